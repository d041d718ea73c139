"""Host enqueue time vs wall time of a training step (is the eager step launch-bound on this host?), with and without
the gradient-bucket hooks of the data-parallel path."""
import sys, time, argparse, contextlib, io, torch
sys.path.insert(0, '.')
from brats21_amd import get_model, synth
from brats21_amd.engine import TrainStep
from brats21_amd.optim import Ranger2020
from brats21_amd.ddp import GradientBuckets
dev = torch.device("cuda:0")
for model in ("equiunet", "equiunet_assp_evo"):
    ns = argparse.Namespace(model=model, width=48, norm="group", act="relu", num_classes=3, dropout=0)
    with contextlib.redirect_stdout(io.StringIO()):
        m = get_model(ns).to(dev).train()
        opt = Ranger2020(m.parameters(), lr=1e-4, weight_decay=1e-5, use_gc=False)
    for ddp in (False, True):
        buckets = GradientBuckets(m) if ddp else None
        if not ddp and hasattr(m, "_grad_sink"): m._grad_sink = None
        step = TrainStep(m, opt, amp=True, buckets=buckets)
        x = synth.random_image(2, 4, (128,)*3, seed=1, device=dev); t = synth.nested_spheres(2, (128,)*3, device=dev)
        for _ in range(3): step(x, t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): step(x, t)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{model} ddp_buckets={ddp}: host enqueue {1e3*(t1-t0)/10:.2f} ms/step, wall {1e3*(t2-t0)/10:.2f} ms/step")
    del m, opt
