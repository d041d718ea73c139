"""Where does the host time of an eager training step go?  cProfile over 10 steps (after warm-up) of both width-48 models with the
gradient-bucket hooks on: python scripts/host_profile.py [model]"""
import sys, argparse, contextlib, io, cProfile, pstats, time, torch
sys.path.insert(0, '.')
from brats21_amd import get_model, synth
from brats21_amd.engine import TrainStep
from brats21_amd.optim import Ranger2020
from brats21_amd.ddp import GradientBuckets
dev = torch.device("cuda:0")
models = sys.argv[1:] or ["equiunet_assp_evo", "equiunet"]
for model in models:
    ns = argparse.Namespace(model=model, width=48, norm="group", act="relu", num_classes=3, dropout=0)
    with contextlib.redirect_stdout(io.StringIO()):
        m = get_model(ns).to(dev).train()
        opt = Ranger2020(m.parameters(), lr=1e-4, weight_decay=1e-5, use_gc=False)
    step = TrainStep(m, opt, amp=True, buckets=GradientBuckets(m))
    x = synth.random_image(2, 4, (128,) * 3, seed=1, device=dev); t = synth.nested_spheres(2, (128,) * 3, device=dev)
    for _ in range(3): step(x, t)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    for _ in range(3): step(x, t)
    pr.disable()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"== {model}: host {1e3 * (t1 - t0) / 3:.2f} ms/step under cProfile (3 steps on an empty queue)")
    st = pstats.Stats(pr); st.sort_stats("tottime")
    out = io.StringIO(); st.stream = out; st.print_stats(28)
    print("\n".join(l[:150] for l in out.getvalue().splitlines()[4:44]))
    del m, opt, step
