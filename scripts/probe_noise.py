"""Run-to-run noise of the box probe by duration (after a burst of real training steps): python scripts/probe_noise.py"""
import argparse, contextlib, io, sys, torch
sys.path.insert(0, '.')
from brats21_amd import ops, get_model
from brats21_amd.engine import TrainStep
from brats21_amd.optim import Ranger2020
from brats21_amd import synth
dev = torch.device('cuda:0')
with contextlib.redirect_stdout(io.StringIO()):
    m = get_model(argparse.Namespace(model="equiunet", width=48, norm="group", act="relu", num_classes=3, dropout=0)).to(dev).train()
    opt = Ranger2020(m.parameters(), lr=1e-4)
x = synth.random_image(2, 4, (128,) * 3, seed=1234, device=dev); t = synth.nested_spheres(2, (128,) * 3, device=dev)
step = TrainStep(m, opt, criterion=None, amp=True)
for ms in (12, 50, 100):
    vals = []
    for rep in range(6):
        for _ in range(10): step(x, t)
        torch.cuda.synchronize()
        b = ops.probe_box(dev, mfma_ms=ms, modes=(0, 1))
        vals.append((b["mfma_TFLOPs"], b["mfma_zeros_TFLOPs"], b["stream_TBps"]))
    print(ms, "ms:", vals, flush=True)
