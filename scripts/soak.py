"""Longer training run on one fixed synthetic batch (EquiUnet-48, 2 x 4x128^3, fused Dice, Ranger2020, bf16): the loss must fall
and stay finite; prints the trajectory and the step time.  python scripts/soak.py [steps]"""
import argparse, contextlib, io, sys, time
import torch
sys.path.insert(0, '.')
from brats21_amd import get_model, synth
from brats21_amd.engine import TrainStep
from brats21_amd.optim import Ranger2020
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device('cuda:0')
torch.manual_seed(0)
ns = argparse.Namespace(model="equiunet", width=48, norm="group", act="relu", num_classes=3, dropout=0)
with contextlib.redirect_stdout(io.StringIO()):
    model = get_model(ns).to(dev).train()
    opt = Ranger2020(model.parameters(), lr=1e-3, alpha=0.5, k=6, N_sma_threshhold=5, betas=(.95, 0.999), eps=1e-5, weight_decay=1e-5)
x = synth.random_image(2, 4, (128,) * 3, seed=1234, device=dev)
t = synth.nested_spheres(2, (128,) * 3, device=dev)
step = TrainStep(model, opt, criterion=None, amp=True)
losses = []
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(steps):
    losses.append(step(x, t))
torch.cuda.synchronize(); dt = time.perf_counter() - t0
losses = [float(l.item()) for l in losses]
assert all(l == l and abs(l) < 1e3 for l in losses), "non-finite loss"
print(f"{steps} steps, {dt / steps * 1e3:.2f} ms/step; loss " + " ".join(f"{losses[i]:.3f}" for i in range(0, steps, max(1, steps // 10))) + f" ... {losses[-1]:.3f}")
assert losses[-1] < 0.5 * losses[0], "the loss did not fall"
print("params finite:", all(bool(torch.isfinite(p).all()) for p in model.parameters()))
