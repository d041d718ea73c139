"""Loss trajectories of the same training run (EquiUnet-48, 2 x 4x128^3 synthetic patches, fused Dice, Ranger2020, lr 1e-3) in
the exact-f32 mode, the split-precision mode ("x3": f32 tensors, 3 x fp16-pair MFMA convolutions) and bf16 storage:
  python scripts/x3_training_curve.py [steps]"""
import argparse, contextlib, io, sys
import torch
sys.path.insert(0, '.')
from brats21_amd import get_model, synth
from brats21_amd.engine import TrainStep
from brats21_amd.optim import Ranger2020

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device('cuda:0')
x = synth.random_image(2, 4, (128,) * 3, seed=1234, device=dev)
t = synth.nested_spheres(2, (128,) * 3, device=dev)
curves = {}
for prec in ("fp32", "x3", "bf16"):
    torch.manual_seed(0)
    ns = argparse.Namespace(model="equiunet", width=48, norm="group", act="relu", num_classes=3, dropout=0)
    with contextlib.redirect_stdout(io.StringIO()):
        model = get_model(ns).to(dev).train()
        opt = Ranger2020(model.parameters(), lr=1e-3, alpha=0.5, k=6, N_sma_threshhold=5, betas=(.95, 0.999), eps=1e-5, weight_decay=1e-5)
    model.precision = prec
    step = TrainStep(model, opt, criterion=None, amp=False)
    curves[prec] = [float(step(x, t).item()) for _ in range(steps)]
print("step  " + "  ".join(f"{k:>10s}" for k in curves) + "   |x3 - fp32|  |bf16 - fp32|")
for i in list(range(0, steps, max(1, steps // 15))) + [steps - 1]:
    print(f"{i:4d}  " + "  ".join(f"{curves[k][i]:10.6f}" for k in curves) + f"   {abs(curves['x3'][i] - curves['fp32'][i]):.2e}     {abs(curves['bf16'][i] - curves['fp32'][i]):.2e}")
