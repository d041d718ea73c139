"""Weight-packing time of one EquiUnet-48 training step: the per-layer launches against the one-launch plan (ops.PackPlan).
python scripts/time_pack.py [model] [width]"""
import argparse, contextlib, io, sys, warnings
import torch
sys.path.insert(0, '.')
from brats21_amd import get_model, ops
from oracle import synth, unet

name = sys.argv[1] if len(sys.argv) > 1 else "equiunet"
width = int(sys.argv[2]) if len(sys.argv) > 2 else 48
dev = torch.device("cuda:0")
ns = argparse.Namespace(model=name, width=width, norm="group", act="relu", num_classes=3, dropout=0)
with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
    warnings.simplefilter("ignore")
    m = get_model(ns).to(dev).train()
m.pack_plan = True
x = synth.random_image(1, 4, (32, 32, 32), seed=1).to(dev)
t = synth.nested_spheres(1, (32, 32, 32)).to(dev)
for _ in range(2):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out, deeps = m(x)
        loss = unet.deep_supervision_loss((out, deeps), t)
    loss.backward()
plan = ops._PLANS[m]
torch.cuda.synchronize()


def timed(fn, n=50):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def per_layer():
    for wref, args in plan.recorded.values():
        ops._pack_weights(wref(), args[0], args[1], args[2], args[3], args[4], args[5], args[6])


nblocks = sum(tb[2].shape[0] for tb in plan.tables)
print(f"{name}-{width}: {len(plan.recorded)} packed layouts, {plan.buf.numel() / 1e6:.1f} MB packed, {nblocks} workgroups in the plan")
print(f"one launch (PackPlan.run): {timed(lambda: plan.run(dev)):.1f} us")
print(f"per-layer launches       : {timed(per_layer):.1f} us (host-bound if the launches are cheaper than their Python)")
