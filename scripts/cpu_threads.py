"""Fastest CPU thread count for the oracle baseline on this host (bench.py cpu_baseline uses min(16, cores))."""
import sys, time, torch
sys.path.insert(0, '.')
from oracle import synth, unet
size = (32, 32, 32)
sd = {k: v.requires_grad_(True) for k, v in synth.fill_state_dict(unet.equiunet_state_shapes(48)).items()}
x, t = synth.random_image(1, 4, size), synth.nested_spheres(1, size)
for th in (8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    ts = []
    for it in range(3):
        t0 = time.perf_counter()
        loss = unet.deep_supervision_loss(unet.equiunet_forward(sd, x), t); loss.backward()
        for v in sd.values(): v.grad = None
        ts.append(time.perf_counter() - t0)
    print(th, "threads:", [round(a, 2) for a in ts], flush=True)
