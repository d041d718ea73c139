"""Event-timed split-precision weight gradients of the EquiUnet-48 layers (2 x 128^3 patch), fused kernel vs the three-launch form:
  python scripts/time_x3_wgrad.py            (BRATS_HIP_LIB selects an A/B build)"""
import os, sys, torch
sys.path.insert(0, '.')
from brats21_amd import ops
dev = torch.device('cuda:0')
N, S = 2, 128


def timeit(fn, reps=5):
    for _ in range(2): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


print("lib:", os.environ.get("BRATS_HIP_LIB", "in-tree"))
for c1, c2, cout, s in ((48, 0, 48, S), (48, 48, 48, S), (8, 0, 48, S), (96, 0, 96, S // 2), (96, 96, 96, S // 2), (48, 0, 96, S // 2), (192, 0, 192, S // 4), (192, 192, 192, S // 4), (384, 0, 384, S // 8)):
    x = torch.relu(torch.randn(N, s, s, s, c1 + c2, device=dev))
    x1, x2 = (x[..., :c1], x[..., c1:]) if c2 else (x, None)
    dy = torch.randn(N, s, s, s, cout, device=dev) * 1e-3
    amax = ops.absmax(dy)
    fl = 2.0 * (c1 + c2) * 27 * cout * N * s ** 3
    row = []
    for fused in (0, 1):
        ops.set_x3_wgrad_fused(fused)
        with ops.split_precision(ops.X3F):
            t = timeit(lambda: ops.conv3d_wgrad(x1, dy, 3, 1, x2=x2, amax_dy=amax))
        row.append(f"{'fused' if fused else 'three-launch'} {t:.3f} ms ({fl / t / 1e9:.0f} TF/s)")
    ops.set_x3_wgrad_fused(-1)
    print(f"{c1}+{c2}->{cout} @{N}x{s}^3: " + " | ".join(row), flush=True)
