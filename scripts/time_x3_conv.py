"""x3 forward kernel timings for the big layers: python scripts/time_x3_conv.py"""
import sys, torch
sys.path.insert(0, '.')
from brats21_amd import ops
dev = torch.device('cuda:0'); N = 2
def timeit(fn, reps=10):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for cin, cout, s in ((48, 48, 128), (96, 48, 128), (48, 96, 128), (96, 96, 64), (192, 192, 32)):
    x = torch.relu(torch.randn(N, s, s, s, cin, device=dev))
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    fl = 2.0 * cin * 27 * cout * N * s ** 3
    with ops.split_precision(ops.X3F):
        wpk = ops.pack_weights(w, torch.float32, ops.PACK_FWD)
        t = timeit(lambda: ops.conv3d(x, wpk, cout, 3, 1, want_stats=True))
    print(f"{cin}->{cout} @{s}^3: x3f fwd {t:.3f} ms ({fl / t / 1e9:.0f} TF/s eq, {3 * fl / t / 1e9:.0f} TF/s of MFMA work) chunk {ops.conv_chunk(ops.X3F, 3, 1, cin, 0, cout)}", flush=True)
