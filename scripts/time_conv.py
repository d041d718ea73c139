"""Event-timed single conv shapes: python scripts/time_conv.py cin cout size [dil] [reps]"""
import sys, torch
sys.path.insert(0, '.')
from brats21_amd import ops
dev = torch.device('cuda:0'); dt = torch.bfloat16
cin, cout, s = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dil = int(sys.argv[4]) if len(sys.argv) > 4 else 1
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
N = 2
x = torch.randn(N, s, s, s, cin, device=dev).to(dt)
dy = torch.randn(N, s, s, s, cout, device=dev).to(dt)
w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
wpk = ops.pack_weights(w, dt, ops.PACK_FWD, dil=dil)
fl = 2.0 * cin * 27 * cout * N * s ** 3
cases = [("fwd", lambda: ops.conv3d(x, wpk, cout, 3, dil, want_stats=True)), ("wgrad", lambda: ops.conv3d_wgrad(x, dy, 3, dil))]
if ops.conv_f8_chunk(cin) > 0:  # the e4m3 kernel of the same layer (scale source already on the device)
    wpk8 = ops.pack_weights_f8(w, ops.PACK_FWD)
    amax = ops.absmax(x)
    cases.insert(1, ("fwd_f8", lambda: ops.conv3d_f8(x, wpk8, cout, dil, want_stats=True, amax=amax)))
for name, fn in cases:
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    print(f"{name:6s} cin={cin} cout={cout} @{s}^3 d={dil}: {ms:.3f} ms  {fl / ms / 1e9:.0f} TF/s")
