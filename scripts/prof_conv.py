"""Runs the dominant conv kernels at their bench shapes a few times (for rocprofv3 --pmc passes)."""
import sys, torch
sys.path.insert(0, '.')
from brats21_amd import ops
dev = torch.device('cuda:0')
dt = torch.bfloat16
which = sys.argv[1] if len(sys.argv) > 1 else "all"
N, S = 2, 128
def run(cin, cout, s, dil=1, reps=3):
    x = torch.relu(torch.randn(N, s, s, s, cin, device=dev)).to(dt)  # post-ReLU activations, as in the network
    dy = torch.randn(N, s, s, s, cout, device=dev).to(dt)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    wpk = ops.pack_weights(w, dt, ops.PACK_FWD, dil=dil)
    for _ in range(reps):
        if which in ("all", "fwd"):
            ops.conv3d(x, wpk, cout, 3, dil, want_stats=True)
        if which in ("all", "wgrad"):
            ops.conv3d_wgrad(x, dy, 3, dil)
    torch.cuda.synchronize()
if which == "dom":  # only the kernel bench.py's roofline names: forward 48 -> 48 @ 2 x 128^3
    which = "fwd"
    run(48, 48, 128)
else:
    run(48, 48, 128)
    run(96, 48, 128)
    run(192, 96, 64)
    run(384, 384, 16, 2)
