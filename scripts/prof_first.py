"""The first layer's forward and weight gradient at the bench shape, a few launches (for rocprofv3 --pmc passes: scripts/pmc.sh TAG scripts/prof_first.py)."""
import sys, torch
sys.path.insert(0, '.')
from brats21_amd import ops
dev = torch.device('cuda:0'); dt = torch.bfloat16
x = torch.zeros(2, 128, 128, 128, 8, device=dev, dtype=dt); x[..., :4] = torch.randn(2, 128, 128, 128, 4, device=dev).to(dt)
dy = (torch.randn(2, 128, 128, 128, 48, device=dev) * 0.1).to(dt)
w = torch.randn(48, 8, 3, 3, 3, device=dev) * 0.1
wpk = ops.pack_weights(w, dt, ops.PACK_FWD)
for _ in range(4):
    ops.conv3d(x, wpk, 48, 3, 1, want_stats=True)
    ops.conv3d_wgrad(x, dy, 3, 1)
torch.cuda.synchronize()
