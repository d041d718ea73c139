"""First layer's weight gradient (8 -> 48 @ 2 x 128^3), event-timed; BRATS_WGRAD_ALLTAPS=1 selects the register-staging form"""
import sys, torch
sys.path.insert(0, '.')
from brats21_amd import ops
dev = torch.device('cuda:0'); dt = torch.bfloat16
x = torch.zeros(2, 128, 128, 128, 8, device=dev, dtype=dt); x[..., :4] = torch.randn(2, 128, 128, 128, 4, device=dev).to(dt)
dy = (torch.randn(2, 128, 128, 128, 48, device=dev) * 0.1).to(dt)
for _ in range(3): ops.conv3d_wgrad(x, dy, 3, 1)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): ops.conv3d_wgrad(x, dy, 3, 1)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 20
print(f"first-layer wgrad: {ms:.4f} ms = {2*128**3*(48+8)*2/ms/1e9:.2f} TB/s algorithmic")
