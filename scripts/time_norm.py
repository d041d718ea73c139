"""Event-timed GroupNorm passes at a bench shape: python scripts/time_norm.py C size"""
import sys, torch
sys.path.insert(0, '.')
from brats21_amd import ops
dev = torch.device('cuda:0'); dt = torch.bfloat16
C, s = int(sys.argv[1]), int(sys.argv[2])
N = 2
y = torch.randn(N, s, s, s, C, device=dev).to(dt)
dz = torch.randn(N, s, s, s, C, device=dev).to(dt)
ss = torch.rand(N, C, 2, device=dev)
mr = torch.rand(N, 8, 2, device=dev) + 0.5
gamma = torch.ones(C, device=dev)
nbytes = y.numel() * 2
def t(name, fn, traffic):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print(f"{name:12s} C={C} @{s}^3: {ms:.3f} ms  {traffic * nbytes / ms / 1e9:.2f} TB/s")
t("affine_act", lambda: ops.affine_act(y, ss, "relu"), 2)
t("gn_act_bwd", lambda: ops.gn_act_bwd(dz, y, ss, mr, gamma, 8, "relu"), 5)
t("maxpool2", lambda: ops.maxpool2(y), 1.125)
t("upsample", lambda: ops.upsample(y[:, :s // 2, :s // 2, :s // 2].contiguous(), 2), 1.125)
