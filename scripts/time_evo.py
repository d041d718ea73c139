"""Event-timed EvoNorm / SE passes at a bench shape: python scripts/time_evo.py C size"""
import sys, torch
sys.path.insert(0, '.')
from brats21_amd import ops
dev = torch.device('cuda:0'); dt = torch.bfloat16
C, s = int(sys.argv[1]), int(sys.argv[2])
N = 2
y = torch.randn(N, s, s, s, C, device=dev).to(dt)
dz = torch.randn(N, s, s, s, C, device=dev).to(dt)
mr = torch.rand(N, 8, 2, device=dev) + 0.5
gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
sc = torch.rand(N, C, device=dev) + 0.5
nbytes = y.numel() * 2
def t(name, fn, traffic):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print(f"{name:14s} C={C} @{s}^3: {ms:.3f} ms  {traffic * nbytes / ms / 1e9:.2f} TB/s")
t("evonorm", lambda: ops.evonorm(y, mr, gamma, beta, 8, want_chansum=True), 2)
t("evonorm_bwd", lambda: ops.evonorm_bwd(dz, y, mr, gamma, 8), 5)
t("channel_scale", lambda: ops.channel_scale(y, sc), 2)
t("channel_dot", lambda: ops.channel_dot(dz, y), 2)
