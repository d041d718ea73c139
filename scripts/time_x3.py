"""Event-timed split-precision ("x3") convolution kernels beside the exact-f32 and bf16 kernels of the same layer, then a
whole EquiUnet-48 training step per precision:  python scripts/time_x3.py [size]"""
import argparse, contextlib, io, sys, time, torch
sys.path.insert(0, '.')
from brats21_amd import ops, get_model
dev = torch.device('cuda:0')
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = 2


def timeit(fn, reps=5):
    for _ in range(2): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for cin, cout, s, dil in ((48, 48, S, 1), (96, 48, S, 1), (96, 96, S // 2, 1), (192, 192, S // 4, 1), (384, 384, S // 8, 2), (8, 48, S, 1)):
    x = torch.relu(torch.randn(N, s, s, s, cin, device=dev))
    dy = torch.randn(N, s, s, s, cout, device=dev) * 1e-3
    w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
    fl = 2.0 * cin * 27 * cout * N * s ** 3
    row = []
    for label, mode, dt in (("f32", None, torch.float32), ("x3f", ops.X3F, torch.float32), ("x3b", ops.X3B, torch.float32), ("bf16", None, torch.bfloat16)):
        xx, dd = x.to(dt), dy.to(dt)
        with ops.split_precision(mode):
            wpk = ops.pack_weights(w, dt, ops.PACK_FWD, dil=dil)
            t_f = timeit(lambda: ops.conv3d(xx, wpk, cout, 3, dil, want_stats=True))
            t_w = timeit(lambda: ops.conv3d_wgrad(xx, dd, 3, dil)) if cin % 8 == 0 else float("nan")
        row.append(f"{label}: fwd {t_f:.3f} ms ({fl / t_f / 1e9:.0f} TF/s) wgrad {t_w:.3f} ms ({fl / t_w / 1e9:.0f} TF/s)")
    print(f"{cin}->{cout} @{N}x{s}^3 d={dil}: " + " | ".join(row), flush=True)

# whole training step
from brats21_amd.engine import TrainStep
from brats21_amd.optim import Ranger2020
from oracle import synth
torch.manual_seed(0)
with contextlib.redirect_stdout(io.StringIO()):
    m = get_model(argparse.Namespace(model="equiunet", width=48, norm="group", act="relu", num_classes=3, dropout=0)).to(dev).train()
    opt = Ranger2020(m.parameters(), lr=1e-4)
x = synth.random_image(N, 4, (S, S, S), seed=1234).to(dev)
t = synth.nested_spheres(N, (S, S, S)).to(dev)
step = TrainStep(m, opt, criterion=None, amp=False)
for prec in ("x3", "bf16x3", "fp32"):
    m.precision = prec
    for _ in range(2): step(x, t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): step(x, t)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"EquiUnet-48 {N}x4x{S}^3 training step, precision={prec}: {ms:.2f} ms = {N / ms * 1e3:.1f} patches/s", flush=True)
