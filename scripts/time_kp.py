"""The 16^3 / 32^3-level 3x3x3 launches of EquiUnet-48 (2 patches): 4-wave workgroups against the 8-wave K-parity form (ops.set_kp),
event-timed back to back in one process; python scripts/time_kp.py [reps]"""
import sys, torch
sys.path.insert(0, '.')
from brats21_amd import ops
dev = torch.device('cuda:0'); dt = torch.bfloat16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
shapes = [(384, 0, 384, 1, 16), (384, 0, 384, 2, 16), (192, 0, 384, 1, 16), (384, 384, 192, 1, 16), (384, 0, 192, 1, 16),
          (192, 0, 768, 1, 16), (192, 0, 192, 1, 32), (192, 0, 96, 1, 32), (96, 0, 192, 1, 32), (384, 0, 192, 1, 32)]
for cin, cin2, cout, dil, s in shapes:
    x = torch.randn(2, s, s, s, cin, device=dev).to(dt)
    x2 = torch.randn(2, s, s, s, cin2, device=dev).to(dt) if cin2 else None
    w = torch.randn(cout, cin + cin2, 3, 3, 3, device=dev) * 0.02
    wpk = ops.pack_weights(w, dt, ops.PACK_FWD, dil=dil, c1=cin if cin2 else None)
    fl = 2.0 * (cin + cin2) * 27 * cout * 2 * s ** 3
    out = []
    for mode in (0, 1):
        ops.set_kp(mode)
        fn = lambda: ops.conv3d(x, wpk, cout, 3, dil, want_stats=True, x2=x2)
        for _ in range(5): fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        out.append(f"{'KP8' if mode else '4w '} {ms*1e3:7.1f} us {fl/ms/1e9:6.0f} TF/s")
    ops.set_kp(-1)
    print(f"{cin+cin2:4d}->{cout:3d} d{dil} @{s}^3: " + "   ".join(out))
