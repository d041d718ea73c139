"""Same-process, interleaved A/B of the Cout = 48 forward kernels (cdna_hip_programming.md rule 24): mode 1 = conv_igemm_vs8
(24-channel chunks), mode 2 + 16 v = conv_igemm_ld variant v (loader wave, 16-channel chunks).
  python scripts/time_ld.py [rounds] [reps]"""
import sys, statistics, torch
sys.path.insert(0, '.')
from brats21_amd import ops
dev = torch.device('cuda:0'); dt = torch.bfloat16
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
configs = [("vs8", 1), ("ld v0", 2), ("ld v1", 2 + 16), ("ld v2", 2 + 32), ("ld v3", 2 + 48)]
N, s = 2, 128
for cin, cin2, cout in [(48, 0, 48), (48, 48, 48)]:
    x = torch.relu(torch.randn(N, s, s, s, cin, device=dev)).to(dt)
    x2 = torch.relu(torch.randn(N, s, s, s, cin2, device=dev)).to(dt) if cin2 else None
    w = torch.randn(cout, cin + cin2, 3, 3, 3, device=dev) * 0.05
    fl = 2.0 * (cin + cin2) * 27 * cout * N * s ** 3
    res = {name: [] for name, _ in configs}
    ref = None
    for r in range(rounds):
        for name, mode in configs:
            old = ops.set_vs8(mode)
            wpk = ops.pack_weights(w, dt, ops.PACK_FWD, c1=cin if cin2 else None)
            fn = lambda: ops.conv3d(x, wpk, cout, 3, 1, want_stats=True, x2=x2)
            y, st = fn()
            if r == 0:
                if ref is None:
                    ref = (y.float(), st.sum(1))
                else:
                    e = float((y.float() - ref[0]).abs().max()); es = float((st.sum(1) - ref[1]).abs().max() / ref[1].abs().max())
                    print(f"  {name}: max |y - y_vs8| {e:.3e} (|y| max {float(ref[0].abs().max()):.2f}), stats rel {es:.2e}")
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps): fn()
            b.record(); torch.cuda.synchronize()
            res[name].append(a.elapsed_time(b) / reps)
            ops.set_vs8(old)
    for name, _ in configs:
        v = res[name]
        print(f"{cin}+{cin2}->{cout} @{N}x{s}^3 {name:6s}: median {statistics.median(v):.4f} ms  min {min(v):.4f} ms  {fl / statistics.median(v) / 1e9:.0f} TF/s")
