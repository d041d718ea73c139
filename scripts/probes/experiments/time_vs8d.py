"""The dominant layers on conv_igemm_vs8 (mode 1) and on the double-buffered LDS-DMA experiment conv_igemm_vs8d (mode 3), alternating
in one process: python scripts/time_vs8d.py"""
import sys
import torch
sys.path.insert(0, ".")
from brats21_amd import ops

dev = torch.device("cuda:0")
n, s = 2, 128
g = torch.Generator().manual_seed(0)


def rnd(c):
    return torch.relu(torch.randn(n, s, s, s, c, generator=g)).to(torch.bfloat16).to(dev)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for cin, cin2, cout in ((48, 0, 48), (48, 48, 48)):
    x, x2 = rnd(cin), (rnd(cin2) if cin2 else None)
    w = (torch.randn(cout, cin + cin2, 3, 3, 3, generator=g) * 0.03).to(dev)
    res, outs = {1: [], 3: [], 4: []}, {}
    for rep in range(4):
        for mode in (1, 3, 4):
            old = ops.set_vs8(mode)
            try:
                wpk = ops.pack_weights(w, torch.bfloat16, ops.PACK_FWD, c1=cin if cin2 else None)
                y = ops.new_act(n, s, s, s, cout, torch.bfloat16, dev)
                res[mode].append(timed(lambda: ops.conv3d(x, wpk, cout, 3, 1, out=y, want_stats=True, x2=x2)))
                outs[mode] = y.float()
            finally:
                ops.set_vs8(old)
    err = max(float((outs[1] - outs[m]).abs().max() / outs[1].abs().max()) for m in (3, 4))
    print(f"{cin}+{cin2}->{cout} @2x128^3 with statistics: vs8 {['%.4f' % t for t in res[1]]} ms | vs8d (LDS-DMA) {['%.4f' % t for t in res[3]]} | "
          f"vs8d (register staging) {['%.4f' % t for t in res[4]]} | max rel diff {err:.2e}")
