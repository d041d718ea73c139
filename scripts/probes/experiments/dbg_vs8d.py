import sys, torch
sys.path.insert(0, ".")
from brats21_amd import ops
dev = torch.device("cuda:0")
c = 48
g = torch.Generator().manual_seed(0)
def run(x, w, mode):
    old = ops.set_vs8(mode)
    try:
        wpk = ops.pack_weights(w.to(dev), torch.bfloat16, ops.PACK_FWD)
        y, _ = ops.conv3d(x, wpk, c, 3, 1)
        torch.cuda.synchronize()
        return y.float()
    finally:
        ops.set_vs8(old)
w = torch.zeros(c, c, 3, 3, 3)
for i in range(c):
    w[i, i, 1, 1, 1] = 1.0
for size in ((8, 16, 32), (16, 24, 48), (12, 24, 48)):
    x = (torch.randn(1, *size, c, generator=g).abs() + 0.5).to(torch.bfloat16).to(dev)
    for rep in range(2):
        y3 = run(x, w, 3)
        d = (y3 - x.float()).abs()
        bad = (d.amax(dim=(0, 4)) > 0).nonzero()
        print(size, "rep", rep, "identity: bad voxels", bad.shape[0], "of", d[0, ..., 0].numel())
        if bad.shape[0]:
            lo, hi = bad.min(0).values.tolist(), bad.max(0).values.tolist()
            bc = (d.amax(dim=(0, 1, 2, 3)) > 0).nonzero().flatten().tolist()
            z, yv, xv = bad[0].tolist()
            print("   box", lo, hi, "channels", bc[:3], "..", bc[-1], "n", len(bc), "| y3 there", y3[0, z, yv, xv, bc[0]].item(), "x", float(x[0, z, yv, xv, bc[0]]))
            tiles = sorted({(int(b[0]) // 4, int(b[1]) // 8, int(b[2]) // 16) for b in bad.tolist()})
            print("   tiles (z/4, y/8, x/16):", tiles)
