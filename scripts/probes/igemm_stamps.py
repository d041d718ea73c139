"""Per-phase cycle shares of conv_igemm_kernel at a small-grid shape (diagnostic library built by scripts/probes/stamps_build.sh):
   python scripts/probes/igemm_stamps.py cin cout size [dil] [kp]"""
import os, sys
sys.path.insert(0, '.')
os.environ.setdefault("BRATS_HIP_LIB", os.path.abspath("brats21_amd/libbrats_diag.so"))
import torch
cin, cout, s = (int(a) for a in sys.argv[1:4])
dil = int(sys.argv[4]) if len(sys.argv) > 4 else 1
kp = int(sys.argv[5]) if len(sys.argv) > 5 else 1
dev = torch.device("cuda:0")
N = 2
nw = 8 if kp else 4
nblk = N * (s // 4) * (s // 4) * (s // 16) * (cout // 48)
stamps = torch.zeros(nblk * nw * 6, dtype=torch.int64, device=dev)
os.environ["BRATS_VS8_STAMP_PTR"] = str(stamps.data_ptr())
from brats21_amd import ops
ops.set_kp(kp)
x = torch.relu(torch.randn(N, s, s, s, cin, device=dev)).to(torch.bfloat16)
w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
wpk = ops.pack_weights(w, torch.bfloat16, ops.PACK_FWD, dil=dil)
for _ in range(5):
    ops.conv3d(x, wpk, cout, 3, dil, want_stats=True)
torch.cuda.synchronize()
st = stamps.view(nblk, nw, 6).double().cpu()
tot = st.sum(-1, keepdim=True)
names = ["address arithmetic + issue of halo loads", "barrier (previous chunk done)", "halo loads landing (vmcnt 0)", "LDS writes + barrier", "MFMA loop", "reduction + epilogue"]
print(f"conv_igemm {cin}->{cout} d{dil} @{s}^3, {nw} waves: {nblk} workgroups, mean cycles per wave {float(tot.mean()):.0f} (max {float(tot.max()):.0f})")
for i, n in enumerate(names):
    print(f"  {n:42s} {100 * float((st[..., i] / tot[..., 0]).mean()):5.1f} %   (mean {float(st[..., i].mean()):9.0f} cycles)")
