"""Per-phase cycle shares of conv_igemm_vs8_kernel (diagnostic library built by scripts/probes/stamps_build.sh)."""
import os, sys
sys.path.insert(0, '.')
os.environ.setdefault("BRATS_HIP_LIB", os.path.abspath("brats21_amd/libbrats_diag.so"))
import torch
cin, cout, s = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (48, 48, 128)))
dev = torch.device("cuda:0")
N = 2
nblk = N * (s // 4) * (s // 8) * (s // 16) * (cout // 48)
stamps = torch.zeros(nblk * 4 * 6, dtype=torch.int64, device=dev)
os.environ["BRATS_VS8_STAMP_PTR"] = str(stamps.data_ptr())
from brats21_amd import ops
x = torch.relu(torch.randn(N, s, s, s, cin, device=dev)).to(torch.bfloat16)
w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
wpk = ops.pack_weights(w, torch.bfloat16, ops.PACK_FWD)
for _ in range(3):
    ops.conv3d(x, wpk, cout, 3, 1, want_stats=True)
torch.cuda.synchronize()
st = stamps.view(nblk, 4, 6).double().cpu()
tot = st.sum(-1, keepdim=True)
names = ["prologue + issue of halo loads", "barrier (previous chunk done)", "halo loads landing (vmcnt 0)", "LDS writes + barrier", "MFMA loop", "epilogue"]
print(f"conv_igemm_vs8 {cin}->{cout} @{s}^3: {nblk} workgroups, mean cycles per wave {float(tot.mean()):.0f}")
for i, n in enumerate(names):
    print(f"  {n:34s} {100 * float((st[..., i] / tot[..., 0]).mean()):5.1f} %   (mean {float(st[..., i].mean()):9.0f} cycles)")
