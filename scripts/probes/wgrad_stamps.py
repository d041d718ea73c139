"""Per-phase cycle shares of the all-taps weight-gradient kernel (diagnostic library built by wgrad_stamps.sh)."""
import os, sys
sys.path.insert(0, '.')
os.environ["BRATS_HIP_LIB"] = os.path.abspath("brats21_amd/libbrats_hip_stamps.so")
import torch
from brats21_amd import _lib, ops
lib = _lib.lib()
dev = torch.device("cuda:0")
cin, cout, s = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (48, 48, 128)))
N = 2
x = torch.relu(torch.randn(N, s, s, s, cin, device=dev)).to(torch.bfloat16)
dy = torch.randn(N, s, s, s, cout, device=dev).to(torch.bfloat16)
nbytes = lib.brats_conv3d_wgrad_ws_bytes(1, 3, N, s, s, s, cin, 0, cout)
extra = 4096 * 8 * 5 * 8
ws = torch.zeros((nbytes + extra) // 4, dtype=torch.float32, device=dev)
dw = torch.empty(cout, cin, 27, dtype=torch.float32, device=dev)
for _ in range(3):
    _lib.check(lib.brats_conv3d_wgrad(x.data_ptr(), cin, cin, None, 0, 0, dy.data_ptr(), cout, ws.data_ptr(), dw.data_ptr(), None, 1, 3, 1,
                                      N, s, s, s, cout, torch.cuda.current_stream().cuda_stream), "wgrad")
torch.cuda.synchronize()
nsplit = nbytes // (27 * cout * cin * 4)
st = ws.view(torch.int64)[nbytes // 8: nbytes // 8 + nsplit * 8 * 5].view(nsplit, 8, 5).double().cpu()
tot = st.sum(-1, keepdim=True)
names = ["wait vmcnt(0) (loads landing)", "barrier 1", "dY write + barrier 2", "issue next tile's loads", "MFMA phase (+ loop)"]
print(f"wgrad all-taps {cin}->{cout} @{s}^3: {nsplit} workgroups, mean cycles per wave {float(tot.mean()):.0f}")
for i, n in enumerate(names):
    print(f"  {n:34s} {100 * float((st[..., i] / tot[..., 0]).mean()):5.1f} %   (mean {float(st[..., i].mean()):9.0f} cycles)")
