"""Per-phase cycle shares of the fused split-precision weight-gradient kernel (csrc/conv_wgrad_x3.hpp) from s_memtime stamps.
Diagnostic build:  scripts/build_variant.sh x3stamps conv_wgrad "-DBRATS_X3W_STAMPS"   then on the GPU box:
  BRATS_HIP_LIB=$PWD/brats21_amd/libbrats_x3stamps.so python scripts/probes/x3w_stamps.py [cin cout size]"""
import sys
sys.path.insert(0, '.')
import torch
from brats21_amd import _lib, ops
lib = _lib.lib()
dev = torch.device("cuda:0")
cin, cout, s = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (48, 48, 128)))
N = 2
x = torch.relu(torch.randn(N, s, s, s, cin, device=dev))
dy = torch.randn(N, s, s, s, cout, device=dev) * 1e-3
amax = ops.absmax(dy)
code = _lib.X3_F16
nbytes = lib.brats_conv3d_wgrad_ws_bytes(code, 3, N, s, s, s, cin, 0, cout)
ws = torch.zeros(nbytes // 4 + 1024, dtype=torch.float32, device=dev)
dw = torch.empty(cout, cin, 27, dtype=torch.float32, device=dev)
for _ in range(3):
    _lib.check(lib.brats_conv3d_x3_wgrad(x.data_ptr(), cin, cin, None, 0, 0, dy.data_ptr(), cout, amax.data_ptr(), ws.data_ptr(), dw.data_ptr(), None,
                                         code, 1, N, s, s, s, cout, torch.cuda.current_stream().cuda_stream), "x3_wgrad")
torch.cuda.synchronize()
nsplit = int(sys.argv[4]) if len(sys.argv) > 4 else 256 // max(1, (cin // 48) * (cout // 48))
off = nsplit * 27 * cout * cin * 4
st = ws.view(torch.uint8)[off: off + nsplit * 8 * 6 * 8].view(torch.int64).view(nsplit, 8, 6).double().cpu()
tot = st.sum(-1, keepdim=True)
names = ["wait vmcnt(0) (loads landing)", "barrier 1", "convert + barrier 2", "issue next tile's loads", "MFMA phase (+ loop)", "segment prologue"]
print(f"x3 wgrad {cin}->{cout} @{s}^3: {nsplit} workgroups, mean cycles per wave {float(tot.mean()):.0f} (100 MHz s_memtime ticks)")
for i, n in enumerate(names):
    print(f"  {n:34s} {100 * float((st[..., i] / tot[..., 0]).mean()):5.1f} %   (mean {float(st[..., i].mean()):9.0f})  by wave: " +
          " ".join(f"{float(st[:, w, i].mean()):7.0f}" for w in range(8)))
