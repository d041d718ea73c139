"""Which probe tracks the box?  All four MFMA probe modes, the stream probe and the dominant convolution (48 -> 48 at
2 x 128^3, bf16, post-ReLU random data) on one lease; run on several leases and compare the ratios.
  python scripts/box_calib.py"""
import json, sys, torch
sys.path.insert(0, '.')
from brats21_amd import ops
dev = torch.device('cuda:0')
x = torch.relu(torch.randn(2, 128, 128, 128, 48, device=dev)).bfloat16()
w = torch.randn(48, 48, 3, 3, 3, device=dev) * 0.05
wpk = ops.pack_weights(w, torch.bfloat16, ops.PACK_FWD)
dy = torch.randn(2, 128, 128, 128, 48, device=dev).bfloat16()
def t(fn, reps=20):
    for _ in range(5): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
rec = {}
for rep in range(3):
    box = ops.probe_box(dev, modes=(0, 1, 2, 3))
    box.pop("probe")
    box["conv48_ms"] = round(t(lambda: ops.conv3d(x, wpk, 48, 3, 1, want_stats=True)), 4)
    box["wgrad48_ms"] = round(t(lambda: ops.conv3d_wgrad(x, dy, 3, 1)), 4)
    print(json.dumps(box), flush=True)
