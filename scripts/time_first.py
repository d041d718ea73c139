"""First layer (8 padded channels -> 48 @ 2 x 128^3): the persistent kernel (dense output) against the 4x8x16-tile kernel (the same
call with a channel-slice output), event-timed; python scripts/time_first.py [reps]"""
import sys, torch
sys.path.insert(0, '.')
from brats21_amd import ops
dev = torch.device('cuda:0'); reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for dt in (torch.bfloat16, torch.float16):
    x = torch.zeros(2, 128, 128, 128, 8, device=dev, dtype=dt); x[..., :4] = torch.randn(2, 128, 128, 128, 4, device=dev).to(dt)
    w = torch.randn(48, 8, 3, 3, 3, device=dev) * 0.1
    wpk = ops.pack_weights(w, dt, ops.PACK_FWD)
    wide = torch.zeros(2, 128, 128, 128, 96, dtype=dt, device=dev)
    dense = torch.zeros(2, 128, 128, 128, 48, dtype=dt, device=dev)
    cases = [("persistent (dense out)", lambda: ops.conv3d(x, wpk, 48, 3, 1, want_stats=True, out=dense)),
             ("tile kernel (slice out)", lambda: ops.conv3d(x, wpk, 48, 3, 1, want_stats=True, out=wide[..., :48]))]
    for name, fn in cases:
        for _ in range(3): fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        print(f"{str(dt):15s} {name:24s}: {ms:.4f} ms  = {(2*128**3*(48+8)*2)/ms/1e9:.2f} TB/s algorithmic (in 8 + out 48 channels)")
