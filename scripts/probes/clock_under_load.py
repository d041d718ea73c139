"""Does the chip give an in-kernel cycle saving back as clock?  One launch shape in the 4-wave and in the 8-wave K-parity form,
~4 s of back-to-back launches each, with the SMI's shader clock and socket power sampled beside it (rocm-smi, 5 Hz):
   python scripts/probes/clock_under_load.py [cin cout size dil]"""
import json, subprocess, sys, threading, time
import torch
sys.path.insert(0, '.')
from brats21_amd import ops
cin, cout, s, dil = (int(a) for a in (sys.argv[1:5] if len(sys.argv) > 4 else (384, 384, 16, 1)))
dev = torch.device("cuda:0"); dt = torch.bfloat16
x = torch.relu(torch.randn(2, s, s, s, cin, device=dev)).to(dt)
w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.02
wpk = ops.pack_weights(w, dt, ops.PACK_FWD, dil=dil)
out = torch.empty(2, s, s, s, cout, device=dev, dtype=dt)
samples, stop = [], threading.Event()
def sample():
    while not stop.is_set():
        try:
            r = json.loads(subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout)
            c = r[sorted(r)[0]]
            sclk = next((v for k, v in c.items() if "sclk" in k.lower()), "?")
            pw = next((v for k, v in c.items() if "power" in k.lower()), "?")
            samples.append((time.perf_counter(), sclk, pw))
        except Exception as e:
            samples.append((time.perf_counter(), "err " + repr(e)[:60], ""))
        time.sleep(0.2)
th = threading.Thread(target=sample, daemon=True); th.start()
for mode in (0, 1, 0, 1):
    ops.set_kp(mode)
    for _ in range(50): ops.conv3d(x, wpk, cout, 3, dil, want_stats=True, out=out)
    torch.cuda.synchronize()
    n = int(sys.argv[5]) if len(sys.argv) > 5 else 40000
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); a.record()
    for _ in range(n): ops.conv3d(x, wpk, cout, 3, dil, want_stats=True, out=out)
    b.record(); torch.cuda.synchronize(); t1 = time.perf_counter()
    mine = [(c, p) for (t, c, p) in samples if t0 + 0.5 < t < t1]
    print(f"{'8 waves (K-parity)' if mode else '4 waves          '}: {a.elapsed_time(b) / n * 1e3:6.1f} us / launch over {t1 - t0:.1f} s; SMI samples (sclk, power): {mine[:6]}")
stop.set()
