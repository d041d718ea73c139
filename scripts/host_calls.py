"""Host time of one eager step split into: time inside the C-ABI calls (hipLaunchKernel and friends), per entry point, and the Python
around them.  python scripts/host_calls.py [model]"""
import sys, argparse, contextlib, io, time, collections, torch
sys.path.insert(0, '.')
from brats21_amd import get_model, synth, _lib
from brats21_amd.engine import TrainStep
from brats21_amd.optim import Ranger2020
from brats21_amd.ddp import GradientBuckets
dev = torch.device("cuda:0")
model = sys.argv[1] if len(sys.argv) > 1 else "equiunet_assp_evo"
ns = argparse.Namespace(model=model, width=48, norm="group", act="relu", num_classes=3, dropout=0)
with contextlib.redirect_stdout(io.StringIO()):
    m = get_model(ns).to(dev).train()
    opt = Ranger2020(m.parameters(), lr=1e-4, weight_decay=1e-5, use_gc=False)
step = TrainStep(m, opt, amp=True, buckets=GradientBuckets(m))
x = synth.random_image(2, 4, (128,) * 3, seed=1, device=dev); t = synth.nested_spheres(2, (128,) * 3, device=dev)
for _ in range(3): step(x, t)
torch.cuda.synchronize()
lib = _lib.lib()
acc = collections.defaultdict(lambda: [0, 0.0])
class Wrap:
    def __init__(self, fn, name): self.fn, self.name = fn, name
    def __call__(self, *a):
        t0 = time.perf_counter(); r = self.fn(*a); d = time.perf_counter() - t0
        e = acc[self.name]; e[0] += 1; e[1] += d
        return r
for name in _lib.declared_symbols():
    try: setattr(lib, name, Wrap(getattr(lib, name), name))
    except Exception: pass
N = 3
t0 = time.perf_counter()
for _ in range(N): step(x, t)
t1 = time.perf_counter()
torch.cuda.synchronize()
tot = (t1 - t0) / N * 1e3
inside = sum(v[1] for v in acc.values()) / N * 1e3
calls = sum(v[0] for v in acc.values()) / N
print(f"{model}: host {tot:.2f} ms/step on an empty queue; {calls:.0f} C-ABI calls/step take {inside:.2f} ms ({inside / calls * 1e3:.1f} us each); Python + torch around them {tot - inside:.2f} ms")
for name, (c, s) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {name:40s} {c / N:6.1f} calls/step {s / N * 1e3:7.3f} ms/step {s / c * 1e6:6.1f} us each")
