"""Does the dominant kernel's time depend on WHERE its tensors lie?  48 -> 48 @2x128^3 bf16 (conv_igemm_vs8<24>), input fixed at the start
of one 3 GB arena, output placed at a sweep of byte offsets behind it; and the input itself shifted.  python scripts/probes/conv_addr_sweep.py"""
import sys
import torch
sys.path.insert(0, ".")
from brats21_amd import ops

dev = torch.device("cuda:0")
n, s, c = 2, 128, 48
nb = n * s ** 3 * c * 2
arena = torch.empty(3 * 2 ** 30, dtype=torch.uint8, device=dev)
print(f"arena at 0x{arena.data_ptr():x}; tensor {nb / 2 ** 20:.1f} MiB")
g = torch.Generator(device="cpu").manual_seed(0)
w = (torch.randn(c, c, 3, 3, 3, generator=g) * 0.03).to(dev)
wpk = ops.pack_weights(w, torch.bfloat16, ops.PACK_FWD)


def view(off):
    return arena[off:off + nb].view(torch.bfloat16).view(n, s, s, s, c)


def fill(t):
    t.copy_(torch.relu(torch.randn(t.shape, device=dev, dtype=torch.float32)).to(torch.bfloat16))


def timed(x, y, reps=20):
    for _ in range(3):
        ops.conv3d(x, wpk, c, 3, 1, out=y)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv3d(x, wpk, c, 3, 1, out=y)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


x = view(0)
fill(x)
base = (nb + 2 ** 21 - 1) // 2 ** 21 * 2 ** 21
print("output offset behind the input (rounded up to 2 MiB) + delta:")
for delta in (0, 256, 1024, 4096, 16384, 65536, 2 ** 18, 2 ** 20, 2 ** 21, 3 * 2 ** 20, 2 ** 24, 2 ** 27 + 4096, 2 ** 29, 2 ** 30):
    t = [timed(x, view(base + delta)) for _ in range(3)]
    print(f"  delta {delta:>11d}: {min(t):.4f} .. {max(t):.4f} ms")
print("input shifted (output fixed at +1 GiB):")
y = view(2 ** 30 + 2 ** 21)
for off in (0, 256, 4096, 65536, 2 ** 20, 2 ** 21 + 256, 2 ** 24):
    x = view(off)
    fill(x)
    t = [timed(x, y) for _ in range(3)]
    print(f"  input at {off:>9d}: {min(t):.4f} .. {max(t):.4f} ms")
