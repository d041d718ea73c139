"""Same-box timing of the e4m3 all-taps weight gradient against the bf16 kernels (ms per launch, incl. the slab reduce).
python scripts/time_wgrad_f8.py"""
import torch

from brats21_amd import ops

dev = torch.device("cuda:0")


def bench(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for (c1, c2, cout, size, nb) in [(48, 0, 48, 128, 2), (48, 48, 48, 128, 2), (96, 0, 96, 64, 2), (96, 96, 96, 64, 2), (192, 0, 192, 32, 2),
                                 (64, 0, 64, 128, 2), (64, 64, 64, 128, 2), (128, 0, 128, 64, 2), (256, 0, 256, 32, 2), (32, 32, 64, 128, 2)]:
    g = torch.Generator(device=dev).manual_seed(0)
    x1 = torch.randn((nb, size, size, size, c1), device=dev, generator=g).to(torch.bfloat16)
    x2 = torch.randn((nb, size, size, size, c2), device=dev, generator=g).to(torch.bfloat16) if c2 else None
    dy = (torch.randn((nb, size, size, size, cout), device=dev, generator=g) * 1e-3).to(torch.bfloat16)
    a1, a2, ady = ops.absmax(x1), (ops.absmax(x2) if c2 else None), ops.absmax(dy)
    tb = bench(lambda: ops.conv3d_wgrad(x1, dy, 3, 1, x2=x2))
    if ops.conv3d_wgrad_f8_ok(x1, dy, x2):
        t8 = bench(lambda: ops.conv3d_wgrad_f8(x1, dy, a1, ady, x2=x2, amax2=a2))
        dwb, _ = ops.conv3d_wgrad(x1, dy, 3, 1, x2=x2)
        dw8 = ops.conv3d_wgrad_f8(x1, dy, a1, ady, x2=x2, amax2=a2)
        rel = float((dw8 - dwb).norm() / dwb.norm())
    else:
        t8, rel = float("nan"), float("nan")
    gf = 2.0 * 27 * (c1 + c2) * cout * nb * size ** 3 / 1e9
    print(f"{c1}+{c2}->{cout} @{nb}x{size}^3: bf16 {tb:.3f} ms ({gf / tb:.0f} TF/s*1e-3)  e4m3 {t8:.3f} ms ({gf / t8:.0f})  x{tb / t8:.2f}  rel diff {rel:.2e}", flush=True)
