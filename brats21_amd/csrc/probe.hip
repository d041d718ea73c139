// Box calibration probes behind the C ABI (bench.py's "box" record): what THIS chip, at the clocks it holds right now,
// delivers on the two resources every roofline of this library is quoted against -- the dense 16-bit MFMA pipe and an HBM
// read + write stream.  The boxes of one MI355X pool differ by up to 9 % on the dominant kernel (0.402 vs
// 0.439 ms for identical code), and the chip's clock under MFMA load depends on power management and on the data; a
// roofline fraction against the nominal 2.5 PFLOP/s cannot tell a slower box from slower code, the fraction against the
// probe's rate can.  ~25 ms each; the caller times them with HIP events on the launch stream.
#include "common.hpp"

// every SIMD issues v_mfma_f32_16x16x32_bf16 on 8 independent accumulators (the pipe's issue limit: 16 cycles each), two
// waves per SIMD like the convolution kernels.  Operands are pseudo-random finite bf16 values; ZEROS = every second value
// is zero (post-ReLU activations: the chip clocks higher on them); VALU = dependent vector instructions between two groups of
// 8 MFMAs (0: the pure matrix loop -- a power virus on which the chip falls far below its nominal clock; 24: the ~60 % matrix
// duty of the implicit-GEMM kernels, whose clock under load is what differs between the boxes of a pool)
template <int ZEROS, int VALU>
__global__ void __launch_bounds__(256) probe_mfma_kernel(float* __restrict__ out, int iters) {
  // 80 KB of (unused) LDS per workgroup: at most two workgroups fit a CU, so a grid of 2 x #CU workgroups puts exactly two
  // waves on every SIMD (without it the dispatcher may stack several of these tiny workgroups on some CUs and leave others idle)
  extern __shared__ char lds_pad[];
  if (iters < 0) lds_pad[threadIdx.x] = 0;
  uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
  uint32_t w[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    s = s * 1664525u + 1013904223u;
    // two bf16 values in [0.5, 2) with random signs and mantissas
    w[i] = (s & 0x807f807fu) | 0x3f003f00u | ((s >> 8) & 0x00800080u);
    if (ZEROS) w[i] &= (s & 0x10000u) ? 0xffff0000u : 0x0000ffffu;
  }
  const bf16x8 a = __builtin_bit_cast(bf16x8, u32x4{w[0], w[1], w[2], w[3]});
  const bf16x8 b = __builtin_bit_cast(bf16x8, u32x4{w[4], w[5], w[6], w[7]});
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float v = __uint_as_float(w[0]);
  // (in-place inline assembly: with the builtin hipcc parks the accumulators in AGPRs and shuffles them through ~40
  //  v_accvgpr moves per iteration -- the loop then runs at half the matrix pipe's rate)
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
    for (int j = 0; j < VALU; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(1.0001f));
  }
  f32x4 t = acc[0];
#pragma unroll
  for (int i = 1; i < 8; ++i) t += acc[i];
  // (one store per block keeps the result alive; the probe moves no data)
  if (threadIdx.x == 0) out[blockIdx.x] = t[0] + t[1] + t[2] + t[3] + v;
}

extern "C" int brats_probe_mfma(float* out, int blocks, int iters, int mode, brats_stream_t s) {
  if (!out || blocks <= 0 || iters <= 0 || mode < 0 || mode > 3) BRATS_FAIL(BRATS_E_ARG, "probe_mfma: bad argument");
  hipStream_t st = (hipStream_t)s;
  constexpr int lds = 80 * 1024;
  static std::atomic<uint64_t> d0{0}, d1{0}, d2{0}, d3{0};
  BRATS_ENSURE_LDS_ATTR((probe_mfma_kernel<0, 0>), lds, d0);
  BRATS_ENSURE_LDS_ATTR((probe_mfma_kernel<1, 0>), lds, d1);
  BRATS_ENSURE_LDS_ATTR((probe_mfma_kernel<0, 24>), lds, d2);
  BRATS_ENSURE_LDS_ATTR((probe_mfma_kernel<1, 24>), lds, d3);
  if (mode == 0) hipLaunchKernelGGL((probe_mfma_kernel<0, 0>), dim3(blocks), dim3(256), lds, st, out, iters);
  if (mode == 1) hipLaunchKernelGGL((probe_mfma_kernel<1, 0>), dim3(blocks), dim3(256), lds, st, out, iters);
  if (mode == 2) hipLaunchKernelGGL((probe_mfma_kernel<0, 24>), dim3(blocks), dim3(256), lds, st, out, iters);
  if (mode == 3) hipLaunchKernelGGL((probe_mfma_kernel<1, 24>), dim3(blocks), dim3(256), lds, st, out, iters);
  BRATS_CHECK_LAUNCH();
  return 0;
}

// bf16 "read, scale-shift-relu, write" over `bytes` (the shape of a GroupNorm apply pass) the way the library's own
// streaming kernels run beyond the Infinity Cache: many short-lived blocks, four 16-byte vectors in flight per thread,
// non-temporal loads and stores
__global__ void __launch_bounds__(256) probe_stream_kernel(const u32x4* __restrict__ x, u32x4* __restrict__ y, size_t n) {
  const size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x);
  const size_t stride = (size_t)gridDim.x * 256;
  u32x4 v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const size_t i = base + j * stride;
    v[j] = i < n ? __builtin_nontemporal_load(x + i) : u32x4{0, 0, 0, 0};
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const size_t i = base + j * stride;
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float lo = __uint_as_float(v[j][e] << 16) * 0.5f + 0.1f, hi = __uint_as_float(v[j][e] & 0xffff0000u) * 0.5f + 0.1f;
      lo = lo > 0.f ? lo : 0.f;
      hi = hi > 0.f ? hi : 0.f;
      o[e] = (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xffff0000u);
    }
    if (i < n) __builtin_nontemporal_store(o, y + i);
  }
}

extern "C" int brats_probe_stream(const void* src, void* dst, size_t bytes, brats_stream_t s) {
  if (!src || !dst || bytes < 16 || (bytes & 15) || ((size_t)src & 15) || ((size_t)dst & 15)) BRATS_FAIL(BRATS_E_ARG, "probe_stream: bad argument");
  const size_t n = bytes / 16;
  const size_t blocks = (n + 1023) / 1024;
  hipLaunchKernelGGL(probe_stream_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, (const u32x4*)src, (u32x4*)dst, n);
  BRATS_CHECK_LAUNCH();
  return 0;
}
