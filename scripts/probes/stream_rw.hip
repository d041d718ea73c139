// Probe (gfx950): what does a bf16 "read 403 MB, scale-shift-relu, write 403 MB" pass reach, by grid size, loads in
// flight and store policy?  (GroupNorm apply at 128^3 x 48 channels x batch 2 runs at ~4.5 TB/s.)
// Build: hipcc --offload-arch=gfx950 -O3 scripts/probes/stream_rw.hip -o scripts/probes/stream_rw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 work(u32x4 v, float sc, float sh) {
  u32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float a = __uint_as_float(v[i] << 16) * sc + sh, b = __uint_as_float(v[i] & 0xffff0000u) * sc + sh;
    a = a > 0.f ? a : 0.f; b = b > 0.f ? b : 0.f;
    o[i] = (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xffff0000u);
  }
  return o;
}
template <int INFLIGHT, int NT>
__global__ void __launch_bounds__(256) k(const u32x4* __restrict__ x, u32x4* __restrict__ y, size_t n, float sc, float sh) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (INFLIGHT - 1) * stride < n; i += INFLIGHT * stride) {
    u32x4 v[INFLIGHT];
#pragma unroll
    for (int j = 0; j < INFLIGHT; ++j) v[j] = NT & 1 ? __builtin_nontemporal_load(x + i + j * stride) : x[i + j * stride];
#pragma unroll
    for (int j = 0; j < INFLIGHT; ++j) {
      const u32x4 o = work(v[j], sc, sh);
      if (NT & 2) __builtin_nontemporal_store(o, y + i + j * stride); else y[i + j * stride] = o;
    }
  }
  for (; i < n; i += stride) y[i] = work(x[i], sc, sh);
}
template <int INFLIGHT, int NT> void run(const u32x4* x, u32x4* y, size_t n, int blocks) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<INFLIGHT, NT>), dim3(blocks), dim3(256), 0, 0, x, y, n, 0.5f, 0.1f);
  hipEventRecord(a);
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k<INFLIGHT, NT>), dim3(blocks), dim3(256), 0, 0, x, y, n, 0.5f, 0.1f);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
  printf("inflight %d nt %d blocks %5d: %.3f ms  %.2f TB/s\n", INFLIGHT, NT, blocks, ms, 2.0 * n * 16 / ms / 1e9);
}
int main() {
  const size_t n = (size_t)2 * 128 * 128 * 128 * 48 * 2 / 16;  // 16-byte vectors in a 2 x 128^3 x 48 bf16 tensor
  u32x4 *x, *y; hipMalloc(&x, n * 16); hipMalloc(&y, n * 16);
  hipMemset(x, 0x3c, n * 16);
  for (int blocks : {4096, 16384, 32768, 65536, 98304}) {
    run<1, 0>(x, y, n, blocks); run<1, 1>(x, y, n, blocks); run<1, 2>(x, y, n, blocks); run<1, 3>(x, y, n, blocks);
    run<2, 3>(x, y, n, blocks); run<4, 3>(x, y, n, blocks);
  }
  return 0;
}
