// Probe (gfx950): does a second pass over a tensor that a first pass has just read hit the 256 MB Infinity Cache -- by size
// (100 / 201 / 403 MB), by direction of the second pass (same order / reverse), and by load policy of the two passes?
// Models GroupNorm / EvoNorm backward: reduce pass (read dz, y) then apply pass (read dz, y again, write dy).
// Build: hipcc --offload-arch=gfx950 -O3 scripts/probes/mall_reuse.hip -o scripts/probes/mall_reuse
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// pass 1: read-only reduction (sum), NT1 = non-temporal loads
template <int NT1>
__global__ void __launch_bounds__(256) k_read(const u32x4* __restrict__ x, size_t n, uint32_t* out) {
  const size_t stride = (size_t)gridDim.x * 256;
  uint32_t acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const u32x4 v = NT1 ? __builtin_nontemporal_load(x + i) : x[i];
    acc += v[0] ^ v[1] ^ v[2] ^ v[3];
  }
  if (acc == 0x12345678u) out[0] = acc;
}
// pass 2: read x (REV: blocks walk from the end), write y (non-temporal)
template <int REV, int NT2>
__global__ void __launch_bounds__(256) k_rw(const u32x4* __restrict__ x, u32x4* __restrict__ y, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  const size_t first = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (size_t j = first; j < n; j += stride) {
    const size_t i = REV ? n - 1 - j : j;
    u32x4 v = NT2 ? __builtin_nontemporal_load(x + i) : x[i];
    v[0] += 1;
    __builtin_nontemporal_store(v, y + i);
  }
}
template <int NT1, int REV, int NT2> void run(const u32x4* x, u32x4* y, size_t n, uint32_t* out, const char* tag) {
  hipEvent_t a, b, c; hipEventCreate(&a); hipEventCreate(&b); hipEventCreate(&c);
  const int blocks = 8192;
  float t1 = 0, t2 = 0;
  for (int r = 0; r < 6; ++r) {
    hipEventRecord(a);
    hipLaunchKernelGGL((k_read<NT1>), dim3(blocks), dim3(256), 0, 0, x, n, out);
    hipEventRecord(b);
    hipLaunchKernelGGL((k_rw<REV, NT2>), dim3(blocks), dim3(256), 0, 0, x, y, n);
    hipEventRecord(c); hipEventSynchronize(c);
    float m1, m2; hipEventElapsedTime(&m1, a, b); hipEventElapsedTime(&m2, b, c);
    if (r >= 2) { t1 += m1 / 4; t2 += m2 / 4; }
  }
  printf("%4zu MB  %-28s pass1 %.3f ms (%.2f TB/s)  pass2 %.3f ms (%.2f TB/s r+w)\n", n * 16 >> 20, tag, t1, n * 16 / t1 / 1e9, t2, 2.0 * n * 16 / t2 / 1e9);
}
int main() {
  const size_t nmax = (size_t)403 << 16;  // 16-byte vectors in 403 MB
  u32x4 *x, *y; uint32_t* out;
  hipMalloc(&x, nmax * 16); hipMalloc(&y, nmax * 16); hipMalloc(&out, 4);
  hipMemset(x, 0x3c, nmax * 16);
  for (size_t mb : {100, 201, 403}) {
    const size_t n = mb << 16;
    run<0, 0, 0>(x, y, n, out, "plain -> fwd plain");
    run<0, 1, 0>(x, y, n, out, "plain -> reverse plain");
    run<0, 1, 1>(x, y, n, out, "plain -> reverse NT");
    run<1, 0, 1>(x, y, n, out, "NT -> fwd NT");
    run<1, 1, 1>(x, y, n, out, "NT -> reverse NT");
  }
  return 0;
}
