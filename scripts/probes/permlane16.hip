#include <hip/hip_runtime.h>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* p) {
  unsigned a = p[threadIdx.x], b = p[threadIdx.x + 64];
  u2 r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  p[threadIdx.x] = r[0]; p[threadIdx.x + 64] = r[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 128 * 4);
  unsigned h[128]; for (int i = 0; i < 128; ++i) h[i] = i < 64 ? 1000 + i : 2000 + (i - 64);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int i = 0; i < 64; i += 8) printf("lane %2d: a'=%u b'=%u\n", i, h[i], h[64 + i]);
  return 0;
}
