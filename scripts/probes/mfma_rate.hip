// Probe (gfx950): issue rate of the MX-scaled e4m3 MFMA against the bf16 MFMA, one wave per SIMD, 8 independent
// accumulators.  Build: hipcc --offload-arch=gfx950 -O2 scripts/probes/mfma_rate.hip -o /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ void k(const i32x8* in, f32x4* out, long long* cyc, int iters, int sa, int sb) {
  i32x8 a = in[threadIdx.x], b = in[64 + threadIdx.x];
  bf16x8 ha = __builtin_bit_cast(bf16x8, in[threadIdx.x].lo), hb = __builtin_bit_cast(bf16x8, in[threadIdx.x].hi);
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 0, 0, 0, sa, 0, sb);
      if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 0, 0, 0, 0, 0, 0);
      if (MODE == 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ha, hb, acc[i], 0, 0, 0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  f32x4 s = acc[0];
  for (int i = 1; i < 8; ++i) s += acc[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  i32x8* in; f32x4* out; long long* cyc; long long h;
  hipMalloc(&in, 128 * 32); hipMemset(in, 0x38, 128 * 32); hipMalloc(&out, 1024 * 64 * 16); hipMalloc(&cyc, 8);
  const int iters = 2000;
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1024), dim3(64), 0, 0, in, out, cyc, iters, 127, 127);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1024), dim3(64), 0, 0, in, out, cyc, iters, 0, 0);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1024), dim3(64), 0, 0, in, out, cyc, iters, 0, 0);
      hipDeviceSynchronize();
    }
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("mode %d (%s): %.1f counter ticks per MFMA\n", mode,
           mode == 0 ? "mfma_scale 16x16x128 e4m3, scales 127" : mode == 1 ? "mfma_scale 16x16x128 e4m3, scales const 0" : "mfma 16x16x32 bf16",
           (double)h / (iters * 8.0));
  }
  return 0;
}
