// Probe (gfx950): semantics of v_cvt_scalef32_pk_fp8_bf16 and the operand layout of v_mfma_scale_f32_16x16x128_f8f6f4
// with e4m3 operands.  Build: hipcc --offload-arch=gfx950 -O2 scripts/probes/fp8.hip -o /tmp/fp8probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__global__ void cvt_kernel(const unsigned* src, unsigned* dst, float scale, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bf16x2 v = __builtin_bit_cast(bf16x2, src[i]);
  s16x2 r = {0, 0};
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(r, v, scale, false);
  dst[i] = __builtin_bit_cast(unsigned, r);
}
template <int SCALE>  // 127 = 2^0 block scales in a VGPR; 0 = hipcc selects the unscaled v_mfma_f32_16x16x128_f8f6f4
__global__ void mfma_kernel(const i32x8* a, const i32x8* b, f32x4* c) {
  int l = threadIdx.x;
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0, 0, 0, SCALE, 0, SCALE);
  c[l] = acc;
}
static float e4m3(unsigned char b) {
  int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v;
  if (e == 15 && m == 7) v = NAN;
  else if (e == 0) v = ldexpf((float)m, -9);
  else v = ldexpf(1.f + m / 8.f, e - 7);
  return s ? -v : v;
}
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
int main() {
  // ---- conversion ----
  const float vals[] = {0.f, 1.f, -1.f, 0.5f, 1.0625f, 1.1875f, 3.3f, 100.f, 447.f, 448.f, 460.f, 480.f, 1000.f, 1e6f, -1e6f,
                        0.001953125f, 0.0009765625f, 0.0015f, 0.00048828125f, 1e-5f, INFINITY, NAN, 17.f, 19.f, 21.f, 23.f, 13.5f, 14.5f};
  const int nv = sizeof(vals) / sizeof(float);
  unsigned h[64], o[64], *ds, *dd;
  for (int i = 0; i < nv; ++i) h[i] = f2bf(vals[i]) | ((unsigned)f2bf(-2.f * vals[i]) << 16);
  hipMalloc(&ds, 256); hipMalloc(&dd, 256);
  hipMemcpy(ds, h, nv * 4, hipMemcpyHostToDevice);
  for (float sc : {1.f, 4.f, 0.25f}) {
    hipLaunchKernelGGL(cvt_kernel, dim3(1), dim3(64), 0, 0, ds, dd, sc, nv);
    hipMemcpy(o, dd, nv * 4, hipMemcpyDeviceToHost);
    printf("scale %g\n", sc);
    for (int i = 0; i < nv; ++i)
      printf("  in %12g -> 0x%02x = %10g | in %12g -> 0x%02x = %10g  (upper half 0x%04x)\n", vals[i], o[i] & 255, e4m3(o[i] & 255),
             -2.f * vals[i], (o[i] >> 8) & 255, e4m3((o[i] >> 8) & 255), o[i] >> 16);
  }
  // ---- MFMA layout: assume lane (q = l>>4, v = l&15) holds row v, k = 32q .. 32q+31 (byte j of the 8 dwords) ----
  unsigned char A[16][128], B[16][128];
  srand(1);
  for (int i = 0; i < 16; ++i) for (int k = 0; k < 128; ++k) {
    A[i][k] = (unsigned char)(rand() % 0x78) | ((rand() & 1) << 7);
    B[i][k] = (unsigned char)(rand() % 0x78) | ((rand() & 1) << 7);
  }
  unsigned char ha[64][32], hb[64][32];
  for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) { ha[l][j] = A[l & 15][32 * (l >> 4) + j]; hb[l][j] = B[l & 15][32 * (l >> 4) + j]; }
  i32x8 *da, *db; f32x4* dc; float hc[64][4];
  hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dc, 1024);
  hipMemcpy(da, ha, 2048, hipMemcpyHostToDevice); hipMemcpy(db, hb, 2048, hipMemcpyHostToDevice);
  for (int variant = 0; variant < 2; ++variant) {
  if (variant == 0) hipLaunchKernelGGL(mfma_kernel<127>, dim3(1), dim3(64), 0, 0, da, db, dc);
  else hipLaunchKernelGGL(mfma_kernel<0>, dim3(1), dim3(64), 0, 0, da, db, dc);
  hipMemcpy(hc, dc, 1024, hipMemcpyDeviceToHost);
  double maxerr = 0, maxref = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    const int row = (l >> 4) * 4 + r, col = l & 15;
    double ref = 0;
    for (int k = 0; k < 128; ++k) ref += (double)e4m3(A[row][k]) * e4m3(B[col][k]);
    maxerr = fmax(maxerr, fabs(ref - hc[l][r])); maxref = fmax(maxref, fabs(ref));
  }
  printf("mfma 16x16x128 e4m3 (%s): max |err| %g (max |ref| %g)  [D[row=4q+r][col=v] = sum_k A[row][k] B[col][k]]\n",
         variant == 0 ? "scales 127" : "scales 0 -> unscaled instruction", maxerr, maxref);
  }
  return 0;
}
