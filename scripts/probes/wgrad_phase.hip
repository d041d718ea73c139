// Probe (gfx950): what limits the MFMA phase of the all-taps weight-gradient kernel?  512-thread workgroups (2 waves per
// SIMD), one per CU, LDS filled with random bf16; each wave runs the kernel's u-step loop (11 pairs x 8 k-steps per
// "tile", 3 MFMAs per u-step) in variants:
//   0: MFMAs only (operands stay in registers)          1: + the kernel's transposing LDS reads (2 per u-step + A)
//   2: reads issued but MFMAs use register operands (no lgkmcnt dependency on them)
//   3: like 1 with 32x32x16 MFMAs on merged fragments (same MACs, half the LDS reads per MAC)
// Prints shader cycles per 16x16x32-equivalent MFMA per SIMD (ideal 16) and the in-kernel clock.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/probes/wgrad_phase.hip -o /tmp/wgrad_phase
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <type_traits>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define DEVI __device__ __forceinline__
template <int I, int N, typename F> DEVI void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}
DEVI bf16x8 tr_pair(const char* p0, const char* p1) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}
constexpr int HY = 6, HX = 18, SX = 96, SY = 96, XB = 65536, PPW = 11;

template <int MODE>
__global__ __launch_bounds__(512, 1) void k(const uint32_t* seed, float* out, long long* stamps, int tiles) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q = lane >> 4, v = lane & 15;
  for (int i = tid; i < (2 * XB + 24576) / 4; i += 512) ((uint32_t*)lds)[i] = seed[(i * 7 + blockIdx.x) & 65535];
  __syncthreads();
  char* ldy = lds + 2 * XB;
  int poff[PPW];
#pragma unroll
  for (int jj = 0; jj < PPW; ++jj) {
    const int pid = wave + 8 * jj, t = pid / 3, nn = pid % 3;
    poff[jj] = pid < 81 ? (((t / 9) * HY + (t / 3) % 3) * HX + t % 3) * SX + nn * 32 : 0;
  }
  const int qq = v >> 2, pp = v & 3;
  const int ybase = (4 * q + qq) * SY + pp * 8, xbase = (4 * q + qq) * SX + pp * 8;
  f32x4 acc[PPW][3];
#pragma unroll
  for (int jj = 0; jj < PPW; ++jj)
#pragma unroll
    for (int m = 0; m < 3; ++m) acc[jj][m] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x16 acc32[6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc32[i][r] = 0.f;
  const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int tile = 0; tile < tiles; ++tile) {
    const char* ldx = lds + (tile & 1) * XB;
    if constexpr (MODE <= 2) {
      constexpr int PD = 3;
      bf16x8 a[2][3], b[PD + 1];
      bf16x8 ra = tr_pair(ldy + ybase, ldy + ybase + 16 * SY), rb = tr_pair(ldx + xbase, ldx + xbase + HX * SX);
      auto read_a = [&](auto s_) {
        constexpr int s = s_;
        const int yoff = ybase + (32 * s) * SY;
#pragma unroll
        for (int m = 0; m < 3; ++m) a[s & 1][m] = MODE == 0 ? ra : tr_pair(ldy + yoff + m * 32, ldy + yoff + 16 * SY + m * 32);
      };
      auto read_b = [&](auto u_) {
        constexpr int u = u_;
        constexpr int s = u / PPW, jj = u % PPW;
        const int xoff = xbase + (((s >> 1) * HY + 2 * (s & 1)) * HX) * SX + poff[jj];
        b[u % (PD + 1)] = MODE == 0 ? rb : tr_pair(ldx + xoff, ldx + xoff + HX * SX);
      };
      constexpr int NU = 8 * PPW;
      read_a(std::integral_constant<int, 0>{});
      static_for<0, PD>([&](auto u_) { read_b(u_); });
      static_for<0, NU>([&](auto u_) {
        constexpr int u = u_;
        constexpr int s = u / PPW, jj = u % PPW;
        if constexpr (u + PD < NU) read_b(std::integral_constant<int, u + PD>{});
        if constexpr (jj == 0 && s + 1 < 8) read_a(std::integral_constant<int, s + 1>{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          if constexpr (MODE == 2) {
            asm volatile("" :: "v"(a[s & 1][m]), "v"(b[u % (PD + 1)]));
            acc[jj][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ra, rb, acc[jj][m], 0, 0, 0);
          } else {
            acc[jj][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s & 1][m], b[u % (PD + 1)], acc[jj][m], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    } else {
      // 32x32x16: per 16-voxel k-step one A32 fragment (2 tr reads) and, per merged pair (6 per wave ~ 11 pairs), one B32
      // fragment (2 tr reads); 6 accumulators x 16 registers; 16 k-steps x 6 MFMAs = 96 MFMAs of 4x the MACs of a 16x16x32
      constexpr int PD = 2;
      bf16x8 a[2], b[PD + 1];
      const int g = lane >> 4;  // 16-lane group: (g & 1) picks the merged pair's half, (g >> 1) the k half
      const int yb32 = (8 * (g >> 1) + (v >> 2)) * SY + (g & 1) * 32 + (v & 3) * 8;
      int xb32[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) xb32[j] = (8 * (g >> 1) + (v >> 2)) * SX + (v & 3) * 8 + poff[(2 * j + (g & 1)) % PPW];
      auto read_a = [&](auto s_) {
        constexpr int s = s_;  // k-step of 16 voxels = one x-row
        a[s & 1] = tr_pair(ldy + yb32 + (16 * s) * SY, ldy + yb32 + (16 * s + 4) * SY);
      };
      auto read_b = [&](auto u_) {
        constexpr int u = u_;
        constexpr int s = u / 6, j = u % 6;
        const int xoff = xb32[j] + (((s >> 2) * HY + (s & 3)) * HX) * SX;
        b[u % (PD + 1)] = tr_pair(ldx + xoff, ldx + xoff + 4 * SX);
      };
      constexpr int NU = 16 * 6;
      read_a(std::integral_constant<int, 0>{});
      static_for<0, PD>([&](auto u_) { read_b(u_); });
      static_for<0, NU>([&](auto u_) {
        constexpr int u = u_;
        constexpr int s = u / 6, j = u % 6;
        if constexpr (u + PD < NU) read_b(std::integral_constant<int, u + PD>{});
        if constexpr (j == 0 && s + 1 < 16) read_a(std::integral_constant<int, s + 1>{});
        __builtin_amdgcn_sched_barrier(0);
        acc32[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s & 1], b[u % (PD + 1)], acc32[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      });
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int jj = 0; jj < PPW; ++jj)
#pragma unroll
    for (int m = 0; m < 3; ++m) s += acc[jj][m];
#pragma unroll
  for (int i = 0; i < 6; ++i) s[0] += acc32[i][0] + acc32[i][5];
  out[(size_t)blockIdx.x * 512 + tid] = s[0] + s[1] + s[2] + s[3];
  if (lane == 0) { stamps[(blockIdx.x * 8 + wave) * 2] = t1 - t0; stamps[(blockIdx.x * 8 + wave) * 2 + 1] = r1 - r0; }
}

template <int MODE> void run(const char* name, uint32_t* seed, float* out, long long* stamps, int tiles) {
  const int lds = 2 * XB + 24576;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), lds, 0, seed, out, stamps, tiles);
  hipDeviceSynchronize();
  static long long h[256 * 8 * 2];
  hipMemcpy(h, stamps, sizeof(h), hipMemcpyDeviceToHost);
  double cyc = 0, real = 0;
  for (int i = 0; i < 256 * 8; ++i) { cyc += h[2 * i]; real += h[2 * i + 1]; }
  cyc /= 256 * 8; real /= 256 * 8;
  // per tile and SIMD: 2 waves x 264 MFMAs of 16x16x32 (variant 3: 2 x 96 MFMAs of 32x32x16 = 4 x the MACs each, 12/11 of the work)
  const double mf = MODE == 3 ? 2.0 * 96 * 4 : 2.0 * 264;
  printf("%-44s %6.1f cycles per 16x16x32-equivalent MFMA per SIMD (ideal 16), clock %.2f GHz\n", name, cyc / tiles / mf * 1.0,
         cyc / real * 0.1);
}

int main() {
  uint32_t* seed; float* out; long long* stamps;
  hipMalloc(&seed, 65536 * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&stamps, 256 * 8 * 2 * 8);
  static uint32_t h[65536];
  uint32_t x = 12345;
  for (int i = 0; i < 65536; ++i) {  // random bf16 pairs in [-2, 2): random sign / mantissa, exponent 0x3f..0x40
    x = x * 1664525u + 1013904223u; const uint32_t a = (x >> 8) & 0xffff;
    x = x * 1664525u + 1013904223u; const uint32_t b = (x >> 8) & 0xffff;
    h[i] = ((a & 0x80ff) | 0x3f00 | ((a >> 14 & 1) << 7)) | (((b & 0x80ff) | 0x3f00) << 16);
  }
  hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
  const int tiles = 400;
  run<0>("0: MFMA only", seed, out, stamps, tiles);
  run<1>("1: kernel pattern (tr reads + MFMA)", seed, out, stamps, tiles);
  run<2>("2: reads issued, MFMAs on register operands", seed, out, stamps, tiles);
  run<3>("3: 32x32x16 on merged fragments", seed, out, stamps, tiles);
  return 0;
}
