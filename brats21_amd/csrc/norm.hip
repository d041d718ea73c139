// GroupNorm(8)+activation and EvoNorm-S0 (statistics finalize, fused apply, backward), NDHWC.
// HBM-bound streaming kernels: 16-byte vector accesses, per-sample scale/shift staged in LDS.
// Reference semantics: nn.GroupNorm(8, C) networks/factory.py:179-182 (biased var, eps 1e-5),
// activations networks/factory.py:195-200; EvoNorm3D S0 networks/equiunet2021.py:95-103 (unbiased
// group variance, x*sigmoid(x) numerator).
#include "twin_begin.hpp"
#include "common.hpp"

// beyond the Infinity Cache (256 MB): non-temporal streaming on more, shorter-lived blocks (common.hpp stream_nt); f32 tensors are
// the split-precision mode's (2 x 128^3 x 48 x 4 B = 805 MB)
static inline bool big_tensor(int dtype, size_t elems) {
#ifdef BRATS_NO_F32_NT  // (A/B builds)
  if (dtype != BRATS_BF16) return false;
#endif
  return stream_nt(elems * (dtype == BRATS_BF16 ? 2 : 4));
}
#include "se.hpp"

// ---- statistics finalize ---------------------------------------------------------------------------
// stage 1: part[z][n][c] = sum over a slice of the tiles of the conv epilogue's per-tile partials (f64), one block per
// (n, 16-channel slab, slice): 16 x 16 threads, coalesced 128-byte rows; plain stores, no atomics, no zeroing.
// stage 2: one block per (n, group) adds the <= 64 slices in order (bitwise reproducible) and finishes.
// Workspace: chan_ws = [N][C][2] totals (kept: EvoNorm's backward reads them) followed by [splits][N][C][2] partials.
constexpr int GN_MAX_SPLITS = 64;
__global__ void gn_chan_reduce_kernel(const float* __restrict__ stats, int tps, int C, double* __restrict__ part) {
  const int n = blockIdx.y, c0 = blockIdx.x * 16;
  const int cl = threadIdx.x & 15, tl = threadIdx.x >> 4;
  const int c = c0 + cl;
  const int per = (tps + gridDim.z - 1) / gridDim.z;
  const int t0 = blockIdx.z * per, t1 = min(tps, t0 + per);
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
    for (int t = t0 + tl; t < t1; t += 16) {
      const f32x2 v = *(const f32x2*)(stats + (((size_t)n * tps + t) * C + c) * 2);
      s1 += v[0];
      s2 += v[1];
    }
  }
  __shared__ double r1[256], r2[256];
  r1[threadIdx.x] = s1;
  r2[threadIdx.x] = s2;
  __syncthreads();
  for (int m = 128; m >= 16; m >>= 1) {
    if ((int)threadIdx.x < m) { r1[threadIdx.x] += r1[threadIdx.x + m]; r2[threadIdx.x] += r2[threadIdx.x + m]; }
    __syncthreads();
  }
  if (threadIdx.x < 16 && c < C) {
    double* dst = part + (((size_t)blockIdx.z * gridDim.y + n) * C + c) * 2;
    dst[0] = r1[threadIdx.x];
    dst[1] = r2[threadIdx.x];
  }
}

static int gn_splits(int tps) {
  int splits = tps / 64;
  return splits < 1 ? 1 : (splits > GN_MAX_SPLITS ? GN_MAX_SPLITS : splits);
}

static int chan_reduce_launch(const float* stats, int tps, int N, int C, double* chan_ws, hipStream_t st) {
  hipLaunchKernelGGL(gn_chan_reduce_kernel, dim3((C + 15) / 16, N, gn_splits(tps)), dim3(256), 0, st, stats, tps, C,
                     chan_ws + (size_t)N * C * 2);
  return 0;
}

__global__ void gn_finalize_kernel(double* __restrict__ chan, int splits, int N, int C, int groups, double count_per_channel,
                                   float eps, int unbiased, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ mean_rstd, float* __restrict__ scale_shift) {
  const int n = blockIdx.x / groups, g = blockIdx.x % groups;
  const int cpg = C / groups;
  __shared__ double r1[256], r2[256];
  double s1 = 0.0, s2 = 0.0;
  {
    // thread (cl, zl): channel cl of the group, slices zl, zl + ZL, ...; then the ZL slice sums are added in order
    const int ZL = 256 / cpg;  // >= 1 (cpg <= 256)
    const int cl = threadIdx.x % cpg, zl = threadIdx.x / cpg;
    const double* part = chan + (size_t)N * C * 2;
    double p1 = 0.0, p2 = 0.0;
    if (zl < ZL) {
      const int c = g * cpg + cl;
      for (int z = zl; z < splits; z += ZL) {
        p1 += part[(((size_t)z * N + n) * C + c) * 2];
        p2 += part[(((size_t)z * N + n) * C + c) * 2 + 1];
      }
    }
    r1[threadIdx.x] = p1;
    r2[threadIdx.x] = p2;
    __syncthreads();
    if ((int)threadIdx.x < cpg) {
      for (int z = 0; z < ZL; ++z) { s1 += r1[z * cpg + threadIdx.x]; s2 += r2[z * cpg + threadIdx.x]; }
      const int c = g * cpg + threadIdx.x;
      chan[((size_t)n * C + c) * 2] = s1;  // per-channel totals (EvoNorm's backward reads them)
      chan[((size_t)n * C + c) * 2 + 1] = s2;
    }
    __syncthreads();
  }
  r1[threadIdx.x] = s1;
  r2[threadIdx.x] = s2;
  __syncthreads();
  for (int m = blockDim.x / 2; m > 0; m >>= 1) {
    if ((int)threadIdx.x < m) { r1[threadIdx.x] += r1[threadIdx.x + m]; r2[threadIdx.x] += r2[threadIdx.x + m]; }
    __syncthreads();
  }
  const double M = count_per_channel * cpg;
  const double mean = r1[0] / M;
  double var = r2[0] / M - mean * mean;
  if (unbiased) var = var * M / (M - 1.0);
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  if (threadIdx.x == 0) {
    mean_rstd[(n * groups + g) * 2] = (float)mean;
    mean_rstd[(n * groups + g) * 2 + 1] = rstd;
  }
  if (scale_shift && (int)threadIdx.x < cpg) {
    const int c = g * cpg + threadIdx.x;
    const float sc = rstd * gamma[c];
    scale_shift[((size_t)n * C + c) * 2] = sc;
    scale_shift[((size_t)n * C + c) * 2 + 1] = beta[c] - (float)mean * sc;
  }
}

// Both stages in ONE launch when a group's partials are few enough for one workgroup to take in a handful of load batches
// (tiles x channels-per-group <= 32 k pairs: every layer below the 128^3 level; with 48 k pairs the one block per group took 27 us): block (n, group) of 1024 threads, thread
// (channel cl, tile lane tl) adds the tiles tl, tl + TL, ... in f64 (eight loads in flight), the TL lane sums of a channel are
// added in lane order through LDS, then the group finishes as above.  One ~5 us launch instead of two (10 us): 17 - 21 such
// pairs per training step, 17 per patch forward of the sliding-window inference.
__global__ void __launch_bounds__(1024) gn_finalize_direct_kernel(const float* __restrict__ stats, int tps, double* __restrict__ chan,
                                                                  int N, int C, int groups, double count_per_channel, float eps,
                                                                  int unbiased, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, float* __restrict__ mean_rstd,
                                                                  float* __restrict__ scale_shift) {
  const int n = blockIdx.x / groups, g = blockIdx.x % groups;
  const int cpg = C / groups;
  __shared__ double r1[1024], r2[1024];
  const int TL = 1024 / cpg;  // >= 4 (cpg <= 256)
  const int cl = threadIdx.x % cpg, tl = threadIdx.x / cpg;
  double p1 = 0.0, p2 = 0.0;
  if (tl < TL) {
    const float* src = stats + (((size_t)n * tps) * C + g * cpg + cl) * 2;
    int t = tl;
    for (; t + 7 * TL < tps; t += 8 * TL) {
      f32x2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *(const f32x2*)(src + (size_t)(t + u * TL) * C * 2);
#pragma unroll
      for (int u = 0; u < 8; ++u) { p1 += v[u][0]; p2 += v[u][1]; }
    }
    for (; t < tps; t += TL) {
      const f32x2 v = *(const f32x2*)(src + (size_t)t * C * 2);
      p1 += v[0];
      p2 += v[1];
    }
  }
  r1[threadIdx.x] = p1;
  r2[threadIdx.x] = p2;
  __syncthreads();
  double s1 = 0.0, s2 = 0.0;
  if ((int)threadIdx.x < cpg) {
    for (int z = 0; z < TL; ++z) { s1 += r1[z * cpg + threadIdx.x]; s2 += r2[z * cpg + threadIdx.x]; }
    const int c = g * cpg + threadIdx.x;
    chan[((size_t)n * C + c) * 2] = s1;  // per-channel totals (EvoNorm's backward reads them)
    chan[((size_t)n * C + c) * 2 + 1] = s2;
  }
  __syncthreads();
  r1[threadIdx.x] = s1;
  r2[threadIdx.x] = s2;
  __syncthreads();
  for (int m = 128; m > 0; m >>= 1) {  // (only the first cpg <= 256 entries are non-zero)
    if ((int)threadIdx.x < m) { r1[threadIdx.x] += r1[threadIdx.x + m]; r2[threadIdx.x] += r2[threadIdx.x + m]; }
    __syncthreads();
  }
  const double M = count_per_channel * cpg;
  const double mean = r1[0] / M;
  double var = r2[0] / M - mean * mean;
  if (unbiased) var = var * M / (M - 1.0);
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  if (threadIdx.x == 0) {
    mean_rstd[(n * groups + g) * 2] = (float)mean;
    mean_rstd[(n * groups + g) * 2 + 1] = rstd;
  }
  if (scale_shift && (int)threadIdx.x < cpg) {
    const int c = g * cpg + threadIdx.x;
    const float sc = rstd * gamma[c];
    scale_shift[((size_t)n * C + c) * 2] = sc;
    scale_shift[((size_t)n * C + c) * 2 + 1] = beta[c] - (float)mean * sc;
  }
}

static int gn_finalize_launch(const float* stats, int tps, int N, int C, int groups, double count_per_channel, float eps, int unbiased,
                              const float* gamma, const float* beta, float* mean_rstd, float* scale_shift, double* chan_ws,
                              hipStream_t st) {
  if ((size_t)tps * (C / groups) <= 32768) {
    hipLaunchKernelGGL(gn_finalize_direct_kernel, dim3(N * groups), dim3(1024), 0, st, stats, tps, chan_ws, N, C, groups,
                       count_per_channel, eps, unbiased, gamma, beta, mean_rstd, scale_shift);
  } else {
    if (int rc = chan_reduce_launch(stats, tps, N, C, chan_ws, st)) return rc;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(N * groups), dim3(256), 0, st, chan_ws, gn_splits(tps), N, C, groups, count_per_channel,
                       eps, unbiased, gamma, beta, mean_rstd, scale_shift);
  }
  BRATS_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t BRATS_API(brats_gn_ws_doubles)(int N, int C) { return (size_t)(1 + GN_MAX_SPLITS) * N * C * 2; }

extern "C" int BRATS_API(brats_gn_finalize)(const float* stats, int tiles_per_sample, int N, int C, int groups,
                                 double count_per_channel, float eps, const float* gamma, const float* beta,
                                 float* mean_rstd, float* scale_shift, double* chan_ws, brats_stream_t s) {
  if (!stats || !mean_rstd || !chan_ws || C % groups || C / groups > 256) BRATS_FAIL(BRATS_E_ARG, "gn_finalize: bad argument");
  if (scale_shift && (!gamma || !beta)) BRATS_FAIL(BRATS_E_ARG, "gn_finalize: scale_shift needs gamma/beta");
  return gn_finalize_launch(stats, tiles_per_sample, N, C, groups, count_per_channel, eps, 0, gamma, beta, mean_rstd, scale_shift, chan_ws,
                            (hipStream_t)s);
}

// ---- z = act(y*scale + shift) ------------------------------------------------------------------
// HEAVY = false instantiations contain only relu / leakyrelu (the transcendental activations cost the streaming kernels
// ~10-20 % through code size and registers even when not selected)
// negative-side slope of leakyrelu: a host value, or -- nn.PReLU's learnable scalar -- read from device memory
struct SlopeArg { float v; const float* p; };
template <bool HEAVY>
DEVI float act_fwd(float x, int act, float slope) {
  if (act == BRATS_ACT_RELU) return x > 0.f ? x : 0.f;
  if (act == BRATS_ACT_LEAKY) return x > 0.f ? x : x * slope;
  if constexpr (!HEAVY) return x;
  if (act == BRATS_ACT_ELU) return x > 0.f ? x : expm1f(x);
  if (act == BRATS_ACT_SWISH) return x / (1.f + __expf(-x));
  if (act == BRATS_ACT_MISH) {
    const float sp = x > 20.f ? x : log1pf(__expf(x));  // softplus with torch's threshold
    return x * tanhf(sp);
  }
  return x;
}
template <bool HEAVY>
DEVI float act_grad(float x, int act, float slope) {  // derivative at pre-activation x
  if (act == BRATS_ACT_RELU) return x > 0.f ? 1.f : 0.f;
  if (act == BRATS_ACT_LEAKY) return x > 0.f ? 1.f : slope;
  if constexpr (!HEAVY) return 1.f;
  if (act == BRATS_ACT_ELU) return x > 0.f ? 1.f : __expf(x);
  if (act == BRATS_ACT_SWISH) {
    const float sg = 1.f / (1.f + __expf(-x));
    return sg * (1.f + x * (1.f - sg));
  }
  if (act == BRATS_ACT_MISH) {
    const float sp = x > 20.f ? x : log1pf(__expf(x));
    const float th = tanhf(sp);
    const float sg = 1.f / (1.f + __expf(-x));
    return th + x * (1.f - th * th) * sg;
  }
  return 1.f;
}

// |max| of a tensor as a side product of the kernel that writes it (the fp8 convolutions scale their input by it):
// wave maximum by DPP-free shuffles, one integer atomicMax on the bits of the non-negative float per wave.  max is
// order-independent, so this atomic keeps results bitwise reproducible.  The slot must be zero before the launch.
template <typename T> DEVI void record_absmax(float mx, uint32_t* amax) {
  mx = to_f<T>(from_f<T>(mx));  // rounding is monotonic: the maximum of the stored (rounded) values
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  // one access per BLOCK: tens of thousands of waves reading / updating one address serialise in a single L2 channel
  // (+20 us per launch when every wave did it).  The slot only grows, so a plain, possibly stale read filters all but
  // the few blocks that would still raise it.
  __shared__ float wmax[16];
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) mx = fmaxf(mx, wmax[w]);
    const uint32_t bits = __float_as_uint(mx);
    if (bits > *(volatile uint32_t*)amax) atomicMax(amax, bits);
  }
}

// A thread owns one 16-byte channel vector (its scale / shift live in registers) and walks voxels, two in flight;
// relu -- the published configuration -- is specialised so that the loop body is cvt, fma, max, cvt per element.
template <typename T, bool HEAVY, bool NT = false>
__global__ void __launch_bounds__(256) affine_act_kernel(const T* __restrict__ y, int ypitch, const float* __restrict__ scale_shift,
                                                         T* __restrict__ z, int zpitch, int act, SlopeArg sl, int voxels, int C,
                                                         uint32_t* __restrict__ amax) {
  constexpr int VW = 16 / sizeof(T);
  const float slope = sl.p ? *sl.p : sl.v;
  const int n = blockIdx.y;
  const int cv = C / VW;
  const int vl_n = blockDim.x / cv;  // voxel lanes per block
  const int mycv = threadIdx.x % cv, myvl = threadIdx.x / cv;
  const bool live = myvl < vl_n;
  const int c0 = mycv * VW;
  float sc[VW], sh[VW];
#pragma unroll
  for (int j = 0; j < VW; ++j) {
    sc[j] = scale_shift[((size_t)n * C + c0 + j) * 2];
    sh[j] = scale_shift[((size_t)n * C + c0 + j) * 2 + 1];
  }
  const T* yb = y + (size_t)n * voxels * ypitch + c0;
  T* zb = z + (size_t)n * voxels * zpitch + c0;
  const int stride = gridDim.x * vl_n;
  float mx = 0.f;
  auto run = [&](auto relu_) {
    constexpr bool RELU = decltype(relu_)::value;
    auto body = [&](float* a) {
#pragma unroll
      for (int j = 0; j < VW; ++j) {
        const float t = a[j] * sc[j] + sh[j];
        if constexpr (RELU) a[j] = __builtin_fmaxf(t, 0.f);
        else a[j] = act_fwd<HEAVY>(t, act, slope);
      }
      if (amax) {
#pragma unroll
        for (int j = 0; j < VW; j += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, __builtin_fabsf(a[j])), __builtin_fabsf(a[j + 1]));
      }
    };
    int vox = live ? blockIdx.x * vl_n + myvl : voxels;
    for (; vox + stride < voxels; vox += 2 * stride) {
      float a0[VW], a1[VW];
      vload<T, VW, NT>(yb + (size_t)vox * ypitch, a0);
      vload<T, VW, NT>(yb + (size_t)(vox + stride) * ypitch, a1);
      body(a0);
      body(a1);
      vstore<T, VW, NT>(zb + (size_t)vox * zpitch, a0);
      vstore<T, VW, NT>(zb + (size_t)(vox + stride) * zpitch, a1);
    }
    if (vox < voxels) {
      float a0[VW];
      vload<T, VW, NT>(yb + (size_t)vox * ypitch, a0);
      body(a0);
      vstore<T, VW, NT>(zb + (size_t)vox * zpitch, a0);
    }
  };
  if (!HEAVY && act == BRATS_ACT_RELU) run(std::true_type{});
  else run(std::false_type{});
  if (amax) record_absmax<T>(mx, amax);
}

static inline int stream_grid(size_t total, int block) {
  size_t b = (total + block - 1) / block;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

extern "C" int BRATS_API(brats_affine_act_fwd)(const void* y, int ypitch, const float* scale_shift, void* z, int zpitch,
                                    int dtype, int act, float slope_value, const float* slope_dev, int N, int voxels, int C,
                                    float* amax, brats_stream_t s) {
  const SlopeArg slope{slope_value, slope_dev};
  const int vw = dtype == BRATS_BF16 ? 8 : 4;
  if (!y || !z || !scale_shift || C % vw || ypitch % vw || zpitch % vw || C / vw > 256)
    BRATS_FAIL(BRATS_E_ARG, "affine_act_fwd: C and pitches must be multiples of %d (C <= %d)", vw, 256 * vw);
  const int vl = 256 / (C / vw);
  const int gx = (voxels + vl * 8 - 1) / (vl * 8);
  const bool big = big_tensor(dtype, (size_t)N * voxels * C);
  const int cap = big ? 8192 : 2048;
  dim3 grid(gx < 1 ? 1 : (gx > cap ? cap : gx), N);  // large tensors: many short-lived blocks stream faster (scripts/probes/stream_rw.hip)
  hipStream_t st = (hipStream_t)s;
  uint32_t* am = (uint32_t*)amax;
  if (dtype == BRATS_BF16 && act <= BRATS_ACT_LEAKY && big) {
    hipLaunchKernelGGL((affine_act_kernel<bf16_t, false, true>), grid, dim3(256), 0, st, (const bf16_t*)y, ypitch, scale_shift, (bf16_t*)z,
                       zpitch, act, slope, voxels, C, am);
  } else if (dtype != BRATS_BF16 && act <= BRATS_ACT_LEAKY && big) {  // (split-precision mode: f32 tensors of 805 MB at the 128^3 level)
    hipLaunchKernelGGL((affine_act_kernel<float, false, true>), grid, dim3(256), 0, st, (const float*)y, ypitch, scale_shift, (float*)z,
                       zpitch, act, slope, voxels, C, am);
  } else if (act > BRATS_ACT_LEAKY) {
    if (dtype == BRATS_BF16)
      hipLaunchKernelGGL((affine_act_kernel<bf16_t, true>), grid, dim3(256), 0, st, (const bf16_t*)y, ypitch, scale_shift, (bf16_t*)z,
                         zpitch, act, slope, voxels, C, am);
    else
      hipLaunchKernelGGL((affine_act_kernel<float, true>), grid, dim3(256), 0, st, (const float*)y, ypitch, scale_shift, (float*)z,
                         zpitch, act, slope, voxels, C, am);
  } else {
    if (dtype == BRATS_BF16)
      hipLaunchKernelGGL((affine_act_kernel<bf16_t, false>), grid, dim3(256), 0, st, (const bf16_t*)y, ypitch, scale_shift, (bf16_t*)z,
                         zpitch, act, slope, voxels, C, am);
    else
      hipLaunchKernelGGL((affine_act_kernel<float, false>), grid, dim3(256), 0, st, (const float*)y, ypitch, scale_shift, (float*)z,
                         zpitch, act, slope, voxels, C, am);
  }
  BRATS_CHECK_LAUNCH();
  return 0;
}

// 16 bytes of a tensor kept as loaded (4 registers) until the arithmetic wants the 8 (4) floats: the head-fold passes hold four
// voxels per thread in flight, unpacked up front they cost 32 registers and an occupancy step
template <typename T> struct Raw16;
template <> struct Raw16<bf16_t> {
  typedef u32x4 type;
  template <bool NT> static DEVI type load(const bf16_t* p) {
    if constexpr (NT) return __builtin_nontemporal_load((const u32x4*)p); else return *(const u32x4*)p;
  }
  static DEVI void unpack(const type& v, float* o) {
#pragma unroll
    for (int i = 0; i < 4; ++i) unpack2(v[i], o[2 * i], o[2 * i + 1]);
  }
};
template <> struct Raw16<float> {
  typedef f32x4 type;
  template <bool NT> static DEVI type load(const float* p) { return *(const f32x4*)p; }
  static DEVI void unpack(const type& v, float* o) { o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3]; }
};

// ---- z = act(y*scale + shift) AND its 2x2x2 max pool (MaxAvgPool: [max | mean]) in one pass ---------------------------
// The encoder levels end with ConvBnRelu -> MaxPool3d(2, 2) (networks/equiunet2020.py:469-475): z is written for the skip
// connection and read right back by the pooling kernel (403 MB at 2 x 48 x 128^3).  Here a block walks (row pair, x segment)
// items: a thread owns one x position x one 16-byte channel vector and the 2 x 2 (z, y) voxels above it -- 4 loads and 4
// stores, each a contiguous run of the row across the lanes --; the thread of the even x gets its right-hand neighbour's four
// values through LDS (the 6 lanes of a voxel may straddle a wave) and writes the pooled voxel.  Pooling sees the values as
// stored (rounded to T), takes the max by the pooling kernel's rule (strict '>' in d, h, w order, NaN propagates) and the
// mean in the same order: bit-identical to brats_affine_act_fwd + brats_maxpool2_fwd.  relu / leakyrelu.
// (A first form gave a thread the whole window: its stores were 96-byte pieces at a 192-byte stride -- half-written lines
// under the non-temporal hint -- and the fused pass was SLOWER than the two it replaced, 254 against 148 + 85 us.)
template <typename T, bool NT = false, bool AM = false>
__global__ void __launch_bounds__(256) affine_act_pool_kernel(const T* __restrict__ y, int ypitch, const float* __restrict__ scale_shift,
                                                              T* __restrict__ z, int zpitch, T* __restrict__ p, int ppitch, int act,
                                                              SlopeArg sl, int D, int H, int W, int C, int with_avg,
                                                              uint32_t* __restrict__ amax, uint8_t* __restrict__ argmax) {
  constexpr int VW = 16 / sizeof(T);
  typedef typename Raw16<T>::type raw_t;
  extern __shared__ __attribute__((aligned(16))) char pool_lds[];  // [2][blockDim.x][4] raw_t
  raw_t* xch = (raw_t*)pool_lds;
  const float slope = sl.p ? *sl.p : sl.v;
  const bool relu = act == BRATS_ACT_RELU;
  const int n = blockIdx.y;
  const int cv = C / VW;
  int xb = (blockDim.x / cv) & ~1;  // x positions per item: even, so that a pooling pair never straddles two items
  if (xb > W) xb = W;
  const int mycv = threadIdx.x % cv, myvl = threadIdx.x / cv, c0 = mycv * VW;
  const bool lane_on = myvl < xb;
  const int Do = D / 2, Ho = H / 2, Wo = W / 2;
  const int segs = (W + xb - 1) / xb;
  const size_t items = (size_t)Do * Ho * segs, voxels = (size_t)D * H * W;
  float sc[VW], sh[VW];
#pragma unroll
  for (int j = 0; j < VW; ++j) {
    sc[j] = lane_on ? scale_shift[((size_t)n * C + c0 + j) * 2] : 0.f;
    sh[j] = lane_on ? scale_shift[((size_t)n * C + c0 + j) * 2 + 1] : 0.f;
  }
  const T* yb = y + (size_t)n * voxels * ypitch + c0;
  T* zb = z + (size_t)n * voxels * zpitch + c0;
  T* pb = p + (size_t)n * Do * Ho * Wo * ppitch + c0;
  float amx = 0.f;
  int buf = 0;
  for (size_t item = blockIdx.x; item < items; item += gridDim.x, buf ^= 1) {
    const int seg = (int)(item % segs);
    const size_t rp = item / segs;
    const int yo = (int)(rp % Ho), zo = (int)(rp / Ho);
    const int x = seg * xb + myvl;
    const bool on = lane_on && x < W;
    raw_t mine[4];
    if (on) {
      raw_t raw[4];
      size_t vox[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {  // k = (dz, dy)
        vox[k] = ((size_t)(2 * zo + (k >> 1)) * H + (2 * yo + (k & 1))) * W + x;
        raw[k] = Raw16<T>::template load<NT>(yb + vox[k] * ypitch);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float a[VW];
        Raw16<T>::unpack(raw[k], a);
#pragma unroll
        for (int j = 0; j < VW; ++j) {
          const float pre = a[j] * sc[j] + sh[j];
          const float neg = relu ? 0.f : pre * slope;
          a[j] = pre > 0.f ? pre : neg;
        }
        vstore<T, VW, NT>(zb + vox[k] * zpitch, a);
        // the values as stored, packed: what the pooling half reads (its own and its neighbour's)
        if constexpr (std::is_same<T, float>::value) {
          mine[k] = raw_t{a[0], a[1], a[2], a[3]};
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) mine[k][q] = pack2(a[2 * q], a[2 * q + 1]);
        }
        if (amax) {
          float r[VW];
          Raw16<T>::unpack(mine[k], r);
#pragma unroll
          for (int j = 0; j < VW; ++j) amx = __builtin_fmaxf(amx, __builtin_fabsf(r[j]));
        }
        xch[((size_t)buf * blockDim.x + threadIdx.x) * 4 + k] = mine[k];
      }
    }
    __syncthreads();  // (two buffers: the next item's writes cannot overtake this item's reads)
    if (on && !(myvl & 1)) {
      float mx[VW], sm[VW];
      int am[VW];  // the first arg-max in d, h, w order (torch's tie rule), 0..7: what the pooling backward recomputes
#pragma unroll
      for (int j = 0; j < VW; ++j) { mx[j] = -INFINITY; sm[j] = 0.f; am[j] = 0; }
#pragma unroll
      for (int k = 0; k < 4; ++k) {  // pooling order: d, h, then w = (own, right-hand neighbour)
        float a[VW], b[VW];
        Raw16<T>::unpack(mine[k], a);
        Raw16<T>::unpack(xch[((size_t)buf * blockDim.x + threadIdx.x + cv) * 4 + k], b);
#pragma unroll
        for (int j = 0; j < VW; ++j) {
          if constexpr (AM) {
            if (a[j] > mx[j] || a[j] != a[j]) { mx[j] = a[j]; am[j] = 2 * k; }
            if (b[j] > mx[j] || b[j] != b[j]) { mx[j] = b[j]; am[j] = 2 * k + 1; }
          } else {
            mx[j] = (a[j] > mx[j] || a[j] != a[j]) ? a[j] : mx[j];
            mx[j] = (b[j] > mx[j] || b[j] != b[j]) ? b[j] : mx[j];
          }
          sm[j] = (sm[j] + a[j]) + b[j];
        }
      }
      const size_t pvox = ((size_t)zo * Ho + yo) * Wo + (x >> 1);
      T* po = pb + pvox * ppitch;
      if constexpr (AM) {  // one byte per (pooled voxel, channel): brats_maxpool2_bwd_idx reads these instead of the 8 window voxels
        uint32_t w[VW / 4];
#pragma unroll
        for (int q = 0; q < VW / 4; ++q) w[q] = am[4 * q] | (am[4 * q + 1] << 8) | (am[4 * q + 2] << 16) | (am[4 * q + 3] << 24);
        uint32_t* ap = (uint32_t*)(argmax + ((size_t)n * Do * Ho * Wo + pvox) * C + c0);
#pragma unroll
        for (int q = 0; q < VW / 4; ++q) ap[q] = w[q];
      }
      Vec<T, VW>::store(po, mx);
      if (with_avg) {
#pragma unroll
        for (int j = 0; j < VW; ++j) sm[j] *= 0.125f;
        Vec<T, VW>::store(po + C, sm);
      }
    }
  }
  if (amax) record_absmax<T>(amx, amax);
}

extern "C" int BRATS_API(brats_affine_act_pool_fwd)(const void* y, int ypitch, const float* scale_shift, void* z, int zpitch, void* pooled,
                                         int ppitch, unsigned char* argmax, int dtype, int act, float slope_value,
                                         const float* slope_dev, int N, int D, int H, int W, int C, int with_avg, float* amax,
                                         brats_stream_t s) {
  const SlopeArg slope{slope_value, slope_dev};
  const int vw = dtype == BRATS_BF16 ? 8 : 4;
  if (!y || !z || !pooled || !scale_shift || C % vw || ypitch % vw || zpitch % vw || ppitch % vw || C / vw > 256 || ((D | H | W) & 1))
    BRATS_FAIL(BRATS_E_ARG, "affine_act_pool_fwd: C / pitches multiples of %d (C <= %d), even spatial dims", vw, 256 * vw);
  if (act > BRATS_ACT_LEAKY) BRATS_FAIL(BRATS_E_UNSUPPORTED, "affine_act_pool_fwd: relu / leakyrelu only (act=%d)", act);
  int xb = (256 / (C / vw)) & ~1;
  if (xb < 2) BRATS_FAIL(BRATS_E_UNSUPPORTED, "affine_act_pool_fwd: C=%d leaves no pooling pair per block", C);
  if (xb > W) xb = W;
  const size_t items = (size_t)(D / 2) * (H / 2) * ((W + xb - 1) / xb);
  const bool big = dtype == BRATS_BF16 && stream_nt((size_t)N * D * H * W * C * 2);
  const size_t cap = big ? 8192 : 2048;
  dim3 grid((unsigned)(items < 1 ? 1 : (items > cap ? cap : items)), N);
  const size_t lds = (size_t)2 * 256 * 4 * 16;
  hipStream_t st = (hipStream_t)s;
  uint32_t* am = (uint32_t*)amax;
#define POOL_GO(T, NT, AM) hipLaunchKernelGGL((affine_act_pool_kernel<T, NT, AM>), grid, dim3(256), lds, st, (const T*)y, ypitch, scale_shift, \
                                              (T*)z, zpitch, (T*)pooled, ppitch, act, slope, D, H, W, C, with_avg, am, argmax)
  if (argmax) {
    if (big) POOL_GO(bf16_t, true, true); else if (dtype == BRATS_BF16) POOL_GO(bf16_t, false, true); else POOL_GO(float, false, true);
  } else {
    if (big) POOL_GO(bf16_t, true, false); else if (dtype == BRATS_BF16) POOL_GO(bf16_t, false, false); else POOL_GO(float, false, false);
  }
#undef POOL_GO
  BRATS_CHECK_LAUNCH();
  return 0;
}

// ---- backward of z = act(GN(y)) ----------------------------------------------------------------
// pass 1: red[n][c] = { sum_v u, sum_v u*xhat },  u = dz * act'(y*scale+shift), xhat = (y-mean_g)*rstd_g
// A thread owns one 16-byte channel vector (fixed for the whole kernel) and walks voxels: its per-channel constants
// live in registers (read from LDS per element they made both passes LDS-bound at ~3.5 TB/s), two voxels are in
// flight per iteration.
// Block reduction over the voxel lanes of a per-thread channel vector, ONE value plane at a time through scr[vl_n][C]: the
// passes that carry 5 - 8 planes (EvoNorm's five sums, the head's K weight-gradient planes) would otherwise hold
// vl_n * C * planes floats of LDS (65 KB at C = 48) for their last microsecond and lose occupancy for the whole stream.
// put(c, sum over lanes in lane order).  Starts with a barrier (scr may still be read by the previous plane).
template <int VW, typename F>
DEVI void lane_reduce_plane(float* scr, const float* a, bool live, int myvl, int vl_n, int C, int c0, F&& put) {
  __syncthreads();
  if (live) {
#pragma unroll
    for (int j = 0; j < VW; ++j) scr[myvl * C + c0 + j] = a[j];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float t = 0.f;
    for (int l = 0; l < vl_n; ++l) t += scr[l * C + c];
    put(c, t);
  }
}

// HK > 0 (brats_gn_act_bwd_head): the layer's output feeds ONLY a 1x1x1 head convolution (the network's last layer), so its
// gradient is dz[v][c] = sum_k dl[k][v] * w[k][c] with K = HK logit planes: computed here from the 12 bytes of dl per voxel
// instead of being written (bf16, 2 * C bytes per voxel) by brats_head_bwd and read back by both passes; the head's own
// weight / bias gradients, sum_v dl[k][v] * z[v][c] and sum_v dl[k][v], come out of pass 1, which has z = act(pre) at hand.
struct HeadFold {
  const float* dl = nullptr;  // [N][HK][voxels]
  const float* w = nullptr;   // [HK][C]
  float* hpart = nullptr;     // pass 1: per-block partials [n * gridDim.x + block][HK * C + HK]
  // HK == -1 (brats_gn_act_bwd_pool): the layer ends an encoder level -- its output gradient is the skip connection's plus the
  // max pool's, dz[v][c] = dskip[v][c] + (argmax[pv][c] == k(v) ? dpool[pv][c] : 0) with pv the voxel's 2x2x2 window and k(v)
  // its index inside it: computed here from the pieces instead of being written by the pooling backward (403 MB at the
  // 128^3 level) and read back by both passes.  Same arithmetic as maxpool2_bwd_idx_kernel ((0 + sel) + skip).
  const void* dskip = nullptr;
  const void* dpool = nullptr;
  const uint8_t* argmax = nullptr;  // [N][D/2][H/2][W/2][C]
  int dskip_pitch = 0, dpool_pitch = 0, D = 0, H = 0, W = 0;
  int with_avg = 0;  // MaxAvgPool: dpool holds [max | mean] gradients (2C channels), the mean's share is dpool[C + c] / 8
};
// the pool-source gradient of one voxel: its channel vector from the three pieces (loads, then the arithmetic)
template <typename T, int VW>
DEVI void pool_dz(const HeadFold& hf, int n, size_t vox, int c0, int C, typename Raw16<T>::type& skr, typename Raw16<T>::type& dpr,
                  typename Raw16<T>::type& avr, uint32_t (&aw)[VW / 4], int& kk) {
  const int x = (int)(vox % hf.W);
  const size_t t = vox / hf.W;
  const int yy = (int)(t % hf.H), z = (int)(t / hf.H);
  const size_t pvox = (((size_t)n * (hf.D / 2) + (z >> 1)) * (hf.H / 2) + (yy >> 1)) * (hf.W / 2) + (x >> 1);
  kk = ((z & 1) << 2) | ((yy & 1) << 1) | (x & 1);
  skr = Raw16<T>::template load<false>((const T*)hf.dskip + ((size_t)n * hf.D * hf.H * hf.W + vox) * hf.dskip_pitch + c0);
  dpr = Raw16<T>::template load<false>((const T*)hf.dpool + pvox * hf.dpool_pitch + c0);
  if (hf.with_avg) avr = Raw16<T>::template load<false>((const T*)hf.dpool + pvox * hf.dpool_pitch + C + c0);
  else avr = dpr;  // (unused)
#pragma unroll
  for (int q = 0; q < VW / 4; ++q) aw[q] = ((const uint32_t*)(hf.argmax + pvox * C + c0))[q];
}
template <typename T, int VW>
DEVI void pool_dz_finish(const typename Raw16<T>::type& skr, const typename Raw16<T>::type& dpr, const typename Raw16<T>::type& avr,
                         int with_avg, const uint32_t (&aw)[VW / 4], int kk, float* g) {
  float sk[VW], dp[VW], ga[VW];
  Raw16<T>::unpack(skr, sk);
  Raw16<T>::unpack(dpr, dp);
  Raw16<T>::unpack(avr, ga);
#pragma unroll
  for (int j = 0; j < VW; ++j) {
    const int am = (aw[j >> 2] >> (8 * (j & 3))) & 0xff;
    float o = (with_avg ? ga[j] * 0.125f : 0.f) + (am == kk ? dp[j] : 0.f);
    o += sk[j];
    g[j] = o;
  }
}
template <typename T, bool HEAVY, bool NT = false, int HK = 0>
__global__ void __launch_bounds__(256) gn_bwd_reduce_kernel(const T* __restrict__ dz, int dzpitch, const T* __restrict__ y,
                                                            int ypitch, const float* __restrict__ scale_shift,
                                                            const float* __restrict__ mean_rstd, float* __restrict__ red, int act,
                                                            SlopeArg sl, int voxels, int C, int groups, HeadFold hf) {
  constexpr int VW = 16 / sizeof(T);
  constexpr int HKA = HK > 0 ? HK : 1;
  const float slope = sl.p ? *sl.p : sl.v;
  extern __shared__ float sm[];  // reduction scratch [vl_n][C][2] (+ [vl_n][HK][C] + [vl_n][HK])
  const int n = blockIdx.y;
  const int cpg = C / groups;
  const int cv = C / VW;
  const int vl_n = blockDim.x / cv;  // voxel lanes per block
  const int mycv = threadIdx.x % cv, myvl = threadIdx.x / cv;
  const int c0 = mycv * VW;
  float a1[VW], a2[VW];
  float aw[HKA][VW], ab[HKA];
#pragma unroll
  for (int j = 0; j < VW; ++j) a1[j] = a2[j] = 0.f;
#pragma unroll
  for (int k = 0; k < HKA; ++k) {
    ab[k] = 0.f;
#pragma unroll
    for (int j = 0; j < VW; ++j) aw[k][j] = 0.f;
  }
  if (myvl < vl_n) {
    float sc[VW], sh[VW], rs[VW], mo[VW];  // pre = y*sc + sh ; xhat = y*rs + mo
#pragma unroll
    for (int j = 0; j < VW; ++j) {
      const int c = c0 + j;
      sc[j] = scale_shift[((size_t)n * C + c) * 2];
      sh[j] = scale_shift[((size_t)n * C + c) * 2 + 1];
      const float mean = mean_rstd[(n * groups + c / cpg) * 2], rstd = mean_rstd[(n * groups + c / cpg) * 2 + 1];
      rs[j] = rstd;
      mo[j] = -mean * rstd;
    }
    const T* dzb = dz + (size_t)n * voxels * dzpitch + c0;
    const T* yb = y + (size_t)n * voxels * ypitch + c0;
    const size_t stride = (size_t)gridDim.x * vl_n;
    size_t vox = (size_t)blockIdx.x * vl_n + myvl;
    if constexpr (HK < 0) {
      const float nslope = act == BRATS_ACT_RELU ? 0.f : slope;
      auto step = [&](size_t v0, auto cnt) {  // two voxels (2 x 3 loads of 16 bytes + the arg-max bytes) in flight per thread
        constexpr int NVX = decltype(cnt)::value;
        typename Raw16<T>::type yr[NVX], skr[NVX], dpr[NVX], avr[NVX];
        uint32_t aw[NVX][VW / 4];
        int kk[NVX];
#pragma unroll
        for (int i = 0; i < NVX; ++i) {
          yr[i] = Raw16<T>::template load<NT>(yb + (v0 + i * stride) * ypitch);
          pool_dz<T, VW>(hf, n, v0 + i * stride, c0, C, skr[i], dpr[i], avr[i], aw[i], kk[i]);
        }
#pragma unroll
        for (int i = 0; i < NVX; ++i) {
          float yy[VW], g[VW];
          Raw16<T>::unpack(yr[i], yy);
          pool_dz_finish<T, VW>(skr[i], dpr[i], avr[i], hf.with_avg, aw[i], kk[i], g);
#pragma unroll
          for (int j = 0; j < VW; ++j) {
            const float pre = yy[j] * sc[j] + sh[j];
            const float u = g[j] * (pre > 0.f ? 1.f : nslope);
            a1[j] += u;
            a2[j] += u * (yy[j] * rs[j] + mo[j]);
          }
        }
      };
      for (; vox + stride < (size_t)voxels; vox += 2 * stride) step(vox, std::integral_constant<int, 2>{});
      for (; vox < (size_t)voxels; vox += stride) step(vox, std::integral_constant<int, 1>{});
    } else if constexpr (HK > 0) {
      float wr[HKA][VW];
#pragma unroll
      for (int k = 0; k < HK; ++k)
#pragma unroll
        for (int j = 0; j < VW; ++j) wr[k][j] = hf.w[k * C + c0 + j];
      const float* dl = hf.dl + (size_t)n * HK * voxels;
      const float nslope = act == BRATS_ACT_RELU ? 0.f : slope;
      // a2 = sum u * xhat = rs * sum(u * y) + mo * sum(u): the loop keeps sum(u * y), the affine map is applied once after it
      auto hbody = [&](const float* g, const float* yy) {
#pragma unroll
        for (int j = 0; j < VW; ++j) {
          float d = 0.f;
#pragma unroll
          for (int k = 0; k < HK; ++k) d += g[k] * wr[k][j];
          const float pre = yy[j] * sc[j] + sh[j];
          const bool pos = pre > 0.f;  // relu = leakyrelu with slope 0: selects, no branches on `act` per element
          const float u = d * (pos ? 1.f : nslope);
          a1[j] += u;
          a2[j] += u * yy[j];
          const float z = pos ? pre : pre * nslope;
#pragma unroll
          for (int k = 0; k < HK; ++k) aw[k][j] += g[k] * z;
        }
#pragma unroll
        for (int k = 0; k < HK; ++k) ab[k] += g[k];
      };
      // four voxels (4 x 16 bytes of y + 4 x HK floats of dl) in flight per thread; a wave's loads cover adjacent voxels.
      // (With two voxels in flight the pass was latency-bound at 1.5 TB/s.)
      auto step = [&](size_t v0, auto cnt) {
        constexpr int NV = decltype(cnt)::value;
        typename Raw16<T>::type yr[NV];
        float gq[NV][HKA];
#pragma unroll
        for (int i = 0; i < NV; ++i) yr[i] = Raw16<T>::template load<NT>(yb + (v0 + i * stride) * ypitch);
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
          for (int k = 0; k < HK; ++k) gq[i][k] = dl[(size_t)k * voxels + v0 + i * stride];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          float yy[VW];
          Raw16<T>::unpack(yr[i], yy);
          hbody(gq[i], yy);
        }
      };
      for (; vox + 3 * stride < (size_t)voxels; vox += 4 * stride) step(vox, std::integral_constant<int, 4>{});
      for (; vox < (size_t)voxels; vox += stride) step(vox, std::integral_constant<int, 1>{});
#pragma unroll
      for (int j = 0; j < VW; ++j) a2[j] = a2[j] * rs[j] + mo[j] * a1[j];
    } else {
    auto body = [&](const float* g, const float* yy) {
#pragma unroll
      for (int j = 0; j < VW; ++j) {
        const float u = g[j] * act_grad<HEAVY>(yy[j] * sc[j] + sh[j], act, slope);
        a1[j] += u;
        a2[j] += u * (yy[j] * rs[j] + mo[j]);
      }
    };
    for (; vox + 3 * stride < (size_t)voxels; vox += 4 * stride) {  // 4 voxels (8 loads) in flight per thread
      float g0[VW], y0[VW], g1[VW], y1[VW], g2[VW], y2[VW], g3[VW], y3[VW];
      vload<T, VW, NT>(dzb + vox * dzpitch, g0);
      vload<T, VW, NT>(yb + vox * ypitch, y0);
      vload<T, VW, NT>(dzb + (vox + stride) * dzpitch, g1);
      vload<T, VW, NT>(yb + (vox + stride) * ypitch, y1);
      vload<T, VW, NT>(dzb + (vox + 2 * stride) * dzpitch, g2);
      vload<T, VW, NT>(yb + (vox + 2 * stride) * ypitch, y2);
      vload<T, VW, NT>(dzb + (vox + 3 * stride) * dzpitch, g3);
      vload<T, VW, NT>(yb + (vox + 3 * stride) * ypitch, y3);
      body(g0, y0);
      body(g1, y1);
      body(g2, y2);
      body(g3, y3);
    }
    for (; vox < (size_t)voxels; vox += stride) {
      float g0[VW], y0[VW];
      vload<T, VW, NT>(dzb + vox * dzpitch, g0);
      vload<T, VW, NT>(yb + vox * ypitch, y0);
      body(g0, y0);
    }
    }
  }
  // block reduction over voxel lanes through LDS, then one atomic per channel per block
  float* scr = sm;  // [vl_n][C][2]
  if (myvl < vl_n) {
#pragma unroll
    for (int j = 0; j < VW; ++j) {
      scr[(myvl * C + c0 + j) * 2] = a1[j];
      scr[(myvl * C + c0 + j) * 2 + 1] = a2[j];
    }
  }
  __syncthreads();
  // per-block partial sums, combined in block order by gn_bwd_finish_kernel (bitwise reproducible; no atomics, no memset)
  float* part = red + (size_t)gridDim.y * C * 2 + ((size_t)blockIdx.x * gridDim.y + n) * C * 2;
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
    float t = 0.f;
    for (int l = 0; l < vl_n; ++l) t += scr[l * C * 2 + i];
    part[i] = t;
  }
  if constexpr (HK > 0) {  // the head's weight / bias gradient partials: [HK][C] + [HK] per block, added in block order afterwards
    const bool live = myvl < vl_n;
    float* hb = sm + vl_n * C * 2;  // [vl_n][HK]
    if (live && mycv == 0) {
#pragma unroll
      for (int k = 0; k < HK; ++k) hb[myvl * HK + k] = ab[k];
    }
    float* hp = hf.hpart + ((size_t)n * gridDim.x + blockIdx.x) * (HK * C + HK);
#pragma unroll
    for (int k = 0; k < HK; ++k)  // (one plane at a time through the first vl_n * C floats of the scratch)
      lane_reduce_plane<VW>(sm, aw[k], live, myvl, vl_n, C, c0, [&](int c, float t) { hp[k * C + c] = t; });
    if ((int)threadIdx.x < HK) {
      float t = 0.f;
      for (int l = 0; l < vl_n; ++l) t += hb[l * HK + threadIdx.x];
      hp[HK * C + threadIdx.x] = t;
    }
  }
}

// red[n][i] = sum over the nb blocks of pass 1 in a fixed order.  A block = 8 entries x 32 slices (slice g adds blocks
// g, g+32, ...; the 32 slice sums are then added in slice order).
__global__ void __launch_bounds__(256) gn_bwd_finish_kernel(float* __restrict__ red, int nb, int total /* N*2C */) {
  const int e = threadIdx.x & 7, g = threadIdx.x >> 3;
  const int i = blockIdx.x * 8 + e;
  const float* part = red + total;
  float s = 0.f;
  if (i < total) {
    // eight partials in flight, added in block order (the plain loop compiles to one round trip per partial)
    const float* src = part + i;
    int b = g;
    for (; b + 7 * 32 < nb; b += 8 * 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(b + 32 * u) * total];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < nb; b += 32) s += src[(size_t)b * total];
  }
  __shared__ float sm[32][8];
  sm[g][e] = s;
  __syncthreads();
  if (g == 0 && i < total) {
#pragma unroll
    for (int k = 1; k < 32; ++k) s += sm[k][e];
    red[i] = s;
  }
}

// pass 2: dy = rstd*(u*gamma - m1 - xhat*m2) = u*A + y*B + K with per-channel A = rstd*gamma, B = -rstd^2*m2,
// K = rstd*(mean*rstd*m2 - m1);  block (0,0) also finishes dgamma/dbeta
template <typename T, bool HEAVY, bool NT = false, int HK = 0>
__global__ void __launch_bounds__(256) gn_bwd_apply_kernel(const T* __restrict__ dz, int dzpitch, const T* __restrict__ y,
                                                           int ypitch, const float* __restrict__ scale_shift,
                                                           const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ red, T* __restrict__ dy, int dypitch,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int act,
                                                           SlopeArg sl, int N, int voxels, int C, int groups,
                                                           uint32_t* __restrict__ amax, HeadFold hf) {
  constexpr int VW = 16 / sizeof(T);
  constexpr int HKA = HK > 0 ? HK : 1;
  const float slope = sl.p ? *sl.p : sl.v;
  extern __shared__ float sm[];
  float* m12 = sm;  // [groups][2]: m1, m2 per group
  const int n = blockIdx.y;
  const int cpg = C / groups;
  const float invM = 1.f / ((float)cpg * (float)voxels);
  for (int g = threadIdx.x; g < groups; g += blockDim.x) {
    float t1 = 0.f, t2 = 0.f;
    for (int j = 0; j < cpg; ++j) {
      const int c = g * cpg + j;
      t1 += gamma[c] * red[((size_t)n * C + c) * 2];
      t2 += gamma[c] * red[((size_t)n * C + c) * 2 + 1];
    }
    m12[g * 2] = t1 * invM;
    m12[g * 2 + 1] = t2 * invM;
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && dgamma) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      float b = 0.f, g = 0.f;
      for (int m = 0; m < N; ++m) { b += red[((size_t)m * C + c) * 2]; g += red[((size_t)m * C + c) * 2 + 1]; }
      dbeta[c] = b;
      dgamma[c] = g;
    }
  }
  __syncthreads();
  const int cv = C / VW;
  const int vl_n = blockDim.x / cv;
  const int mycv = threadIdx.x % cv, myvl = threadIdx.x / cv;
  const bool live = myvl < vl_n;  // idle threads stay for the |max| wave reduction at the end
  const int c0 = mycv * VW;
  float sc[VW], sh[VW], ca[VW], cb[VW], ck[VW];
#pragma unroll
  for (int j = 0; j < VW; ++j) {
    const int c = c0 + j, g = c / cpg;
    sc[j] = scale_shift[((size_t)n * C + c) * 2];
    sh[j] = scale_shift[((size_t)n * C + c) * 2 + 1];
    const float mean = mean_rstd[(n * groups + g) * 2], rstd = mean_rstd[(n * groups + g) * 2 + 1];
    const float m1 = m12[g * 2], m2 = m12[g * 2 + 1];
    ca[j] = rstd * gamma[c];
    cb[j] = -rstd * rstd * m2;
    ck[j] = rstd * (mean * rstd * m2 - m1);
  }
  const T* dzb = dz + (size_t)n * voxels * dzpitch + c0;
  const T* yb = y + (size_t)n * voxels * ypitch + c0;
  T* dyb = dy + (size_t)n * voxels * dypitch + c0;
  const size_t stride = (size_t)gridDim.x * vl_n;
  size_t vox = live ? (size_t)blockIdx.x * vl_n + myvl : (size_t)voxels;
  float mx = 0.f;
  auto body = [&](const float* g, const float* yy, float* o) {
#pragma unroll
    for (int j = 0; j < VW; ++j) {
      const float u = g[j] * act_grad<HEAVY>(yy[j] * sc[j] + sh[j], act, slope);
      o[j] = u * ca[j] + (yy[j] * cb[j] + ck[j]);
    }
    if (amax) {
#pragma unroll
      for (int j = 0; j < VW; j += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, __builtin_fabsf(o[j])), __builtin_fabsf(o[j + 1]));
    }
  };
  if constexpr (HK < 0) {
    const float nslope = act == BRATS_ACT_RELU ? 0.f : slope;
    auto step = [&](size_t v0, auto cnt) {
      constexpr int NVX = decltype(cnt)::value;
      typename Raw16<T>::type yr[NVX], skr[NVX], dpr[NVX], avr[NVX];
      uint32_t aw[NVX][VW / 4];
      int kk[NVX];
#pragma unroll
      for (int i = 0; i < NVX; ++i) {
        yr[i] = Raw16<T>::template load<NT>(yb + (v0 + i * stride) * ypitch);
        pool_dz<T, VW>(hf, n, v0 + i * stride, c0, C, skr[i], dpr[i], avr[i], aw[i], kk[i]);
      }
#pragma unroll
      for (int i = 0; i < NVX; ++i) {
        float yy[VW], g[VW], o[VW];
        Raw16<T>::unpack(yr[i], yy);
        pool_dz_finish<T, VW>(skr[i], dpr[i], avr[i], hf.with_avg, aw[i], kk[i], g);
#pragma unroll
        for (int j = 0; j < VW; ++j) {
          const float u = g[j] * (yy[j] * sc[j] + sh[j] > 0.f ? 1.f : nslope);
          o[j] = u * ca[j] + (yy[j] * cb[j] + ck[j]);
        }
        if (amax) {
#pragma unroll
          for (int j = 0; j < VW; j += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, __builtin_fabsf(o[j])), __builtin_fabsf(o[j + 1]));
        }
        vstore<T, VW, NT>(dyb + (v0 + i * stride) * dypitch, o);
      }
    };
    for (; vox + stride < (size_t)voxels; vox += 2 * stride) step(vox, std::integral_constant<int, 2>{});
    for (; vox < (size_t)voxels; vox += stride) step(vox, std::integral_constant<int, 1>{});
  } else if constexpr (HK > 0) {
    float wr[HKA][VW];
#pragma unroll
    for (int k = 0; k < HK; ++k)
#pragma unroll
      for (int j = 0; j < VW; ++j) wr[k][j] = hf.w[k * C + c0 + j];
    const float* dl = hf.dl + (size_t)n * HK * voxels;
    const float nslope = act == BRATS_ACT_RELU ? 0.f : slope;
    auto step = [&](size_t v0, auto cnt) {  // four voxels in flight per thread, as in pass 1
      constexpr int NV = decltype(cnt)::value;
      typename Raw16<T>::type yr[NV];
      float gq[NV][HKA];
#pragma unroll
      for (int i = 0; i < NV; ++i) yr[i] = Raw16<T>::template load<NT>(yb + (v0 + i * stride) * ypitch);
#pragma unroll
      for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int k = 0; k < HK; ++k) gq[i][k] = dl[(size_t)k * voxels + v0 + i * stride];
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        float yy[VW], o[VW];
        Raw16<T>::unpack(yr[i], yy);
#pragma unroll
        for (int j = 0; j < VW; ++j) {
          float d = 0.f;
#pragma unroll
          for (int k = 0; k < HK; ++k) d += gq[i][k] * wr[k][j];
          const float u = d * (yy[j] * sc[j] + sh[j] > 0.f ? 1.f : nslope);
          o[j] = u * ca[j] + (yy[j] * cb[j] + ck[j]);
        }
        if (amax) {
#pragma unroll
          for (int j = 0; j < VW; j += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, __builtin_fabsf(o[j])), __builtin_fabsf(o[j + 1]));
        }
        vstore<T, VW, NT>(dyb + (v0 + i * stride) * dypitch, o);
      }
    };
    for (; vox + 3 * stride < (size_t)voxels; vox += 4 * stride) step(vox, std::integral_constant<int, 4>{});
    for (; vox < (size_t)voxels; vox += stride) step(vox, std::integral_constant<int, 1>{});
  } else {
  for (; vox + stride < (size_t)voxels; vox += 2 * stride) {
    float g0[VW], y0[VW], g1[VW], y1[VW], o0[VW], o1[VW];
    vload<T, VW, NT>(dzb + vox * dzpitch, g0);
    vload<T, VW, NT>(yb + vox * ypitch, y0);
    vload<T, VW, NT>(dzb + (vox + stride) * dzpitch, g1);
    vload<T, VW, NT>(yb + (vox + stride) * ypitch, y1);
    body(g0, y0, o0);
    body(g1, y1, o1);
    vstore<T, VW, NT>(dyb + vox * dypitch, o0);
    vstore<T, VW, NT>(dyb + (vox + stride) * dypitch, o1);
  }
  if (vox < (size_t)voxels) {
    float g0[VW], y0[VW], o0[VW];
    vload<T, VW, NT>(dzb + vox * dzpitch, g0);
    vload<T, VW, NT>(yb + vox * ypitch, y0);
    body(g0, y0, o0);
    vstore<T, VW, NT>(dyb + vox * dypitch, o0);
  }
  }
  if (amax) record_absmax<T>(mx, amax);
}

constexpr int GN_BWD_MAX_BLOCKS = 2048;
extern "C" size_t BRATS_API(brats_gn_bwd_ws_floats)(int N, int C) { return (size_t)(1 + GN_BWD_MAX_BLOCKS) * N * C * 2; }

template <typename T, bool HEAVY, bool NT, int HK>
static void gn_bwd_launch(dim3 g1, dim3 g2, size_t lds1, size_t lds2, hipStream_t st, const void* dz, int dzpitch, const void* y,
                          int ypitch, const float* scale_shift, const float* mean_rstd, const float* gamma, void* dy, int dypitch,
                          float* red, float* dgamma, float* dbeta, int act, SlopeArg slope, int N, int voxels, int C, int groups,
                          float* amax, HeadFold hf) {
  hipLaunchKernelGGL((gn_bwd_reduce_kernel<T, HEAVY, NT, HK>), g1, dim3(256), lds1, st, (const T*)dz, dzpitch, (const T*)y, ypitch,
                     scale_shift, mean_rstd, red, act, slope, voxels, C, groups, hf);
  // pass 1 leaves one partial sum per block; gn_bwd_finish_kernel adds them in block order
  hipLaunchKernelGGL(gn_bwd_finish_kernel, dim3((N * C * 2 + 7) / 8), dim3(256), 0, st, red, (int)g1.x, N * C * 2);
  hipLaunchKernelGGL((gn_bwd_apply_kernel<T, HEAVY, NT, HK>), g2, dim3(256), lds2, st, (const T*)dz, dzpitch, (const T*)y, ypitch,
                     scale_shift, mean_rstd, gamma, red, (T*)dy, dypitch, dgamma, dbeta, act, slope, N, voxels, C, groups,
                     (uint32_t*)amax, hf);
}

static int gn_act_bwd_impl(const void* dz, int dzpitch, const void* y, int ypitch, const float* scale_shift, const float* mean_rstd,
                           const float* gamma, void* dy, int dypitch, float* red, float* dgamma, float* dbeta, int dtype, int act,
                           SlopeArg slope, int N, int voxels, int C, int groups, float* amax, HeadFold hf, int K, hipStream_t st) {
  const int vw = dtype == BRATS_BF16 ? 8 : 4;
  if (C % vw || C % groups || dzpitch % vw || ypitch % vw || dypitch % vw || C / vw > 256)
    BRATS_FAIL(BRATS_E_ARG, "gn_act_bwd: C=%d / pitches must be multiples of %d", C, vw);
  const int cv = C / vw, vl = 256 / cv;
  const int gx = (int)(((size_t)voxels + vl * 8 - 1) / (vl * 8));
  const bool big = big_tensor(dtype, (size_t)N * voxels * C);
  const int cap1 = big ? GN_BWD_MAX_BLOCKS : 512;
  dim3 g1(gx < 1 ? 1 : (gx > cap1 ? cap1 : gx), N);
  const size_t lds1 = (size_t)(vl * C * 2 + (K > 0 ? vl * K : 0)) * sizeof(float);
  dim3 g2(gx < 1 ? 1 : (gx > (big ? 8192 : 2048) ? (big ? 8192 : 2048) : gx), N);
  const size_t lds2 = (size_t)2 * groups * sizeof(float);
#define GN_BWD_GO(T, HEAVY, NT, HK) gn_bwd_launch<T, HEAVY, NT, HK>(g1, g2, lds1, lds2, st, dz, dzpitch, y, ypitch, scale_shift, mean_rstd, \
                                                                  gamma, dy, dypitch, red, dgamma, dbeta, act, slope, N, voxels, C, groups, amax, hf)
  if (K < 0) {  // the pool-source form (relu / leakyrelu: checked by the caller)
    if (big && dtype == BRATS_BF16) GN_BWD_GO(bf16_t, false, true, -1);
    else if (big) GN_BWD_GO(float, false, true, -1);
    else if (dtype == BRATS_BF16) GN_BWD_GO(bf16_t, false, false, -1);
    else GN_BWD_GO(float, false, false, -1);
  } else if (K) {  // (relu / leakyrelu, three logit planes: checked by the caller)
    if (big && dtype == BRATS_BF16) GN_BWD_GO(bf16_t, false, true, 3);
    else if (big) GN_BWD_GO(float, false, true, 3);
    else if (dtype == BRATS_BF16) GN_BWD_GO(bf16_t, false, false, 3);
    else GN_BWD_GO(float, false, false, 3);
  } else if (big && act <= BRATS_ACT_LEAKY && dtype == BRATS_BF16) GN_BWD_GO(bf16_t, false, true, 0);
  else if (big && act <= BRATS_ACT_LEAKY) GN_BWD_GO(float, false, true, 0);
  else if (act > BRATS_ACT_LEAKY) {
    if (dtype == BRATS_BF16) GN_BWD_GO(bf16_t, true, false, 0);
    else GN_BWD_GO(float, true, false, 0);
  } else {
    if (dtype == BRATS_BF16) GN_BWD_GO(bf16_t, false, false, 0);
    else GN_BWD_GO(float, false, false, 0);
  }
#undef GN_BWD_GO
  BRATS_CHECK_LAUNCH();
  return (int)g1.x;  // > 0: the number of pass-1 blocks per sample (the head partials' count)
}

extern "C" int BRATS_API(brats_gn_act_bwd)(const void* dz, int dzpitch, const void* y, int ypitch, const float* scale_shift,
                                const float* mean_rstd, const float* gamma, void* dy, int dypitch, float* red,
                                float* dgamma, float* dbeta, int dtype, int act, float slope_value, const float* slope_dev,
                                int N, int voxels, int C, int groups, float* amax, brats_stream_t s) {
  if (!dz || !y || !dy || !red || !scale_shift || !mean_rstd || !gamma) BRATS_FAIL(BRATS_E_ARG, "gn_act_bwd: null pointer");
  const int rc = gn_act_bwd_impl(dz, dzpitch, y, ypitch, scale_shift, mean_rstd, gamma, dy, dypitch, red, dgamma, dbeta, dtype, act,
                                 SlopeArg{slope_value, slope_dev}, N, voxels, C, groups, amax, HeadFold{}, 0, (hipStream_t)s);
  return rc < 0 ? rc : 0;
}

// GroupNorm + activation backward whose first pass was done by the producer of dz: the input-gradient convolution of the
// block's second unit (brats_conv3d_fwd_bstats) left per tile and channel sum u and sum u * y (u = dz * act'(y * scale + shift),
// y = this unit's raw convolution output) in the tile-statistics layout of the forward kernels.  Here: the tiles are added in a
// fixed order in f64 (the forward pass's slab reduction), sum u * xhat = rstd * (sum u*y - mean * sum u), then pass 2 as in
// brats_gn_act_bwd -- dz and y are read once instead of twice.
__global__ void __launch_bounds__(256) gn_bwd_tiles_finish_kernel(const double* __restrict__ part, int splits, int N, int C, int groups,
                                                                  const float* __restrict__ mean_rstd, float* __restrict__ red) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * C) return;
  const int n = i / C, c = i % C;
  double s1 = 0.0, s2 = 0.0;
  const double* src = part + ((size_t)n * C + c) * 2;
  const size_t zs = (size_t)N * C * 2;
  int z = 0;
  for (; z + 8 <= splits; z += 8) {  // eight slab sums in flight, added in slab order
    double v1[8], v2[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { v1[u] = src[(z + u) * zs]; v2[u] = src[(z + u) * zs + 1]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) { s1 += v1[u]; s2 += v2[u]; }
  }
  for (; z < splits; ++z) { s1 += src[z * zs]; s2 += src[z * zs + 1]; }
  const int g = c / (C / groups);
  const double mean = mean_rstd[(n * groups + g) * 2], rstd = mean_rstd[(n * groups + g) * 2 + 1];
  red[(size_t)i * 2] = (float)s1;
  red[(size_t)i * 2 + 1] = (float)(rstd * (s2 - mean * s1));
}

extern "C" int BRATS_API(brats_gn_act_bwd_tiles)(const float* tile_stats, int tiles_per_sample, const void* dz, int dzpitch, const void* y,
                                      int ypitch, const float* scale_shift, const float* mean_rstd, const float* gamma, void* dy,
                                      int dypitch, float* red, float* dgamma, float* dbeta, int dtype, int act, float slope,
                                      int N, int voxels, int C, int groups, float* amax, brats_stream_t s) {
  if (!tile_stats || !dz || !y || !dy || !red || !scale_shift || !mean_rstd || !gamma || tiles_per_sample <= 0)
    BRATS_FAIL(BRATS_E_ARG, "gn_act_bwd_tiles: null pointer");
  if (dtype != BRATS_BF16 && dtype != BRATS_F32) BRATS_FAIL(BRATS_E_UNSUPPORTED, "gn_act_bwd_tiles: 16-bit or f32 (split-precision mode) activations");
  if (act != BRATS_ACT_RELU && act != BRATS_ACT_LEAKY) BRATS_FAIL(BRATS_E_UNSUPPORTED, "gn_act_bwd_tiles: relu / leakyrelu only (act %d)", act);
  const int vw = dtype == BRATS_BF16 ? 8 : 4;
  if (C % vw || C % groups || dzpitch % vw || ypitch % vw || dypitch % vw || C / vw > 256)
    BRATS_FAIL(BRATS_E_ARG, "gn_act_bwd_tiles: C=%d / pitches must be multiples of %d", C, vw);
  hipStream_t st = (hipStream_t)s;
  // f64 slab partials behind the [N][C][2] totals, inside the workspace brats_gn_bwd_ws_floats() sizes (64 x N x C x 2 doubles)
  double* part = (double*)(red + (((size_t)N * C * 2 + 1) / 2) * 2);
  const int splits = gn_splits(tiles_per_sample);
  hipLaunchKernelGGL(gn_chan_reduce_kernel, dim3((C + 15) / 16, N, splits), dim3(256), 0, st, tile_stats, tiles_per_sample, C, part);
  hipLaunchKernelGGL(gn_bwd_tiles_finish_kernel, dim3((N * C + 255) / 256), dim3(256), 0, st, part, splits, N, C, groups, mean_rstd, red);
  const int cv = C / vw, vl = 256 / cv;
  const int gx = (int)(((size_t)voxels + vl * 8 - 1) / (vl * 8));
  const bool big = big_tensor(dtype, (size_t)N * voxels * C);
  dim3 g2(gx < 1 ? 1 : (gx > (big ? 8192 : 2048) ? (big ? 8192 : 2048) : gx), N);
  const size_t lds2 = (size_t)2 * groups * sizeof(float);
  const SlopeArg sl{slope, nullptr};
  if (dtype == BRATS_F32 && big)
    hipLaunchKernelGGL((gn_bwd_apply_kernel<float, false, true, 0>), g2, dim3(256), lds2, st, (const float*)dz, dzpitch, (const float*)y, ypitch,
                       scale_shift, mean_rstd, gamma, red, (float*)dy, dypitch, dgamma, dbeta, act, sl, N, voxels, C, groups, (uint32_t*)amax, HeadFold{});
  else if (dtype == BRATS_F32)
    hipLaunchKernelGGL((gn_bwd_apply_kernel<float, false, false, 0>), g2, dim3(256), lds2, st, (const float*)dz, dzpitch, (const float*)y, ypitch,
                       scale_shift, mean_rstd, gamma, red, (float*)dy, dypitch, dgamma, dbeta, act, sl, N, voxels, C, groups, (uint32_t*)amax, HeadFold{});
  else if (big)
    hipLaunchKernelGGL((gn_bwd_apply_kernel<bf16_t, false, true, 0>), g2, dim3(256), lds2, st, (const bf16_t*)dz, dzpitch, (const bf16_t*)y, ypitch,
                       scale_shift, mean_rstd, gamma, red, (bf16_t*)dy, dypitch, dgamma, dbeta, act, sl, N, voxels, C, groups, (uint32_t*)amax, HeadFold{});
  else
    hipLaunchKernelGGL((gn_bwd_apply_kernel<bf16_t, false, false, 0>), g2, dim3(256), lds2, st, (const bf16_t*)dz, dzpitch, (const bf16_t*)y, ypitch,
                       scale_shift, mean_rstd, gamma, red, (bf16_t*)dy, dypitch, dgamma, dbeta, act, sl, N, voxels, C, groups, (uint32_t*)amax, HeadFold{});
  BRATS_CHECK_LAUNCH();
  return 0;
}

// GroupNorm + activation backward of a layer whose output feeds only a 1x1x1 head convolution with K = 3 logit planes (the
// network's last ConvBnRelu + outconv, networks/equiunet2020.py:488): dz is never materialised (HeadFold above), and the
// head's dweight [K][C] / dbias [K] come out of the same passes.  Replaces brats_head_bwd(scale 1) + brats_gn_act_bwd.
// dlogits: f32 [N][K][voxels]; hw: the head weight [K][C]; hws: brats_gn_bwd_head_ws_floats(N, C, K) floats.
extern "C" size_t BRATS_API(brats_gn_bwd_head_ws_floats)(int N, int C, int K) { return (size_t)N * GN_BWD_MAX_BLOCKS * (K * C + K); }
extern "C" int BRATS_API(brats_gn_act_bwd_head)(const float* dlogits, const float* hw, int K, const void* y, int ypitch,
                                     const float* scale_shift, const float* mean_rstd, const float* gamma, void* dy, int dypitch,
                                     float* red, float* hws, float* dgamma, float* dbeta, float* dhw, float* dhb, int dtype,
                                     int act, float slope_value, int N, int voxels, int C, int groups, float* amax,
                                     brats_stream_t s) {
  if (!dlogits || !hw || !y || !dy || !red || !hws || !scale_shift || !mean_rstd || !gamma || !dhw || !dhb)
    BRATS_FAIL(BRATS_E_ARG, "gn_act_bwd_head: null pointer");
  if (K != 3 || act > BRATS_ACT_LEAKY)
    BRATS_FAIL(BRATS_E_UNSUPPORTED, "gn_act_bwd_head: built for K = 3 logit planes and relu / leakyrelu (K=%d, act=%d)", K, act);
  HeadFold hf;
  hf.dl = dlogits; hf.w = hw; hf.hpart = hws;
  const int nb = gn_act_bwd_impl(nullptr, 8, y, ypitch, scale_shift, mean_rstd, gamma, dy, dypitch, red, dgamma, dbeta, dtype, act,
                                 SlopeArg{slope_value, nullptr}, N, voxels, C, groups, amax, hf, K, (hipStream_t)s);
  if (nb < 0) return nb;
  return brats_ordered_sum2(hws, dhw, K * C, dhb, N * nb, K * C + K, (hipStream_t)s);  // totals straight into dhw [K][C], dhb [K]
}

// GroupNorm + activation backward of a layer that ends an encoder level (ConvBnRelu -> MaxPool3d(2, 2), its output also the
// skip connection): the output gradient dz = dskip + maxpool-backward(dpool) is composed inside both passes from the skip
// gradient, the pooled gradient and the arg-max bytes the pooling forward recorded (HeadFold, HK == -1).  Replaces
// brats_maxpool2_bwd_idx + brats_gn_act_bwd for that layer; relu / leakyrelu, no MaxAvgPool.
extern "C" int BRATS_API(brats_gn_act_bwd_pool)(const void* dskip, int dskip_pitch, const void* dpool, int dpool_pitch,
                                     const unsigned char* argmax, const void* y, int ypitch, const float* scale_shift,
                                     const float* mean_rstd, const float* gamma, void* dy, int dypitch, float* red, float* dgamma,
                                     float* dbeta, int dtype, int act, float slope_value, int N, int D, int H, int W, int C,
                                     int groups, float* amax, brats_stream_t s) {
  const int vw = dtype == BRATS_BF16 ? 8 : 4;
  if (!dskip || !dpool || !argmax || !y || !dy || !red || !scale_shift || !mean_rstd || !gamma)
    BRATS_FAIL(BRATS_E_ARG, "gn_act_bwd_pool: null pointer");
  if (act > BRATS_ACT_LEAKY || ((D | H | W) & 1) || dskip_pitch % vw || dpool_pitch % vw)
    BRATS_FAIL(BRATS_E_UNSUPPORTED, "gn_act_bwd_pool: relu / leakyrelu, even spatial dims, pitches multiples of %d (act=%d)", vw, act);
  HeadFold hf;
  hf.dskip = dskip; hf.dpool = dpool; hf.argmax = argmax; hf.dskip_pitch = dskip_pitch; hf.dpool_pitch = dpool_pitch;
  hf.D = D; hf.H = H; hf.W = W;
  const int rc = gn_act_bwd_impl(nullptr, vw, y, ypitch, scale_shift, mean_rstd, gamma, dy, dypitch, red, dgamma, dbeta, dtype, act,
                                 SlopeArg{slope_value, nullptr}, N, D * H * W, C, groups, amax, hf, -1, (hipStream_t)s);
  return rc < 0 ? rc : 0;
}

// =================================================================================================
// EvoNorm-S0 (networks/equiunet2021.py:95-103 with group_std :48-52):
//     z = x*sigmoid(x) * rstd_g * gamma_c + beta_c,   rstd_g = 1/sqrt(var_unbiased(group) + eps)
// Statistics come from the convolution epilogue like GroupNorm's; brats_evonorm_finalize only differs
// in the unbiased variance.  `chansum` (optional) accumulates sum_v z per (n, c): the global average
// pool of the following ResidualSELayer for free.
// =================================================================================================
// nn.PReLU (--act prelu, one learnable slope per ConvBnRelu): d loss / d slope = sum over (n, voxel, channel) of
// dz * min(xhat, 0), xhat = y * scale + shift the normalised pre-activation.  One partial per block (part[b]), added in
// block order by brats_ordered_sum: bitwise reproducible.
template <typename T>
__global__ void __launch_bounds__(256) prelu_slope_grad_kernel(const T* __restrict__ dz, int dzpitch, const T* __restrict__ y, int ypitch,
                                                               const float* __restrict__ scale_shift, float* __restrict__ part,
                                                               int voxels, int C) {
  constexpr int VW = 16 / sizeof(T);
  const int n = blockIdx.y;
  const int cv = C / VW, vl_n = blockDim.x / cv;
  const int mycv = threadIdx.x % cv, myvl = threadIdx.x / cv, c0 = mycv * VW;
  float acc = 0.f;
  if (myvl < vl_n) {
    float sc[VW], sh[VW];
#pragma unroll
    for (int j = 0; j < VW; ++j) { sc[j] = scale_shift[((size_t)n * C + c0 + j) * 2]; sh[j] = scale_shift[((size_t)n * C + c0 + j) * 2 + 1]; }
    const T* dzb = dz + (size_t)n * voxels * dzpitch + c0;
    const T* yb = y + (size_t)n * voxels * ypitch + c0;
    for (size_t vox = (size_t)blockIdx.x * vl_n + myvl; vox < (size_t)voxels; vox += (size_t)gridDim.x * vl_n) {
      float g[VW], yy[VW];
      Vec<T, VW>::load(dzb + vox * dzpitch, g);
      Vec<T, VW>::load(yb + vox * ypitch, yy);
#pragma unroll
      for (int j = 0; j < VW; ++j) acc += g[j] * fminf(yy[j] * sc[j] + sh[j], 0.f);
    }
  }
  __shared__ float r[256];
  r[threadIdx.x] = acc;
  __syncthreads();
  for (int m = 128; m > 0; m >>= 1) {
    if ((int)threadIdx.x < m) r[threadIdx.x] += r[threadIdx.x + m];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[(size_t)n * gridDim.x + blockIdx.x] = r[0];
}

constexpr int PRELU_MAX_BLOCKS = 1024;
extern "C" size_t BRATS_API(brats_prelu_ws_floats)(int N) { return (size_t)N * PRELU_MAX_BLOCKS; }
extern "C" int BRATS_API(brats_prelu_slope_grad)(const void* dz, int dzpitch, const void* y, int ypitch, const float* scale_shift, float* ws,
                                      float* dslope, int dtype, int N, int voxels, int C, brats_stream_t s) {
  const int vw = dtype == BRATS_BF16 ? 8 : 4;
  if (!dz || !y || !scale_shift || !ws || !dslope || C % vw || dzpitch % vw || ypitch % vw || C / vw > 256)
    BRATS_FAIL(BRATS_E_ARG, "prelu_slope_grad: bad argument");
  hipStream_t st = (hipStream_t)s;
  const int vl = 256 / (C / vw);
  size_t gx = ((size_t)voxels + (size_t)vl * 8 - 1) / ((size_t)vl * 8);
  gx = gx < 1 ? 1 : (gx > PRELU_MAX_BLOCKS ? PRELU_MAX_BLOCKS : gx);
  if (dtype == BRATS_BF16)
    hipLaunchKernelGGL(prelu_slope_grad_kernel<bf16_t>, dim3((unsigned)gx, N), dim3(256), 0, st, (const bf16_t*)dz, dzpitch,
                       (const bf16_t*)y, ypitch, scale_shift, ws, voxels, C);
  else
    hipLaunchKernelGGL(prelu_slope_grad_kernel<float>, dim3((unsigned)gx, N), dim3(256), 0, st, (const float*)dz, dzpitch,
                       (const float*)y, ypitch, scale_shift, ws, voxels, C);
  BRATS_CHECK_LAUNCH();
  return brats_ordered_sum(ws, dslope, N * (int)gx, 1, st);
}

extern "C" int BRATS_API(brats_evonorm_finalize)(const float* stats, int tiles_per_sample, int N, int C, int groups,
                                      double count_per_channel, float eps, float* mean_rstd, double* chan_ws,
                                      brats_stream_t s) {
  if (!stats || !mean_rstd || !chan_ws || C % groups || C / groups > 256) BRATS_FAIL(BRATS_E_ARG, "evonorm_finalize: bad argument");
  return gn_finalize_launch(stats, tiles_per_sample, N, C, groups, count_per_channel, eps, 1, nullptr, nullptr, mean_rstd, nullptr, chan_ws,
                            (hipStream_t)s);
}

// v_rcp_f32 (1 ulp) instead of the IEEE division sequence (v_div_scale x 2, v_rcp, 4 fma, v_div_fmas, v_div_fixup): the three
// EvoNorm passes are VALU-bound at 128^3 (28 VALU instructions per element with the division, ~19 without)
DEVI float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

template <typename T, bool NT = false>
__global__ void __launch_bounds__(256) evonorm_fwd_kernel(const T* __restrict__ x, int xpitch, const float* __restrict__ mean_rstd,
                                   const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ z,
                                   int zpitch, float* __restrict__ chansum, int voxels, int C, int groups,
                                   uint32_t* __restrict__ amax, const float* __restrict__ gscale) {
  constexpr int VW = 16 / sizeof(T);
  extern __shared__ float sm[];  // sc[C] = rstd*gamma, be[C], then reduction scratch
  float* sc = sm;
  float* be = sm + C;
  const int n = blockIdx.y, cpg = C / groups;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    // gscale (brats_evonorm_se_fwd): the ResidualSELayer's 1 + gate per (n, channel), applied to z in the same pass
    const float gs = gscale ? gscale[(size_t)n * C + c] : 1.f;
    sc[c] = mean_rstd[(n * groups + c / cpg) * 2 + 1] * gamma[c] * gs;
    be[c] = beta[c] * gs;
  }
  __syncthreads();
  const int cv = C / VW, vl_n = blockDim.x / cv;
  const int mycv = threadIdx.x % cv, myvl = threadIdx.x / cv, c0 = mycv * VW;
  float acc[VW];
  float mx = 0.f;
#pragma unroll
  for (int j = 0; j < VW; ++j) acc[j] = 0.f;
  if (myvl < vl_n) {
    const T* xb = x + (size_t)n * voxels * xpitch;
    T* zb = z + (size_t)n * voxels * zpitch;
    for (size_t vox = (size_t)blockIdx.x * vl_n + myvl; vox < (size_t)voxels; vox += (size_t)gridDim.x * vl_n) {
      float a[VW];
      vload<T, VW, NT>(xb + vox * xpitch + c0, a);
#pragma unroll
      for (int j = 0; j < VW; ++j) {
        a[j] = a[j] * sigmoidf_(a[j]) * sc[c0 + j] + be[c0 + j];
        acc[j] += a[j];
      }
      if (amax) {
#pragma unroll
        for (int j = 0; j < VW; j += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, __builtin_fabsf(a[j])), __builtin_fabsf(a[j + 1]));
      }
      vstore<T, VW, NT>(zb + vox * zpitch + c0, a);
    }
  }
  if (amax) record_absmax<T>(mx, amax);
  if (chansum) {
    float* scr = sm + 2 * C;  // [vl_n][C]
    __syncthreads();
    if (myvl < vl_n) {
#pragma unroll
      for (int j = 0; j < VW; ++j) scr[myvl * C + c0 + j] = acc[j];
    }
    __syncthreads();
    float* part = chansum + (size_t)gridDim.y * C + ((size_t)blockIdx.x * gridDim.y + n) * C;  // per-block partials
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      float t = 0.f;
      for (int l = 0; l < vl_n; ++l) t += scr[l * C + c];
      part[c] = t;
    }
  }
}

// per-(n, channel) sums over voxels are produced as per-block partials [blocks][N][C*vals] after the [N][C*vals] totals and
// added in block order (brats_ordered_sum): no float atomics, bitwise reproducible
constexpr int CHAN_MAX_BLOCKS = 1024;
extern "C" size_t BRATS_API(brats_chan_ws_floats)(int N, int C, int vals) { return (size_t)(1 + CHAN_MAX_BLOCKS) * N * C * vals; }

extern "C" int BRATS_API(brats_evonorm_fwd)(const void* x, int xpitch, const float* mean_rstd, const float* gamma, const float* beta,
                                 void* z, int zpitch, float* chansum, int dtype, int N, int voxels, int C, int groups,
                                 float* amax, brats_stream_t s) {
  const int vw = dtype == BRATS_BF16 ? 8 : 4;
  if (!x || !z || !mean_rstd || !gamma || !beta || C % vw || C % groups || xpitch % vw || zpitch % vw || C / vw > 256)
    BRATS_FAIL(BRATS_E_ARG, "evonorm_fwd: bad argument (C, pitches multiples of %d)", vw);
  hipStream_t st = (hipStream_t)s;
  const int cv = C / vw, vl = 256 / cv;
  size_t gx = ((size_t)voxels + (size_t)vl * 8 - 1) / ((size_t)vl * 8);
  dim3 grid((unsigned)(gx < 1 ? 1 : (gx > CHAN_MAX_BLOCKS ? CHAN_MAX_BLOCKS : gx)), N);
  const size_t lds = (size_t)(2 * C + vl * C) * sizeof(float);
  if (dtype == BRATS_BF16 && stream_nt((size_t)N * voxels * C * 2))  // (beyond the Infinity Cache: non-temporal streaming, common.hpp)
    hipLaunchKernelGGL((evonorm_fwd_kernel<bf16_t, true>), grid, dim3(256), lds, st, (const bf16_t*)x, xpitch, mean_rstd, gamma, beta,
                       (bf16_t*)z, zpitch, chansum, voxels, C, groups, (uint32_t*)amax, (const float*)nullptr);
  else if (dtype == BRATS_BF16)
    hipLaunchKernelGGL(evonorm_fwd_kernel<bf16_t>, grid, dim3(256), lds, st, (const bf16_t*)x, xpitch, mean_rstd, gamma, beta,
                       (bf16_t*)z, zpitch, chansum, voxels, C, groups, (uint32_t*)amax, (const float*)nullptr);
  else
    hipLaunchKernelGGL(evonorm_fwd_kernel<float>, grid, dim3(256), lds, st, (const float*)x, xpitch, mean_rstd, gamma, beta,
                       (float*)z, zpitch, chansum, voxels, C, groups, (uint32_t*)amax, (const float*)nullptr);
  if (chansum) brats_ordered_sum(chansum + (size_t)N * C, chansum, (int)grid.x, N * C, st);
  BRATS_CHECK_LAUNCH();
  return 0;
}

// out[n][c] = sum_v x*sigmoid(x) (per-block partials, added in block order by the caller): the statistics pass of
// brats_evonorm_se_fwd
template <typename T, bool NT = false>
__global__ void __launch_bounds__(256) evonorm_numsum_kernel(const T* __restrict__ x, int xpitch, float* __restrict__ out, int voxels, int C) {
  constexpr int VW = 16 / sizeof(T);
  extern __shared__ float sm[];
  const int n = blockIdx.y;
  const int cv = C / VW, vl_n = blockDim.x / cv;
  const int mycv = threadIdx.x % cv, myvl = threadIdx.x / cv, c0 = mycv * VW;
  float acc[VW];
#pragma unroll
  for (int j = 0; j < VW; ++j) acc[j] = 0.f;
  if (myvl < vl_n) {
    const T* xb = x + (size_t)n * voxels * xpitch + c0;
    const size_t stride = (size_t)gridDim.x * vl_n;
    size_t vox = (size_t)blockIdx.x * vl_n + myvl;
    for (; vox + stride < (size_t)voxels; vox += 2 * stride) {
      float x0[VW], x1[VW];
      vload<T, VW, NT>(xb + vox * xpitch, x0);
      vload<T, VW, NT>(xb + (vox + stride) * xpitch, x1);
#pragma unroll
      for (int j = 0; j < VW; ++j) acc[j] += x0[j] * sigmoidf_(x0[j]);
#pragma unroll
      for (int j = 0; j < VW; ++j) acc[j] += x1[j] * sigmoidf_(x1[j]);
    }
    if (vox < (size_t)voxels) {
      float x0[VW];
      vload<T, VW, NT>(xb + vox * xpitch, x0);
#pragma unroll
      for (int j = 0; j < VW; ++j) acc[j] += x0[j] * sigmoidf_(x0[j]);
    }
#pragma unroll
    for (int j = 0; j < VW; ++j) sm[myvl * C + c0 + j] = acc[j];
  }
  __syncthreads();
  float* part = out + (size_t)gridDim.y * C + ((size_t)blockIdx.x * gridDim.y + n) * C;  // per-block partials
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float t = 0.f;
    for (int l = 0; l < vl_n; ++l) t += sm[l * C + c];
    part[c] = t;
  }
}

// EvoNorm + ResidualSELayer forward without storing the EvoNorm output z: pass 1 reads x and sums num(x) per (n, channel);
// the SE gate runs on sum_v z reconstructed from those sums (se.hpp); pass 2 writes out = z * (1 + gate) directly.  Replaces
// brats_evonorm_fwd(chansum) + brats_se_fwd + brats_channel_scale: 3 tensor passes (x, x, out) instead of 4 (x, z, z, out).
// ws: brats_chan_ws_floats(N, C, 1) floats; chansum_out [N][C] = sum_v z (what brats_evonorm_se_bwd's se_chansum wants).
extern "C" int BRATS_API(brats_evonorm_se_fwd)(const void* x, int xpitch, const float* mean_rstd, const float* gamma, const float* beta,
                                    const float* w1, const float* b1, const float* w2, const float* b2, void* out, int opitch,
                                    float* ws, float* chansum_out, float* gate1p, float* hidden, int Ch, int dtype, int N,
                                    int voxels, int C, int groups, float* amax, brats_stream_t s) {
  const int vw = dtype == BRATS_BF16 ? 8 : 4;
  if (!x || !mean_rstd || !gamma || !beta || !ws || !chansum_out || !gate1p || !hidden)
    BRATS_FAIL(BRATS_E_ARG, "evonorm_se_fwd: null pointer");
  if (C % vw || C % groups || xpitch % vw || opitch % vw || C / vw > 256)
    BRATS_FAIL(BRATS_E_ARG, "evonorm_se_fwd: bad argument (C, pitches multiples of %d)", vw);
  hipStream_t st = (hipStream_t)s;
  const int cv = C / vw, vl = 256 / cv;
  size_t gx = ((size_t)voxels + (size_t)vl * 8 - 1) / ((size_t)vl * 8);
  const bool big = dtype == BRATS_BF16 && stream_nt((size_t)N * voxels * C * 2);
  const size_t cap = big ? CHAN_MAX_BLOCKS : 512;
  dim3 g1((unsigned)(gx < 1 ? 1 : (gx > cap ? cap : gx)), N);
  const size_t lds1 = (size_t)vl * C * sizeof(float);
  if (big)
    hipLaunchKernelGGL((evonorm_numsum_kernel<bf16_t, true>), g1, dim3(256), lds1, st, (const bf16_t*)x, xpitch, ws, voxels, C);
  else if (dtype == BRATS_BF16)
    hipLaunchKernelGGL(evonorm_numsum_kernel<bf16_t>, g1, dim3(256), lds1, st, (const bf16_t*)x, xpitch, ws, voxels, C);
  else
    hipLaunchKernelGGL(evonorm_numsum_kernel<float>, g1, dim3(256), lds1, st, (const float*)x, xpitch, ws, voxels, C);
  BRATS_CHECK_LAUNCH();
  brats_ordered_sum(ws + (size_t)N * C, ws, (int)g1.x, N * C, st);
  SeFwdFold fold;
  fold.numsum = ws; fold.mean_rstd = mean_rstd; fold.gamma = gamma; fold.beta = beta; fold.chansum_out = chansum_out;
  fold.groups = groups; fold.voxels = (float)voxels;
  if (int rc = brats_se_fwd_launch(nullptr, fold, 1.f / (float)voxels, w1, b1, w2, b2, gate1p, hidden, N, C, Ch, st)) return rc;
  if (!out) return 0;  // (the consumer applies the gated EvoNorm on load: brats_evonorm_head_fwd)
  dim3 g2((unsigned)(gx < 1 ? 1 : (gx > CHAN_MAX_BLOCKS ? CHAN_MAX_BLOCKS : gx)), N);
  const size_t lds2 = (size_t)(2 * C + vl * C) * sizeof(float);
  if (big)
    hipLaunchKernelGGL((evonorm_fwd_kernel<bf16_t, true>), g2, dim3(256), lds2, st, (const bf16_t*)x, xpitch, mean_rstd, gamma, beta,
                       (bf16_t*)out, opitch, (float*)nullptr, voxels, C, groups, (uint32_t*)amax, (const float*)gate1p);
  else if (dtype == BRATS_BF16)
    hipLaunchKernelGGL(evonorm_fwd_kernel<bf16_t>, g2, dim3(256), lds2, st, (const bf16_t*)x, xpitch, mean_rstd, gamma, beta,
                       (bf16_t*)out, opitch, (float*)nullptr, voxels, C, groups, (uint32_t*)amax, (const float*)gate1p);
  else
    hipLaunchKernelGGL(evonorm_fwd_kernel<float>, g2, dim3(256), lds2, st, (const float*)x, xpitch, mean_rstd, gamma, beta,
                       (float*)out, opitch, (float*)nullptr, voxels, C, groups, (uint32_t*)amax, (const float*)gate1p);
  BRATS_CHECK_LAUNCH();
  return 0;
}

// backward pass 1: red[n][c] = { sum_v dz, sum_v dz * x*sigmoid(x), sum_v dz * d/dx[x*sigmoid(x)] }
// (the third sum gives the bias gradient of the preceding convolution without another pass over dx)
// RAW5 (brats_evonorm_se_bwd): dz is the SE block's output gradient `do` as it is, and two more sums -- sum_v x*sigmoid(x),
// sum_v d/dx[x*sigmoid(x)] -- come along: the SE backward (se.hip) derives d loss / d gate and the three sums above for
// dz = do * gscale + gadd from these five (se.hpp), so that no pass computes sum_v do * z on its own.
// HK > 0 (RAW5 only; brats_evonorm_se_bwd with dlogits): the SE block's output feeds only the 1x1x1 output head, `do` =
// W_head^T dlogits is computed on the fly (HeadFold, as in gn_bwd_reduce_kernel); the head's weight gradient
// sum_v dl[k] * out[v][c], out = (num * sc + be) * gate1p, is linear in sum_v dl[k] * num and sum_v dl[k].
struct EvoHead {
  HeadFold hf;
  const float* mean_rstd = nullptr;
  const float* gamma = nullptr;
  const float* beta = nullptr;
  const float* gate1p = nullptr;
  int groups = 1;
};
template <typename T, bool NT = false, bool RAW5 = false, int HK = 0>
__global__ void __launch_bounds__(256) evonorm_bwd_reduce_kernel(const T* __restrict__ dz, int dzpitch, const T* __restrict__ x, int xpitch,
                                          float* __restrict__ red, int voxels, int C, const float* __restrict__ gscale,
                                          const float* __restrict__ gadd, EvoHead eh) {
  constexpr int VW = 16 / sizeof(T);
  constexpr int NV = RAW5 ? 5 : 3;
  constexpr int HKA = HK > 0 ? HK : 1;
  static_assert(HK == 0 || RAW5, "the head / pool folds exist for the five-sum pass only");
  extern __shared__ float sm[];
  const int n = blockIdx.y;
  const int cv = C / VW, vl_n = blockDim.x / cv;
  const int mycv = threadIdx.x % cv, myvl = threadIdx.x / cv, c0 = mycv * VW;
  float a1[VW], a2[VW], a3[VW], a4[VW], a5[VW];
  float aw[HKA][VW], ab[HKA];
#pragma unroll
  for (int j = 0; j < VW; ++j) a1[j] = a2[j] = a3[j] = a4[j] = a5[j] = 0.f;
#pragma unroll
  for (int k = 0; k < HKA; ++k) {
    ab[k] = 0.f;
#pragma unroll
    for (int j = 0; j < VW; ++j) aw[k][j] = 0.f;
  }
  if (myvl < vl_n) {
    const T* dzb = dz + (size_t)n * voxels * dzpitch;
    const T* xb = x + (size_t)n * voxels * xpitch;
    // optional per-(n, channel) affine map of the incoming gradient (the SE layer's backward, dz = do * (1 + gate) + dgap / V,
    // folded in instead of a separate channel_scale pass over the tensor)
    float gs[VW], ga[VW];
#pragma unroll
    for (int j = 0; j < VW; ++j) { gs[j] = gscale ? gscale[(size_t)n * C + c0 + j] : 1.f; ga[j] = gadd ? gadd[(size_t)n * C + c0 + j] : 0.f; }
    auto body = [&](const float* g_, const float* xx) {
#pragma unroll
      for (int j = 0; j < VW; ++j) {
        const float sg = sigmoidf_(xx[j]);
        if constexpr (RAW5) {
          const float num = xx[j] * sg, dnum = sg * (1.f + xx[j] * (1.f - sg));
          a1[j] += g_[j];
          a2[j] += g_[j] * num;
          a3[j] += g_[j] * dnum;
          a4[j] += num;
          a5[j] += dnum;
        } else {
          const float g = g_[j] * gs[j] + ga[j];
          a1[j] += g;
          a2[j] += g * xx[j] * sg;
          a3[j] += g * sg * (1.f + xx[j] * (1.f - sg));
        }
      }
    };
    const size_t stride = (size_t)gridDim.x * vl_n;
    size_t vox = (size_t)blockIdx.x * vl_n + myvl;
    if constexpr (HK < 0) {  // do = skip gradient + MaxAvgPool backward, composed from the pieces (HeadFold, pool_dz)
      auto step = [&](size_t v0, auto cnt) {
        constexpr int NVX = decltype(cnt)::value;
        typename Raw16<T>::type xr[NVX], skr[NVX], dpr[NVX], avr[NVX];
        uint32_t aw8[NVX][VW / 4];
        int kk[NVX];
#pragma unroll
        for (int i = 0; i < NVX; ++i) {
          xr[i] = Raw16<T>::template load<NT>(xb + (v0 + i * stride) * xpitch + c0);
          pool_dz<T, VW>(eh.hf, n, v0 + i * stride, c0, C, skr[i], dpr[i], avr[i], aw8[i], kk[i]);
        }
#pragma unroll
        for (int i = 0; i < NVX; ++i) {
          float xx[VW], g[VW];
          Raw16<T>::unpack(xr[i], xx);
          pool_dz_finish<T, VW>(skr[i], dpr[i], avr[i], eh.hf.with_avg, aw8[i], kk[i], g);
          body(g, xx);
        }
      };
      for (; vox + stride < (size_t)voxels; vox += 2 * stride) step(vox, std::integral_constant<int, 2>{});
      for (; vox < (size_t)voxels; vox += stride) step(vox, std::integral_constant<int, 1>{});
    } else if constexpr (HK > 0) {
      float wr[HKA][VW];
#pragma unroll
      for (int k = 0; k < HK; ++k)
#pragma unroll
        for (int j = 0; j < VW; ++j) wr[k][j] = eh.hf.w[k * C + c0 + j];
      const float* dl = eh.hf.dl + (size_t)n * HK * voxels;
      auto step = [&](size_t v0, auto cnt) {  // four voxels in flight per thread
        constexpr int NVX = decltype(cnt)::value;
        typename Raw16<T>::type xr[NVX];
        float gq[NVX][HKA];
#pragma unroll
        for (int i = 0; i < NVX; ++i) xr[i] = Raw16<T>::template load<NT>(xb + (v0 + i * stride) * xpitch + c0);
#pragma unroll
        for (int i = 0; i < NVX; ++i)
#pragma unroll
          for (int k = 0; k < HK; ++k) gq[i][k] = dl[(size_t)k * voxels + v0 + i * stride];
#pragma unroll
        for (int i = 0; i < NVX; ++i) {
          float xx[VW];
          Raw16<T>::unpack(xr[i], xx);
#pragma unroll
          for (int j = 0; j < VW; ++j) {
            float d = 0.f;
#pragma unroll
            for (int k = 0; k < HK; ++k) d += gq[i][k] * wr[k][j];
            const float sg = sigmoidf_(xx[j]);
            const float num = xx[j] * sg, dnum = sg * (1.f + xx[j] * (1.f - sg));
            a1[j] += d;
            a2[j] += d * num;
            a3[j] += d * dnum;
            a4[j] += num;
            a5[j] += dnum;
#pragma unroll
            for (int k = 0; k < HK; ++k) aw[k][j] += gq[i][k] * num;
          }
#pragma unroll
          for (int k = 0; k < HK; ++k) ab[k] += gq[i][k];
        }
      };
      for (; vox + 3 * stride < (size_t)voxels; vox += 4 * stride) step(vox, std::integral_constant<int, 4>{});
      for (; vox < (size_t)voxels; vox += stride) step(vox, std::integral_constant<int, 1>{});
    } else {
    for (; vox + stride < (size_t)voxels; vox += 2 * stride) {  // two voxels (4 loads) in flight per thread
      float g0[VW], x0[VW], g1[VW], x1[VW];
      vload<T, VW, NT>(dzb + vox * dzpitch + c0, g0);
      vload<T, VW, NT>(xb + vox * xpitch + c0, x0);
      vload<T, VW, NT>(dzb + (vox + stride) * dzpitch + c0, g1);
      vload<T, VW, NT>(xb + (vox + stride) * xpitch + c0, x1);
      body(g0, x0);
      body(g1, x1);
    }
    if (vox < (size_t)voxels) {
      float g0[VW], x0[VW];
      vload<T, VW, NT>(dzb + vox * dzpitch + c0, g0);
      vload<T, VW, NT>(xb + vox * xpitch + c0, x0);
      body(g0, x0);
    }
    }
  }
  float* scr = sm;  // [vl_n][C] (+ [vl_n][HK])
  const bool live = myvl < vl_n;
  float* part = red + (size_t)gridDim.y * C * NV + ((size_t)blockIdx.x * gridDim.y + n) * C * NV;  // per-block partials
  lane_reduce_plane<VW>(scr, a1, live, myvl, vl_n, C, c0, [&](int c, float t) { part[c * NV] = t; });
  lane_reduce_plane<VW>(scr, a2, live, myvl, vl_n, C, c0, [&](int c, float t) { part[c * NV + 1] = t; });
  lane_reduce_plane<VW>(scr, a3, live, myvl, vl_n, C, c0, [&](int c, float t) { part[c * NV + 2] = t; });
  if constexpr (RAW5) {
    lane_reduce_plane<VW>(scr, a4, live, myvl, vl_n, C, c0, [&](int c, float t) { part[c * NV + 3] = t; });
    lane_reduce_plane<VW>(scr, a5, live, myvl, vl_n, C, c0, [&](int c, float t) { part[c * NV + 4] = t; });
  }
  if constexpr (HK > 0) {  // the head's weight / bias gradient partials [HK][C] + [HK] of this block (one sample)
    float* hb = sm + vl_n * C;  // [vl_n][HK]
    if (live && mycv == 0) {
#pragma unroll
      for (int k = 0; k < HK; ++k) hb[myvl * HK + k] = ab[k];
    }
    float* hp = eh.hf.hpart + ((size_t)n * gridDim.x + blockIdx.x) * (HK * C + HK);
    const int cpg = C / eh.groups;
#pragma unroll
    for (int k = 0; k < HK; ++k)
      lane_reduce_plane<VW>(scr, aw[k], live, myvl, vl_n, C, c0, [&](int c, float t) {
        float tb = 0.f;
        for (int l = 0; l < vl_n; ++l) tb += hb[l * HK + k];
        // sum_v dl * out, out = (num * rstd * gamma + beta) * gate1p
        const float sc = eh.mean_rstd[(n * eh.groups + c / cpg) * 2 + 1] * eh.gamma[c];
        hp[k * C + c] = eh.gate1p[(size_t)n * C + c] * (sc * t + eh.beta[c] * tb);
      });
    if ((int)threadIdx.x < HK) {
      float tb = 0.f;
      for (int l = 0; l < vl_n; ++l) tb += hb[l * HK + threadIdx.x];
      hp[HK * C + threadIdx.x] = tb;
    }
  }
}

// pass 2: dx = dz*gamma*r*num'(x) - r^3 * A_g * (x - mean_g)/(M-1),  A_g = sum_{c in g} gamma_c * red[n][c][1]
template <typename T, bool NT = false, int HK = 0>
__global__ void __launch_bounds__(256) evonorm_bwd_apply_kernel(const T* __restrict__ dz, int dzpitch, const T* __restrict__ x, int xpitch,
                                         const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
                                         const float* __restrict__ red, T* __restrict__ dx, int dxpitch,
                                         float* __restrict__ dgamma, float* __restrict__ dbeta, const double* __restrict__ chan,
                                         float* __restrict__ dconvbias, int N, int voxels, int C, int groups,
                                         uint32_t* __restrict__ amax, const float* __restrict__ gscale,
                                         const float* __restrict__ gadd, HeadFold hf) {
  constexpr int VW = 16 / sizeof(T);
  constexpr int HKA = HK > 0 ? HK : 1;
  extern __shared__ float sm[];
  float* gr = sm;          // [C] gamma * r
  float* mu = sm + C;      // [C] group mean
  float* kk = sm + 2 * C;  // [C] r^3 * A_g / (M-1)
  const int n = blockIdx.y, cpg = C / groups;
  const float Mm1 = (float)cpg * (float)voxels - 1.f;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const int g = c / cpg;
    const float r = mean_rstd[(n * groups + g) * 2 + 1];
    float A = 0.f;
    for (int j = 0; j < cpg; ++j) A += gamma[g * cpg + j] * red[((size_t)n * C + g * cpg + j) * 3 + 1];
    gr[c] = gamma[c] * r;
    mu[c] = mean_rstd[(n * groups + g) * 2];
    kk[c] = r * r * r * A / Mm1;
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && dgamma) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      float b = 0.f, g = 0.f;
      float dbc = 0.f;  // sum_v dx = bias gradient of the convolution that produced x
      for (int m = 0; m < N; ++m) {
        const int gg = c / cpg;
        const float r = mean_rstd[(m * groups + gg) * 2 + 1], mean = mean_rstd[(m * groups + gg) * 2];
        b += red[((size_t)m * C + c) * 3];
        g += red[((size_t)m * C + c) * 3 + 1] * r;
        float A = 0.f;
        for (int j = 0; j < cpg; ++j) A += gamma[gg * cpg + j] * red[((size_t)m * C + gg * cpg + j) * 3 + 1];
        const float sumx = chan ? (float)chan[((size_t)m * C + c) * 2] : 0.f;
        dbc += gamma[c] * r * red[((size_t)m * C + c) * 3 + 2] - r * r * r * A / Mm1 * (sumx - (float)voxels * mean);
      }
      dbeta[c] = b;
      dgamma[c] = g;
      if (dconvbias) dconvbias[c] = dbc;
    }
  }
  __syncthreads();
  const int cv = C / VW, vl_n = blockDim.x / cv;
  const int mycv = threadIdx.x % cv, myvl = threadIdx.x / cv, c0 = mycv * VW;
  const bool live = myvl < vl_n;  // idle threads stay for the |max| reduction
  float cg[VW], cm[VW], ck[VW], gs[VW], ga[VW];  // per-channel constants in registers (not re-read from LDS per element)
#pragma unroll
  for (int j = 0; j < VW; ++j) {
    cg[j] = gr[c0 + j]; cm[j] = mu[c0 + j]; ck[j] = kk[c0 + j];
    gs[j] = gscale ? gscale[(size_t)n * C + c0 + j] : 1.f;
    ga[j] = gadd ? gadd[(size_t)n * C + c0 + j] : 0.f;
  }
  const T* dzb = dz + (size_t)n * voxels * dzpitch + c0;
  const T* xb = x + (size_t)n * voxels * xpitch + c0;
  T* dxb = dx + (size_t)n * voxels * dxpitch + c0;
  float mx = 0.f;
  auto body = [&](const float* g, const float* xx, float* o) {
#pragma unroll
    for (int j = 0; j < VW; ++j) {
      const float sg = sigmoidf_(xx[j]);
      const float dnum = sg * (1.f + xx[j] * (1.f - sg));
      o[j] = (g[j] * gs[j] + ga[j]) * cg[j] * dnum - ck[j] * (xx[j] - cm[j]);
    }
    if (amax) {
#pragma unroll
      for (int j = 0; j < VW; j += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, __builtin_fabsf(o[j])), __builtin_fabsf(o[j + 1]));
    }
  };
  const size_t stride = (size_t)gridDim.x * vl_n;
  size_t vox = live ? (size_t)blockIdx.x * vl_n + myvl : (size_t)voxels;
  if constexpr (HK < 0) {
    auto step = [&](size_t v0, auto cnt) {
      constexpr int NVX = decltype(cnt)::value;
      typename Raw16<T>::type xr[NVX], skr[NVX], dpr[NVX], avr[NVX];
      uint32_t aw8[NVX][VW / 4];
      int kk[NVX];
#pragma unroll
      for (int i = 0; i < NVX; ++i) {
        xr[i] = Raw16<T>::template load<NT>(xb + (v0 + i * stride) * xpitch);
        pool_dz<T, VW>(hf, n, v0 + i * stride, c0, C, skr[i], dpr[i], avr[i], aw8[i], kk[i]);
      }
#pragma unroll
      for (int i = 0; i < NVX; ++i) {
        float xx[VW], g[VW], o[VW];
        Raw16<T>::unpack(xr[i], xx);
        pool_dz_finish<T, VW>(skr[i], dpr[i], avr[i], hf.with_avg, aw8[i], kk[i], g);
        body(g, xx, o);
        vstore<T, VW, NT>(dxb + (v0 + i * stride) * dxpitch, o);
      }
    };
    for (; vox + stride < (size_t)voxels; vox += 2 * stride) step(vox, std::integral_constant<int, 2>{});
    for (; vox < (size_t)voxels; vox += stride) step(vox, std::integral_constant<int, 1>{});
  } else if constexpr (HK > 0) {
    float wr[HKA][VW];
#pragma unroll
    for (int k = 0; k < HK; ++k)
#pragma unroll
      for (int j = 0; j < VW; ++j) wr[k][j] = hf.w[k * C + c0 + j];
    const float* dl = hf.dl + (size_t)n * HK * voxels;
    auto step = [&](size_t v0, auto cnt) {  // four voxels in flight per thread
      constexpr int NVX = decltype(cnt)::value;
      typename Raw16<T>::type xr[NVX];
      float gq[NVX][HKA];
#pragma unroll
      for (int i = 0; i < NVX; ++i) xr[i] = Raw16<T>::template load<NT>(xb + (v0 + i * stride) * xpitch);
#pragma unroll
      for (int i = 0; i < NVX; ++i)
#pragma unroll
        for (int k = 0; k < HK; ++k) gq[i][k] = dl[(size_t)k * voxels + v0 + i * stride];
#pragma unroll
      for (int i = 0; i < NVX; ++i) {
        float xx[VW], g[VW], o[VW];
        Raw16<T>::unpack(xr[i], xx);
#pragma unroll
        for (int j = 0; j < VW; ++j) {
          float d = 0.f;
#pragma unroll
          for (int k = 0; k < HK; ++k) d += gq[i][k] * wr[k][j];
          g[j] = d;
        }
        body(g, xx, o);
        vstore<T, VW, NT>(dxb + (v0 + i * stride) * dxpitch, o);
      }
    };
    for (; vox + 3 * stride < (size_t)voxels; vox += 4 * stride) step(vox, std::integral_constant<int, 4>{});
    for (; vox < (size_t)voxels; vox += stride) step(vox, std::integral_constant<int, 1>{});
  } else {
  for (; vox + stride < (size_t)voxels; vox += 2 * stride) {
    float g0[VW], x0[VW], g1[VW], x1[VW], o0[VW], o1[VW];
    vload<T, VW, NT>(dzb + vox * dzpitch, g0);
    vload<T, VW, NT>(xb + vox * xpitch, x0);
    vload<T, VW, NT>(dzb + (vox + stride) * dzpitch, g1);
    vload<T, VW, NT>(xb + (vox + stride) * xpitch, x1);
    body(g0, x0, o0);
    body(g1, x1, o1);
    vstore<T, VW, NT>(dxb + vox * dxpitch, o0);
    vstore<T, VW, NT>(dxb + (vox + stride) * dxpitch, o1);
  }
  if (vox < (size_t)voxels) {
    float g0[VW], x0[VW], o0[VW];
    vload<T, VW, NT>(dzb + vox * dzpitch, g0);
    vload<T, VW, NT>(xb + vox * xpitch, x0);
    body(g0, x0, o0);
    vstore<T, VW, NT>(dxb + vox * dxpitch, o0);
  }
  }
  if (amax) record_absmax<T>(mx, amax);
}

extern "C" int BRATS_API(brats_evonorm_bwd)(const void* dz, int dzpitch, const void* x, int xpitch, const float* mean_rstd,
                                 const float* gamma, void* dx, int dxpitch, float* red, float* dgamma, float* dbeta,
                                 const double* chan_sums, float* dconvbias, int dtype, int N, int voxels, int C, int groups,
                                 float* amax, const float* gscale, const float* gadd, brats_stream_t s) {
  const int vw = dtype == BRATS_BF16 ? 8 : 4;
  if (!dz || !x || !dx || !red || !mean_rstd || !gamma) BRATS_FAIL(BRATS_E_ARG, "evonorm_bwd: null pointer");
  if (dconvbias && !chan_sums) BRATS_FAIL(BRATS_E_ARG, "evonorm_bwd: dconvbias needs the forward per-channel sums");
  if (C % vw || C % groups || dzpitch % vw || xpitch % vw || dxpitch % vw || C / vw > 256)
    BRATS_FAIL(BRATS_E_ARG, "evonorm_bwd: C=%d / pitches must be multiples of %d", C, vw);
  hipStream_t st = (hipStream_t)s;
  const int cv = C / vw, vl = 256 / cv;
  size_t gx = ((size_t)voxels + (size_t)vl * 8 - 1) / ((size_t)vl * 8);
  const bool big = dtype == BRATS_BF16 && stream_nt((size_t)N * voxels * C * 2);
  const size_t cap1 = big ? CHAN_MAX_BLOCKS : 512, cap2 = big ? 8192 : 2048;  // large tensors: many short-lived blocks stream faster
  dim3 g1((unsigned)(gx < 1 ? 1 : (gx > cap1 ? cap1 : gx)), N);  // one partial per block, added in block order
  const size_t lds1 = (size_t)(vl * C) * sizeof(float);  // (one value plane at a time: lane_reduce_plane)
  dim3 g2((unsigned)(gx < 1 ? 1 : (gx > cap2 ? cap2 : gx)), N);
  const size_t lds2 = (size_t)3 * C * sizeof(float);
  if (big) {
    hipLaunchKernelGGL((evonorm_bwd_reduce_kernel<bf16_t, true>), g1, dim3(256), lds1, st, (const bf16_t*)dz, dzpitch, (const bf16_t*)x,
                       xpitch, red, voxels, C, gscale, gadd, EvoHead{});
    brats_ordered_sum(red + (size_t)N * C * 3, red, (int)g1.x, N * C * 3, st);
    hipLaunchKernelGGL((evonorm_bwd_apply_kernel<bf16_t, true>), g2, dim3(256), lds2, st, (const bf16_t*)dz, dzpitch, (const bf16_t*)x,
                       xpitch, mean_rstd, gamma, red, (bf16_t*)dx, dxpitch, dgamma, dbeta, chan_sums, dconvbias, N, voxels, C, groups, (uint32_t*)amax, gscale, gadd, HeadFold{});
  } else if (dtype == BRATS_BF16) {
    hipLaunchKernelGGL(evonorm_bwd_reduce_kernel<bf16_t>, g1, dim3(256), lds1, st, (const bf16_t*)dz, dzpitch, (const bf16_t*)x,
                       xpitch, red, voxels, C, gscale, gadd, EvoHead{});
    brats_ordered_sum(red + (size_t)N * C * 3, red, (int)g1.x, N * C * 3, st);
    hipLaunchKernelGGL(evonorm_bwd_apply_kernel<bf16_t>, g2, dim3(256), lds2, st, (const bf16_t*)dz, dzpitch, (const bf16_t*)x,
                       xpitch, mean_rstd, gamma, red, (bf16_t*)dx, dxpitch, dgamma, dbeta, chan_sums, dconvbias, N, voxels, C, groups, (uint32_t*)amax, gscale, gadd, HeadFold{});
  } else {
    hipLaunchKernelGGL(evonorm_bwd_reduce_kernel<float>, g1, dim3(256), lds1, st, (const float*)dz, dzpitch, (const float*)x,
                       xpitch, red, voxels, C, gscale, gadd, EvoHead{});
    brats_ordered_sum(red + (size_t)N * C * 3, red, (int)g1.x, N * C * 3, st);
    hipLaunchKernelGGL(evonorm_bwd_apply_kernel<float>, g2, dim3(256), lds2, st, (const float*)dz, dzpitch, (const float*)x,
                       xpitch, mean_rstd, gamma, red, (float*)dx, dxpitch, dgamma, dbeta, chan_sums, dconvbias, N, voxels, C, groups, (uint32_t*)amax, gscale, gadd, HeadFold{});
  }
  BRATS_CHECK_LAUNCH();
  return 0;
}

// ---- EvoNorm backward whose first pass was taken by the producer of dz (round 5; the EvoNorm analogue of brats_gn_act_bwd_tiles)
// A block of EquiUnetASSPEvo is conv1 -> EvoNorm -> conv2 -> EvoNorm -> SE (networks/equiunet2021.py:197-206): dz, the gradient
// of the first EvoNorm's output z, is produced by conv2's input-gradient launch and then read back twice (pass 1: sum dz,
// sum dz * num(x), sum dz * num'(x); pass 2: dx).  num(x) = x * sigmoid(x) costs a transcendental per element -- nothing for a
// convolution epilogue -- but z = num(x) * (rstd * gamma) + beta is LINEAR in it and z is in memory anyway (conv2's saved
// input): the "backward statistics" form of the 16-bit convolutions (brats_conv3d_fwd_bstats with by = z, leakyrelu slope 1:
// u = dz) leaves S1 = sum dz and S2 = sum dz * z per tile and channel, from which
//     gamma_c * sum dz * num = (S2 - beta_c * S1) / rstd_g          (no division by gamma)
// gives A_g, the only pass-1 quantity pass 2 needs.  The two sums that are NOT linear in z -- sum dz * num (for dgamma when
// gamma_c = 0 exactly) and sum dz * num' (the convolution's bias gradient) -- are taken by pass 2 itself, which evaluates the
// sigmoid per element anyway: two more FMAs per element and a block reduction.  dz and x are read once instead of twice.
// red: brats_evonorm_bwd_tiles_ws_floats(N, C) floats.
constexpr int EVO_SIDE_MAX_BLOCKS = 2048;
extern "C" size_t BRATS_API(brats_evonorm_bwd_tiles_ws_floats)(int N, int C) {
  // [N][C][2] (S1, gamma * sum dz * num) + side totals [N][C][2] + side partials [blocks][N][C][2] + f64 slab partials of the tiles
  return (size_t)N * C * 4 + (size_t)EVO_SIDE_MAX_BLOCKS * N * C * 2 + (size_t)2 * GN_MAX_SPLITS * N * C * 2 + 8;
}

// s12[n][c] = { S1, (S2 - beta_c * S1) / rstd_g } from the f64 slab sums of the tile statistics (fixed order)
__global__ void __launch_bounds__(256) evonorm_bwd_tiles_prep_kernel(const double* __restrict__ part, int splits, int N, int C, int groups,
                                                                     const float* __restrict__ mean_rstd, const float* __restrict__ beta,
                                                                     float* __restrict__ s12) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * C) return;
  const int n = i / C, c = i % C;
  double s1 = 0.0, s2 = 0.0;
  const double* src = part + ((size_t)n * C + c) * 2;
  const size_t zs = (size_t)N * C * 2;
  for (int z = 0; z < splits; ++z) { s1 += src[z * zs]; s2 += src[z * zs + 1]; }
  const double rstd = mean_rstd[(n * groups + c / (C / groups)) * 2 + 1];
  s12[(size_t)i * 2] = (float)s1;
  s12[(size_t)i * 2 + 1] = (float)((s2 - (double)beta[c] * s1) / rstd);
}

// pass 2 alone: dx = dz * gamma * r * num'(x) - r^3 * A_g * (x - mean_g) / (M - 1), A_g = sum_{c in g} s12[n][c][1];
// side partials [block][n][c] = { sum dz * num, sum dz * num' } over the block's voxels
template <typename T, bool NT>
__global__ void __launch_bounds__(256) evonorm_bwd_apply_side_kernel(const T* __restrict__ dz, int dzpitch, const T* __restrict__ x, int xpitch,
                                                                     const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
                                                                     const float* __restrict__ s12, T* __restrict__ dx, int dxpitch,
                                                                     float* __restrict__ side_part, int voxels, int C, int groups,
                                                                     uint32_t* __restrict__ amax) {
  constexpr int VW = 16 / sizeof(T);
  extern __shared__ float sm[];  // [3][C] constants, then the reduction scratch [vl_n][C]
  float* gr = sm;          // gamma * r
  float* mu = sm + C;      // group mean
  float* kk = sm + 2 * C;  // r^3 * A_g / (M - 1)
  const int n = blockIdx.y, cpg = C / groups;
  const float Mm1 = (float)cpg * (float)voxels - 1.f;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const int g = c / cpg;
    const float r = mean_rstd[(n * groups + g) * 2 + 1];
    float A = 0.f;
    for (int j = 0; j < cpg; ++j) A += s12[((size_t)n * C + g * cpg + j) * 2 + 1];
    gr[c] = gamma[c] * r;
    mu[c] = mean_rstd[(n * groups + g) * 2];
    kk[c] = r * r * r * A / Mm1;
  }
  __syncthreads();
  const int cv = C / VW, vl_n = blockDim.x / cv;
  const int mycv = threadIdx.x % cv, myvl = threadIdx.x / cv, c0 = mycv * VW;
  const bool live = myvl < vl_n;
  float cg[VW], cm[VW], ck[VW], a2[VW], a3[VW];
#pragma unroll
  for (int j = 0; j < VW; ++j) { cg[j] = gr[c0 + j]; cm[j] = mu[c0 + j]; ck[j] = kk[c0 + j]; a2[j] = a3[j] = 0.f; }
  const T* dzb = dz + (size_t)n * voxels * dzpitch + c0;
  const T* xb = x + (size_t)n * voxels * xpitch + c0;
  T* dxb = dx + (size_t)n * voxels * dxpitch + c0;
  float mx = 0.f;
  auto body = [&](const float* g, const float* xx, float* o) {
#pragma unroll
    for (int j = 0; j < VW; ++j) {
      const float sg = sigmoidf_(xx[j]);
      const float num = xx[j] * sg, dnum = sg * (1.f + xx[j] * (1.f - sg));
      a2[j] += g[j] * num;
      a3[j] += g[j] * dnum;
      o[j] = g[j] * cg[j] * dnum - ck[j] * (xx[j] - cm[j]);
    }
    if (amax) {
#pragma unroll
      for (int j = 0; j < VW; j += 2) mx = __builtin_fmaxf(__builtin_fmaxf(mx, __builtin_fabsf(o[j])), __builtin_fabsf(o[j + 1]));
    }
  };
  const size_t stride = (size_t)gridDim.x * vl_n;
  size_t vox = live ? (size_t)blockIdx.x * vl_n + myvl : (size_t)voxels;
  for (; vox + stride < (size_t)voxels; vox += 2 * stride) {
    float g0[VW], x0[VW], g1[VW], x1[VW], o0[VW], o1[VW];
    vload<T, VW, NT>(dzb + vox * dzpitch, g0);
    vload<T, VW, NT>(xb + vox * xpitch, x0);
    vload<T, VW, NT>(dzb + (vox + stride) * dzpitch, g1);
    vload<T, VW, NT>(xb + (vox + stride) * xpitch, x1);
    body(g0, x0, o0);
    body(g1, x1, o1);
    vstore<T, VW, NT>(dxb + vox * dxpitch, o0);
    vstore<T, VW, NT>(dxb + (vox + stride) * dxpitch, o1);
  }
  if (vox < (size_t)voxels) {
    float g0[VW], x0[VW], o0[VW];
    vload<T, VW, NT>(dzb + vox * dzpitch, g0);
    vload<T, VW, NT>(xb + vox * xpitch, x0);
    body(g0, x0, o0);
    vstore<T, VW, NT>(dxb + vox * dxpitch, o0);
  }
  float* scr = sm + 3 * C;
  float* part = side_part + ((size_t)blockIdx.x * gridDim.y + n) * C * 2;
  lane_reduce_plane<VW>(scr, a2, live, myvl, vl_n, C, c0, [&](int c, float t) { part[c * 2] = t; });
  lane_reduce_plane<VW>(scr, a3, live, myvl, vl_n, C, c0, [&](int c, float t) { part[c * 2 + 1] = t; });
  if (amax) record_absmax<T>(mx, amax);
}

// dgamma, dbeta and the bias gradient of the convolution that produced x, from S1, A_g and the side totals
__global__ void __launch_bounds__(256) evonorm_bwd_tiles_finish_kernel(const float* __restrict__ s12, const float* __restrict__ side,
                                                                       const float* __restrict__ mean_rstd, const float* __restrict__ gamma,
                                                                       const double* __restrict__ chan, float* __restrict__ dgamma,
                                                                       float* __restrict__ dbeta, float* __restrict__ dconvbias, int N,
                                                                       int voxels, int C, int groups) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int cpg = C / groups, gg = c / cpg;
  const float Mm1 = (float)cpg * (float)voxels - 1.f;
  float b = 0.f, g = 0.f, dbc = 0.f;
  for (int m = 0; m < N; ++m) {
    const float r = mean_rstd[(m * groups + gg) * 2 + 1], mean = mean_rstd[(m * groups + gg) * 2];
    float A = 0.f;
    for (int j = 0; j < cpg; ++j) A += s12[((size_t)m * C + gg * cpg + j) * 2 + 1];
    b += s12[((size_t)m * C + c) * 2];
    g += side[((size_t)m * C + c) * 2] * r;
    const float sumx = chan ? (float)chan[((size_t)m * C + c) * 2] : 0.f;
    dbc += gamma[c] * r * side[((size_t)m * C + c) * 2 + 1] - r * r * r * A / Mm1 * (sumx - (float)voxels * mean);
  }
  dbeta[c] = b;
  dgamma[c] = g;
  if (dconvbias) dconvbias[c] = dbc;
}

extern "C" int BRATS_API(brats_evonorm_bwd_tiles)(const float* tile_stats, int tiles_per_sample, const void* dz, int dzpitch, const void* x,
                                       int xpitch, const float* mean_rstd, const float* gamma, const float* beta, void* dx, int dxpitch,
                                       float* red, float* dgamma, float* dbeta, const double* chan_sums, float* dconvbias, int dtype,
                                       int N, int voxels, int C, int groups, float* amax, brats_stream_t s) {
  if (!tile_stats || !dz || !x || !dx || !red || !mean_rstd || !gamma || !beta || !dgamma || !dbeta || tiles_per_sample <= 0)
    BRATS_FAIL(BRATS_E_ARG, "evonorm_bwd_tiles: null pointer");
  if (dtype != BRATS_BF16 && dtype != BRATS_F32) BRATS_FAIL(BRATS_E_UNSUPPORTED, "evonorm_bwd_tiles: 16-bit or f32 (split-precision mode) activations");
  if (dconvbias && !chan_sums) BRATS_FAIL(BRATS_E_ARG, "evonorm_bwd_tiles: dconvbias needs the forward per-channel sums");
  const int vw = dtype == BRATS_BF16 ? 8 : 4;
  if (C % vw || C % groups || dzpitch % vw || xpitch % vw || dxpitch % vw || C / vw > 256)
    BRATS_FAIL(BRATS_E_ARG, "evonorm_bwd_tiles: C=%d / pitches must be multiples of %d", C, vw);
  hipStream_t st = (hipStream_t)s;
  float* s12 = red;
  float* side = red + (size_t)N * C * 2;
  float* side_part = side + (size_t)N * C * 2;  // (brats_ordered_sum: the partials directly behind the totals)
  double* part = (double*)(side_part + (((size_t)EVO_SIDE_MAX_BLOCKS * N * C * 2 + 1) / 2) * 2);
  const int splits = gn_splits(tiles_per_sample);
  hipLaunchKernelGGL(gn_chan_reduce_kernel, dim3((C + 15) / 16, N, splits), dim3(256), 0, st, tile_stats, tiles_per_sample, C, part);
  hipLaunchKernelGGL(evonorm_bwd_tiles_prep_kernel, dim3((N * C + 255) / 256), dim3(256), 0, st, part, splits, N, C, groups, mean_rstd, beta, s12);
  const int cv = C / vw, vl = 256 / cv;
  size_t gx = ((size_t)voxels + (size_t)vl * 8 - 1) / ((size_t)vl * 8);
  const bool big = dtype == BRATS_BF16 && stream_nt((size_t)N * voxels * C * 2);
  const size_t cap = big ? EVO_SIDE_MAX_BLOCKS : 512;
  dim3 g2((unsigned)(gx < 1 ? 1 : (gx > cap ? cap : gx)), N);
  const size_t lds2 = (size_t)(3 * C + vl * C) * sizeof(float);
  if (dtype == BRATS_F32)
    hipLaunchKernelGGL((evonorm_bwd_apply_side_kernel<float, false>), g2, dim3(256), lds2, st, (const float*)dz, dzpitch, (const float*)x, xpitch,
                       mean_rstd, gamma, s12, (float*)dx, dxpitch, side_part, voxels, C, groups, (uint32_t*)amax);
  else if (big)
    hipLaunchKernelGGL((evonorm_bwd_apply_side_kernel<bf16_t, true>), g2, dim3(256), lds2, st, (const bf16_t*)dz, dzpitch, (const bf16_t*)x, xpitch,
                       mean_rstd, gamma, s12, (bf16_t*)dx, dxpitch, side_part, voxels, C, groups, (uint32_t*)amax);
  else
    hipLaunchKernelGGL((evonorm_bwd_apply_side_kernel<bf16_t, false>), g2, dim3(256), lds2, st, (const bf16_t*)dz, dzpitch, (const bf16_t*)x, xpitch,
                       mean_rstd, gamma, s12, (bf16_t*)dx, dxpitch, side_part, voxels, C, groups, (uint32_t*)amax);
  BRATS_CHECK_LAUNCH();
  if (int rc = brats_ordered_sum(side_part, side, (int)g2.x, N * C * 2, st)) return rc;
  hipLaunchKernelGGL(evonorm_bwd_tiles_finish_kernel, dim3((C + 255) / 256), dim3(256), 0, st, s12, side, mean_rstd, gamma, chan_sums, dgamma,
                     dbeta, dconvbias, N, voxels, C, groups);
  BRATS_CHECK_LAUNCH();
  return 0;
}

// EvoNorm backward of the layer a ResidualSELayer sits on, with the SE backward in the middle (se.hpp): pass 1 over (dout, x)
// with the five raw sums -> brats_se_bwd_launch (d loss / d gate from the sums; gadd, the SE parameter gradients, the three
// sums for dz = dout * gate1p + gadd) -> pass 2.  Replaces brats_channel_dot + brats_se_bwd + brats_evonorm_bwd(gscale, gadd):
// one pass over two tensors less per block.  ws: brats_chan_ws_floats(N, C, 5) + N * C * 3 floats.
// dlogits != NULL (then dout may be NULL): the block's output feeds only the 1x1x1 output head with K = 3 logit planes -- the
// head's backward is folded in as in brats_gn_act_bwd_head (dout = W_head^T dlogits on the fly; dhw [K][C], dhb [K] out of
// pass 1); hws: brats_gn_bwd_head_ws_floats(N, C, K) floats.
// mode: 0 = dout given, 3 = head fold (dlogits), -1 = pool fold (eh.hf.dskip / dpool / argmax)
static int evonorm_se_bwd_impl(const void* dout, int dopitch, const void* x, int xpitch, const float* mean_rstd, const float* gamma,
                               const float* beta, void* dx, int dxpitch, float* ws, float* dgamma, float* dbeta,
                               const double* chan_sums, float* dconvbias, const float* se_chansum, const float* hidden,
                               const float* gate1p, const float* w1, const float* w2, float* gadd, float* dw1, float* db1, float* dw2,
                               float* db2, int Ch, EvoHead eh, int mode, int K, float* dhw, float* dhb, int dtype, int N, int voxels,
                               int C, int groups, float* amax, hipStream_t st) {
  const int vw = dtype == BRATS_BF16 ? 8 : 4;
  const bool head = mode > 0;
  if (!x || !dx || !ws || !mean_rstd || !gamma || !beta || !gate1p || !gadd) BRATS_FAIL(BRATS_E_ARG, "evonorm_se_bwd: null pointer");
  if (dconvbias && !chan_sums) BRATS_FAIL(BRATS_E_ARG, "evonorm_se_bwd: dconvbias needs the forward per-channel sums");
  if (C % vw || C % groups || (mode == 0 && dopitch % vw) || xpitch % vw || dxpitch % vw || C / vw > 256)
    BRATS_FAIL(BRATS_E_ARG, "evonorm_se_bwd: C=%d / pitches must be multiples of %d", C, vw);
  const int cv = C / vw, vl = 256 / cv;
  size_t gx = ((size_t)voxels + (size_t)vl * 8 - 1) / ((size_t)vl * 8);
  const bool big = dtype == BRATS_BF16 && stream_nt((size_t)N * voxels * C * 2);
  const size_t cap1 = big ? CHAN_MAX_BLOCKS : 512, cap2 = big ? 8192 : 2048;
  dim3 g1((unsigned)(gx < 1 ? 1 : (gx > cap1 ? cap1 : gx)), N);
  const size_t lds1 = (size_t)(vl * C + (head ? vl * K : 0)) * sizeof(float);  // (one value plane at a time)
  dim3 g2((unsigned)(gx < 1 ? 1 : (gx > cap2 ? cap2 : gx)), N);
  const size_t lds2 = (size_t)3 * C * sizeof(float);
  float* raw5 = ws;
  float* red3 = ws + (size_t)(1 + CHAN_MAX_BLOCKS) * N * C * 5;
  eh.mean_rstd = mean_rstd; eh.gamma = gamma; eh.beta = beta; eh.gate1p = gate1p; eh.groups = groups;
  const float* nof = nullptr;
#define EVO_P1(T, NT, HK) hipLaunchKernelGGL((evonorm_bwd_reduce_kernel<T, NT, true, HK>), g1, dim3(256), lds1, st, (const T*)dout, dopitch, \
                                             (const T*)x, xpitch, raw5, voxels, C, nof, nof, eh)
#define EVO_P2(T, NT, HK) hipLaunchKernelGGL((evonorm_bwd_apply_kernel<T, NT, HK>), g2, dim3(256), lds2, st, (const T*)dout, dopitch, \
                                             (const T*)x, xpitch, mean_rstd, gamma, red3, (T*)dx, dxpitch, dgamma, dbeta, chan_sums, \
                                             dconvbias, N, voxels, C, groups, (uint32_t*)amax, gate1p, gadd, eh.hf)
#define EVO_ALL(P, HK) do { if (big) P(bf16_t, true, HK); else if (dtype == BRATS_BF16) P(bf16_t, false, HK); else P(float, false, HK); } while (0)
  if (mode > 0) EVO_ALL(EVO_P1, 3); else if (mode < 0) EVO_ALL(EVO_P1, -1); else EVO_ALL(EVO_P1, 0);
  BRATS_CHECK_LAUNCH();
  brats_ordered_sum(raw5 + (size_t)N * C * 5, raw5, (int)g1.x, N * C * 5, st);
  if (head) brats_ordered_sum2(eh.hf.hpart, dhw, K * C, dhb, N * (int)g1.x, K * C + K, st);  // totals into dhw [K][C], dhb [K]
  SeFold fold;
  fold.raw5 = raw5; fold.mean_rstd = mean_rstd; fold.gamma = gamma; fold.beta = beta; fold.red3 = red3; fold.groups = groups;
  fold.voxels = (float)voxels;
  if (int rc = brats_se_bwd_launch(nullptr, fold, se_chansum, 1.f / (float)voxels, hidden, gate1p, w1, w2, gadd, dw1, db1, dw2, db2,
                                   N, C, Ch, st))
    return rc;
  if (mode > 0) EVO_ALL(EVO_P2, 3); else if (mode < 0) EVO_ALL(EVO_P2, -1); else EVO_ALL(EVO_P2, 0);
#undef EVO_ALL
#undef EVO_P1
#undef EVO_P2
  BRATS_CHECK_LAUNCH();
  return 0;
}

// dlogits != NULL (then dout may be NULL): the block's output feeds only the 1x1x1 output head with K = 3 logit planes -- the
// head's backward is folded in as in brats_gn_act_bwd_head (dout = W_head^T dlogits on the fly; dhw [K][C], dhb [K] out of
// pass 1); hws: brats_gn_bwd_head_ws_floats(N, C, K) floats.
extern "C" int BRATS_API(brats_evonorm_se_bwd)(const void* dout, int dopitch, const void* x, int xpitch, const float* mean_rstd,
                                    const float* gamma, const float* beta, void* dx, int dxpitch, float* ws, float* dgamma,
                                    float* dbeta, const double* chan_sums, float* dconvbias, const float* se_chansum,
                                    const float* hidden, const float* gate1p, const float* w1, const float* w2, float* gadd,
                                    float* dw1, float* db1, float* dw2, float* db2, int Ch, const float* dlogits, const float* hw,
                                    int K, float* hws, float* dhw, float* dhb, int dtype, int N, int voxels, int C, int groups,
                                    float* amax, brats_stream_t s) {
  const bool head = dlogits != nullptr;
  if (!dout && !head) BRATS_FAIL(BRATS_E_ARG, "evonorm_se_bwd: null pointer");
  if (head && (!hw || !hws || !dhw || !dhb)) BRATS_FAIL(BRATS_E_ARG, "evonorm_se_bwd: incomplete head arguments");
  if (head && K != 3) BRATS_FAIL(BRATS_E_UNSUPPORTED, "evonorm_se_bwd: the head fold is built for K = 3 logit planes (K=%d)", K);
  EvoHead eh;
  eh.hf.dl = dlogits; eh.hf.w = hw; eh.hf.hpart = hws;
  return evonorm_se_bwd_impl(dout, dopitch, x, xpitch, mean_rstd, gamma, beta, dx, dxpitch, ws, dgamma, dbeta, chan_sums, dconvbias,
                             se_chansum, hidden, gate1p, w1, w2, gadd, dw1, db1, dw2, db2, Ch, eh, head ? 3 : 0, K, dhw, dhb, dtype, N,
                             voxels, C, groups, amax, (hipStream_t)s);
}

// The same for a block that ends an encoder level (block -> MaxAvgPool / MaxPool3d, its output also the skip connection): the
// block's output gradient = dskip + pooling-backward(dpool) is composed inside both passes from the pieces and the arg-max
// bytes of brats_maxpool2_fwd (with_avg: dpool holds [max | mean], 2C channels) -- replaces brats_maxpool2_bwd_idx +
// brats_evonorm_se_bwd(dout).
extern "C" int BRATS_API(brats_evonorm_se_bwd_pool)(const void* dskip, int dskip_pitch, const void* dpool, int dpool_pitch,
                                         const unsigned char* argmax, int with_avg, int D, int H, int W, const void* x, int xpitch,
                                         const float* mean_rstd, const float* gamma, const float* beta, void* dx, int dxpitch,
                                         float* ws, float* dgamma, float* dbeta, const double* chan_sums, float* dconvbias,
                                         const float* se_chansum, const float* hidden, const float* gate1p, const float* w1,
                                         const float* w2, float* gadd, float* dw1, float* db1, float* dw2, float* db2, int Ch,
                                         int dtype, int N, int C, int groups, float* amax, brats_stream_t s) {
  const int vw = dtype == BRATS_BF16 ? 8 : 4;
  if (!dskip || !dpool || !argmax || ((D | H | W) & 1) || dskip_pitch % vw || dpool_pitch % vw)
    BRATS_FAIL(BRATS_E_ARG, "evonorm_se_bwd_pool: null pointer / odd spatial size / pitch not a multiple of %d", vw);
  EvoHead eh;
  eh.hf.dskip = dskip; eh.hf.dpool = dpool; eh.hf.argmax = argmax; eh.hf.dskip_pitch = dskip_pitch; eh.hf.dpool_pitch = dpool_pitch;
  eh.hf.D = D; eh.hf.H = H; eh.hf.W = W; eh.hf.with_avg = with_avg;
  return evonorm_se_bwd_impl(nullptr, vw, x, xpitch, mean_rstd, gamma, beta, dx, dxpitch, ws, dgamma, dbeta, chan_sums, dconvbias,
                             se_chansum, hidden, gate1p, w1, w2, gadd, dw1, db1, dw2, db2, Ch, eh, -1, 0, nullptr, nullptr, dtype, N,
                             D * H * W, C, groups, amax, (hipStream_t)s);
}

// per-(n, channel) reduction over voxels: out[n][c] = sum_v a[v][c] * (b ? b[v][c] : 1)
// (global average pool of the SE layer and its backward dot product)
template <typename T, bool NT = false>
__global__ void channel_dot_kernel(const T* __restrict__ a, int apitch, const T* __restrict__ b, int bpitch,
                                   float* __restrict__ out, int voxels, int C) {
  constexpr int VW = 16 / sizeof(T);
  extern __shared__ float sm[];
  const int n = blockIdx.y;
  const int cv = C / VW, vl_n = blockDim.x / cv;
  const int mycv = threadIdx.x % cv, myvl = threadIdx.x / cv, c0 = mycv * VW;
  float acc[VW];
#pragma unroll
  for (int j = 0; j < VW; ++j) acc[j] = 0.f;
  if (myvl < vl_n) {
    const T* ab = a + (size_t)n * voxels * apitch;
    const T* bb = b ? b + (size_t)n * voxels * bpitch : nullptr;
    const size_t stride = (size_t)gridDim.x * vl_n;
    size_t vox = (size_t)blockIdx.x * vl_n + myvl;
    if (bb) {
      for (; vox + stride < (size_t)voxels; vox += 2 * stride) {
        float x0[VW], y0[VW], x1[VW], y1[VW];
        vload<T, VW, NT>(ab + vox * apitch + c0, x0);
        vload<T, VW, NT>(bb + vox * bpitch + c0, y0);
        vload<T, VW, NT>(ab + (vox + stride) * apitch + c0, x1);
        vload<T, VW, NT>(bb + (vox + stride) * bpitch + c0, y1);
#pragma unroll
        for (int j = 0; j < VW; ++j) acc[j] += x0[j] * y0[j];
#pragma unroll
        for (int j = 0; j < VW; ++j) acc[j] += x1[j] * y1[j];
      }
      if (vox < (size_t)voxels) {
        float x0[VW], y0[VW];
        vload<T, VW, NT>(ab + vox * apitch + c0, x0);
        vload<T, VW, NT>(bb + vox * bpitch + c0, y0);
#pragma unroll
        for (int j = 0; j < VW; ++j) acc[j] += x0[j] * y0[j];
      }
    } else {
      for (; vox + stride < (size_t)voxels; vox += 2 * stride) {
        float x0[VW], x1[VW];
        vload<T, VW, NT>(ab + vox * apitch + c0, x0);
        vload<T, VW, NT>(ab + (vox + stride) * apitch + c0, x1);
#pragma unroll
        for (int j = 0; j < VW; ++j) acc[j] += x0[j];
#pragma unroll
        for (int j = 0; j < VW; ++j) acc[j] += x1[j];
      }
      if (vox < (size_t)voxels) {
        float x0[VW];
        vload<T, VW, NT>(ab + vox * apitch + c0, x0);
#pragma unroll
        for (int j = 0; j < VW; ++j) acc[j] += x0[j];
      }
    }
  }
  if (myvl < vl_n) {
#pragma unroll
    for (int j = 0; j < VW; ++j) sm[myvl * C + c0 + j] = acc[j];
  }
  __syncthreads();
  float* part = out + (size_t)gridDim.y * C + ((size_t)blockIdx.x * gridDim.y + n) * C;  // per-block partials
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float t = 0.f;
    for (int l = 0; l < vl_n; ++l) t += sm[l * C + c];
    part[c] = t;
  }
}

extern "C" int BRATS_API(brats_channel_dot)(const void* a, int apitch, const void* b, int bpitch, float* out, int dtype, int N, int voxels,
                                 int C, brats_stream_t s) {
  const int vw = dtype == BRATS_BF16 ? 8 : 4;
  if (!a || !out || C % vw || apitch % vw || (b && bpitch % vw) || C / vw > 256) BRATS_FAIL(BRATS_E_ARG, "channel_dot: bad argument");
  hipStream_t st = (hipStream_t)s;
  const int cv = C / vw, vl = 256 / cv;
  size_t gx = ((size_t)voxels + (size_t)vl * 8 - 1) / ((size_t)vl * 8);
  const bool big = dtype == BRATS_BF16 && stream_nt((size_t)N * voxels * C * 2);
  const size_t cap = big ? CHAN_MAX_BLOCKS : 512;
  dim3 grid((unsigned)(gx < 1 ? 1 : (gx > cap ? cap : gx)), N);
  const size_t lds = (size_t)vl * C * sizeof(float);
  if (big)
    hipLaunchKernelGGL((channel_dot_kernel<bf16_t, true>), grid, dim3(256), lds, st, (const bf16_t*)a, apitch, (const bf16_t*)b, bpitch, out, voxels, C);
  else if (dtype == BRATS_BF16)
    hipLaunchKernelGGL(channel_dot_kernel<bf16_t>, grid, dim3(256), lds, st, (const bf16_t*)a, apitch, (const bf16_t*)b, bpitch, out, voxels, C);
  else
    hipLaunchKernelGGL(channel_dot_kernel<float>, grid, dim3(256), lds, st, (const float*)a, apitch, (const float*)b, bpitch, out, voxels, C);
  brats_ordered_sum(out + (size_t)N * C, out, (int)grid.x, N * C, st);
  BRATS_CHECK_LAUNCH();
  return 0;
}

// dst[v][c] = a[v][c]*sa[n][c] (+ b[v][c]*sb[n][c]) (+ add[n][c])  -- SE scale / residual and their backward
template <typename T, bool NT = false>
__global__ void channel_scale_kernel(const T* __restrict__ a, int apitch, const float* __restrict__ sa, const float* __restrict__ add,
                                     T* __restrict__ dst, int dpitch, int voxels, int C, uint32_t* __restrict__ amax) {
  constexpr int VW = 16 / sizeof(T);
  extern __shared__ float sm[];  // sa[C], add[C]
  const int n = blockIdx.y;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    sm[c] = sa[(size_t)n * C + c];
    sm[C + c] = add ? add[(size_t)n * C + c] : 0.f;
  }
  __syncthreads();
  const int cv = C / VW, vl_n = blockDim.x / cv;
  const int mycv = threadIdx.x % cv, myvl = threadIdx.x / cv, c0 = mycv * VW;
  const bool live = myvl < vl_n;
  float mx = 0.f;
  float cs[VW], ca[VW];
#pragma unroll
  for (int j = 0; j < VW; ++j) { cs[j] = sm[c0 + j]; ca[j] = sm[C + c0 + j]; }
  const T* ab = a + (size_t)n * voxels * apitch + c0;
  T* db = dst + (size_t)n * voxels * dpitch + c0;
  const size_t stride = (size_t)gridDim.x * vl_n;
  size_t vox = live ? (size_t)blockIdx.x * vl_n + myvl : (size_t)voxels;
  for (; vox + stride < (size_t)voxels; vox += 2 * stride) {
    float x0[VW], x1[VW];
    vload<T, VW, NT>(ab + vox * apitch, x0);
    vload<T, VW, NT>(ab + (vox + stride) * apitch, x1);
#pragma unroll
    for (int j = 0; j < VW; ++j) { x0[j] = x0[j] * cs[j] + ca[j]; x1[j] = x1[j] * cs[j] + ca[j]; }
    if (amax) {
#pragma unroll
      for (int j = 0; j < VW; ++j) mx = __builtin_fmaxf(__builtin_fmaxf(mx, __builtin_fabsf(x0[j])), __builtin_fabsf(x1[j]));
    }
    vstore<T, VW, NT>(db + vox * dpitch, x0);
    vstore<T, VW, NT>(db + (vox + stride) * dpitch, x1);
  }
  if (vox < (size_t)voxels) {
    float x0[VW];
    vload<T, VW, NT>(ab + vox * apitch, x0);
#pragma unroll
    for (int j = 0; j < VW; ++j) x0[j] = x0[j] * cs[j] + ca[j];
    if (amax) {
#pragma unroll
      for (int j = 0; j < VW; ++j) mx = __builtin_fmaxf(mx, __builtin_fabsf(x0[j]));
    }
    vstore<T, VW, NT>(db + vox * dpitch, x0);
  }
  if (amax) record_absmax<T>(mx, amax);
}

extern "C" int BRATS_API(brats_channel_scale)(const void* a, int apitch, const float* scale, const float* add, void* dst, int dpitch,
                                   int dtype, int N, int voxels, int C, float* amax, brats_stream_t s) {
  const int vw = dtype == BRATS_BF16 ? 8 : 4;
  if (!a || !scale || !dst || C % vw || apitch % vw || dpitch % vw) BRATS_FAIL(BRATS_E_ARG, "channel_scale: bad argument");
  const bool big = dtype == BRATS_BF16 && stream_nt((size_t)N * voxels * C * 2);
  dim3 grid(stream_grid((size_t)voxels * (C / vw), 256) * (big ? 2 : 1), N);  // (4096 blocks per sample; 8192 for the large tensors)
  if (big)
    hipLaunchKernelGGL((channel_scale_kernel<bf16_t, true>), grid, dim3(256), 2 * C * sizeof(float), (hipStream_t)s, (const bf16_t*)a, apitch,
                       scale, add, (bf16_t*)dst, dpitch, voxels, C, (uint32_t*)amax);
  else if (dtype == BRATS_BF16)
    hipLaunchKernelGGL(channel_scale_kernel<bf16_t>, grid, dim3(256), 2 * C * sizeof(float), (hipStream_t)s, (const bf16_t*)a, apitch,
                       scale, add, (bf16_t*)dst, dpitch, voxels, C, (uint32_t*)amax);
  else
    hipLaunchKernelGGL(channel_scale_kernel<float>, grid, dim3(256), 2 * C * sizeof(float), (hipStream_t)s, (const float*)a, apitch,
                       scale, add, (float*)dst, dpitch, voxels, C, (uint32_t*)amax);
  BRATS_CHECK_LAUNCH();
  return 0;
}
#include "twin_end.hpp"
