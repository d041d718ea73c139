// 2x2x2 max / max+avg pooling and align_corners=True trilinear up-sampling (forward + adjoint), NDHWC.
// All HBM-bound: 16-byte channel vectors per thread, coalesced along the channel-minor layout.
// Reference: nn.MaxPool3d(2,2) networks/equiunet2020.py:433; MONAI MaxAvgPool networks/equiunet2021.py:261
// (cat([max, avg], dim=1)); nn.Upsample(trilinear, align_corners=True) networks/equiunet2020.py:439.
#include "twin_begin.hpp"
#include "common.hpp"

static inline int stream_grid(size_t total, int block) {
  size_t b = (total + block - 1) / block;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

// ---- pooling -----------------------------------------------------------------------------------
// argmax (optional): one byte per (pooled voxel, channel) = the window index 0..7 (d, h, w order) of torch's first arg-max;
// brats_maxpool2_bwd_idx reads these instead of the window
template <typename T>
__global__ void maxpool2_fwd_kernel(const T* __restrict__ x, int xpitch, T* __restrict__ y, int ypitch, int N, int C,
                                    int D, int H, int W, int with_avg, uint8_t* __restrict__ argmax) {
  constexpr int VW = 16 / sizeof(T);
  const int cv = C / VW, Do = D / 2, Ho = H / 2, Wo = W / 2;
  const size_t total = (size_t)N * Do * Ho * Wo * cv;
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (size_t)gridDim.x * blockDim.x) {
    const int c0 = (int)(it % cv) * VW;
    size_t v = it / cv;
    const int xo = v % Wo; v /= Wo;
    const int yo = v % Ho; v /= Ho;
    const int zo = v % Do;
    const int n = (int)(v / Do);
    float mx[VW], sm[VW];
    int am[VW];
#pragma unroll
    for (int j = 0; j < VW; ++j) { mx[j] = -INFINITY; sm[j] = 0.f; am[j] = 0; }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int z = 2 * zo + (k >> 2), yy = 2 * yo + ((k >> 1) & 1), xx = 2 * xo + (k & 1);
      float a[VW];
      Vec<T, VW>::load(x + ((((size_t)n * D + z) * H + yy) * W + xx) * xpitch + c0, a);
#pragma unroll
      for (int j = 0; j < VW; ++j) {
        if (a[j] > mx[j] || a[j] != a[j]) { mx[j] = a[j]; am[j] = k; }
        sm[j] += a[j];
      }
    }
    const size_t pvox = (((size_t)n * Do + zo) * Ho + yo) * Wo + xo;
    T* yo_p = y + pvox * ypitch;
    if (argmax) {
      uint32_t* ap = (uint32_t*)(argmax + pvox * C + c0);
#pragma unroll
      for (int q = 0; q < VW / 4; ++q) ap[q] = am[4 * q] | (am[4 * q + 1] << 8) | (am[4 * q + 2] << 16) | (am[4 * q + 3] << 24);
    }
    Vec<T, VW>::store(yo_p + c0, mx);
    if (with_avg) {
#pragma unroll
      for (int j = 0; j < VW; ++j) sm[j] *= 0.125f;
      Vec<T, VW>::store(yo_p + C + c0, sm);
    }
  }
}

// one thread per pooled voxel x channel vector: recompute the first arg-max (torch tie rule:
// strict '>' in d,h,w scan order) and write all 8 input-gradient voxels of the window.
template <typename T>
__global__ void maxpool2_bwd_kernel(const T* __restrict__ x, int xpitch, const T* __restrict__ dy, int dypitch,
                                    const T* __restrict__ dxs, int dxspitch, T* __restrict__ dx, int dxpitch, int N, int C,
                                    int D, int H, int W, int with_avg) {
  constexpr int VW = 16 / sizeof(T);
  const int cv = C / VW, Do = D / 2, Ho = H / 2, Wo = W / 2;
  const size_t total = (size_t)N * Do * Ho * Wo * cv;
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (size_t)gridDim.x * blockDim.x) {
    const int c0 = (int)(it % cv) * VW;
    size_t v = it / cv;
    const int xo = v % Wo; v /= Wo;
    const int yo = v % Ho; v /= Ho;
    const int zo = v % Do;
    const int n = (int)(v / Do);
    float a[8][VW];
    float mx[VW];
    int am[VW];
#pragma unroll
    for (int j = 0; j < VW; ++j) { mx[j] = -INFINITY; am[j] = 0; }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int z = 2 * zo + (k >> 2), yy = 2 * yo + ((k >> 1) & 1), xx = 2 * xo + (k & 1);
      Vec<T, VW>::load(x + ((((size_t)n * D + z) * H + yy) * W + xx) * xpitch + c0, a[k]);
#pragma unroll
      for (int j = 0; j < VW; ++j)
        if (a[k][j] > mx[j] || a[k][j] != a[k][j]) { mx[j] = a[k][j]; am[j] = k; }
    }
    const T* dyp = dy + ((((size_t)n * Do + zo) * Ho + yo) * Wo + xo) * dypitch;
    float g[VW], ga[VW];
    Vec<T, VW>::load(dyp + c0, g);
#pragma unroll
    for (int j = 0; j < VW; ++j) ga[j] = 0.f;
    if (with_avg) {
      Vec<T, VW>::load(dyp + C + c0, ga);
#pragma unroll
      for (int j = 0; j < VW; ++j) ga[j] *= 0.125f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int z = 2 * zo + (k >> 2), yy = 2 * yo + ((k >> 1) & 1), xx = 2 * xo + (k & 1);
      const size_t vox = (((size_t)n * D + z) * H + yy) * W + xx;
      float o[VW];
#pragma unroll
      for (int j = 0; j < VW; ++j) o[j] = ga[j] + (am[j] == k ? g[j] : 0.f);
      if (dxs) {
        float sk[VW];
        Vec<T, VW>::load(dxs + vox * dxspitch + c0, sk);
#pragma unroll
        for (int j = 0; j < VW; ++j) o[j] += sk[j];
      }
      Vec<T, VW>::store(dx + vox * dxpitch + c0, o);
    }
  }
}

// The same backward from a RECORDED arg-max (one byte per pooled voxel and channel, written by brats_affine_act_pool_fwd): the
// 8 window voxels of x (403 MB at 2 x 48 x 128^3) are not read again.  A thread owns one x position x one 16-byte channel
// vector and the 2 x 2 (z, y) voxels above it: its 4 skip-gradient loads and 4 stores are contiguous runs of a row across the
// lanes; the pooled gradient and the arg-max bytes are read by both threads of an x pair.  Same arithmetic as
// maxpool2_bwd_kernel ((avg / 8 + [k == argmax] * dy) + skip): bit-identical.
template <typename T>
__global__ void __launch_bounds__(256) maxpool2_bwd_idx_kernel(const uint8_t* __restrict__ argmax, const T* __restrict__ dy, int dypitch,
                                                               const T* __restrict__ dxs, int dxspitch, T* __restrict__ dx, int dxpitch,
                                                               int C, int D, int H, int W, int with_avg) {
  constexpr int VW = 16 / sizeof(T);
  const int n = blockIdx.y;
  const int cv = C / VW;
  int xb = blockDim.x / cv;
  if (xb > W) xb = W;
  const int mycv = threadIdx.x % cv, myvl = threadIdx.x / cv, c0 = mycv * VW;
  const int Do = D / 2, Ho = H / 2, Wo = W / 2;
  const int segs = (W + xb - 1) / xb;
  const size_t items = (size_t)Do * Ho * segs, voxels = (size_t)D * H * W, pvoxels = (size_t)Do * Ho * Wo;
  if (myvl >= xb) return;
  const T* dyb = dy + (size_t)n * pvoxels * dypitch + c0;
  const T* sb = dxs ? dxs + (size_t)n * voxels * dxspitch + c0 : nullptr;
  T* ob = dx + (size_t)n * voxels * dxpitch + c0;
  const uint8_t* ab = argmax + (size_t)n * pvoxels * C + c0;
  for (size_t item = blockIdx.x; item < items; item += gridDim.x) {
    const int seg = (int)(item % segs);
    const size_t rp = item / segs;
    const int yo = (int)(rp % Ho), zo = (int)(rp / Ho);
    const int x = seg * xb + myvl;
    if (x >= W) continue;
    const size_t pvox = ((size_t)zo * Ho + yo) * Wo + (x >> 1);
    float sk[4][VW];
    size_t vox[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      vox[k] = ((size_t)(2 * zo + (k >> 1)) * H + (2 * yo + (k & 1))) * W + x;
      if (sb) Vec<T, VW>::load(sb + vox[k] * dxspitch, sk[k]);
    }
    uint32_t aw[VW / 4];
#pragma unroll
    for (int q = 0; q < VW / 4; ++q) aw[q] = ((const uint32_t*)(ab + pvox * C))[q];
    float g[VW], ga[VW];
    Vec<T, VW>::load(dyb + pvox * dypitch, g);
#pragma unroll
    for (int j = 0; j < VW; ++j) ga[j] = 0.f;
    if (with_avg) {
      Vec<T, VW>::load(dyb + pvox * dypitch + C, ga);
#pragma unroll
      for (int j = 0; j < VW; ++j) ga[j] *= 0.125f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int kk = 2 * k + (x & 1);  // the voxel's index in the window, d, h, w order
      float o[VW];
#pragma unroll
      for (int j = 0; j < VW; ++j) {
        const int am = (aw[j >> 2] >> (8 * (j & 3))) & 0xff;
        o[j] = ga[j] + (am == kk ? g[j] : 0.f);
        if (sb) o[j] += sk[k][j];
      }
      Vec<T, VW>::store(ob + vox[k] * dxpitch, o);
    }
  }
}

extern "C" int BRATS_API(brats_maxpool2_bwd_idx)(const unsigned char* argmax, const void* dy, int dypitch, const void* dx_skip,
                                      int dxskip_pitch, void* dx, int dxpitch, int dtype, int N, int C, int D, int H, int W,
                                      int with_avg, brats_stream_t s) {
  const int vw = dtype == BRATS_BF16 ? 8 : 4;
  if (!argmax || !dy || !dx || C % vw || dypitch % vw || dxpitch % vw || (dx_skip && dxskip_pitch % vw) || ((D | H | W) & 1) ||
      C / vw > 256)
    BRATS_FAIL(BRATS_E_ARG, "maxpool2_bwd_idx: bad argument");
  int xb = 256 / (C / vw);
  if (xb > W) xb = W;
  const size_t items = (size_t)(D / 2) * (H / 2) * ((W + xb - 1) / xb);
  dim3 grid((unsigned)(items < 1 ? 1 : (items > 8192 ? 8192 : items)), N);
  if (dtype == BRATS_BF16)
    hipLaunchKernelGGL(maxpool2_bwd_idx_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)s, argmax, (const bf16_t*)dy, dypitch,
                       (const bf16_t*)dx_skip, dxskip_pitch, (bf16_t*)dx, dxpitch, C, D, H, W, with_avg);
  else
    hipLaunchKernelGGL(maxpool2_bwd_idx_kernel<float>, grid, dim3(256), 0, (hipStream_t)s, argmax, (const float*)dy, dypitch,
                       (const float*)dx_skip, dxskip_pitch, (float*)dx, dxpitch, C, D, H, W, with_avg);
  BRATS_CHECK_LAUNCH();
  return 0;
}

extern "C" int BRATS_API(brats_maxpool2_fwd)(const void* x, int xpitch, void* y, int ypitch, unsigned char* argmax, int dtype, int N,
                                  int C, int D, int H, int W, int with_avg, brats_stream_t s) {
  const int vw = dtype == BRATS_BF16 ? 8 : 4;
  if (!x || !y || C % vw || xpitch % vw || ypitch % vw || (D | H | W) & 1)
    BRATS_FAIL(BRATS_E_ARG, "maxpool2_fwd: C/pitch multiple of %d and even spatial dims required", vw);
  const size_t total = (size_t)N * (D / 2) * (H / 2) * (W / 2) * (C / vw);
  if (dtype == BRATS_BF16)
    hipLaunchKernelGGL(maxpool2_fwd_kernel<bf16_t>, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)s,
                       (const bf16_t*)x, xpitch, (bf16_t*)y, ypitch, N, C, D, H, W, with_avg, argmax);
  else
    hipLaunchKernelGGL(maxpool2_fwd_kernel<float>, dim3(stream_grid(total, 256)), dim3(256), 0, (hipStream_t)s,
                       (const float*)x, xpitch, (float*)y, ypitch, N, C, D, H, W, with_avg, argmax);
  BRATS_CHECK_LAUNCH();
  return 0;
}

extern "C" int BRATS_API(brats_maxpool2_bwd)(const void* x, int xpitch, const void* y, int ypitch, const void* dy, int dypitch,
                                  const void* dx_skip, int dxskip_pitch, void* dx, int dxpitch, int dtype, int N, int C,
                                  int D, int H, int W, int with_avg, brats_stream_t s) {
  (void)y; (void)ypitch;  // arg-max is recomputed from x
  const int vw = dtype == BRATS_BF16 ? 8 : 4;
  if (!x || !dy || !dx || C % vw || xpitch % vw || dypitch % vw || dxpitch % vw || (dx_skip && dxskip_pitch % vw) ||
      (D | H | W) & 1)
    BRATS_FAIL(BRATS_E_ARG, "maxpool2_bwd: bad argument");
  const size_t total = (size_t)N * (D / 2) * (H / 2) * (W / 2) * (C / vw);
  if (dtype == BRATS_BF16)
    hipLaunchKernelGGL(maxpool2_bwd_kernel<bf16_t>, dim3(stream_grid(total, 128)), dim3(128), 0, (hipStream_t)s,
                       (const bf16_t*)x, xpitch, (const bf16_t*)dy, dypitch, (const bf16_t*)dx_skip, dxskip_pitch,
                       (bf16_t*)dx, dxpitch, N, C, D, H, W, with_avg);
  else
    hipLaunchKernelGGL(maxpool2_bwd_kernel<float>, dim3(stream_grid(total, 128)), dim3(128), 0, (hipStream_t)s,
                       (const float*)x, xpitch, (const float*)dy, dypitch, (const float*)dx_skip, dxskip_pitch, (float*)dx,
                       dxpitch, N, C, D, H, W, with_avg);
  BRATS_CHECK_LAUNCH();
  return 0;
}

// ---- trilinear, align_corners=True -------------------------------------------------------------
// torch semantics (UpSample.h area_pixel_compute_source_index, align_corners): scale = (in-1)/(out-1)
// in f32, src = scale*dst, i0 = (int)src, i1 = i0 + (i0 < in-1), lambda1 = src - i0.
struct Lerp { int i0, i1; float w0, w1; };
DEVI Lerp lerp_coef(int o, int in_len, float scale) {
  const float src = scale * (float)o;
  Lerp l;
  l.i0 = (int)src;
  if (l.i0 > in_len - 1) l.i0 = in_len - 1;
  l.i1 = l.i0 + (l.i0 < in_len - 1 ? 1 : 0);
  l.w1 = fminf(fmaxf(src - (float)l.i0, 0.f), 1.f);
  l.w0 = 1.f - l.w1;
  return l;
}
static inline float ac_scale(int in_len, int out_len) { return out_len > 1 ? (float)(in_len - 1) / (float)(out_len - 1) : 0.f; }

// One block = one output row (n, zo, yo): the z / y interpolation coefficients and the four source-row pointers are
// scalar; a thread handles (xo, channel vector) items of the row (no per-element div / mod chains).
template <typename T>
__global__ void __launch_bounds__(256) upsample_fwd_kernel(const T* __restrict__ x, int xpitch, T* __restrict__ y, int ypitch,
                                                           int N, int C, int D, int H, int W, int sc, float sd, float sh,
                                                           float sw) {
  constexpr int VW = 16 / sizeof(T);
  const int cv = C / VW, Do = D * sc, Ho = H * sc, Wo = W * sc;
  for (size_t row = blockIdx.x; row < (size_t)N * Do * Ho; row += gridDim.x) {
    const int yo = (int)(row % Ho);
    const int zo = (int)((row / Ho) % Do);
    const int n = (int)(row / ((size_t)Ho * Do));
    const Lerp lz = lerp_coef(zo, D, sd), ly = lerp_coef(yo, H, sh);
    const T* xb = x + (size_t)n * D * H * W * xpitch;
    const T* r00 = xb + (size_t)(lz.i0 * H + ly.i0) * W * xpitch;
    const T* r01 = xb + (size_t)(lz.i0 * H + ly.i1) * W * xpitch;
    const T* r10 = xb + (size_t)(lz.i1 * H + ly.i0) * W * xpitch;
    const T* r11 = xb + (size_t)(lz.i1 * H + ly.i1) * W * xpitch;
    T* yb = y + row * Wo * ypitch;
    for (int it = threadIdx.x; it < Wo * cv; it += blockDim.x) {
      const int xo = it / cv, c0 = (it % cv) * VW;
      const Lerp lx = lerp_coef(xo, W, sw);
      float o[VW];
#pragma unroll
      for (int j = 0; j < VW; ++j) o[j] = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const T* r = (k & 4) ? ((k & 2) ? r11 : r10) : ((k & 2) ? r01 : r00);
        const int xx = (k & 1) ? lx.i1 : lx.i0;
        // same association as ATen's upsample_trilinear3d: w_z * (w_y * (w_x * v ...)) -> use product of weights
        const float w = ((k & 4) ? lz.w1 : lz.w0) * ((k & 2) ? ly.w1 : ly.w0) * ((k & 1) ? lx.w1 : lx.w0);
        float a[VW];
        Vec<T, VW>::load(r + (size_t)xx * xpitch + c0, a);
#pragma unroll
        for (int j = 0; j < VW; ++j) o[j] += w * a[j];
      }
      Vec<T, VW>::store(yb + (size_t)xo * ypitch + c0, o);
    }
  }
}

// scale 2 (the decoder's up-sampling): one block = 2 output z-planes x 4 output y-rows of one sample, a thread = one (xo,
// channel vector) column of that 2 x 4 patch.  The 8 outputs read the same 3 input planes x 4 input rows, so a thread does
// 24 gathers for 8 outputs instead of 64 (the row-per-block kernel above is bound by its 8 gathers per 16-byte store), and
// every store instruction still writes a contiguous x-row.  Weights of absent (plane, row) pairs are zero; same f32
// coefficients as lerp_coef, summed x first, then y, then z.
// LDSX: the 3 planes x 4 rows the block reads are first copied into LDS with dense 16-byte loads (every input element
// fetched once per block: 18 load instructions per thread instead of 72 gathers); 12 x W x C elements must fit (73.7 KB at
// every decoder level of the bf16 networks: W x C = 3072).
template <typename T, bool NT = false, bool LDSX = false>
__global__ void __launch_bounds__(256) upsample2_fwd_kernel(const T* __restrict__ x, int xpitch, T* __restrict__ y, int ypitch,
                                                            int C, int D, int H, int W, float sd, float sh, float sw) {
  constexpr int VW = 16 / sizeof(T);
  extern __shared__ __attribute__((aligned(16))) char up_lds[];
  const int cv = C / VW, Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
  const int yq = blockIdx.x, zp = blockIdx.y, n = blockIdx.z;
  const int zo0 = 2 * zp, yo0 = 4 * yq;
  Lerp lz[2], ly[4];
#pragma unroll
  for (int a = 0; a < 2; ++a) lz[a] = lerp_coef(zo0 + a, D, sd);
#pragma unroll
  for (int b = 0; b < 4; ++b) ly[b] = lerp_coef(yo0 + b, H, sh);
  const int zb = lz[0].i0, yb = ly[0].i0;
  float wz[2][3], wy[4][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int pz = 0; pz < 3; ++pz) wz[a][pz] = (lz[a].i0 == zb + pz ? lz[a].w0 : 0.f) + (lz[a].i1 == zb + pz ? lz[a].w1 : 0.f);
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int r = 0; r < 4; ++r) wy[b][r] = (ly[b].i0 == yb + r ? ly[b].w0 : 0.f) + (ly[b].i1 == yb + r ? ly[b].w1 : 0.f);
  const T* xb = x + (size_t)n * D * H * W * xpitch;
  T* yb_ = y + (size_t)n * Do * Ho * Wo * ypitch;
  if constexpr (LDSX) {
    const int row_pieces = W * cv;
    for (int q = threadIdx.x; q < 12 * row_pieces; q += blockDim.x) {
      const int rr = q / row_pieces, w = q % row_pieces;
      const int zi = zb + rr / 4 < D ? zb + rr / 4 : D - 1, yi = yb + rr % 4 < H ? yb + rr % 4 : H - 1;
      *(u32x4*)(up_lds + (size_t)q * 16) =
          *(const u32x4*)(xb + ((size_t)(zi * H + yi) * W + w / cv) * xpitch + (w % cv) * VW);
    }
    __syncthreads();
  }
  // pairs of channels through explicit v_pk_fma_f32: with separate multiplies and adds (the build's -ffp-contract=off) the
  // ~1400 packed ops per 8 outputs made this kernel VALU-bound at 2.3 TB/s
  typedef __attribute__((ext_vector_type(2))) float f2;
  auto fma2 = [](float w, f2 v, f2 acc) { return __builtin_elementwise_fma(f2{w, w}, v, acc); };
  constexpr int V2 = VW / 2;
  for (int it = threadIdx.x; it < Wo * cv; it += blockDim.x) {
    const int xo = it / cv, c0 = (it % cv) * VW;
    const Lerp lx = lerp_coef(xo, W, sw);
    f2 o[2][4][V2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int j = 0; j < V2; ++j) o[a][b][j] = f2{0.f, 0.f};
#pragma unroll
    for (int pz = 0; pz < 3; ++pz) {
      const int zi = zb + pz < D ? zb + pz : D - 1;   // (a clamped plane / row carries zero weight)
      f2 t[4][V2];                                    // y-interpolated rows of this plane, per output row b
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int j = 0; j < V2; ++j) t[b][j] = f2{0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int yi = yb + r < H ? yb + r : H - 1;
        float v0[VW], v1[VW];
        if constexpr (LDSX) {
          const T* row = (const T*)up_lds + (size_t)(pz * 4 + r) * W * C + c0;
          Vec<T, VW>::load(row + (size_t)lx.i0 * C, v0);
          Vec<T, VW>::load(row + (size_t)lx.i1 * C, v1);
        } else {
          const T* row = xb + (size_t)(zi * H + yi) * W * xpitch + c0;
          Vec<T, VW>::load(row + (size_t)lx.i0 * xpitch, v0);
          Vec<T, VW>::load(row + (size_t)lx.i1 * xpitch, v1);
        }
#pragma unroll
        for (int j = 0; j < V2; ++j) {
          const f2 xl = fma2(lx.w1, f2{v1[2 * j], v1[2 * j + 1]}, lx.w0 * f2{v0[2 * j], v0[2 * j + 1]});
#pragma unroll
          for (int b = 0; b < 4; ++b) t[b][j] = fma2(wy[b][r], xl, t[b][j]);
        }
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int j = 0; j < V2; ++j) o[a][b][j] = fma2(wz[a][pz], t[b][j], o[a][b][j]);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        float of[VW];
#pragma unroll
        for (int j = 0; j < V2; ++j) { of[2 * j] = o[a][b][j][0]; of[2 * j + 1] = o[a][b][j][1]; }
        vstore<T, VW, NT>(yb_ + ((size_t)((zo0 + a) * Ho + yo0 + b) * Wo + xo) * ypitch + c0, of);
      }
  }
}

extern "C" int BRATS_API(brats_upsample_fwd)(const void* x, int xpitch, void* y, int ypitch, int dtype, int N, int C, int D, int H,
                                  int W, int scale, brats_stream_t s) {
  const int vw = dtype == BRATS_BF16 ? 8 : 4;
  if (!x || !y || C % vw || xpitch % vw || ypitch % vw || scale < 1)
    BRATS_FAIL(BRATS_E_ARG, "upsample_fwd: C/pitch must be multiples of %d", vw);
  const size_t rows = (size_t)N * D * scale * H * scale;
  const unsigned grid = (unsigned)(rows > 65536 ? 65536 : rows);
  const float sd = ac_scale(D, D * scale), sh = ac_scale(H, H * scale), sw = ac_scale(W, W * scale);
  if (scale == 2 && H % 2 == 0 && N <= 65535 && D <= 65535) {
    const dim3 g2(H / 2, D, N);  // (Ho / 4, Do / 2, N)
    const size_t ldsx = (size_t)12 * W * C * 2;
    if (dtype == BRATS_BF16 && ldsx <= 80 * 1024) {  // two blocks per CU with their input rows in LDS
      static std::atomic<uint64_t> attr_a{0}, attr_b{0};
      BRATS_ENSURE_LDS_ATTR((upsample2_fwd_kernel<bf16_t, true, true>), 80 * 1024, attr_a);
      BRATS_ENSURE_LDS_ATTR((upsample2_fwd_kernel<bf16_t, false, true>), 80 * 1024, attr_b);
      if (stream_nt((size_t)N * D * H * W * 8 * C * 2))
        hipLaunchKernelGGL((upsample2_fwd_kernel<bf16_t, true, true>), g2, dim3(256), ldsx, (hipStream_t)s, (const bf16_t*)x, xpitch, (bf16_t*)y,
                           ypitch, C, D, H, W, sd, sh, sw);
      else
        hipLaunchKernelGGL((upsample2_fwd_kernel<bf16_t, false, true>), g2, dim3(256), ldsx, (hipStream_t)s, (const bf16_t*)x, xpitch, (bf16_t*)y,
                           ypitch, C, D, H, W, sd, sh, sw);
      BRATS_CHECK_LAUNCH();
      return 0;
    }
    if (dtype == BRATS_BF16 && stream_nt((size_t)N * D * H * W * 8 * C * 2))  // output beyond the Infinity Cache: non-temporal stores
      hipLaunchKernelGGL((upsample2_fwd_kernel<bf16_t, true>), g2, dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, xpitch, (bf16_t*)y, ypitch,
                         C, D, H, W, sd, sh, sw);
    else if (dtype == BRATS_BF16)
      hipLaunchKernelGGL(upsample2_fwd_kernel<bf16_t>, g2, dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, xpitch, (bf16_t*)y, ypitch,
                         C, D, H, W, sd, sh, sw);
    else
      hipLaunchKernelGGL(upsample2_fwd_kernel<float>, g2, dim3(256), 0, (hipStream_t)s, (const float*)x, xpitch, (float*)y, ypitch,
                         C, D, H, W, sd, sh, sw);
    BRATS_CHECK_LAUNCH();
    return 0;
  }
  if (dtype == BRATS_BF16)
    hipLaunchKernelGGL(upsample_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, xpitch,
                       (bf16_t*)y, ypitch, N, C, D, H, W, scale, sd, sh, sw);
  else
    hipLaunchKernelGGL(upsample_fwd_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)s, (const float*)x, xpitch,
                       (float*)y, ypitch, N, C, D, H, W, scale, sd, sh, sw);
  BRATS_CHECK_LAUNCH();
  return 0;
}

// 1-D adjoint of the lerp along one axis of a tensor viewed as [outer][L][inner_vox][C(+pitch)]:
// out[o][i][v][c] = sum_{l : i0(l)==i} w0(l)*in[o][l][v][c] + sum_{l : i1(l)==i, i1!=i0} w1(l)*in[..l..].
// Candidates l are re-derived with the *forward's own* f32 expression, so forward and adjoint are
// exactly transposes of each other.
template <typename TI, typename TO, int VW>
__global__ void lerp_adjoint_kernel(const TI* __restrict__ in, int in_pitch, TO* __restrict__ out, int out_pitch,
                                    size_t outer, int Lout /*len of in*/, int Lin /*len of out*/, size_t inner_vox, int C,
                                    float scale) {
  const int cv = C / VW;
  const size_t total = outer * Lin * inner_vox * cv;
  const float inv = scale > 0.f ? 1.f / scale : 0.f;
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (size_t)gridDim.x * blockDim.x) {
    const int c0 = (int)(it % cv) * VW;
    size_t v = it / cv;
    const size_t iv = v % inner_vox; v /= inner_vox;
    const int i = (int)(v % Lin);
    const size_t o = v / Lin;
    int lo = (int)floorf((float)(i - 1) * inv) - 1, hi = (int)ceilf((float)(i + 1) * inv) + 1;
    if (scale <= 0.f) { lo = 0; hi = Lout - 1; }
    lo = lo < 0 ? 0 : lo;
    hi = hi > Lout - 1 ? Lout - 1 : hi;
    float acc[VW];
#pragma unroll
    for (int j = 0; j < VW; ++j) acc[j] = 0.f;
    for (int l = lo; l <= hi; ++l) {
      const Lerp c = lerp_coef(l, Lin, scale);
      float w = 0.f;
      if (c.i0 == i) w += c.w0;
      if (c.i1 == i) w += c.w1;  // when i1 == i0 (clamped end) both weights land on the same index
      if (w != 0.f) {
        float a[VW];
        if constexpr (VW == 1) a[0] = to_f<TI>(in[((o * Lout + l) * inner_vox + iv) * in_pitch + c0]);
        else Vec<TI, VW>::load(in + ((o * Lout + l) * inner_vox + iv) * in_pitch + c0, a);
#pragma unroll
        for (int j = 0; j < VW; ++j) acc[j] += w * a[j];
      }
    }
    if constexpr (VW == 1) out[((o * Lin + i) * inner_vox + iv) * out_pitch + c0] = from_f<TO>(acc[0]);
    else Vec<TO, VW>::store(out + ((o * Lin + i) * inner_vox + iv) * out_pitch + c0, acc);
  }
}

// exported for head.hip: f32 planes [outer][L][inner] adjoint
int brats_lerp_adjoint_f32_planes(const float* in, float* out, size_t outer, int Lout, int Lin, size_t inner,
                                  hipStream_t st) {
  const size_t total = outer * Lin * inner;
  if (inner % 4 == 0 && (((size_t)in | (size_t)out) & 15) == 0) {
    // four neighbouring inner elements per thread (16-byte loads): these planes are bound by the NUMBER of vector-memory
    // instructions (one per candidate l per thread), not by their bytes
    hipLaunchKernelGGL((lerp_adjoint_kernel<float, float, 4>), dim3(stream_grid(total / 4, 256)), dim3(256), 0, st, in, 4, out, 4,
                       outer, Lout, Lin, inner / 4, 4, ac_scale(Lin, Lout));
    BRATS_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL((lerp_adjoint_kernel<float, float, 1>), dim3(stream_grid(total, 256)), dim3(256), 0, st, in, 1, out, 1,
                     outer, Lout, Lin, inner, 1, ac_scale(Lin, Lout));
  BRATS_CHECK_LAUNCH();
  return 0;
}

extern "C" size_t BRATS_API(brats_upsample_bwd_ws_bytes)(int dtype, int N, int C, int D, int H, int W, int scale) {
  const size_t esz = dtype == BRATS_BF16 ? 2 : 4;
  const size_t a = (size_t)N * D * (H * scale) * (W * scale) * C;  // after the D pass
  const size_t b = (size_t)N * D * H * (W * scale) * C;            // after the H pass
  return ((a * esz + 255) / 256 * 256) + ((b * esz + 255) / 256 * 256);
}

template <typename T>
static int upsample_bwd_t(const T* dy, int dypitch, T* dx, int dxpitch, char* tmp, int N, int C, int D, int H, int W, int sc,
                          hipStream_t st) {
  constexpr int VW = 16 / sizeof(T);
  const int Do = D * sc, Ho = H * sc, Wo = W * sc;
  const size_t a_elems = (size_t)N * D * Ho * Wo * C;
  T* t1 = (T*)tmp;
  T* t2 = (T*)(tmp + ((a_elems * sizeof(T) + 255) / 256 * 256));
  // D axis: [N][Do][Ho*Wo][C] -> [N][D][Ho*Wo][C]
  size_t total = (size_t)N * D * Ho * Wo * (C / VW);
  hipLaunchKernelGGL((lerp_adjoint_kernel<T, T, VW>), dim3(stream_grid(total, 256)), dim3(256), 0, st, dy, dypitch, t1, C,
                     (size_t)N, Do, D, (size_t)Ho * Wo, C, ac_scale(D, Do));
  // H axis: [N*D][Ho][Wo][C] -> [N*D][H][Wo][C]
  total = (size_t)N * D * H * Wo * (C / VW);
  hipLaunchKernelGGL((lerp_adjoint_kernel<T, T, VW>), dim3(stream_grid(total, 256)), dim3(256), 0, st, (const T*)t1, C, t2, C,
                     (size_t)N * D, Ho, H, (size_t)Wo, C, ac_scale(H, Ho));
  // W axis: [N*D*H][Wo][1][C] -> [N*D*H][W][1][C(pitch)]
  total = (size_t)N * D * H * W * (C / VW);
  hipLaunchKernelGGL((lerp_adjoint_kernel<T, T, VW>), dim3(stream_grid(total, 256)), dim3(256), 0, st, (const T*)t2, C, dx,
                     dxpitch, (size_t)N * D * H, Wo, W, (size_t)1, C, ac_scale(W, Wo));
  BRATS_CHECK_LAUNCH();
  return 0;
}

extern "C" int BRATS_API(brats_upsample_bwd)(const void* dy, int dypitch, void* dx, int dxpitch, void* tmp, int dtype, int N, int C,
                                  int D, int H, int W, int scale, brats_stream_t s) {
  const int vw = dtype == BRATS_BF16 ? 8 : 4;
  if (!dy || !dx || !tmp || C % vw || dypitch % vw || dxpitch % vw) BRATS_FAIL(BRATS_E_ARG, "upsample_bwd: bad argument");
  if (dtype == BRATS_BF16)
    return upsample_bwd_t<bf16_t>((const bf16_t*)dy, dypitch, (bf16_t*)dx, dxpitch, (char*)tmp, N, C, D, H, W, scale,
                                  (hipStream_t)s);
  return upsample_bwd_t<float>((const float*)dy, dypitch, (float*)dx, dxpitch, (char*)tmp, N, C, D, H, W, scale,
                               (hipStream_t)s);
}
#include "twin_end.hpp"
