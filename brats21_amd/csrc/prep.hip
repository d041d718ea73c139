// Input pipeline on the GPU: the reference prepares every patch with CPU MONAI / numpy
// transforms (src/definer.py:449-467): label -> 3 channels, random crop, rot90, flips, intensity shift / contrast,
// Gaussian noise, and the non-zero per-channel z-score of utils/transforms.py:328-406.  Here: one fused
// crop + signed-permutation + affine-intensity gather, a label converter, a two-pass non-zero z-score and a
// gamma-contrast pass.  NCDHW f32, one thread per output element, x-fastest: pure HBM-bound index kernels.
#include "common.hpp"

static inline int qgrid(size_t total) {
  size_t b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

// dst[p][i0][i1][i2] = scale[p] * src[p][c + j] + shift[p]: dst axis a runs along source axis perm[a] over the crop
// box [c, c + e) (reversed when flip[a]); dst dims = (e[perm[0]], e[perm[1]], e[perm[2]]).  RandSpatialCrop +
// RandRotate90 + RandFlip + RandShiftIntensity (and RandScaleIntensity) in one pass.
__global__ void crop_perm_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t planes, int s0, int s1, int s2,
                                 int c0, int c1, int c2, int d0, int d1, int d2, int p0, int p1, int p2, int f0, int f1, int f2,
                                 const float* __restrict__ scale, const float* __restrict__ shift) {
  const size_t total = planes * d0 * d1 * d2;
  const int org[3] = {c0, c1, c2};
  const int sdim[3] = {s0, s1, s2};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t t = i;
    int di[3];
    di[2] = t % d2; t /= d2;
    di[1] = t % d1; t /= d1;
    di[0] = t % d0;
    const size_t pl = t / d0;
    int sj[3];
    sj[p0] = org[p0] + (f0 ? d0 - 1 - di[0] : di[0]);
    sj[p1] = org[p1] + (f1 ? d1 - 1 - di[1] : di[1]);
    sj[p2] = org[p2] + (f2 ? d2 - 1 - di[2] : di[2]);
    float v = src[((pl * sdim[0] + sj[0]) * sdim[1] + sj[1]) * sdim[2] + sj[2]];
    if (scale) v *= scale[pl];
    if (shift) v += shift[pl];
    dst[i] = v;
  }
}

// out[n][k][v], k = 0..2: order 0 = (TC, WT, ET) (MONAI ConvertToMultiChannelBasedOnBratsClasses, the training
// pipeline src/definer.py:451), order 1 = (WT, TC, ET) (utils/transforms.py:155-166).  Labels: 1 NCR/NET, 2 ED, 4 ET.
__global__ void label_channels_kernel(const float* __restrict__ label, float* __restrict__ out, int N, size_t voxels, int order) {
  const size_t total = (size_t)N * voxels;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t n = i / voxels, v = i % voxels;
    const float l = label[i];
    const float tc = (l == 1.f || l == 4.f) ? 1.f : 0.f;
    const float wt = (l == 1.f || l == 4.f || l == 2.f) ? 1.f : 0.f;
    const float et = l == 4.f ? 1.f : 0.f;
    float* o = out + n * 3 * voxels + v;
    o[0] = order == 0 ? tc : wt;
    o[voxels] = order == 0 ? wt : tc;
    o[2 * voxels] = et;
  }
}

// stats[p] = { count, sum, sum of squares } (f64) over the selected voxels of plane p (x != 0 when nonzero)
__global__ void plane_stats_kernel(const float* __restrict__ x, double* __restrict__ stats, size_t voxels, int nonzero) {
  const int p = blockIdx.y;
  const float* xp = x + (size_t)p * voxels;
  double c = 0.0, s1 = 0.0, s2 = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < voxels; i += (size_t)gridDim.x * blockDim.x) {
    const float v = xp[i];
    if (!nonzero || v != 0.f) { c += 1.0; s1 += (double)v; s2 += (double)v * (double)v; }
  }
  __shared__ double r[3][256];
  r[0][threadIdx.x] = c; r[1][threadIdx.x] = s1; r[2][threadIdx.x] = s2;
  __syncthreads();
  for (int m = 128; m > 0; m >>= 1) {
    if ((int)threadIdx.x < m) {
      r[0][threadIdx.x] += r[0][threadIdx.x + m];
      r[1][threadIdx.x] += r[1][threadIdx.x + m];
      r[2][threadIdx.x] += r[2][threadIdx.x + m];
    }
    __syncthreads();
  }
  if (threadIdx.x < 3) atomicAdd(stats + (size_t)p * 3 + threadIdx.x, r[threadIdx.x][0]);
}

// NormalizeIntensity(nonzero, channel_wise) of utils/transforms.py:363-385: selected voxels become (x - mean)/std
// (std = population std, 1 when 0), optionally clipped to +-clip; unselected (zero) voxels stay 0; a plane without
// selected voxels is returned unchanged.
__global__ void zscore_apply_kernel(const float* __restrict__ x, float* __restrict__ y, const double* __restrict__ stats,
                                    size_t voxels, int nonzero, float clip) {
  const int p = blockIdx.y;
  const double cnt = stats[(size_t)p * 3];
  double mean = 0.0, sd = 1.0;
  if (cnt > 0.0) {
    mean = stats[(size_t)p * 3 + 1] / cnt;
    const double var = stats[(size_t)p * 3 + 2] / cnt - mean * mean;
    sd = var > 0.0 ? sqrt(var) : 1.0;
    if (sd == 0.0) sd = 1.0;
  }
  const float m = (float)mean, sdev = (float)sd;
  const float* xp = x + (size_t)p * voxels;
  float* yp = y + (size_t)p * voxels;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < voxels; i += (size_t)gridDim.x * blockDim.x) {
    float v = xp[i];
    if (cnt > 0.0 && (!nonzero || v != 0.f)) {
      v = (v - m) / sdev;
      if (clip > 0.f) v = fminf(fmaxf(v, -clip), clip);
    }
    yp[i] = v;
  }
}

// AdjustContrast (MONAI 0.6): y = ((x - mn) / (rg + 1e-7))^gamma * rg + mn over the whole image, then + noise
__global__ void gamma_noise_kernel(const float* __restrict__ x, float* __restrict__ y, size_t total, float mn, float rg,
                                   float gamma, const float* __restrict__ noise) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    float v = x[i];
    if (gamma > 0.f) v = powf((v - mn) / (rg + 1e-7f), gamma) * rg + mn;
    if (noise) v += noise[i];
    y[i] = v;
  }
}

// GaussianSmooth (MONAI 0.6 GaussianFilter = separable zero-padded correlations, one per spatial axis): one axis of a tensor
// viewed as [outer][L][inner]; the taps (host-computed, erf-integrated, <= 63) sit in LDS
__global__ void __launch_bounds__(256) blur_axis_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t outer, int L,
                                                        size_t inner, const float* __restrict__ taps, int ntaps) {
  __shared__ float tp[64];
  if ((int)threadIdx.x < ntaps) tp[threadIdx.x] = taps[threadIdx.x];
  __syncthreads();
  const int r = ntaps / 2;
  const size_t total = outer * L * inner;
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (size_t)gridDim.x * blockDim.x) {
    const size_t v = it % inner;
    const int i = (int)((it / inner) % L);
    const size_t o = it / (inner * L);
    const float* line = src + o * L * inner + v;
    const int k0 = i - r < 0 ? r - i : 0, k1 = i + r > L - 1 ? L - 1 - i + r : ntaps - 1;
    float acc = 0.f;
    for (int k = k0; k <= k1; ++k) acc += tp[k] * line[(size_t)(i + k - r) * inner];  // tap order = the convolution's summation order
    dst[it] = acc;
  }
}

// CropForeground's bounding box (MONAI generate_spatial_bounding_box, select_fn = x > 0 over ANY channel, margin 0):
// bbox[n] = {z0, y0, x0, z1, y1, x1}, ends exclusive; min / max are order-independent, so integer atomics are exact
__global__ void bbox_init_kernel(int* __restrict__ bbox, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N * 6) bbox[i] = (i % 6) < 3 ? 0x7fffffff : 0;
}
__global__ void __launch_bounds__(256) foreground_bbox_kernel(const float* __restrict__ img, int C, int D, int H, int W,
                                                              int* __restrict__ bbox) {
  const int n = blockIdx.y;
  const size_t vox = (size_t)D * H * W;
  const float* base = img + (size_t)n * C * vox;
  int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {0, 0, 0};
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < vox; v += (size_t)gridDim.x * blockDim.x) {
    bool fg = false;
    for (int c = 0; c < C; ++c) fg |= base[(size_t)c * vox + v] > 0.f;
    if (fg) {
      const int x = (int)(v % W), y = (int)((v / W) % H), z = (int)(v / ((size_t)W * H));
      lo[0] = min(lo[0], z); lo[1] = min(lo[1], y); lo[2] = min(lo[2], x);
      hi[0] = max(hi[0], z + 1); hi[1] = max(hi[1], y + 1); hi[2] = max(hi[2], x + 1);
    }
  }
  __shared__ int slo[3], shi[3];
  if (threadIdx.x < 3) { slo[threadIdx.x] = 0x7fffffff; shi[threadIdx.x] = 0; }
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    if (lo[a] != 0x7fffffff) { atomicMin(&slo[a], lo[a]); atomicMax(&shi[a], hi[a]); }
  }
  __syncthreads();
  if (threadIdx.x < 3 && slo[threadIdx.x] != 0x7fffffff) {
    atomicMin(&bbox[n * 6 + threadIdx.x], slo[threadIdx.x]);
    atomicMax(&bbox[n * 6 + 3 + threadIdx.x], shi[threadIdx.x]);
  }
}

extern "C" int brats_blur_axis(const float* src, float* dst, size_t outer, int L, size_t inner, const float* taps, int ntaps,
                               brats_stream_t s) {
  if (!src || !dst || !taps || src == dst || L <= 0 || ntaps < 1 || ntaps > 63 || !(ntaps & 1))
    BRATS_FAIL(BRATS_E_ARG, "blur_axis: bad argument (odd tap count <= 63, out of place)");
  hipLaunchKernelGGL(blur_axis_kernel, dim3(qgrid(outer * L * inner)), dim3(256), 0, (hipStream_t)s, src, dst, outer, L, inner, taps, ntaps);
  BRATS_CHECK_LAUNCH();
  return 0;
}

extern "C" int brats_foreground_bbox(const float* img, int N, int C, int D, int H, int W, int* bbox, brats_stream_t s) {
  if (!img || !bbox || N <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0) BRATS_FAIL(BRATS_E_ARG, "foreground_bbox: bad argument");
  hipStream_t st = (hipStream_t)s;
  hipLaunchKernelGGL(bbox_init_kernel, dim3((N * 6 + 63) / 64), dim3(64), 0, st, bbox, N);
  size_t gx = ((size_t)D * H * W + 255) / 256 / 4;
  gx = gx < 1 ? 1 : (gx > 1024 ? 1024 : gx);
  hipLaunchKernelGGL(foreground_bbox_kernel, dim3((unsigned)gx, N), dim3(256), 0, st, img, C, D, H, W, bbox);
  BRATS_CHECK_LAUNCH();
  return 0;
}

extern "C" int brats_crop_perm(const float* src, float* dst, int planes, int s0, int s1, int s2, int c0, int c1, int c2, int e0,
                               int e1, int e2, int p0, int p1, int p2, int f0, int f1, int f2, const float* scale,
                               const float* shift, brats_stream_t s) {
  if (!src || !dst || planes <= 0 || (1 << p0 | 1 << p1 | 1 << p2) != 7) BRATS_FAIL(BRATS_E_ARG, "crop_perm: bad argument");
  if (c0 < 0 || c1 < 0 || c2 < 0 || e0 <= 0 || e1 <= 0 || e2 <= 0 || c0 + e0 > s0 || c1 + e1 > s1 || c2 + e2 > s2)
    BRATS_FAIL(BRATS_E_ARG, "crop_perm: crop box [%d,%d,%d]+[%d,%d,%d] outside the %dx%dx%d volume", c0, c1, c2, e0, e1, e2, s0, s1, s2);
  const int e[3] = {e0, e1, e2};
  const int d0 = e[p0], d1 = e[p1], d2 = e[p2];
  hipLaunchKernelGGL(crop_perm_kernel, dim3(qgrid((size_t)planes * d0 * d1 * d2)), dim3(256), 0, (hipStream_t)s, src, dst,
                     (size_t)planes, s0, s1, s2, c0, c1, c2, d0, d1, d2, p0, p1, p2, f0, f1, f2, scale, shift);
  BRATS_CHECK_LAUNCH();
  return 0;
}

extern "C" int brats_label_to_channels(const float* label, float* out, int N, size_t voxels, int order, brats_stream_t s) {
  if (!label || !out || N <= 0 || order < 0 || order > 1) BRATS_FAIL(BRATS_E_ARG, "label_to_channels: bad argument");
  hipLaunchKernelGGL(label_channels_kernel, dim3(qgrid((size_t)N * voxels)), dim3(256), 0, (hipStream_t)s, label, out, N, voxels, order);
  BRATS_CHECK_LAUNCH();
  return 0;
}

extern "C" int brats_zscore_normalize(const float* x, float* y, double* stats, int planes, size_t voxels, int nonzero, float clip,
                                      brats_stream_t s) {
  if (!x || !y || !stats || planes <= 0) BRATS_FAIL(BRATS_E_ARG, "zscore_normalize: bad argument");
  hipStream_t st = (hipStream_t)s;
  hipError_t e = hipMemsetAsync(stats, 0, (size_t)planes * 3 * sizeof(double), st);
  if (e != hipSuccess) BRATS_FAIL(BRATS_E_HIP, "zscore_normalize: memset: %s", hipGetErrorString(e));
  size_t gx = (voxels + 255) / 256 / 8;
  gx = gx < 1 ? 1 : (gx > 512 ? 512 : gx);
  hipLaunchKernelGGL(plane_stats_kernel, dim3((unsigned)gx, planes), dim3(256), 0, st, x, stats, voxels, nonzero);
  hipLaunchKernelGGL(zscore_apply_kernel, dim3((unsigned)gx, planes), dim3(256), 0, st, x, y, (const double*)stats, voxels, nonzero, clip);
  BRATS_CHECK_LAUNCH();
  return 0;
}

extern "C" int brats_gamma_noise(const float* x, float* y, size_t total, float vmin, float vrange, float gamma, const float* noise,
                                 brats_stream_t s) {
  if (!x || !y) BRATS_FAIL(BRATS_E_ARG, "gamma_noise: null pointer");
  hipLaunchKernelGGL(gamma_noise_kernel, dim3(qgrid(total)), dim3(256), 0, (hipStream_t)s, x, y, total, vmin, vrange, gamma, noise);
  BRATS_CHECK_LAUNCH();
  return 0;
}
