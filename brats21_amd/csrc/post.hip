// Post-forward chain of Engine.evaluate kept on the GPU: the reference pads the volume to
// a multiple of 8 (utils/transforms.py:482-512), averages the sigmoid outputs of models x TTA passes on the CPU,
// thresholds (src/definer.py:700-703 AsDiscrete), removes background voxels (utils/transforms.py:536-550),
// converts TC/WT/ET channels to BraTS labels (utils/transforms.py:169-206), crops back (:515-533) and computes the
// hard Dice of utils/metrics.py:35-67.  All NCDHW f32; pure HBM-bound index / reduction kernels, x-fastest.
#include "common.hpp"

static inline int pgrid(size_t total) {
  size_t b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

// dst[p][z][y][x] = src[p][z-oz][y-oy][x-ox] inside the source box, `fill` outside.  Positive offsets pad
// (shape_to_divisible: o = p_b), negative offsets crop (shape_to_original: o = -p_b).
__global__ void pad_crop_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t planes, int sd, int sh, int sw,
                                int dd, int dh, int dw, int oz, int oy, int ox, float fill) {
  const size_t total = planes * dd * dh * dw;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t t = i;
    const int x = (int)(t % dw) - ox; t /= dw;
    const int y = (int)(t % dh) - oy; t /= dh;
    const int z = (int)(t % dd) - oz;
    const size_t p = t / dd;
    float v = fill;
    if (z >= 0 && z < sd && y >= 0 && y < sh && x >= 0 && x < sw) v = src[((p * sd + z) * sh + y) * sw + x];
    dst[i] = v;
  }
}

// seg[n][k][v] = (prob[n][k][v] * scale >= thresh) && any_c(img[n][c][v] != 0)       (f32 0/1, like the reference)
// labels[n][v] (optional, K == 3, channel order TC/WT/ET): ET -> 4, TC&!ET -> 1, WT&!TC -> 2 (written in the
// reference's assignment order: et, then net, then ed).
__global__ void post_threshold_kernel(const float* __restrict__ prob, const float* __restrict__ img, float* __restrict__ seg,
                                      uint8_t* __restrict__ labels, int K, int C, size_t voxels, float scale, float thresh) {
  const int n = blockIdx.y;
  const float* pp = prob + (size_t)n * K * voxels;
  const float* ip = img ? img + (size_t)n * C * voxels : nullptr;
  float* sp = seg + (size_t)n * K * voxels;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < voxels; v += (size_t)gridDim.x * blockDim.x) {
    bool brain = true;
    if (ip) {
      brain = false;
      for (int c = 0; c < C; ++c) brain |= ip[(size_t)c * voxels + v] != 0.f;
    }
    bool ch[3] = {false, false, false};
    for (int k = 0; k < K; ++k) {
      const bool on = brain && (pp[(size_t)k * voxels + v] * scale >= thresh);
      sp[(size_t)k * voxels + v] = on ? 1.f : 0.f;
      if (k < 3) ch[k] = on;
    }
    if (labels) {
      uint8_t l = 0;
      if (ch[2]) l = 4;
      if (ch[0] && !ch[2]) l = 1;
      if (ch[1] && !ch[0]) l = 2;
      labels[(size_t)n * voxels + v] = l;
    }
  }
}

// counts[nk] = { #(p && t), #p, #t } over one (n, k) plane, p = pred != 0, t = target != 0 (exact integers)
__global__ void overlap_counts_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                      unsigned long long* __restrict__ counts, size_t voxels) {
  const int nk = blockIdx.y;
  const float* pp = pred + (size_t)nk * voxels;
  const float* tp = target + (size_t)nk * voxels;
  unsigned c0 = 0, c1 = 0, c2 = 0;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < voxels; v += (size_t)gridDim.x * blockDim.x) {
    const bool p = pp[v] != 0.f, t = tp[v] != 0.f;
    c0 += p && t; c1 += p; c2 += t;
  }
  __shared__ unsigned r[3][256];
  r[0][threadIdx.x] = c0; r[1][threadIdx.x] = c1; r[2][threadIdx.x] = c2;
  __syncthreads();
  for (int m = 128; m > 0; m >>= 1) {
    if ((int)threadIdx.x < m) {
      r[0][threadIdx.x] += r[0][threadIdx.x + m];
      r[1][threadIdx.x] += r[1][threadIdx.x + m];
      r[2][threadIdx.x] += r[2][threadIdx.x + m];
    }
    __syncthreads();
  }
  if (threadIdx.x < 3) atomicAdd(counts + (size_t)nk * 3 + threadIdx.x, (unsigned long long)r[threadIdx.x][0]);
}

extern "C" int brats_pad_crop(const float* src, float* dst, int planes, int sd, int sh, int sw, int dd, int dh, int dw, int oz,
                              int oy, int ox, float fill, brats_stream_t s) {
  if (!src || !dst || planes <= 0 || sd <= 0 || sh <= 0 || sw <= 0 || dd <= 0 || dh <= 0 || dw <= 0)
    BRATS_FAIL(BRATS_E_ARG, "pad_crop: bad argument");
  hipLaunchKernelGGL(pad_crop_kernel, dim3(pgrid((size_t)planes * dd * dh * dw)), dim3(256), 0, (hipStream_t)s, src, dst,
                     (size_t)planes, sd, sh, sw, dd, dh, dw, oz, oy, ox, fill);
  BRATS_CHECK_LAUNCH();
  return 0;
}

extern "C" int brats_post_threshold(const float* prob, const float* img, float* seg, uint8_t* labels, int N, int K, int C,
                                    size_t voxels, float scale, float thresh, brats_stream_t s) {
  if (!prob || !seg || N <= 0 || K <= 0 || (img && C <= 0)) BRATS_FAIL(BRATS_E_ARG, "post_threshold: bad argument");
  if (labels && K != 3) BRATS_FAIL(BRATS_E_ARG, "post_threshold: BraTS labels need the 3 channels TC/WT/ET, got K=%d", K);
  size_t gx = (voxels + 255) / 256;
  gx = gx < 1 ? 1 : (gx > 4096 ? 4096 : gx);
  hipLaunchKernelGGL(post_threshold_kernel, dim3((unsigned)gx, N), dim3(256), 0, (hipStream_t)s, prob, img, seg, labels, K, C,
                     voxels, scale, thresh);
  BRATS_CHECK_LAUNCH();
  return 0;
}

extern "C" int brats_overlap_counts(const float* pred, const float* target, unsigned long long* counts, int NK, size_t voxels,
                                    brats_stream_t s) {
  if (!pred || !target || !counts || NK <= 0) BRATS_FAIL(BRATS_E_ARG, "overlap_counts: bad argument");
  hipStream_t st = (hipStream_t)s;
  hipError_t e = hipMemsetAsync(counts, 0, (size_t)NK * 3 * sizeof(unsigned long long), st);
  if (e != hipSuccess) BRATS_FAIL(BRATS_E_HIP, "overlap_counts: memset: %s", hipGetErrorString(e));
  size_t gx = (voxels + 255) / 256 / 8;
  gx = gx < 1 ? 1 : (gx > 512 ? 512 : gx);
  hipLaunchKernelGGL(overlap_counts_kernel, dim3((unsigned)gx, NK), dim3(256), 0, st, pred, target, counts, voxels);
  BRATS_CHECK_LAUNCH();
  return 0;
}
