// Fused sigmoid-Dice / Jaccard statistics and gradient for deep supervision.
// Semantics = monai.losses.DiceLoss(sigmoid, squared_pred, batch=True) as configured at src/definer.py:184-203,
// averaged over heads as in learning/engine.py:322-330.  The scalar algebra on the [heads][classes] sums stays
// in torch; these kernels only do the two HBM-bound passes over the [N][K][V] f32 logits:
//   stats : sums[k] = { sum t*p, sum p*p, sum t*t }      (p = sigmoid(x), over n and voxels)
//   grad  : dx = (a_k * t + b_k * 2p) * p * (1 - p)        (a_k = dL/dI_k, b_k = dL/dP2_k, read from device memory)
#include "common.hpp"

__global__ void dice_stats_kernel(const float* __restrict__ x, const float* __restrict__ t, float* __restrict__ part, int K,
                                  size_t voxels) {
  // grid: (blocks, N*K); each block reduces a slice of one (n, k) plane
  const int nk = blockIdx.y, k = nk % K;
  const float* xp = x + (size_t)nk * voxels;
  const float* tp = t + (size_t)nk * voxels;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  const size_t v4 = voxels / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < v4; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 xv = ((const f32x4*)xp)[i], tv = ((const f32x4*)tp)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float p = 1.f / (1.f + __expf(-xv[j]));
      s0 += tv[j] * p;
      s1 += p * p;
      s2 += tv[j] * tv[j];
    }
  }
  if (blockIdx.x == 0) {
    for (size_t i = v4 * 4 + threadIdx.x; i < voxels; i += blockDim.x) {
      const float p = 1.f / (1.f + __expf(-xp[i]));
      s0 += tp[i] * p; s1 += p * p; s2 += tp[i] * tp[i];
    }
  }
  __shared__ float r[3][256];
  r[0][threadIdx.x] = s0; r[1][threadIdx.x] = s1; r[2][threadIdx.x] = s2;
  __syncthreads();
  for (int m = 128; m > 0; m >>= 1) {
    if ((int)threadIdx.x < m) {
      r[0][threadIdx.x] += r[0][threadIdx.x + m];
      r[1][threadIdx.x] += r[1][threadIdx.x + m];
      r[2][threadIdx.x] += r[2][threadIdx.x + m];
    }
    __syncthreads();
  }
  // one partial per block: part[(n * gridDim.x + blockIdx.x)][K][3]; brats_ordered_sum adds them in block order
  if (threadIdx.x < 3) part[(((size_t)(nk / K) * gridDim.x + blockIdx.x) * K + k) * 3 + threadIdx.x] = r[threadIdx.x][0];
}

// out[i] = sum over nb partial vectors, 8 entries x 32 slices per block, slices added in order
__global__ void __launch_bounds__(256) ordered_sum_kernel(const float* __restrict__ part, float* __restrict__ out, int nb, int total,
                                                          float* __restrict__ out2 /* entries >= n1 go here (may be NULL) */, int n1) {
  const int e = threadIdx.x & 7, g = threadIdx.x >> 3;
  const int i = blockIdx.x * 8 + e;
  float s = 0.f;
  if (i < total) {
    // eight partials in flight, added in block order (hipcc turns the plain loop into load - wait - add, one round trip each)
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
    if (out2 && i >= n1) out2[i - n1] = s;
    else out[i] = s;
  }
}

int brats_ordered_sum(const float* part, float* out, int nb, int total, hipStream_t st) {
  hipLaunchKernelGGL(ordered_sum_kernel, dim3((total + 7) / 8), dim3(256), 0, st, part, out, nb, total, (float*)nullptr, 0);
  return 0;
}
// the first n1 totals to out1, the rest to out2 (two result tensors, one launch, no copies)
int brats_ordered_sum2(const float* part, float* out1, int n1, float* out2, int nb, int total, hipStream_t st) {
  hipLaunchKernelGGL(ordered_sum_kernel, dim3((total + 7) / 8), dim3(256), 0, st, part, out1, nb, total, out2, n1);
  return 0;
}

__global__ void dice_grad_kernel(const float* __restrict__ x, const float* __restrict__ t, const float* __restrict__ coef,
                                 float* __restrict__ dx, int K, size_t voxels) {
  const int nk = blockIdx.y, k = nk % K;
  const float a = coef[k * 2], b2 = 2.f * coef[k * 2 + 1];
  const float* xp = x + (size_t)nk * voxels;
  const float* tp = t + (size_t)nk * voxels;
  float* dp = dx + (size_t)nk * voxels;
  const size_t v4 = voxels / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < v4; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 xv = ((const f32x4*)xp)[i], tv = ((const f32x4*)tp)[i];
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float p = 1.f / (1.f + __expf(-xv[j]));
      o[j] = (a * tv[j] + b2 * p) * p * (1.f - p);
    }
    ((f32x4*)dp)[i] = o;
  }
  if (blockIdx.x == 0) {
    for (size_t i = v4 * 4 + threadIdx.x; i < voxels; i += blockDim.x) {
      const float p = 1.f / (1.f + __expf(-xp[i]));
      dp[i] = (a * tp[i] + b2 * p) * p * (1.f - p);
    }
  }
}

constexpr int DICE_MAX_BLOCKS = 512;
extern "C" size_t brats_dice_ws_floats(int N, int K) { return (size_t)N * DICE_MAX_BLOCKS * K * 3; }

extern "C" int brats_dice_stats(const float* logits, const float* target, float* sums /*[K][3]*/, float* ws, int N, int K,
                                size_t voxels, brats_stream_t s) {
  if (!logits || !target || !sums || !ws || N <= 0 || K <= 0) BRATS_FAIL(BRATS_E_ARG, "dice_stats: bad argument");
  hipStream_t st = (hipStream_t)s;
  size_t gx = (voxels / 4 + 255) / 256 / 4;
  gx = gx < 1 ? 1 : (gx > DICE_MAX_BLOCKS ? DICE_MAX_BLOCKS : gx);
  hipLaunchKernelGGL(dice_stats_kernel, dim3((unsigned)gx, N * K), dim3(256), 0, st, logits, target, ws, K, voxels);
  brats_ordered_sum(ws, sums, N * (int)gx, K * 3, st);
  BRATS_CHECK_LAUNCH();
  return 0;
}

extern "C" int brats_dice_grad(const float* logits, const float* target, const float* coef /*[K][2] = dL/dI, dL/dP2*/,
                               float* dlogits, int N, int K, size_t voxels, brats_stream_t s) {
  if (!logits || !target || !coef || !dlogits) BRATS_FAIL(BRATS_E_ARG, "dice_grad: null pointer");
  size_t gx = (voxels / 4 + 255) / 256 / 4;
  gx = gx < 1 ? 1 : (gx > 2048 ? 2048 : gx);
  hipLaunchKernelGGL(dice_grad_kernel, dim3((unsigned)gx, N * K), dim3(256), 0, (hipStream_t)s, logits, target, coef, dlogits, K, voxels);
  BRATS_CHECK_LAUNCH();
  return 0;
}
