// Fused sigmoid-Dice / Jaccard statistics and gradient for deep supervision.
// Semantics = monai.losses.DiceLoss(sigmoid, squared_pred, batch=True) as configured at src/definer.py:184-203,
// averaged over heads as in learning/engine.py:322-330.  The scalar algebra on the [heads][classes] sums stays
// in torch; these kernels only do the two HBM-bound passes over the [N][K][V] f32 logits:
//   stats : sums[k] = { sum t*p, sum p*p, sum t*t }      (p = sigmoid(x), over n and voxels)
//   grad  : dx = (a_k * t + b_k * 2p) * p * (1 - p)        (a_k = dL/dI_k, b_k = dL/dP2_k, read from device memory)
#include "common.hpp"

__global__ void dice_stats_kernel(const float* __restrict__ x, const float* __restrict__ t, float* __restrict__ sums, int K,
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
  if (threadIdx.x < 3) atomicAdd(sums + k * 3 + threadIdx.x, r[threadIdx.x][0]);
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

extern "C" int brats_dice_stats(const float* logits, const float* target, float* sums /*[K][3], zeroed here*/, int N, int K,
                                size_t voxels, brats_stream_t s) {
  if (!logits || !target || !sums || N <= 0 || K <= 0) BRATS_FAIL(BRATS_E_ARG, "dice_stats: bad argument");
  hipStream_t st = (hipStream_t)s;
  hipError_t e = hipMemsetAsync(sums, 0, (size_t)K * 3 * sizeof(float), st);
  if (e != hipSuccess) BRATS_FAIL(BRATS_E_HIP, "dice_stats: memset: %s", hipGetErrorString(e));
  size_t gx = (voxels / 4 + 255) / 256 / 4;
  gx = gx < 1 ? 1 : (gx > 512 ? 512 : gx);
  hipLaunchKernelGGL(dice_stats_kernel, dim3((unsigned)gx, N * K), dim3(256), 0, st, logits, target, sums, K, voxels);
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
