// Multi-tensor Ranger2020 step: the reference's learning/optimizer.py:136-255 walks the
// parameters in Python and issues ~15 small torch ops per tensor (RAdam moments, gradient centralisation,
// lookahead), i.e. >1000 launches per step.  Here one step is two launches over ALL tensors:
//   ranger_row_means : per output-channel mean of the gradient (gradient centralisation, :11-20), one WG per row
//   ranger_update    : one WG per 2048-element chunk: GC, moments (:192-196), RAdam / SGD-like update with the
//                      host-computed rectified step size (:198-231), weight decay (:222-223, including the
//                      reference's aliasing of exp_avg in the non-adaptive branch), lookahead (:233-240).
// Pure f32 streaming: 16 B/param read (p, g, m, v) + 12 B written (+8 B on lookahead steps): HBM-bound.
#include "common.hpp"

static constexpr int RANGER_CHUNK = 2048;

__global__ void __launch_bounds__(256) ranger_row_means_kernel(const brats_ranger_tensor* __restrict__ tab,
                                                               const int* __restrict__ rows, float* __restrict__ means) {
  const int t = rows[blockIdx.x * 2], r = rows[blockIdx.x * 2 + 1];
  const brats_ranger_tensor T = tab[t];
  const float* g = (const float*)T.grad + (size_t)r * T.rowlen;
  float s = 0.f;
  for (int i = threadIdx.x; i < T.rowlen; i += 256) s += g[i];
  __shared__ float red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int m = 128; m > 0; m >>= 1) {
    if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
    __syncthreads();
  }
  if (threadIdx.x == 0) means[T.row_base + r] = red[0] / (float)T.rowlen;
}

__global__ void __launch_bounds__(256) ranger_update_kernel(const brats_ranger_tensor* __restrict__ tab,
                                                            const int* __restrict__ chunks, const float* __restrict__ means,
                                                            float beta1, float beta2, float omb1, float omb2, float eps,
                                                            float alpha) {
  const int t = chunks[blockIdx.x * 2];
  const long base = (long)chunks[blockIdx.x * 2 + 1] * RANGER_CHUNK;
  const brats_ranger_tensor T = tab[t];
  float* __restrict__ p = (float*)T.param;
  const float* __restrict__ g = (const float*)T.grad;
  float* __restrict__ m = (float*)T.exp_avg;
  float* __restrict__ v = (float*)T.exp_avg_sq;
  float* __restrict__ slow = (float*)T.slow;
  const bool adaptive = T.flags & 1, look = T.flags & 2, gc = T.rowlen > 0;
  const long end = base + RANGER_CHUNK < T.numel ? base + RANGER_CHUNK : T.numel;
  for (long i = base + threadIdx.x; i < end; i += 256) {
    float gi = g[i];
    if (gc) gi = gi + (-means[T.row_base + (int)(i / T.rowlen)]);
    const float vi = v[i] * beta2 + (omb2 * gi) * gi;
    float mi = m[i] * beta1 + omb1 * gi;
    float pi = p[i];
    float G = adaptive ? mi / (sqrtf(vi) + eps) : mi;
    if (T.wd != 0.f) {
      G = G + T.wd * pi;
      if (!adaptive) mi = G;  // the reference's G_grad aliases exp_avg here (learning/optimizer.py:220-223)
    }
    pi = pi + T.neg_step * G;
    if (look) {
      const float si = slow[i] + alpha * (pi - slow[i]);
      slow[i] = si;
      pi = si;
    }
    v[i] = vi;
    m[i] = mi;
    p[i] = pi;
  }
}

extern "C" int brats_ranger_chunk(void) { return RANGER_CHUNK; }

extern "C" int brats_ranger_step(const brats_ranger_tensor* table, int ntensors, const int* chunks, int nchunks, const int* rows,
                                 int nrows, float* row_means, float beta1, float beta2, float one_minus_beta1,
                                 float one_minus_beta2, float eps, float alpha, brats_stream_t s) {
  if (!table || ntensors <= 0 || !chunks || nchunks <= 0) BRATS_FAIL(BRATS_E_ARG, "ranger_step: empty tensor / chunk table");
  if (nrows > 0 && (!rows || !row_means)) BRATS_FAIL(BRATS_E_ARG, "ranger_step: gradient centralisation needs rows + row_means");
  hipStream_t st = (hipStream_t)s;
  if (nrows > 0) {
    hipLaunchKernelGGL(ranger_row_means_kernel, dim3(nrows), dim3(256), 0, st, table, rows, row_means);
    BRATS_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(ranger_update_kernel, dim3(nchunks), dim3(256), 0, st, table, chunks, row_means, beta1, beta2,
                     one_minus_beta1, one_minus_beta2, eps, alpha);
  BRATS_CHECK_LAUNCH();
  return 0;
}
