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

// torch.amp.GradScaler without its host round trip (the optimizer's _step_supports_amp_scaling protocol, as torch's fused Adam):
// grad_scale = the loss scale the gradients still carry (device scalar, a power of two; NULL = already unscaled), found_inf =
// non-zero when a gradient overflowed (device scalar; NULL = not checked) -- then the whole step, counter included, is a no-op
struct RangerAmp {
  const float* grad_scale;
  const float* found_inf;
  __device__ bool skip() const { return found_inf && *found_inf != 0.f; }
  __device__ float inv() const { return grad_scale ? 1.f / *grad_scale : 1.f; }
};

__global__ void __launch_bounds__(256) ranger_row_means_kernel(const brats_ranger_tensor* __restrict__ tab,
                                                               const int* __restrict__ rows, float* __restrict__ means, RangerAmp amp) {
  if (amp.skip()) return;
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
  if (threadIdx.x == 0) means[T.row_base + r] = red[0] / (float)T.rowlen;  // (of the gradients as stored: still scaled under amp)
}

// use_gcnorm (learning/optimizer.py:23-36,189-190): the (centralised) gradient of every tensor with more than two elements
// is divided by its unbiased standard deviation + 1e-8.  Two small kernels in front of the update: per 2048-element chunk
// the sums of x = g - row mean - pilot and of x^2 (fixed-order tree), then per tensor the chunk sums added in chunk order in
// f64.  pilot = the tensor's first (centralised) element: the variance is shift-invariant, and with the shift the one-pass
// formula (s2 - s1^2 / n) no longer cancels catastrophically when |mean| >> std (use_gc = False: the reference's torch.std
// is two-pass); the f32 chunk partials stay well-conditioned.
__global__ void __launch_bounds__(256) ranger_chunk_stats_kernel(const brats_ranger_tensor* __restrict__ tab,
                                                                 const int* __restrict__ chunks, const float* __restrict__ means,
                                                                 float* __restrict__ part /* [nchunks][2] */, RangerAmp amp) {
  if (amp.skip()) return;
  const int t = chunks[blockIdx.x * 2];
  const long base = (long)chunks[blockIdx.x * 2 + 1] * RANGER_CHUNK;
  const brats_ranger_tensor T = tab[t];
  const float* __restrict__ g = (const float*)T.grad;
  const bool gc = T.rowlen > 0;
  const long end = base + RANGER_CHUNK < T.numel ? base + RANGER_CHUNK : T.numel;
  float pilot = g[0];
  if (gc) pilot = pilot + (-means[T.row_base]);
  float s1 = 0.f, s2 = 0.f;
  for (long i = base + threadIdx.x; i < end; i += 256) {
    float x = g[i];
    if (gc) x = x + (-means[T.row_base + (int)(i / T.rowlen)]);
    x = x - pilot;
    s1 += x;
    s2 += x * x;
  }
  __shared__ float r1[256], r2[256];
  r1[threadIdx.x] = s1;
  r2[threadIdx.x] = s2;
  __syncthreads();
  for (int m = 128; m > 0; m >>= 1) {
    if ((int)threadIdx.x < m) { r1[threadIdx.x] += r1[threadIdx.x + m]; r2[threadIdx.x] += r2[threadIdx.x + m]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[blockIdx.x * 2] = r1[0]; part[blockIdx.x * 2 + 1] = r2[0]; }
}

__global__ void ranger_tensor_std_kernel(const brats_ranger_tensor* __restrict__ tab, const float* __restrict__ part,
                                         float* __restrict__ gstd /* [ntensors] */, RangerAmp amp) {
  if (amp.skip()) return;
  const int t = blockIdx.x;
  const brats_ranger_tensor T = tab[t];
  const long n = T.numel;
  const int nch = (int)((n + RANGER_CHUNK - 1) / RANGER_CHUNK);
  double s1 = 0.0, s2 = 0.0;
  for (int c = 0; c < nch; ++c) { s1 += part[(size_t)(T.chunk_base + c) * 2]; s2 += part[(size_t)(T.chunk_base + c) * 2 + 1]; }
  float d = 1.f;  // tensors of one or two elements are left alone (:33)
  if (n > 2) {
    double var = (s2 - s1 * s1 / (double)n) / (double)(n - 1);
    d = (float)sqrt(var > 0.0 ? var : 0.0) * amp.inv() + 1e-8f;  // (the std of the UNSCALED gradient: the scale is a power of two)
  }
  gstd[t] = d;
}

__global__ void __launch_bounds__(256) ranger_update_kernel(const brats_ranger_tensor* __restrict__ tab,
                                                            const int* __restrict__ chunks, const float* __restrict__ means,
                                                            const float* __restrict__ gstd,
                                                            const brats_ranger_dyn* __restrict__ dyn, float beta1, float beta2,
                                                            float omb1, float omb2, float eps, float alpha, RangerAmp amp) {
  if (amp.skip()) return;
  const float ginv = amp.inv();
  const int t = chunks[blockIdx.x * 2];
  const long base = (long)chunks[blockIdx.x * 2 + 1] * RANGER_CHUNK;
  const brats_ranger_tensor T = tab[t];
  float* __restrict__ p = (float*)T.param;
  const float* __restrict__ g = (const float*)T.grad;
  float* __restrict__ m = (float*)T.exp_avg;
  float* __restrict__ v = (float*)T.exp_avg_sq;
  float* __restrict__ slow = (float*)T.slow;
  // graph-captured steps read the step-dependent scalars from device memory (written by ranger_advance_kernel)
  const int flags = dyn ? dyn->flags : T.flags;
  const float neg_step = dyn ? dyn->neg_step : T.neg_step;
  const bool adaptive = flags & 1, look = flags & 2, gc = T.rowlen > 0;
  const long end = base + RANGER_CHUNK < T.numel ? base + RANGER_CHUNK : T.numel;
  const float gdiv = gstd ? gstd[t] : 1.f;
  for (long i = base + threadIdx.x; i < end; i += 256) {
    float gi = g[i];
    if (gc) gi = gi + (-means[T.row_base + (int)(i / T.rowlen)]);
    gi = gi * ginv;  // (exact: a power of two; (g - mean(g)) / s == g / s - mean(g / s) bit for bit, short of under / overflow)
    if (gstd) gi = gi / gdiv;
    const float vi = v[i] * beta2 + (omb2 * gi) * gi;
    float mi = m[i] * beta1 + omb1 * gi;
    float pi = p[i];
    float G = adaptive ? mi / (sqrtf(vi) + eps) : mi;
    if (T.wd != 0.f) {
      G = G + T.wd * pi;
      if (!adaptive) mi = G;  // the reference's G_grad aliases exp_avg here (learning/optimizer.py:220-223)
    }
    pi = pi + neg_step * G;
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

// step += 1; neg_step / flags of learning/optimizer.py:198-214 in f64 on the device (one thread), so that a captured
// hipGraph can be replayed without any host-side change between steps
__global__ void ranger_advance_kernel(brats_ranger_dyn* dyn, double lr_host, double beta1, double beta2, int k, double nsma_thr,
                                      RangerAmp amp) {
  if (amp.skip()) return;  // an overflowed step does not count (GradScaler skips optimizer.step() altogether)
  const double lr = lr_host >= 0.0 ? lr_host : dyn->lr;  // lr < 0: the learning rate lives on the device (LR schedules under replay)
  const int step = dyn->step + 1;
  const double b2t = pow(beta2, (double)step);
  const double nmax = 2.0 / (1.0 - beta2) - 1.0;
  const double nsma = nmax - 2.0 * step * b2t / (1.0 - b2t);
  const double b1c = 1.0 - pow(beta1, (double)step);
  double ss;
  int flags = 0;
  if (nsma > nsma_thr) {
    ss = sqrt((1.0 - b2t) * (nsma - 4.0) / (nmax - 4.0) * (nsma - 2.0) / nsma * nmax / (nmax - 2.0)) / b1c;
    flags |= 1;
  } else {
    ss = 1.0 / b1c;
  }
  if (step % k == 0) flags |= 2;
  dyn->step = step;
  dyn->flags = flags;
  dyn->neg_step = (float)(-ss * lr);
}

extern "C" int brats_ranger_chunk(void) { return RANGER_CHUNK; }

extern "C" int brats_ranger_advance_amp(brats_ranger_dyn* dyn, double lr, double beta1, double beta2, int k, double nsma_threshold,
                                        const float* found_inf, brats_stream_t s) {
  if (!dyn || k < 1) BRATS_FAIL(BRATS_E_ARG, "ranger_advance: bad argument");
  // (lr < 0 = read dyn->lr: a captured hipGraph then follows a learning-rate schedule without being re-captured)
  hipLaunchKernelGGL(ranger_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)s, dyn, lr, beta1, beta2, k, nsma_threshold,
                     RangerAmp{nullptr, found_inf});
  BRATS_CHECK_LAUNCH();
  return 0;
}
extern "C" int brats_ranger_advance(brats_ranger_dyn* dyn, double lr, double beta1, double beta2, int k, double nsma_threshold,
                                    brats_stream_t s) {
  return brats_ranger_advance_amp(dyn, lr, beta1, beta2, k, nsma_threshold, nullptr, s);
}

extern "C" int brats_ranger_step_amp(const brats_ranger_tensor* table, int ntensors, const int* chunks, int nchunks, const int* rows,
                                     int nrows, float* row_means, float* chunk_stats, float* grad_std, const brats_ranger_dyn* dyn,
                                     float beta1, float beta2, float one_minus_beta1, float one_minus_beta2, float eps, float alpha,
                                     const float* grad_scale, const float* found_inf, brats_stream_t s) {
  const RangerAmp amp{grad_scale, found_inf};
  if (!table || ntensors <= 0 || !chunks || nchunks <= 0) BRATS_FAIL(BRATS_E_ARG, "ranger_step: empty tensor / chunk table");
  if (nrows > 0 && (!rows || !row_means)) BRATS_FAIL(BRATS_E_ARG, "ranger_step: gradient centralisation needs rows + row_means");
  hipStream_t st = (hipStream_t)s;
  if (nrows > 0) {
    hipLaunchKernelGGL(ranger_row_means_kernel, dim3(nrows), dim3(256), 0, st, table, rows, row_means, amp);
    BRATS_CHECK_LAUNCH();
  }
  if ((chunk_stats == nullptr) != (grad_std == nullptr)) BRATS_FAIL(BRATS_E_ARG, "ranger_step: use_gcnorm needs chunk_stats AND grad_std");
  if (grad_std) {
    hipLaunchKernelGGL(ranger_chunk_stats_kernel, dim3(nchunks), dim3(256), 0, st, table, chunks, row_means, chunk_stats, amp);
    hipLaunchKernelGGL(ranger_tensor_std_kernel, dim3(ntensors), dim3(1), 0, st, table, (const float*)chunk_stats, grad_std, amp);
    BRATS_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(ranger_update_kernel, dim3(nchunks), dim3(256), 0, st, table, chunks, row_means, (const float*)grad_std, dyn,
                     beta1, beta2, one_minus_beta1, one_minus_beta2, eps, alpha, amp);
  BRATS_CHECK_LAUNCH();
  return 0;
}
extern "C" int brats_ranger_step(const brats_ranger_tensor* table, int ntensors, const int* chunks, int nchunks, const int* rows,
                                 int nrows, float* row_means, float* chunk_stats, float* grad_std, const brats_ranger_dyn* dyn,
                                 float beta1, float beta2, float one_minus_beta1, float one_minus_beta2, float eps, float alpha,
                                 brats_stream_t s) {
  return brats_ranger_step_amp(table, ntensors, chunks, nchunks, rows, nrows, row_means, chunk_stats, grad_std, dyn, beta1, beta2,
                               one_minus_beta1, one_minus_beta2, eps, alpha, nullptr, nullptr, s);
}
