// Squeeze-and-excitation gate of MONAI's ResidualSELayer as the reference builds it (networks/equiunet2021.py:204-205:
// ResidualSELayer(3, C, r=2, relu, sigmoid)): for sample n
//     gap = mean_v z[n][v][:]                      (the channel sums come out of the EvoNorm pass, norm.hip)
//     hid = relu(W1 gap + b1),  W1: [C/2][C]
//     gate = sigmoid(W2 hid + b2), W2: [C][C/2]
//     out = z + z * gate = z * (1 + gate)          (applied by brats_channel_scale / folded into the EvoNorm backward)
// and its backward incl. the four parameter gradients.  These were ~20 tiny ATen launches per block (F.linear,
// torch.autograd.grad) x 7 blocks in a launch-bound step; here: ONE launch forward (a workgroup per sample) and ONE launch
// backward (a single workgroup: <= 4 x 74 k multiply-adds per sample at C = 384).  HBM-/latency-bound by construction, all
// reductions in a fixed order (wave-per-row dot products, shuffles in a fixed tree, samples added in order) -> bitwise
// reproducible.
#include "common.hpp"

DEVI float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}

// y[r] = sum_k M[r][k] * x[k] for r in [0, R): one wave per row (lanes stride over k), rows dealt round-robin to the waves
template <typename F>
DEVI void rows_dot(const float* __restrict__ M, int R, int K, const float* x /*LDS*/, int wave, int nwaves, int lane, F&& put) {
  for (int r = wave; r < R; r += nwaves) {
    const float* row = M + (size_t)r * K;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += row[k] * x[k];
    s = wave_sum(s);
    if (lane == 0) put(r, s);
  }
}

// grid = N, block = 512.  LDS: gap[C] + hid[Ch]
__global__ __launch_bounds__(512) void se_fwd_kernel(const float* __restrict__ chansum, float inv_vox, const float* __restrict__ w1,
                                                     const float* __restrict__ b1, const float* __restrict__ w2,
                                                     const float* __restrict__ b2, float* __restrict__ gate1p,
                                                     float* __restrict__ hidden, int C, int Ch) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  float* gap = (float*)lds_raw;
  float* hid = gap + C;
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  for (int c = tid; c < C; c += blockDim.x) gap[c] = chansum[(size_t)n * C + c] * inv_vox;
  __syncthreads();
  rows_dot(w1, Ch, C, gap, wave, nw, lane, [&](int j, float s) {
    const float h = fmaxf(s + b1[j], 0.f);
    hid[j] = h;
    hidden[(size_t)n * Ch + j] = h;
  });
  __syncthreads();
  rows_dot(w2, C, Ch, hid, wave, nw, lane, [&](int c, float s) {
    gate1p[(size_t)n * C + c] = 1.f + 1.f / (1.f + __expf(-(s + b2[c])));
  });
}

// grid = 1, block = 1024.  LDS: ds[N][C], dh[N][Ch], gap[N][C], hid[N][Ch]
__global__ __launch_bounds__(1024) void se_bwd_kernel(const float* __restrict__ dgate, const float* __restrict__ chansum, float inv_vox,
                                                      const float* __restrict__ hidden, const float* __restrict__ gate1p,
                                                      const float* __restrict__ w1, const float* __restrict__ w2,
                                                      float* __restrict__ gadd, float* __restrict__ dw1, float* __restrict__ db1,
                                                      float* __restrict__ dw2, float* __restrict__ db2, int N, int C, int Ch) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  float* ds = (float*)lds_raw;          // [N][C]   d loss / d (pre-sigmoid)
  float* dh = ds + (size_t)N * C;       // [N][Ch]  d loss / d (pre-relu)
  float* gap = dh + (size_t)N * Ch;     // [N][C]
  float* hid = gap + (size_t)N * C;     // [N][Ch]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6, nt = blockDim.x;
  for (int i = tid; i < N * C; i += nt) {
    const float g = gate1p[i] - 1.f;
    ds[i] = dgate[i] * g * (1.f - g);
    gap[i] = chansum[i] * inv_vox;
  }
  for (int i = tid; i < N * Ch; i += nt) hid[i] = hidden[i];
  __syncthreads();
  // dh[n][j] = [hid > 0] * sum_c W2[c][j] ds[n][c]: a thread owns column j (reads of W2 coalesce over j)
  for (int j = tid; j < Ch; j += nt) {
    float acc[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) acc[n] = 0.f;
    for (int c = 0; c < C; ++c) {
      const float w = w2[(size_t)c * Ch + j];
#pragma unroll
      for (int n = 0; n < 8; ++n)
        if (n < N) acc[n] += w * ds[n * C + c];
    }
#pragma unroll
    for (int n = 0; n < 8; ++n)
      if (n < N) dh[n * Ch + j] = hid[n * Ch + j] > 0.f ? acc[n] : 0.f;
  }
  __syncthreads();
  // dgap[n][c] = sum_j W1[j][c] dh[n][j] (a thread owns column c), handed on as gadd = dgap / V
  for (int c = tid; c < C; c += nt) {
    float acc[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) acc[n] = 0.f;
    for (int j = 0; j < Ch; ++j) {
      const float w = w1[(size_t)j * C + c];
#pragma unroll
      for (int n = 0; n < 8; ++n)
        if (n < N) acc[n] += w * dh[n * Ch + j];
    }
#pragma unroll
    for (int n = 0; n < 8; ++n)
      if (n < N) gadd[(size_t)n * C + c] = acc[n] * inv_vox;
  }
  // parameter gradients: samples added in order
  for (int i = tid; i < C * Ch; i += nt) {
    const int c = i / Ch, j = i % Ch;   // dW2[c][j] = sum_n ds[n][c] hid[n][j]
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += ds[n * C + c] * hid[n * Ch + j];
    dw2[i] = s;
  }
  for (int i = tid; i < Ch * C; i += nt) {
    const int j = i / C, c = i % C;     // dW1[j][c] = sum_n dh[n][j] gap[n][c]
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += dh[n * Ch + j] * gap[n * C + c];
    dw1[i] = s;
  }
  for (int c = tid; c < C; c += nt) {
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += ds[n * C + c];
    db2[c] = s;
  }
  for (int j = tid; j < Ch; j += nt) {
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += dh[n * Ch + j];
    db1[j] = s;
  }
}

extern "C" int brats_se_fwd(const float* chansum, float inv_vox, const float* w1, const float* b1, const float* w2, const float* b2,
                            float* gate1p, float* hidden, int N, int C, int Ch, brats_stream_t s) {
  if (!chansum || !w1 || !b1 || !w2 || !b2 || !gate1p || !hidden || N <= 0 || C <= 0 || Ch <= 0)
    BRATS_FAIL(BRATS_E_ARG, "se_fwd: null pointer or non-positive size");
  const size_t lds = (size_t)(C + Ch) * 4;
  if (lds > 64 * 1024) BRATS_FAIL(BRATS_E_UNSUPPORTED, "se_fwd: C = %d, C/r = %d exceed the 64 KB LDS budget", C, Ch);
  hipLaunchKernelGGL(se_fwd_kernel, dim3(N), dim3(512), lds, (hipStream_t)s, chansum, inv_vox, w1, b1, w2, b2, gate1p, hidden, C, Ch);
  BRATS_CHECK_LAUNCH();
  return 0;
}

extern "C" int brats_se_bwd(const float* dgate, const float* chansum, float inv_vox, const float* hidden, const float* gate1p,
                            const float* w1, const float* w2, float* gadd, float* dw1, float* db1, float* dw2, float* db2,
                            int N, int C, int Ch, brats_stream_t s) {
  if (!dgate || !chansum || !hidden || !gate1p || !w1 || !w2 || !gadd || !dw1 || !db1 || !dw2 || !db2 || N <= 0 || C <= 0 || Ch <= 0)
    BRATS_FAIL(BRATS_E_ARG, "se_bwd: null pointer or non-positive size");
  const size_t lds = (size_t)N * (2 * C + 2 * Ch) * 4;
  if (N > 8 || lds > 64 * 1024)
    BRATS_FAIL(BRATS_E_UNSUPPORTED, "se_bwd: N = %d (max 8) x (C = %d, C/r = %d) exceeds the 64 KB LDS budget", N, C, Ch);
  hipLaunchKernelGGL(se_bwd_kernel, dim3(1), dim3(1024), lds, (hipStream_t)s, dgate, chansum, inv_vox, hidden, gate1p, w1, w2, gadd,
                     dw1, db1, dw2, db2, N, C, Ch);
  BRATS_CHECK_LAUNCH();
  return 0;
}
