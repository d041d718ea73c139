// Squeeze-and-excitation gate of MONAI's ResidualSELayer as the reference builds it (networks/equiunet2021.py:204-205:
// ResidualSELayer(3, C, r=2, relu, sigmoid)): for sample n
//     gap = mean_v z[n][v][:]                      (the channel sums come out of the EvoNorm pass, norm.hip)
//     hid = relu(W1 gap + b1),  W1: [C/2][C]
//     gate = sigmoid(W2 hid + b2), W2: [C][C/2]
//     out = z + z * gate = z * (1 + gate)          (applied by brats_channel_scale / folded into the EvoNorm backward)
// and its backward incl. the four parameter gradients.  These were ~20 tiny ATen launches per block (F.linear,
// torch.autograd.grad) x 7 blocks in a launch-bound step; here: ONE launch forward and ONE launch backward.
//
// The arithmetic is nothing (4 x 74 k multiply-adds per sample at C = 384); what a launch costs is its chain of DEPENDENT
// memory round trips.  The first version of this file (a thread owning an output column and walking the C rows of the matrix
// one load at a time, a wave owning one row after the other) measured 47 us forward / 118 us backward -- 1.15 ms per step of
// EquiUnetASSPEvo-48, more than the ATen launches it replaced.  Now: 1024 threads per workgroup; every matrix-vector product
// splits its reduction axis over as many thread slices as fit (cols_dot: coalesced float4 columns x K-slices, partial sums
// through LDS) or handles four rows per wave at once (rows_dot4), loads issued in batches of 4 - 8 before their use; layers
// with large matrices spread the output rows / columns and the parameter-gradient stores over several workgroups, each of
// which recomputes the (cheap, L2-resident) hidden vector it needs.  All reductions in a fixed order that depends on the
// shapes only (shuffles in a fixed tree, slices and samples added in order) -> bitwise reproducible.
#include "common.hpp"
#include "se.hpp"

constexpr int SE_NT = 1024;      // threads per workgroup
constexpr int SE_MAXN = 8;       // samples per launch (backward: all samples in one workgroup's registers)
constexpr int SE_MAX_WGS = 16;
constexpr int SE_PART_FLOATS = 16384;  // LDS budget of cols_dot's partial sums (64 KB)

DEVI float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}

// y[r] = sum_k M[r*K + k] * x[k] for r in [r_lo, r_hi): a wave owns four rows at a time (lanes stride over k, 4 loads in flight
// per step), rows dealt round-robin to the waves
template <typename F>
DEVI void rows_dot4(const float* __restrict__ M, int r_lo, int r_hi, int K, const float* x /*LDS*/, int wave, int nwaves, int lane,
                    F&& put) {
  for (int r0 = r_lo + wave * 4; r0 < r_hi; r0 += nwaves * 4) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = lane; k < K; k += 64) {
      const float xv = x[k];
      float w[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) w[u] = r0 + u < r_hi ? M[(size_t)(r0 + u) * K + k] : 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) s[u] += w[u] * xv;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] = wave_sum(s[u]);
    if (lane == 0) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (r0 + u < r_hi) put(r0 + u, s[u]);
    }
  }
}

// put(n, r, s): s = sum_k M[k*ld + r] * x[n*K + k] for r < R, n < N (N <= 8): thread t owns VW adjacent columns (r = VW * (t % (R/VW)))
// of K-slice t / (R/VW) -- reads of M coalesce over r --, the S slices' partial sums are added in slice order through LDS
// (part: S * N * R <= SE_PART_FLOATS floats).  Ends with a barrier (part is free again, put()'s LDS writes are visible).
template <int VW, typename F>
DEVI void cols_dot(const float* __restrict__ M, int ld, int K, int R, const float* x /*LDS [N][K]*/, int N, float* part /*LDS*/,
                   int tid, int nt, F&& put) {
  const int RV = R / VW;
  int S = nt / RV;
  S = S < K ? S : K;
  S = S < SE_PART_FLOATS / (N * R) ? S : SE_PART_FLOATS / (N * R);
  const int rv = tid % RV, sl = tid / RV;
  if (sl < S) {
    float acc[SE_MAXN][VW];
#pragma unroll
    for (int n = 0; n < SE_MAXN; ++n)
#pragma unroll
      for (int v = 0; v < VW; ++v) acc[n][v] = 0.f;
    constexpr int U = VW == 4 ? 4 : 8;
    const float* col = M + rv * VW;
    int k = sl;
    for (; k + (U - 1) * S < K; k += U * S) {
      float w[U][VW];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if constexpr (VW == 4) {
          const float4 q = *(const float4*)(col + (size_t)(k + u * S) * ld);
          w[u][0] = q.x; w[u][1] = q.y; w[u][2] = q.z; w[u][3] = q.w;
        } else {
          w[u][0] = col[(size_t)(k + u * S) * ld];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int n = 0; n < SE_MAXN; ++n)
          if (n < N) {
            const float xv = x[n * K + k + u * S];
#pragma unroll
            for (int v = 0; v < VW; ++v) acc[n][v] += w[u][v] * xv;
          }
    }
    for (; k < K; k += S) {
      float w[VW];
      if constexpr (VW == 4) {
        const float4 q = *(const float4*)(col + (size_t)k * ld);
        w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.w;
      } else {
        w[0] = col[(size_t)k * ld];
      }
#pragma unroll
      for (int n = 0; n < SE_MAXN; ++n)
        if (n < N) {
          const float xv = x[n * K + k];
#pragma unroll
          for (int v = 0; v < VW; ++v) acc[n][v] += w[v] * xv;
        }
    }
#pragma unroll
    for (int n = 0; n < SE_MAXN; ++n)
      if (n < N) {
#pragma unroll
        for (int v = 0; v < VW; ++v) part[((size_t)sl * N + n) * R + rv * VW + v] = acc[n][v];
      }
  }
  __syncthreads();
  for (int i = tid; i < N * R; i += nt) {
    const int n = i / R, r = i % R;
    float s = 0.f;
    for (int l = 0; l < S; ++l) s += part[((size_t)l * N + n) * R + r];
    put(n, r, s);
  }
  __syncthreads();
}

DEVI void slice_of(int total, int parts, int idx, int& lo, int& hi) {
  const int per = (total + parts - 1) / parts;
  lo = idx * per < total ? idx * per : total;
  hi = lo + per < total ? lo + per : total;
}

// grid = (G, N), block = SE_NT.  LDS: gap[C] + hid[Ch].  Every workgroup of a sample computes the whole hidden vector, then
// its slice of the gate rows
__global__ __launch_bounds__(SE_NT) void se_fwd_kernel(const float* __restrict__ chansum, SeFwdFold fold, float inv_vox,
                                                       const float* __restrict__ w1, const float* __restrict__ b1,
                                                       const float* __restrict__ w2, const float* __restrict__ b2,
                                                       float* __restrict__ gate1p, float* __restrict__ hidden, int C, int Ch) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  float* gap = (float*)lds_raw;
  float* hid = gap + C;
  const int n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const bool first = blockIdx.x == 0;
  const int cpg = fold.numsum ? C / fold.groups : 1;
  for (int c = tid; c < C; c += blockDim.x) {
    float cs;
    if (fold.numsum) {
      cs = fold.mean_rstd[(n * fold.groups + c / cpg) * 2 + 1] * fold.gamma[c] * fold.numsum[(size_t)n * C + c] + fold.beta[c] * fold.voxels;
      if (first) fold.chansum_out[(size_t)n * C + c] = cs;
    } else {
      cs = chansum[(size_t)n * C + c];
    }
    gap[c] = cs * inv_vox;
  }
  __syncthreads();
  rows_dot4(w1, 0, Ch, C, gap, wave, nw, lane, [&](int j, float s) {
    const float h = fmaxf(s + b1[j], 0.f);
    hid[j] = h;
    if (first) hidden[(size_t)n * Ch + j] = h;
  });
  __syncthreads();
  int c_lo, c_hi;
  slice_of(C, gridDim.x, blockIdx.x, c_lo, c_hi);
  rows_dot4(w2, c_lo, c_hi, Ch, hid, wave, nw, lane, [&](int c, float s) {
    gate1p[(size_t)n * C + c] = 1.f + 1.f / (1.f + __expf(-(s + b2[c])));
  });
}

// grid = G, block = SE_NT.  LDS: ds[N][C], gap[N][C], dh[N][Ch], hid[N][Ch], part[SE_PART_FLOATS].
// Workgroup g: d loss / d pre-sigmoid (all), the hidden gradient (all: every workgroup needs it), then ITS slice of the
// columns c (gadd, the folded EvoNorm sums, dW2 rows, db2) and of the rows j (dW1 rows, db1)
template <int VW>
__global__ __launch_bounds__(SE_NT) void se_bwd_kernel(const float* __restrict__ dgate, SeFold fold, const float* __restrict__ chansum,
                                                       float inv_vox, const float* __restrict__ hidden, const float* __restrict__ gate1p,
                                                       const float* __restrict__ w1, const float* __restrict__ w2,
                                                       float* __restrict__ gadd, float* __restrict__ dw1, float* __restrict__ db1,
                                                       float* __restrict__ dw2, float* __restrict__ db2, int N, int C, int Ch,
                                                       int accum /* add to the parameter gradients of the samples before */) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  float* ds = (float*)lds_raw;          // [N][C]   d loss / d (pre-sigmoid)
  float* gap = ds + (size_t)N * C;      // [N][C]
  float* dh = gap + (size_t)N * C;      // [N][Ch]  d loss / d (pre-relu)
  float* hid = dh + (size_t)N * Ch;     // [N][Ch]
  float* part = hid + (size_t)N * Ch;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int cpg = fold.raw5 ? C / fold.groups : 1;
  for (int i = tid; i < N * C; i += nt) {
    float dg;
    if (fold.raw5) {  // d loss / d gate = sum_v do * z, z = num * rstd_g * gamma_c + beta_c (norm.hip: evonorm_fwd_kernel)
      const int n = i / C, c = i % C;
      const float r = fold.mean_rstd[(n * fold.groups + c / cpg) * 2 + 1];
      dg = r * fold.gamma[c] * fold.raw5[(size_t)i * 5 + 1] + fold.beta[c] * fold.raw5[(size_t)i * 5];
    } else {
      dg = dgate[i];
    }
    const float g = gate1p[i] - 1.f;
    ds[i] = dg * g * (1.f - g);
    gap[i] = chansum[i] * inv_vox;
  }
  for (int i = tid; i < N * Ch; i += nt) hid[i] = hidden[i];
  __syncthreads();
  // dh[n][j] = [hid > 0] * sum_c W2[c][j] ds[n][c]
  cols_dot<VW>(w2, Ch, C, Ch, ds, N, part, tid, nt, [&](int n, int j, float s) { dh[n * Ch + j] = hid[n * Ch + j] > 0.f ? s : 0.f; });
  int c_lo, c_hi, j_lo, j_hi;
  slice_of(C / VW, gridDim.x, blockIdx.x, c_lo, c_hi);
  c_lo *= VW;
  c_hi *= VW;
  slice_of(Ch, gridDim.x, blockIdx.x, j_lo, j_hi);
  // dgap[n][c] = sum_j W1[j][c] dh[n][j], handed on as gadd = dgap / V; with it the sums of the EvoNorm backward for the
  // gradient dz = do * (1 + gate) + gadd it will read (linear in the raw sums over do)
  if (c_hi > c_lo)
    cols_dot<VW>(w1 + c_lo, C, Ch, c_hi - c_lo, dh, N, part, tid, nt, [&](int n, int r, float s) {
      const size_t i = (size_t)n * C + c_lo + r;
      const float ga = s * inv_vox;
      gadd[i] = ga;
      if (fold.raw5) {
        const float gs = gate1p[i];
        const float* q = fold.raw5 + i * 5;
        fold.red3[i * 3] = gs * q[0] + ga * fold.voxels;
        fold.red3[i * 3 + 1] = gs * q[1] + ga * q[3];
        fold.red3[i * 3 + 2] = gs * q[2] + ga * q[4];
      }
    });
  // parameter gradients: samples added in order
  for (int i = tid; i < (c_hi - c_lo) * Ch; i += nt) {
    const int c = c_lo + i / Ch, j = i % Ch;   // dW2[c][j] = sum_n ds[n][c] hid[n][j]
    float s = accum ? dw2[(size_t)c * Ch + j] : 0.f;
    for (int n = 0; n < N; ++n) s += ds[n * C + c] * hid[n * Ch + j];
    dw2[(size_t)c * Ch + j] = s;
  }
  for (int i = tid; i < (j_hi - j_lo) * C; i += nt) {
    const int j = j_lo + i / C, c = i % C;     // dW1[j][c] = sum_n dh[n][j] gap[n][c]
    float s = accum ? dw1[(size_t)j * C + c] : 0.f;
    for (int n = 0; n < N; ++n) s += dh[n * Ch + j] * gap[n * C + c];
    dw1[(size_t)j * C + c] = s;
  }
  for (int c = c_lo + tid; c < c_hi; c += nt) {
    float s = accum ? db2[c] : 0.f;
    for (int n = 0; n < N; ++n) s += ds[n * C + c];
    db2[c] = s;
  }
  for (int j = j_lo + tid; j < j_hi; j += nt) {
    float s = accum ? db1[j] : 0.f;
    for (int n = 0; n < N; ++n) s += dh[n * Ch + j];
    db1[j] = s;
  }
}

static int se_workgroups(int C, int Ch) {
  const long cells = (long)C * Ch;  // one workgroup per ~4.6 k matrix elements: 1 at C <= 96, 4 at 192, 16 at 384
  const long g = cells / 4608;
  return g < 1 ? 1 : (g > SE_MAX_WGS ? SE_MAX_WGS : (int)g);
}

int brats_se_fwd_launch(const float* chansum, const SeFwdFold& fold, float inv_vox, const float* w1, const float* b1, const float* w2,
                        const float* b2, float* gate1p, float* hidden, int N, int C, int Ch, hipStream_t st) {
  if ((!chansum && !fold.numsum) || !w1 || !b1 || !w2 || !b2 || !gate1p || !hidden || N <= 0 || C <= 0 || Ch <= 0)
    BRATS_FAIL(BRATS_E_ARG, "se_fwd: null pointer or non-positive size");
  if (fold.numsum && (!fold.mean_rstd || !fold.gamma || !fold.beta || !fold.chansum_out || fold.groups <= 0 || C % fold.groups))
    BRATS_FAIL(BRATS_E_ARG, "se_fwd: incomplete EvoNorm fold arguments");
  const size_t lds = (size_t)(C + Ch) * 4;
  if (lds > 64 * 1024) BRATS_FAIL(BRATS_E_UNSUPPORTED, "se_fwd: C = %d, C/r = %d exceed the 64 KB LDS budget", C, Ch);
  hipLaunchKernelGGL(se_fwd_kernel, dim3(se_workgroups(C, Ch), N), dim3(SE_NT), lds, st, chansum, fold, inv_vox, w1, b1, w2, b2, gate1p,
                     hidden, C, Ch);
  BRATS_CHECK_LAUNCH();
  return 0;
}

extern "C" int brats_se_fwd(const float* chansum, float inv_vox, const float* w1, const float* b1, const float* w2, const float* b2,
                            float* gate1p, float* hidden, int N, int C, int Ch, brats_stream_t s) {
  if (!chansum) BRATS_FAIL(BRATS_E_ARG, "se_fwd: null pointer or non-positive size");
  return brats_se_fwd_launch(chansum, SeFwdFold{}, inv_vox, w1, b1, w2, b2, gate1p, hidden, N, C, Ch, (hipStream_t)s);
}

int brats_se_bwd_launch(const float* dgate, const SeFold& fold, const float* chansum, float inv_vox, const float* hidden,
                        const float* gate1p, const float* w1, const float* w2, float* gadd, float* dw1, float* db1, float* dw2,
                        float* db2, int N, int C, int Ch, hipStream_t st) {
  if ((!dgate && !fold.raw5) || !chansum || !hidden || !gate1p || !w1 || !w2 || !gadd || !dw1 || !db1 || !dw2 || !db2 || N <= 0 ||
      C <= 0 || Ch <= 0)
    BRATS_FAIL(BRATS_E_ARG, "se_bwd: null pointer or non-positive size");
  if (fold.raw5 && (!fold.mean_rstd || !fold.gamma || !fold.beta || !fold.red3 || fold.groups <= 0 || C % fold.groups))
    BRATS_FAIL(BRATS_E_ARG, "se_bwd: incomplete EvoNorm fold arguments");
  const bool v4 = C % 4 == 0 && Ch % 4 == 0 && ((uintptr_t)w1 % 16 == 0) && ((uintptr_t)w2 % 16 == 0);
  // samples in groups of at most SE_MAXN (the per-sample state lives in one workgroup's LDS); the parameter gradients of
  // later groups are added to the earlier ones in order (launches on one stream: bitwise reproducible).  A per-GPU batch
  // above 8 (small patches) used to fail here AFTER a whole forward pass.
  const int G = se_workgroups(C, Ch);
  for (int n0 = 0; n0 < N; n0 += SE_MAXN) {
    const int nb = N - n0 < SE_MAXN ? N - n0 : SE_MAXN;
    const size_t lds = ((size_t)nb * (2 * C + 2 * Ch) + SE_PART_FLOATS) * 4;
    if (C > SE_NT || lds > 160 * 1024)
      BRATS_FAIL(BRATS_E_UNSUPPORTED, "se_bwd: C = %d, C/r = %d exceed the workgroup's budget", C, Ch);
    SeFold f = fold;
    if (f.raw5) {
      f.raw5 += (size_t)n0 * C * 5;
      f.red3 += (size_t)n0 * C * 3;
      f.mean_rstd += (size_t)n0 * f.groups * 2;
    }
    const float* dg = dgate ? dgate + (size_t)n0 * C : nullptr;
    const size_t oc = (size_t)n0 * C, oh = (size_t)n0 * Ch;
    if (v4) {
      static std::atomic<uint64_t> done{0};
      BRATS_ENSURE_LDS_ATTR(se_bwd_kernel<4>, 160 * 1024, done);
      hipLaunchKernelGGL(se_bwd_kernel<4>, dim3(G), dim3(SE_NT), lds, st, dg, f, chansum + oc, inv_vox, hidden + oh, gate1p + oc, w1, w2,
                         gadd + oc, dw1, db1, dw2, db2, nb, C, Ch, n0 > 0 ? 1 : 0);
    } else {
      static std::atomic<uint64_t> done{0};
      BRATS_ENSURE_LDS_ATTR(se_bwd_kernel<1>, 160 * 1024, done);
      hipLaunchKernelGGL(se_bwd_kernel<1>, dim3(G), dim3(SE_NT), lds, st, dg, f, chansum + oc, inv_vox, hidden + oh, gate1p + oc, w1, w2,
                         gadd + oc, dw1, db1, dw2, db2, nb, C, Ch, n0 > 0 ? 1 : 0);
    }
  }
  BRATS_CHECK_LAUNCH();
  return 0;
}

extern "C" int brats_se_bwd(const float* dgate, const float* chansum, float inv_vox, const float* hidden, const float* gate1p,
                            const float* w1, const float* w2, float* gadd, float* dw1, float* db1, float* dw2, float* db2,
                            int N, int C, int Ch, brats_stream_t s) {
  if (!dgate) BRATS_FAIL(BRATS_E_ARG, "se_bwd: null pointer or non-positive size");
  return brats_se_bwd_launch(dgate, SeFold{}, chansum, inv_vox, hidden, gate1p, w1, w2, gadd, dw1, db1, dw2, db2, N, C, Ch,
                             (hipStream_t)s);
}
