// Segmentation heads: 1x1x1 conv C -> K (K <= 4) + bias, then trilinear (align_corners) up-sampling to
// full resolution, emitted as NCDHW f32 logits for the PyTorch Dice loss (learning/engine.py:312-333).
// Reference: conv1x1 networks/equiunet2020.py:37-41 (outconv :441, deep heads :443-458).
// Pure HBM-bound kernels (AI <= 3 FLOP/B): no MFMA on purpose.
#include "twin_begin.hpp"
#include "common.hpp"

int brats_lerp_adjoint_f32_planes(const float* in, float* out, size_t outer, int Lout, int Lin, size_t inner, hipStream_t st);

static inline int sgrid(size_t total, int block) {
  size_t b = (total + block - 1) / block;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}
constexpr int HEAD_KMAX = 4;

// thread = (voxel, 16-byte channel vector): fully coalesced loads; the K partial dot products of a voxel's
// C/VW threads are summed through LDS (a thread-per-voxel loop over channels ran at 2.3 TB/s)
// PRE (brats_gn_head_fwd): x is the last layer's raw convolution output and the head reads z = act(x * scale + shift) --
// GroupNorm + relu / leakyrelu applied on load, rounded to the storage type as the stored z would be -- so that z itself
// (2 * C bytes per voxel written and read back) never exists.  pre.scale_shift: [N][C][2]; pre.nslope: 0 for relu.
// evo: the EvoNorm form z = x * sigmoid(x) * scale + shift (brats_evonorm_head_fwd: scale = rstd_g * gamma_c * (1 + gate),
// shift = beta_c * (1 + gate) per (n, channel) -- the output of brats_evonorm_se_fwd's last pass, recomputed on load).
struct HeadPre { const float* scale_shift; float nslope; int evo; };
DEVI float head_pre(float x, float sc, float sh, const HeadPre& pre) {
  if (pre.evo) return x * __builtin_amdgcn_rcpf(1.f + __expf(-x)) * sc + sh;  // (norm.hip: evonorm_fwd_kernel's expression)
  const float p = x * sc + sh;
  return p > 0.f ? p : p * pre.nslope;
}
template <typename T, bool PRE = false>
__global__ void head_conv_kernel(const T* __restrict__ x, int xpitch, const float* __restrict__ w, const float* __restrict__ b,
                                 float* __restrict__ low, int C, int K, size_t voxels, HeadPre pre) {
  constexpr int VW = 16 / sizeof(T);
  extern __shared__ float sm[];
  float* ws = sm;                       // [K][C]
  float* red = sm + HEAD_KMAX * C;      // [256][HEAD_KMAX]
  for (int i = threadIdx.x; i < K * C; i += blockDim.x) ws[i] = w[i];
  __syncthreads();
  const int n = blockIdx.y;
  const int cv = C / VW, vl_n = blockDim.x / cv;
  const int mycv = threadIdx.x % cv, myvl = threadIdx.x / cv, c0 = mycv * VW;
  const T* xb = x + (size_t)n * voxels * xpitch;
  float psc[VW], psh[VW];
  if constexpr (PRE) {
#pragma unroll
    for (int j = 0; j < VW; ++j) {
      const bool in = myvl < vl_n;
      psc[j] = in ? pre.scale_shift[((size_t)n * C + c0 + j) * 2] : 0.f;
      psh[j] = in ? pre.scale_shift[((size_t)n * C + c0 + j) * 2 + 1] : 0.f;
    }
  }
  for (size_t vbase = (size_t)blockIdx.x * vl_n; vbase < voxels; vbase += (size_t)gridDim.x * vl_n) {
    const size_t v = vbase + myvl;
    float acc[HEAD_KMAX] = {0.f, 0.f, 0.f, 0.f};
    if (myvl < vl_n && v < voxels) {
      float a[VW];
      Vec<T, VW>::load(xb + v * xpitch + c0, a);
      if constexpr (PRE) {
#pragma unroll
        for (int j = 0; j < VW; ++j) a[j] = to_f<T>(from_f<T>(head_pre(a[j], psc[j], psh[j], pre)));
      }
#pragma unroll
      for (int k = 0; k < HEAD_KMAX; ++k)
        if (k < K) {
#pragma unroll
          for (int j = 0; j < VW; ++j) acc[k] += a[j] * ws[k * C + c0 + j];
        }
    }
#pragma unroll
    for (int k = 0; k < HEAD_KMAX; ++k) red[threadIdx.x * HEAD_KMAX + k] = acc[k];
    __syncthreads();
    for (int i = threadIdx.x; i < vl_n * K; i += blockDim.x) {
      const int vl = i / K, k = i % K;
      if (vbase + vl < voxels) {
        float t = b ? b[k] : 0.f;
        for (int c = 0; c < cv; ++c) t += red[(vl * cv + c) * HEAD_KMAX + k];
        low[((size_t)n * K + k) * voxels + vbase + vl] = t;
      }
    }
    __syncthreads();
  }
}

// bf16, C <= 64 (the full-resolution head: 48 -> 3 at 128^3, 403 MB read): the K <= 4 logits of 16 voxels are ONE pair of
// v_mfma_f32_16x16x32_bf16 -- rows = classes (3 of 16 used: the matrix pipe is idle anyway), columns = voxels, k =
// channels; the B operand of lane (voxel l & 15, q = l >> 4) is one 16-byte load of 8 consecutive channels, so a wave
// instruction reads 1 KB of contiguous activations, and nothing goes through LDS or a cross-lane reduction (the first form
// read its weights from LDS per element and reduced six channel-vector threads per voxel through LDS: 2.7 TB/s).
template <bool PRE = false>
__global__ void __launch_bounds__(256) head_conv_mfma_kernel(const bf16_t* __restrict__ x, int xpitch, const float* __restrict__ w,
                                                             const float* __restrict__ b, float* __restrict__ low, int C, int K,
                                                             size_t voxels, HeadPre pre) {
  const int lane = threadIdx.x & 63, v = lane & 15, q = lane >> 4;
  const int n = blockIdx.y;
  // PRE: the lane's 16 channels (8q.. and 32 + 8q..) are fixed, their scale / shift live in registers
  float psc[2][8], psh[2][8];
  if constexpr (PRE) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = 32 * s2 + 8 * q + j;
        psc[s2][j] = c < C ? pre.scale_shift[((size_t)n * C + c) * 2] : 0.f;
        psh[s2][j] = c < C ? pre.scale_shift[((size_t)n * C + c) * 2 + 1] : 0.f;
      }
  }
  auto pre_act = [&](bf16x8 raw, int s2) {
    typedef __attribute__((ext_vector_type(4))) uint32_t u4;
    u4 u = __builtin_bit_cast(u4, raw);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float lo, hi;
      unpack2(u[i], lo, hi);
      u[i] = pack2(head_pre(lo, psc[s2][2 * i], psh[s2][2 * i], pre), head_pre(hi, psc[s2][2 * i + 1], psh[s2][2 * i + 1], pre));
    }
    return __builtin_bit_cast(bf16x8, u);
  };
  // the f32 weights enter as three bf16 terms (w = hi + mid + lo exactly: 3 x 8 mantissa bits), so the products are
  // those of the f32 weights with the bf16 activations, as in the reference's arithmetic; six MFMAs per 16 voxels
  bf16x8 wa[3][2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 t[3];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = 32 * s + 8 * q + j;
      float rest = (v < K && c < C) ? w[v * C + c] : 0.f;
#pragma unroll
      for (int part = 0; part < 3; ++part) {
        const bf16_t hb = f2bf(rest);
        t[part][j] = (short)hb;
        rest -= bf2f(hb);
      }
    }
#pragma unroll
    for (int part = 0; part < 3; ++part) wa[part][s] = __builtin_bit_cast(bf16x8, t[part]);
  }
  float bias[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bias[r] = (b && r < K) ? b[r] : 0.f;
  const bf16_t* xb = x + (size_t)n * voxels * xpitch;
  // (the range ends with the last voxel's C channels: a channel-slice view of a wider buffer must not be read past its slice)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)xb, (short)0, (int)((voxels - 1) * xpitch * 2 + C * 2), 0x00020000);
  // channels 8q.. of the first and 32 + 8q.. of the second k-step exist?  (C < 32: the lanes of the missing channel groups
  // must not read the next voxel / the neighbouring slice -- zero weights do not neutralise an Inf or NaN there)
  const int dead0 = 8 * q < C ? 0 : -1;
  const int dead1 = 32 + 8 * q < C ? 0 : -1;
  const size_t wave_id = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4;
  float* lowp = low + (size_t)n * K * voxels;
  for (size_t v0 = wave_id * 64; v0 < voxels; v0 += nwaves * 64) {  // 4 chunks of 16 voxels per iteration: 8 loads in flight
    bf16x8 xb0[4], xb1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const size_t vv = v0 + 16 * i + v;
      const int off = vv < voxels ? (int)(vv * xpitch * 2) + 16 * q : -1;
      xb0[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, off | dead0, 0, 0));
      xb1[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, (off + 64) | dead1 | (off >> 31), 0, 0));
    }
    if constexpr (PRE) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { xb0[i] = pre_act(xb0[i], 0); xb1[i] = pre_act(xb1[i], 1); }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int part = 2; part >= 0; --part) {  // small terms first
        acc = MFMA16_16x16x32(wa[part][0], xb0[i], acc);
        acc = MFMA16_16x16x32(wa[part][1], xb1[i], acc);
      }
      const size_t vv = v0 + 16 * i + v;
      if (q == 0 && vv < voxels) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (r < K) lowp[(size_t)r * voxels + vv] = acc[r] + bias[r];
      }
    }
  }
}

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

// grid = (chunks of the (yo, xo / 4) plane, Do, planes): z coefficients are scalar, a thread writes 4 consecutive x (one
// 16-byte store) -- the first form computed one f32 per thread behind three 64-bit div / mod chains and wrote the deep
// heads' 50 MB planes at 0.85 TB/s.  Same weights and summation order (w = wz * wy * wx, k = 0..7): bit-identical.
__global__ void __launch_bounds__(256) upsample_planes_kernel(const float* __restrict__ low, float* __restrict__ out, int D, int H, int W, int sc,
                                                              float sd, float sh, float sw) {
  const int Ho = H * sc, Wo = W * sc, Do = D * sc, W4 = Wo / 4;
  const int zo = blockIdx.y;
  const size_t pl = blockIdx.z;
  const Lerp lz = lerp_coef(zo, D, sd);
  const float* p0 = low + pl * D * H * W + (size_t)lz.i0 * H * W;
  const float* p1 = low + pl * D * H * W + (size_t)lz.i1 * H * W;
  float* o = out + (pl * Do + zo) * (size_t)Ho * Wo;
  for (int it = blockIdx.x * blockDim.x + threadIdx.x; it < Ho * W4; it += gridDim.x * blockDim.x) {
    const int yo = it / W4, x0 = (it % W4) * 4;
    const Lerp ly = lerp_coef(yo, H, sh);
    f32x4 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const Lerp lx = lerp_coef(x0 + j, W, sw);
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float* p = (k & 4) ? p1 : p0;
        const int yy = (k & 2) ? ly.i1 : ly.i0, xx = (k & 1) ? lx.i1 : lx.i0;
        const float wgt = ((k & 4) ? lz.w1 : lz.w0) * ((k & 2) ? ly.w1 : ly.w0) * ((k & 1) ? lx.w1 : lx.w0);
        acc += wgt * p[yy * W + xx];
      }
      r[j] = acc;
    }
    *(f32x4*)(o + (size_t)yo * Wo + x0) = r;
  }
}

template <bool PRE>
static int head_conv_launch(const void* x, int xpitch, const float* w, const float* b, float* low, int dtype, int N, int C, int K,
                            size_t vox, HeadPre pre, hipStream_t st) {
  const int vw = dtype == BRATS_BF16 ? 8 : 4;
  const int cvh = C / vw, vlh = 256 / cvh;
  if (cvh > 256) BRATS_FAIL(BRATS_E_UNSUPPORTED, "head_fwd: C too large");
  dim3 grid(sgrid((vox + vlh - 1) / vlh, 1), N);
  const size_t ldsh = (size_t)(HEAD_KMAX * C + 256 * HEAD_KMAX) * sizeof(float);
  if (dtype == BRATS_BF16 && C <= 64 && (double)vox * xpitch * 2 < 2147483648.0) {
    const size_t waves = (vox + 63) / 64;
    const unsigned gx = (unsigned)(waves / 4 < 1 ? 1 : (waves / 4 > 8192 ? 8192 : waves / 4));
    hipLaunchKernelGGL(head_conv_mfma_kernel<PRE>, dim3(gx, N), dim3(256), 0, st, (const bf16_t*)x, xpitch, w, b, low, C, K, vox, pre);
  } else if (dtype == BRATS_BF16)
    hipLaunchKernelGGL((head_conv_kernel<bf16_t, PRE>), grid, dim3(256), ldsh, st, (const bf16_t*)x, xpitch, w, b, low, C, K, vox, pre);
  else
    hipLaunchKernelGGL((head_conv_kernel<float, PRE>), grid, dim3(256), ldsh, st, (const float*)x, xpitch, w, b, low, C, K, vox, pre);
  BRATS_CHECK_LAUNCH();
  return 0;
}

// logits [N][K][voxels] = conv1x1(act(y * scale + shift)) + bias: the output head on the last layer's raw convolution output
// (HeadPre above) -- replaces brats_affine_act_fwd + brats_head_fwd(scale 1) for that layer; relu / leakyrelu.
extern "C" int BRATS_API(brats_gn_head_fwd)(const void* y, int ypitch, const float* scale_shift, int act, float slope, const float* w,
                                 const float* b, float* out, int dtype, int N, int C, int K, int voxels, brats_stream_t s) {
  const int vw = dtype == BRATS_BF16 ? 8 : 4;
  if (!y || !scale_shift || !w || !out || K < 1 || K > HEAD_KMAX || C % vw || ypitch % vw)
    BRATS_FAIL(BRATS_E_ARG, "gn_head_fwd: bad argument (K<=4, C multiple of %d)", vw);
  if (act > BRATS_ACT_LEAKY) BRATS_FAIL(BRATS_E_UNSUPPORTED, "gn_head_fwd: relu / leakyrelu only (act=%d)", act);
  return head_conv_launch<true>(y, ypitch, w, b, out, dtype, N, C, K, (size_t)voxels,
                                HeadPre{scale_shift, act == BRATS_ACT_RELU ? 0.f : slope, 0}, (hipStream_t)s);
}

// the same for EquiUnetASSPEvo's last block (EvoNorm + ResidualSELayer -> out_conv): logits = conv1x1(y * sigmoid(y) * scale +
// shift) + bias with scale_shift [N][C][2] = { rstd_g * gamma_c * (1 + gate), beta_c * (1 + gate) } -- the block's output is
// recomputed on load, rounded to the storage type as the stored tensor would be (bit-identical logits), never stored.
extern "C" int BRATS_API(brats_evonorm_head_fwd)(const void* y, int ypitch, const float* scale_shift, const float* w, const float* b,
                                      float* out, int dtype, int N, int C, int K, int voxels, brats_stream_t s) {
  const int vw = dtype == BRATS_BF16 ? 8 : 4;
  if (!y || !scale_shift || !w || !out || K < 1 || K > HEAD_KMAX || C % vw || ypitch % vw)
    BRATS_FAIL(BRATS_E_ARG, "evonorm_head_fwd: bad argument (K<=4, C multiple of %d)", vw);
  return head_conv_launch<true>(y, ypitch, w, b, out, dtype, N, C, K, (size_t)voxels, HeadPre{scale_shift, 0.f, 1}, (hipStream_t)s);
}

extern "C" int BRATS_API(brats_head_fwd)(const void* x, int xpitch, const float* w, const float* b, float* lowres, float* out,
                              int dtype, int N, int C, int K, int D, int H, int W, int scale, brats_stream_t s) {
  const int vw = dtype == BRATS_BF16 ? 8 : 4;
  if (!x || !w || !out || K < 1 || K > HEAD_KMAX || C % vw || xpitch % vw || scale < 1)
    BRATS_FAIL(BRATS_E_ARG, "head_fwd: bad argument (K<=4, C multiple of %d)", vw);
  if (scale > 1 && !lowres) BRATS_FAIL(BRATS_E_ARG, "head_fwd: lowres workspace required when scale > 1");
  hipStream_t st = (hipStream_t)s;
  const size_t vox = (size_t)D * H * W;
  float* low = scale > 1 ? lowres : out;
  if (int rc = head_conv_launch<false>(x, xpitch, w, b, low, dtype, N, C, K, vox, HeadPre{nullptr, 0.f, 0}, st)) return rc;
  if (scale > 1) {
    if ((W * scale) % 4) BRATS_FAIL(BRATS_E_UNSUPPORTED, "head_fwd: up-sampled width %d must be a multiple of 4", W * scale);
    const int items = H * scale * (W * scale / 4);
    hipLaunchKernelGGL(upsample_planes_kernel, dim3((unsigned)((items + 255) / 256), (unsigned)(D * scale), (unsigned)(N * K)), dim3(256), 0, st,
                       (const float*)low, out, D, H, W, scale, ac_scale(D, D * scale), ac_scale(H, H * scale), ac_scale(W, W * scale));
  }
  BRATS_CHECK_LAUNCH();
  return 0;
}

// dx[v][c] = sum_k dlow[k][v]*w[k][c];  dw[k][c] += sum_v dlow[k][v]*x[v][c];  db[k] += sum_v dlow[k][v]
template <typename T>
__global__ void __launch_bounds__(256) head_bwd_kernel(const T* __restrict__ x, int xpitch, const float* __restrict__ w, const float* __restrict__ dlow,
                                T* __restrict__ dx, int dxpitch, float* __restrict__ dw /* per-block partials */, int C, int K,
                                size_t voxels) {
  constexpr int VW = 16 / sizeof(T);
  extern __shared__ float sm[];
  float* ws = sm;  // [K][C]
  for (int i = threadIdx.x; i < K * C; i += blockDim.x) ws[i] = w[i];
  __syncthreads();
  const int n = blockIdx.y;
  const int cv = C / VW, vl_n = blockDim.x / cv;
  const int mycv = threadIdx.x % cv, myvl = threadIdx.x / cv, c0 = mycv * VW;
  float aw[HEAD_KMAX][VW];
  float ab[HEAD_KMAX];
#pragma unroll
  for (int k = 0; k < HEAD_KMAX; ++k) {
    ab[k] = 0.f;
#pragma unroll
    for (int j = 0; j < VW; ++j) aw[k][j] = 0.f;
  }
  if (myvl < vl_n) {
    const T* xb = x + (size_t)n * voxels * xpitch + c0;
    T* dxb = dx ? dx + (size_t)n * voxels * dxpitch + c0 : nullptr;
    // the thread's channel vector is fixed: its K x VW weights live in registers (read from LDS per element the pass was
    // LDS-bound), two voxels are in flight per iteration
    float wr[HEAD_KMAX][VW];
#pragma unroll
    for (int k = 0; k < HEAD_KMAX; ++k)
#pragma unroll
      for (int j = 0; j < VW; ++j) wr[k][j] = k < K ? ws[k * C + c0 + j] : 0.f;
    const float* dl = dlow + (size_t)n * K * voxels;
    auto body = [&](const float* a, const float* g, float* o) {
#pragma unroll
      for (int j = 0; j < VW; ++j) o[j] = 0.f;
#pragma unroll
      for (int k = 0; k < HEAD_KMAX; ++k) {
        ab[k] += g[k];
#pragma unroll
        for (int j = 0; j < VW; ++j) {
          aw[k][j] += g[k] * a[j];
          o[j] += g[k] * wr[k][j];
        }
      }
    };
    const size_t stride = (size_t)gridDim.x * vl_n;
    size_t v = (size_t)blockIdx.x * vl_n + myvl;
    for (; v + stride < voxels; v += 2 * stride) {
      float a0[VW], a1[VW], o0[VW], o1[VW], g0[HEAD_KMAX], g1[HEAD_KMAX];
      Vec<T, VW>::load(xb + v * xpitch, a0);
      Vec<T, VW>::load(xb + (v + stride) * xpitch, a1);
#pragma unroll
      for (int k = 0; k < HEAD_KMAX; ++k) {
        g0[k] = k < K ? dl[(size_t)k * voxels + v] : 0.f;
        g1[k] = k < K ? dl[(size_t)k * voxels + v + stride] : 0.f;
      }
      body(a0, g0, o0);
      body(a1, g1, o1);
      if (dxb) {
        Vec<T, VW>::store(dxb + v * dxpitch, o0);
        Vec<T, VW>::store(dxb + (v + stride) * dxpitch, o1);
      }
    }
    if (v < voxels) {
      float a0[VW], o0[VW], g0[HEAD_KMAX];
      Vec<T, VW>::load(xb + v * xpitch, a0);
#pragma unroll
      for (int k = 0; k < HEAD_KMAX; ++k) g0[k] = k < K ? dl[(size_t)k * voxels + v] : 0.f;
      body(a0, g0, o0);
      if (dxb) Vec<T, VW>::store(dxb + v * dxpitch, o0);
    }
  }
  float* scr = sm + K * C;  // [vl_n][K][C] + [vl_n][K]
  __syncthreads();
  if (myvl < vl_n) {
    for (int k = 0; k < K; ++k) {
#pragma unroll
      for (int j = 0; j < VW; ++j) scr[(myvl * K + k) * C + c0 + j] = aw[k][j];
      if (mycv == 0) scr[vl_n * K * C + myvl * K + k] = ab[k];
    }
  }
  __syncthreads();
  // one partial vector [K*C + K] per block, added in block order afterwards (no float atomics)
  float* part = dw + ((size_t)n * gridDim.x + blockIdx.x) * (K * C + K);
  for (int i = threadIdx.x; i < K * C; i += blockDim.x) {
    float t = 0.f;
    for (int l = 0; l < vl_n; ++l) t += scr[l * K * C + i];
    part[i] = t;
  }
  if ((int)threadIdx.x < K) {
    float t = 0.f;
    for (int l = 0; l < vl_n; ++l) t += scr[vl_n * K * C + l * K + threadIdx.x];
    part[K * C + threadIdx.x] = t;
  }
}

constexpr int HEAD_MAX_BLOCKS = 1024;
static size_t head_lerp_floats(int N, int K, int D, int H, int W, int scale) {
  if (scale <= 1) return 0;
  const size_t p = (size_t)N * K;
  const size_t dlow = p * D * H * W, t1 = p * D * (H * scale) * (W * scale), t2 = p * D * H * (W * scale);
  return dlow + t1 + t2;
}
// workspace = up-sampling adjoint temporaries (scale > 1) + per-block partial sums of dw / db + the [K*C + K] totals
extern "C" size_t BRATS_API(brats_head_bwd_ws_bytes)(int N, int C, int K, int D, int H, int W, int scale) {
  return (head_lerp_floats(N, K, D, H, W, scale) + (size_t)(N * HEAD_MAX_BLOCKS + 1) * (K * C + K)) * sizeof(float);
}

extern "C" int BRATS_API(brats_head_bwd)(const void* x, int xpitch, const float* w, const float* dout, float* ws, void* dx, int dxpitch,
                              float* dw, float* db, int dtype, int N, int C, int K, int D, int H, int W, int scale,
                              brats_stream_t s) {
  const int vw = dtype == BRATS_BF16 ? 8 : 4;
  if (!x || !w || !dout || !dw || !db || K < 1 || K > HEAD_KMAX || C % vw || xpitch % vw || (dx && dxpitch % vw) ||
      C / vw > 128)
    BRATS_FAIL(BRATS_E_ARG, "head_bwd: bad argument");
  if (!ws) BRATS_FAIL(BRATS_E_ARG, "head_bwd: workspace required");
  hipStream_t st = (hipStream_t)s;
  const size_t vox = (size_t)D * H * W, p = (size_t)N * K;
  const float* dlow = dout;
  if (scale > 1) {
    float* dl = ws;
    float* t1 = dl + p * vox;
    float* t2 = t1 + p * D * (H * scale) * (W * scale);
    int rc;
    if ((rc = brats_lerp_adjoint_f32_planes(dout, t1, p, D * scale, D, (size_t)H * scale * W * scale, st))) return rc;
    if ((rc = brats_lerp_adjoint_f32_planes(t1, t2, p * D, H * scale, H, (size_t)W * scale, st))) return rc;
    if ((rc = brats_lerp_adjoint_f32_planes(t2, dl, p * D * H, W * scale, W, 1, st))) return rc;
    dlow = dl;
  }
  const int cv = C / vw, vl = 256 / cv;
  size_t gx = (vox + (size_t)vl * 16 - 1) / ((size_t)vl * 16);
  dim3 grid((unsigned)(gx < 1 ? 1 : (gx > HEAD_MAX_BLOCKS ? HEAD_MAX_BLOCKS : gx)), N);
  const size_t lds = (size_t)(K * C + vl * K * C + vl * K) * sizeof(float);
  float* part = ws + head_lerp_floats(N, K, D, H, W, scale);
  float* tot = part + (size_t)N * HEAD_MAX_BLOCKS * (K * C + K);
  if (dtype == BRATS_BF16)
    hipLaunchKernelGGL(head_bwd_kernel<bf16_t>, grid, dim3(256), lds, st, (const bf16_t*)x, xpitch, w, dlow, (bf16_t*)dx,
                       dxpitch, part, C, K, vox);
  else
    hipLaunchKernelGGL(head_bwd_kernel<float>, grid, dim3(256), lds, st, (const float*)x, xpitch, w, dlow, (float*)dx, dxpitch,
                       part, C, K, vox);
  (void)tot;
  brats_ordered_sum2(part, dw, K * C, db, N * (int)grid.x, K * C + K, st);  // totals straight into dw [K][C] and db [K]
  BRATS_CHECK_LAUNCH();
  return 0;
}
#include "twin_end.hpp"
