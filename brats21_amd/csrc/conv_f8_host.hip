// C-ABI entry points of the fp8 (e4m3) convolution path: weight packing with per-row scales, |max| reduction,
// forward / dgrad launch (kernels: conv_igemm_f8.hpp).
#include "twin_begin.hpp"
#include "conv_igemm_f8.hpp"

template <> int conv_f8_launch<1>(const ConvF8Params& p, int ck, hipStream_t st);
template <> int conv_f8_launch<2>(const ConvF8Params& p, int ck, hipStream_t st);

extern "C" int BRATS_API(brats_conv3d_f8_chunk)(int c1, int c2) {
  static const int cand[] = {48, 32, 16};
  for (int i = 0; i < 3; ++i)
    if (c1 > 0 && c1 % cand[i] == 0 && (c2 <= 0 || c2 % cand[i] == 0)) return cand[i];
  return 0;
}

static int f8_macro_steps(int ck) { return (27 * (ck / 16) + 7) / 8; }

extern "C" size_t BRATS_API(brats_conv3d_f8_packed_bytes)(int cin, int cout, int ck) {
  if (ck <= 0 || ck % 16 || cin % ck) return 0;
  const int rows16 = ceil_div(cout, 16);
  return (size_t)rows16 * 16 * 4 + (size_t)(cin / ck) * f8_macro_steps(ck) * rows16 * 64 * 32;
}

// GEMM element (row, kc, tap) of the torch-layout weight [cout_w][cin_w][27]
DEVI float f8_weight_at(const float* w, int mode, int cin_w, int cin_off, int row, int kc, int tap) {
  if (mode == BRATS_PACK_FWD) return w[((size_t)row * cin_w + cin_off + kc) * 27 + tap];
  return w[((size_t)kc * cin_w + cin_off + row) * 27 + (26 - tap)];
}

// wscale[row] = power of two that puts the row's |max| into [128, 256) after division (1 for padding rows)
__global__ void __launch_bounds__(256) f8_row_scale_kernel(const float* __restrict__ w, float* __restrict__ wscale, int mode,
                                                           int cin_w, int cin_off, int rows, int kdim) {
  const int row = blockIdx.x;
  float m = 0.f;
  if (row < rows)
    for (int i = threadIdx.x; i < kdim * 27; i += 256) m = fmaxf(m, fabsf(f8_weight_at(w, mode, cin_w, cin_off, row, i / 27, i % 27)));
  __shared__ float red[256];
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) wscale[row] = row < rows ? f8_scale_from_amax(red[0]) : 1.f;
}

// out[chunk][ms][row16][half][lane][16 B]: lane (q, v) of fragment (ms, row16) holds row 16*row16 + v, units
// 8*ms + 2*q + half (a unit = 16 consecutive channels of one tap; conv_igemm_f8.hpp)
__global__ void f8_pack_kernel(const float* __restrict__ w, const float* __restrict__ wscale, uint32_t* __restrict__ out,
                               int mode, int cin_w, int cin_off, int rows, int rows16, int kdim, int ck, int ms_n, size_t total) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // one dword = 4 channels
  if (idx >= total) return;
  const int d = idx % 4;
  size_t t = idx / 4;
  const int lane = t % 64; t /= 64;
  const int half = t % 2; t /= 2;
  const int ft = t % rows16; t /= rows16;
  const int ms = t % ms_n;
  const int chunk = t / ms_n;
  const int q = lane >> 4, row = ft * 16 + (lane & 15);
  const int upt = ck / 16, g = 8 * ms + 2 * q + half;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (g < 27 * upt && row < rows) {
    const int tap = g / upt, kc = chunk * ck + (g % upt) * 16 + d * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = f8_weight_at(w, mode, cin_w, cin_off, row, kc + j, tap);
  }
  const float sc = wscale[row];
  s16x2 r = {0, 0};
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, v[0], v[1], sc, false);
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, v[2], v[3], sc, true);
  out[idx] = __builtin_bit_cast(uint32_t, r);
}

extern "C" int BRATS_API(brats_conv3d_f8_pack_weights)(const float* w, void* packed, int mode, int cout_w, int cin_w, int cin_off,
                                            int cin_cnt, int ck, brats_stream_t s) {
  if (!w || !packed || ck <= 0 || ck % 16) BRATS_FAIL(BRATS_E_ARG, "f8_pack_weights: bad argument");
  const int rows = mode == BRATS_PACK_FWD ? cout_w : cin_cnt;
  const int kdim = mode == BRATS_PACK_FWD ? cin_cnt : cout_w;
  if (kdim % ck) BRATS_FAIL(BRATS_E_ARG, "f8_pack_weights: K channels %d not a multiple of chunk %d", kdim, ck);
  const int rows16 = ceil_div(rows, 16), ms = f8_macro_steps(ck);
  float* wscale = (float*)packed;
  uint32_t* frag = (uint32_t*)((char*)packed + (size_t)rows16 * 16 * 4);
  hipLaunchKernelGGL(f8_row_scale_kernel, dim3(rows16 * 16), dim3(256), 0, (hipStream_t)s, w, wscale, mode, cin_w, cin_off, rows, kdim);
  const size_t total = (size_t)(kdim / ck) * ms * rows16 * 2 * 64 * 4;
  hipLaunchKernelGGL(f8_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)s, w, wscale, frag, mode, cin_w,
                     cin_off, rows, rows16, kdim, ck, ms, total);
  BRATS_CHECK_LAUNCH();
  return 0;
}

// ---- |max| of an NDHWC tensor (for tensors whose producer did not record it) --------------------------------
// max is order-independent, so the integer atomicMax on the bits of the non-negative float is deterministic
template <typename T>
__global__ void __launch_bounds__(256) absmax_kernel(const T* __restrict__ x, int pitch, size_t rows, int C, uint32_t* __restrict__ out) {
  constexpr int VW = 16 / sizeof(T);
  const int cv = C / VW;
  const size_t total = rows * cv;
  float m = 0.f;
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (size_t)gridDim.x * blockDim.x) {
    float a[VW];
    Vec<T, VW>::load(x + (it / cv) * pitch + (it % cv) * VW, a);
#pragma unroll
    for (int j = 0; j < VW; ++j) m = fmaxf(m, fabsf(a[j]));
  }
  __shared__ float red[256];
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0 && __float_as_uint(red[0]) > *(volatile uint32_t*)out) atomicMax(out, __float_as_uint(red[0]));
}

extern "C" int BRATS_API(brats_absmax)(const void* x, int pitch, int dtype, size_t rows, int C, float* out, brats_stream_t s) {
  const int vw = dtype == BRATS_BF16 ? 8 : 4;
  if (!x || !out || C % vw || pitch % vw) BRATS_FAIL(BRATS_E_ARG, "absmax: C and pitch must be multiples of %d", vw);
  hipStream_t st = (hipStream_t)s;
  if (hipMemsetAsync(out, 0, 4, st) != hipSuccess) BRATS_FAIL(BRATS_E_HIP, "absmax: memset failed");
  size_t b = (rows * (C / vw) + 2047) / 2048;
  const int blocks = (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
  if (dtype == BRATS_BF16)
    hipLaunchKernelGGL(absmax_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)x, pitch, rows, C, (uint32_t*)out);
  else
    hipLaunchKernelGGL(absmax_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)x, pitch, rows, C, (uint32_t*)out);
  BRATS_CHECK_LAUNCH();
  return 0;
}

// ---- forward / dgrad ---------------------------------------------------------------------------
extern "C" int BRATS_API(brats_conv3d_f8_fwd)(const void* x1, int c1, int pitch1, const float* amax1, const void* x2, int c2, int pitch2,
                                   const float* amax2, float xscale, const void* packed_w, const float* bias, void* y, int ypitch,
                                   void* y2, int y2pitch, int ysplit, float* stats, int dil, int N, int D, int H, int W, int cout,
                                   brats_stream_t s) {
  if (!x1 || !packed_w || !y || c1 <= 0 || N <= 0 || D <= 0 || H <= 0 || W <= 0 || cout <= 0)
    BRATS_FAIL(BRATS_E_ARG, "conv3d_f8_fwd: null pointer or non-positive size");
  if (c2 > 0 && !x2) BRATS_FAIL(BRATS_E_ARG, "conv3d_f8_fwd: c2 > 0 but x2 is NULL");
  if (c2 < 0) c2 = 0;
  if (cout % 4) BRATS_FAIL(BRATS_E_UNSUPPORTED, "conv3d_f8_fwd: cout %d must be a multiple of 4", cout);
  if (pitch1 % 8 || (c2 && pitch2 % 8) || ypitch % 4)
    BRATS_FAIL(BRATS_E_ARG, "conv3d_f8_fwd: channel pitches must keep 16-byte loads / 4-channel stores aligned");
  if (!amax1 && !(xscale > 0.f)) BRATS_FAIL(BRATS_E_ARG, "conv3d_f8_fwd: needs the |max| of the input or a positive static scale");
  if (amax1 && c2 && !amax2) BRATS_FAIL(BRATS_E_ARG, "conv3d_f8_fwd: |max| of x2 missing");
  {
    const int mp = pitch1 > pitch2 ? pitch1 : pitch2;
    if ((double)D * H * W * mp * 2 >= 2147483648.0)
      BRATS_FAIL(BRATS_E_UNSUPPORTED, "conv3d_f8_fwd: one sample of %dx%dx%d x pitch %d exceeds the 2 GiB buffer-offset range", D, H, W, mp);
  }
  const int ck = BRATS_API(brats_conv3d_f8_chunk)(c1, c2);
  if (!ck) BRATS_FAIL(BRATS_E_UNSUPPORTED, "conv3d_f8_fwd: channel counts c1=%d c2=%d must be multiples of 16", c1, c2);
  ConvF8Params pp;
  ConvParams& p = pp.c;
  p.x1 = x1; p.x2 = x2; p.c1 = c1; p.c2 = c2; p.p1 = pitch1; p.p2 = pitch2;
  p.rows16 = ceil_div(cout, 16);
  pp.wscale = (const float*)packed_w;
  p.wpk = (const char*)packed_w + (size_t)p.rows16 * 16 * 4;
  p.bias = bias; p.y = y; p.ypitch = ypitch; p.stats = stats;
  p.y2 = y2; p.y2pitch = y2pitch; p.ysplit = ysplit;
  if (y2) {
    const ConvTileChoice tc = conv_choose_tile(p.rows16);
    if (ysplit <= 0 || ysplit >= cout || ysplit % (tc.nf * 16) || y2pitch % 4)
      BRATS_FAIL(BRATS_E_UNSUPPORTED, "conv3d_f8_fwd: ysplit=%d must be a multiple of %d inside (0, cout)", ysplit, tc.nf * 16);
  }
  p.N = N; p.D = D; p.H = H; p.W = W; p.cout = cout;
  p.nchunks = (c1 + c2) / ck;
  p.tz = ceil_div(D, CONV_TZ); p.ty = ceil_div(H, CONV_TY); p.tx = ceil_div(W, CONV_TX);
  pp.amax1 = amax1; pp.amax2 = c2 ? amax2 : nullptr; pp.xscale = xscale;
  if (dil == 1) return conv_f8_launch<1>(pp, ck, (hipStream_t)s);
  if (dil == 2) return conv_f8_launch<2>(pp, ck, (hipStream_t)s);
  BRATS_FAIL(BRATS_E_UNSUPPORTED, "conv3d_f8_fwd: unsupported dilation %d", dil);
}
#include "twin_end.hpp"
