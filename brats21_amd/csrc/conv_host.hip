// C-ABI entry points for the implicit-GEMM convolution: weight packing + forward/dgrad launch.
#include <stdlib.h>
#include "twin_begin.hpp"
#include "conv_igemm_x3.hpp"

// ---- chunk selection ---------------------------------------------------------------------------
int g_conv_vs8_mode = -1;
extern "C" int BRATS_API(brats_conv3d_set_vs8)(int mode) {
  const int old = g_conv_vs8_mode;
  g_conv_vs8_mode = mode < 0 ? -1 : (mode ? 1 : 0);  // 0 the 4x4x16-tile kernels, 1 conv_igemm_vs8 (4x8x16 tile)
  return old;
}
int g_conv_kp_mode = -1;  // -1 = BRATS_CONV_KP (default on), 0 / 1 = brats_conv3d_set_kp (conv_igemm.hpp: conv_kp_enabled)
extern "C" int BRATS_API(brats_conv3d_set_kp)(int mode) {
  const int old = g_conv_kp_mode;
  g_conv_kp_mode = mode < 0 ? -1 : (mode ? 1 : 0);
  return old;
}
static int conv_vs8_enabled() {
  if (g_conv_vs8_mode >= 0) return g_conv_vs8_mode;
  static int v = -1;
  if (v < 0) { const char* e = getenv("BRATS_CONV_VS8"); v = e ? atoi(e) : 1; }
  return v;
}

extern "C" int BRATS_API(brats_conv3d_chunk)(int dtype, int ksize, int dil, int c1, int c2, int cout) {
  if (dtype == BRATS_X3_BF16) {
    // split precision (conv_igemm_x3.hpp): hi + lo LDS tiles = the f32 tile's bytes; 24 channels keep two workgroups per CU
    if (ksize != 3) return 0;
    static const int x3[] = {24, 16, 8};
    // Cout = 48 (mod 96): the y-split roles need 145 VGPRs with 16-channel chunks -- three workgroups per CU (3 x 43 KB of
    // LDS): 48 -> 48 @2x128^3 1.417 -> 1.372 ms, 96 -> 48 2.376 -> 2.335 (same box, scripts/time_x3_conv.py); the cout-half
    // roles (Cout >= 96) are faster with 24 (48 -> 96: 2.15 vs 2.19 ms)
    const int rows16 = ceil_div(cout, 16);
    if (cout > 0 && rows16 % 3 == 0 && rows16 % 6 != 0 && c1 % 16 == 0 && (c2 <= 0 || c2 % 16 == 0)) return 16;
    for (int i = 0; i < 3; ++i)
      if (c1 % x3[i] == 0 && (c2 <= 0 || c2 % x3[i] == 0)) return x3[i];
    return 0;
  }
  // Cout = 48 (mod 96), bf16, 3x3x3 dilation 1: 24-channel chunks for the 4x8x16-tile kernel (conv_igemm_vs8.hpp)
  if (dtype == BRATS_BF16 && ksize == 3 && dil == 1 && cout > 0 && conv_vs8_enabled() && conv_vsplit_enabled()) {
    const int rows16 = ceil_div(cout, 16);
    if (rows16 % 3 == 0 && rows16 % 6 != 0 && c1 % 24 == 0 && (c2 <= 0 || c2 % 24 == 0)) return 24;
  }
  static int pref16 = -1;  // experiment switch: smaller K chunks -> smaller LDS tile -> more workgroups per CU
  if (pref16 < 0) { const char* e = getenv("BRATS_CONV_CK16"); pref16 = e ? atoi(e) : 0; }
  if (pref16 && dtype == BRATS_BF16 && c1 % 16 == 0 && (c2 <= 0 || c2 % 16 == 0)) return 16;
  static const int bf[] = {48, 32, 16, 8};
  static const int f32[] = {16, 8, 4};
  const int* cand = dtype == BRATS_BF16 ? bf : f32;
  const int n = dtype == BRATS_BF16 ? 4 : 3;
  for (int i = 0; i < n; ++i)
    if (c1 % cand[i] == 0 && (c2 <= 0 || c2 % cand[i] == 0)) return cand[i];
  return 0;
}

static int macro_steps(int dtype, int ksize, int ck) {
  const int taps = ksize * ksize * ksize;
  if (dtype == BRATS_BF16 || dtype == BRATS_X3_BF16) return (taps * (ck / 8) + 3) / 4;
  return (taps * ck / 4 + 3) / 4;
}

extern "C" size_t BRATS_API(brats_conv3d_packed_bytes)(int dtype, int ksize, int cin, int cout, int ck) {
  if (ck <= 0 || cin % ck) return 0;
  const int rows16 = ceil_div(cout, 16);
  // (split precision: a hi and a lo fragment per (macro-step, row16))
  return (size_t)(cin / ck) * macro_steps(dtype, ksize, ck) * rows16 * 64 * 16 * (dtype == BRATS_X3_BF16 ? 2 : 1);
}

// ---- weight packing ----------------------------------------------------------------------------
// out[chunk][ms][row16][lane][16 B]; see conv_igemm.hpp for the unit -> (tap, channel) map.
// X3 (split precision, conv_igemm_x3.hpp): out[chunk][ms][row16][hi | lo][lane][16 B], hi = rn16(w), lo = rn16(w - hi)
//
// One workgroup packs the fragments of (K chunk, 16-row group, PACK_ROWS of its rows): the torch-layout weights it needs are
// CONTIGUOUS runs (forward: ck * taps floats per output row; dgrad: PACK_ROWS * taps floats per K row), read coalesced into LDS,
// and every thread then gathers the 16-byte piece of one (macro-step, lane) from LDS.  (The first packer gathered from
// global memory with a stride of `taps` floats -- 33 launches x 9 us per EquiUnet-48 step.)
static constexpr int PACK_ROWS = 4;                            // fragment rows per workgroup
static constexpr int PACK_SUBS = 16 / PACK_ROWS;               // workgroups per 16-row fragment group
static constexpr int PACK_LDS_FLOATS = PACK_ROWS * 48 * 27;    // rows x the largest chunk x 27 taps (20736 B)
DEVI void pack_tile(const brats_pack_job& J, int blk, float* lds) {
  constexpr int RB = PACK_ROWS;
  const int sub = blk % PACK_SUBS, ft = (blk / PACK_SUBS) % J.rows16, chunk = (blk / PACK_SUBS) / J.rows16;
  const int taps = J.taps, ck = J.ck, row0 = ft * 16 + sub * RB;
  const bool fwd = J.mode == BRATS_PACK_FWD;
  // 16-byte loads, all of a thread's loads issued before the first LDS write, when every run starts and ends on a 16-byte
  // boundary (always, for input-channel counts that are multiples of 4); element-wise otherwise
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool al = ((size_t)J.w & 15) == 0 && (J.cin_real * taps) % 4 == 0 && (J.cin_off * taps) % 4 == 0 && blockDim.x == 256;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (fwd) {
    // lds[r][kcl][tap] = w[row0 + r][cin_off + chunk * ck + kcl][tap]  (zero beyond the real input channels / rows)
    const int seg = ck * taps;
    int kvalid = J.cin_real - J.cin_off - chunk * ck;
    kvalid = kvalid < 0 ? 0 : (kvalid > ck ? ck : kvalid);
    const int jlim = kvalid * taps;
    if (al && seg % 4 == 0 && jlim % 4 == 0) {
      const int seg4 = seg / 4, jl4 = jlim / 4;
      constexpr int IT = (48 * 27 / 4 + 63) / 64, RW = (RB + 3) / 4;  // 6 loads per row, RW rows per wave
      float4 v[RW][IT];
#pragma unroll
      for (int h = 0; h < RW; ++h) {
        const int r = wave + 4 * h;
        const float4* src = reinterpret_cast<const float4*>(J.w + ((size_t)(row0 + r) * J.cin_real + J.cin_off + chunk * ck) * taps);
        const bool rv = r < RB && row0 + r < J.rows;
#pragma unroll
        for (int u = 0; u < IT; ++u) v[h][u] = rv && lane + 64 * u < jl4 ? src[lane + 64 * u] : zero4;
      }
#pragma unroll
      for (int h = 0; h < RW; ++h)
#pragma unroll
        for (int u = 0; u < IT; ++u)
          if (wave + 4 * h < RB && lane + 64 * u < seg4) reinterpret_cast<float4*>(lds)[(wave + 4 * h) * seg4 + lane + 64 * u] = v[h][u];
    } else {
      for (int i = threadIdx.x; i < RB * seg; i += blockDim.x) {
        const int r = i / seg, j = i - r * seg;
        float v = 0.f;
        if (row0 + r < J.rows && j < jlim) v = J.w[((size_t)(row0 + r) * J.cin_real + J.cin_off + chunk * ck) * taps + j];
        lds[i] = v;
      }
    }
  } else {
    // lds[kcl][r][tap] = w[chunk * ck + kcl][cin_off + row0 + r][tap]
    const int seg = RB * taps;
    int rvalid = J.rows - row0, rv2 = J.cin_real - J.cin_off - row0;
    rvalid = rvalid < rv2 ? rvalid : rv2;
    rvalid = rvalid < 0 ? 0 : (rvalid > RB ? RB : rvalid);
    const int jlim = rvalid * taps;
    constexpr int LPS = RB * 27 / 4 <= 32 ? 32 : 64;  // lanes per K row's run (RB * 27 / 4 sixteen-byte pieces)
    constexpr int KPW = 64 / LPS;                     // K rows a wave loads per instruction
    if (al && seg % 4 == 0 && jlim % 4 == 0 && seg <= 4 * LPS) {
      const int seg4 = seg / 4, jl4 = jlim / 4, l = lane % LPS;
      constexpr int IT = 48 / (4 * KPW);
      float4 v[IT];
#pragma unroll
      for (int u = 0; u < IT; ++u) {
        const int kcl = (wave + 4 * u) * KPW + lane / LPS;
        const float4* src = reinterpret_cast<const float4*>(J.w + ((size_t)(chunk * ck + kcl) * J.cin_real + J.cin_off + row0) * taps);
        v[u] = kcl < ck && l < jl4 ? src[l] : zero4;
      }
#pragma unroll
      for (int u = 0; u < IT; ++u) {
        const int kcl = (wave + 4 * u) * KPW + lane / LPS;
        if (kcl < ck && l < seg4) reinterpret_cast<float4*>(lds)[kcl * seg4 + l] = v[u];
      }
    } else {
      for (int i = threadIdx.x; i < ck * seg; i += blockDim.x) {
        const int kcl = i / seg, j = i - kcl * seg;
        float v = 0.f;
        if (j < jlim) v = J.w[((size_t)(chunk * ck + kcl) * J.cin_real + J.cin_off + row0) * taps + j];
        lds[i] = v;
      }
    }
  }
  __syncthreads();
  const bool x3 = J.dtype == BRATS_X3_BF16;
  const int nx = x3 ? 2 : 1;
  const int pieces = J.ms_n * 4 * RB * nx;
  if (J.dtype == BRATS_F32) {
    // f32 fragments: 4 K units per lane, unit g = 4 * (4 * ms + e) + q -> (tap, channel) = (g / ck, g % ck)
    for (int p = threadIdx.x; p < pieces; p += blockDim.x) {
      const int r = p % RB, q = (p / RB) & 3, ms = p / (4 * RB);
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int g = 4 * (4 * ms + e) + q, tap = g / ck, kcl = g - tap * ck;
        v[e] = g < taps * ck ? (fwd ? lds[(r * ck + kcl) * taps + tap] : lds[(kcl * RB + r) * taps + (taps - 1 - tap)]) : 0.f;
      }
      const size_t o = ((((size_t)chunk * J.ms_n + ms) * J.rows16 + ft) * 64 + q * 16 + sub * RB + r) * 4;
      *reinterpret_cast<float4*>((float*)J.out + o) = make_float4(v[0], v[1], v[2], v[3]);
    }
    return;
  }
  // 16-bit fragments: a lane's piece = 8 consecutive K channels of one tap; unit g = 4 * ms + q
  const int upt = ck / 8;
  for (int p = threadIdx.x; p < pieces; p += blockDim.x) {
    const int r = p % RB, q = (p / RB) & 3;
    int t = p / (4 * RB), hl = 0;
    if (x3) { hl = t & 1; t >>= 1; }
    const int ms = t, g = 4 * ms + q, tap = g / upt, kcl0 = (g - tap * upt) * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float val = 0.f;
      if (g < taps * upt) val = fwd ? lds[(r * ck + kcl0 + e) * taps + tap] : lds[((kcl0 + e) * RB + r) * taps + (taps - 1 - tap)];
      v[e] = hl ? val - bf2f(f2bf(val)) : val;
    }
    const size_t o = (((((size_t)chunk * J.ms_n + ms) * J.rows16 + ft) * nx + hl) * 64 + q * 16 + sub * RB + r) * 8;
    Vec<bf16_t, 8>::store((bf16_t*)J.out + o, v);
  }
}
__global__ void __launch_bounds__(256) pack_weights_kernel(const brats_pack_job J) {
  __shared__ __attribute__((aligned(16))) float lds[PACK_LDS_FLOATS];
  pack_tile(J, blockIdx.x, lds);
}

static int pack_blocks(int kdim, int ck, int rows) { return (kdim / ck) * ceil_div(rows, 16) * PACK_SUBS; }
extern "C" int BRATS_API(brats_conv3d_pack_blocks)(int kdim, int ck, int rows) {
  return ck > 0 && kdim % ck == 0 && rows > 0 ? pack_blocks(kdim, ck, rows) : 0;
}

extern "C" int BRATS_API(brats_conv3d_pack_weights)(const float* w, void* packed, int dtype, int mode, int ksize,
                                         int cout_w, int cin_w, int cin_off, int cin_cnt, int ck,
                                         brats_stream_t s) {
  if (!w || !packed || (ksize != 1 && ksize != 3) || ck <= 0) BRATS_FAIL(BRATS_E_ARG, "pack_weights: bad argument");
  if (dtype != BRATS_BF16 && dtype != BRATS_X3_BF16 && dtype != BRATS_F32) BRATS_FAIL(BRATS_E_ARG, "pack_weights: dtype %d", dtype);
  brats_pack_job J;
  J.w = w; J.out = packed; J.dtype = dtype; J.mode = mode; J.taps = ksize * ksize * ksize;
  J.cin_w = cin_w; J.cin_real = cin_w; J.cin_off = cin_off;
  J.rows = mode == BRATS_PACK_FWD ? cout_w : cin_cnt;
  J.kdim = mode == BRATS_PACK_FWD ? cin_cnt : cout_w;
  if (J.kdim % ck) BRATS_FAIL(BRATS_E_ARG, "pack_weights: K channels %d not a multiple of chunk %d", J.kdim, ck);
  if (PACK_ROWS * ck * J.taps > PACK_LDS_FLOATS) BRATS_FAIL(BRATS_E_UNSUPPORTED, "pack_weights: chunk %d x %d taps exceeds the staging tile", ck, J.taps);
  J.rows16 = ceil_div(J.rows, 16); J.ck = ck; J.ms_n = macro_steps(dtype, ksize, ck); J.reserved = 0;
  J.total = (unsigned long long)(J.kdim / ck) * J.ms_n * J.rows16 * 64 * (dtype == BRATS_F32 ? 4 : 8) * (dtype == BRATS_X3_BF16 ? 2 : 1);
  hipLaunchKernelGGL(pack_weights_kernel, dim3(pack_blocks(J.kdim, ck, J.rows)), dim3(256), 0, (hipStream_t)s, J);
  BRATS_CHECK_LAUNCH();
  return 0;
}

// ---- multi-tensor packing: one launch for every layer of a network -----------------------------------------------
// blocks[i] = {job, block of that job (0 .. brats_conv3d_pack_blocks(kdim, ck, rows) - 1)}
__global__ void __launch_bounds__(256) pack_weights_multi_kernel(const brats_pack_job* __restrict__ jobs, const int* __restrict__ blocks) {
  __shared__ __attribute__((aligned(16))) float lds[PACK_LDS_FLOATS];
  const brats_pack_job J = jobs[blocks[blockIdx.x * 2]];
  if (PACK_ROWS * J.ck * J.taps > PACK_LDS_FLOATS) __builtin_trap();
  pack_tile(J, blocks[blockIdx.x * 2 + 1], lds);
}
extern "C" int BRATS_API(brats_conv3d_pack_weights_multi)(const brats_pack_job* jobs, const int* blocks, int nblocks, brats_stream_t s) {
  if (!jobs || !blocks || nblocks <= 0) BRATS_FAIL(BRATS_E_ARG, "pack_weights_multi: empty job / block table");
  hipLaunchKernelGGL(pack_weights_multi_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)s, jobs, blocks);
  BRATS_CHECK_LAUNCH();
  return 0;
}

// ---- forward / dgrad ---------------------------------------------------------------------------
extern "C" int BRATS_API(brats_conv3d_split_granule)(int cout) { return conv_choose_tile(ceil_div(cout, 16)).nf * 16; }

extern "C" int BRATS_API(brats_conv3d_tiles_per_sample)(int D, int H, int W) {
  return ceil_div(D, CONV_TZ) * ceil_div(H, CONV_TY) * ceil_div(W, CONV_TX);
}

bool conv_pre_supported(int ck, int dil, int rows16);                            // conv_bf16_k3_pre.hip
int conv_pre_launch(const ConvParams& p, int ck, int dil, hipStream_t st);

struct ConvPre { const float* ss1 = nullptr; const float* ss2 = nullptr; int act = 0; float slope = 0.f; bool on = false; };

bool conv_bst_supported(int ck, int dil, int rows16);                            // conv_bf16_k3_bst.hip
int conv_bst_launch(const ConvParams& p, int ck, int dil, hipStream_t st);
struct ConvBst { const void* by = nullptr; int bypitch = 0; const float* bss = nullptr; float slope = 0.f; bool on = false; };

static int conv_fwd_impl(const void* x1, int c1, int pitch1, const void* x2, int c2, int pitch2, const float* xamax,
                         const void* packed_w, const float* bias, void* y, int ypitch, void* y2, int y2pitch,
                         int ysplit, float* stats, int dtype, int ksize, int dil, int N, int D, int H, int W,
                         int cout, brats_stream_t s, const ConvPre& pre = ConvPre{}, const ConvBst& bst = ConvBst{}) {
  if (!x1 || !packed_w || !y || c1 <= 0 || N <= 0 || D <= 0 || H <= 0 || W <= 0 || cout <= 0)
    BRATS_FAIL(BRATS_E_ARG, "conv3d_fwd: null pointer or non-positive size");
  if (c2 > 0 && !x2) BRATS_FAIL(BRATS_E_ARG, "conv3d_fwd: c2 > 0 but x2 is NULL");
  if (c2 < 0) c2 = 0;
  if (cout % 4) BRATS_FAIL(BRATS_E_UNSUPPORTED, "conv3d_fwd: cout %d must be a multiple of 4", cout);
  const int align = dtype == BRATS_BF16 ? 8 : 4;
  if (pitch1 % align || (c2 && pitch2 % align) || ypitch % 4)
    BRATS_FAIL(BRATS_E_ARG, "conv3d_fwd: channel pitches must keep 16-byte loads / 4-channel stores aligned");
  if (dtype == BRATS_X3_BF16 && (ksize != 3 || ((size_t)x1 & 15) || (c2 && ((size_t)x2 & 15))))
    BRATS_FAIL(BRATS_E_UNSUPPORTED, "conv3d_fwd: the split-precision kernel is 3x3x3 only, on 16-byte aligned f32 tensors");
  {  // staged pieces are addressed by 32-bit byte offsets inside one sample (buffer_load voffset)
    const int mp = pitch1 > pitch2 ? pitch1 : pitch2;
    if ((double)D * H * W * mp * (dtype == BRATS_BF16 ? 2 : 4) >= 2147483648.0)
      BRATS_FAIL(BRATS_E_UNSUPPORTED, "conv3d_fwd: one sample of %dx%dx%d x pitch %d exceeds the 2 GiB buffer-offset range", D, H, W, mp);
  }
  const int ck = BRATS_API(brats_conv3d_chunk)(dtype, ksize, dil, c1, c2, cout);
  if (!ck) BRATS_FAIL(BRATS_E_UNSUPPORTED, "conv3d_fwd: no channel chunk divides c1=%d c2=%d", c1, c2);
  ConvParams p;
  p.x1 = x1; p.x2 = x2; p.c1 = c1; p.c2 = c2; p.p1 = pitch1; p.p2 = pitch2;
  p.wpk = packed_w; p.bias = bias; p.y = y; p.ypitch = ypitch; p.stats = stats;
  p.y2 = y2; p.y2pitch = y2pitch; p.ysplit = ysplit; p.xamax = xamax;
  p.ss1 = pre.ss1; p.ss2 = pre.ss2; p.pre_act = pre.act; p.pre_slope = pre.slope;
  p.by = bst.by; p.bypitch = bst.bypitch; p.bss = bst.bss; p.bslope = bst.slope;
  if (y2) {
    const ConvTileChoice tc = conv_choose_tile(ceil_div(cout, 16));
    if (ysplit <= 0 || ysplit >= cout || ysplit % (tc.nf * 16) || y2pitch % 4)
      BRATS_FAIL(BRATS_E_UNSUPPORTED, "conv3d_fwd: ysplit=%d must be a multiple of %d inside (0, cout)", ysplit, tc.nf * 16);
  }
  p.N = N; p.D = D; p.H = H; p.W = W; p.cout = cout; p.rows16 = ceil_div(cout, 16);
  p.nchunks = (c1 + c2) / ck;
  p.tz = ceil_div(D, CONV_TZ); p.ty = ceil_div(H, CONV_TY); p.tx = ceil_div(W, CONV_TX);
#ifdef BRATS_VS8_STAMPS
  { const char* e = getenv("BRATS_VS8_STAMP_PTR"); p.stamps = e ? (long long*)strtoull(e, nullptr, 0) : nullptr; }
#endif
  hipStream_t st = (hipStream_t)s;
  if (ksize == 1) dil = 1;
  if (pre.on) return conv_pre_launch(p, ck, dil, st);
  if (bst.on) return dtype == BRATS_X3_BF16 ? conv_x3_bst_launch(p, ck, dil, st) : conv_bst_launch(p, ck, dil, st);
#define GO(T) \
  if (ksize == 3 && dil == 1) return conv_launch<T, 3, 1>(p, ck, st); \
  if (ksize == 3 && dil == 2) return conv_launch<T, 3, 2>(p, ck, st); \
  if (ksize == 1) return conv_launch<T, 1, 1>(p, ck, st);
  if (dtype == BRATS_BF16) { GO(bf16_t) }
  else if (dtype == BRATS_X3_BF16) {
    if (dil == 1) return conv_x3_launch<1>(p, ck, st);
    if (dil == 2) return conv_x3_launch<2>(p, ck, st);
  }
#ifndef BRATS_FP16  // (the f32 kernels live in translation units of their own that are not built twice)
  else if (dtype == BRATS_F32) { GO(float) }
#endif
#undef GO
  BRATS_FAIL(BRATS_E_UNSUPPORTED, "conv3d_fwd: unsupported dtype=%d ksize=%d dilation=%d", dtype, ksize, dil);
}

extern "C" int BRATS_API(brats_conv3d_fwd)(const void* x1, int c1, int pitch1, const void* x2, int c2, int pitch2,
                                const void* packed_w, const float* bias, void* y, int ypitch, void* y2, int y2pitch,
                                int ysplit, float* stats, int dtype, int ksize, int dil, int N, int D, int H, int W,
                                int cout, brats_stream_t s) {
  return conv_fwd_impl(x1, c1, pitch1, x2, c2, pitch2, nullptr, packed_w, bias, y, ypitch, y2, y2pitch, ysplit, stats, dtype, ksize,
                       dil, N, D, H, W, cout, s);
}

extern "C" int BRATS_API(brats_conv3d_pre_ok)(int dtype, int ksize, int dil, int c1, int c2, int cout) {
  if (dtype != BRATS_BF16 || ksize != 3 || cout <= 0) return 0;
  const int ck = BRATS_API(brats_conv3d_chunk)(dtype, ksize, dil, c1, c2, cout);
  return ck > 0 && conv_pre_supported(ck, dil, ceil_div(cout, 16)) ? 1 : 0;
}

extern "C" int BRATS_API(brats_conv3d_fwd_pre)(const void* x1, int c1, int pitch1, const float* ss1, const void* x2, int c2, int pitch2,
                                    const float* ss2, int act, float slope, const void* packed_w, const float* bias, void* y,
                                    int ypitch, float* stats, int dtype, int dil, int N, int D, int H, int W, int cout,
                                    brats_stream_t s) {
  if (dtype != BRATS_BF16) BRATS_FAIL(BRATS_E_UNSUPPORTED, "conv3d_fwd_pre: 16-bit activations only");
  if (act != BRATS_ACT_RELU && act != BRATS_ACT_LEAKY) BRATS_FAIL(BRATS_E_UNSUPPORTED, "conv3d_fwd_pre: relu / leakyrelu only (act %d)", act);
  if (!ss1 && !ss2) BRATS_FAIL(BRATS_E_ARG, "conv3d_fwd_pre: neither source has a scale / shift table (use brats_conv3d_fwd)");
  if (((size_t)ss1 | (size_t)ss2) & 15) BRATS_FAIL(BRATS_E_ARG, "conv3d_fwd_pre: scale / shift tables must be 16-byte aligned");
  if (c1 % 8 || (c2 > 0 && c2 % 8)) BRATS_FAIL(BRATS_E_ARG, "conv3d_fwd_pre: channel counts must be multiples of 8");
  ConvPre pre;
  pre.ss1 = ss1; pre.ss2 = ss2; pre.act = act; pre.slope = slope; pre.on = true;
  return conv_fwd_impl(x1, c1, pitch1, x2, c2, pitch2, nullptr, packed_w, bias, y, ypitch, nullptr, 0, 0, stats, dtype, 3, dil, N, D, H,
                       W, cout, s, pre);
}

extern "C" int BRATS_API(brats_conv3d_bstats_ok)(int dtype, int ksize, int dil, int c1, int cout) {
  if ((dtype != BRATS_BF16 && dtype != BRATS_X3_BF16) || ksize != 3 || cout <= 0) return 0;
  const int ck = BRATS_API(brats_conv3d_chunk)(dtype, ksize, dil, c1, 0, cout);
  if (ck <= 0) return 0;
  return (dtype == BRATS_X3_BF16 ? conv_x3_bst_supported(ck, dil, ceil_div(cout, 16)) : conv_bst_supported(ck, dil, ceil_div(cout, 16))) ? 1 : 0;
}

extern "C" int BRATS_API(brats_conv3d_fwd_bstats)(const void* x1, int c1, int pitch1, const void* packed_w, void* y, int ypitch,
                                       const void* fwd_y, int fwd_pitch, const float* scale_shift, int act, float slope,
                                       float* tile_stats, int dtype, int dil, int N, int D, int H, int W, int cout,
                                       brats_stream_t s) {
  if (dtype != BRATS_BF16) BRATS_FAIL(BRATS_E_UNSUPPORTED, "conv3d_fwd_bstats: 16-bit activations only");
  if (act != BRATS_ACT_RELU && act != BRATS_ACT_LEAKY) BRATS_FAIL(BRATS_E_UNSUPPORTED, "conv3d_fwd_bstats: relu / leakyrelu only (act %d)", act);
  if (!fwd_y || !scale_shift || !tile_stats) BRATS_FAIL(BRATS_E_ARG, "conv3d_fwd_bstats: null pointer");
  if (fwd_pitch % 4 || ((size_t)fwd_y & 7)) BRATS_FAIL(BRATS_E_ARG, "conv3d_fwd_bstats: the forward tensor must keep 8-byte loads aligned");
  ConvBst bst;
  bst.by = fwd_y; bst.bypitch = fwd_pitch; bst.bss = scale_shift; bst.slope = act == BRATS_ACT_RELU ? 0.f : slope; bst.on = true;
  return conv_fwd_impl(x1, c1, pitch1, nullptr, 0, 0, nullptr, packed_w, nullptr, y, ypitch, nullptr, 0, 0, tile_stats, dtype, 3, dil, N, D,
                       H, W, cout, s, ConvPre{}, bst);
}

extern "C" int BRATS_API(brats_conv3d_x3_fwd)(const void* x1, int c1, int pitch1, const void* x2, int c2, int pitch2,
                                   const float* xamax, const void* packed_w, const float* bias, void* y, int ypitch, void* y2,
                                   int y2pitch, int ysplit, float* stats, int dtype, int dil, int N, int D, int H, int W,
                                   int cout, brats_stream_t s) {
  if (dtype != BRATS_X3_BF16) BRATS_FAIL(BRATS_E_ARG, "conv3d_x3_fwd: dtype must be BRATS_X3_F16 or BRATS_X3_BF16");
  return conv_fwd_impl(x1, c1, pitch1, x2, c2, pitch2, xamax, packed_w, bias, y, ypitch, y2, y2pitch, ysplit, stats, dtype, 3, dil,
                       N, D, H, W, cout, s);
}

extern "C" int BRATS_API(brats_conv3d_x3_fwd_bstats)(const void* x1, int c1, int pitch1, const float* xamax, const void* packed_w, void* y,
                                          int ypitch, const void* fwd_y, int fwd_pitch, const float* scale_shift, int act,
                                          float slope, float* tile_stats, int dtype, int dil, int N, int D, int H, int W, int cout,
                                          brats_stream_t s) {
  if (dtype != BRATS_X3_BF16) BRATS_FAIL(BRATS_E_ARG, "conv3d_x3_fwd_bstats: dtype must be BRATS_X3_F16 or BRATS_X3_BF16");
  if (act != BRATS_ACT_RELU && act != BRATS_ACT_LEAKY) BRATS_FAIL(BRATS_E_UNSUPPORTED, "conv3d_x3_fwd_bstats: relu / leakyrelu only (act %d)", act);
  if (!fwd_y || !scale_shift || !tile_stats) BRATS_FAIL(BRATS_E_ARG, "conv3d_x3_fwd_bstats: null pointer");
  if (fwd_pitch % 4 || ((size_t)fwd_y & 15)) BRATS_FAIL(BRATS_E_ARG, "conv3d_x3_fwd_bstats: the forward tensor must keep 16-byte loads aligned");
  ConvBst bst;
  bst.by = fwd_y; bst.bypitch = fwd_pitch; bst.bss = scale_shift; bst.slope = act == BRATS_ACT_RELU ? 0.f : slope; bst.on = true;
  return conv_fwd_impl(x1, c1, pitch1, nullptr, 0, 0, xamax, packed_w, nullptr, y, ypitch, nullptr, 0, 0, tile_stats, dtype, 3, dil, N, D,
                       H, W, cout, s, ConvPre{}, bst);
}
#include "twin_end.hpp"
