// Direct (gather) convolution for the ASPP head (SimpleASPPEVO, networks/equiunet2021.py:121-189): the four parallel
// branches 1x1x1 and 3x3x3 with dilation 2 / 4 / 6 on the coarsest level, forward and input gradient.
//
// The LDS halo tiling of conv_igemm.hpp does not work here: at dilation 6 the halo of a 4x4x16 tile is larger than the
// whole 16^3 volume.  But that volume (3 MB per sample at 384 channels) lives in L2, so the B operand of the MFMA is
// gathered straight from global memory: for v_mfma_f32_32x32x16_bf16 lane (col = l & 31, h = l >> 5) holds 8
// consecutive channels of ONE voxel = one 16-byte buffer_load per lane; a tap that falls outside the volume gets an
// out-of-range offset and the descriptor's range check returns zeros (= the convolution's zero padding).  No im2col
// buffer (27x the activation), no LDS.  The 32x32 MFMA shape halves the operand bytes per MAC against 16x16 tiles,
// which matters because every operand comes through L1/L2.
//
// One launch runs up to 4 JOBS (grid.z); a job's output is the sum of up to 4 TERMS (input slice, kernel size, dilation,
// packed weights):  forward = 4 jobs x 1 term (each branch writes its channel slice of the concat buffer, torch.cat at
// :187 removed), input gradient = 1 job x 4 terms (the four branch gradients are summed in the accumulators).
//   D[row][voxel] = bias[row] + sum_term sum_tap sum_c Wt[tap][row][c] * Xt[voxel + dil_t * off(tap)][c]
// Workgroup = 4 waves, wave = NF row-fragments (32 output channels each) x NB voxel-fragments (32 voxels each);
// the (tap, 16-channel step) loop is flattened and software-pipelined three steps deep.
// f32 (parity mode): v_mfma_f32_32x32x2_f32, a 16-byte load = 4 channels = 4 MFMA k-steps.
#include "twin_begin.hpp"
#include "common.hpp"

typedef __attribute__((ext_vector_type(16))) float f32x16;

struct DconvTerm { const void* x; const void* w; int xpitch, cin, ksize, dil; };
struct DconvJob { DconvTerm term[4]; int nterms, rows; const float* bias; void* y; int ypitch, reserved; };
static_assert(sizeof(DconvTerm) == sizeof(brats_dconv_term) && sizeof(DconvJob) == sizeof(brats_dconv_job), "ABI structs");
struct DconvParams { DconvJob job[4]; int N, D, H, W; };

template <typename T> struct DcT;
template <> struct DcT<bf16_t> { static constexpr int KCH = 16; typedef bf16x8 frag; };  // channels per pipeline step
template <> struct DcT<float> { static constexpr int KCH = 8; typedef f32x4 frag; };

template <typename T, int NF, int NB>
__global__ __launch_bounds__(256, 1) void dconv_kernel(const DconvParams p) {
  using frag = typename DcT<T>::frag;
  constexpr int KCH = DcT<T>::KCH, ESZ = sizeof(T), PD = 3;
  const DconvJob& J = p.job[blockIdx.z];
  const int RF = (J.rows + 31) / 32;
  const int rf0 = blockIdx.y * NF;
  if (rf0 >= RF) return;  // (jobs of one launch may differ in their row count)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 31, h = lane >> 5;
  const int HW = p.H * p.W, DHW = p.D * HW;
  const long V = (long)p.N * DHW;
  const long vbase = ((long)blockIdx.x * 4 + wave) * (32 * NB);
  if (vbase >= V) return;

  int vz[NB], vy[NB], vx[NB];
  long vv[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    vv[b] = vbase + 32 * b + col;
    const int r = (int)(vv[b] % DHW);
    vz[b] = r / HW; vy[b] = (r / p.W) % p.H; vx[b] = r % p.W;
    if (vv[b] >= V) vz[b] = -100000;  // fails every range test: all taps read zeros
  }

  f32x16 acc[NF][NB];
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[f][b][r] = 0.f;

  for (int t = 0; t < J.nterms; ++t) {
    const DconvTerm& Tm = J.term[t];
    const int taps = Tm.ksize == 3 ? 27 : 1;
    const int KC = Tm.cin / KCH;
    const int S = taps * KC;
    const int pb = Tm.xpitch * ESZ;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)Tm.x, (short)0, (int)(V * pb), 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)Tm.w, (short)0, (int)((long)S * RF * 1024), 0x00020000);
    // per voxel-fragment: which taps stay inside the volume (bit tap), and the lane's own byte offset
    unsigned mask[NB];
    int lbase[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      unsigned m = 0;
      if (Tm.ksize == 3) {
        for (int tap = 0; tap < 27; ++tap) {
          const int z = vz[b] + (tap / 9 - 1) * Tm.dil, y = vy[b] + ((tap / 3) % 3 - 1) * Tm.dil, x = vx[b] + (tap % 3 - 1) * Tm.dil;
          m |= ((unsigned)z < (unsigned)p.D && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W ? 1u : 0u) << tap;
        }
      } else {
        m = vz[b] >= 0 ? 1u : 0u;
      }
      mask[b] = m;
      lbase[b] = (int)(vv[b] * pb) + h * 16;
    }
    auto tapoff = [&](int tap) {  // scalar: byte offset of the tap's shifted voxel
      if (Tm.ksize != 3) return 0;
      return (((tap / 9 - 1) * p.H + ((tap / 3) % 3 - 1)) * p.W + (tap % 3 - 1)) * Tm.dil * pb;
    };

    frag A[PD][NF], B[PD][NB];
    int ps = 0, ptap = 0, pkc = 0, psoff = tapoff(0);  // the prefetch stream's position
    // Every load is unconditional; what must not be read gets the offset 0xffffffff (OR with an all-ones mask), which the
    // range check drops.  The scalar masks go through an empty asm so that hipcc does not specialise the loop into
    // branches around each load (it does for a plain `cond ? offset : -1` on a wave-uniform condition).
    int fdead[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      fdead[f] = __builtin_amdgcn_readfirstlane(rf0 + f < RF ? 0 : -1);
      asm volatile("" : "+s"(fdead[f]));
    }
    auto issue = [&](auto st_) {
      constexpr int st = st_;
      int dead = __builtin_amdgcn_readfirstlane(ps < S ? 0 : -1);  // scalar
      asm volatile("" : "+s"(dead));
      const int wo = (ps * RF + rf0) * 1024 + lane * 16;
#pragma unroll
      for (int f = 0; f < NF; ++f)
        A[st][f] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(wrs, (wo + f * 1024) | dead | fdead[f], 0, 0));
      const int so = psoff + pkc * 32;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int inside = -(int)((mask[b] >> ptap) & 1u);  // all ones when the tap is inside the volume
        B[st][b] = __builtin_bit_cast(frag, __builtin_amdgcn_raw_buffer_load_b128(xrs, (lbase[b] + so) | ~inside | dead, 0, 0));
      }
      ++ps;
      if (++pkc == KC) { pkc = 0; ++ptap; psoff = tapoff(ptap); }
    };
    auto mma = [&](auto st_) {
      constexpr int st = st_;
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          if constexpr (std::is_same<T, bf16_t>::value) {
            acc[f][b] = MFMA16_32x32x16(A[st][f], B[st][b], acc[f][b]);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[f][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[st][f][j], B[st][b][j], acc[f][b], 0, 0, 0);
          }
        }
    };
    static_for<0, PD>([&](auto st_) { issue(st_); });
    for (int s = 0; s < S; s += PD) {
      static_for<0, PD>([&](auto st_) {
        mma(st_);     // steps beyond S multiply zeros (their loads were dropped by the range check)
        issue(st_);
      });
    }
  }

  // ---- epilogue: lane = voxel `col` of each fragment, rows (reg & 3) + 8 * (reg >> 2) + 4 * h ----
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    if (vv[b] >= V) continue;
    T* yrow = (T*)J.y + vv[b] * J.ypitch;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int r0 = (rf0 + f) * 32 + 8 * g + 4 * h;
        if (r0 < J.rows) {
          float o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = acc[f][b][4 * g + r] + (J.bias ? J.bias[r0 + r] : 0.f);
          Vec<T, 4>::store(yrow + r0, o);
        }
      }
    }
  }
}

// ---- weight packing: out[tap][kstep][row32][lane][16 B] ------------------------------------------------------------
// bf16: lane (row = l & 31, h = l >> 5) holds channels kstep*16 + 8h + e (e < 8); f32: channels kstep*8 + 4h + j (j < 4).
template <typename T>
__global__ void dconv_pack_kernel(const float* __restrict__ w, T* __restrict__ out, int mode, int taps, int cin_w, int cin_off,
                                  int rows, int RF, int kdim, int KC, size_t total) {
  constexpr int EPL = 16 / sizeof(T), KCH = 2 * EPL;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int e = idx % EPL;
  size_t t = idx / EPL;
  const int lane = t % 64; t /= 64;
  const int rf = t % RF; t /= RF;
  const int kc = t % KC;
  const int tap = (int)(t / KC);
  const int row = rf * 32 + (lane & 31), ch = kc * KCH + EPL * (lane >> 5) + e;
  float val = 0.f;
  if (row < rows && ch < kdim) {
    if (mode == BRATS_PACK_FWD) val = w[((size_t)row * cin_w + cin_off + ch) * taps + tap];
    else val = w[((size_t)ch * cin_w + cin_off + row) * taps + (taps - 1 - tap)];
  }
  out[idx] = from_f<T>(val);
}

static int dconv_kch(int dtype) { return dtype == BRATS_BF16 ? 16 : 8; }

extern "C" size_t BRATS_API(brats_dconv_packed_bytes)(int dtype, int ksize, int kdim, int rows) {
  const int kch = dconv_kch(dtype);
  if (kdim <= 0 || kdim % kch || rows <= 0 || (ksize != 1 && ksize != 3)) return 0;
  return (size_t)ksize * ksize * ksize * (kdim / kch) * ceil_div(rows, 32) * 1024;
}

extern "C" int BRATS_API(brats_dconv_pack_weights)(const float* w, void* packed, int dtype, int mode, int ksize, int cout_w, int cin_w,
                                        int cin_off, int cin_cnt, brats_stream_t s) {
  if (!w || !packed || (ksize != 1 && ksize != 3)) BRATS_FAIL(BRATS_E_ARG, "dconv_pack_weights: bad argument");
  const int taps = ksize * ksize * ksize;
  const int rows = mode == BRATS_PACK_FWD ? cout_w : cin_cnt;
  const int kdim = mode == BRATS_PACK_FWD ? cin_cnt : cout_w;
  const int kch = dconv_kch(dtype);
  if (kdim % kch) BRATS_FAIL(BRATS_E_UNSUPPORTED, "dconv_pack_weights: K channels %d not a multiple of %d", kdim, kch);
  const int RF = ceil_div(rows, 32), KC = kdim / kch;
  const size_t total = (size_t)taps * KC * RF * 64 * (dtype == BRATS_BF16 ? 8 : 4);
  const int blocks = (int)((total + 255) / 256);
  if (dtype == BRATS_BF16)
    hipLaunchKernelGGL(dconv_pack_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)s, w, (bf16_t*)packed, mode, taps, cin_w,
                       cin_off, rows, RF, kdim, KC, total);
  else
    hipLaunchKernelGGL(dconv_pack_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)s, w, (float*)packed, mode, taps, cin_w,
                       cin_off, rows, RF, kdim, KC, total);
  BRATS_CHECK_LAUNCH();
  return 0;
}

extern "C" int BRATS_API(brats_dconv_run)(const brats_dconv_job* jobs, int njobs, int dtype, int N, int D, int H, int W, brats_stream_t s) {
  if (!jobs || njobs < 1 || njobs > 4 || N <= 0 || D <= 0 || H <= 0 || W <= 0) BRATS_FAIL(BRATS_E_ARG, "dconv_run: bad argument");
  if (dtype != BRATS_BF16 && dtype != BRATS_F32) BRATS_FAIL(BRATS_E_UNSUPPORTED, "dconv_run: dtype %d", dtype);
  const int kch = dconv_kch(dtype), esz = dtype == BRATS_BF16 ? 2 : 4, align = 16 / esz;
  DconvParams p;
  p.N = N; p.D = D; p.H = H; p.W = W;
  const double V = (double)N * D * H * W;
  int max_rf = 0;
  for (int j = 0; j < njobs; ++j) {
    const brats_dconv_job& src = jobs[j];
    DconvJob& J = p.job[j];
    if (src.nterms < 1 || src.nterms > 4 || src.rows <= 0 || src.rows % 4 || !src.y || src.ypitch % 4 || src.ypitch < src.rows)
      BRATS_FAIL(BRATS_E_ARG, "dconv_run: job %d: bad rows / output", j);
    J.nterms = src.nterms; J.rows = src.rows; J.bias = src.bias; J.y = src.y; J.ypitch = src.ypitch; J.reserved = 0;
    for (int t = 0; t < src.nterms; ++t) {
      const brats_dconv_term& tt = src.term[t];
      if (!tt.x || !tt.w || tt.cin <= 0 || tt.cin % kch || tt.xpitch % align || tt.xpitch < tt.cin || (tt.ksize != 1 && tt.ksize != 3) || tt.dil < 1)
        BRATS_FAIL(BRATS_E_ARG, "dconv_run: job %d term %d: bad input description (channels must be multiples of %d)", j, t, kch);
      if (V * tt.xpitch * esz >= 2147483648.0)
        BRATS_FAIL(BRATS_E_UNSUPPORTED, "dconv_run: input of %g voxels x pitch %d exceeds the 2 GiB buffer-offset range", V, tt.xpitch);
      J.term[t].x = tt.x; J.term[t].w = tt.w; J.term[t].xpitch = tt.xpitch; J.term[t].cin = tt.cin;
      J.term[t].ksize = tt.ksize; J.term[t].dil = tt.dil;
    }
    const int rf = ceil_div(src.rows, 32);
    if (rf > max_rf) max_rf = rf;
  }
  constexpr int NF = 3, NB = 2;
  dim3 grid((unsigned)ceil_div((int)((V + 32 * NB - 1) / (32 * NB)), 4), (unsigned)ceil_div(max_rf, NF), (unsigned)njobs);
  if (dtype == BRATS_BF16) hipLaunchKernelGGL((dconv_kernel<bf16_t, NF, NB>), grid, dim3(256), 0, (hipStream_t)s, p);
  else hipLaunchKernelGGL((dconv_kernel<float, NF, NB>), grid, dim3(256), 0, (hipStream_t)s, p);
  BRATS_CHECK_LAUNCH();
  return 0;
}
#include "twin_end.hpp"
