// Implicit-GEMM 3x3x3 / 1x1x1 convolution for gfx950 (MFMA), NDHWC activations.
//
// Replaces nn.Conv3d of the reference (networks/equiunet2020.py:19-25, networks/equiunet2021.py:197-206)
// and, with weights packed in DGRAD mode, its autograd input-gradient.
//
// GEMM view: D[cout][voxel] = sum_{tap, cin} Wp[cout][(tap,cin)] * X[voxel + off(tap)][cin]
//   A operand (MFMA rows)  = packed weights, streamed from L2 in fragment order (16 B / lane / step)
//   B operand (MFMA cols)  = activations, read from an LDS halo tile (one ds_read_b128 per fragment)
//   C/D: lane (q = lane>>4, v = lane&15) holds cout rows 4q..4q+3 of voxel v -> 4 contiguous
//        NDHWC channels per lane = one 8/16-byte store.
// Workgroup = 256 threads = 4 waves on a 4x4x16 (z,y,x) voxel tile: wave (wm, wn); wm picks 2 z-slices
// (128 voxels = 8 voxel-fragments, each one x-row of 16), wn picks either the cout half (tile = 2*NF*16 couts) or, when
// the layer has too few couts (KSPLIT), the parity of the K macro-steps (reduced through LDS).
// Per Cin chunk (CK channels) the halo tile is staged once and all taps read it (27x reuse from
// LDS; HBM/L2 sees only the ~2x halo amplification).  2 workgroups/CU overlap staging with MFMA.
// Relatives: conv_igemm_vs8.hpp (y-split roles on a 4x8x16 tile, the default for bf16 layers with 48 mod 96 couts),
// conv_igemm_f8.hpp (e4m3 operands), conv_igemm_x3.hpp (f32 tensors, three 16-bit MFMA products).  (Experiments that did not
// pay -- two persistent forms, the loader-wave kernel, three workgroups per CU -- live in scripts/probes/experiments/.)
#pragma once
#include <stdlib.h>
#include "common.hpp"

struct ConvParams {
  const void* x1; const void* x2; int c1, c2, p1, p2;
  const void* wpk; const float* bias; void* y; int ypitch; float* stats;
  void* y2; int y2pitch; int ysplit;  // optional second destination for output channels >= ysplit (dgrad of a concat input)
  int N, D, H, W, cout, rows16, nchunks;
  int tz, ty, tx;
  const float* xamax;  // split-precision kernels only (conv_igemm_x3.hpp): device scalar max|x| -> input scale; NULL = 1
  // PRE kernels only ("normalise + activate on load", inference): per-(sample, channel) {scale, shift} of source 1 / 2 (NULL =
  // that source is read as it is), applied to every staged value as act(x * scale + shift) before it enters LDS
  const float* ss1; const float* ss2; int pre_act; float pre_slope;
  // BST kernels only ("backward statistics", training): this launch is the input gradient of the SECOND convolution of a
  // block, its output dz the gradient of the first unit's activation.  by = that unit's raw convolution output (the dz
  // shape, pitch bypitch), bss its {scale, shift} [N][cout][2]; the epilogue adds u = dz * act'(by * scale + shift) and
  // u * by per tile and channel into `stats` -- GroupNorm backward's first pass (brats_gn_act_bwd: sum u, sum u * xhat)
  // without reading dz and by back from HBM.  bslope: 0 = relu, else leakyrelu's slope
  const void* by; int bypitch; const float* bss; float bslope;
#ifdef BRATS_VS8_STAMPS
  long long* stamps;  // diagnostic build only
#endif
};

constexpr int CONV_TZ = 4, CONV_TY = 4, CONV_TX = 16;  // a voxel fragment = one x-row of 16

template <typename T, int KS, int CK, int DIL, int TY = CONV_TY /* tile rows in y: 8 for conv_igemm_vs8.hpp */>
struct ConvGeom {
  static constexpr bool BF = std::is_same<T, bf16_t>::value;
  static constexpr int ESZ = sizeof(T);
  static constexpr int EPL = 16 / ESZ;  // elements per 16-byte lane fragment
  static constexpr int R = (KS == 3) ? DIL : 0;
  static constexpr int HZ = CONV_TZ + 2 * R, HY = TY + 2 * R, HX = CONV_TX + 2 * R;
  static constexpr int HVOX = HZ * HY * HX;
  static constexpr int ROWB = CK * ESZ;
  static constexpr int PPV = ROWB / 16;
  // voxel stride in LDS.  bf16: a fragment's 16 voxels are 16 consecutive x, so ds_read_b128 is
  // bank-conflict-free iff (S/16) % 4 == 2 (32, 96, 160 B ...; CK=48 -> 96 B = no padding at all).
  // f32 (ds_read_b32 operands, parity mode): odd number of 16-B slots (2-way at worst).
  // (ds_read_b128 serves 8 lanes per cycle, so any odd number of 16-byte slots is conflict-free as well -- measured
  // with the 48-byte stride of the e4m3 kernel; only the 24-channel chunks of conv_igemm_vs8.hpp use that here, the
  // tuned kernels keep their strides)
  static constexpr int S = (BF && PPV == 3) ? 48 : BF ? 16 * (PPV + ((2 - PPV % 4) + 4) % 4) : ((PPV % 2 == 0) ? ROWB + 16 : ROWB);
  static constexpr int NPIECE = HVOX * PPV;
  static constexpr int NITER = (NPIECE + 255) / 256;
  static constexpr int TAPS = KS * KS * KS;
  // bf16: one macro-step = one K=32 MFMA = 4 "units" of 8 channels (one per lane quarter)
  // f32 : one macro-step = four K=4 MFMAs; a K=4 step = 4 units of 1 channel
  static constexpr int UPT = BF ? CK / 8 : CK;           // units per tap
  static constexpr int UNITS = TAPS * UPT;
  static constexpr int MS = BF ? (UNITS + 3) / 4 : (UNITS / 4 + 3) / 4;
  static constexpr int UB = BF ? 16 : 4;                 // bytes per unit
  static constexpr int LDS_TILE = HVOX * S;
  static_assert(ROWB % 16 == 0, "CK*sizeof(T) must be a multiple of 16");
  static_assert(BF || CK % 4 == 0, "f32 CK must be a multiple of 4");

  static constexpr int tapoff(int tap) {
    return KS == 1 ? 0 : ((((tap / 9) * DIL) * HY + ((tap / 3) % 3) * DIL) * HX + (tap % 3) * DIL) * S;
  }
  // byte offset (relative to the lane's voxel) of unit g; invalid units read offset 0 (weights are 0)
  static constexpr int unitoff(int g) { return g < UNITS ? tapoff(g / UPT) + (g % UPT) * UB : 0; }
};

// wave roles inside a workgroup: wm always picks the z half of the tile; wn picks
//   !KSPLIT && !VS : the cout half (tile = 2*NF*16 couts, 8 voxel fragments per wave)
//   KSPLIT         : the parity of the K macro-steps (same couts and voxels, reduced through LDS)
//   VS             : the y half (same couts, 4 voxel fragments per wave, all K: no reduction, all four waves share
//                    the epilogue; the weight fragments are fetched twice as often)
//   VS + KP        : (round 6) the y-split roles with EIGHT waves, for grids of at most one workgroup per CU (the 16^3 level):
//                    two teams of four waves with the VS roles; team 0 takes the even K macro-steps of every chunk, team 1 the
//                    odd ones, on the same LDS tile and the same output tile; team 1's partial sums reach team 0 through LDS
//                    (aliasing the dead tile) before the epilogue.  Two waves per SIMD instead of a lone one, halo staging by
//                    eight waves, no HBM partial-sum pass.  What it buys is small (0 .. 14 % by shape) and DESIGN.md says why:
//                    at one workgroup per CU these roles are bound by the weight-fragment stream through the CU's vector-memory
//                    path (3 KB per 12 MFMAs and wave ~ 64 B/clk/CU asked, ~32 delivered), which eight waves do not change.
template <int NF, bool KSPLIT, bool VS = false, bool KP = false> struct ConvTile {
  static_assert(!(KSPLIT && VS), "one split mode at a time");
  static_assert(!KP || VS, "the 8-wave K-parity teams sit on the y-split roles");
  static constexpr int NFW = (KSPLIT || VS) ? NF : 2 * NF;  // cout16-fragments per workgroup
  static constexpr int NB = VS ? 4 : 8;                     // voxel fragments (x-rows of 16) per wave
  static constexpr int NW = KP ? 8 : 4;                     // waves per workgroup
  static constexpr int RED_BYTES = KSPLIT ? 2 * NF * 8 * 64 * 16 : KP ? 4 * NF * NB * 64 * 16 : 0;
  static constexpr int SRED_BYTES = (VS ? 4 : 2) * NFW * 16 * 2 * 4;
};

template <typename T, int KS, int CK, int DIL, int NF, bool KSPLIT, bool VS = false, bool KP = false>
constexpr int conv_lds_bytes() {
  using G = ConvGeom<T, KS, CK, DIL>;
  using TL = ConvTile<NF, KSPLIT, VS, KP>;
  int a = G::LDS_TILE > TL::RED_BYTES ? G::LDS_TILE : TL::RED_BYTES;
  a = (a + 15) / 16 * 16;
  return a + TL::SRED_BYTES;
}

// One Cin chunk of MFMA work, software-pipelined by hand (hipcc does not do it at this register
// pressure): the weight fragments of step k+1 are requested from L2 before the MFMAs of step k, and the
// 8 activation fragments are read from LDS in two halves so that 4 ds_read_b128 are always in flight
// behind 12 MFMAs.
template <typename T, int KS, int CK, int DIL, int NF, int PARITY /* -1: all steps */,
          int NB = 8 /* voxel fragments per wave: NB/2 y-rows in each of 2 z-slices */,
          typename GEOM = ConvGeom<T, KS, CK, DIL>>
DEVI void conv_mma_chunk(const char* ldsb, int lane_b, int q, const void* wpk_chunk, int rows16, int f0,
                         int lane, f32x4 (&acc)[NF][NB]) {
  constexpr int YB = NB / 2;
  using G = GEOM;
  constexpr int FOZ = G::HY * G::HX * G::S;  // one z-slice
  constexpr int NSTEP = PARITY < 0 ? G::MS : (G::MS - PARITY + 1) / 2;
  if constexpr (G::BF) {
    const bf16x8* wp0 = (const bf16x8*)wpk_chunk + (size_t)f0 * 64 + lane;
    constexpr int WD = 2;  // weight prefetch distance in macro-steps (L2 latency under load > one step of 24 MFMAs)
    bf16x8 a[WD + 1][NF];
    bf16x8 b[NB];
    auto load_a = [&](auto k_) {
      constexpr int k = k_;
      constexpr int ms = PARITY < 0 ? k : 2 * k + PARITY;
#pragma unroll
      for (int f = 0; f < NF; ++f) a[k % (WD + 1)][f] = wp0[((size_t)ms * rows16 + f) * 64];
    };
    auto read_b = [&](auto k_, auto half_) {
      constexpr int k = k_, half = half_;
      constexpr int ms = PARITY < 0 ? k : 2 * k + PARITY;
      constexpr int o0 = G::unitoff(4 * ms), o1 = G::unitoff(4 * ms + 1), o2 = G::unitoff(4 * ms + 2),
                    o3 = G::unitoff(4 * ms + 3);
      int lb;
      if constexpr (o1 - o0 == G::UB && o2 - o0 == 2 * G::UB && o3 - o0 == 3 * G::UB) {
        lb = lane_b + o0;  // lane_b already carries q*UB
      } else {
        lb = lane_b + (q == 0 ? o0 : q == 1 ? o1 - G::UB : q == 2 ? o2 - 2 * G::UB : o3 - 3 * G::UB);
      }
#pragma unroll
      for (int i = YB * half; i < YB * half + YB; ++i)
        b[i] = *(const bf16x8*)(ldsb + lb + ((i / YB) * FOZ + (i % YB) * G::HX * G::S));
    };
    auto mma = [&](auto k_, auto half_) {
      constexpr int k = k_, half = half_;
#pragma unroll
      for (int i = YB * half; i < YB * half + YB; ++i)
#pragma unroll
        for (int f = 0; f < NF; ++f)
          acc[f][i] = MFMA16_16x16x32(a[k % (WD + 1)][f], b[i], acc[f][i]);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    static_for<0, (WD < NSTEP ? WD : NSTEP)>([&](auto k_) { load_a(k_); });
    read_b(I0{}, I0{});
    static_for<0, NSTEP>([&](auto k_) {
      constexpr int k = k_;
      // sched_barrier(0) pins the issue order: without it hipcc sinks every load to just before its
      // first use (one live B fragment, weights waited for at vmcnt(0)) and the loop runs latency-bound
      if constexpr (k + WD < NSTEP) load_a(std::integral_constant<int, k + WD>{});
      read_b(k_, I1{});
      __builtin_amdgcn_sched_barrier(0);
      mma(k_, I0{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (k + 1 < NSTEP) read_b(std::integral_constant<int, k + 1>{}, I0{});
      __builtin_amdgcn_sched_barrier(0);
      mma(k_, I1{});
      __builtin_amdgcn_sched_barrier(0);
    });
  } else {
    static_for<0, NSTEP>([&](auto k_) {
      constexpr int k = k_;
      constexpr int ms = PARITY < 0 ? k : 2 * k + PARITY;
      const f32x4* wp = (const f32x4*)wpk_chunk + ((size_t)ms * rows16 + f0) * 64 + lane;
      f32x4 a[NF];
#pragma unroll
      for (int f = 0; f < NF; ++f) a[f] = wp[f * 64];
      static_for<0, 4>([&](auto j_) {
        constexpr int j = j_;
        constexpr int o = G::unitoff(4 * (4 * ms + j));  // CK%4==0: the 4 quarters share the tap
        if constexpr (4 * (4 * ms + j) < G::UNITS) {
#pragma unroll
          for (int i = 0; i < NB; ++i) {
            const float bb = *(const float*)(ldsb + lane_b + o + ((i / YB) * FOZ + (i % YB) * G::HX * G::S));
#pragma unroll
            for (int f = 0; f < NF; ++f) acc[f][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[f][j], bb, acc[f][i], 0, 0, 0);
          }
        }
      });
    });
  }
}

// 16-lane (one MFMA row group) all-reduce with DPP row rotations: 4 VALU ops, no LDS crossbar.
DEVI float row16_sum(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x128, 0xf, 0xf, false));  // row_ror:8
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x124, 0xf, 0xf, false));  // row_ror:4
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x122, 0xf, 0xf, false));  // row_ror:2
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x121, 0xf, 0xf, false));  // row_ror:1
  return x;
}

// ---- epilogue statistics of one output piece (the lane's 4 channels of one voxel) ------------------------------------------
// Scalar on purpose: packed f32 instructions (v_pk_add_f32 / v_pk_fma_f32: half the instruction count) were built and
// measured in round 4 -- the 4x8x16-tile kernel 1-2 % faster in isolation, the step 0.05-0.1 ms slower: beside the partner
// workgroup's MFMA stream a packed f32 instruction costs more than the two scalar ones it replaces (MI355X_MICROARCH.md,
// "packed f32 VALU ... an anti-lever beside MFMAs"; the reason the library is built with -fno-slp-vectorize).
template <bool MASKED>
DEVI void stat_fwd(float (&s1)[4], float (&s2)[4], const float (&o)[4], float m) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if constexpr (MASKED) {
      const float om = o[r] * m;
      s1[r] += om;
      s2[r] = __builtin_fmaf(om, o[r], s2[r]);
    } else {
      s1[r] += o[r];
      s2[r] = __builtin_fmaf(o[r], o[r], s2[r]);
    }
  }
}
// backward statistics (ConvParams::by / bss): u = dz * act'(y * scale + shift), s1 += u, s2 += u * y for the lane's four
// channels; yraw = their forward values (4 x 16 bit); slope 0 = relu
template <bool MASKED>
DEVI void stat_bst(float (&s1)[4], float (&s2)[4], const float (&o)[4], u32x2 yraw, const float (&sc)[4], const float (&sh)[4],
                   float slope, float m) {
  float yy[4];
  unpack2(yraw[0], yy[0], yy[1]);
  unpack2(yraw[1], yy[2], yy[3]);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float pre = __builtin_fmaf(yy[r], sc[r], sh[r]);
    float u = pre > 0.f ? o[r] : o[r] * slope;
    if constexpr (MASKED) u *= m;
    s1[r] += u;
    s2[r] = __builtin_fmaf(u, yy[r], s2[r]);
  }
}

// statistics of a 4x8x16 tile: sred[wm + 2 wn][NF*16][2] -> one entry per 4x4x16 sub-tile (y-half wn), summed over wm
template <int NF>
DEVI void vs8_stats_write(const ConvParams& p, int ty4, const float* sred, int tid, int n, int tzi, int tyi, int txi, int ct) {
  if (tid < 2 * NF * 16) {
    const int half = tid / (NF * 16), cl = tid % (NF * 16);
    const int c = ct * NF * 16 + cl;
    const int ty_i = tyi * 2 + half;
    if (c < p.cout && ty_i < ty4) {
      const size_t tps = (size_t)p.tz * ty4 * p.tx;
      const size_t tile = ((size_t)tzi * ty4 + ty_i) * p.tx + txi;
      float* dst = p.stats + (((size_t)n * tps + tile) * p.cout + c) * 2;
      dst[0] = sred[((2 * half) * NF * 16 + cl) * 2] + sred[((2 * half + 1) * NF * 16 + cl) * 2];
      dst[1] = sred[((2 * half) * NF * 16 + cl) * 2 + 1] + sred[((2 * half + 1) * NF * 16 + cl) * 2 + 1];
    }
  }
}

// ---- normalise + activate on load (PRE kernels) ------------------------------------------------------------------------
// Under no_grad nobody but the next convolution reads z = act(GroupNorm(y)) of a layer inside a block: the consumer then
// stages the RAW convolution output y and applies the producer's per-(sample, channel) affine map and activation between the
// global load and the LDS write -- z is never stored, the affine_act pass (a read + a write of the whole tensor) is gone.
// Voxels outside the volume must enter LDS as ZERO (the padding of z, not act(shift)): `ok`.  The arithmetic is
// affine_act_kernel's (norm.hip: multiply, add, max / select in f32, one rounding to the 16-bit type), so the staged values
// are bit-identical to the stored z.  relu / leakyrelu only (the published configurations).
DEVI u32x4 conv_pre_apply(u32x4 v, const float (&sc)[8], const float (&sh)[8], bool leaky, float slope, bool ok) {
  const uint32_t w[4] = {v[0], v[1], v[2], v[3]};  // (through scalars: see f8_quant8)
  uint32_t o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float a, b;
    unpack2(w[i], a, b);
    a = a * sc[2 * i] + sh[2 * i];
    b = b * sc[2 * i + 1] + sh[2 * i + 1];
    if (leaky) {
      a = a > 0.f ? a : a * slope;
      b = b > 0.f ? b : b * slope;
    } else {
      a = __builtin_fmaxf(a, 0.f);
      b = __builtin_fmaxf(b, 0.f);
    }
    o[i] = ok ? pack2(a, b) : 0u;
  }
  return u32x4{o[0], o[1], o[2], o[3]};
}
// the 8 {scale, shift} pairs of the piece a lane stages: channels cb + 8 part .. + 7 of sample n (ss = [N][C][2] f32)
DEVI void conv_pre_load(const float* ss, int n, int csrc, int cb, int part, float (&sc)[8], float (&sh)[8]) {
  const f32x4* q = (const f32x4*)(ss + ((size_t)n * csrc + cb + part * 8) * 2);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x4 t = q[i];
    sc[2 * i] = t[0]; sh[2 * i] = t[1]; sc[2 * i + 1] = t[2]; sh[2 * i + 1] = t[3];
  }
}

// BST: backward statistics in the epilogue (ConvParams::by / bss; training, the input gradient of a block's second unit)
template <typename T, int KS, int CK, int DIL, int NF, bool KSPLIT, bool VS = false, bool PRE = false, bool BST = false, bool KP = false>
__global__ __launch_bounds__(KP ? 512 : 256, KP ? 1 : 2) void conv_igemm_kernel(const ConvParams p) {
  static_assert(!PRE || std::is_same<T, bf16_t>::value, "normalise-on-load exists for the 16-bit kernels");
  static_assert(!BST || (std::is_same<T, bf16_t>::value && !KSPLIT), "backward statistics exist for the 16-bit kernels without K split");
  using G = ConvGeom<T, KS, CK, DIL>;
  using TL = ConvTile<NF, KSPLIT, VS, KP>;
  constexpr int NB = TL::NB, YB = NB / 2, NW = TL::NW;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform -> scalar row math
  const int team = KP ? wave >> 2 : 0;                        // KP: the K-parity team; the roles below are per team
  const int wm = wave & 1, wn = (wave >> 1) & 1;
  const int q = lane >> 4, v = lane & 15;

  int bt = blockIdx.x;
  const int tile_in_sample = bt % (p.tz * p.ty * p.tx);
  const int txi = bt % p.tx; bt /= p.tx;
  const int tyi = bt % p.ty; bt /= p.ty;
  const int tzi = bt % p.tz;
  const int n = bt / p.tz;
  const int z0 = tzi * CONV_TZ, y0 = tyi * CONV_TY, x0 = txi * CONV_TX;
  const int ct = blockIdx.y;
  const int f0 = ct * TL::NFW + ((KSPLIT || VS) ? 0 : wn * NF);
  const size_t sample_vox = (size_t)n * p.D * p.H * p.W;

  // --- staging: wave w owns halo rows (hz,hy) = w, w+4, ...; a row is HX voxels x PPV 16-byte pieces,
  //     fetched by IPR wave-instructions.  Row origin / bounds are scalar; the per-lane part (which
  //     piece of the row, is its x inside the volume) is computed once per kernel.
  constexpr int NROWS = G::HZ * G::HY;
  constexpr int PPR = G::HX * G::PPV;          // pieces per row
  constexpr int IPR = (PPR + 63) / 64;         // wave-instructions per row
  constexpr int RPW = (NROWS + NW - 1) / NW;   // rows per wave
  int lds_off[IPR];    // byte offset of the lane's piece inside an LDS row (+ the wave's first row)
  int hx_part[IPR];    // hx * 65536 + part, or -1 when the lane has no piece / x is outside the volume
#pragma unroll
  for (int j = 0; j < IPR; ++j) {
    const int pc = lane + 64 * j;
    const int hx = pc / G::PPV, part = pc % G::PPV;
    const int gx = x0 - G::R + hx;
    const bool ok = pc < PPR && gx >= 0 && gx < p.W;
    hx_part[j] = ok ? (hx << 16) | part : -1;
    lds_off[j] = pc < PPR ? wave * (G::HX * G::S) + hx * G::S + part * 16 : -1;
  }

  // accumulators start at the bias (zero without one; the second K-split partition adds to the first: zero): no bias add in
  // the epilogue, whose instruction count is what it costs beside the partner workgroup's MFMA stream (conv_igemm_vs8.hpp)
  f32x4 acc[NF][NB];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int cbase = (f0 + f) * 16 + 4 * q;
    f32x4 b = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && cbase < p.cout && !(KSPLIT && wn == 1) && !(KP && team == 1)) b = *(const f32x4*)(p.bias + cbase);
#pragma unroll
    for (int i = 0; i < NB; ++i) acc[f][i] = b;
  }

  // lane's voxel inside the halo tile (tap (0,0,0) corner) + quarter offset
  const int lane_b = ((wm * 2) * G::HY * G::HX + (VS ? wn * 2 * G::HX : 0) + v) * G::S + q * G::UB;
  const size_t chunk_stride = (size_t)G::MS * p.rows16 * 64 * 16;  // bytes of packed weights per chunk

#ifdef BRATS_VS8_STAMPS  // diagnostic build only (scripts/probes/stamps_build.sh + igemm_stamps.py): where does a workgroup's time go?
  long long tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#define IG_STAMP(i) do { const long long t_ = __builtin_amdgcn_s_memtime(); tacc[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define IG_STAMP(i) do { } while (0)
#endif
  for (int chunk = 0; chunk < p.nchunks; ++chunk) {
    const int c0 = chunk * CK;
    const T* src;
    int pitch;
    if (c0 < p.c1) { src = (const T*)p.x1 + c0; pitch = p.p1; }
    else { src = (const T*)p.x2 + (c0 - p.c1); pitch = p.p2; }
    // one unconditional buffer_load per piece: pieces outside the volume get an out-of-range offset, for which the
    // descriptor's range check returns zeros (no per-piece branch, no zero-initialisation of the staging registers)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(src + sample_vox * pitch), (short)0, (int)((size_t)p.D * p.H * p.W * pitch * G::ESZ), 0x00020000);
    const int pb = pitch * G::ESZ;
    int goff[IPR];  // byte offset of the lane's piece from the row origin (gx = x0 - R)
#pragma unroll
    for (int j = 0; j < IPR; ++j) goff[j] = (hx_part[j] >> 16) * pb + (hx_part[j] & 0xffff) * 16;
    u32x4 r[RPW][IPR];
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
      const int row = wave + NW * k;
      const int hz = row / G::HY, hy = row % G::HY;
      const int gz = z0 - G::R + hz, gy = y0 - G::R + hy;
      const bool row_ok = row < NROWS && gz >= 0 && gz < p.D && gy >= 0 && gy < p.H;  // scalar
      const int rb = ((gz * p.H + gy) * p.W + (x0 - G::R)) * pb;  // may be negative at the low faces; valid pieces are not
#pragma unroll
      for (int j = 0; j < IPR; ++j) {
        const int vo = (row_ok && hx_part[j] >= 0) ? rb + goff[j] : -1;
        r[k][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, 0));
      }
    }
    // PRE (normalise + activate on load): this chunk's {scale, shift} of the lane's channels, requested behind the halo loads
    float psc[PRE ? IPR : 1][8], psh[PRE ? IPR : 1][8];
    const float* pre_ss = nullptr;
    if constexpr (PRE) {
      pre_ss = c0 < p.c1 ? p.ss1 : p.ss2;  // scalar: NULL = this source is read as it is
      if (pre_ss) {
        const int csrc = c0 < p.c1 ? p.c1 : p.c2, cb = c0 < p.c1 ? c0 : c0 - p.c1;
#pragma unroll
        for (int j = 0; j < IPR; ++j) conv_pre_load(pre_ss, n, csrc, cb, hx_part[j] >= 0 ? hx_part[j] & 0xffff : 0, psc[j], psh[j]);
      }
    }
    IG_STAMP(0);  // address arithmetic + issue of the halo loads
    if (chunk > 0) __syncthreads();  // all waves finished reading the previous chunk's tile
    IG_STAMP(1);  // barrier: everybody done with the previous chunk
#ifdef BRATS_VS8_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    IG_STAMP(2);  // the halo loads landing
#endif
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
      if (wave + NW * k < NROWS) {
#pragma unroll
        for (int j = 0; j < IPR; ++j)
          if (lds_off[j] >= 0) {
            u32x4 val = r[k][j];
            if constexpr (PRE) {
              if (pre_ss) {  // piece by piece, right in front of its LDS write: the staged registers die as they are consumed
                const int row = wave + NW * k;
                const int gz = z0 - G::R + row / G::HY, gy = y0 - G::R + row % G::HY;
                const bool row_ok = gz >= 0 && gz < p.D && gy >= 0 && gy < p.H;
                val = conv_pre_apply(val, psc[j], psh[j], p.pre_act == BRATS_ACT_LEAKY, p.pre_slope, row_ok && hx_part[j] >= 0);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
            *(u32x4*)(lds + lds_off[j] + k * NW * (G::HX * G::S)) = val;
          }
      }
    }
    __syncthreads();
    IG_STAMP(3);  // LDS writes + barrier
    const char* wchunk = (const char*)p.wpk + chunk * chunk_stride;
    if constexpr (KSPLIT) {
      if (wn == 0) conv_mma_chunk<T, KS, CK, DIL, NF, 0>(lds, lane_b, q, wchunk, p.rows16, f0, lane, acc);
      else conv_mma_chunk<T, KS, CK, DIL, NF, 1>(lds, lane_b, q, wchunk, p.rows16, f0, lane, acc);
    } else if constexpr (KP) {
      if (team == 0) conv_mma_chunk<T, KS, CK, DIL, NF, 0, NB>(lds, lane_b, q, wchunk, p.rows16, f0, lane, acc);
      else conv_mma_chunk<T, KS, CK, DIL, NF, 1, NB>(lds, lane_b, q, wchunk, p.rows16, f0, lane, acc);
    } else {
      conv_mma_chunk<T, KS, CK, DIL, NF, -1, NB>(lds, lane_b, q, wchunk, p.rows16, f0, lane, acc);
    }
    IG_STAMP(4);  // MFMA loop
  }

  // --- K-split reduction through LDS ---
  if constexpr (KSPLIT) {
    __syncthreads();
    f32x4* red = (f32x4*)lds;
    if (wn == 1) {
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int i = 0; i < 8; ++i) red[((wm * NF + f) * 8 + i) * 64 + lane] = acc[f][i];
    }
    __syncthreads();
    if (wn == 0) {
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[f][i] += red[((wm * NF + f) * 8 + i) * 64 + lane];
    }
  }

  // --- K-parity teams: team 1's partial sums through LDS (the tile is dead) into team 0, which runs the epilogue ---
  if constexpr (KP) {
    __syncthreads();
    f32x4* red = (f32x4*)lds;
    const int wv = wave & 3;
    if (team == 1) {
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int i = 0; i < NB; ++i) red[((wv * NF + f) * NB + i) * 64 + lane] = acc[f][i];
    }
    __syncthreads();
    if (team == 0) {
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int i = 0; i < NB; ++i) acc[f][i] += red[((wv * NF + f) * NB + i) * 64 + lane];
    }
  }

  // --- epilogue: bias, per-channel tile statistics, NDHWC store.  A fragment is one x-row, so the row
  //     pointer / row bounds are scalar and each lane adds a fixed offset: no per-store address math.
  constexpr int LDS_MAIN = ((G::LDS_TILE > TL::RED_BYTES ? G::LDS_TILE : TL::RED_BYTES) + 15) / 16 * 16;
  float* sred = (float*)(lds + LDS_MAIN);  // [2 (wm)][NFW*16][2]
  const bool active = KSPLIT ? (wn == 0) : KP ? (team == 0) : true;
  if (active) {
    const bool x_ok = x0 + v < p.W;
    // dual destination (dgrad of a conv whose input was the virtual concat [x1 | x2]): channels >= ysplit
    // go to y2.  ysplit is a multiple of the wave's NF*16 channels, so the choice is per wave.
    const bool second = p.y2 != nullptr && f0 * 16 >= p.ysplit;
    T* const ydst = second ? (T*)p.y2 : (T*)p.y;
    const int ypit = second ? p.y2pitch : p.ypitch;
    const int csub = second ? p.ysplit : 0;
    const int lane_o = (x0 + v) * ypit + 4 * q - csub;  // elements from the row origin
    float s1[NF][4], s2[NF][4];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s1[f][r] = 0.f;
        s2[f][r] = 0.f;
      }
    }
    // BST: the forward values under this lane's outputs (4 channels per fragment and x-row, 8 bytes), requested
    // BRATS_BST_AHEAD x-rows ahead of their use (round 5: all of them at once), and the lane's channels' {scale, shift}; the tile sums become
    // sum u, sum u * (forward value) with u = dz * act'(forward * scale + shift)  (GroupNorm backward's first pass)
    u32x2 byv[BST ? NF : 1][BST ? NB : 1];
    float bsc[BST ? NF : 1][4], bsh[BST ? NF : 1][4];
    auto load_y = [&](int i, auto checked) {
      if constexpr (BST) {
        const int z = z0 + 2 * wm + (i / YB), y = y0 + (VS ? 2 * wn : 0) + (i % YB);
        const bf16_t* rowp = (const bf16_t*)p.by + (sample_vox + (size_t)(z * p.H + y) * p.W + (x0 + v)) * p.bypitch + 4 * q;
        const bool ok = !decltype(checked)::value || (z < p.D && y < p.H && x_ok);
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const bool cok = !decltype(checked)::value || (f0 + f) * 16 + 4 * q < p.cout;
          byv[f][i] = ok && cok ? *(const u32x2*)(rowp + (f0 + f) * 16) : u32x2{0u, 0u};
        }
      }
    };
    if constexpr (BST) {
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const int c = (f0 + f) * 16 + 4 * q;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool cok = c + r < p.cout;
          bsc[f][r] = cok ? p.bss[((size_t)n * p.cout + c + r) * 2] : 0.f;
          bsh[f][r] = cok ? p.bss[((size_t)n * p.cout + c + r) * 2 + 1] : 0.f;
        }
      }
    }
    // statistics of one output piece (4 channels of one voxel): sum x, sum x^2 -- or, BST, sum u, sum u * (forward value)
    auto tally = [&](int f, int i, const float (&o)[4], auto masked, float m) {
      constexpr bool MASKED = decltype(masked)::value;
      if constexpr (BST) stat_bst<MASKED>(s1[f], s2[f], o, byv[f][i], bsc[f], bsh[f], p.bslope, m);
      else stat_fwd<MASKED>(s1[f], s2[f], o, m);
    };
    // interior tiles (the common case) take the mask-free path; edge tiles weight the statistics by a
    // 0/1 mask instead of branching around the accumulation (a branch makes hipcc copy all 8*NF
    // running sums through v_mov at every row)
    const bool full = z0 + CONV_TZ <= p.D && y0 + CONV_TY <= p.H && x0 + CONV_TX <= p.W &&
                      (ct + 1) * TL::NFW * 16 <= p.cout;  // scalar
    if (full && G::BF && (ypit % 8) == 0 && (csub % 8) == 0 && ((size_t)ydst & 15) == 0) {
      // bf16: 16-byte stores (the epilogue is store-ISSUE bound).  The lanes of MFMA rows q and q ^ 1 hold channels
      // 4q..4q+3 and the next four of the SAME voxel; exchanging halves between two x-rows (v_permlane16_swap: odd 16-lane
      // rows of the first operand <-> even rows of the second) gives every lane 8 consecutive channels of ONE voxel:
      // rows 0 / 2 keep x-row i, rows 1 / 3 take x-row i + 1.
      const int lane_w = (x0 + v) * ypit + 8 * (q >> 1) - csub + (q & 1) * p.W * ypit;
#ifndef BRATS_BST_AHEAD
#define BRATS_BST_AHEAD 8
#endif
      constexpr int AH = BRATS_BST_AHEAD < NB ? BRATS_BST_AHEAD : NB;  // x-rows requested ahead of their use (even)
#pragma unroll
      for (int k = 0; k < AH; ++k) load_y(k, std::false_type{});
#pragma unroll
      for (int i = 0; i < NB; i += 2) {
        if (i + AH < NB) {
          load_y(i + AH, std::false_type{});
          load_y(i + AH + 1, std::false_type{});
        }
        if constexpr (BST) __builtin_amdgcn_sched_barrier(0);  // (the loads stay where they are written)
        const int z = z0 + 2 * wm + (i / YB), y = y0 + (VS ? 2 * wn : 0) + (i % YB);
        T* rowp = ydst + (sample_vox + (size_t)(z * p.H + y) * p.W) * ypit;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          uint32_t pk[2][2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = acc[f][i + e][r];
            tally(f, i + e, o, std::false_type{}, 1.f);
            pk[e][0] = pack2(o[0], o[1]);
            pk[e][1] = pack2(o[2], o[3]);
          }
          const u32x2 lo = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
          const u32x2 hi = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
          *(u32x4*)((bf16_t*)rowp + lane_w + (f0 + f) * 16) = u32x4{lo[0], hi[0], lo[1], hi[1]};
        }
      }
    } else if (full) {
      load_y(0, std::false_type{});
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        if (i + 1 < NB) load_y(i + 1, std::false_type{});
        const int z = z0 + 2 * wm + (i / YB), y = y0 + (VS ? 2 * wn : 0) + (i % YB);
        T* rowp = ydst + (sample_vox + (size_t)(z * p.H + y) * p.W) * ypit;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          float o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = acc[f][i][r];
          tally(f, i, o, std::false_type{}, 1.f);
          Vec<T, 4>::store(rowp + lane_o + (f0 + f) * 16, o);
        }
      }
    } else {
      load_y(0, std::true_type{});
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        if (i + 1 < NB) load_y(i + 1, std::true_type{});
        const int z = z0 + 2 * wm + (i / YB), y = y0 + (VS ? 2 * wn : 0) + (i % YB);
        const bool ok = z < p.D && y < p.H && x_ok;
        const float mk = ok ? 1.f : 0.f;
        T* rowp = ydst + (sample_vox + (size_t)(z * p.H + y) * p.W) * ypit;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const bool cok = (f0 + f) * 16 + 4 * q < p.cout;
          const float mf = cok ? mk : 0.f;
          float o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = acc[f][i][r];
          tally(f, i, o, std::true_type{}, mf);
          if (ok && cok) Vec<T, 4>::store(rowp + lane_o + (f0 + f) * 16, o);
        }
      }
    }
    if (p.stats) {
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        float t1[4], t2[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          t1[r] = row16_sum(s1[f][r]);
          t2[r] = row16_sum(s2[f][r]);
        }
        if (v == 0) {
          const int cl = (f0 + f - ct * TL::NFW) * 16 + 4 * q;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            sred[(((VS ? wm + 2 * wn : wm) * TL::NFW * 16) + cl + r) * 2 + 0] = t1[r];
            sred[(((VS ? wm + 2 * wn : wm) * TL::NFW * 16) + cl + r) * 2 + 1] = t2[r];
          }
        }
      }
    }
  }
#ifdef BRATS_VS8_STAMPS
  IG_STAMP(5);  // reduction between teams + epilogue
  if (lane == 0 && p.stamps) {
    long long* st = p.stamps + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * TL::NW + wave) * 6;
    for (int i = 0; i < 6; ++i) st[i] = tacc[i];
  }
#endif
  if (p.stats) {
    __syncthreads();
    if (tid < TL::NFW * 16) {
      const int c = ct * TL::NFW * 16 + tid;
      if (c < p.cout) {
        const size_t tps = (size_t)p.tz * p.ty * p.tx;
        float* dst = p.stats + (((size_t)n * tps + tile_in_sample) * p.cout + c) * 2;
        float t1 = sred[tid * 2] + sred[(TL::NFW * 16 + tid) * 2];
        float t2 = sred[tid * 2 + 1] + sred[(TL::NFW * 16 + tid) * 2 + 1];
        if constexpr (VS) {
          t1 += sred[(2 * TL::NFW * 16 + tid) * 2] + sred[(3 * TL::NFW * 16 + tid) * 2];
          t2 += sred[(2 * TL::NFW * 16 + tid) * 2 + 1] + sred[(3 * TL::NFW * 16 + tid) * 2 + 1];
        }
        dst[0] = t1;
        dst[1] = t2;
      }
    }
  }
}

// ---- host-side dispatch -----------------------------------------------------------------------
template <typename T, int KS, int CK, int DIL, int NF, bool KSPLIT, bool VS = false, bool PRE = false, bool BST = false, bool KP = false>
int conv_launch_one(const ConvParams& p, hipStream_t st) {
  constexpr int lds = conv_lds_bytes<T, KS, CK, DIL, NF, KSPLIT, VS, KP>();
  auto kern = conv_igemm_kernel<T, KS, CK, DIL, NF, KSPLIT, VS, PRE, BST, KP>;
  static std::atomic<uint64_t> attr_done{0};
  BRATS_ENSURE_LDS_ATTR(kern, lds, attr_done);
  dim3 grid((unsigned)(p.N * p.tz * p.ty * p.tx), (unsigned)(p.rows16 / ConvTile<NF, KSPLIT, VS>::NFW));
  hipLaunchKernelGGL(kern, grid, dim3(KP ? 512 : 256), lds, st, p);
  BRATS_CHECK_LAUNCH();
  return 0;
}

// grids of at most one workgroup per CU (the 16^3 level of the networks: 32 tiles) take the 8-wave form of the y-split roles
// (KP: two K-parity teams per workgroup, conv_igemm_kernel) where it is built: 16-bit, 48-channel chunks; brats_conv3d_set_kp()
extern int g_conv_kp_mode;  // conv_host.hip
static inline bool conv_kp_enabled() {
  if (g_conv_kp_mode >= 0) return g_conv_kp_mode != 0;
  static int v = -1;
  if (v < 0) { const char* e = getenv("BRATS_CONV_KP"); v = e ? atoi(e) : 1; }
  return v != 0;
}
static inline long conv_kp_max_grid() {
  static long v = -1;
  if (v < 0) { const char* e = getenv("BRATS_CONV_KP_GRID"); v = e ? atol(e) : 256; }
  return v;
}
template <typename T, int KS, int CK, int DIL, bool VS, bool PRE = false, bool BST = false>
int conv_launch_nf3(const ConvParams& p, hipStream_t st) {
  if constexpr (std::is_same<T, bf16_t>::value && CK == 48 && KS == 3 && VS) {
    const long wgs = (long)p.N * p.tz * p.ty * p.tx * (p.rows16 / 3);
    if (conv_kp_enabled() && wgs <= conv_kp_max_grid()) return conv_launch_one<T, KS, CK, DIL, 3, false, true, PRE, BST, true>(p, st);
  }
  return conv_launch_one<T, KS, CK, DIL, 3, false, VS, PRE, BST, false>(p, st);
}

static inline bool conv_vsplit_enabled() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("BRATS_CONV_VSPLIT"); v = e ? atoi(e) : 1; }
  return v != 0;
}
static inline long conv_small_grid_threshold() {
  static long v = -1;
  if (v < 0) { const char* e = getenv("BRATS_CONV_SMALLGRID"); v = e ? atol(e) : 200; }
  return v;
}
struct ConvTileChoice { int nf; bool ksplit; int nfw; };
static inline ConvTileChoice conv_choose_tile(int rows16) {
  if (rows16 % 6 == 0) return {3, false, 6};
  if (rows16 % 3 == 0) return {3, true, 3};
  if (rows16 % 4 == 0) return {2, false, 4};
  if (rows16 % 2 == 0) return {2, true, 2};
  return {1, true, 1};
}

template <typename T, int KS, int CK, int DIL, bool PRE = false>
int conv_launch_ck(const ConvParams& p, hipStream_t st) {
  const ConvTileChoice t = conv_choose_tile(p.rows16);
  if (t.nf == 3 && !t.ksplit) {
    // small volumes (16^3 levels): 96-cout tiles give fewer workgroups than CUs; the y-split roles (48 couts per
    // workgroup) double the grid at the price of staging each halo tile twice
    if (std::is_same<T, bf16_t>::value && conv_vsplit_enabled() &&
        (long)p.N * p.tz * p.ty * p.tx * (p.rows16 / 6) < conv_small_grid_threshold())
      return conv_launch_nf3<T, KS, CK, DIL, true, PRE>(p, st);
    return conv_launch_one<T, KS, CK, DIL, 3, false, false, PRE>(p, st);
  }
  if (t.nf == 3 && t.ksplit) {
    // Cout = 48 (mod 96): the y-split roles (no K reduction, shared epilogue) for bf16; K-split stays for f32 / opt-out
    if (std::is_same<T, bf16_t>::value && conv_vsplit_enabled()) return conv_launch_nf3<T, KS, CK, DIL, true, PRE>(p, st);
    return conv_launch_one<T, KS, CK, DIL, 3, true, false, PRE>(p, st);
  }
  if (t.nf == 2 && !t.ksplit) return conv_launch_one<T, KS, CK, DIL, 2, false, false, PRE>(p, st);
  if (t.nf == 2 && t.ksplit) return conv_launch_one<T, KS, CK, DIL, 2, true, false, PRE>(p, st);
  return conv_launch_one<T, KS, CK, DIL, 1, true, false, PRE>(p, st);
}

// implemented in conv_<dtype>_k<KS>_d<DIL>.hip (one translation unit each, for parallel builds)
template <typename T, int KS, int DIL> int conv_launch(const ConvParams& p, int ck, hipStream_t st);

#define CONV_DEFINE_LAUNCH_BF16(KS, DIL)                                                        \
  template <> int conv_launch<bf16_t, KS, DIL>(const ConvParams& p, int ck, hipStream_t st) {   \
    switch (ck) {                                                                               \
      case 48: return conv_launch_ck<bf16_t, KS, 48, DIL>(p, st);                               \
      case 32: return conv_launch_ck<bf16_t, KS, 32, DIL>(p, st);                               \
      case 16: return conv_launch_ck<bf16_t, KS, 16, DIL>(p, st);                               \
      case 8: return conv_launch_ck<bf16_t, KS, 8, DIL>(p, st);                                 \
    }                                                                                           \
    BRATS_FAIL(BRATS_E_UNSUPPORTED, "conv bf16: unsupported channel chunk %d", ck);             \
  }
#define CONV_DEFINE_LAUNCH_F32(KS, DIL)                                                         \
  template <> int conv_launch<float, KS, DIL>(const ConvParams& p, int ck, hipStream_t st) {    \
    switch (ck) {                                                                               \
      case 16: return conv_launch_ck<float, KS, 16, DIL>(p, st);                                \
      case 8: return conv_launch_ck<float, KS, 8, DIL>(p, st);                                  \
      case 4: return conv_launch_ck<float, KS, 4, DIL>(p, st);                                  \
    }                                                                                           \
    BRATS_FAIL(BRATS_E_UNSUPPORTED, "conv f32: unsupported channel chunk %d", ck);              \
  }
