// Split-precision implicit-GEMM 3x3x3 convolution: f32 activations in HBM, 16-bit MFMAs, f32-class results.
//
// The exact-f32 parity mode (conv_igemm_kernel<float>: v_mfma_f32_16x16x4_f32, 1/16 of the 16-bit MFMA rate) is the only
// configuration that holds north_star's "logits within 1e-3 of the reference CPU path" at every size, and it runs at a
// tenth of the bf16 throughput.  This kernel closes the gap: every f32 operand is split ONCE, when it is staged, into
//     x = hi + lo,   hi = rn16(x),   lo = rn16(x - hi)          (x - hi is exact in f32)
// and the product is taken as  hi_w * hi_x + lo_w * hi_x + hi_w * lo_x  on v_mfma_f32_16x16x32_{f16,bf16} with f32
// accumulation (the lo * lo term, <= 2^-22 (fp16) / 2^-16 (bf16) of the product, is dropped): three MFMAs at the full
// 16-bit rate instead of sixteen K = 4 f32 MFMAs.
//   fp16 twin (-DBRATS_FP16, dtype BRATS_X3_F16): 11 + 11 significand bits -- per-product error 2^-22 relative or 2^-25
//     absolute (lo becomes subnormal below |x| = 2^-3; gfx950's MFMA keeps fp16 subnormals), i.e. f32-class results for
//     operands inside fp16's range (|x| < 65504: post-normalisation activations and weights).  The forward pass.
//   bf16 build (dtype BRATS_X3_BF16): 8 + 8 bits, per-product error 2^-16, over f32's whole exponent range -- for the
//     input gradients, whose operand dY spans many decades (the reference needs a GradScaler for fp16 there,
//     learning/engine.py:304-315).
// Structure = conv_igemm_kernel's (4x4x16 voxel tile, 27 taps out of one staged halo tile, weights streamed from L2 in
// fragment order, software-pipelined MMA loop), with two LDS tiles (hi, lo: together the f32 tile's bytes) and
// hi/lo weight fragments side by side in the packed buffer ([chunk][ms][row16][hi|lo][lane][16 B]).  Replaces nn.Conv3d of
// networks/equiunet2020.py:19-25 and its input gradient when the model runs with precision = "x3".
#pragma once
#include "conv_igemm.hpp"

// 8 consecutive f32 channels (two 16-byte pieces), times the power of two `sc` -> 8 hi + 8 lo 16-bit values in channel order
DEVI void x3_split8(const u32x4 a, const u32x4 b, float sc, u32x4& hi, u32x4& lo) {
  const uint32_t w[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};  // (through scalars: see f8_quant8)
  uint32_t h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float x0 = __uint_as_float(w[2 * i]) * sc, x1 = __uint_as_float(w[2 * i + 1]) * sc;
    x3_split2(x0, x1, h[i], l[i]);
  }
  hi = u32x4{h[0], h[1], h[2], h[3]};
  lo = u32x4{l[0], l[1], l[2], l[3]};
}

// chunk of MFMA work: conv_mma_chunk's schedule with three MFMAs per (weight fragment, activation fragment) pair.  The three
// terms of a half are issued term-major, so an accumulator is revisited only after NF * NB / 2 - 1 other MFMAs.
template <int NF, int NB, typename G>
DEVI void conv_mma_chunk_x3(const char* ldsb, int lo_off, int lane_b, int q, const void* wpk_chunk, int rows16, int f0, int lane,
                            f32x4 (&acc)[NF][NB]) {
  constexpr int YB = NB / 2;
  constexpr int FOZ = G::HY * G::HX * G::S;
  constexpr int NSTEP = G::MS;
  const bf16x8* wp0 = (const bf16x8*)wpk_chunk + (size_t)f0 * 128 + lane;
  constexpr int WD = 1;  // a macro-step is 3x as long as the 16-bit kernel's: one step of weight prefetch covers L2
  bf16x8 ah[WD + 1][NF], al[WD + 1][NF];
  bf16x8 bh[NB], bl[NB];
  auto load_a = [&](auto k_) {
    constexpr int k = k_;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      ah[k % (WD + 1)][f] = wp0[((size_t)k * rows16 + f) * 128];
      al[k % (WD + 1)][f] = wp0[((size_t)k * rows16 + f) * 128 + 64];
    }
  };
  auto read_b = [&](auto k_, auto half_) {
    constexpr int k = k_, half = half_;
    constexpr int o0 = G::unitoff(4 * k), o1 = G::unitoff(4 * k + 1), o2 = G::unitoff(4 * k + 2), o3 = G::unitoff(4 * k + 3);
    int lb;
    if constexpr (o1 - o0 == G::UB && o2 - o0 == 2 * G::UB && o3 - o0 == 3 * G::UB) lb = lane_b + o0;
    else lb = lane_b + (q == 0 ? o0 : q == 1 ? o1 - G::UB : q == 2 ? o2 - 2 * G::UB : o3 - 3 * G::UB);
#pragma unroll
    for (int i = YB * half; i < YB * half + YB; ++i) {
      const int o = lb + ((i / YB) * FOZ + (i % YB) * G::HX * G::S);
      bh[i] = *(const bf16x8*)(ldsb + o);
      bl[i] = *(const bf16x8*)(ldsb + lo_off + o);
    }
  };
  auto mma = [&](auto k_, auto half_) {
    constexpr int k = k_, half = half_;
#pragma unroll
    for (int i = YB * half; i < YB * half + YB; ++i)
#pragma unroll
      for (int f = 0; f < NF; ++f) acc[f][i] = MFMA16_16x16x32(al[k % (WD + 1)][f], bh[i], acc[f][i]);
#pragma unroll
    for (int i = YB * half; i < YB * half + YB; ++i)
#pragma unroll
      for (int f = 0; f < NF; ++f) acc[f][i] = MFMA16_16x16x32(ah[k % (WD + 1)][f], bl[i], acc[f][i]);
#pragma unroll
    for (int i = YB * half; i < YB * half + YB; ++i)
#pragma unroll
      for (int f = 0; f < NF; ++f) acc[f][i] = MFMA16_16x16x32(ah[k % (WD + 1)][f], bh[i], acc[f][i]);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  static_for<0, (WD < NSTEP ? WD : NSTEP)>([&](auto k_) { load_a(k_); });
  read_b(I0{}, I0{});
  static_for<0, NSTEP>([&](auto k_) {
    constexpr int k = k_;
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
}

template <int KS, int CK, int DIL, int NF, bool VS, int TY = CONV_TY>
constexpr int conv_x3_lds_bytes() {
  using G = ConvGeom<bf16_t, KS, CK, DIL, TY>;
  return 2 * ((G::LDS_TILE + 15) / 16 * 16) + ConvTile<NF, false, VS>::SRED_BYTES;
}

// wave roles: VS = all four waves compute the same NF*16 couts for a quarter of the tile's voxels;
// !VS = wave (wm, wn) computes cout half wn (NF*16 of the workgroup's 2*NF*16) for z half wm (8 x-rows).
// TY = tile rows in y: 4, or -- VS only -- 8 (the 4x8x16 tile of conv_igemm_vs8.hpp: a wave owns 4 y-rows in each of its two
// z-slices = 8 voxel fragments, so a weight fragment pair fetched from L2 feeds 24 x 3 MFMAs instead of 12 x 3, the halo
// amplification drops 2.53x -> 2.11x, and the per-tile costs are paid half as often; statistics stay per 4x4x16 sub-tile)
// BST: the "backward statistics" form (ConvParams::by / bss, conv_igemm.hpp): this launch is the input gradient of a block's
// SECOND convolution; the tile statistics become sum u, sum u * by with u = dz * act'(by * scale + shift), by = the first
// unit's f32 forward tensor -- GroupNorm / EvoNorm backward's first pass without reading dz and by back (2 x 4 bytes per element)
template <int KS, int CK, int DIL, int NF, bool VS, int TY = CONV_TY, bool BST = false>
__global__ __launch_bounds__(256, (CK == 16 && VS && TY == CONV_TY && !BST) ? 3 : 2) void conv_igemm_x3_kernel(const ConvParams p, int ty4) {
  static_assert(TY == CONV_TY || (TY == 8 && VS), "the 8-row tile exists for the y-split roles");
  using G = ConvGeom<bf16_t, KS, CK, DIL, TY>;
  using TL = ConvTile<NF, false, VS>;
  constexpr int YB = VS ? TY / 2 : CONV_TY;         // y-rows per wave and z-slice: VS 2 (TY = 4) or 4 (TY = 8); cout-half roles 4
  constexpr int NB = 2 * YB;
  constexpr int LDS_HALF = (G::LDS_TILE + 15) / 16 * 16;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int q = lane >> 4, v = lane & 15;

  int bt = blockIdx.x;
  const int tile_in_sample = bt % (p.tz * p.ty * p.tx);
  const int txi = bt % p.tx; bt /= p.tx;
  const int tyi = bt % p.ty; bt /= p.ty;
  const int tzi = bt % p.tz;
  const int n = bt / p.tz;
  const int z0 = tzi * CONV_TZ, y0 = tyi * TY, x0 = txi * CONV_TX;  // (TY = 8: p.ty counts 8-row tiles, ty4 the 4-row ones)
  const int ct = blockIdx.y;
  const int f0 = ct * TL::NFW + (VS ? 0 : wn * NF);
  const size_t sample_vox = (size_t)n * p.D * p.H * p.W;

  // staging (see conv_igemm_kernel): a piece = 8 channels of one halo voxel = 32 bytes of f32 in global memory (two
  // 16-byte loads) = 16 bytes in each of the two LDS tiles
  constexpr int NROWS = G::HZ * G::HY;
  constexpr int PPR = G::HX * G::PPV;
  constexpr int IPR = (PPR + 63) / 64;
  constexpr int RPW = (NROWS + 3) / 4;
  int lds_off[IPR];
  int hx_part[IPR];
#pragma unroll
  for (int j = 0; j < IPR; ++j) {
    const int pc = lane + 64 * j;
    const int hx = pc / G::PPV, part = pc % G::PPV;
    const int gx = x0 - G::R + hx;
    const bool ok = pc < PPR && gx >= 0 && gx < p.W;
    hx_part[j] = ok ? (hx << 16) | part : -1;
    lds_off[j] = pc < PPR ? wave * (G::HX * G::S) + hx * G::S + part * 16 : -1;
  }

  // input scale (gradients: |max| -> fp16's top binades; see x3_scale_from_amax); the accumulators carry it until the epilogue
  const float sc = p.xamax ? x3_scale_from_amax(*p.xamax) : 1.f;
  const float isc = x3_inv_scale(sc);

  f32x4 acc[NF][NB];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int cbase = (f0 + f) * 16 + 4 * q;
    f32x4 b = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && cbase < p.cout) b = *(const f32x4*)(p.bias + cbase) * sc;
#pragma unroll
    for (int i = 0; i < NB; ++i) acc[f][i] = b;
  }

  const int lane_b = ((wm * 2) * G::HY * G::HX + (VS ? wn * YB * G::HX : 0) + v) * G::S + q * G::UB;
  const size_t chunk_stride = (size_t)G::MS * p.rows16 * 128 * 16;  // bytes of packed hi + lo weights per chunk

  for (int chunk = 0; chunk < p.nchunks; ++chunk) {
    const int c0 = chunk * CK;
    const float* src;
    int pitch;
    if (c0 < p.c1) { src = (const float*)p.x1 + c0; pitch = p.p1; }
    else { src = (const float*)p.x2 + (c0 - p.c1); pitch = p.p2; }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(src + sample_vox * pitch), (short)0, (int)((size_t)p.D * p.H * p.W * pitch * 4), 0x00020000);
    const int pb = pitch * 4;
    int goff[IPR];
#pragma unroll
    for (int j = 0; j < IPR; ++j) goff[j] = (hx_part[j] >> 16) * pb + (hx_part[j] & 0xffff) * 32;
    u32x4 r[RPW][IPR][2];
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
      const int row = wave + 4 * k;
      const int hz = row / G::HY, hy = row % G::HY;
      const int gz = z0 - G::R + hz, gy = y0 - G::R + hy;
      const bool row_ok = row < NROWS && gz >= 0 && gz < p.D && gy >= 0 && gy < p.H;  // scalar
      const int rb = ((gz * p.H + gy) * p.W + (x0 - G::R)) * pb;
#pragma unroll
      for (int j = 0; j < IPR; ++j) {
        const bool ok = row_ok && hx_part[j] >= 0;
        const int vo = ok ? rb + goff[j] : -1;  // out of range: the descriptor's range check returns zeros
        r[k][j][0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, 0));
        r[k][j][1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? vo + 16 : -1, 0, 0));
      }
    }
    if (chunk > 0) __syncthreads();
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
      if (wave + 4 * k < NROWS) {
#pragma unroll
        for (int j = 0; j < IPR; ++j)
          if (lds_off[j] >= 0) {
            u32x4 hi, lo;
            x3_split8(r[k][j][0], r[k][j][1], sc, hi, lo);
            char* dst = lds + lds_off[j] + k * 4 * (G::HX * G::S);
            *(u32x4*)dst = hi;
            *(u32x4*)(dst + LDS_HALF) = lo;
          }
      }
    }
    __syncthreads();
    const char* wchunk = (const char*)p.wpk + chunk * chunk_stride;
    conv_mma_chunk_x3<NF, NB, G>(lds, LDS_HALF, lane_b, q, wchunk, p.rows16, f0, lane, acc);
  }

  // --- epilogue: per-channel tile statistics + f32 NDHWC stores (a lane holds 4 consecutive channels of one voxel: 16 B)
  float* sred = (float*)(lds + 2 * LDS_HALF);  // [2 or 4 wave slots][NFW*16][2]
  {
    const bool x_ok = x0 + v < p.W;
    const bool second = p.y2 != nullptr && f0 * 16 >= p.ysplit;
    float* const ydst = second ? (float*)p.y2 : (float*)p.y;
    const int ypit = second ? p.y2pitch : p.ypitch;
    const int csub = second ? p.ysplit : 0;
    const int lane_o = (x0 + v) * ypit + 4 * q - csub;
    float s1[NF][4], s2[NF][4];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int r = 0; r < 4; ++r) { s1[f][r] = 0.f; s2[f][r] = 0.f; }
    // BST: the forward values under this lane's outputs (4 f32 channels per fragment and x-row = 16 bytes), requested one x-row
    // ahead of their use, and the lane's channels' {scale, shift}
#ifndef BRATS_X3_BST_AHEAD
#define BRATS_X3_BST_AHEAD 8  // all rows at once (96 registers in the 4x8x16 kernel, 249 VGPRs, no spill); 1 -> 8: bst launches -2.5 %
#endif
    constexpr int AH = BRATS_X3_BST_AHEAD < NB ? BRATS_X3_BST_AHEAD : NB;  // x-rows requested ahead of their use
    u32x4 byv[BST ? NF : 1][BST ? AH + 1 : 1];
    float bsc[BST ? NF : 1][4], bsh[BST ? NF : 1][4];
    auto load_y = [&](int i, auto checked) {
      if constexpr (BST) {
        const int z = z0 + 2 * wm + (i / YB), y = y0 + (VS ? YB * wn : 0) + (i % YB);
        const float* rowp = (const float*)p.by + (sample_vox + (size_t)(z * p.H + y) * p.W + (x0 + v)) * p.bypitch + 4 * q;
        const bool ok = !decltype(checked)::value || (z < p.D && y < p.H && x_ok);
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const bool cok = !decltype(checked)::value || (f0 + f) * 16 + 4 * q < p.cout;
          byv[f][i % (AH + 1)] = ok && cok ? *(const u32x4*)(rowp + (f0 + f) * 16) : u32x4{0u, 0u, 0u, 0u};
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
    // tile sums of one output piece: sum x, sum x^2 -- or, BST, sum u, sum u * (forward value); m = 0 / 1 mask of an edge tile
    auto tally = [&](int f, int i, const float (&o)[4], float m) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if constexpr (BST) {
          const float yy = __uint_as_float(byv[f][i % (AH + 1)][r]);
          const float pre = __builtin_fmaf(yy, bsc[f][r], bsh[f][r]);
          const float u = (pre > 0.f ? o[r] : o[r] * p.bslope) * m;
          s1[f][r] += u;
          s2[f][r] = __builtin_fmaf(u, yy, s2[f][r]);
        } else {
          const float om = o[r] * m;
          s1[f][r] += om;
          s2[f][r] = __builtin_fmaf(om, o[r], s2[f][r]);
        }
      }
    };
    const bool full = z0 + CONV_TZ <= p.D && y0 + TY <= p.H && x0 + CONV_TX <= p.W && (ct + 1) * TL::NFW * 16 <= p.cout;
    if (full) {
#pragma unroll
      for (int k = 0; k < AH; ++k) load_y(k, std::false_type{});
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        if (i + AH < NB) load_y(i + AH, std::false_type{});
        if constexpr (BST) __builtin_amdgcn_sched_barrier(0);  // (keep the loads one row ahead, not all at the top)
        const int z = z0 + 2 * wm + (i / YB), y = y0 + (VS ? YB * wn : 0) + (i % YB);
        float* rowp = ydst + (sample_vox + (size_t)(z * p.H + y) * p.W) * ypit;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          float o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = acc[f][i][r] * isc;
          tally(f, i, o, 1.f);
          Vec<float, 4>::store(rowp + lane_o + (f0 + f) * 16, o);
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < AH; ++k) load_y(k, std::true_type{});
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        if (i + AH < NB) load_y(i + AH, std::true_type{});
        if constexpr (BST) __builtin_amdgcn_sched_barrier(0);
        const int z = z0 + 2 * wm + (i / YB), y = y0 + (VS ? YB * wn : 0) + (i % YB);
        const bool ok = z < p.D && y < p.H && x_ok;
        const float mk = ok ? 1.f : 0.f;
        float* rowp = ydst + (sample_vox + (size_t)(z * p.H + y) * p.W) * ypit;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const bool cok = (f0 + f) * 16 + 4 * q < p.cout;
          float o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = acc[f][i][r] * isc;
          tally(f, i, o, cok ? mk : 0.f);
          if (ok && cok) Vec<float, 4>::store(rowp + lane_o + (f0 + f) * 16, o);
        }
      }
    }
    if (p.stats) {
#pragma unroll
      for (int f = 0; f < NF; ++f) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s1[f][r] = row16_sum(s1[f][r]);
          s2[f][r] = row16_sum(s2[f][r]);
        }
        if (v == 0) {
          const int cl = (f0 + f - ct * TL::NFW) * 16 + 4 * q;
          const int slot = VS ? wm + 2 * wn : wm;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            sred[((slot * TL::NFW * 16) + cl + r) * 2 + 0] = s1[f][r];
            sred[((slot * TL::NFW * 16) + cl + r) * 2 + 1] = s2[f][r];
          }
        }
      }
    }
  }
  if (p.stats) {
    __syncthreads();
    if constexpr (TY == 8) {
      // one 4x4x16 statistics entry per y-half (= wn): the layout every consumer reads (conv_igemm_vs8.hpp)
      vs8_stats_write<NF>(p, ty4, sred, tid, n, tzi, tyi, txi, ct);
    } else if (tid < TL::NFW * 16) {
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
template <int KS, int CK, int DIL, int NF, bool VS, int TY = CONV_TY, bool BST = false>
int conv_x3_launch_one(const ConvParams& p0, hipStream_t st) {
  constexpr int lds = conv_x3_lds_bytes<KS, CK, DIL, NF, VS, TY>();
  static_assert(lds <= 160 * 1024, "x3 LDS tile too large");
  auto kern = conv_igemm_x3_kernel<KS, CK, DIL, NF, VS, TY, BST>;
  static std::atomic<uint64_t> attr_done{0};
  BRATS_ENSURE_LDS_ATTR(kern, lds, attr_done);
  ConvParams p = p0;
  const int ty4 = p.ty;
  p.ty = ceil_div(p.H, TY);
  dim3 grid((unsigned)(p.N * p.tz * p.ty * p.tx), (unsigned)(p.rows16 / ConvTile<NF, false, VS>::NFW));
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p, ty4);
  BRATS_CHECK_LAUNCH();
  return 0;
}

inline int conv_x3_ty8_enabled() {
  static int ty8 = -1;
  if (ty8 < 0) { const char* e = getenv("BRATS_X3_TY8"); ty8 = e ? atoi(e) : 1; }
  return ty8;
}

// tile choice: the cout-half roles where the layer has an even number of NF-fragment groups, the y-split roles otherwise
// (and for small grids, as conv_launch_ck does); conv_choose_tile()'s nf, so brats_conv3d_split_granule() holds here too
template <int KS, int CK, int DIL>
int conv_x3_launch_ck(const ConvParams& p, hipStream_t st) {
  const ConvTileChoice t = conv_choose_tile(p.rows16);
  const bool small = (long)p.N * p.tz * p.ty * p.tx * (p.rows16 / (2 * t.nf) > 0 ? p.rows16 / (2 * t.nf) : 1) < conv_small_grid_threshold();
  if constexpr (CK == 16 && DIL == 1) {
    // Cout = 48 (mod 96), big volumes: the y-split roles on the 4x8x16 tile (BRATS_X3_TY8=0: the 4x4x16 tile, for A/B runs)
    const int ty8 = conv_x3_ty8_enabled();
    if (ty8 && t.nf == 3 && t.ksplit && (long)p.N * p.tz * p.ty * p.tx >= 2048) return conv_x3_launch_one<KS, CK, DIL, 3, true, 8>(p, st);
  }
  if (t.nf == 3) return (t.ksplit || small) ? conv_x3_launch_one<KS, CK, DIL, 3, true>(p, st) : conv_x3_launch_one<KS, CK, DIL, 3, false>(p, st);
  if (t.nf == 2) return (t.ksplit || small) ? conv_x3_launch_one<KS, CK, DIL, 2, true>(p, st) : conv_x3_launch_one<KS, CK, DIL, 2, false>(p, st);
  return conv_x3_launch_one<KS, CK, DIL, 1, true>(p, st);
}

// the backward-statistics forms (conv_x3_k3_bst.hip): the channel roles of a block's second convolution in the width-48 / 96 / ...
// networks -- rows a multiple of 48; 16-channel chunks where rows = 48 (mod 96), 24-channel chunks where rows = 0 (mod 96)
inline bool conv_x3_bst_supported(int ck, int dil, int rows16) {
  if (rows16 % 3 || (dil != 1 && dil != 2)) return false;
  if (ck == 16) return dil == 1 && rows16 % 6 != 0;
  return ck == 24 && rows16 % 6 == 0;
}
template <int DIL>
int conv_x3_bst_launch_ck24(const ConvParams& p, hipStream_t st) {  // conv_x3_launch_ck's choice among the NF = 3 roles
  const bool small = (long)p.N * p.tz * p.ty * p.tx * (p.rows16 / 6) < conv_small_grid_threshold();
  return small ? conv_x3_launch_one<3, 24, DIL, 3, true, CONV_TY, true>(p, st) : conv_x3_launch_one<3, 24, DIL, 3, false, CONV_TY, true>(p, st);
}
int conv_x3_bst_launch(const ConvParams& p, int ck, int dil, hipStream_t st);  // conv_x3_k3_bst.hip

// implemented in conv_x3_k3_d<DIL>.hip
template <int DIL> int conv_x3_launch(const ConvParams& p, int ck, hipStream_t st);
#define CONV_DEFINE_LAUNCH_X3(DIL)                                                            \
  template <> int conv_x3_launch<DIL>(const ConvParams& p, int ck, hipStream_t st) {          \
    switch (ck) {                                                                             \
      case 24: return conv_x3_launch_ck<3, 24, DIL>(p, st);                                   \
      case 16: return conv_x3_launch_ck<3, 16, DIL>(p, st);                                   \
      case 8: return conv_x3_launch_ck<3, 8, DIL>(p, st);                                     \
    }                                                                                         \
    BRATS_FAIL(BRATS_E_UNSUPPORTED, "conv x3: unsupported channel chunk %d", ck);             \
  }
