// y-split implicit-GEMM kernel on a 4 x 8 x 16 voxel tile (bf16, 3x3x3, dilation 1, Cout = 48 mod 96): twice the
// voxels of conv_igemm_kernel's tile per workgroup, so that what a tile costs besides its MFMAs -- kernel prologue, the
// round trip of the halo loads, barriers, the drain of the epilogue's stores: ~40 % of the one-tile kernel by ablation
// -- is paid half as often, and the halo amplification drops from 2.53x to 2.11x.  To keep two workgroups per CU the
// K chunk is 24 channels (LDS tile 6 x 10 x 18 voxels x 48 B = 52 KB; the 48-byte voxel stride costs LDS bank
// conflicts -- SQ_LDS_BANK_CONFLICT ratio 0.45 against 0.01 at 96 B -- which the kernel can afford: it is not LDS-bound).  Roles: all four waves compute the same NF*16 couts over all of K; wave (wm, wn) owns z-slices
// 2wm, 2wm+1 and y-rows 4wn..4wn+3 = 8 voxel fragments (acc: NF x 8 x 4 registers).  Statistics are written per
// 4 x 4 x 16 sub-tile (the y-half of a wave pair), i.e. in the layout every consumer already reads.
// 16-channel chunks (32-byte stride, conflict-free, three chunks per 48 channels) are slower: 0.443 vs 0.417 ms; so is
// a unit-plane layout of the 24-channel chunk ([unit][voxel][16 B], planes 80 B out of phase): 0.461 ms.
// Tried on top, not kept: touching the next chunk's half of the voxel rows early (L2 prefetch by inline-asm loads into a
// scratch register quad): -0.4 % once correct.  A first version let the compiler reuse the scratch registers while the
// loads were still in flight -- the data landed in live registers, the network produced NaNs, and NaN-filled tensors
// made every kernel of the step faster (a fake 8 % "speed-up"): compare loss values, not only times, in an A/B.
#pragma once
#include "conv_igemm.hpp"

constexpr int VS8_TY = 8;

template <int CK, int DIL, int NF>
constexpr int conv_vs8_lds_bytes() {
  using G = ConvGeom<bf16_t, 3, CK, DIL, VS8_TY>;
  return (G::LDS_TILE + 15) / 16 * 16 + 4 * NF * 16 * 2 * 4;
}

// Accumulators start at the bias (zero without one): the epilogue has no bias add.  The epilogue of a tile runs beside the
// partner workgroup's MFMA stream, where a vector instruction gets an issue slot only every ~16 cycles (an MFMA holds the
// SIMD's vector issue for 8 of its 16 cycles): its ~800 VALU instructions WERE the 12 k cycles the stamps showed (round 3),
// so it is written for instruction count -- bias in the accumulator, one fma per squared sum, one v_cvt_pk per pair.
template <int NF>
DEVI void vs8_init_acc(const ConvParams& p, f32x4 (&acc)[NF][8], int f0, int q) {
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    const int cbase = (f0 + f) * 16 + 4 * q;
    f32x4 b = {0.f, 0.f, 0.f, 0.f};
    if (p.bias && cbase < p.cout) b = *(const f32x4*)(p.bias + cbase);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[f][i] = b;
  }
}

// Epilogue of the 4 x 8 x 16-tile kernels (conv_igemm_vs8_kernel, conv_igemm_ld_kernel): per-channel statistics of the
// wave's 4x4x16 sub-tile half (DPP row sums -> sred[wm + 2 wn][NF*16][2]) and the NDHWC stores.  acc[f][i]: cout fragment f,
// voxel fragment i = x-row (z = z0 + 2 wm + i / 4, y = y0 + 4 wn + i % 4).
template <int NF, bool BST = false>
DEVI void vs8_epilogue_store(const ConvParams& p, f32x4 (&acc)[NF][8], float* sred, int wm, int wn, int q, int v,
                             int z0, int y0, int x0, int ct, int f0, size_t sample_vox, int n = 0) {
  typedef bf16_t T;
  constexpr int NB = 8, YB = 4;
  const bool x_ok = x0 + v < p.W;
  const bool second = p.y2 != nullptr && f0 * 16 >= p.ysplit;
  T* const ydst = second ? (T*)p.y2 : (T*)p.y;
  const int ypit = second ? p.y2pitch : p.ypitch;
  const int csub = second ? p.ysplit : 0;
  const int lane_o = (x0 + v) * ypit + 4 * q - csub;
  float s1[NF][4], s2[NF][4];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      s1[f][rr] = 0.f;
      s2[f][rr] = 0.f;
    }
  }
  // BST: the forward values under this lane's outputs (4 channels per fragment and x-row, 8 bytes), requested
  // BRATS_BST_AHEAD x-rows ahead of their use, and the lane's channels' {scale, shift}
  u32x2 byv[BST ? NF : 1][BST ? NB : 1];
  float bsc[BST ? NF : 1][4], bsh[BST ? NF : 1][4];
  auto load_y = [&](int i, auto checked) {
    if constexpr (BST) {
      const int z = z0 + 2 * wm + (i / YB), y = y0 + YB * wn + (i % YB);
      const T* rowp = (const T*)p.by + (sample_vox + (size_t)(z * p.H + y) * p.W + (x0 + v)) * p.bypitch + 4 * q;
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
      for (int rr = 0; rr < 4; ++rr) {
        const bool cok = c + rr < p.cout;
        bsc[f][rr] = cok ? p.bss[((size_t)n * p.cout + c + rr) * 2] : 0.f;
        bsh[f][rr] = cok ? p.bss[((size_t)n * p.cout + c + rr) * 2 + 1] : 0.f;
      }
    }
  }
  // statistics of one output piece: forward kernels sum x and x^2; BST sums u = dz * act'(pre) and u * (forward value)
  auto tally = [&](int f, int i, const float (&o)[4], auto masked, float m) {
    constexpr bool MASKED = decltype(masked)::value;
    if constexpr (BST) stat_bst<MASKED>(s1[f], s2[f], o, byv[f][i], bsc[f], bsh[f], p.bslope, m);
    else stat_fwd<MASKED>(s1[f], s2[f], o, m);
  };
  const bool full = z0 + CONV_TZ <= p.D && y0 + VS8_TY <= p.H && x0 + CONV_TX <= p.W && (ct + 1) * NF * 16 <= p.cout;
  if (full && ypit % 8 == 0 && csub % 8 == 0 && ((size_t)ydst & 15) == 0) {
    // 16-byte stores: the lanes of MFMA rows q and q ^ 1 hold channels 4q..4q+3 and the next four of the SAME voxel;
    // exchanging halves between two x-rows (v_permlane16_swap: odd 16-lane rows of the first operand <-> even rows of
    // the second) gives every lane 8 consecutive channels of ONE voxel: rows 0 / 2 keep x-row i, rows 1 / 3 take x-row
    // i + 1.  Half the store instructions; worth ~1 % (the epilogue waits on the CU's ~10 B/clk store path, not on issue).
    const int lane_w = (x0 + v) * ypit + 8 * (q >> 1) - csub + (q & 1) * p.W * ypit;
#ifndef BRATS_BST_AHEAD
#define BRATS_BST_AHEAD 8  // all of a wave's x-rows at once (48 registers, no spill: 230 VGPRs either way); 2 -> 8: bst launches -2 %, same box
#endif
    constexpr int AH = BRATS_BST_AHEAD;  // x-rows requested ahead of their use
#pragma unroll
    for (int k = 0; k < AH; ++k) load_y(k, std::false_type{});
#pragma unroll
    for (int i = 0; i < NB; i += 2) {
      if (i + AH < NB) {
        load_y(i + AH, std::false_type{});
        load_y(i + AH + 1, std::false_type{});
      }
      if constexpr (BST) __builtin_amdgcn_sched_barrier(0);  // (the loads stay where they are written)
      const int z = z0 + 2 * wm + (i / YB), y = y0 + YB * wn + (i % YB);
      T* rowp = ydst + (sample_vox + (size_t)(z * p.H + y) * p.W) * ypit;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        uint32_t pk[2][2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          float o[4];
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) o[rr] = acc[f][i + e][rr];
          tally(f, i + e, o, std::false_type{}, 1.f);
          pk[e][0] = pack2(o[0], o[1]);
          pk[e][1] = pack2(o[2], o[3]);
        }
        const u32x2 lo = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
        const u32x2 hi = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
        *(u32x4*)(rowp + lane_w + (f0 + f) * 16) = u32x4{lo[0], hi[0], lo[1], hi[1]};
      }
    }
  } else {
    load_y(0, std::true_type{});
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      if (i + 1 < NB) load_y(i + 1, std::true_type{});
      const int z = z0 + 2 * wm + (i / YB), y = y0 + YB * wn + (i % YB);
      const bool ok = z < p.D && y < p.H && x_ok;
      const float mk = ok ? 1.f : 0.f;
      T* rowp = ydst + (sample_vox + (size_t)(z * p.H + y) * p.W) * ypit;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        const bool cok = (f0 + f) * 16 + 4 * q < p.cout;
        const float mf = cok ? mk : 0.f;
        float o[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) o[rr] = acc[f][i][rr];
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
      for (int rr = 0; rr < 4; ++rr) {
        t1[rr] = row16_sum(s1[f][rr]);
        t2[rr] = row16_sum(s2[f][rr]);
      }
      if (v == 0) {
        const int cl = f * 16 + 4 * q;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          sred[(((wm + 2 * wn) * NF * 16) + cl + rr) * 2 + 0] = t1[rr];
          sred[(((wm + 2 * wn) * NF * 16) + cl + rr) * 2 + 1] = t2[rr];
        }
      }
    }
  }
}

// second half: one 4x4x16 statistics entry per y-half (wn), after a workgroup barrier behind vs8_epilogue_store
template <int NF>
DEVI void vs8_epilogue_stats(const ConvParams& p, int ty4, const float* sred, int tid, int n, int tzi, int tyi, int txi, int ct) {
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

// (Three workgroups per CU -- a 168-register build on a ring form of the MMA loop -- measured 0.39 -> 0.57 ms in round 3:
// scripts/probes/experiments/conv_igemm_ld.hpp keeps that loop.)
// PRE: normalise + activate on load (inference; conv_igemm.hpp: conv_pre_apply)
// BST: backward statistics in the epilogue (ConvParams::by / bss; training, the input gradient of a block's second unit)
template <int CK, int DIL, int NF, bool PRE = false, bool BST = false>
__global__ __launch_bounds__(256, 2) void conv_igemm_vs8_kernel(const ConvParams p, int ty4 /* 4-row tiles in y */) {
  using T = bf16_t;
  using G = ConvGeom<T, 3, CK, DIL, VS8_TY>;
  constexpr int NB = 8, YB = 4;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int q = lane >> 4, v = lane & 15;

  int bt = blockIdx.x;
  const int txi = bt % p.tx; bt /= p.tx;
  const int tyi = bt % p.ty; bt /= p.ty;  // p.ty counts 8-row tiles here
  const int tzi = bt % p.tz;
  const int n = bt / p.tz;
  const int z0 = tzi * CONV_TZ, y0 = tyi * VS8_TY, x0 = txi * CONV_TX;
  const int ct = blockIdx.y;
  const int f0 = ct * NF;
  const size_t sample_vox = (size_t)n * p.D * p.H * p.W;

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

  f32x4 acc[NF][NB];
  vs8_init_acc<NF>(p, acc, f0, q);

  const int lane_b = ((wm * 2) * G::HY * G::HX + wn * YB * G::HX + v) * G::S + q * G::UB;
  const size_t chunk_stride = (size_t)G::MS * p.rows16 * 64 * 16;
#ifdef BRATS_VS8_STAMPS  // diagnostic build only (scripts/probes/stamps_build.sh, vs8_stamps.py)
  long long tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#define VS8_STAMP(i) do { const long long t_ = __builtin_amdgcn_s_memtime(); tacc[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define VS8_STAMP(i) do { } while (0)
#endif

  for (int chunk = 0; chunk < p.nchunks; ++chunk) {
    const int c0 = chunk * CK;
    const T* src;
    int pitch;
    if (c0 < p.c1) { src = (const T*)p.x1 + c0; pitch = p.p1; }
    else { src = (const T*)p.x2 + (c0 - p.c1); pitch = p.p2; }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(src + sample_vox * pitch), (short)0, (int)((size_t)p.D * p.H * p.W * pitch * 2), 0x00020000);
    const int pb = pitch * 2;
    int goff[IPR];
#pragma unroll
    for (int j = 0; j < IPR; ++j) goff[j] = (hx_part[j] >> 16) * pb + (hx_part[j] & 0xffff) * 16;
    u32x4 r[RPW][IPR];
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
      const int row = wave + 4 * k;
      const int hz = row / G::HY, hy = row % G::HY;
      const int gz = z0 - G::R + hz, gy = y0 - G::R + hy;
      const bool row_ok = row < NROWS && gz >= 0 && gz < p.D && gy >= 0 && gy < p.H;
      const int rb = ((gz * p.H + gy) * p.W + (x0 - G::R)) * pb;
#pragma unroll
      for (int j = 0; j < IPR; ++j) {
        const int vo = (row_ok && hx_part[j] >= 0) ? rb + goff[j] : -1;
        r[k][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, 0));
      }
    }
    VS8_STAMP(0);  // prologue / issue of the halo loads
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
    if (chunk > 0) __syncthreads();
    VS8_STAMP(1);  // barrier: everybody done with the previous chunk
#ifdef BRATS_VS8_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    VS8_STAMP(2);  // the halo loads landing
#endif
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
      if (wave + 4 * k < NROWS) {
#pragma unroll
        for (int j = 0; j < IPR; ++j)
          if (lds_off[j] >= 0) {
            u32x4 val = r[k][j];
            if constexpr (PRE) {
              if (pre_ss) {  // piece by piece, right in front of its LDS write: the staged registers die as they are consumed
                const int row = wave + 4 * k;
                const int gz = z0 - G::R + row / G::HY, gy = y0 - G::R + row % G::HY;
                const bool row_ok = gz >= 0 && gz < p.D && gy >= 0 && gy < p.H;
                val = conv_pre_apply(val, psc[j], psh[j], p.pre_act == BRATS_ACT_LEAKY, p.pre_slope, row_ok && hx_part[j] >= 0);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
            *(u32x4*)(lds + lds_off[j] + k * 4 * (G::HX * G::S)) = val;
          }
      }
    }
    __syncthreads();
    VS8_STAMP(3);  // LDS writes + barrier
    const char* wchunk = (const char*)p.wpk + chunk * chunk_stride;
    conv_mma_chunk<T, 3, CK, DIL, NF, -1, NB, G>(lds, lane_b, q, wchunk, p.rows16, f0, lane, acc);
    VS8_STAMP(4);  // MFMA loop
  }

  // --- epilogue: bias, statistics per 4x4x16 sub-tile, NDHWC store ---
  constexpr int LDS_MAIN = (G::LDS_TILE + 15) / 16 * 16;
  float* sred = (float*)(lds + LDS_MAIN);  // [4 (wm + 2 wn)][NF*16][2]
  vs8_epilogue_store<NF, BST>(p, acc, sred, wm, wn, q, v, z0, y0, x0, ct, f0, sample_vox, n);
#ifdef BRATS_VS8_STAMPS
  VS8_STAMP(5);  // epilogue
  if (lane == 0 && p.stamps) {
    long long* st = p.stamps + ((size_t)blockIdx.x * 4 + wave) * 6;
    for (int i = 0; i < 6; ++i) st[i] = tacc[i];
  }
#endif
  if (p.stats) {
    __syncthreads();
    vs8_epilogue_stats<NF>(p, ty4, sred, tid, n, tzi, tyi, txi, ct);
  }
}

extern int g_conv_vs8_mode;  // conv_host.hip: -1 = BRATS_CONV_VS8 (default on), 0 / 1 = brats_conv3d_set_vs8
static inline int conv_vs8_mode() {
  if (g_conv_vs8_mode >= 0) return g_conv_vs8_mode;
  static int v = -1;
  if (v < 0) { const char* e = getenv("BRATS_CONV_VS8"); v = e ? atoi(e) : 1; }
  return v;
}

template <int CK, int DIL, int NF, bool PRE = false, bool BST = false>
int conv_launch_vs8(const ConvParams& p0, hipStream_t st) {
  constexpr int lds = conv_vs8_lds_bytes<CK, DIL, NF>();
  auto kern = conv_igemm_vs8_kernel<CK, DIL, NF, PRE, BST>;
  static std::atomic<uint64_t> attr_done{0};
  BRATS_ENSURE_LDS_ATTR(kern, lds, attr_done);
  ConvParams p = p0;
  const int ty4 = p.ty;
  p.ty = ceil_div(p.H, VS8_TY);
  dim3 grid((unsigned)(p.N * p.tz * p.ty * p.tx), (unsigned)(p.rows16 / NF));
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p, ty4);
  BRATS_CHECK_LAUNCH();
  return 0;
}
