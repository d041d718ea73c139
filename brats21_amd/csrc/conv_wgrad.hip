// Weight gradient of the 3x3x3 convolution (autograd of nn.Conv3d, networks/equiunet2020.py:19-25)
// as an MFMA GEMM whose reduction dimension is the voxel index:
//     dW[tap][co][ci] = sum_{n,v} dY[n,v][co] * X[n, v + off(tap)][ci]
//   A operand = dY^T  (rows = co, k = voxel)      B operand = X (k = voxel, cols = ci)
// Both operands are channel-minor in HBM/LDS but MFMA wants 8 consecutive k per lane, so bf16
// fragments are fetched with the gfx950 transposing LDS read (ds_read_b64_tr_b16); f32 uses K=4
// MFMAs whose one-value-per-lane operands are natural ds_read_b32.
// Workgroup = 4 waves; every wave owns the accumulators of taps {w, w+4, ...} (<= 7 taps x COF x CIF
// fragments, kept in registers across ALL the spatial tiles the workgroup walks), so an X halo tile
// and a dY tile are staged once per tile and reused by 27 taps x COF x CIF MFMAs per k-step.
// Split-K over workgroups writes f32 slabs ws[split][tap][co][ci]; a second kernel reduces them in a
// fixed order (bitwise reproducible, no float atomics) into torch's [co][ci][tap] layout.
#include <stdlib.h>
#include "twin_begin.hpp"
#include "common.hpp"

typedef __attribute__((ext_vector_type(4))) short s16x4;

struct WgradParams {
  const void* x1; const void* x2; int c1, c2, p1, p2;
  const void* dy; int dyp;
  float* ws;
  int N, D, H, W, cin, cout;
  int tz, ty, tx, ntiles, nsplit;
  int nlane;  // tile ranges (8 = one per XCD; fewer for small volumes so that fewer split-K slabs are written)
  int ntaps, dil;  // KS = 1 form only: 1 tap (a 1x1x1 convolution) or 27 shifted taps (3x3x3 at any dilation)
  int seglen, nsegz;  // split-precision z-walk kernel only (conv_wgrad_x3.hpp): tiles per column segment, segments per column
};

constexpr int WG_TZ = 4, WG_TY = 4, WG_TX = 16, WG_VOX = 256;  // tile = 16 x-rows of 16 voxels

// A workgroup owns ONE z-plane of taps (tzg in 0..2, 9 taps): its X tile then needs no z halo
// (4 x (4+2d) x (16+2d) voxels), X + dY tiles fit twice per CU, and two workgroups per CU overlap each
// other's staging and MFMA phases.  The three tzg workgroups of a tile group sit on the same XCD
// (block ids b, b+8, b+16) so the shared dY / X lines come from that XCD's L2.
// KS = 1 ("shifted-tap" form): no halo at all.  A workgroup owns ONE tap; its X tile is the dY tile's box shifted by
// dil * off(tap) (voxels outside the volume read zeros).  With ntaps = 1 that is the weight gradient of a 1x1x1
// convolution (ConvEvo / bridge / upconv / ASPP k1, networks/equiunet2021.py:212-222: a GEMM over the voxels, HBM-bound);
// with ntaps = 27 it is the weight gradient of a 3x3x3 convolution at ANY dilation -- used for the ASPP branches with
// dilation 4 and 6 (:121-189), whose halo does not fit LDS and whose 16^3 input sits in L2 anyway.
template <typename T, int DIL, int COF, int CIF, int KS = 3>
struct WgGeom {
  static constexpr bool BF = std::is_same<T, bf16_t>::value;
  static constexpr int ESZ = sizeof(T);
  static constexpr int EPL = 16 / ESZ;
  static constexpr int R = KS == 3 ? DIL : 0;
  static constexpr int HY = WG_TY + 2 * R, HX = WG_TX + 2 * R;
  static constexpr int HVOX = WG_TZ * HY * HX;
  static constexpr int CI_T = 16 * CIF, CO_T = 16 * COF;
  static constexpr int XROWB = CI_T * ESZ, YROWB = CO_T * ESZ;
  // voxel strides: the transposing reads of a 32-lane half touch 8 consecutive x-voxels x 32 B, which is
  // bank-conflict-free iff the stride is 32, 96 or 160 B (CI_T/CO_T = 48 bf16 -> 96 B, no padding)
  static constexpr int SX = BF ? (CIF == 1 ? 32 : 96) : XROWB + 16;
  static constexpr int SY = BF ? (COF == 1 ? 32 : 96) : YROWB + 16;
  static constexpr int XPPV = XROWB / 16, YPPV = YROWB / 16;
  static constexpr int XPPR = HX * XPPV, XIPR = (XPPR + 63) / 64, XROWS = WG_TZ * HY, XRPW = (XROWS + 3) / 4;
  static constexpr int YPPR = WG_TX * YPPV, YIPR = (YPPR + 63) / 64, YROWS = WG_TZ * WG_TY, YRPW = YROWS / 4;
  static constexpr int LDS_X = HVOX * SX, LDS_Y = WG_VOX * SY;
  static constexpr int LDS = LDS_X + LDS_Y;
  static constexpr int PAIRS = (KS == 3 ? 9 : 1) * CIF;  // (tap-in-plane, ci fragment) pairs
  static constexpr int PPW = (PAIRS + 3) / 4;    // pairs per wave (wave w: w, w+4, ...)
  static constexpr int NB = BF ? 32 : 64;        // bytes between consecutive ci / co fragments in a voxel row
};

DEVI bf16x8 tr_pair(const char* p0, const char* p1) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <typename T, int DIL, int COF, int CIF, int KS = 3>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradParams p) {
  using G = WgGeom<T, DIL, COF, CIF, KS>;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* ldx = lds;
  char* ldy = lds + G::LDS_X;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, v = lane & 15;
  // ---- schedule: blockIdx.x = lane8 + 8*r, r = NP*gsub + tzg (NP = 3 tap planes, or the taps of the KS = 1 form) ----
  const int NP = KS == 3 ? 3 : p.ntaps;
  const int lane8 = blockIdx.x % p.nlane, rr = blockIdx.x / p.nlane;
  const int tzg = rr % NP, gsub = rr / NP, g8 = gridDim.x / (NP * p.nlane);
  // KS = 1: the tap's shift in voxels (0 for a 1x1x1 convolution)
  const int shz = (KS == 3 || p.ntaps == 1) ? 0 : (tzg / 9 - 1) * p.dil;
  const int shy = (KS == 3 || p.ntaps == 1) ? 0 : ((tzg / 3) % 3 - 1) * p.dil;
  const int shx = (KS == 3 || p.ntaps == 1) ? 0 : (tzg % 3 - 1) * p.dil;
  const int split = lane8 + p.nlane * gsub;
  const int tpx = (p.ntiles + p.nlane - 1) / p.nlane;
  const int tile_end = min(p.ntiles, (lane8 + 1) * tpx);
  const int cot = blockIdx.y, cit = blockIdx.z;
  const int co0 = cot * G::CO_T, ci0 = cit * G::CI_T;
  const T* xsrc;
  int xpitch;
  if (ci0 < p.c1) { xsrc = (const T*)p.x1 + ci0; xpitch = p.p1; }
  else { xsrc = (const T*)p.x2 + (ci0 - p.c1); xpitch = p.p2; }
  const int ci_lim = (ci0 < p.c1 ? p.c1 : p.c1 + p.c2) - ci0;  // valid channels from this source in the tile
  const int co_lim = p.cout - co0;

  // ---- per-lane staging constants.  A staged piece = 16 bytes of one voxel; wave w owns halo rows w, w+4, ...
  //      Every piece is fetched by ONE unconditional buffer_load whose 32-bit byte offset is
  //      (tile origin, scalar) + (piece offset inside the halo box, per lane, computed once per kernel);
  //      pieces outside the volume / channel range get offset 0xffffffff, which the descriptor's range check
  //      turns into zeros.  No per-piece branches, no 64-bit address arithmetic (that per-tile scalar + vector
  //      overhead used to cost more than the MFMAs of the tile).
  int xhx[G::XIPR], xlo[G::XIPR];
  int xvo[G::XRPW][G::XIPR];
#pragma unroll
  for (int j = 0; j < G::XIPR; ++j) {
    const int pc = lane + 64 * j;
    const int hx = pc / G::XPPV, part = pc % G::XPPV;
    const bool ok = pc < G::XPPR && part * G::EPL < ci_lim;
    xhx[j] = ok ? hx : -100000;  // fails every x-range test
    xlo[j] = pc < G::XPPR ? wave * (G::HX * G::SX) + hx * G::SX + part * 16 : -1;
#pragma unroll
    for (int k = 0; k < G::XRPW; ++k) {
      const int row = wave + 4 * k;
      xvo[k][j] = (((row / G::HY) * p.H + row % G::HY) * p.W + hx) * xpitch * G::ESZ + part * 16;
    }
  }
  int yvx[G::YIPR], ylo[G::YIPR];
  int yvo[G::YRPW][G::YIPR];
#pragma unroll
  for (int j = 0; j < G::YIPR; ++j) {
    const int pc = lane + 64 * j;
    const int vx = pc / G::YPPV, part = pc % G::YPPV;
    const bool ok = pc < G::YPPR && part * G::EPL < co_lim;
    yvx[j] = ok ? vx : 100000;
    ylo[j] = pc < G::YPPR ? wave * (WG_TX * G::SY) + vx * G::SY + part * 16 : -1;
#pragma unroll
    for (int k = 0; k < G::YRPW; ++k) {
      const int row = wave + 4 * k;
      yvo[k][j] = (((row / WG_TY) * p.H + row % WG_TY) * p.W + vx) * p.dyp * G::ESZ + part * 16;
    }
  }
  const unsigned xsample_bytes = (unsigned)p.D * p.H * p.W * xpitch * G::ESZ;  // < 2^31, checked by the host
  const unsigned ysample_bytes = (unsigned)p.D * p.H * p.W * p.dyp * G::ESZ;

  // ---- the wave's (tap-in-plane, ci-fragment) pairs ----
  int poff[G::PPW];
#pragma unroll
  for (int jj = 0; jj < G::PPW; ++jj) {
    const int pid = wave + 4 * jj;
    const int t9 = pid / CIF, nn = pid % CIF;
    poff[jj] = pid < G::PAIRS ? (((t9 / 3) * G::R) * G::HX + (t9 % 3) * G::R) * G::SX + nn * G::NB : 0;
  }
  f32x4 acc[G::PPW][COF];
#pragma unroll
  for (int jj = 0; jj < G::PPW; ++jj)
#pragma unroll
    for (int m = 0; m < COF; ++m) acc[jj][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Software pipeline across tiles: the global loads of tile i+1 are issued right after tile i has been written to
  // LDS (the staging registers are free again) and fly during tile i's MFMA phase.  With only 2 workgroups per CU
  // a workgroup's own load -> LDS -> MFMA chain is otherwise serial and sets the throughput (load latency under a
  // chip-wide burst is as long as the MFMA phase).
  u32x4 rx[G::XRPW][G::XIPR], ry[G::YRPW][G::YIPR];
  auto issue_loads = [&](int tile) {
    int bt = tile;
    const int x0 = (bt % p.tx) * WG_TX; bt /= p.tx;
    const int y0 = (bt % p.ty) * WG_TY; bt /= p.ty;
    const int z0 = (bt % p.tz) * WG_TZ;
    const int n = bt / p.tz;
    const int gz0 = KS == 3 ? z0 + (tzg - 1) * DIL : z0 + shz;
    const int gy0 = KS == 3 ? y0 - DIL : y0 + shy, gx0 = KS == 3 ? x0 - DIL : x0 + shx;
    const size_t sample_vox = (size_t)n * p.D * p.H * p.W;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(xsrc + sample_vox * xpitch), (short)0,
                                                                          (int)xsample_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)p.dy + sample_vox * p.dyp + co0),
                                                                          (short)0, (int)ysample_bytes, 0x00020000);
    const int xorg = ((gz0 * p.H + gy0) * p.W + gx0) * xpitch * G::ESZ;  // may be negative at the low faces
    const int yorg = ((z0 * p.H + y0) * p.W + x0) * p.dyp * G::ESZ;
    bool xok[G::XIPR], yok[G::YIPR];
#pragma unroll
    for (int j = 0; j < G::XIPR; ++j) xok[j] = (unsigned)(gx0 + xhx[j]) < (unsigned)p.W;
#pragma unroll
    for (int j = 0; j < G::YIPR; ++j) yok[j] = x0 + yvx[j] < p.W;
#pragma unroll
    for (int k = 0; k < G::XRPW; ++k) {
      const int row = wave + 4 * k;
      const int gz = gz0 + row / G::HY, gy = gy0 + row % G::HY;
      const bool row_ok = row < G::XROWS && (unsigned)gz < (unsigned)p.D && (unsigned)gy < (unsigned)p.H;  // scalar
#pragma unroll
      for (int j = 0; j < G::XIPR; ++j) {
        const int vo = (row_ok && xok[j]) ? xorg + xvo[k][j] : -1;
        rx[k][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, vo, 0, 0));
      }
    }
#pragma unroll
    for (int k = 0; k < G::YRPW; ++k) {
      const int row = wave + 4 * k;
      const bool row_ok = z0 + row / WG_TY < p.D && y0 + row % WG_TY < p.H;
#pragma unroll
      for (int j = 0; j < G::YIPR; ++j) {
        const int vo = (row_ok && yok[j]) ? yorg + yvo[k][j] : -1;
        ry[k][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(yrs, vo, 0, 0));
      }
    }
  };
  // (the dilated 48x48 tile has no registers left to keep a second tile in flight: it would spill)
  constexpr bool PREFETCH = !(G::BF && KS == 3 && DIL == 2 && COF * CIF == 9);
  const int tile_first = lane8 * tpx + gsub;
  if (PREFETCH && tile_first < tile_end) issue_loads(tile_first);
  for (int tile = tile_first; tile < tile_end; tile += g8) {
    if (!PREFETCH) issue_loads(tile);
    __syncthreads();  // previous tile's LDS reads are done
#pragma unroll
    for (int k = 0; k < G::XRPW; ++k) {
      if (wave + 4 * k < G::XROWS) {
#pragma unroll
        for (int j = 0; j < G::XIPR; ++j)
          if (xlo[j] >= 0) *(u32x4*)(ldx + xlo[j] + k * 4 * (G::HX * G::SX)) = rx[k][j];
      }
    }
#pragma unroll
    for (int k = 0; k < G::YRPW; ++k) {
#pragma unroll
      for (int j = 0; j < G::YIPR; ++j)
        if (ylo[j] >= 0) *(u32x4*)(ldy + ylo[j] + k * 4 * (WG_TX * G::SY)) = ry[k][j];
    }
    __syncthreads();
    if (PREFETCH && tile + g8 < tile_end) issue_loads(tile + g8);
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch in front of the MFMA phase

    // ---- MFMA over the 256 voxels of the tile ----
    if constexpr (G::BF) {
      // k-step s = the two x-rows 2s, 2s+1 of the tile (row = z*4 + y).  MFMA k = 8q + e: e = 0..3 from the
      // first transposing read (row 2s, x = 4q + e), e = 4..7 from the second (row 2s+1, same x); lane
      // 4qq+pp of a quarter supplies voxel x = 4q + qq, channels 4pp..4pp+3.  Software pipeline over the
      // 8 x PPW (k-step, pair) steps: fragments of step u+1 are read while the COF MFMAs of step u issue.
      const int qq = v >> 2, pp = v & 3;
      const int ybase = (4 * q + qq) * G::SY + pp * 8;
      const int xbase = (4 * q + qq) * G::SX + pp * 8;
      constexpr int PD = 3;  // B-fragment prefetch distance in steps (3 MFMAs each): covers LDS latency
      bf16x8 a[2][COF], b[PD + 1];
      auto read_a = [&](auto s_) {
        constexpr int s = s_;
        const int yoff = ybase + (32 * s) * G::SY;
#pragma unroll
        for (int m = 0; m < COF; ++m) a[s & 1][m] = tr_pair(ldy + yoff + m * 32, ldy + yoff + 16 * G::SY + m * 32);
      };
      auto read_b = [&](auto u_) {
        constexpr int u = u_;
        constexpr int s = u / G::PPW, jj = u % G::PPW;
        const int xoff = xbase + (((s >> 1) * G::HY + 2 * (s & 1)) * G::HX) * G::SX + poff[jj];
        b[u % (PD + 1)] = tr_pair(ldx + xoff, ldx + xoff + G::HX * G::SX);
      };
      constexpr int NU = 8 * G::PPW;
      read_a(std::integral_constant<int, 0>{});
      static_for<0, PD>([&](auto u_) { read_b(u_); });
      static_for<0, NU>([&](auto u_) {
        constexpr int u = u_;
        constexpr int s = u / G::PPW, jj = u % G::PPW;
        if constexpr (u + PD < NU) read_b(std::integral_constant<int, u + PD>{});
        if constexpr (jj == 0 && s + 1 < 8) read_a(std::integral_constant<int, s + 1>{});  // a whole k-step ahead
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < COF; ++m)
          acc[jj][m] = MFMA16_16x16x32(a[s & 1][m], b[u % (PD + 1)], acc[jj][m]);
        __builtin_amdgcn_sched_barrier(0);
      });
    } else {
      // f32: k-step s' = voxels 4s'..4s'+3 of the flattened tile (x fastest), quarter q -> voxel 4s'+q
#pragma unroll 2
      for (int s = 0; s < 64; ++s) {
        const int vx = 4 * s + q;
        const int z = vx >> 6, y = (vx >> 4) & 3, x = vx & 15;
        const int yoff = vx * G::SY + v * 4;
        const int xoff = ((z * G::HY + y) * G::HX + x) * G::SX + v * 4;
        float a[COF];
#pragma unroll
        for (int m = 0; m < COF; ++m) a[m] = *(const float*)(ldy + yoff + m * 64);
#pragma unroll
        for (int jj = 0; jj < G::PPW; ++jj) {
          const float b = *(const float*)(ldx + xoff + poff[jj]);
#pragma unroll
          for (int m = 0; m < COF; ++m) acc[jj][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b, acc[jj][m], 0, 0, 0);
        }
      }
    }
  }

  // ---- write the slab part of this workgroup: ws[split][tap = 9*tzg + t9][co][ci] ----
#pragma unroll
  for (int jj = 0; jj < G::PPW; ++jj) {
    const int pid = wave + 4 * jj;
    if (pid < G::PAIRS) {
      const int t9 = pid / CIF, nn = pid % CIF;
      float* base = p.ws + (KS == 3 ? (size_t)split * 27 + tzg * 9 + t9 : (size_t)split * p.ntaps + tzg) * p.cout * p.cin;
      const int ci = ci0 + nn * 16 + v;
#pragma unroll
      for (int m = 0; m < COF; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = co0 + m * 16 + 4 * q + r;
          if (co < p.cout && nn * 16 + v < ci_lim) base[(size_t)co * p.cin + ci] = acc[jj][m][r];
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// All-taps variant for the 48 x 48 channel block of the large bf16 layers (dilation 1).  The tap-plane kernel
// above stages the dY tile three times and every X plane twice (one workgroup per tap plane): measured at
// 48->48 @128^3 its global->LDS traffic (3.2 GB per launch) alone takes 0.40 of the 0.78 ms.  Here ONE workgroup
// of 8 waves per CU owns all 27 taps of a tile: X tile with z halo (6 x 6 x 18 voxels) + dY tile = 87 KB of LDS,
// staged once per tile = 2.3x less traffic.  The 81 (tap, ci-fragment) pairs are dealt to the 8 waves
// (11 pairs x 3 co-fragments = 132 accumulator registers per lane, kept across all tiles of the workgroup);
// the next tile's 87 KB are prefetched into registers (11 x 16 B per lane) during the MFMA phase.
// CIF = ci fragments of the channel block: 3 (48 input channels) or 1 (the first layer: <= 16 input channels, 16-channel
// LDS rows of which only the real ones are fetched)
// COF = co fragments of the block: 3 (48 output channels), or 4 for the FIRST layer of the width-64 networks (8 -> 64: round 5; the
// dY voxel stride is then padded from 128 to 160 bytes, the next conflict-free value for the transposing reads)
template <int CIF, int COF = 3> struct Wg3 {
  static constexpr int HZ = WG_TZ + 2, HY = WG_TY + 2, HX = WG_TX + 2, HVOX = HZ * HY * HX;
  static constexpr int CO = 16 * COF;
  static constexpr int SX = CIF == 3 ? 96 : 32, SY = COF == 3 ? 96 : 160, XPPV = 2 * CIF, YPPV = 2 * COF;
  static constexpr int XPIECES = HVOX * XPPV, YPIECES = WG_VOX * YPPV;     // 3888 (1296), 1536 sixteen-byte pieces
  static constexpr int XI = (XPIECES + 511) / 512, YI = YPIECES / 512;     // 8 (3), 3 (4) per thread
  static constexpr int LDS_X = HVOX * SX, LDS = LDS_X + WG_VOX * SY;       // 62208 (20736) + 24576
  static constexpr int PAIRS = 27 * CIF, PPW = (PAIRS + 7) / 8;             // pair p -> wave p % 8
};

template <int CIF, int COF = 3>
__global__ __launch_bounds__(512, 1) void conv_wgrad_alltaps_kernel(const WgradParams p) {
  typedef bf16_t T;
  using G = Wg3<CIF, COF>;
  static_assert(COF == 3 || CIF == 1, "four co fragments: the first-layer form only");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* ldx = lds;
  char* ldy = lds + G::LDS_X;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, v = lane & 15;
  const int lane8 = blockIdx.x % p.nlane, gsub = blockIdx.x / p.nlane, g8 = gridDim.x / p.nlane;
  const int split = blockIdx.x;
  const int tpx = (p.ntiles + p.nlane - 1) / p.nlane;
  const int tile_end = min(p.ntiles, (lane8 + 1) * tpx);
  const int co0 = blockIdx.y * G::CO, ci0 = blockIdx.z * 16 * CIF;
  const int ci_lim = (ci0 < p.c1 ? p.c1 : p.c1 + p.c2) - ci0;  // valid channels of this source in the block
  const T* xsrc;
  int xpitch;
  if (ci0 < p.c1) { xsrc = (const T*)p.x1 + ci0; xpitch = p.p1; }
  else { xsrc = (const T*)p.x2 + (ci0 - p.c1); xpitch = p.p2; }

  // static per-lane piece codes: hz | hy << 3 | hx << 6 | part << 11 | valid << 14  (flat piece P = tid + 512 i)
  int xcode[G::XI], ycode[G::YI];
#pragma unroll
  for (int i = 0; i < G::XI; ++i) {
    const int P = tid + 512 * i;
    const int vox = P / G::XPPV, part = P % G::XPPV;
    const int hx = vox % G::HX, hy = (vox / G::HX) % G::HY, hz = vox / (G::HX * G::HY);
    xcode[i] = (P < G::XPIECES && part * 8 < ci_lim) ? (hz | hy << 3 | hx << 6 | part << 11 | 1 << 14) : 0;
  }
#pragma unroll
  for (int i = 0; i < G::YI; ++i) {
    const int P = tid + 512 * i;
    const int vox = P / G::YPPV, part = P % G::YPPV;
    ycode[i] = (vox >> 6) | ((vox >> 4) & 3) << 3 | (vox & 15) << 6 | part << 11 | 1 << 14;
  }
  const unsigned xsample_bytes = (unsigned)p.D * p.H * p.W * xpitch * 2;
  const unsigned ysample_bytes = (unsigned)p.D * p.H * p.W * p.dyp * 2;

  int poff[G::PPW];
#pragma unroll
  for (int jj = 0; jj < G::PPW; ++jj) {
    const int pid = wave + 8 * jj;
    const int t = pid / CIF, nn = pid % CIF;
    poff[jj] = pid < G::PAIRS ? (((t / 9) * G::HY + (t / 3) % 3) * G::HX + t % 3) * G::SX + nn * 32 : 0;
  }
  f32x4 acc[G::PPW][COF];
#pragma unroll
  for (int jj = 0; jj < G::PPW; ++jj)
#pragma unroll
    for (int m = 0; m < COF; ++m) acc[jj][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 rx[G::XI], ry[G::YI];
  auto issue_loads = [&](int tile) {
    int bt = tile;
    const int x0 = (bt % p.tx) * WG_TX; bt /= p.tx;
    const int y0 = (bt % p.ty) * WG_TY; bt /= p.ty;
    const int z0 = (bt % p.tz) * WG_TZ;
    const int n = bt / p.tz;
    const size_t sample_vox = (size_t)n * p.D * p.H * p.W;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(xsrc + sample_vox * xpitch), (short)0,
                                                                          (int)xsample_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)p.dy + sample_vox * p.dyp + co0),
                                                                          (short)0, (int)ysample_bytes, 0x00020000);
    unsigned zm = 0, ym = 0, xm = 0, zy = 0, yy = 0, xy = 0;
#pragma unroll
    for (int h = 0; h < G::HZ; ++h) zm |= ((unsigned)(z0 - 1 + h) < (unsigned)p.D ? 1u : 0u) << h;
#pragma unroll
    for (int h = 0; h < G::HY; ++h) ym |= ((unsigned)(y0 - 1 + h) < (unsigned)p.H ? 1u : 0u) << h;
#pragma unroll
    for (int h = 0; h < G::HX; ++h) xm |= ((unsigned)(x0 - 1 + h) < (unsigned)p.W ? 1u : 0u) << h;
    zy = zm >> 1; yy = ym >> 1; xy = xm >> 1;  // the dY tile is the halo box without its rim
    const int xorg = ((z0 - 1) * p.H + (y0 - 1)) * p.W + (x0 - 1);  // voxel index of the halo corner (may be negative)
    const int yorg = (z0 * p.H + y0) * p.W + x0;
    const int xpb = xpitch * 2, ypb = p.dyp * 2;
#pragma unroll
    for (int i = 0; i < G::XI; ++i) {
      int c = xcode[i];
      asm volatile("" : "+v"(c));  // opaque: the derived terms must not be hoisted out of the tile loop (registers)
      const unsigned ok = (unsigned)(c >> 14) & (zm >> (c & 7)) & (ym >> ((c >> 3) & 7)) & (xm >> ((c >> 6) & 31)) & 1u;
      const int pv = ((c & 7) * p.H + ((c >> 3) & 7)) * p.W + ((c >> 6) & 31);
      const int vo = ok ? (xorg + pv) * xpb + ((c >> 11) & 7) * 16 : -1;
      rx[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, vo, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < G::YI; ++i) {
      int c = ycode[i];
      asm volatile("" : "+v"(c));
      const unsigned ok = (zy >> (c & 7)) & (yy >> ((c >> 3) & 7)) & (xy >> ((c >> 6) & 31)) & 1u;
      const int pv = ((c & 7) * p.H + ((c >> 3) & 7)) * p.W + ((c >> 6) & 31);
      const int vo = ok ? (yorg + pv) * ypb + ((c >> 11) & 7) * 16 : -1;
      ry[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(yrs, vo, 0, 0));
    }
  };

  const int tile_first = lane8 * tpx + gsub;
  if (tile_first < tile_end) issue_loads(tile_first);
  for (int tile = tile_first; tile < tile_end; tile += g8) {
    __syncthreads();  // previous tile's LDS reads are done
#pragma unroll
    for (int i = 0; i < G::XI; ++i)
      if (tid + 512 * i < G::XPIECES) *(u32x4*)(ldx + (tid + 512 * i) * 16) = rx[i];
#pragma unroll
    for (int i = 0; i < G::YI; ++i) {  // (piece P = (voxel, part): a padded voxel stride when COF = 4)
      const int P = tid + 512 * i;
      *(u32x4*)(ldy + (P / G::YPPV) * G::SY + (P % G::YPPV) * 16) = ry[i];
    }
    __syncthreads();
    if (tile + g8 < tile_end) issue_loads(tile + g8);
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch in front of the MFMA phase
    // k-step s = x-rows 2s, 2s+1 of the tile (row = z*4 + y); see the tap-plane kernel for the fragment layout
    const int qq = v >> 2, pp = v & 3;
    const int ybase = (4 * q + qq) * G::SY + pp * 8;
    const int xbase = (4 * q + qq) * G::SX + pp * 8;
    constexpr int PD = 2;  // (3 in the tap-plane kernel; here the 256-register budget is full)
    bf16x8 a[2][COF], b[PD + 1];
    auto read_a = [&](auto s_) {
      constexpr int s = s_;
      const int yoff = ybase + (32 * s) * G::SY;
#pragma unroll
      for (int m = 0; m < COF; ++m) a[s & 1][m] = tr_pair(ldy + yoff + m * 32, ldy + yoff + 16 * G::SY + m * 32);
    };
    auto read_b = [&](auto u_) {
      constexpr int u = u_;
      constexpr int s = u / G::PPW, jj = u % G::PPW;
      const int xoff = xbase + (((s >> 1) * G::HY + 2 * (s & 1)) * G::HX) * G::SX + poff[jj];
      b[u % (PD + 1)] = tr_pair(ldx + xoff, ldx + xoff + G::HX * G::SX);
    };
    constexpr int NU = 8 * G::PPW;
    read_a(std::integral_constant<int, 0>{});
    static_for<0, PD>([&](auto u_) { read_b(u_); });
    static_for<0, NU>([&](auto u_) {
      constexpr int u = u_;
      constexpr int s = u / G::PPW, jj = u % G::PPW;
      if constexpr (u + PD < NU) read_b(std::integral_constant<int, u + PD>{});
      if constexpr (jj == 0 && s + 1 < 8) read_a(std::integral_constant<int, s + 1>{});
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < COF; ++m)
        acc[jj][m] = MFMA16_16x16x32(a[s & 1][m], b[u % (PD + 1)], acc[jj][m]);
      __builtin_amdgcn_sched_barrier(0);
    });
  }

  // ---- slab: ws[split][tap][co][ci] ----
#pragma unroll
  for (int jj = 0; jj < G::PPW; ++jj) {
    const int pid = wave + 8 * jj;
    if (pid < G::PAIRS) {
      const int t = pid / CIF, nn = pid % CIF;
      float* base = p.ws + ((size_t)split * 27 + t) * p.cout * p.cin;
      const int ci = ci0 + nn * 16 + v;
      if (nn * 16 + v < ci_lim) {
#pragma unroll
        for (int m = 0; m < COF; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) base[(size_t)(co0 + m * 16 + 4 * q + r) * p.cin + ci] = acc[jj][m][r];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// All-taps kernel, second form (48-channel ci blocks; the default since round 2).  Same roles and MFMA phase as above;
// what changed is everything around the MFMAs, which by ablation cost as much as they did:
//  * the X halo tile (61 KB) is fetched by LDS-DMA (buffer_load ... lds: no staging registers, no ds_write pass --
//    11 ds_write_b128 per thread and 32 VGPRs less) into one of TWO X buffers, for the NEXT tile, while the MFMA phase
//    of the current tile runs out of the other (2 x 61 KB + dY 24 KB = 146 KB of the CU's 160 KB); the flat piece order
//    P = tid + 512 i is exactly the lane-linear image LDS-DMA writes, pieces outside the volume get an out-of-range
//    offset and the DMA writes zeros;
//  * only the dY tile (3 pieces per thread) still goes through registers;
//  * per-piece byte offsets are computed once per kernel: an interior tile (halo box inside the volume, ~2/3 of all
//    tiles at 128^3) costs ONE v_add per piece instead of ~20 VALU instructions of unpack / range tests / multiply-adds,
//    which used to run in both waves of every SIMD at the same time, with the matrix pipe idle;
//  * the freed registers carry one more B fragment in flight (prefetch distance 3).
// (OPAQUE_V, lds_dma16, lds_dma16_async, make_rsrc4: common.hpp)
// Two block shapes: 48 co x 48 ci (COF, CIF = 3, 3: widths 48 / 96 / ...) and 64 co x 32 ci (4, 2: widths that are multiples
// of 64 but not of 48 -- EquiUnetASSPEvo-64; 54 (tap, ci-fragment) pairs, 7 x 4 accumulators per lane, X 2 x 48 KB + dY 32 KB).
template <int COF, int CIF> struct Wg3b {
  static constexpr int HZ = WG_TZ + 2, HY = WG_TY + 2, HX = WG_TX + 2, HVOX = HZ * HY * HX;   // 6 x 6 x 18
  static constexpr int CO = 16 * COF, CI = 16 * CIF;
  static constexpr int SX = 2 * CI, SY = 2 * CO, XPPV = CI / 8, YPPV = CO / 8;
  static constexpr int XPIECES = HVOX * XPPV, YPIECES = WG_VOX * YPPV;   // 3888, 1536 | 2592, 2048
  static constexpr int XI = (XPIECES + 511) / 512, YI = YPIECES / 512;   // 8, 3 | 6, 4 per thread
  static constexpr int XB = XI * 512 * 16;                               // 65536 | 49152: a whole number of 1-KB DMA rows per wave
  static constexpr int LDS = 2 * XB + WG_VOX * SY;                       // 155648 | 131072
  static constexpr int PAIRS = 27 * CIF, PPW = (PAIRS + 7) / 8;          // 81, 11 | 54, 7
  static_assert(YPIECES % 512 == 0, "");
};

template <int COF, int CIF>
__global__ __launch_bounds__(512, 1) void conv_wgrad_alltaps2_kernel(const WgradParams p) {
  typedef bf16_t T;
  using G = Wg3b<COF, CIF>;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* ldy = lds + 2 * G::XB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, v = lane & 15;
  const int lane8 = blockIdx.x % p.nlane, gsub = blockIdx.x / p.nlane, g8 = gridDim.x / p.nlane;
  const int split = blockIdx.x;
  const int tpx = (p.ntiles + p.nlane - 1) / p.nlane;
  const int tile_end = min(p.ntiles, (lane8 + 1) * tpx);
  const int co0 = blockIdx.y * G::CO, ci0 = blockIdx.z * G::CI;
  const T* xsrc;
  int xpitch;
  if (ci0 < p.c1) { xsrc = (const T*)p.x1 + ci0; xpitch = p.p1; }
  else { xsrc = (const T*)p.x2 + (ci0 - p.c1); xpitch = p.p2; }
  const int xpb = xpitch * 2, ypb = p.dyp * 2;

  // static per-lane piece data: byte offset from the halo corner (INT_MIN = never valid: origin + INT_MIN is beyond any
  // buffer), and the packed coordinates hz | hy << 3 | hx << 6 for the range tests of boundary tiles
  int xoffs[G::XI], xcode[G::XI];
#pragma unroll
  for (int i = 0; i < G::XI; ++i) {
    const int P = tid + 512 * i;
    const int vox = P / G::XPPV, part = P % G::XPPV;
    const int hx = vox % G::HX, hy = (vox / G::HX) % G::HY, hz = vox / (G::HX * G::HY);
    // (CIF = 1, the first layer: a 16-channel LDS row of which only the tensor's real 8-channel pieces are fetched -- the others
    //  arrive as zeros from the range check, like every piece outside the volume)
    const bool ok = P < G::XPIECES && (CIF > 1 || part * 8 < p.c1);
    xoffs[i] = ok ? ((hz * p.H + hy) * p.W + hx) * xpb + part * 16 : (int)0x80000000;
    xcode[i] = ok ? (hz | hy << 3 | hx << 6 | 1 << 14) : 0;
  }
  const unsigned xsample_bytes = (unsigned)p.D * p.H * p.W * xpitch * 2;
  const unsigned ysample_bytes = (unsigned)p.D * p.H * p.W * p.dyp * 2;

  int poff[G::PPW];
#pragma unroll
  for (int jj = 0; jj < G::PPW; ++jj) {
    const int pid = wave + 8 * jj;
    const int t = pid / CIF, nn = pid % CIF;
    poff[jj] = pid < G::PAIRS ? (((t / 9) * G::HY + (t / 3) % 3) * G::HX + t % 3) * G::SX + nn * 32 : 0;
  }
  f32x4 acc[G::PPW][COF];
#pragma unroll
  for (int jj = 0; jj < G::PPW; ++jj)
#pragma unroll
    for (int m = 0; m < COF; ++m) acc[jj][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 ry[G::YI];
  // The loads of a tile -- X -> LDS buffer `buf` by 8 LDS-DMA instructions per wave, dY -> 3 register loads -- as a
  // scalar set-up plus per-piece issue functions.  Issuing an LDS-DMA instruction stalls the issuing wave for 100-300
  // cycles (measured with s_memtime stamps, scripts/probes/wgrad_stamps.*: 2400 of a tile's 13000 cycles for the eight
  // of a wave, 8000 for the MFMA phase, 2100 waiting for the slowest wave at the barrier);
  // all pieces of the NEXT tile are issued in one batch right before the MFMA phase (alternatives: see the call site).
  // Range tests are branch-free (an interior tile ORs an all-ones mask in).
  struct TileLoads {
    rsrc4_t xrs;                  // (X: asynchronous LDS-DMA, see lds_dma16_async)
    __amdgpu_buffer_rsrc_t yrs;
    int xorg, yorg;
    unsigned zm, ym, xm, inter;  // bit h: halo plane / row / column h is inside the volume; inter = all ones for an interior tile
    char* xdst;
  };
  auto setup_loads = [&](int tile, int buf) {
    TileLoads L;
    int bt = tile;
    const int x0 = (bt % p.tx) * WG_TX; bt /= p.tx;
    const int y0 = (bt % p.ty) * WG_TY; bt /= p.ty;
    const int z0 = (bt % p.tz) * WG_TZ;
    const int n = bt / p.tz;
    const size_t sample_vox = (size_t)n * p.D * p.H * p.W;
    L.xrs = make_rsrc4(xsrc + sample_vox * xpitch, xsample_bytes);
    L.yrs = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)p.dy + sample_vox * p.dyp + co0), (short)0, (int)ysample_bytes, 0x00020000);
    L.xorg = (((z0 - 1) * p.H + (y0 - 1)) * p.W + (x0 - 1)) * xpb;  // byte offset of the halo corner (negative at the low faces)
    L.yorg = ((z0 * p.H + y0) * p.W + x0) * ypb;
    L.xdst = lds + buf * G::XB + wave * 1024;  // wave-uniform: the DMA adds lane * 16
    // bit h set <=> 0 <= o - 1 + h < size  <=>  max(0, 1 - o) <= h < min(HN, size - o + 1)
    auto inside = [](int o, int size, int hn) {
      const int lo = o >= 1 ? 0 : 1 - o, hi = size - o + 1 < hn ? size - o + 1 : hn;
      return hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
    };
    L.zm = inside(z0, p.D, G::HZ); L.ym = inside(y0, p.H, G::HY); L.xm = inside(x0, p.W, G::HX);
    L.inter = (L.zm == (1u << G::HZ) - 1 && L.ym == (1u << G::HY) - 1 && L.xm == (1u << G::HX) - 1) ? 1u : 0u;
    return L;
  };
  auto issue_x = [&](const TileLoads& L, auto i_) {
    constexpr int i = i_;
    const int c = xcode[i];
    const unsigned ok = (unsigned)(c >> 14) & (L.inter | ((L.zm >> (c & 7)) & (L.ym >> ((c >> 3) & 7)) & (L.xm >> ((c >> 6) & 31)))) & 1u;
    lds_dma16_async(L.xrs, L.xdst + i * 8192, (L.xorg + xoffs[i]) | ((int)ok - 1));
  };
  auto issue_y = [&](const TileLoads& L, auto i_) {
    constexpr int i = i_;
    // (3 pieces per tile: offset and coordinates are recomputed here instead of living in 6 registers)
    const int P = tid + 512 * i;
    const int vox = P / G::YPPV, part = P % G::YPPV;
    const int z = vox >> 6, y = (vox >> 4) & 3, x = vox & 15;
    const int yo = ((z * p.H + y) * p.W + x) * ypb + part * 16;
    const unsigned ok = (L.inter | ((L.zm >> (z + 1)) & (L.ym >> (y + 1)) & (L.xm >> (x + 1)))) & 1u;  // the dY tile = the halo box without its rim
    ry[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(L.yrs, (L.yorg + yo) | ((int)ok - 1), 0, 0));
  };

  const int tile_first = lane8 * tpx + gsub;
  const int qq = v >> 2, pp = v & 3;
  const int ybase = (4 * q + qq) * G::SY + pp * 8;
  const int xbase = (4 * q + qq) * G::SX + pp * 8;
  int cur = 0;
#ifdef BRATS_WGRAD_STAMPS  // diagnostic build only (scripts/probes/wgrad_stamps.sh): where does a tile's time go?
  long long tacc[5] = {0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#define WG_STAMP(i) do { const long long t_ = __builtin_amdgcn_s_memtime(); tacc[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define WG_STAMP(i) do { } while (0)
#endif
  if (tile_first < tile_end) {
    const TileLoads L = setup_loads(tile_first, 0);
    static_for<0, G::XI>([&](auto i_) { issue_x(L, i_); });
    static_for<0, G::YI>([&](auto i_) { issue_y(L, i_); });
  }
  for (int tile = tile_first; tile < tile_end; tile += g8, cur ^= 1) {
    // my DMA pieces of this tile have landed (they were issued during the previous MFMA phase); after the barrier
    // everybody's have, and everybody is done reading dY and the other X buffer
    WG_STAMP(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WG_STAMP(0);
    __syncthreads();
    WG_STAMP(1);
#pragma unroll
    for (int i = 0; i < G::YI; ++i) *(u32x4*)(ldy + (tid + 512 * i) * 16) = ry[i];
    __syncthreads();
    WG_STAMP(2);
    const bool more = tile + g8 < tile_end;  // scalar
    TileLoads L = setup_loads(more ? tile + g8 : tile, cur ^ 1);
    if (!more) L.zm = L.inter = 0;  // nothing follows: every piece is dropped by the range check
    __builtin_amdgcn_sched_barrier(0);
    WG_STAMP(3);
    const char* ldx = lds + cur * G::XB;
    constexpr int PD = 3;
    bf16x8 a[2][COF], b[PD + 1];
    auto read_a = [&](auto s_) {
      constexpr int s = s_;
      const int yoff = ybase + (32 * s) * G::SY;
#pragma unroll
      for (int m = 0; m < COF; ++m) a[s & 1][m] = tr_pair(ldy + yoff + m * 32, ldy + yoff + 16 * G::SY + m * 32);
    };
    auto read_b = [&](auto u_) {
      constexpr int u = u_;
      constexpr int s = u / G::PPW, jj = u % G::PPW;
      const int xoff = xbase + (((s >> 1) * G::HY + 2 * (s & 1)) * G::HX) * G::SX + poff[jj];
      b[u % (PD + 1)] = tr_pair(ldx + xoff, ldx + xoff + G::HX * G::SX);
    };
    constexpr int NU = 8 * G::PPW;
    // the next tile's loads, all issued here, back to back.  Measured alternatives (same box, 48 -> 48 @128^3, this form
    // 0.48 ms): one piece every 10 u-steps inside the MFMA loop 0.53 ms (a lone LDS-DMA among MFMAs stalls its wave ~3x
    // longer than one in a batch, and both waves of a SIMD reach it together); waves 4-7 issuing theirs in the middle of
    // the phase 0.75 ms; only waves 4-7 issuing (16 pieces each) while waves 0-3 compute 0.71 ms.  The issue rate IS
    // the CU's load bandwidth (~25 GB/s: 61 KB take >= 2400 cycles whoever issues them).  (Those three were measured while
    // hipcc still waited for every LDS-DMA before the next LDS read -- see lds_dma16_async; with the asynchronous form,
    // one load every 4 u-steps inside the loop: 48 -> 48 unchanged (0.445 ms), 96 -> 48 +4 %, every 7 u-steps: -3 %;
    // waves 0-3 issuing before the loop and waves 4-7 at u-step 30 or 44: 0.83 / 0.78 ms against 0.46.)
    static_for<0, G::XI>([&](auto i_) { issue_x(L, i_); });
    static_for<0, G::YI>([&](auto i_) { issue_y(L, i_); });
    __builtin_amdgcn_sched_barrier(0);
    read_a(std::integral_constant<int, 0>{});
    static_for<0, PD>([&](auto u_) { read_b(u_); });
    static_for<0, NU>([&](auto u_) {
      constexpr int u = u_;
      constexpr int s = u / G::PPW, jj = u % G::PPW;
      if constexpr (u + PD < NU) read_b(std::integral_constant<int, u + PD>{});
      if constexpr (jj == 0 && s + 1 < 8) read_a(std::integral_constant<int, s + 1>{});
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < COF; ++m)
        acc[jj][m] = MFMA16_16x16x32(a[s & 1][m], b[u % (PD + 1)], acc[jj][m]);
      __builtin_amdgcn_sched_barrier(0);
    });
  }

#ifdef BRATS_WGRAD_STAMPS
  WG_STAMP(4);
  if (lane == 0 && blockIdx.y == 0 && blockIdx.z == 0) {  // behind the slabs: [split][wave][5] cycle sums
    long long* st = (long long*)(p.ws + (size_t)gridDim.x * 27 * p.cout * p.cin) + ((size_t)blockIdx.x * 8 + wave) * 5;
    for (int i = 0; i < 5; ++i) st[i] = tacc[i];
  }
#endif
  // ---- slab: ws[split][tap][co][ci] ----
#pragma unroll
  for (int jj = 0; jj < G::PPW; ++jj) {
    const int pid = wave + 8 * jj;
    if (pid < G::PAIRS) {
      const int t = pid / CIF, nn = pid % CIF;
      float* base = p.ws + ((size_t)split * 27 + t) * p.cout * p.cin;
      const int ci = ci0 + nn * 16 + v;
      if (CIF > 1 || ci < p.cin) {  // (CIF = 1: the slab has the tensor's real input channels as columns)
#pragma unroll
        for (int m = 0; m < COF; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) base[(size_t)(co0 + m * 16 + 4 * q + r) * p.cin + ci] = acc[jj][m][r];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// e4m3 all-taps kernel (BASELINE.json configs[4], model.conv_fp8 = "all"): the weight gradient of the 48 x 48 channel
// blocks on v_mfma_scale_f32_16x16x128_f8f6f4 (twice the bf16 MFMA rate).  X and dY stay bf16 in HBM; the X halo tile and
// the dY tile arrive by LDS-DMA in bf16 staging buffers (as X does in the kernel above), and every thread quantises the pieces IT
// fetched (x / 2^e with the power-of-two scales of the e4m3 convolutions: from the tensors' recorded |max|) into the
// e4m3 tiles the MFMA phase reads (X 31 KB + dY 12 KB; one staging buffer is enough: it is consumed by its own thread
// before the next tile's DMA is issued).  A k-step is 128 voxels = 8 x-rows of the tile.  Both operands are
// channel-minor and the MFMA wants 32 consecutive k per lane: four 8-bit transposing LDS reads per fragment
// (ds_read_b64_tr_b8: per 16-lane group 8 voxels x 16 channels, lane i <- channel i, byte r <- voxel r; probed in
// scripts/probes/tr8.hip).  Roles as above: 81 (tap, ci-fragment) pairs over 8 waves, 3 co fragments each, accumulators
// kept over all tiles of the workgroup; the slab gets acc * (xscale * dyscale).
// Two block shapes: 48 co x 48 ci (COF, CIF = 3, 3: the width-48 networks) and 64 co x 32 ci (4, 2: width 64 = configs[4];
// 54 (tap, ci-fragment) pairs over 8 waves, 7 x 4 accumulators per lane, X 41 KB + dY 32 KB of bf16 per tile).
template <int COF, int CIF> struct Wg3f {
  static constexpr int HZ = WG_TZ + 2, HY = WG_TY + 2, HX = WG_TX + 2, HVOX = HZ * HY * HX;
  static constexpr int CO = 16 * COF, CI = 16 * CIF;
  static constexpr int SX = CI, SY = CO;                                   // e4m3 tiles: bytes per voxel
  static constexpr int XPPV = CI / 8, YPPV = CO / 8;                       // 16-byte bf16 pieces (8 channels) per voxel in HBM
  static constexpr int XPIECES = HVOX * XPPV, YPIECES = WG_VOX * YPPV;
  static constexpr int XI = (XPIECES + 511) / 512, YI = YPIECES / 512;     // 8, 3 | 6, 4 per thread
  static constexpr int XB = XI * 512 * 16, YB = YI * 512 * 16;             // bf16 staging buffers (whole 1-KB DMA rows per wave)
  static constexpr int LDS_X = HVOX * SX;
  static constexpr int LDS = XB + YB + LDS_X + WG_VOX * SY;                // 65536 + 24576 + 31104 + 12288 | 49152 + 32768 + 20736 + 16384
  static constexpr int PAIRS = 27 * CIF, PPW = (PAIRS + 7) / 8;            // 81, 11 | 54, 7
  static_assert(YPIECES % 512 == 0, "");
};

// 32 voxels (two tile rows x 16) x 16 channels of an e4m3 tile as one MFMA operand: lane (kq, L) of the k-step gets channel L
// of voxels k = 32 kq .. 32 kq + 31; o0 / o1 = this lane's byte offsets into rows 2 kq / 2 kq + 1 (see the kernel)
template <int STRIDE> DEVI i32x8 tr8_frag(const char* base, int o0, int o1) {
  typedef __attribute__((ext_vector_type(2))) int v2i;
  typedef __attribute__((address_space(3))) v2i* lp;
  const v2i a0 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lp)(base + o0));
  const v2i a1 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lp)(base + o0 + 8 * STRIDE));
  const v2i a2 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lp)(base + o1));
  const v2i a3 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lp)(base + o1 + 8 * STRIDE));
  return i32x8{a0[0], a0[1], a1[0], a1[1], a2[0], a2[1], a3[0], a3[1]};
}

struct WgradF8Params { WgradParams w; const float* amax_x1; const float* amax_x2; const float* amax_dy; };

template <int COF, int CIF>
__global__ __launch_bounds__(512, 1) void conv_wgrad_alltaps_f8_kernel(const WgradF8Params pp) {
  typedef bf16_t T;
  using G = Wg3f<COF, CIF>;
  const WgradParams& p = pp.w;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* ldx = lds + G::XB + G::YB;
  char* ldy = ldx + G::LDS_X;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, L = lane & 15;
  const int lane8 = blockIdx.x % p.nlane, gsub = blockIdx.x / p.nlane, g8 = gridDim.x / p.nlane;
  const int split = blockIdx.x;
  const int tpx = (p.ntiles + p.nlane - 1) / p.nlane;
  const int tile_end = min(p.ntiles, (lane8 + 1) * tpx);
  const int co0 = blockIdx.y * G::CO, ci0 = blockIdx.z * G::CI;
  const T* xsrc;
  int xpitch;
  if (ci0 < p.c1) { xsrc = (const T*)p.x1 + ci0; xpitch = p.p1; }
  else { xsrc = (const T*)p.x2 + (ci0 - p.c1); xpitch = p.p2; }
  const int xpb = xpitch * 2, ypb = p.dyp * 2;
  float ax = *pp.amax_x1;
  if (pp.amax_x2) ax = fmaxf(ax, *pp.amax_x2);  // (one scale for the concatenated input, as the e4m3 forward uses)
  const float xs = f8_scale_from_amax(ax), ys = f8_scale_from_amax(*pp.amax_dy);

  int xoffs[G::XI], xcode[G::XI];
#pragma unroll
  for (int i = 0; i < G::XI; ++i) {
    const int P = tid + 512 * i;
    const int vox = P / G::XPPV, part = P % G::XPPV;
    const int hx = vox % G::HX, hy = (vox / G::HX) % G::HY, hz = vox / (G::HX * G::HY);
    const bool ok = P < G::XPIECES;
    xoffs[i] = ok ? ((hz * p.H + hy) * p.W + hx) * xpb + part * 16 : (int)0x80000000;
    xcode[i] = ok ? (hz | hy << 3 | hx << 6 | 1 << 14) : 0;
  }
  const unsigned xsample_bytes = (unsigned)p.D * p.H * p.W * xpitch * 2;
  const unsigned ysample_bytes = (unsigned)p.D * p.H * p.W * p.dyp * 2;

  int poff[G::PPW];
#pragma unroll
  for (int jj = 0; jj < G::PPW; ++jj) {
    const int pid = wave + 8 * jj;
    const int t = pid / CIF, nn = pid % CIF;
    poff[jj] = pid < G::PAIRS ? (((t / 9) * G::HY + (t / 3) % 3) * G::HX + t % 3) * G::SX + nn * 16 : 0;
  }
  f32x4 acc[G::PPW][COF];
#pragma unroll
  for (int jj = 0; jj < G::PPW; ++jj)
#pragma unroll
    for (int m = 0; m < COF; ++m) acc[jj][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  struct TileLoads {
    rsrc4_t xrs, yrs;
    int xorg, yorg;
    unsigned zm, ym, xm, inter;
  };
  auto setup_loads = [&](int tile) {
    TileLoads T_;
    int bt = tile;
    const int x0 = (bt % p.tx) * WG_TX; bt /= p.tx;
    const int y0 = (bt % p.ty) * WG_TY; bt /= p.ty;
    const int z0 = (bt % p.tz) * WG_TZ;
    const int n = bt / p.tz;
    const size_t sample_vox = (size_t)n * p.D * p.H * p.W;
    T_.xrs = make_rsrc4(xsrc + sample_vox * xpitch, xsample_bytes);
    T_.yrs = make_rsrc4((const T*)p.dy + sample_vox * p.dyp + co0, ysample_bytes);
    T_.xorg = (((z0 - 1) * p.H + (y0 - 1)) * p.W + (x0 - 1)) * xpb;
    T_.yorg = ((z0 * p.H + y0) * p.W + x0) * ypb;
    auto inside = [](int o, int size, int hn) {
      const int lo = o >= 1 ? 0 : 1 - o, hi = size - o + 1 < hn ? size - o + 1 : hn;
      return hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
    };
    T_.zm = inside(z0, p.D, G::HZ); T_.ym = inside(y0, p.H, G::HY); T_.xm = inside(x0, p.W, G::HX);
    T_.inter = (T_.zm == (1u << G::HZ) - 1 && T_.ym == (1u << G::HY) - 1 && T_.xm == (1u << G::HX) - 1) ? 1u : 0u;
    return T_;
  };
  char* const xdst = lds + wave * 1024;  // wave-uniform: the DMA adds lane * 16
  auto issue_x = [&](const TileLoads& T_, auto i_) {
    constexpr int i = i_;
    int c = xcode[i];
    OPAQUE_V(c);  // (decoded per tile: the hoisted fields would cost 3 registers per piece)
    const unsigned ok = (unsigned)(c >> 14) & (T_.inter | ((T_.zm >> (c & 7)) & (T_.ym >> ((c >> 3) & 7)) & (T_.xm >> ((c >> 6) & 31)))) & 1u;
    lds_dma16_async(T_.xrs, xdst + i * 8192, (T_.xorg + xoffs[i]) | ((int)ok - 1));
  };
  auto issue_y = [&](const TileLoads& T_, auto i_) {
    constexpr int i = i_;
    int t_ = tid;
    OPAQUE_V(t_);  // (recomputed per tile: hoisted out of the tile loop these offsets cost registers the MFMA phase needs)
    const int P = t_ + 512 * i;
    const int vox = P / G::YPPV, part = P % G::YPPV;
    const int z = vox >> 6, y = (vox >> 4) & 3, x = vox & 15;
    const int yo = ((z * p.H + y) * p.W + x) * ypb + part * 16;
    const unsigned ok = (T_.inter | ((T_.zm >> (z + 1)) & (T_.ym >> (y + 1)) & (T_.xm >> (x + 1)))) & 1u;
    lds_dma16_async(T_.yrs, xdst + G::XB + i * 8192, (T_.yorg + yo) | ((int)ok - 1));
  };

  // per-lane offsets of the transposing reads: lane (kq, L) owns k = 32 kq .. 32 kq + 31 of a k-step = tile rows 2 kq + e
  // (e = 0, 1) x 16 voxels; within one read lane L supplies voxel x = (L >> 1) (+ 8 for the second half row), channel
  // bytes 8 (L & 1) .. of the 16-channel fragment.  A and B use the same voxel <-> k mapping, which is all that matters.
  int xl[2], yl[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int r = 2 * kq + e;  // row inside the k-step's 8 rows: z = r >> 2, y = r & 3
    xl[e] = (((r >> 2) * G::HY + (r & 3)) * G::HX + (L >> 1)) * G::SX + (L & 1) * 8;
    yl[e] = (r * 16 + (L >> 1)) * G::SY + (L & 1) * 8;
  }

  const int tile_first = lane8 * tpx + gsub;
  if (tile_first < tile_end) {
    const TileLoads T_ = setup_loads(tile_first);
    static_for<0, G::XI>([&](auto i_) { issue_x(T_, i_); });
    static_for<0, G::YI>([&](auto i_) { issue_y(T_, i_); });
  }
  for (int tile = tile_first; tile < tile_end; tile += g8) {
    // my own pieces have landed; quantise them while the slower waves finish the previous MFMA phase
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    u32x2 qx[G::XI], qy[G::YI];
#pragma unroll
    for (int i = 0; i < G::XI; ++i) qx[i] = f8_quant8(*(const u32x4*)(lds + (tid + 512 * i) * 16), xs);
#pragma unroll
    for (int i = 0; i < G::YI; ++i) qy[i] = f8_quant8(*(const u32x4*)(lds + G::XB + (tid + 512 * i) * 16), ys);
    __syncthreads();  // everybody is done reading the e4m3 tiles of the previous tile
#pragma unroll
    for (int i = 0; i < G::XI; ++i)
      if (i + 1 < G::XI || tid + 512 * i < G::XPIECES) *(u32x2*)(ldx + (tid + 512 * i) * 8) = qx[i];
#pragma unroll
    for (int i = 0; i < G::YI; ++i) *(u32x2*)(ldy + (tid + 512 * i) * 8) = qy[i];
    __syncthreads();
    const bool more = tile + g8 < tile_end;  // scalar
    TileLoads T_ = setup_loads(more ? tile + g8 : tile);
    if (!more) T_.zm = T_.inter = 0;  // nothing follows: every piece is dropped by the range check
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, G::XI>([&](auto i_) { issue_x(T_, i_); });
    static_for<0, G::YI>([&](auto i_) { issue_y(T_, i_); });
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, 2>([&](auto ks_) {
      constexpr int ks = ks_;  // k-step = tile rows 8 ks .. 8 ks + 7 (z = 2 ks, 2 ks + 1)
      const char* xk = ldx + ks * (2 * G::HY * G::HX * G::SX);
      i32x8 a[COF];
#pragma unroll
      for (int m = 0; m < COF; ++m) a[m] = tr8_frag<G::SY>(ldy + ks * (8 * 16 * G::SY) + m * 16, yl[0], yl[1]);
      i32x8 b[2];
      // (lane offset + pair offset added per use: hoisted out of the tile loop the 2 x PPW sums cost registers)
      auto read_b = [&](auto jj_) {
        constexpr int jj = jj_;
        int o[2] = {xl[0] + poff[jj], xl[1] + poff[jj]};
        OPAQUE_V(o[0]); OPAQUE_V(o[1]);
        b[jj & 1] = tr8_frag<G::SX>(xk, o[0], o[1]);
      };
      read_b(std::integral_constant<int, 0>{});
      static_for<0, G::PPW>([&](auto jj_) {
        constexpr int jj = jj_;
        if constexpr (jj + 1 < G::PPW) read_b(std::integral_constant<int, jj + 1>{});
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < COF; ++m)
          acc[jj][m] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[m], b[jj & 1], acc[jj][m], 0, 0, 0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  }

  // ---- slab: ws[split][tap][co][ci] = acc * (xscale * dyscale); C/D layout: lane (q = lane >> 4, v = lane & 15) holds
  //      rows (co) 4q..4q+3 of column (ci) v ----
  const float os = xs * ys;
#pragma unroll
  for (int jj = 0; jj < G::PPW; ++jj) {
    const int pid = wave + 8 * jj;
    if (pid < G::PAIRS) {
      const int t = pid / CIF, nn = pid % CIF;
      float* base = p.ws + ((size_t)split * 27 + t) * p.cout * p.cin;
      const int ci = ci0 + nn * 16 + L;
#pragma unroll
      for (int m = 0; m < COF; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) base[(size_t)(co0 + m * 16 + 4 * kq + r) * p.cin + ci] = acc[jj][m][r] * os;
    }
  }
}

// dw[co][ci][tap] = sum_split ws[split][tap][co][ci]   (fixed summation order -> bitwise reproducible; 16-byte loads)
// A block = 32 consecutive f32x4 elements x 8 split groups: thread (e, g) adds splits g, g+8, ... and the 8 partial
// sums are combined in group order through LDS (one thread per element deep the kernel had 61 workgroups, each
// lane walking all ~256 slabs serially).
// amax (split precision only, else NULL): the device scalar the dY operand was scaled by (x3_scale_from_amax) -- undone here
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nsplit,
                                                           int cout, int cin, int taps, const float* __restrict__ amax) {
  const size_t per = (size_t)taps * cout * cin;
  const size_t per4 = per / 4;  // cin % 4 == 0
  const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
  const size_t i4 = (size_t)blockIdx.x * 32 + e;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (i4 < per4) {
    // eight slabs in flight per thread, added in slab order (the one-load loop hipcc makes of the plain form waits for every
    // load before it issues the next: 32 serial HBM round trips per thread at 256 slabs, 49 us for 64 MB)
    const float* src = ws + i4 * 4;
    int k = g;
    for (; k + 56 < nsplit; k += 64) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *(const f32x4*)(src + (size_t)(k + 8 * u) * per);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < nsplit; k += 8) s += *(const f32x4*)(src + (size_t)k * per);
  }
  __shared__ f32x4 part[8][32];
  part[g][e] = s;
  __syncthreads();
  if (g == 0 && i4 < per4) {
#pragma unroll
    for (int k = 1; k < 8; ++k) s += part[k][e];
    if (amax) s *= x3_inv_scale(x3_scale_from_amax(*amax));
    const size_t i = i4 * 4;
    const int ci = i % cin;
    const int co = (i / cin) % cout;
    const int tap = (int)(i / ((size_t)cin * cout));
#pragma unroll
    for (int j = 0; j < 4; ++j) dw[((size_t)co * cin + ci + j) * taps + tap] = s[j];
  }
}

// The same reduction for layers with many (co, ci) pairs (and therefore few slabs): a block owns 128 consecutive (co, ci)
// positions and ALL taps; thread (e, g) sums the slabs of taps g, g + 8, g + 16, g + 24 in slab order -- up to 4 x 8 loads
// in flight, no cross-thread step -- into an LDS tile [taps][128]; the 128 x taps floats of dw, one contiguous run in the
// [co][ci][tap] layout, then go out with coalesced stores.  (The plain kernel's stores are 4-byte pieces at a 108-byte stride:
// 4 M of them for a 384 x 384 layer, 54 us for 16 MB; a first all-taps form that kept the 8 split groups took 27 serial round
// trips per block -- with 4 slabs only half of its threads had a load at all: 26 us.)  Fixed order: bitwise reproducible.
__global__ void __launch_bounds__(256) wgrad_reduce_taps_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nsplit,
                                                                int cout, int cin, int taps, const float* __restrict__ amax) {
  extern __shared__ __attribute__((aligned(16))) float red_tile[];  // [taps][128]
  const size_t per = (size_t)taps * cout * cin, pairs4 = (size_t)cout * cin / 4;
  const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
  const size_t p4 = (size_t)blockIdx.x * 32 + e;  // f32x4 index among the (co, ci) pairs
  if (p4 < pairs4) {
    f32x4 s[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < nsplit; k0 += 8) {
      f32x4 v[4][8];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int tap = g + 8 * t;
        const float* src = ws + ((size_t)(tap < taps ? tap : 0) * cout * cin + p4 * 4);
#pragma unroll
        for (int u = 0; u < 8; ++u)
          v[t][u] = (tap < taps && k0 + u < nsplit) ? *(const f32x4*)(src + (size_t)(k0 + u) * per) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (k0 + u < nsplit) s[t] += v[t][u];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (g + 8 * t < taps) *(f32x4*)(red_tile + (g + 8 * t) * 128 + e * 4) = s[t];
  }
  __syncthreads();
  // dw[(pair) * taps + tap] for the block's 128 pairs: 128 * taps consecutive floats
  const size_t pair0 = (size_t)blockIdx.x * 128, npairs = (size_t)cout * cin;
  float* dst = dw + pair0 * taps;
  const float isc = amax ? x3_inv_scale(x3_scale_from_amax(*amax)) : 1.f;
  for (int i = threadIdx.x; i < 128 * taps; i += blockDim.x) {
    const int pr = i / taps, tap = i % taps;
    if (pair0 + pr < npairs) dst[i] = red_tile[tap * 128 + pr] * isc;
  }
}

static void wgrad_reduce_launch(const float* ws, float* dw, int nsplit, int cout, int cin, int taps, hipStream_t st,
                                const float* amax = nullptr) {
  const size_t per = (size_t)taps * cout * cin;
  const size_t tblocks = ((size_t)cout * cin / 4 + 31) / 32;
  if (taps > 1 && taps <= 32 && tblocks >= 256)  // enough (co, ci) pairs for a grid of all-taps blocks: coalesced stores
    hipLaunchKernelGGL(wgrad_reduce_taps_kernel, dim3((unsigned)tblocks), dim3(256), (size_t)taps * 128 * sizeof(float), st, ws, dw,
                       nsplit, cout, cin, taps, amax);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((per / 4 + 31) / 32)), dim3(256), 0, st, ws, dw, nsplit, cout, cin, taps, amax);
}

// dbias[c] = sum_v dy[v][c]
template <typename T>
__global__ void dbias_kernel(const T* __restrict__ dy, int pitch, float* __restrict__ db, size_t voxels, int C) {
  __shared__ float red[256];
  const int c = blockIdx.x;
  float s = 0.f;
  for (size_t v = threadIdx.x; v < voxels; v += blockDim.x) s += to_f<T>(dy[v * pitch + c]);
  red[threadIdx.x] = s;
  __syncthreads();
  for (int m = 128; m > 0; m >>= 1) {
    if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
    __syncthreads();
  }
  if (threadIdx.x == 0) db[c] = red[0];
}

// spatial groups: nlane tile ranges (8 = one per XCD) x g8 interleaved sub-groups; total workgroups of the tap-plane
// kernel = 3 (tap planes) * nlane * g8 * cot * cit ~ 2 per CU.  Every (range, sub-group) pair writes its own f32 slab of
// the whole dW, so small volumes (16^3 levels: 32 tiles, 27 x 384 x 384 weights) use fewer ranges: with 8 the slab
// traffic (127 MB written + read) cost as much as the MFMAs.
static int wgrad_nlane(int ntiles) {
  int nl = 8;
  while (nl > 1 && ntiles / nl < 16) nl >>= 1;
  return nl;
}
static int wgrad_g8(int ntiles, int cotiles, int citiles) {
  const int nl = wgrad_nlane(ntiles);
  int g8 = ceil_div(512, 3 * nl * cotiles * citiles);
  const int cap = ceil_div(ntiles, nl);
  if (g8 > cap) g8 = cap;
  if (g8 < 1) g8 = 1;
  return g8;
}
static int wgrad_nsplit(int ntiles, int cotiles, int citiles) { return wgrad_nlane(ntiles) * wgrad_g8(ntiles, cotiles, citiles); }
static void wgrad_tiles(int dtype, int c1, int c2, int cout, int* cof, int* cif) {
  const int co16 = ceil_div(cout, 16);
  *cof = co16 % 3 == 0 ? 3 : (co16 % 2 == 0 ? 2 : 1);
  if (dtype == BRATS_F32) { *cif = 1; return; }
  // the ci tile must not straddle the x1|x2 boundary
  const int a = ceil_div(c1, 16), b = c2 > 0 ? ceil_div(c2, 16) : 0;
  auto ok = [&](int f) { return a % f == 0 && (b == 0 || b % f == 0); };
  *cif = ok(3) ? 3 : (ok(2) ? 2 : 1);
}

// all-taps kernel: 48 x 48 channel blocks only, one persistent workgroup per CU in total (8 XCD ranges x g8)
int g_wgrad_alltaps_mode = -1;  // brats_conv3d_set_wgrad_alltaps(): -1 = environment / default (on), 0 = off, 1 = on
static bool wgrad_alltaps_ok(int dtype, int dil, int c1, int c2, int cout, int ntiles, int* g8_out, int* wide_out = nullptr) {
  static int env_mode = -1, ncu = 0;
  if (env_mode < 0) {
    const char* e = getenv("BRATS_WGRAD_ALLTAPS");
    env_mode = e ? atoi(e) : 1;
    int dev = 0;
    hipDeviceProp_t prop;
    ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  const int mode = g_wgrad_alltaps_mode >= 0 ? g_wgrad_alltaps_mode : env_mode;
  const int cin = c1 + (c2 > 0 ? c2 : 0);
  const bool narrow = c2 <= 0 && c1 <= 16;  // the first layer: one 16-channel ci block
  if (!mode || dtype != BRATS_BF16 || dil != 1) return false;
  // block shape: 48 co x 48 ci, or 64 co x 32 ci for widths that are multiples of 64 but not of 48 (LDS-DMA form only)
  bool wide = false;
  if (cout % 48 || (!narrow && (c1 % 48 || (c2 > 0 && c2 % 48)))) {
    if (cout % 64 || (!narrow && (c1 % 32 || (c2 > 0 && c2 % 32)))) return false;
    wide = true;  // (narrow + wide: the first layer of a width-64 network, 64 co x 16 ci blocks)
  }
  const int blocks = wide ? (cout / 64) * (narrow ? 1 : cin / 32) : (cout / 48) * (narrow ? 1 : cin / 48);
  const int nl = wgrad_nlane(ntiles);
  int g8 = ceil_div(ncu, nl * blocks);
  if (g8 < 1) g8 = 1;
  if (ntiles < 4 * nl * g8) return false;  // too few tiles per workgroup to amortise 132 accumulators x 27 taps of slab
  *g8_out = g8;
  if (wide_out) *wide_out = wide ? 1 : 0;
  return true;
}
extern "C" int BRATS_API(brats_conv3d_set_wgrad_alltaps)(int mode) {
  const int old = g_wgrad_alltaps_mode;
  g_wgrad_alltaps_mode = mode < 0 ? -1 : (mode ? 1 : 0);
  return old;
}

static size_t x3_align(size_t b);
static size_t wgrad_x3_fused_ws_bytes(int N, int D, int H, int W, int c1, int c2, int cout);
extern "C" size_t BRATS_API(brats_conv3d_wgrad_ws_bytes)(int dtype, int ksize, int N, int D, int H, int W, int c1, int c2, int cout) {
  if (ksize != 3) return 0;
  if (dtype == BRATS_X3_BF16) {
    // split precision: three groups of 16-bit slabs + the hi / lo tensors of x1, x2 and dy (wgrad_x3)
    const size_t vox = (size_t)N * D * H * W;
    const size_t slab = BRATS_API(brats_conv3d_wgrad_ws_bytes)(BRATS_BF16, 3, N, D, H, W, c1, c2, cout) * 3;
    auto sp = [&](int c) { return c > 0 ? 2 * x3_align(vox * c * 2) : (size_t)0; };
    const size_t three = x3_align(slab) + sp(c1) + sp(c2) + sp(cout), fused = wgrad_x3_fused_ws_bytes(N, D, H, W, c1, c2, cout);
    return three > fused ? three : fused;
  }
  int cof, cif;
  wgrad_tiles(dtype, c1, c2, cout, &cof, &cif);
  const int ntiles = N * ceil_div(D, WG_TZ) * ceil_div(H, WG_TY) * ceil_div(W, WG_TX);
  const int cin_tiles = ceil_div(c1, 16 * cif) + (c2 > 0 ? ceil_div(c2, 16 * cif) : 0);
  int ns = wgrad_nsplit(ntiles, ceil_div(cout, 16 * cof), cin_tiles);
  int g8 = 0;
  if (wgrad_alltaps_ok(dtype, 1, c1, c2, cout, ntiles, &g8) && wgrad_nlane(ntiles) * g8 > ns) ns = wgrad_nlane(ntiles) * g8;  // the dilation is not known here
  return (size_t)ns * 27 * cout * (c1 + (c2 > 0 ? c2 : 0)) * sizeof(float);
}

template <typename T, int DIL, int COF, int CIF, int KS = 3>
static int wgrad_launch(const WgradParams& p, dim3 grid, hipStream_t st) {
  using G = WgGeom<T, DIL, COF, CIF, KS>;
  auto kern = conv_wgrad_kernel<T, DIL, COF, CIF, KS>;
  static std::atomic<uint64_t> attr_done{0};
  BRATS_ENSURE_LDS_ATTR(kern, G::LDS, attr_done);
  hipLaunchKernelGGL(kern, grid, dim3(256), G::LDS, st, p);
  BRATS_CHECK_LAUNCH();
  return 0;
}

template <typename T, int DIL, int KS = 3>
static int wgrad_dispatch(const WgradParams& p, int cof, int cif, dim3 grid, hipStream_t st) {
#define WG_CASE(A, B) if (cof == A && cif == B) return wgrad_launch<T, DIL, A, B, KS>(p, grid, st);
  WG_CASE(3, 1) WG_CASE(2, 1) WG_CASE(1, 1)
  if constexpr (std::is_same<T, bf16_t>::value) {
    WG_CASE(3, 3) WG_CASE(3, 2) WG_CASE(2, 3) WG_CASE(2, 2) WG_CASE(1, 3) WG_CASE(1, 2)
  }
#undef WG_CASE
  BRATS_FAIL(BRATS_E_UNSUPPORTED, "wgrad: unsupported tile %dx%d", cof, cif);
}

// launches the MFMA kernel of one weight-gradient problem (16-bit or f32 operands as `dtype` says) into the split-K slabs at
// ws; *nsplit_out = number of slabs written ([split][27][cout][cin] f32 each)
static int wgrad_mfma(const void* x1, int c1, int pitch1, const void* x2, int c2, int pitch2, const void* dy, int dypitch, float* ws,
                      int dtype, int dil, int N, int D, int H, int W, int cout, hipStream_t st, int* nsplit_out) {
  int cof, cif;
  wgrad_tiles(dtype, c1, c2, cout, &cof, &cif);
  WgradParams p;
  p.x1 = x1; p.x2 = x2; p.c1 = c1; p.c2 = c2; p.p1 = pitch1; p.p2 = pitch2;
  p.dy = dy; p.dyp = dypitch; p.ws = ws;
  p.N = N; p.D = D; p.H = H; p.W = W; p.cin = c1 + c2; p.cout = cout;
  p.tz = ceil_div(D, WG_TZ); p.ty = ceil_div(H, WG_TY); p.tx = ceil_div(W, WG_TX);
  p.ntiles = N * p.tz * p.ty * p.tx;
  const int cot = ceil_div(cout, 16 * cof);
  const int cit = ceil_div(c1, 16 * cif) + (c2 > 0 ? ceil_div(c2, 16 * cif) : 0);
  p.nsplit = wgrad_nsplit(p.ntiles, cot, cit);
  p.nlane = wgrad_nlane(p.ntiles);
  // ci tiles of x2 start at tile index ceil(c1/CI_T): only exact when c1 % CI_T == 0 or c2 == 0
  if (c2 > 0 && c1 % (16 * cif)) BRATS_FAIL(BRATS_E_UNSUPPORTED, "wgrad: c1=%d must be a multiple of the ci tile %d", c1, 16 * cif);
  int g8a = 0, wide = 0;
  const bool alltaps = wgrad_alltaps_ok(dtype, dil, c1, c2, cout, p.ntiles, &g8a, &wide);
  // tap-plane kernel: slab entries of padded ci / co lanes are never written: clear the slab
  if (!alltaps && ((c1 + c2) % 16 || cout % 16)) {
    hipError_t e = hipMemsetAsync(ws, 0, (size_t)p.nsplit * 27 * cout * p.cin * sizeof(float), st);
    if (e != hipSuccess) BRATS_FAIL(BRATS_E_HIP, "wgrad: memset: %s", hipGetErrorString(e));
  }
  dim3 grid(3 * p.nsplit, cot, cit);  // x = lane + nlane*(3*gsub + tzg)
  int rc;
  if (alltaps) {
    p.nsplit = p.nlane * g8a;
    constexpr int lds_48 = Wg3b<3, 3>::LDS, lds_wide = Wg3b<4, 2>::LDS;
    static std::atomic<uint64_t> attr_a{0}, attr_b{0}, attr_c{0}, attr_d{0};
    BRATS_ENSURE_LDS_ATTR(conv_wgrad_alltaps_kernel<3>, Wg3<3>::LDS, attr_a);
    BRATS_ENSURE_LDS_ATTR(conv_wgrad_alltaps_kernel<1>, Wg3<1>::LDS, attr_b);
    static std::atomic<uint64_t> attr_e{0};
    BRATS_ENSURE_LDS_ATTR((conv_wgrad_alltaps_kernel<1, 4>), (Wg3<1, 4>::LDS), attr_e);
    BRATS_ENSURE_LDS_ATTR((conv_wgrad_alltaps2_kernel<3, 3>), lds_48, attr_c);
    BRATS_ENSURE_LDS_ATTR((conv_wgrad_alltaps2_kernel<4, 2>), lds_wide, attr_d);
    static int form = -1;  // BRATS_WGRAD_ALLTAPS=1: the round-1 form (register staging, one X buffer) for same-box A/B runs
    if (form < 0) {
      const char* e = getenv("BRATS_WGRAD_ALLTAPS");
      form = (e && atoi(e) == 1) ? 1 : 2;
    }
    if (wide && c2 <= 0 && c1 <= 16) {
      hipLaunchKernelGGL((conv_wgrad_alltaps_kernel<1, 4>), dim3(p.nsplit, cout / 64, 1), dim3(512), (Wg3<1, 4>::LDS), st, p);
    } else if (wide) {
      hipLaunchKernelGGL((conv_wgrad_alltaps2_kernel<4, 2>), dim3(p.nsplit, cout / 64, p.cin / 32), dim3(512), lds_wide, st, p);
    } else if (c2 <= 0 && c1 <= 16 && form != 1 && c1 % 8 == 0) {
      // the first layer on the LDS-DMA form (round 6): two X buffers of 16-channel rows, the next tile's loads in flight behind
      // the MFMA phase -- this layer is all loads (dY: 24 KB per tile against 1.5 k cycles of MFMA per wave)
      static std::atomic<uint64_t> attr_f{0};
      BRATS_ENSURE_LDS_ATTR((conv_wgrad_alltaps2_kernel<3, 1>), (Wg3b<3, 1>::LDS), attr_f);
      hipLaunchKernelGGL((conv_wgrad_alltaps2_kernel<3, 1>), dim3(p.nsplit, cout / 48, 1), dim3(512), (Wg3b<3, 1>::LDS), st, p);
    } else if (c2 <= 0 && c1 <= 16) {
      // the slab columns of the padded ci lanes (c1 < 16) are never written and never read (cin = c1)
      hipLaunchKernelGGL(conv_wgrad_alltaps_kernel<1>, dim3(p.nsplit, cout / 48, 1), dim3(512), Wg3<1>::LDS, st, p);
    } else if (form == 1) {
      hipLaunchKernelGGL(conv_wgrad_alltaps_kernel<3>, dim3(p.nsplit, cout / 48, p.cin / 48), dim3(512), Wg3<3>::LDS, st, p);
    } else {
      hipLaunchKernelGGL((conv_wgrad_alltaps2_kernel<3, 3>), dim3(p.nsplit, cout / 48, p.cin / 48), dim3(512), lds_48, st, p);
    }
    rc = 0;
  } else if (dtype == BRATS_BF16) rc = dil == 1 ? wgrad_dispatch<bf16_t, 1>(p, cof, cif, grid, st) : wgrad_dispatch<bf16_t, 2>(p, cof, cif, grid, st);
  else rc = dil == 1 ? wgrad_dispatch<float, 1>(p, cof, cif, grid, st) : wgrad_dispatch<float, 2>(p, cof, cif, grid, st);
  if (rc) return rc;
  BRATS_CHECK_LAUNCH();
  *nsplit_out = p.nsplit;
  return 0;
}

// ---- split precision (BRATS_X3_*): f32 X and dY, three 16-bit MFMA products -----------------------------------------
// dW = Xhi (x) dYhi + Xhi (x) dYlo + Xlo (x) dYhi with x = hi + lo split into two 16-bit values (conv_igemm_x3.hpp has the
// error analysis).  Unlike the forward kernel the split is NOT done while staging: the all-taps kernel fills its LDS by
// DMA (which cannot transform data) and its hi + lo tiles (2 x 87 KB) would not fit a CU.  One streaming pass writes the
// hi / lo tensors (dense, 16-bit) into the workspace, the UNCHANGED 16-bit kernels run three times into three groups of
// split-K slabs, and the one fixed-order reduction sums all of them (bitwise reproducible).  Price: 12 extra bytes of HBM
// traffic per operand element and three tile stagings instead of one.
// a thread owns 8 consecutive channels of one voxel (32 B of f32 in, 16 B + 16 B out); two voxels in flight, short-lived
// blocks, non-temporal loads and stores (the tensors are streamed once; csrc/probe.hip's policy: 6.2 TB/s against 5.5)
template <int V>
__global__ void __launch_bounds__(256) x3_split_kernel(const float* __restrict__ src, int pitch, bf16_t* __restrict__ hi,
                                                       bf16_t* __restrict__ lo, size_t voxels, int C, const float* __restrict__ amax) {
  const float sc = amax ? x3_scale_from_amax(*amax) : 1.f;
  const int cv = C / 8;
  const size_t total = voxels * cv;
  const size_t stride = (size_t)gridDim.x * 256;
  auto load = [&](size_t i, u32x4& a, u32x4& b) {
    const float* p = src + (i / cv) * pitch + (i % cv) * 8;
    a = __builtin_nontemporal_load((const u32x4*)p);
    b = __builtin_nontemporal_load((const u32x4*)(p + 4));
  };
  auto emit = [&](size_t i, const u32x4& a, const u32x4& b) {
    const uint32_t w[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    uint32_t h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x0 = __uint_as_float(w[2 * e]) * sc, x1 = __uint_as_float(w[2 * e + 1]) * sc;
      x3_split2(x0, x1, h[e], l[e]);
    }
    const size_t o = (i / cv) * C + (i % cv) * 8;
    __builtin_nontemporal_store(u32x4{h[0], h[1], h[2], h[3]}, (u32x4*)(hi + o));
    __builtin_nontemporal_store(u32x4{l[0], l[1], l[2], l[3]}, (u32x4*)(lo + o));
  };
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + stride < total; i += 2 * stride) {
    u32x4 a0, b0, a1, b1;
    load(i, a0, b0);
    load(i + stride, a1, b1);
    emit(i, a0, b0);
    emit(i + stride, a1, b1);
  }
  if (i < total) {
    u32x4 a0, b0;
    load(i, a0, b0);
    emit(i, a0, b0);
  }
}
static size_t x3_align(size_t b) { return (b + 255) / 256 * 256; }
static size_t wgrad_x3_split_bytes(size_t voxels, int c) { return c > 0 ? x3_align(voxels * c * 2) : 0; }  // one of hi / lo
static void x3_split_launch(const float* src, int pitch, bf16_t* hi, bf16_t* lo, size_t vox, int c, const float* amax, hipStream_t st) {
  const size_t total = vox * (c / 8);
  size_t nb = (total + 511) / 512;  // two pieces per thread
  const unsigned blocks = (unsigned)(nb < 1 ? 1 : (nb > 262144 ? 262144 : nb));
  hipLaunchKernelGGL(x3_split_kernel<0>, dim3(blocks), dim3(256), 0, st, src, pitch, hi, lo, vox, c, amax);
}

static int wgrad_x3(const void* x1, int c1, int pitch1, const void* x2, int c2, int pitch2, const void* dy, int dypitch,
                    const float* amax_dy, float* ws, float* dw, size_t slab_bytes, int dil, int N, int D, int H, int W, int cout,
                    hipStream_t st) {
  const size_t vox = (size_t)N * D * H * W;
  char* b = (char*)ws + x3_align(slab_bytes);
  bf16_t* h1 = (bf16_t*)b; b += wgrad_x3_split_bytes(vox, c1);
  bf16_t* l1 = (bf16_t*)b; b += wgrad_x3_split_bytes(vox, c1);
  bf16_t* h2 = (bf16_t*)b; b += wgrad_x3_split_bytes(vox, c2);
  bf16_t* l2 = (bf16_t*)b; b += wgrad_x3_split_bytes(vox, c2);
  bf16_t* hy = (bf16_t*)b; b += wgrad_x3_split_bytes(vox, cout);
  bf16_t* ly = (bf16_t*)b;
  auto split = [&](const void* src, int pitch, bf16_t* hi, bf16_t* lo, int c, const float* amax) {
    x3_split_launch((const float*)src, pitch, hi, lo, vox, c, amax, st);
  };
  split(x1, pitch1, h1, l1, c1, nullptr);
  if (c2 > 0) split(x2, pitch2, h2, l2, c2, nullptr);
  split(dy, dypitch, hy, ly, cout, amax_dy);  // (dY * 2^k: undone by the reduction)
  BRATS_CHECK_LAUNCH();
  int ns = 0, total = 0;
  const size_t per = (size_t)27 * cout * (c1 + c2);
  // (small terms first: the reduction adds the slabs in this order)
  int rc = wgrad_mfma(l1, c1, c1, c2 > 0 ? l2 : nullptr, c2, c2, hy, cout, ws, BRATS_BF16, dil, N, D, H, W, cout, st, &ns);
  if (rc) return rc;
  total += ns;
  rc = wgrad_mfma(h1, c1, c1, c2 > 0 ? h2 : nullptr, c2, c2, ly, cout, ws + (size_t)total * per, BRATS_BF16, dil, N, D, H, W, cout, st, &ns);
  if (rc) return rc;
  total += ns;
  rc = wgrad_mfma(h1, c1, c1, c2 > 0 ? h2 : nullptr, c2, c2, hy, cout, ws + (size_t)total * per, BRATS_BF16, dil, N, D, H, W, cout, st, &ns);
  if (rc) return rc;
  total += ns;
  wgrad_reduce_launch((const float*)ws, dw, total, cout, c1 + c2, 27, st, amax_dy);
  return 0;
}

#include "conv_wgrad_x3.hpp"  // the fused form: split in the staging path, one launch (round 5)
extern "C" int BRATS_API(brats_conv3d_set_x3_wgrad_fused)(int mode) {  // (here, not in the header: gen_twin_dispatch.py scans the .hip files)
  const int old = g_x3_wgrad_fused_mode;
  g_x3_wgrad_fused_mode = mode < 0 ? -1 : (mode > 2 ? 2 : mode);
  return old;
}

static int wgrad_impl(const void* x1, int c1, int pitch1, const void* x2, int c2, int pitch2, const void* dy, int dypitch,
                      const float* amax_dy, float* ws, float* dw, float* dbias, int dtype, int ksize, int dil, int N, int D, int H,
                      int W, int cout, brats_stream_t s) {
  if (!x1 || !dy || !ws || !dw || c1 <= 0 || cout <= 0) BRATS_FAIL(BRATS_E_ARG, "wgrad: null pointer / bad size");
  if (ksize != 3 || (dil != 1 && dil != 2)) BRATS_FAIL(BRATS_E_UNSUPPORTED, "wgrad: ksize=%d dil=%d unsupported", ksize, dil);
  if (c2 < 0) c2 = 0;
  if (c2 > 0 && !x2) BRATS_FAIL(BRATS_E_ARG, "wgrad: c2 > 0 but x2 NULL");
  const bool x3 = dtype == BRATS_X3_BF16;
  const int epl = dtype == BRATS_BF16 ? 8 : 4;
  if (pitch1 % epl || (c2 && pitch2 % epl) || dypitch % epl || c1 % epl || c2 % epl || cout % epl)
    BRATS_FAIL(BRATS_E_ARG, "wgrad: channel counts / pitches must be multiples of %d", epl);
  if (x3 && (c1 % 8 || c2 % 8 || cout % 8)) BRATS_FAIL(BRATS_E_ARG, "wgrad (split precision): channel counts must be multiples of 8");
  {  // staged pieces are addressed by 32-bit byte offsets inside one sample (buffer_load voffset)
    const int mp = pitch1 > pitch2 ? (pitch1 > dypitch ? pitch1 : dypitch) : (pitch2 > dypitch ? pitch2 : dypitch);
    if ((double)D * H * W * mp * (dtype == BRATS_BF16 ? 2 : 4) >= 2147483648.0)
      BRATS_FAIL(BRATS_E_UNSUPPORTED, "wgrad: one sample of %dx%dx%d x pitch %d exceeds the 2 GiB buffer-offset range", D, H, W, mp);
  }
  hipStream_t st = (hipStream_t)s;
  if (x3) {
    int rc = wgrad_x3_fused(x1, c1, pitch1, x2, c2, pitch2, dy, dypitch, amax_dy, ws, dw, dil, N, D, H, W, cout, st);
    if (rc == 1) {  // not a layer of the fused kernel: the split pass + three 16-bit launches
      const size_t slab = BRATS_API(brats_conv3d_wgrad_ws_bytes)(BRATS_BF16, 3, N, D, H, W, c1, c2, cout) * 3;
      rc = wgrad_x3(x1, c1, pitch1, x2, c2, pitch2, dy, dypitch, amax_dy, ws, dw, slab, dil, N, D, H, W, cout, st);
    }
    if (rc) return rc;
  } else {
    int nsplit = 0;
    const int rc = wgrad_mfma(x1, c1, pitch1, x2, c2, pitch2, dy, dypitch, ws, dtype, dil, N, D, H, W, cout, st, &nsplit);
    if (rc) return rc;
    wgrad_reduce_launch((const float*)ws, dw, nsplit, cout, c1 + c2, 27, st);
  }
  if (dbias) {
    const size_t vox = (size_t)N * D * H * W;
    if (dtype == BRATS_BF16) hipLaunchKernelGGL(dbias_kernel<bf16_t>, dim3(cout), dim3(256), 0, st, (const bf16_t*)dy, dypitch, dbias, vox, cout);
    else hipLaunchKernelGGL(dbias_kernel<float>, dim3(cout), dim3(256), 0, st, (const float*)dy, dypitch, dbias, vox, cout);
  }
  BRATS_CHECK_LAUNCH();
  return 0;
}

extern "C" int BRATS_API(brats_conv3d_wgrad)(const void* x1, int c1, int pitch1, const void* x2, int c2, int pitch2, const void* dy,
                                  int dypitch, float* ws, float* dw, float* dbias, int dtype, int ksize, int dil, int N,
                                  int D, int H, int W, int cout, brats_stream_t s) {
  return wgrad_impl(x1, c1, pitch1, x2, c2, pitch2, dy, dypitch, nullptr, ws, dw, dbias, dtype, ksize, dil, N, D, H, W, cout, s);
}
extern "C" int BRATS_API(brats_conv3d_x3_wgrad)(const void* x1, int c1, int pitch1, const void* x2, int c2, int pitch2, const void* dy,
                                     int dypitch, const float* amax_dy, float* ws, float* dw, float* dbias, int dtype, int dil,
                                     int N, int D, int H, int W, int cout, brats_stream_t s) {
  if (dtype != BRATS_X3_BF16) BRATS_FAIL(BRATS_E_ARG, "conv3d_x3_wgrad: dtype must be BRATS_X3_F16 or BRATS_X3_BF16");
  return wgrad_impl(x1, c1, pitch1, x2, c2, pitch2, dy, dypitch, amax_dy, ws, dw, dbias, dtype, 3, dil, N, D, H, W, cout, s);
}

// ---- shifted-tap form: 1x1x1 convolutions, and 3x3x3 convolutions at any dilation (ASPP d = 4, 6) ------------------
static int wgrad_shift_geometry(int dtype, int ksize, int N, int D, int H, int W, int cin, int cout, WgradParams* p, int* cof,
                                int* cif, int* cot, int* cit) {
  wgrad_tiles(dtype, cin, 0, cout, cof, cif);
  p->N = N; p->D = D; p->H = H; p->W = W; p->cin = cin; p->cout = cout;
  p->tz = ceil_div(D, WG_TZ); p->ty = ceil_div(H, WG_TY); p->tx = ceil_div(W, WG_TX);
  p->ntiles = N * p->tz * p->ty * p->tx;
  p->ntaps = ksize == 3 ? 27 : 1;
  *cot = ceil_div(cout, 16 * *cof);
  *cit = ceil_div(cin, 16 * *cif);
  p->nlane = wgrad_nlane(p->ntiles);
  int g8 = ceil_div(512, p->ntaps * p->nlane * *cot * *cit);
  const int cap = ceil_div(p->ntiles, p->nlane);
  if (g8 > cap) g8 = cap;
  if (g8 < 1) g8 = 1;
  p->nsplit = p->nlane * g8;
  return 0;
}

extern "C" size_t BRATS_API(brats_conv3d_wgrad_shift_ws_bytes)(int dtype, int ksize, int N, int D, int H, int W, int cin, int cout) {
  if (ksize != 1 && ksize != 3) return 0;
  if (dtype == BRATS_X3_BF16) {  // split precision: three groups of 16-bit slabs + the hi / lo tensors of x and dy (wgrad_shift_x3)
    const size_t vox = (size_t)N * D * H * W;
    const size_t slab = BRATS_API(brats_conv3d_wgrad_shift_ws_bytes)(BRATS_BF16, ksize, N, D, H, W, cin, cout) * 3;
    return x3_align(slab) + 2 * wgrad_x3_split_bytes(vox, cin) + 2 * wgrad_x3_split_bytes(vox, cout);
  }
  WgradParams p;
  int cof, cif, cot, cit;
  wgrad_shift_geometry(dtype, ksize, N, D, H, W, cin, cout, &p, &cof, &cif, &cot, &cit);
  return (size_t)p.nsplit * p.ntaps * cout * cin * sizeof(float);
}

// launches the shifted-tap MFMA kernel of one problem into the slabs at ws; *nsplit_out = slabs written ([split][taps][cout][cin])
static int wgrad_shift_mfma(const void* x, int cin, int xpitch, const void* dy, int dypitch, float* ws, int dtype, int ksize, int dil,
                            int N, int D, int H, int W, int cout, hipStream_t st, int* nsplit_out, int* ntaps_out) {
  WgradParams p;
  int cof, cif, cot, cit;
  wgrad_shift_geometry(dtype, ksize, N, D, H, W, cin, cout, &p, &cof, &cif, &cot, &cit);
  p.x1 = x; p.x2 = nullptr; p.c1 = cin; p.c2 = 0; p.p1 = xpitch; p.p2 = 0;
  p.dy = dy; p.dyp = dypitch; p.ws = ws; p.dil = dil;
  if (cin % 16 || cout % 16) {  // slab entries of padded ci / co lanes are never written
    hipError_t e = hipMemsetAsync(ws, 0, (size_t)p.nsplit * p.ntaps * cout * cin * sizeof(float), st);
    if (e != hipSuccess) BRATS_FAIL(BRATS_E_HIP, "wgrad_shift: memset: %s", hipGetErrorString(e));
  }
  dim3 grid(p.ntaps * p.nsplit, cot, cit);
  const int rc = dtype == BRATS_BF16 ? wgrad_dispatch<bf16_t, 1, 1>(p, cof, cif, grid, st) : wgrad_dispatch<float, 1, 1>(p, cof, cif, grid, st);
  if (rc) return rc;
  *nsplit_out = p.nsplit;
  *ntaps_out = p.ntaps;
  return 0;
}

static int wgrad_shift_impl(const void* x, int cin, int xpitch, const void* dy, int dypitch, const float* amax_dy, float* ws, float* dw,
                            float* dbias, int dtype, int ksize, int dil, int N, int D, int H, int W, int cout, brats_stream_t s) {
  if (!x || !dy || !ws || !dw || cin <= 0 || cout <= 0 || N <= 0 || D <= 0 || H <= 0 || W <= 0)
    BRATS_FAIL(BRATS_E_ARG, "wgrad_shift: null pointer / bad size");
  if ((ksize != 1 && ksize != 3) || dil < 1) BRATS_FAIL(BRATS_E_UNSUPPORTED, "wgrad_shift: ksize=%d dil=%d unsupported", ksize, dil);
  if (dtype != BRATS_BF16 && dtype != BRATS_F32 && dtype != BRATS_X3_BF16) BRATS_FAIL(BRATS_E_UNSUPPORTED, "wgrad_shift: dtype %d", dtype);
  const bool x3 = dtype == BRATS_X3_BF16;
  const int epl = dtype == BRATS_BF16 ? 8 : 4;
  if (xpitch % epl || dypitch % epl || cin % epl || cout % epl)
    BRATS_FAIL(BRATS_E_ARG, "wgrad_shift: channel counts / pitches must be multiples of %d", epl);
  if (x3 && (cin % 8 || cout % 8)) BRATS_FAIL(BRATS_E_ARG, "wgrad_shift (split precision): channel counts must be multiples of 8");
  {
    const int mp = xpitch > dypitch ? xpitch : dypitch;
    if ((double)D * H * W * mp * (dtype == BRATS_BF16 ? 2 : 4) >= 2147483648.0)
      BRATS_FAIL(BRATS_E_UNSUPPORTED, "wgrad_shift: one sample of %dx%dx%d x pitch %d exceeds the 2 GiB buffer-offset range", D, H, W, mp);
  }
  hipStream_t st = (hipStream_t)s;
  int ns = 0, ntaps = 0;
  if (x3) {
    // split precision (see wgrad_x3): hi / lo tensors of x and of dy * 2^k, three runs of the 16-bit kernel, one reduction
    const size_t vox = (size_t)N * D * H * W;
    const size_t slab = BRATS_API(brats_conv3d_wgrad_shift_ws_bytes)(BRATS_BF16, ksize, N, D, H, W, cin, cout) * 3;
    char* b = (char*)ws + x3_align(slab);
    bf16_t* hx = (bf16_t*)b; b += wgrad_x3_split_bytes(vox, cin);
    bf16_t* lx = (bf16_t*)b; b += wgrad_x3_split_bytes(vox, cin);
    bf16_t* hy = (bf16_t*)b; b += wgrad_x3_split_bytes(vox, cout);
    bf16_t* ly = (bf16_t*)b;
    x3_split_launch((const float*)x, xpitch, hx, lx, vox, cin, nullptr, st);
    x3_split_launch((const float*)dy, dypitch, hy, ly, vox, cout, amax_dy, st);
    BRATS_CHECK_LAUNCH();
    int total = 0;
    const size_t per = (size_t)(ksize == 3 ? 27 : 1) * cout * cin;
    const bf16_t* xs[3] = {lx, hx, hx};
    const bf16_t* ys[3] = {hy, ly, hy};
    for (int t = 0; t < 3; ++t) {  // (small terms first: the reduction adds the slabs in this order)
      const int rc = wgrad_shift_mfma(xs[t], cin, cin, ys[t], cout, ws + (size_t)total * per, BRATS_BF16, ksize, dil, N, D, H, W, cout, st, &ns, &ntaps);
      if (rc) return rc;
      total += ns;
    }
    wgrad_reduce_launch((const float*)ws, dw, total, cout, cin, ntaps, st, amax_dy);
  } else {
    const int rc = wgrad_shift_mfma(x, cin, xpitch, dy, dypitch, ws, dtype, ksize, dil, N, D, H, W, cout, st, &ns, &ntaps);
    if (rc) return rc;
    wgrad_reduce_launch((const float*)ws, dw, ns, cout, cin, ntaps, st);
  }
  if (dbias) {
    const size_t vox = (size_t)N * D * H * W;
    if (dtype == BRATS_BF16) hipLaunchKernelGGL(dbias_kernel<bf16_t>, dim3(cout), dim3(256), 0, st, (const bf16_t*)dy, dypitch, dbias, vox, cout);
    else hipLaunchKernelGGL(dbias_kernel<float>, dim3(cout), dim3(256), 0, st, (const float*)dy, dypitch, dbias, vox, cout);
  }
  BRATS_CHECK_LAUNCH();
  return 0;
}

extern "C" int BRATS_API(brats_conv3d_wgrad_shift)(const void* x, int cin, int xpitch, const void* dy, int dypitch, float* ws, float* dw,
                                        float* dbias, int dtype, int ksize, int dil, int N, int D, int H, int W, int cout,
                                        brats_stream_t s) {
  return wgrad_shift_impl(x, cin, xpitch, dy, dypitch, nullptr, ws, dw, dbias, dtype, ksize, dil, N, D, H, W, cout, s);
}
extern "C" int BRATS_API(brats_conv3d_x3_wgrad_shift)(const void* x, int cin, int xpitch, const void* dy, int dypitch, const float* amax_dy,
                                           float* ws, float* dw, float* dbias, int dtype, int ksize, int dil, int N, int D, int H,
                                           int W, int cout, brats_stream_t s) {
  if (dtype != BRATS_X3_BF16) BRATS_FAIL(BRATS_E_ARG, "conv3d_x3_wgrad_shift: dtype must be BRATS_X3_F16 or BRATS_X3_BF16");
  return wgrad_shift_impl(x, cin, xpitch, dy, dypitch, amax_dy, ws, dw, dbias, dtype, ksize, dil, N, D, H, W, cout, s);
}

// ---- e4m3 weight gradient (all-taps blocks only; everything else stays on the bf16 kernels) --------------------------
// block shape (co fragments, ci fragments) and workgroups per XCD range; false = not built for this layer
static bool wgrad_f8_shape(int N, int D, int H, int W, int c1, int c2, int cout, int* cof, int* cif, int* g8_out, int* nl_out, int* ntiles_out) {
  static int ncu = 0;
  if (!ncu) {
    int dev = 0;
    hipDeviceProp_t prop;
    ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  if (c2 < 0) c2 = 0;
  if (c1 <= 0 || cout <= 0) return false;
  if (cout % 48 == 0 && c1 % 48 == 0 && c2 % 48 == 0) { *cof = 3; *cif = 3; }
  else if (cout % 64 == 0 && c1 % 32 == 0 && c2 % 32 == 0) { *cof = 4; *cif = 2; }
  else return false;
  const int ntiles = N * ceil_div(D, WG_TZ) * ceil_div(H, WG_TY) * ceil_div(W, WG_TX);
  const int blocks = (cout / (16 * *cof)) * ((c1 + c2) / (16 * *cif));
  const int nl = wgrad_nlane(ntiles);
  int g8 = ceil_div(ncu, nl * blocks);
  if (g8 < 1) g8 = 1;
  if (ntiles < 4 * nl * g8) return false;  // too few tiles per workgroup to amortise the 27-tap slab
  *g8_out = g8; *nl_out = nl; *ntiles_out = ntiles;
  return true;
}

extern "C" size_t BRATS_API(brats_conv3d_wgrad_f8_ws_bytes)(int N, int D, int H, int W, int c1, int c2, int cout) {
  int cof, cif, g8, nl, nt;
  if (!wgrad_f8_shape(N, D, H, W, c1, c2, cout, &cof, &cif, &g8, &nl, &nt)) return 0;  // 0 = not supported: use brats_conv3d_wgrad
  return (size_t)nl * g8 * 27 * cout * (c1 + (c2 > 0 ? c2 : 0)) * sizeof(float);
}

template <int COF, int CIF>
static int wgrad_f8_launch(const WgradF8Params& pp, hipStream_t st) {
  using G = Wg3f<COF, CIF>;
  auto kern = conv_wgrad_alltaps_f8_kernel<COF, CIF>;
  static std::atomic<uint64_t> attr_done{0};
  BRATS_ENSURE_LDS_ATTR(kern, G::LDS, attr_done);
  hipLaunchKernelGGL(kern, dim3(pp.w.nsplit, pp.w.cout / G::CO, pp.w.cin / G::CI), dim3(512), G::LDS, st, pp);
  BRATS_CHECK_LAUNCH();
  return 0;
}

extern "C" int BRATS_API(brats_conv3d_wgrad_f8)(const void* x1, int c1, int pitch1, const float* amax1, const void* x2, int c2, int pitch2,
                                     const float* amax2, const void* dy, int dypitch, const float* amax_dy, float* ws, float* dw,
                                     int N, int D, int H, int W, int cout, brats_stream_t s) {
  if (!x1 || !dy || !ws || !dw || !amax1 || !amax_dy || c1 <= 0 || cout <= 0) BRATS_FAIL(BRATS_E_ARG, "wgrad_f8: null pointer / bad size");
  if (c2 < 0) c2 = 0;
  if (c2 > 0 && (!x2 || !amax2)) BRATS_FAIL(BRATS_E_ARG, "wgrad_f8: c2 > 0 needs x2 and its |max|");
  int cof, cif, g8, nl, nt;
  if (!wgrad_f8_shape(N, D, H, W, c1, c2, cout, &cof, &cif, &g8, &nl, &nt))
    BRATS_FAIL(BRATS_E_UNSUPPORTED, "wgrad_f8: built for 48 x 48 / 64 x 32 channel blocks of layers with enough tiles "
               "(brats_conv3d_wgrad_f8_ws_bytes() == 0 otherwise): use brats_conv3d_wgrad");
  if (pitch1 % 8 || (c2 && pitch2 % 8) || dypitch % 8) BRATS_FAIL(BRATS_E_ARG, "wgrad_f8: pitches must be multiples of 8");
  {
    const int mp = pitch1 > pitch2 ? (pitch1 > dypitch ? pitch1 : dypitch) : (pitch2 > dypitch ? pitch2 : dypitch);
    if ((double)D * H * W * mp * 2 >= 2147483648.0) BRATS_FAIL(BRATS_E_UNSUPPORTED, "wgrad_f8: sample exceeds the 2 GiB buffer-offset range");
  }
  WgradF8Params pp;
  WgradParams& p = pp.w;
  p.x1 = x1; p.x2 = x2; p.c1 = c1; p.c2 = c2; p.p1 = pitch1; p.p2 = pitch2;
  p.dy = dy; p.dyp = dypitch; p.ws = ws;
  p.N = N; p.D = D; p.H = H; p.W = W; p.cin = c1 + c2; p.cout = cout;
  p.tz = ceil_div(D, WG_TZ); p.ty = ceil_div(H, WG_TY); p.tx = ceil_div(W, WG_TX);
  p.ntiles = nt;
  p.nlane = nl;
  p.ntaps = 27; p.dil = 1;
  p.nsplit = nl * g8;
  pp.amax_x1 = amax1; pp.amax_x2 = c2 ? amax2 : nullptr; pp.amax_dy = amax_dy;
  hipStream_t st = (hipStream_t)s;
  const int rc = cof == 3 ? wgrad_f8_launch<3, 3>(pp, st) : wgrad_f8_launch<4, 2>(pp, st);
  if (rc) return rc;
  wgrad_reduce_launch((const float*)ws, dw, p.nsplit, cout, p.cin, 27, st);
  BRATS_CHECK_LAUNCH();
  return 0;
}
#include "twin_end.hpp"
