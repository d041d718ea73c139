// Split-precision ("x3") weight gradient with the split in the staging path (round 5; included by conv_wgrad.hip).
//
//     dW[tap][co][ci] = sum_v dY[v][co] * X[v + off(tap)][ci]      (autograd of nn.Conv3d, networks/equiunet2020.py:19-25)
//
// on f32 tensors as three 16-bit MFMA products per MAC: x = hi + lo with hi = rn16(x), lo = rn16(x - hi);
// hi*hi + hi*lo + lo*hi drops only lo*lo (<= 2^-22 of the product; error analysis: conv_igemm_x3.hpp).
//
// Round 4 (wgrad_x3 in conv_wgrad.hip, still the path of every shape this kernel does not take) wrote the hi / lo tensors
// of X and dY to HBM in a streaming pass and ran the UNCHANGED 16-bit kernels three times, each re-staging its tiles:
// 14.1 of the 40.3 ms parity-mode step at 0.34 of the MFMA ceiling.  Here ONE launch stages each f32 value once:
//   * one 8-wave workgroup per CU owns all 27 taps of a HALF tile (2 x 4 x 16 voxels; the 16-bit all-taps tile, 4 x 4 x 16,
//     would need 173 KB of LDS as hi + lo) and WALKS A COLUMN of such tiles along z: the X halo box of a tile is 4 z-planes
//     of 6 x 18 voxels, of which the next tile down the column re-uses two -- the planes live in a ring of four LDS slots
//     (slot = (z + 1) & 3), so a tile costs 2 new planes (41.5 KB of f32) + its dY tile (24.6 KB) instead of 108 KB;
//   * the f32 pieces (16 bytes = 4 channels of a voxel) are split in registers -- dY times the power of two that puts its
//     recorded |max| into fp16's top binades -- and written as FOUR 16-bit LDS tiles (X hi, X lo: 4 planes x 10.4 KB each;
//     dY hi, dY lo: 12.3 KB each; 107.5 KB) whose rows are the lane-linear image of the piece order: conflict-free
//     ds_write_b64 runs;
//   * the NEXT tile's pieces are fetched during the MFMA phase without costing the 60 registers a register prefetch needs
//     (that form spilled 93-147 registers and ran at 165 TF/s; a 4-wave / 512-register form at 236): dY and the first 128
//     columns of every new X row arrive by LDS-DMA in a 48 KB staging area beside the tiles, only the rows' tails in registers.
//     WHO issues them matters (s_memtime stamps, scripts/probes/x3w_stamps.py): an LDS-DMA instruction stalls its wave for
//     150-300 cycles, and with every wave issuing its share in front of its MFMA loop the matrix pipe idled 2 k of a tile's
//     18.5 k cycles.  Waves 0-3 ("group A") win the matrix pipe's arbitration anyway (older wave slots; s_setprio on top changes
//     nothing), finish their MFMA loop while waves 4-7 -- their SIMD partners -- still have a third of theirs to go, and issue
//     ALL the next tile's loads in that shadow (1.28 -> 1.23 ms at 48 -> 48 @2x128^3).  What is left (stamps of that layer: a
//     tile = 17.5 k cycles): group B's MFMA loop 14.7 k (two waves sharing a SIMD reach 87 % of the pipe rate; a wave alone
//     64 %, whichever group goes first: alternating the priority in time measured 8-19 % SLOWER), the split + tile writes of
//     66 KB 2.4 k (VALU-bound, and not overlappable without a second set of tiles LDS has no room for);
//   * MFMA phase = the all-taps kernel's roles (81 (tap, ci-fragment) pairs dealt to 8 waves, 3 co fragments each, 132
//     accumulator registers kept over everything the workgroup walks), operands by the transposing LDS read from the hi / lo
//     tiles (96-byte voxel stride: conflict-free), 9 MFMAs per (k-step, pair): a_hi*b_lo, a_lo*b_hi, a_hi*b_hi per co fragment;
//   * slabs and the fixed-order reduction (times 2^-k) are the 16-bit kernels' (bitwise reproducible).
// CIF = 3: 48-channel ci blocks; CIF = 1: the first layer (<= 16 input channels; 77 KB of LDS, 128 registers: two workgroups per CU).
#pragma once
#ifndef X3_PRIO
#define X3_PRIO 2  // s_setprio of waves 0-3 during their MFMA loop
#endif

template <int CIF> struct Wg3z {
  static constexpr int NW = 8;                                                            // waves per workgroup
  static constexpr int TZ = 2, TY = 4, TX = 16, VOX = TZ * TY * TX;                       // 128 voxels = 8 x-rows = 4 k-steps
  static constexpr int HY = TY + 2, HX = TX + 2;                                          // a halo plane: 6 x 18 voxels
  static constexpr int CI = 16 * CIF, CO = 48;
  static constexpr int SX = 2 * CI, SY = 2 * CO;                                          // bytes per voxel in a 16-bit tile: 96 (32), 96
  static constexpr int XPPV = CI / 4, YPPV = CO / 4;                                      // 16-byte f32 pieces per voxel: 12 (4), 12
  static constexpr int XPR = HX * XPPV, YPR = TX * YPPV;                                  // pieces per row: 216 (72), 192
  static constexpr int XROWB = HX * SX, PLANE = HY * XROWB;                               // bytes of a row / plane of ONE of hi, lo: 1728, 10368
  static constexpr int LDS_XH = 4 * PLANE, LDS_YH = VOX * SY;                             // 41472 (13824), 12288
  static constexpr int TILES = 2 * LDS_XH + 2 * LDS_YH;                                   // 107520 (52224)
  // a new X row = XPR pieces = XJ loads of 64 lanes; the first XJD of them by LDS-DMA, the other XJR through registers
  static constexpr int XJ = (XPR + 63) / 64, XJD = CIF == 3 ? 2 : 0, XJR = XJ - XJD;      // 4 = 2 + 2 | 2 = 0 + 2
  static_assert(XJR == 2 && (XJD == 0 || XJD == 2) && YPR == 192, "");
  static constexpr int STAGE_Y = NW * 3 * 1024, STAGE_X = XJD ? NW * 3 * 1024 : 0;        // 1 KB per wave and DMA instruction
  static constexpr int LDS = TILES + STAGE_Y + STAGE_X;                                   // 156672 (76800)
  static constexpr int PAIRS = 27 * CIF, PPW = (PAIRS + NW - 1) / NW;                     // 81, 11 | 27, 4
};

// 4 consecutive f32 channels (one 16-byte piece) times the power of two sc -> 4 hi + 4 lo 16-bit values (8 bytes each)
DEVI void x3_split4(const u32x4 a, float sc, u32x2& hi, u32x2& lo) {
  const uint32_t w[4] = {a[0], a[1], a[2], a[3]};  // (through scalars: see f8_quant8)
  uint32_t h[2], l[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float x0 = __uint_as_float(w[2 * i]) * sc, x1 = __uint_as_float(w[2 * i + 1]) * sc;
    x3_split2(x0, x1, h[i], l[i]);
  }
  hi = u32x2{h[0], h[1]};
  lo = u32x2{l[0], l[1]};
}

// the lane number from the hardware, re-derived wherever it is needed (volatile: never kept): held in a register across the MFMA
// phase hipcc SPILLS it, and its scratch re-load -- waited for with vmcnt(0) right behind a barrier -- costs ~1 us twice per tile
DEVI int lane_now() {
  int l = 0;
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
#endif
  return l;
}

template <int CIF>
__global__ __launch_bounds__(512, CIF == 1 ? 4 : 2) void conv_wgrad_x3_zwalk_kernel(const WgradParams p, const float* __restrict__ amax_dy) {
  using G = Wg3z<CIF>;
  constexpr int NW = G::NW;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* const ldxh = lds;
  char* const ldxl = lds + G::LDS_XH;
  char* const ldyh = lds + 2 * G::LDS_XH;
  char* const ldyl = ldyh + G::LDS_YH;
  char* const stage_y = lds + G::TILES;
  char* const stage_x = stage_y + G::STAGE_Y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, v = lane & 15;
  // segments (column pieces of p.seglen tiles) are dealt like the tiles of the other kernels: nlane XCD-contiguous ranges
  const int lane8 = blockIdx.x % p.nlane, gsub = blockIdx.x / p.nlane, g8 = gridDim.x / p.nlane;
  const int split = blockIdx.x;
  const int spx = (p.ntiles + p.nlane - 1) / p.nlane;  // (ntiles = number of segments here)
  const int seg_end = min(p.ntiles, (lane8 + 1) * spx);
  const int co0 = blockIdx.y * G::CO, ci0 = blockIdx.z * G::CI;
  const int ci_lim = (ci0 < p.c1 ? p.c1 : p.c1 + p.c2) - ci0;  // valid channels of this source in the block
  const float* xsrc;
  int xpitch;
  if (ci0 < p.c1) { xsrc = (const float*)p.x1 + ci0; xpitch = p.p1; }
  else { xsrc = (const float*)p.x2 + (ci0 - p.c1); xpitch = p.p2; }
  const float ysc = amax_dy ? x3_scale_from_amax(*amax_dy) : 1.f;
  const unsigned xsample_bytes = (unsigned)p.D * p.H * p.W * xpitch * 4;  // < 2^31, checked by the host
  const unsigned ysample_bytes = (unsigned)p.D * p.H * p.W * p.dyp * 4;
  const int xpb = xpitch * 4, ypb = p.dyp * 4;

  // the wave's (tap, ci-fragment) pairs: pid = wave + 8 jj is wave-uniform -> scalar registers.  tapz = the tap's z offset
  // (the ring slot of its plane depends on the tile), pin = its byte offset inside a plane
  int tapz[G::PPW], pin[G::PPW];
#pragma unroll
  for (int jj = 0; jj < G::PPW; ++jj) {
    const int pid = wave + NW * jj;
    const int t = pid / CIF, nn = pid % CIF;
    tapz[jj] = pid < G::PAIRS ? t / 9 : 0;
    pin[jj] = pid < G::PAIRS ? (((t / 3) % 3) * G::HX + t % 3) * G::SX + nn * 32 : 0;
  }
  f32x4 acc[G::PPW][3];
#pragma unroll
  for (int jj = 0; jj < G::PPW; ++jj)
#pragma unroll
    for (int m = 0; m < 3; ++m) acc[jj][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- staging roles.  A tile brings 12 new X rows (2 planes x 6) and 8 dY rows (wave w: dY row w, three 64-lane loads).
  // X: a row is XJ loads of 64 lanes; load j of row r is unit (r, j); class D (j < XJD, by LDS-DMA) and class R (the others,
  // through registers) have 24 units each = 3 per wave: wave w takes load j0 + (w & 1) of rows (w >> 1) + 4 k, k = 0..2 -- so a
  // lane's column inside its rows is ONE constant per class.
  // The lane constants of a class are RE-DERIVED from the lane number where they are used (a dozen VALU instructions per tile):
  // kept in registers across the MFMA phase hipcc spills them, and a scratch re-load between the tile's buffer_loads waits
  // (vmcnt is in-order) for every global load issued before it -- the loads of a tile then go out one DRAM latency at a time.
  struct LaneX { int col, hx, lds; };
  auto lane_x = [&](int j) {  // load j of a row: piece lane + 64 j
    const int l = lane_now();
    const int pc = l + 64 * j;
    const int hx = pc / G::XPPV, part = pc - hx * G::XPPV;
    LaneX r;
    r.col = hx * xpb + part * 16;                                  // byte offset of the piece inside a global row
    r.hx = (pc < G::XPR && part * 4 < ci_lim) ? hx : 31;           // halo column for the range test (31: never valid)
    r.lds = pc * 8;                                                // byte offset inside a hi / lo tile row
    return r;
  };

  struct Col {  // the column a segment belongs to (scalars)
    __amdgpu_buffer_rsrc_t xrs, yrs;
    rsrc4_t xrs4, yrs4;
    int x0, y0;
    unsigned xm;  // bit h: halo column h is inside the volume (bit 31 stays clear)
  };
  auto column = [&](int col) {
    Col c;
    c.x0 = (col % p.tx) * G::TX; col /= p.tx;
    c.y0 = (col % p.ty) * G::TY;
    const int n = col / p.ty;
    const size_t sample_vox = (size_t)n * p.D * p.H * p.W;
    const float* xb = xsrc + sample_vox * xpitch;
    const float* yb = (const float*)p.dy + sample_vox * p.dyp + co0;
    c.xrs = __builtin_amdgcn_make_buffer_rsrc((void*)xb, (short)0, (int)xsample_bytes, 0x00020000);
    c.yrs = __builtin_amdgcn_make_buffer_rsrc((void*)yb, (short)0, (int)ysample_bytes, 0x00020000);
    c.xrs4 = make_rsrc4(xb, xsample_bytes);
    c.yrs4 = make_rsrc4(yb, ysample_bytes);
    c.xm = 0;
#pragma unroll
    for (int h = 0; h < G::HX; ++h) c.xm |= ((unsigned)(c.x0 - 1 + h) < (unsigned)p.W ? 1u : 0u) << h;
    return c;
  };
  // byte offset of X row (gz, hy) of the column's halo box, or INT_MIN-like "never valid" (-1 after the OR below)
  auto xrow = [&](const Col& c, int gz, int hy, bool& ok) {
    const int gy = c.y0 - 1 + hy;
    ok = (unsigned)gz < (unsigned)p.D && (unsigned)gy < (unsigned)p.H;
    return ((gz * p.H + gy) * p.W + (c.x0 - 1)) * xpb;  // (may be negative at the low faces)
  };

  u32x4 rxr[2][3];  // the register pieces (group A only: [own share | the partner wave's][row])
  // loads of the two NEW planes z0 + 1, z0 + 2 and of the dY tile at z0: the share of wave ww (0..7), issued by a group-A wave
  // for itself (h = 0) and for its partner ww = wave + 4 (h = 1).  DMA'd pieces land in ww's staging slots.
  auto issue_for = [&](const Col& c, int z0, int ww, auto h_) {
    constexpr int h = h_;
    const LaneX LD = lane_x(ww & 1), LR = lane_x(G::XJD + (ww & 1));
    const bool okD = G::XJD && ((c.xm >> LD.hx) & 1u), okR = (c.xm >> LR.hx) & 1u;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int r = (ww >> 1) + 4 * k;  // 0..11
      bool row_ok;
      const int rb = xrow(c, z0 + 1 + r / G::HY, r % G::HY, row_ok);
      if (G::XJD) lds_dma16_async(c.xrs4, stage_x + (ww * 3 + k) * 1024, (row_ok && okD) ? rb + LD.col : -1);
      rxr[h][k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(c.xrs, (row_ok && okR) ? rb + LR.col : -1, 0, 0));
    }
    {
      const int gz = z0 + (ww >> 2), gy = c.y0 + (ww & 3);
      const bool row_ok = gz < p.D && gy < p.H;
      const int rb = ((gz * p.H + gy) * p.W + c.x0) * ypb;
      const int l = lane_now();
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int pc = l + 64 * j, vx = pc / G::YPPV;
        const int vo = (row_ok && ((c.xm >> (vx + 1)) & 1u)) ? rb + vx * ypb + (pc - vx * G::YPPV) * 16 : -1;
        lds_dma16_async(c.yrs4, stage_y + (ww * 3 + j) * 1024, vo);
      }
    }
  };
  auto issue_tile = [&](const Col& c, int z0) {  // group A only
    issue_for(c, z0, wave, std::integral_constant<int, 0>{});
    issue_for(c, z0, wave + 4, std::integral_constant<int, 1>{});
  };
  // split + LDS tile writes of what issue_tile fetched; sb = ring slot of the first new plane.  Nine pieces per wave: group A
  // converts its six register pieces and its own three DMA'd X pieces, group B its own DMA'd X and dY pieces and the dY pieces
  // of its partner (after the barrier behind group A's vmcnt(0) every staging slot is readable by everybody).
  auto put_x = [&](const u32x4 d, int ww, int k, int ldsoff, int sb) {
    const int r = (ww >> 1) + 4 * k;
    const int rowoff = (((sb + r / G::HY) & 3) * G::HY + r % G::HY) * G::XROWB;  // scalar
    u32x2 hi, lo;
    x3_split4(d, 1.f, hi, lo);
    *(u32x2*)(ldxh + rowoff + ldsoff) = hi;
    *(u32x2*)(ldxl + rowoff + ldsoff) = lo;
  };
  auto put_y = [&](int ww, int l) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const u32x4 d = *(const u32x4*)(stage_y + (ww * 3 + j) * 1024 + l * 16);
      u32x2 hi, lo;
      x3_split4(d, ysc, hi, lo);
      const int o = (ww * G::YPR + l + 64 * j) * 8;
      *(u32x2*)(ldyh + o) = hi;
      *(u32x2*)(ldyl + o) = lo;
    }
  };
  auto convert_tile = [&](int sb) {
    const int l = lane_now();
    if (G::XJD) {
      const int xldsD = (l + 64 * (wave & 1)) * 8;
#pragma unroll
      for (int k = 0; k < 3; ++k) put_x(*(const u32x4*)(stage_x + (wave * 3 + k) * 1024 + l * 16), wave, k, xldsD, sb);
    }
    if (wave < 4) {
      const int xldsR = (l + 64 * (G::XJD + (wave & 1))) * 8;  // (wave and wave + 4 have the same parity: the same column)
      if (xldsR < G::XROWB) {  // (the last load of a row is partly beyond it)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          put_x(rxr[0][k], wave, k, xldsR, sb);
          put_x(rxr[1][k], wave + 4, k, xldsR, sb);
        }
      }
    } else {
      put_y(wave, l);
      put_y(wave - 4, l);
    }
  };

#ifdef BRATS_X3W_STAMPS  // diagnostic build only (scripts/probes/x3w_stamps.*): where does a tile's time go?
  long long tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
#define X3_STAMP(i) do { const long long t_ = __builtin_amdgcn_s_memtime(); tacc[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define X3_STAMP(i) do { } while (0)
#endif
  for (int seg = lane8 * spx + gsub; seg < seg_end; seg += g8) {
    const Col c = column(seg / p.nsegz);
    const int t0 = (seg % p.nsegz) * p.seglen, t1 = min(p.tz, t0 + p.seglen);
    // ---- segment prologue: the two planes below the first tile's new ones (z0 - 1, z0), all loads through registers ----
    {
      u32x4 pr[2][3];
      const LaneX LD = lane_x(wave & 1), LR = lane_x(G::XJD + (wave & 1));
      const bool okD = G::XJD && ((c.xm >> LD.hx) & 1u), okR = (c.xm >> LR.hx) & 1u;
      const int xldsD = LD.lds, xldsR = LR.lds;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int r = (wave >> 1) + 4 * k;
        bool row_ok;
        const int rb = xrow(c, 2 * t0 - 1 + r / G::HY, r % G::HY, row_ok);
        if (G::XJD) pr[0][k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(c.xrs, (row_ok && okD) ? rb + LD.col : -1, 0, 0));
        pr[1][k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(c.xrs, (row_ok && okR) ? rb + LR.col : -1, 0, 0));
      }
      __syncthreads();  // the previous segment's last MFMA phase is done with the ring
      const int sb = 2 * (t0 & 1);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int r = (wave >> 1) + 4 * k;
        const int rowoff = (((sb + r / G::HY) & 3) * G::HY + r % G::HY) * G::XROWB;
        if (G::XJD) {
          u32x2 hi, lo;
          x3_split4(pr[0][k], 1.f, hi, lo);
          *(u32x2*)(ldxh + rowoff + xldsD) = hi;
          *(u32x2*)(ldxl + rowoff + xldsD) = lo;
        }
        if (xldsR < G::XROWB) {
          u32x2 hi, lo;
          x3_split4(pr[1][k], 1.f, hi, lo);
          *(u32x2*)(ldxh + rowoff + xldsR) = hi;
          *(u32x2*)(ldxl + rowoff + xldsR) = lo;
        }
      }
    }
    // fetch (group A) -> everybody's pieces have landed and the previous MFMA phase is over -> split into the tiles.  (The loop
    // is written MFMA-first so that the register pieces are loaded and consumed inside ONE iteration: with their live range
    // across the back edge hipcc spilled each of them right behind its load.)
    auto stage_tile = [&](int t) {
      // (defined on both paths: with the pieces undefined for group B hipcc spills each of them right behind its load)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int k = 0; k < 3; ++k) rxr[h][k] = u32x4{0u, 0u, 0u, 0u};
      if (wave < 4) issue_tile(c, 2 * t);  // (t > t0: in the shadow of group B's MFMA loop)
      X3_STAMP(3);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // group A: the DMA pieces have landed (the asm form is invisible to hipcc's counters)
      X3_STAMP(0);
      __syncthreads();  // the previous tile's LDS reads are done (the prologue's writes went to other ring slots)
      X3_STAMP(1);
      convert_tile(2 * ((t + 1) & 1));
      __syncthreads();
      X3_STAMP(2);
    };
    stage_tile(t0);
    X3_STAMP(5);
    for (int t = t0; t < t1; ++t) {
      if (wave < 4) __builtin_amdgcn_s_setprio(X3_PRIO);  // group A first at the matrix pipe: it has the next tile's loads to issue afterwards
      __builtin_amdgcn_sched_barrier(0);
      // ---- MFMA over the 128 voxels of the tile.  k-step s = x-rows 2s, 2s+1 (row = z*4 + y); fragment layout: see
      // conv_wgrad_kernel.  The plane of (tile z = s >> 1, tap z) sits in ring slot (2 (t & 1) + (s >> 1) + tapz) & 3.
      const int tpar = 2 * (t & 1);
      // per-read addresses = one v_add of a scalar sum onto these two (re-derived per tile: nothing to hoist, nothing to spill)
      const int lm = lane_now();
      const int vox8 = 4 * (lm >> 4) + ((lm & 15) >> 2);  // 4 q + qq: the voxel of the lane inside an 8-voxel group pair
      const int xb = vox8 * G::SX + (lm & 3) * 8, yb = vox8 * G::SY + (lm & 3) * 8;
#ifndef X3_PD
#define X3_PD 2
#endif
#ifndef X3_AB
#define X3_AB 2
#endif
      constexpr int PD = CIF == 1 ? 2 : X3_PD;             // X fragment pairs in flight
      constexpr int AB = CIF == 1 ? 1 : X3_AB;             // dY fragment buffers
      bf16x8 ah[AB][3], al[AB][3], bh[PD + 1], bl[PD + 1];
      auto read_a = [&](auto s_) {
        constexpr int s = s_;
        const int yoff = yb + (32 * s) * G::SY;
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          ah[s % AB][m] = tr_pair(ldyh + yoff + m * 32, ldyh + yoff + 16 * G::SY + m * 32);
          al[s % AB][m] = tr_pair(ldyl + yoff + m * 32, ldyl + yoff + 16 * G::SY + m * 32);
        }
      };
      auto read_b = [&](auto u_) {
        constexpr int u = u_;
        constexpr int s = u / G::PPW, jj = u % G::PPW;
        const int xoff = xb + (((tpar + (s >> 1) + tapz[jj]) & 3) * G::PLANE + (2 * (s & 1)) * G::XROWB + pin[jj]);
        bh[u % (PD + 1)] = tr_pair(ldxh + xoff, ldxh + xoff + G::XROWB);
        bl[u % (PD + 1)] = tr_pair(ldxl + xoff, ldxl + xoff + G::XROWB);
      };
      constexpr int NU = 4 * G::PPW;
      read_a(std::integral_constant<int, 0>{});
      static_for<0, PD>([&](auto u_) { read_b(u_); });
      static_for<0, NU>([&](auto u_) {
        constexpr int u = u_;
        constexpr int s = u / G::PPW, jj = u % G::PPW;
        if constexpr (u + PD < NU) read_b(std::integral_constant<int, u + PD>{});
        if constexpr (AB == 2 && jj == 0 && s + 1 < 4) read_a(std::integral_constant<int, s + 1>{});  // a whole k-step ahead
        __builtin_amdgcn_sched_barrier(0);
        // term-major: an accumulator is revisited after two other MFMAs; the small terms first
#pragma unroll
        for (int m = 0; m < 3; ++m) acc[jj][m] = MFMA16_16x16x32(ah[s % AB][m], bl[u % (PD + 1)], acc[jj][m]);
#pragma unroll
        for (int m = 0; m < 3; ++m) acc[jj][m] = MFMA16_16x16x32(al[s % AB][m], bh[u % (PD + 1)], acc[jj][m]);
#pragma unroll
        for (int m = 0; m < 3; ++m) acc[jj][m] = MFMA16_16x16x32(ah[s % AB][m], bh[u % (PD + 1)], acc[jj][m]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (AB == 1 && jj == G::PPW - 1 && s + 1 < 4) read_a(std::integral_constant<int, s + 1>{});  // (single buffer: after its last use)
      });
      __builtin_amdgcn_s_setprio(0);
      X3_STAMP(4);
      if (t + 1 < t1) stage_tile(t + 1);
    }
  }
#ifdef BRATS_X3W_STAMPS
  if (lane_now() == 0 && blockIdx.y == 0 && blockIdx.z == 0) {  // behind the slabs: [split][wave][6] cycle sums
    long long* st = (long long*)(p.ws + (size_t)gridDim.x * 27 * p.cout * p.cin) + ((size_t)blockIdx.x * 8 + wave) * 6;
    for (int i = 0; i < 6; ++i) st[i] = tacc[i];
  }
#endif

  // ---- slab: ws[split][tap][co][ci] (the scale 2^k of dY is undone by the reduction) ----
#pragma unroll
  for (int jj = 0; jj < G::PPW; ++jj) {
    const int pid = wave + NW * jj;
    if (pid < G::PAIRS) {
      const int t = pid / CIF, nn = pid % CIF;
      float* base = p.ws + ((size_t)split * 27 + t) * p.cout * p.cin;
      const int ci = ci0 + nn * 16 + v;
      if (nn * 16 + v < ci_lim) {
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
          for (int r = 0; r < 4; ++r) base[(size_t)(co0 + m * 16 + 4 * q + r) * p.cin + ci] = acc[jj][m][r];
      }
    }
  }
}

// Is the fused kernel built for this layer?  dilation 1, Cout a multiple of 48, input channels in 48-blocks per source (or the
// first layer: one source of <= 16 channels), enough half tiles per workgroup to amortise its 27-tap slab.
int g_x3_wgrad_fused_mode = -1;  // brats_conv3d_set_x3_wgrad_fused(): -1 = environment / default (1), 0 = off, 1 = on, 2 = any tile count
struct X3Shape { int nl, g8, nseg, nsegz, seglen, narrow; };
static bool wgrad_x3_fused_shape(int dil, int N, int D, int H, int W, int c1, int c2, int cout, X3Shape* o) {
  static int env_mode = -1, ncu = 0;
  if (env_mode < 0) {
    const char* e = getenv("BRATS_X3_WGRAD_FUSED");  // 0: round 4's split pass + three 16-bit launches, for same-box A/B runs
    env_mode = e ? atoi(e) : 1;
    int dev = 0;
    hipDeviceProp_t prop;
    ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  const int mode = g_x3_wgrad_fused_mode >= 0 ? g_x3_wgrad_fused_mode : env_mode;
  if (!mode || dil != 1 || cout % 48) return false;
  if (c2 < 0) c2 = 0;
  const bool narrow = c2 == 0 && c1 <= 16;
  if (!narrow && (c1 % 48 || c2 % 48)) return false;
  using G = Wg3z<3>;
  const int tz = ceil_div(D, G::TZ), cols = N * ceil_div(H, G::TY) * ceil_div(W, G::TX);
  const int blocks = (cout / 48) * (narrow ? 1 : (c1 + c2) / 48);
  const int want = ceil_div((narrow ? 2 : 1) * ncu, blocks);  // workgroups along x (first layer: two per CU)
  // columns are cut into nsegz segments when there are fewer columns than workgroups; a segment re-loads two planes, so it
  // stays >= 4 tiles long
  int nsegz = 1;
  while (cols * nsegz < want && ceil_div(tz, 2 * nsegz) >= 4) nsegz *= 2;
  const int seglen = ceil_div(tz, nsegz);
  nsegz = ceil_div(tz, seglen);
  const int nseg = cols * nsegz;
  if (mode != 2 && (long)cols * tz < 8L * want) return false;  // too few half tiles per workgroup to amortise 132 accumulators x 27 taps of slab
  int nl = 8;
  while (nl > 1 && nseg / nl < 2) nl >>= 1;
  int g8 = ceil_div(want, nl);
  while (g8 > 1 && nl * g8 > nseg) --g8;
  o->nl = nl; o->g8 = g8; o->nseg = nseg; o->nsegz = nsegz; o->seglen = seglen; o->narrow = narrow ? 1 : 0;
  return true;
}

static size_t wgrad_x3_fused_ws_bytes(int N, int D, int H, int W, int c1, int c2, int cout) {
  X3Shape sh;
  if (!wgrad_x3_fused_shape(1, N, D, H, W, c1, c2, cout, &sh)) return 0;
  return (size_t)sh.nl * sh.g8 * 27 * cout * (c1 + (c2 > 0 ? c2 : 0)) * sizeof(float);
}


// one launch + the fixed-order reduction; returns 1 when the layer is not taken (the caller falls back to wgrad_x3)
static int wgrad_x3_fused(const void* x1, int c1, int pitch1, const void* x2, int c2, int pitch2, const void* dy, int dypitch,
                          const float* amax_dy, float* ws, float* dw, int dil, int N, int D, int H, int W, int cout, hipStream_t st) {
  X3Shape sh;
  if (!wgrad_x3_fused_shape(dil, N, D, H, W, c1, c2, cout, &sh)) return 1;
  using G = Wg3z<3>;
  WgradParams p;
  p.x1 = x1; p.x2 = x2; p.c1 = c1; p.c2 = c2; p.p1 = pitch1; p.p2 = pitch2;
  p.dy = dy; p.dyp = dypitch; p.ws = ws;
  p.N = N; p.D = D; p.H = H; p.W = W; p.cin = c1 + c2; p.cout = cout;
  p.tz = ceil_div(D, G::TZ); p.ty = ceil_div(H, G::TY); p.tx = ceil_div(W, G::TX);
  p.ntiles = sh.nseg; p.nlane = sh.nl; p.nsplit = sh.nl * sh.g8; p.ntaps = 27; p.dil = 1;
  p.seglen = sh.seglen; p.nsegz = sh.nsegz;
  static std::atomic<uint64_t> attr_a{0}, attr_b{0};
  BRATS_ENSURE_LDS_ATTR(conv_wgrad_x3_zwalk_kernel<3>, Wg3z<3>::LDS, attr_a);
  BRATS_ENSURE_LDS_ATTR(conv_wgrad_x3_zwalk_kernel<1>, Wg3z<1>::LDS, attr_b);
  if (sh.narrow)  // the slab columns of the padded ci lanes (c1 < 16) are never written and never read (cin = c1)
    hipLaunchKernelGGL(conv_wgrad_x3_zwalk_kernel<1>, dim3(p.nsplit, cout / 48, 1), dim3(512), Wg3z<1>::LDS, st, p, amax_dy);
  else
    hipLaunchKernelGGL(conv_wgrad_x3_zwalk_kernel<3>, dim3(p.nsplit, cout / 48, p.cin / 48), dim3(512), Wg3z<3>::LDS, st, p, amax_dy);
  BRATS_CHECK_LAUNCH();
  wgrad_reduce_launch((const float*)ws, dw, p.nsplit, cout, p.cin, 27, st, amax_dy);
  return 0;
}
