// Split-precision ("x3") weight gradient with the split in the staging path (round 5; included by conv_wgrad.hip).
//
//     dW[tap][co][ci] = sum_v dY[v][co] * X[v + off(tap)][ci]      (autograd of nn.Conv3d, networks/equiunet2020.py:19-25)
//
// on f32 tensors as three 16-bit MFMA products per MAC: x = hi + lo with hi = rn16(x), lo = rn16(x - hi);
// hi*hi + hi*lo + lo*hi drops only lo*lo (<= 2^-22 of the product; error analysis: conv_igemm_x3.hpp).
//
// Round 4 (wgrad_x3 in conv_wgrad.hip, still the path of every shape this kernel does not take) wrote the hi / lo tensors
// of X and dY to HBM in a streaming pass and ran the UNCHANGED 16-bit kernels three times, each re-staging its tiles:
// 14.1 of the 40.3 ms parity-mode step at 0.34 of the MFMA ceiling.  Here ONE launch stages each f32 tile once:
//   * one 8-wave workgroup per CU owns all 27 taps of a HALF tile (2 x 4 x 16 voxels): the f32 X halo box (4 x 6 x 18
//     voxels x 48 channels = 83 KB) and the dY tile (128 x 48 = 24.6 KB) arrive by 16-byte buffer_loads in registers
//     (the NEXT tile's, issued before the MFMA phase of the current one; out-of-volume pieces through the descriptor's range
//     check = zeros), are split in registers (dY times the power of two that puts its recorded |max| into fp16's top binades)
//     and written as FOUR 16-bit LDS tiles (X hi, X lo, dY hi, dY lo: 41.5 + 41.5 + 12.3 + 12.3 = 107.5 KB; the all-taps
//     tile of the 16-bit kernel, 4 x 4 x 16, would need 173 KB as hi + lo);
//   * MFMA phase = the all-taps kernel's roles (81 (tap, ci-fragment) pairs dealt to 8 waves, 3 co fragments each, 132
//     accumulator registers kept over all tiles the workgroup walks), operands by the transposing LDS read from the hi / lo
//     tiles (96-byte voxel stride: conflict-free), 9 MFMAs per (k-step, pair): a_hi*b_lo, a_lo*b_hi, a_hi*b_hi per co fragment
//     -- 4 transposing reads per 9 MFMAs where the 16-bit kernel needs 2 per 3;
//   * slabs and the fixed-order reduction (times 2^-k) are the 16-bit kernels' (bitwise reproducible).
// Per 256 voxels the CU fetches 215 KB of f32 once instead of 3 x 87 KB of 16-bit values plus the split pass's 16 bytes per
// element of HBM traffic.  CIF = 3: 48-channel ci blocks; CIF = 1: the first layer (<= 16 input channels).
#pragma once

template <int CIF> struct Wg3x {
  static constexpr int NW = 8;                                                            // waves per workgroup
  static constexpr int TZ = 2, TY = 4, TX = 16, VOX = TZ * TY * TX;                       // 128 voxels = 8 x-rows = 4 k-steps
  static constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2, HVOX = HZ * HY * HX;        // 4 x 6 x 18 = 432
  static constexpr int CI = 16 * CIF, CO = 48;
  static constexpr int SX = 2 * CI, SY = 2 * CO;                                          // bytes per voxel in a 16-bit tile: 96 (32), 96
  static constexpr int XPPV = CI / 4, YPPV = CO / 4;                                      // 16-byte f32 pieces per voxel: 12 (4), 12
  static constexpr int XPIECES = HVOX * XPPV, YPIECES = VOX * YPPV;                       // 5184 (1728), 1536
  static constexpr int LDS_XH = HVOX * SX, LDS_YH = VOX * SY;                             // one of hi / lo
  static constexpr int LDS = 2 * LDS_XH + 2 * LDS_YH;                                     // 107520 (52224)
  static constexpr int PAIRS = 27 * CIF, PPW = (PAIRS + NW - 1) / NW;                     // 81, 11 | 27, 4  (NW = 8)
};

// 4 consecutive f32 channels (one 16-byte piece) times the power of two sc -> 4 hi + 4 lo 16-bit values (8 bytes each)
DEVI void x3_split4(const u32x4 a, float sc, u32x2& hi, u32x2& lo) {
  const uint32_t w[4] = {a[0], a[1], a[2], a[3]};  // (through scalars: see f8_quant8)
  uint32_t h[2], l[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float x0 = __uint_as_float(w[2 * i]) * sc, x1 = __uint_as_float(w[2 * i + 1]) * sc;
    h[i] = pack2(x0, x1);
    float h0, h1;
    unpack2(h[i], h0, h1);
    l[i] = pack2(x0 - h0, x1 - h1);
  }
  hi = u32x2{h[0], h[1]};
  lo = u32x2{l[0], l[1]};
}

template <int CIF>
__global__ __launch_bounds__(512, CIF == 1 ? 4 : 2) void conv_wgrad_x3_alltaps_kernel(const WgradParams p, const float* __restrict__ amax_dy) {
  using G = Wg3x<CIF>;
  constexpr int NW = G::NW;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* const ldxh = lds;
  char* const ldxl = lds + G::LDS_XH;
  char* const ldyh = lds + 2 * G::LDS_XH;
  char* const ldyl = ldyh + G::LDS_YH;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, v = lane & 15;
  const int lane8 = blockIdx.x % p.nlane, gsub = blockIdx.x / p.nlane, g8 = gridDim.x / p.nlane;
  const int split = blockIdx.x;
  const int tpx = (p.ntiles + p.nlane - 1) / p.nlane;
  const int tile_end = min(p.ntiles, (lane8 + 1) * tpx);
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

  // the wave's (tap, ci-fragment) pairs: pid = wave + 8 jj is wave-uniform, so these live in scalar registers
  int poff[G::PPW];
#pragma unroll
  for (int jj = 0; jj < G::PPW; ++jj) {
    const int pid = wave + NW * jj;
    const int t = pid / CIF, nn = pid % CIF;
    poff[jj] = pid < G::PAIRS ? (((t / 9) * G::HY + (t / 3) % 3) * G::HX + t % 3) * G::SX + nn * 32 : 0;
  }
  f32x4 acc[G::PPW][3];
#pragma unroll
  for (int jj = 0; jj < G::PPW; ++jj)
#pragma unroll
    for (int m = 0; m < 3; ++m) acc[jj][m] = f32x4{0.f, 0.f, 0.f, 0.f};

  // staged pieces by ROWS of the halo box: wave w owns halo rows w, w + 8, w + 16 (row = hz * HY + hy, 24 rows) and dY row w
  // (row = z * 4 + y, 8 rows), so a row's coordinates, range tests and base offset are scalar; a lane's pieces inside a row
  // are lane + 64 j (XPR = 18 voxels x XPPV parts: 216 pieces = 3.4 loads per lane; dY: 16 x 12 = 192 = 3 loads exactly), whose
  // column offsets are per-lane constants.  The 8 hi (lo) bytes of a piece land at byte 8 * (row * XPR + lane + 64 j) of the hi
  // (lo) tile: the tiles are the lane-linear image of the piece order (SX = 8 XPPV) -- conflict-free ds_write_b64 runs.
  constexpr int XPR = G::HX * G::XPPV, XJ = (XPR + 63) / 64, XRW = (G::HZ * G::HY) / NW;  // 216 (72), 4 (2), 3
  constexpr int YPR = G::TX * G::YPPV, YJ = YPR / 64, YRW = (G::TZ * G::TY) / NW;          // 192, 3, 1
  static_assert((G::HZ * G::HY) % NW == 0 && (G::TZ * G::TY) % NW == 0 && YPR % 64 == 0, "");
  int xcol[XJ], xhx[XJ];  // byte offset of the lane's piece inside a halo row; its hx (out of the row / channel range: 31 -> never valid)
#pragma unroll
  for (int j = 0; j < XJ; ++j) {
    const int pc = lane + 64 * j;
    const int hx = pc / G::XPPV, part = pc % G::XPPV;
    const bool ok = pc < XPR && part * 4 < ci_lim;
    xcol[j] = hx * xpb + part * 16;
    xhx[j] = ok ? hx : 31;
  }
  int ycol[YJ], yx[YJ];
#pragma unroll
  for (int j = 0; j < YJ; ++j) {
    const int pc = lane + 64 * j;
    ycol[j] = (pc / G::YPPV) * ypb + (pc % G::YPPV) * 16;
    yx[j] = pc / G::YPPV + 1;  // (halo-box column of the dY voxel)
  }
  u32x4 rx[XRW][XJ], ry[YRW][YJ];
  auto issue_loads = [&](int tile) {
    int bt = tile;
    const int x0 = (bt % p.tx) * G::TX; bt /= p.tx;
    const int y0 = (bt % p.ty) * G::TY; bt /= p.ty;
    const int z0 = (bt % p.tz) * G::TZ;
    const int n = bt / p.tz;
    const size_t sample_vox = (size_t)n * p.D * p.H * p.W;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)(xsrc + sample_vox * xpitch), (short)0,
                                                                          (int)xsample_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)((const float*)p.dy + sample_vox * p.dyp + co0),
                                                                          (short)0, (int)ysample_bytes, 0x00020000);
    unsigned xm = 0;  // bit h: halo column h is inside the volume (bit 31 stays clear: the "never valid" column)
#pragma unroll
    for (int h = 0; h < G::HX; ++h) xm |= ((unsigned)(x0 - 1 + h) < (unsigned)p.W ? 1u : 0u) << h;
#pragma unroll
    for (int k = 0; k < XRW; ++k) {
      const int row = wave + NW * k;
      const int gz = z0 - 1 + row / G::HY, gy = y0 - 1 + row % G::HY;
      const bool row_ok = (unsigned)gz < (unsigned)p.D && (unsigned)gy < (unsigned)p.H;  // scalar
      const int rb = ((gz * p.H + gy) * p.W + (x0 - 1)) * xpb;                             // scalar (may be negative at the low faces)
#pragma unroll
      for (int j = 0; j < XJ; ++j) {
        const bool ok = row_ok && ((xm >> xhx[j]) & 1u);
        rx[k][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, ok ? rb + xcol[j] : -1, 0, 0));
      }
    }
#pragma unroll
    for (int k = 0; k < YRW; ++k) {
      const int row = wave + NW * k;
      const int gz = z0 + (row >> 2), gy = y0 + (row & 3);
      const bool row_ok = gz < p.D && gy < p.H;
      const int rb = ((gz * p.H + gy) * p.W + x0) * ypb;
#pragma unroll
      for (int j = 0; j < YJ; ++j) {
        const bool ok = row_ok && ((xm >> yx[j]) & 1u);
        ry[k][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(yrs, ok ? rb + ycol[j] : -1, 0, 0));
      }
    }
  };

  const int tile_first = lane8 * tpx + gsub;
  const int qq = v >> 2, pp = v & 3;
  const int ybase = (4 * q + qq) * G::SY + pp * 8;
  const int xbase = (4 * q + qq) * G::SX + pp * 8;
  for (int tile = tile_first; tile < tile_end; tile += g8) {
    issue_loads(tile);  // (in flight while the other waves finish the previous tile's MFMA phase)
    __syncthreads();    // previous tile's LDS reads are done
#pragma unroll
    for (int k = 0; k < XRW; ++k)
#pragma unroll
      for (int j = 0; j < XJ; ++j)
        if (XPR % 64 == 0 || j + 1 < XJ || lane + 64 * j < XPR) {
          u32x2 hi, lo;
          x3_split4(rx[k][j], 1.f, hi, lo);
          const int o = ((wave + NW * k) * XPR + lane + 64 * j) * 8;
          *(u32x2*)(ldxh + o) = hi;
          *(u32x2*)(ldxl + o) = lo;
        }
#pragma unroll
    for (int k = 0; k < YRW; ++k)
#pragma unroll
      for (int j = 0; j < YJ; ++j) {
        u32x2 hi, lo;
        x3_split4(ry[k][j], ysc, hi, lo);
        const int o = ((wave + NW * k) * YPR + lane + 64 * j) * 8;
        *(u32x2*)(ldyh + o) = hi;
        *(u32x2*)(ldyl + o) = lo;
      }
    __syncthreads();
    // k-step s = x-rows 2s, 2s+1 of the tile (row = z*4 + y); fragment layout: see conv_wgrad_kernel
    // (the staging registers are dead here: room for double-buffered dY fragments and three X fragment pairs in flight)
#ifndef X3_PD
#define X3_PD 2
#endif
    constexpr int PD = CIF == 1 ? 2 : X3_PD;   // (first layer: 128 registers for two workgroups per CU)
#ifndef X3_AB
#define X3_AB 2
#endif
    constexpr int AB = CIF == 1 ? 1 : X3_AB;   // dY fragment buffers
    bf16x8 ah[AB][3], al[AB][3], bh[PD + 1], bl[PD + 1];
    auto read_a = [&](auto s_) {
      constexpr int s = s_;
      const int yoff = ybase + (32 * s) * G::SY;
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        ah[s % AB][m] = tr_pair(ldyh + yoff + m * 32, ldyh + yoff + 16 * G::SY + m * 32);
        al[s % AB][m] = tr_pair(ldyl + yoff + m * 32, ldyl + yoff + 16 * G::SY + m * 32);
      }
    };
    auto read_b = [&](auto u_) {
      constexpr int u = u_;
      constexpr int s = u / G::PPW, jj = u % G::PPW;
      const int xoff = xbase + (((s >> 1) * G::HY + 2 * (s & 1)) * G::HX) * G::SX + poff[jj];
      bh[u % (PD + 1)] = tr_pair(ldxh + xoff, ldxh + xoff + G::HX * G::SX);
      bl[u % (PD + 1)] = tr_pair(ldxl + xoff, ldxl + xoff + G::HX * G::SX);
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
  }

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
static bool wgrad_x3_fused_shape(int dil, int N, int D, int H, int W, int c1, int c2, int cout, int* g8_out, int* nl_out, int* ntiles_out,
                                 int* narrow_out) {
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
  const int ntiles = N * ceil_div(D, Wg3x<3>::TZ) * ceil_div(H, Wg3x<3>::TY) * ceil_div(W, Wg3x<3>::TX);
  const int blocks = (cout / 48) * (narrow ? 1 : (c1 + c2) / 48);
  const int nl = wgrad_nlane(ntiles);
  int g8 = ceil_div((narrow ? 2 : 1) * ncu, nl * blocks);  // (first layer: 52 KB of LDS, 128 registers -> two workgroups per CU)
  if (g8 < 1) g8 = 1;
  if (mode == 2) {  // tests: any volume (workgroups beyond the tile count write zero slabs)
    while (g8 > 1 && nl * g8 > ntiles) --g8;
  } else if (ntiles < 8 * nl * g8) return false;  // too few half tiles per workgroup to amortise 132 accumulators x 27 taps of slab
  *g8_out = g8; *nl_out = nl; *ntiles_out = ntiles; *narrow_out = narrow ? 1 : 0;
  return true;
}

static size_t wgrad_x3_fused_ws_bytes(int N, int D, int H, int W, int c1, int c2, int cout) {
  int g8, nl, nt, narrow;
  if (!wgrad_x3_fused_shape(1, N, D, H, W, c1, c2, cout, &g8, &nl, &nt, &narrow)) return 0;
  return (size_t)nl * g8 * 27 * cout * (c1 + (c2 > 0 ? c2 : 0)) * sizeof(float);
}

// one launch + the fixed-order reduction; returns 1 when the layer is not taken (the caller falls back to wgrad_x3)
static int wgrad_x3_fused(const void* x1, int c1, int pitch1, const void* x2, int c2, int pitch2, const void* dy, int dypitch,
                          const float* amax_dy, float* ws, float* dw, int dil, int N, int D, int H, int W, int cout, hipStream_t st) {
  int g8, nl, nt, narrow;
  if (!wgrad_x3_fused_shape(dil, N, D, H, W, c1, c2, cout, &g8, &nl, &nt, &narrow)) return 1;
  using G = Wg3x<3>;
  WgradParams p;
  p.x1 = x1; p.x2 = x2; p.c1 = c1; p.c2 = c2; p.p1 = pitch1; p.p2 = pitch2;
  p.dy = dy; p.dyp = dypitch; p.ws = ws;
  p.N = N; p.D = D; p.H = H; p.W = W; p.cin = c1 + c2; p.cout = cout;
  p.tz = ceil_div(D, G::TZ); p.ty = ceil_div(H, G::TY); p.tx = ceil_div(W, G::TX);
  p.ntiles = nt; p.nlane = nl; p.nsplit = nl * g8; p.ntaps = 27; p.dil = 1;
  static std::atomic<uint64_t> attr_a{0}, attr_b{0};
  BRATS_ENSURE_LDS_ATTR(conv_wgrad_x3_alltaps_kernel<3>, Wg3x<3>::LDS, attr_a);
  BRATS_ENSURE_LDS_ATTR(conv_wgrad_x3_alltaps_kernel<1>, Wg3x<1>::LDS, attr_b);
  if (narrow)  // the slab columns of the padded ci lanes (c1 < 16) are never written and never read (cin = c1)
    hipLaunchKernelGGL(conv_wgrad_x3_alltaps_kernel<1>, dim3(p.nsplit, cout / 48, 1), dim3(512), Wg3x<1>::LDS, st, p, amax_dy);
  else
    hipLaunchKernelGGL(conv_wgrad_x3_alltaps_kernel<3>, dim3(p.nsplit, cout / 48, p.cin / 48), dim3(512), Wg3x<3>::LDS, st, p, amax_dy);
  BRATS_CHECK_LAUNCH();
  wgrad_reduce_launch((const float*)ws, dw, p.nsplit, cout, p.cin, 27, st, amax_dy);
  return 0;
}
