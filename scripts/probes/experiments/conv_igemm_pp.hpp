// "Ping-pong" persistent variant of the implicit-GEMM 3x3x3 convolution (bf16, dilation 1, unpadded LDS voxel
// rows: CK = 48 or 32) for the large layers of the U-Nets (networks/equiunet2020.py:19-25 and its dgrad).
//
// conv_igemm.hpp runs one tile per workgroup: global loads -> wait -> LDS -> MFMA -> epilogue, strictly in
// sequence, and relies on a second co-resident workgroup to fill the gaps.  Measured, the two workgroups of a CU
// drift into lockstep (same code, same phase lengths), so the matrix pipe idles during every load / epilogue
// phase (MFMA busy 45 % at 48->48 @128^3).  Here ONE workgroup of 8 waves per CU is split into two teams that
// are forced into anti-phase by shared barriers:
//
//   phase p     team (p & 1)      : MFMA over its halo block in LDS (weights streamed from L2 as before)
//               the other team    : epilogue of its previous tile (K-split exchange, bias, statistics, stores)
//                                   + LDS-DMA of its NEXT halo block into its own LDS buffer
//   barrier, swap roles.
//
// The halo block arrives by `buffer_load ... lds` (no staging registers, no ds_write pass): one wave instruction
// moves 64 consecutive 16-byte pieces of the flattened [voxel][CK] tile image; pieces outside the volume use an
// out-of-range offset, which the descriptor's range check turns into ZEROS in LDS (probed on gfx950:
// scripts/probes/ldsdma.hip) -- the convolution's zero padding costs no branch.  Each team owns one tile buffer
// (filled while the other team computes, read while the other team loads), so two buffers give full overlap.
// A workgroup walks (tile, cout-tile) items of one XCD-contiguous range; a tile's Cin chunks are consecutive
// blocks of the same team (accumulators stay in registers).
//
// STATUS (round 1, measured on MI355X, 2x128^3, random data): bit-identical to the tile kernel, NOT faster, so it
// is opt-in (BRATS_CONV_PP=1 / brats_conv3d_set_pingpong(1)) and the tile kernel stays the default:
//   48->96: tile kernel 0.99 ms; ping-pong 1.03-1.3 ms, of which MFMA phases alone 0.86 ms, epilogue +0.25, DMA +0.21
//   48->48: tile kernel 0.69 ms; ping-pong 1.2 ms (the K-split variant spills ~100 VGPRs at the 256-register cap)
// One MFMA wave per SIMD sustains only ~29 cycles per 16x16x32 MFMA here (weight prefetch distance 2 or 3 makes
// no difference), so hiding the load / epilogue phases cannot beat two co-resident waves per SIMD that interleave
// their MFMA streams; the other team's epilogue VALU and the DMA issue still take issue slots from the MFMA wave.
// What it would need: the weights through LDS as well (one FIFO of LDS-DMA, no VM loads in the MFMA wave) and a
// register budget that fits the K-split exchange -- see DESIGN.md.
#pragma once
#include "conv_igemm.hpp"

template <int CK, int NF, bool KSPLIT>
struct PPGeom {
  using G = ConvGeom<bf16_t, 3, CK, 1>;
  using TL = ConvTile<NF, KSPLIT>;
  static_assert(G::S == G::PPV * 16, "LDS-DMA needs an unpadded (lane-linear) tile image");
  static constexpr int NPIECE = G::HVOX * G::PPV;
  static constexpr int NINSTR = (NPIECE + 63) / 64;   // wave-instructions per halo block (61 for CK=48)
  static constexpr int IPW = (NINSTR + 3) / 4;        // per wave of the loading team
  static constexpr int TILE_PAD = NINSTR * 1024;      // the last instruction's unused lanes still write (zeros)
  static constexpr int XCH = KSPLIT ? 2 * (NF * 4) * 1024 : 0;  // K-split exchange: 2 wave pairs x NF*4 fragments
  static constexpr int SRED = 4 * NF * 16 * 2 * 4;    // per team: [wave][NF*16 channels][sum, sumsq]
  static constexpr int LDS = 2 * TILE_PAD + XCH + 2 * SRED;
};

#define PP_WAIT_VM0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define PP_WAIT_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define PP_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)

template <int CK, int NF, bool KSPLIT>
__global__ __launch_bounds__(512, 1) void conv_igemm_pp_kernel(const ConvParams p) {
  using PP = PPGeom<CK, NF, KSPLIT>;
  using G = typename PP::G;
  using TL = typename PP::TL;
  typedef bf16_t T;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = wave >> 2, w4 = wave & 3, wm = w4 & 1, wn = w4 >> 1;
  const int q = lane >> 4, v = lane & 15;
  char* const buf = lds + team * PP::TILE_PAD;
  f32x4* const xch = (f32x4*)(lds + 2 * PP::TILE_PAD);
  float* const sred = (float*)(lds + 2 * PP::TILE_PAD + PP::XCH + team * PP::SRED);

  // ---- work list: items = (tile, cout tile), ct fastest; 8 XCD-contiguous ranges; inside a range the XCD's
  //      workgroups interleave, and a workgroup's items alternate between its two teams ----
  const int nct = p.rows16 / TL::NFW;
  const int tps = p.tz * p.ty * p.tx;
  const int nitems = p.N * tps * nct;
  const int xcd = blockIdx.x & 7, wgx = blockIdx.x >> 3, gx = gridDim.x >> 3;
  const int per_xcd = (nitems + 7) / 8;
  const int it_lo = xcd * per_xcd, it_hi = min(nitems, it_lo + per_xcd);
  const int mine = it_lo + wgx < it_hi ? (it_hi - it_lo - wgx + gx - 1) / gx : 0;  // items of this workgroup
  const int nb_me = ((mine + 1 - team) >> 1) * p.nchunks;                          // blocks of my team
  const int nb_max = ((mine + 1) >> 1) * p.nchunks;                                // team 0 has the most
  auto item_of = [&](int kb) { return it_lo + wgx + (2 * (kb / p.nchunks) + team) * gx; };

  // ---- per-lane constants of the LDS-DMA pieces this wave issues (instruction i = w4 + 4*ii) ----
  int pcode[PP::IPW];
#pragma unroll
  for (int ii = 0; ii < PP::IPW; ++ii) {
    const int P = (w4 + 4 * ii) * 64 + lane;
    const int vox = P / G::PPV, part = P % G::PPV;
    const int hx = vox % G::HX, hy = (vox / G::HX) % G::HY, hz = vox / (G::HX * G::HY);
    pcode[ii] = P < PP::NPIECE ? (hz | hy << 3 | hx << 6 | part << 11 | 1 << 14) : 0;
  }
  auto issue_dma = [&](int kb) {
    const int item = item_of(kb), chunk = kb % p.nchunks;
    int bt = item / nct;
    const int x0 = (bt % p.tx) * CONV_TX; bt /= p.tx;
    const int y0 = (bt % p.ty) * CONV_TY; bt /= p.ty;
    const int z0 = (bt % p.tz) * CONV_TZ;
    const int n = bt / p.tz;
    const int c0 = chunk * CK;
    const T* src;
    int pitch;
    if (c0 < p.c1) { src = (const T*)p.x1 + c0; pitch = p.p1; }
    else { src = (const T*)p.x2 + (c0 - p.c1); pitch = p.p2; }
    const size_t svox = (size_t)p.D * p.H * p.W;
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)n * svox * pitch), (short)0, (int)(svox * pitch * 2), 0x00020000);
    unsigned zm = 0, ym = 0, xm = 0;
#pragma unroll
    for (int h = 0; h < G::HZ; ++h) zm |= ((unsigned)(z0 - 1 + h) < (unsigned)p.D ? 1u : 0u) << h;
#pragma unroll
    for (int h = 0; h < G::HY; ++h) ym |= ((unsigned)(y0 - 1 + h) < (unsigned)p.H ? 1u : 0u) << h;
#pragma unroll
    for (int h = 0; h < G::HX; ++h) xm |= ((unsigned)(x0 - 1 + h) < (unsigned)p.W ? 1u : 0u) << h;
    const int org = ((z0 - 1) * p.H + (y0 - 1)) * p.W + (x0 - 1);  // voxel index of the halo corner (may be negative)
    const int pitchb = pitch * 2;
#pragma unroll
    for (int ii = 0; ii < PP::IPW; ++ii) {
      if (w4 + 4 * ii < PP::NINSTR) {
        int c = pcode[ii];
        asm volatile("" : "+v"(c));  // opaque: keeps hipcc from hoisting the derived terms of every piece out of the phase loop
        const int pv = ((c & 7) * p.H + ((c >> 3) & 7)) * p.W + ((c >> 6) & 31);
        const unsigned ok = (unsigned)(c >> 14) & (zm >> (c & 7)) & (ym >> ((c >> 3) & 7)) & (xm >> ((c >> 6) & 31)) & 1u;
        const int vo = ok ? (org + pv) * pitchb + ((c >> 11) & 7) * 16 : -1;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(buf + (w4 + 4 * ii) * 1024), 16, vo, 0, 0, 0);
      }
    }
  };

  f32x4 acc[NF][8];
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[f][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int lane_b = ((wm * 2) * G::HY * G::HX + v) * G::S + q * G::UB;
  const size_t chunk_stride = (size_t)G::MS * p.rows16 * 64 * 16;
  int epi_item = -1;    // item whose accumulators are complete and wait for the epilogue (my team)
  int stat_item = -1;   // item whose per-wave statistics sit in sred and wait for the combine

  for (int ph = -1; ph <= 2 * nb_max; ++ph) {
    const bool mma_role = ph >= 0 && (ph & 1) == team;
    if (mma_role) {
      // statistics of the tile finished one phase ago: combine the 4 waves' partial sums (written before the
      // last barrier) and publish them
      if (stat_item >= 0) {
        const int tt = tid & 255;
        if (p.stats && tt < TL::NFW * 16) {
          const int ct = stat_item % nct, tile = stat_item / nct;
          const int c = ct * TL::NFW * 16 + tt;
          if (c < p.cout) {
            float s1 = 0.f, s2 = 0.f;
            if constexpr (KSPLIT) {
#pragma unroll
              for (int w = 0; w < 4; ++w) { s1 += sred[(w * NF * 16 + tt) * 2]; s2 += sred[(w * NF * 16 + tt) * 2 + 1]; }
            } else {
              const int half = tt / (NF * 16), cc = tt % (NF * 16);
#pragma unroll
              for (int m = 0; m < 2; ++m) { s1 += sred[((m + 2 * half) * NF * 16 + cc) * 2]; s2 += sred[((m + 2 * half) * NF * 16 + cc) * 2 + 1]; }
            }
            float* dst = p.stats + ((size_t)tile * p.cout + c) * 2;  // tile = n * tiles_per_sample + tile_in_sample
            dst[0] = s1;
            dst[1] = s2;
          }
        }
        stat_item = -1;
      }
      const int kb = ph >> 1;
      if (kb < nb_me && !(p.debug & 2)) {
        const int item = item_of(kb), chunk = kb % p.nchunks;
        const int ct = item % nct;
        const int f0 = ct * TL::NFW + (KSPLIT ? 0 : wn * NF);
        const char* wchunk = (const char*)p.wpk + chunk * chunk_stride;
        if constexpr (KSPLIT) {
          if (wn == 0) conv_mma_chunk<T, 3, CK, 1, NF, 0, 2>(buf, lane_b, q, wchunk, p.rows16, f0, lane, acc);
          else conv_mma_chunk<T, 3, CK, 1, NF, 1, 2>(buf, lane_b, q, wchunk, p.rows16, f0, lane, acc);
        } else {
          conv_mma_chunk<T, 3, CK, 1, NF, -1, 2>(buf, lane_b, q, wchunk, p.rows16, f0, lane, acc);
        }
        if (chunk == p.nchunks - 1) epi_item = item;
      } else {
        PP_BARRIER();
        PP_BARRIER();
      }
    } else {
      // ---- loading / epilogue role ----
      const int nph = ph + 1;
      if ((nph & 1) == team && (nph >> 1) < nb_me && !(p.debug & 1)) issue_dma(nph >> 1);
      const bool epi = epi_item >= 0 && !(p.debug & 4);
      int ct = 0, n = 0, z0 = 0, y0 = 0, x0 = 0;
      if (epi) {
        ct = epi_item % nct;
        int bt = epi_item / nct;
        x0 = (bt % p.tx) * CONV_TX; bt /= p.tx;
        y0 = (bt % p.ty) * CONV_TY; bt /= p.ty;
        z0 = (bt % p.tz) * CONV_TZ;
        n = bt / p.tz;
      }
      // K-split: wave (wm, 0) finalises voxel rows i = 0..3 of the pair's half tile, wave (wm, 1) rows 4..7;
      // the partial sums of the other K parity travel through the exchange buffer in two rounds
      if constexpr (KSPLIT) {
        if (epi && wn == 1) {
#pragma unroll
          for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int i = 0; i < 4; ++i) xch[((wm * NF + f) * 4 + i) * 64 + lane] = acc[f][i];
        }
        PP_WAIT_LGKM0();
        PP_BARRIER();
        if (epi && wn == 0) {
#pragma unroll
          for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[f][i] += xch[((wm * NF + f) * 4 + i) * 64 + lane];
#pragma unroll
          for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int i = 0; i < 4; ++i) xch[((wm * NF + f) * 4 + i) * 64 + lane] = acc[f][4 + i];
        }
        PP_WAIT_LGKM0();
        PP_BARRIER();
        if (epi && wn == 1) {
#pragma unroll
          for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[f][4 + i] += xch[((wm * NF + f) * 4 + i) * 64 + lane];
        }
      } else {
        PP_BARRIER();
        PP_BARRIER();
      }
      if (epi) {
        const int f0 = ct * TL::NFW + (KSPLIT ? 0 : wn * NF);
        const size_t sample_vox = (size_t)n * p.D * p.H * p.W;
        const bool x_ok = x0 + v < p.W;
        const bool second = p.y2 != nullptr && f0 * 16 >= p.ysplit;
        T* const ydst = second ? (T*)p.y2 : (T*)p.y;
        const int ypit = second ? p.y2pitch : p.ypitch;
        const int csub = second ? p.ysplit : 0;
        const int lane_o = (x0 + v) * ypit + 4 * q - csub;
        float bias[NF][4], s1[NF][4], s2[NF][4];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const int cbase = (f0 + f) * 16 + 4 * q;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            bias[f][r] = (p.bias && cbase < p.cout) ? p.bias[cbase + r] : 0.f;
            s1[f][r] = 0.f;
            s2[f][r] = 0.f;
          }
        }
        constexpr int NI = KSPLIT ? 4 : 8;
        const int ib = KSPLIT ? 4 * wn : 0;  // first voxel row this wave finalises
        const bool full = z0 + CONV_TZ <= p.D && y0 + CONV_TY <= p.H && x0 + CONV_TX <= p.W && (ct + 1) * TL::NFW * 16 <= p.cout;
        auto finalize = [&](auto ib_) {
          constexpr int IB = ib_;
          if (full) {
#pragma unroll
            for (int i = IB; i < IB + NI; ++i) {
              const int z = z0 + 2 * wm + (i >> 2), y = y0 + (i & 3);
              T* rowp = ydst + (sample_vox + (size_t)(z * p.H + y) * p.W) * ypit;
#pragma unroll
              for (int f = 0; f < NF; ++f) {
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  o[r] = acc[f][i][r] + bias[f][r];
                  s1[f][r] += o[r];
                  s2[f][r] += o[r] * o[r];
                }
                Vec<T, 4>::store(rowp + lane_o + (f0 + f) * 16, o);
              }
            }
          } else {
#pragma unroll
            for (int i = IB; i < IB + NI; ++i) {
              const int z = z0 + 2 * wm + (i >> 2), y = y0 + (i & 3);
              const bool ok = z < p.D && y < p.H && x_ok;
              const float mk = ok ? 1.f : 0.f;
              T* rowp = ydst + (sample_vox + (size_t)(z * p.H + y) * p.W) * ypit;
#pragma unroll
              for (int f = 0; f < NF; ++f) {
                const bool cok = (f0 + f) * 16 + 4 * q < p.cout;
                const float mf = cok ? mk : 0.f;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  o[r] = acc[f][i][r] + bias[f][r];
                  const float om = o[r] * mf;
                  s1[f][r] += om;
                  s2[f][r] += om * o[r];
                }
                if (ok && cok) Vec<T, 4>::store(rowp + lane_o + (f0 + f) * 16, o);
              }
            }
          }
        };
        if (ib == 0) finalize(std::integral_constant<int, 0>{});
        else finalize(std::integral_constant<int, KSPLIT ? 4 : 0>{});
        if (p.stats) {
#pragma unroll
          for (int f = 0; f < NF; ++f) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              s1[f][r] = row16_sum(s1[f][r]);
              s2[f][r] = row16_sum(s2[f][r]);
            }
            if (v == 0) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                sred[(w4 * NF * 16 + f * 16 + 4 * q + r) * 2 + 0] = s1[f][r];
                sred[(w4 * NF * 16 + f * 16 + 4 * q + r) * 2 + 1] = s2[f][r];
              }
            }
          }
        }
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[f][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        stat_item = epi_item;
        epi_item = -1;
      }
      PP_WAIT_VM0();    // my LDS-DMA pieces have landed (and my stores have left)
      PP_WAIT_LGKM0();  // statistics partial sums are in LDS
    }
    PP_BARRIER();
  }
  // the last tile's statistics (its combine normally happens at the start of the next MFMA phase)
  if (stat_item >= 0 && p.stats) {
    const int tt = tid & 255;
    if (tt < TL::NFW * 16) {
      const int ct = stat_item % nct, tile = stat_item / nct;
      const int c = ct * TL::NFW * 16 + tt;
      if (c < p.cout) {
        float s1 = 0.f, s2 = 0.f;
        if constexpr (KSPLIT) {
#pragma unroll
          for (int w = 0; w < 4; ++w) { s1 += sred[(w * NF * 16 + tt) * 2]; s2 += sred[(w * NF * 16 + tt) * 2 + 1]; }
        } else {
          const int half = tt / (NF * 16), cc = tt % (NF * 16);
#pragma unroll
          for (int m = 0; m < 2; ++m) { s1 += sred[((m + 2 * half) * NF * 16 + cc) * 2]; s2 += sred[((m + 2 * half) * NF * 16 + cc) * 2 + 1]; }
        }
        float* dst = p.stats + ((size_t)tile * p.cout + c) * 2;
        dst[0] = s1;
        dst[1] = s2;
      }
    }
  }
}

// ---- host side ----------------------------------------------------------------------------------
template <int CK, int NF, bool KSPLIT>
int conv_launch_pp(const ConvParams& p, hipStream_t st, int grid) {
  using PP = PPGeom<CK, NF, KSPLIT>;
  auto kern = conv_igemm_pp_kernel<CK, NF, KSPLIT>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, PP::LDS);
    if (e != hipSuccess) BRATS_FAIL(BRATS_E_HIP, "hipFuncSetAttribute(%d B LDS): %s", PP::LDS, hipGetErrorString(e));
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), PP::LDS, st, p);
  BRATS_CHECK_LAUNCH();
  return 0;
}

extern int g_conv_pp_mode;  // brats_conv3d_set_pingpong(): -1 = environment / default, 0 = off, 1 = on

// Returns -1 when the ping-pong kernel does not apply (caller falls through to the one-tile-per-workgroup kernel).
template <int CK>
int conv_try_pp(const ConvParams& p, hipStream_t st) {
  static int env_mode = -1, ncu = 0;
  if (env_mode < 0) {
    const char* e = getenv("BRATS_CONV_PP");  // 0 (default): never, 1: when there are enough tiles per CU
    env_mode = e ? atoi(e) : 0;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) ncu = 256;
    else ncu = prop.multiProcessorCount;
  }
  const int mode = g_conv_pp_mode >= 0 ? g_conv_pp_mode : env_mode;
  if (!mode) return -1;
  const ConvTileChoice t = conv_choose_tile(p.rows16);
  if (t.nf != 3) return -1;
  const long items = (long)p.N * p.tz * p.ty * p.tx * (p.rows16 / t.nfw);
  const int grid = (ncu / 8) * 8;
  if (grid < 8 || items < 4L * grid) return -1;  // small layers: one tile per workgroup fills the chip better
  const int mp = p.p1 > p.p2 ? p.p1 : p.p2;
  if ((double)p.D * p.H * p.W * mp * 2 >= 2147483648.0) return -1;  // 32-bit buffer offsets inside one sample
  return t.ksplit ? conv_launch_pp<CK, 3, true>(p, st, grid) : conv_launch_pp<CK, 3, false>(p, st, grid);
}
