// Persistent form of the y-split implicit-GEMM kernel (conv_igemm.hpp, VS roles, bf16, 3x3x3).  EXPERIMENT, off by
// default (BRATS_CONV_PERSIST=1 / brats_conv3d_set_persistent): bit-identical to the one-tile kernel, 3-8 % slower on
// the 128^3 layers, 5 % faster at 64^3.
//
// Motivation (ablation of the one-tile kernel, 48->48 @128^3): per tile and slot ~2.6 us go to the kernel prologue and
// the global-load round trip of the halo tile, ~2 us to the epilogue's drain; two workgroups per CU do not hide all of
// it.  Here a workgroup walks a list of tiles and the halo tile of the NEXT (tile, chunk) item is fetched into
// registers WHILE the MFMAs of the current one run: its buffer_loads are dealt out over the macro-steps of the MMA
// loop.  vmcnt completes in order, so a wait for the weight fragments of step k+2 also waits for the activation loads
// issued before them -- those are at least two steps (24 MFMAs) old by then, the budget the weight stream already lives
// with.  The MFMAs are inline asm accumulating in place (with the builtin the allocator gives every MFMA a fresh
// destination: 92 registers for 48 accumulators, which spills the prefetch registers = synchronous waits).
// Why it does not win: the kernel is not latency-bound but operand-delivery-bound -- with 4 voxel fragments per wave
// every MFMA needs 256 B of weights through the vector-memory path (4 SIMDs x 256 B / 16 cycles = the CU's 64 B/clk)
// and 341 B of activations from LDS (67 % of 128 B/clk); hiding the tile-boundary latencies leaves those two pipes as
// busy as before.
// Tiles: 8 contiguous ranges (one per XCD, block id % 8), dealt round-robin to the workgroups of that XCD so that
// neighbouring tiles -- which share halo lines -- are in flight on the same L2 at the same time.
#pragma once
#include "conv_igemm.hpp"

template <int CK, int DIL, int NF>
__global__ __launch_bounds__(256, 2) void conv_igemm_vsp_kernel(const ConvParams p) {
  using T = bf16_t;
  using G = ConvGeom<T, 3, CK, DIL>;
  using TL = ConvTile<NF, false, true>;
  constexpr int NB = TL::NB, YB = NB / 2;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int q = lane >> 4, v = lane & 15;
  const int ct = blockIdx.y;
  const int f0 = ct * TL::NFW;

  // tile list of this workgroup
  const int tps = p.tz * p.ty * p.tx, tiles_total = p.N * tps;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int nslot = ((int)gridDim.x + 7 - xcd) >> 3;
  const int t_lo = (int)((long)tiles_total * xcd / 8), t_hi = (int)((long)tiles_total * (xcd + 1) / 8);

  constexpr int NROWS = G::HZ * G::HY;
  constexpr int PPR = G::HX * G::PPV;
  constexpr int IPR = (PPR + 63) / 64;
  constexpr int RPW = (NROWS + 3) / 4;
  constexpr int NLOAD = RPW * IPR;
  // NPRE of the NLOAD wave-instructions are issued inside the MMA loop; the rest right after it (their registers are
  // the operand registers the loop just released -- all NLOAD in flight next to 48 accumulators, 3 weight stages and
  // the B fragments do not fit 256 VGPRs, and a spilled prefetch register is a synchronous wait inside the loop)
  constexpr int NPRE = NLOAD < 12 ? NLOAD : 12;
  constexpr int LSTEP = (G::MS >= 2 * NPRE + 2) ? 2 : 1;  // macro-steps between two prefetch loads
  static_assert(G::MS >= NPRE + 1, "not enough macro-steps to deal the prefetch loads out");
  int lds_off[IPR], hxs[IPR], part16[IPR];
#pragma unroll
  for (int j = 0; j < IPR; ++j) {
    const int pc = lane + 64 * j;
    hxs[j] = pc / G::PPV;
    part16[j] = (pc % G::PPV) * 16;
    lds_off[j] = pc < PPR ? wave * (G::HX * G::S) + hxs[j] * G::S + part16[j] : -1;
  }
  const int lane_b = ((wm * 2) * G::HY * G::HX + wn * 2 * G::HX + v) * G::S + q * G::UB;
  const int chunk_stride = G::MS * p.rows16 * 64 * 16;

  f32x4 acc[NF][NB];
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int i = 0; i < NB; ++i) acc[f][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  u32x4 r[RPW][IPR];
  // --- prefetch state of the item being fetched (scalars) + per-lane piece offsets ---
  __amdgpu_buffer_rsrc_t nrs;
  int n_z0, n_y0, n_xr, n_pb;  // tile origin (z, y), x0 - R, pitch in bytes
  int n_go[IPR];                // per-lane byte offset from the row origin, or -1
  auto setup_item = [&](int t, int chunk) {
    int bt = t % tps;
    const int n = t / tps;
    const int txi = bt % p.tx; bt /= p.tx;
    const int tyi = bt % p.ty;
    const int tzi = bt / p.ty;
    n_z0 = tzi * CONV_TZ; n_y0 = tyi * CONV_TY;
    const int x0 = txi * CONV_TX;
    n_xr = x0 - G::R;
    const int c0 = chunk * CK;
    const T* src;
    int pitch;
    if (c0 < p.c1) { src = (const T*)p.x1 + c0; pitch = p.p1; }
    else { src = (const T*)p.x2 + (c0 - p.c1); pitch = p.p2; }
    n_pb = pitch * 2;
    nrs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)n * p.D * p.H * p.W * pitch), (short)0,
                                            (int)((size_t)p.D * p.H * p.W * pitch * 2), 0x00020000);
#pragma unroll
    for (int j = 0; j < IPR; ++j) {
      const int gx = n_xr + hxs[j];
      n_go[j] = (lane + 64 * j < PPR && gx >= 0 && gx < p.W) ? hxs[j] * n_pb + part16[j] : -1;
    }
  };
  auto load_one = [&](auto idx_) {  // one wave-instruction of the halo tile: row k, instruction j
    constexpr int idx = idx_, k = idx / IPR, j = idx % IPR;
    const int row = wave + 4 * k;
    const int hz = row / G::HY, hy = row % G::HY;
    const int gz = n_z0 - G::R + hz, gy = n_y0 - G::R + hy;
    const bool row_ok = row < NROWS && gz >= 0 && gz < p.D && gy >= 0 && gy < p.H;
    const int rb = ((gz * p.H + gy) * p.W + n_xr) * n_pb;
    const int vo = (row_ok && n_go[j] >= 0) ? rb + n_go[j] : -1;
    r[k][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(nrs, vo, 0, 0));
  };
  auto commit = [&]() {
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
      if (wave + 4 * k < NROWS) {
#pragma unroll
        for (int j = 0; j < IPR; ++j)
          if (lds_off[j] >= 0) *(u32x4*)(lds + lds_off[j] + k * 4 * (G::HX * G::S)) = r[k][j];
      }
    }
  };

  int t = t_lo + slot, chunk = 0;
  if (t < t_hi) {
    setup_item(t, 0);
    static_for<0, NLOAD>([&](auto i_) { load_one(i_); });
  }
  constexpr int LDS_MAIN = (G::LDS_TILE + 15) / 16 * 16;
  float* sred = (float*)(lds + LDS_MAIN);  // [4 (wm + 2 wn)][NFW*16][2]
  bool first = true;
  while (t < t_hi) {
    if (!first) __syncthreads();  // every wave is done reading the previous item's tile
    first = false;
    commit();
    __syncthreads();
    // the item after this one
    int nt = t, nc = chunk + 1;
    if (nc == p.nchunks) { nc = 0; nt = t + nslot; }
    const bool more = nt < t_hi;  // scalar
    if (more) setup_item(nt, nc);
    const char* wchunk = (const char*)p.wpk + chunk * chunk_stride;
    conv_mma_chunk<T, 3, CK, DIL, NF, -1, 0, NB>(lds, lane_b, q, wchunk, p.rows16, f0, lane, acc, [&](auto k_) {
      constexpr int k = k_;
      if constexpr (k % LSTEP == 0 && k / LSTEP < NPRE) {
        if (more) load_one(std::integral_constant<int, k / LSTEP>{});
      }
    });
    if (more) static_for<NPRE, NLOAD>([&](auto i_) { load_one(i_); });
    // the MFMAs above are inline asm: the compiler does not know that their results need ~10 wait states before a
    // VALU read; the epilogue's first read comes after address arithmetic, this makes it explicit
    asm volatile("s_nop 15");
    if (chunk == p.nchunks - 1) {
      // --- epilogue of tile t (as conv_igemm_kernel, VS roles) ---
      int bt = t % tps;
      const int n = t / tps, tile_in_sample = bt;
      const int txi = bt % p.tx; bt /= p.tx;
      const int tyi = bt % p.ty;
      const int tzi = bt / p.ty;
      const int z0 = tzi * CONV_TZ, y0 = tyi * CONV_TY, x0 = txi * CONV_TX;
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
        for (int rr = 0; rr < 4; ++rr) {
          bias[f][rr] = (p.bias && cbase < p.cout) ? p.bias[cbase + rr] : 0.f;
          s1[f][rr] = 0.f;
          s2[f][rr] = 0.f;
        }
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int z = z0 + 2 * wm + (i / YB), y = y0 + 2 * wn + (i % YB);
        const bool ok = z < p.D && y < p.H && x_ok;
        const float mk = ok ? 1.f : 0.f;
        T* rowp = ydst + (sample_vox + (size_t)(z * p.H + y) * p.W) * ypit;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const bool cok = (f0 + f) * 16 + 4 * q < p.cout;
          const float mf = cok ? mk : 0.f;
          float o[4];
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            o[rr] = acc[f][i][rr] + bias[f][rr];
            const float om = o[rr] * mf;
            s1[f][rr] += om;
            s2[f][rr] += om * o[rr];
            acc[f][i][rr] = 0.f;
          }
          if (ok && cok) Vec<T, 4>::store(rowp + lane_o + (f0 + f) * 16, o);
        }
      }
      if (p.stats) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            s1[f][rr] = row16_sum(s1[f][rr]);
            s2[f][rr] = row16_sum(s2[f][rr]);
          }
          if (v == 0) {
            const int cl = f * 16 + 4 * q;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              sred[(((wm + 2 * wn) * TL::NFW * 16) + cl + rr) * 2 + 0] = s1[f][rr];
              sred[(((wm + 2 * wn) * TL::NFW * 16) + cl + rr) * 2 + 1] = s2[f][rr];
            }
          }
        }
        __syncthreads();
        if (tid < TL::NFW * 16) {
          const int c = ct * TL::NFW * 16 + tid;
          if (c < p.cout) {
            float* dst = p.stats + (((size_t)n * tps + tile_in_sample) * p.cout + c) * 2;
            // same association as conv_igemm_kernel: the two kernels are bit-identical
            float t1 = sred[tid * 2] + sred[(TL::NFW * 16 + tid) * 2];
            float t2 = sred[tid * 2 + 1] + sred[(TL::NFW * 16 + tid) * 2 + 1];
            t1 += sred[(2 * TL::NFW * 16 + tid) * 2] + sred[(3 * TL::NFW * 16 + tid) * 2];
            t2 += sred[(2 * TL::NFW * 16 + tid) * 2 + 1] + sred[(3 * TL::NFW * 16 + tid) * 2 + 1];
            dst[0] = t1;
            dst[1] = t2;
          }
        }
      }
    }
    t = nt;
    chunk = nc;
  }
}

extern int g_conv_persist_mode;  // conv_host.hip: -1 = BRATS_CONV_PERSIST (default off), 0 / 1 = brats_conv3d_set_persistent
static inline int conv_persist_mode() {
  if (g_conv_persist_mode >= 0) return g_conv_persist_mode;
  static int v = -1;
  if (v < 0) { const char* e = getenv("BRATS_CONV_PERSIST"); v = e ? atoi(e) : 0; }
  return v;
}

// returns -1 when the layer does not qualify (caller falls back to the one-tile kernels)
template <int CK, int DIL, int NF>
int conv_try_vsp(const ConvParams& p, hipStream_t st) {
  if (!conv_persist_mode()) return -1;
  const long tiles = (long)p.N * p.tz * p.ty * p.tx;
  const int cblocks = p.rows16 / NF;
  if (tiles * cblocks < 2048) return -1;  // small layers: too few tiles per workgroup to pipeline
  constexpr int lds = conv_lds_bytes<bf16_t, 3, CK, DIL, NF, false, true>();
  auto kern = conv_igemm_vsp_kernel<CK, DIL, NF>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) BRATS_FAIL(BRATS_E_HIP, "hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
    attr_done = true;
  }
  int gx = 512 / cblocks;
  if (gx < 8) gx = 8;
  dim3 grid((unsigned)gx, (unsigned)cblocks);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
  BRATS_CHECK_LAUNCH();
  return 0;
}
