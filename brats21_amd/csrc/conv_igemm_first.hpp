// First layer of the networks (networks/equiunet2020.py:424 encoder1 / networks/equiunet2021.py:246: 4 image modalities,
// padded to 8 channels -> 48 channels, 3x3x3, 16-bit): K = 27 taps x 8 channels = 216 -- seven macro-steps of MFMA work
// against 96 bytes of output per voxel.  This layer is bound by the CU's memory path, not by the matrix pipe, and the
// general tile kernels spend that path badly on it:
//   * every wave of every tile re-fetches the 21 KB of weight fragments (84 KB per 4-wave tile against 49 KB of output);
//   * the MFMA accumulator layout stores 16-byte pieces 96 bytes apart (a lane owns 8 channels of one voxel): four times
//     the write requests of a contiguous store for the same bytes -- 403 MB drained at 2.6 TB/s where plain stores of the
//     same size reach 6 TB/s ("what's weak" 2).
// Here: PERSISTENT workgroups (two per CU) walk the 4x4x16-voxel tiles with the weight fragments held in registers for the
// whole launch (84 VGPRs), the next tile's 10 KB halo travels by LDS-DMA into the other of two halo buffers while the current
// tile computes (no staging registers, and -- the counters being in issue order -- an exact s_waitcnt that leaves the tile's
// stores draining behind the next tile's work), and the output tile goes
// through LDS so that every store instruction writes 1 KB of consecutive bytes (a tile's x-row of 16 voxels x 48 channels is
// one 1536-byte run of the NDHWC tensor).  Same arithmetic as conv_igemm_vs8_kernel<8, 1, 3> (same packed weights, same K
// order inside a macro-step, f32 accumulation from the bias), same tile statistics layout.
#pragma once
#include "conv_igemm.hpp"

struct FirstGeom {
  static constexpr int HZ = 6, HY = 6, HX = 18, S = 16;  // halo tile of a 4x4x16 tile, one 16-byte piece (8 channels) per voxel
  static constexpr int HVOX = HZ * HY * HX;               // 648
  static constexpr int MS = 7;                            // macro-steps: 28 units (taps) of 8 channels, the last one zero
  static constexpr int HALO_BYTES = 768 * S;              // one halo buffer: 3 LDS-DMA pieces per thread (648 used)
  static constexpr int OUT_BYTES = 256 * 96;              // the tile's 16 x-rows x 1536 B
  static constexpr int SRED_BYTES = 4 * 48 * 2 * 4;
  static constexpr int LDS_BYTES = 2 * HALO_BYTES + OUT_BYTES + SRED_BYTES;
  static constexpr int tapoff(int tap) { return tap < 27 ? (((tap / 9) * HY + (tap / 3) % 3) * HX + tap % 3) * S : 0; }
};

__global__ __launch_bounds__(256, 2) void conv_first_kernel(const ConvParams p, int ntiles) {
  using T = bf16_t;
  using G = FirstGeom;
  constexpr int NF = 3, NB = 4;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* const outb = lds + 2 * G::HALO_BYTES;
  float* const sred = (float*)(lds + 2 * G::HALO_BYTES + G::OUT_BYTES);  // [wave][48][2]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q = lane >> 4, v = lane & 15;

  // ---- once per launch: weight fragments (all 7 macro-steps x 3 cout fragments), per-lane tap offsets, staging map
  bf16x8 a[G::MS][NF];
  {
    const bf16x8* wp = (const bf16x8*)p.wpk + lane;
#pragma unroll
    for (int ms = 0; ms < G::MS; ++ms)
#pragma unroll
      for (int f = 0; f < NF; ++f) a[ms][f] = wp[((size_t)ms * p.rows16 + f) * 64];
  }
  int toff[G::MS];  // byte offset of the lane quarter's tap in macro-step ms
#pragma unroll
  for (int ms = 0; ms < G::MS; ++ms)
    toff[ms] = q == 0 ? G::tapoff(4 * ms) : q == 1 ? G::tapoff(4 * ms + 1) : q == 2 ? G::tapoff(4 * ms + 2) : G::tapoff(4 * ms + 3);
  const int lane_b = ((wave * G::HY) * G::HX + v) * G::S;  // wave w computes z-slice w of the tile: x-rows (w, 0..3)
  f32x4 bias[NF];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    bias[f] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (p.bias) bias[f] = *(const f32x4*)(p.bias + f * 16 + 4 * q);
  }
  // halo pieces of this thread: P = tid + 256 j (j < 3), P -> (hz, hy, hx); rel = voxel offset from the halo origin
  constexpr int IPT = 3;
  int hzyx[IPT], rel[IPT];
  const int pb = p.p1 * 2;  // bytes per input voxel
#pragma unroll
  for (int j = 0; j < IPT; ++j) {
    const int P = tid + 256 * j;
    const int hz = P / (G::HY * G::HX), r2 = P % (G::HY * G::HX), hy = r2 / G::HX, hx = r2 % G::HX;
    hzyx[j] = P < G::HVOX ? (hz << 16) | (hy << 8) | hx : -1;
    rel[j] = ((hz * p.H + hy) * p.W + hx) * pb;
  }
  const int tps = p.tz * p.ty * p.tx;  // tiles per sample

  // the halo of tile t: three LDS-DMA instructions per wave (buffer_load ... lds: lane l's 16 bytes land at dst + 16 l; pieces
  // outside the volume and the pieces past the 648th get an out-of-range offset and arrive as zeros) into halo buffer `buf`
  auto dma_halo = [&](int t, int buf) {
    int bt = t;
    const int txi = bt % p.tx; bt /= p.tx;
    const int tyi = bt % p.ty; bt /= p.ty;
    const int tzi = bt % p.tz;
    const int n = bt / p.tz;
    const int z0 = tzi * 4 - 1, y0 = tyi * 4 - 1, x0 = txi * 16 - 1;  // halo origin
    const rsrc4_t rs = make_rsrc4((const T*)p.x1 + (size_t)n * p.D * p.H * p.W * p.p1, (unsigned)((size_t)p.D * p.H * p.W * pb));
    const int base = ((z0 * p.H + y0) * p.W + x0) * pb;
    const bool interior = z0 >= 0 && z0 + G::HZ <= p.D && y0 >= 0 && y0 + G::HY <= p.H && x0 >= 0 && x0 + G::HX <= p.W;  // scalar
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
      int vo = base + rel[j];
      if (!interior) {
        const int gz = z0 + (hzyx[j] >> 16), gy = y0 + ((hzyx[j] >> 8) & 0xff), gx = x0 + (hzyx[j] & 0xff);
        if (gz < 0 || gz >= p.D || gy < 0 || gy >= p.H || gx < 0 || gx >= p.W) vo = -1;
      }
      if (hzyx[j] < 0) vo = -1;
      lds_dma16_async(rs, lds + buf * G::HALO_BYTES + (wave * 64 + 256 * j) * 16, vo);
    }
  };

  // XCD-aware walk: workgroup b sits on XCD b % 8 (round-robin dispatch) and every XCD has its own L2, so each XCD takes ONE contiguous
  // eighth of the tile list (x fastest, then y, z, n) and its workgroups walk it interleaved -- tiles that share halo lines are then
  // fetched through one L2 instead of up to eight (PMC before: L2 hit 0.47, 168 MiB fetched for a 67 MB input)
  const int per_xcd = (ntiles + 7) >> 3, wg_per_xcd = (int)gridDim.x >> 3;
  const int t_lo = (blockIdx.x & 7) * per_xcd, t_hi = min(ntiles, t_lo + per_xcd);
  int t = t_lo + ((int)blockIdx.x >> 3), cur = 0;
  if (t < t_hi) dma_halo(t, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (also the weight fragments)
  for (; t < t_hi; t += wg_per_xcd, cur ^= 1) {
    // every wave has waited for its own pieces of this tile's halo (below / above); behind the barrier all of them are in LDS,
    // and every wave is past the previous tile's reads of the other halo buffer and of the output image
    __syncthreads();
    if (t + wg_per_xcd < t_hi) dma_halo(t + wg_per_xcd, cur ^ 1);  // lands behind the MFMAs and the epilogue
    const char* const halo = lds + cur * G::HALO_BYTES;

    f32x4 acc[NF][NB];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int i = 0; i < NB; ++i) acc[f][i] = bias[f];
    bf16x8 b[2][NB];
    auto read_b = [&](int ms, bf16x8 (&dst)[NB]) {
      const char* src = halo + lane_b + toff[ms];
#pragma unroll
      for (int i = 0; i < NB; ++i) dst[i] = *(const bf16x8*)(src + i * G::HX * G::S);
    };
    read_b(0, b[0]);
#pragma unroll
    for (int ms = 0; ms < G::MS; ++ms) {
      // the four fragment reads of the next macro-step go out in front of this one's twelve MFMAs (sched_barrier pins the order:
      // hipcc otherwise sinks each read to just before its first use and the loop runs at LDS latency)
      if (ms + 1 < G::MS) read_b(ms + 1, b[(ms + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int f = 0; f < NF; ++f) acc[f][i] = MFMA16_16x16x32(a[ms][f], b[ms & 1][i], acc[f][i]);
      __builtin_amdgcn_sched_barrier(0);
    }

    // ---- epilogue: statistics of the wave's z-slice, 16-bit rounding, tile image into LDS
    float s1[NF][4], s2[NF][4];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) { s1[f][rr] = 0.f; s2[f][rr] = 0.f; }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      char* dst = outb + ((wave * 4 + i) * 16 + v) * 96 + 8 * q;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        float o[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) o[rr] = acc[f][i][rr];
        stat_fwd<false>(s1[f], s2[f], o, 1.f);
        *(u32x2*)(dst + 32 * f) = u32x2{pack2(o[0], o[1]), pack2(o[2], o[3])};
      }
    }
    if (p.stats) {
#pragma unroll
      for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const float a1 = row16_sum(s1[f][rr]), a2 = row16_sum(s2[f][rr]);
          if (v == 0) {
            sred[(wave * 48 + f * 16 + 4 * q + rr) * 2] = a1;
            sred[(wave * 48 + f * 16 + 4 * q + rr) * 2 + 1] = a2;
          }
        }
    }
    __syncthreads();
    // ---- the tile leaves LDS in its NDHWC byte order: 1536 pieces of 16 bytes, piece Q = tid + 256 k -> x-row Q / 96
    {
      int bt = t;
      const int txi = bt % p.tx; bt /= p.tx;
      const int tyi = bt % p.ty; bt /= p.ty;
      const int tzi = bt % p.tz;
      const int n = bt / p.tz;
      T* const ybase = (T*)p.y + (((size_t)n * p.D + tzi * 4) * p.H + tyi * 4) * p.W * 48 + (size_t)txi * 16 * 48;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int Q = tid + 256 * k;
        const int row = Q / 96, w16 = Q % 96;  // row = z * 4 + y inside the tile
        const u32x4 val = *(const u32x4*)(outb + Q * 16);
        *(u32x4*)((char*)(ybase + ((size_t)(row >> 2) * p.H + (row & 3)) * p.W * 48) + w16 * 16) = val;
      }
      {  // (every thread issues the store -- lanes without a value at an out-of-range offset, dropped by the range
                      //  check -- so that the count of memory operations per iteration is the same in every wave)
        const int c = (tid >> 1) % 48, e = tid & 1;
        const int tile = t % tps;
        const float val = (sred[(0 * 48 + c) * 2 + e] + sred[(1 * 48 + c) * 2 + e]) + (sred[(2 * 48 + c) * 2 + e] + sred[(3 * 48 + c) * 2 + e]);
        const __amdgpu_buffer_rsrc_t rst = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.stats + ((size_t)n * tps + tile) * 96), (short)0, 96 * 4, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, val), rst, tid < 96 ? tid * 4 : -1, 0, 0);
      }
    }
    // Memory operations of this wave in issue order: [3 DMA pieces of the next tile] [6 + 1 stores of this tile].  All but the
    // seven youngest done = the next tile's halo pieces of this wave have landed; the stores drain behind the next tile's work.
    asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  }
}

// the layer this kernel is built for: one 8-channel chunk, exactly 48 output channels in a dense tensor, whole 4x4x16 tiles
static inline bool conv_first_ok(const ConvParams& p, int ck) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("BRATS_CONV_FIRST"); on = e ? atoi(e) : 1; }
  return on && ck == 8 && p.nchunks == 1 && p.c2 == 0 && p.stats && p.cout == 48 && p.rows16 == 3 && p.ypitch == 48 && !p.y2 && p.D % 4 == 0 &&
         p.H % 4 == 0 && p.W % 16 == 0 && ((size_t)p.y & 15) == 0 && (!p.bias || ((size_t)p.bias & 15) == 0) &&
         (long)p.N * p.tz * p.ty * p.tx >= 2048;
}

static int conv_launch_first(const ConvParams& p, hipStream_t st) {
  auto kern = conv_first_kernel;
  static std::atomic<uint64_t> attr_done{0};
  BRATS_ENSURE_LDS_ATTR(kern, FirstGeom::LDS_BYTES, attr_done);
  const int ntiles = p.N * p.tz * p.ty * p.tx;
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    hipDeviceProp_t prop;
    cus = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
  }
  const int grid = (ntiles < 2 * cus ? ntiles : 2 * cus) & ~7;  // (a multiple of 8: the same number of workgroups on every XCD; conv_first_ok wants >= 2048 tiles)
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), FirstGeom::LDS_BYTES, st, p, ntiles);
  BRATS_CHECK_LAUNCH();
  return 0;
}
