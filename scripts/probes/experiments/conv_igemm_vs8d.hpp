// EXPERIMENT (round 5), measured and NOT part of the library: profiles/r05_vs8d_double_buffered_negative.txt.
// To rebuild it: copy this file into brats21_amd/csrc/, include it from conv_bf16_k3_d1.hip instead of conv_igemm_vs8.hpp, dispatch
// `ck == 16 && conv_vs8_mode() >= 3 && rows16 % 3 == 0 && rows16 % 6 != 0` to conv_launch_vs8d<3, false, mode == 4>, let
// brats_conv3d_chunk() hand out 16 for those layers under mode >= 3 and let brats_conv3d_set_vs8() keep modes 3 / 4
// (time_vs8d.py, dbg_vs8d.py beside this file drive it).  Bit-exactness against the product kernels was checked with
// tests/test_ops_gpu.py::test_conv3d_vs8_kernel_matches_tile_kernel parametrised over modes 3 and 4 (9 passed).
//
// The 4 x 8 x 16-tile kernel of
// conv_igemm_vs8.hpp (bf16 / fp16, 3x3x3, dilation 1, Cout = 48 mod 96) on 16-channel chunks DOUBLE-BUFFERED in LDS, the halo tile
// of chunk c + 1 arriving by LDS-DMA issued by the four MFMA waves themselves at the top of chunk c -- no loader waves, no third
// wave per SIMD, no staging registers.
//
// What makes that possible is the wait discipline.  vmcnt is one in-order counter per wave, and hipcc knows nothing about the
// inline-asm DMAs: with compiler-tracked weight loads every `s_waitcnt vmcnt(n)` it emits is n too small by the DMAs in
// flight, i.e. the MMA loop would wait for the next chunk's halo at its first weight wait.  So the weight stream is inline asm
// as well (buffer_load with a scalar offset: the packed fragments of a tile are ONE linear stream over chunk * MS + step) and
// every wait is written by hand with the exact count:
//     step k issues the fragments of step k + WD, then waits for those of step k:
//         issued after them = WD groups of NF loads, plus -- while k < WD -- the NDMA DMAs of the next chunk, which were issued
//         behind the first WD groups of this chunk (those were requested during the previous chunk's last WD steps)
//     => vmcnt(NF * WD + NDMA) for k < WD, vmcnt(NF * WD) from k = WD on: the DMAs have WD macro-steps (4 x 24 MFMAs) to land
//        before anybody waits for them, and the wait at step WD is also the guarantee that this wave's share of the next chunk is
//        in LDS when the chunk barrier is reached.
// LDS: 2 x 36 KB (a chunk's halo tile = 6 x 10 x 18 voxels x 32 B = 33.75 KB = 34 DMA instructions of 1 KB; every wave issues
// exactly NDMA = 9 -- the surplus ones carry an out-of-range offset and write zeros into the 2 KB behind the tile -- so that the
// counts above are the same in all four waves) + the statistics scratch: two workgroups per CU as before.
// Ring of 7 weight-fragment groups (14 macro-steps per chunk: the slot of a step does not depend on the chunk), 5 of them live.
#pragma once
#include "conv_igemm_vs8.hpp"

constexpr int VS8D_WD = 4, VS8D_NDMA = 9, VS8D_BUF = 36 * 1024, VS8D_RING = 7;

template <int NF> constexpr int conv_vs8d_lds_bytes() { return 2 * VS8D_BUF + 4 * NF * 16 * 2 * 4; }

DEVI void vs8d_load_a(bf16x8& a, rsrc4_t rs, int voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(a) : "v"(voff), "s"(rs), "s"(soff));
#endif
}
// the weight fragments of the current step are complete once at most N younger vector-memory operations of this wave are
// outstanding; a compiler barrier for memory, the sched_barrier(0) behind it keeps the MFMAs behind it
template <int N> DEVI void vs8d_wait() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
#endif
}

// RS: the next chunk's pieces requested into REGISTERS instead (9 asynchronous buffer_loads per wave: 36 staging registers; the
// same wait counts) and written to the other LDS buffer behind the chunk's last MFMAs -- an LDS-DMA instruction stalls its wave
// 150-300 cycles at issue, an ordinary load does not
template <int NF, bool BST = false, bool RS = false>
__global__ __launch_bounds__(256, 2) void conv_igemm_vs8d_kernel(const ConvParams p, int ty4 /* 4-row tiles in y */) {
  static_assert(NF == 3, "the wait counts are written for three weight fragments per macro-step");
  using T = bf16_t;
  constexpr int CK = 16, WD = RS ? VS8D_WD - 1 : VS8D_WD, NDMA = VS8D_NDMA, RING = VS8D_RING;  // (RS: 12 registers for the staging set)
  using G = ConvGeom<T, 3, CK, 1, VS8_TY>;
  static_assert(G::S == 32 && G::PPV == 2 && G::MS % RING == 0 && G::MS > WD, "geometry of the 16-channel chunk");
  constexpr int NB = 8, YB = 4, MS = G::MS, NPIECE = G::HVOX * 2;
  static_assert((NPIECE + 63) / 64 <= 4 * NDMA - 2 && 4 * NDMA * 1024 <= VS8D_BUF, "DMA blocks of a chunk");
  constexpr int FOZ = G::HY * G::HX * G::S;
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

  // this lane's piece of each of the wave's NDMA blocks: voxel index inside the sample, or -1 (outside the volume / the tile)
  auto piece_vox = [&](int i, int ln) {
    const int P = (4 * i + wave) * 64 + ln;
    const int vox = P >> 1;
    const int hx = vox % G::HX, t = vox / G::HX;
    const int hy = t % G::HY, hz = t / G::HY;
    const int gz = z0 - 1 + hz, gy = y0 - 1 + hy, gx = x0 - 1 + hx;
    const bool ok = P < NPIECE && gz >= 0 && gz < p.D && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    return ok ? (gz * p.H + gy) * p.W + gx : -1;
  };
  int pvox[RS ? 1 : NDMA];  // (RS: decoded again for every chunk -- ~15 VALU per piece against 9 registers held through the MMA loop)
  if constexpr (!RS) {
#pragma unroll
    for (int i = 0; i < NDMA; ++i) pvox[i] = piece_vox(i, lane);
  }
  const int partb = (lane & 1) * 16;
  auto issue_dma = [&](int chunk, int buf) {
    const int c0 = chunk * CK;
    const T* src;
    int pitch;
    if (c0 < p.c1) { src = (const T*)p.x1 + c0; pitch = p.p1; }
    else { src = (const T*)p.x2 + (c0 - p.c1); pitch = p.p2; }
    const rsrc4_t rs = make_rsrc4(src + sample_vox * pitch, (unsigned)((size_t)p.D * p.H * p.W * pitch * 2));
    const int pb = pitch * 2;
    char* dst = lds + buf * VS8D_BUF + wave * 1024;
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      const int pv = RS ? piece_vox(i, lane) : pvox[RS ? 0 : i];
      lds_dma16_async(rs, dst + i * 4096, (pv * pb + partb) | (pv >> 31));
    }
  };
  bf16x8 stg[RS ? NDMA : 1];
  auto issue_loads = [&](int chunk) {
    const int c0 = chunk * CK;
    const T* src;
    int pitch;
    if (c0 < p.c1) { src = (const T*)p.x1 + c0; pitch = p.p1; }
    else { src = (const T*)p.x2 + (c0 - p.c1); pitch = p.p2; }
    const rsrc4_t rs = make_rsrc4(src + sample_vox * pitch, (unsigned)((size_t)p.D * p.H * p.W * pitch * 2));
    const int pb = pitch * 2;
    int ln = lane;
    asm volatile("" : "+v"(ln));  // (opaque: the decode is not to be hoisted out of the chunk loop into registers)
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      const int pv = piece_vox(i, ln);
      vs8d_load_a(stg[RS ? i : 0], rs, (pv * pb + partb) | (pv >> 31), 0);
    }
  };
  auto write_staged = [&](int buf) {
    char* dst = lds + buf * VS8D_BUF + wave * 1024 + lane * 16;
#pragma unroll
    for (int i = 0; i < NDMA; ++i) *(bf16x8*)(dst + i * 4096) = stg[RS ? i : 0];
  };

  f32x4 acc[NF][NB];
  vs8_init_acc<NF>(p, acc, f0, q);
  const int lane_b = ((wm * 2) * G::HY * G::HX + wn * YB * G::HX + v) * G::S + q * G::UB;

  // the packed weights of this tile: one linear stream of nchunks * MS macro-steps, rows16 KB each
  const int rs16 = p.rows16 * 1024;
  const rsrc4_t wrs = make_rsrc4(p.wpk, (unsigned)((size_t)p.nchunks * MS * rs16));
  const int wvoff = lane * 16;
  int wnext = f0 * 1024;  // scalar byte offset of the next macro-step to request
  bf16x8 a[RING][NF];
  auto load_a = [&](auto slot_) {
    constexpr int slot = slot_;
#pragma unroll
    for (int f = 0; f < NF; ++f) vs8d_load_a(a[slot][f], wrs, wvoff, wnext + f * 1024);
    wnext += rs16;
  };

  // prologue: chunk 0 lands (nothing else is in flight), then the first WD weight groups, then chunk 1's DMAs behind them
  issue_dma(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  static_for<0, WD>([&](auto k_) { load_a(k_); });

  bf16x8 b[NB];
  for (int chunk = 0; chunk < p.nchunks; ++chunk) {
    const char* ldsb = lds + (chunk & 1) * VS8D_BUF;
    const bool last = chunk + 1 == p.nchunks;  // scalar
    if (!last) {
      if constexpr (RS) issue_loads(chunk + 1);
      else issue_dma(chunk + 1, (chunk + 1) & 1);
    }
    auto read_b = [&](auto k_, auto half_) {
      constexpr int k = k_, half = half_;
      constexpr int o0 = G::unitoff(4 * k), o1 = G::unitoff(4 * k + 1), o2 = G::unitoff(4 * k + 2), o3 = G::unitoff(4 * k + 3);
      int lb;
      if constexpr (o1 - o0 == G::UB && o2 - o0 == 2 * G::UB && o3 - o0 == 3 * G::UB) lb = lane_b + o0;
      else lb = lane_b + (q == 0 ? o0 : q == 1 ? o1 - G::UB : q == 2 ? o2 - 2 * G::UB : o3 - 3 * G::UB);
#pragma unroll
      for (int i = YB * half; i < YB * half + YB; ++i) b[i] = *(const bf16x8*)(ldsb + lb + ((i / YB) * FOZ + (i % YB) * G::HX * G::S));
    };
    auto mma = [&](auto k_, auto half_) {
      constexpr int k = k_, half = half_;
#pragma unroll
      for (int i = YB * half; i < YB * half + YB; ++i)
#pragma unroll
        for (int f = 0; f < NF; ++f) acc[f][i] = MFMA16_16x16x32(a[k % RING][f], b[i], acc[f][i]);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    read_b(I0{}, I0{});
    static_for<0, MS>([&](auto k_) {
      constexpr int k = k_;
      // request step k + WD of the tile's weight stream (the next chunk's first steps from k = MS - WD on; behind the last chunk
      // the offsets run past the packed buffer: the descriptor's range check returns zeros, nobody reads them -- the loop body,
      // hence every wait count, is the same for all chunks)
      load_a(std::integral_constant<int, (k + WD) % RING>{});
      read_b(k_, I1{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (k < WD) {
        if (last) vs8d_wait<NF * WD>();  // (no DMAs were issued at the top of the last chunk)
        else vs8d_wait<NF * WD + NDMA>();
      } else {
        vs8d_wait<NF * WD>();
      }
      __builtin_amdgcn_sched_barrier(0);
      mma(k_, I0{});
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (k + 1 < MS) read_b(std::integral_constant<int, k + 1>{}, I0{});
      __builtin_amdgcn_sched_barrier(0);
      mma(k_, I1{});
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (RS) {
      if (!last) write_staged((chunk + 1) & 1);  // (landed since step WD; nobody reads that buffer before the barrier below)
    }
    if (!last) __syncthreads();  // everybody is done with this buffer, and every wave has seen its share of the next chunk land (step WD)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the surplus weight requests behind the last chunk)
  // ... whose destination registers must stay allocated until they have landed: an asm output nobody reads is dead to the register
  // allocator at once, and the load would land in whatever lives there by then (found the hard way: one accumulator fragment per wave)
  // (the last WD requests went to slots MS % RING = 0 .. WD - 1)
#pragma unroll
  for (int r = 0; r < WD; ++r)
#pragma unroll
    for (int f = 0; f < NF; ++f) asm volatile("" ::"v"(a[r][f]));

  // --- epilogue: statistics per 4x4x16 sub-tile, NDHWC store (conv_igemm_vs8.hpp) ---
  float* sred = (float*)(lds + 2 * VS8D_BUF);
  vs8_epilogue_store<NF, BST>(p, acc, sred, wm, wn, q, v, z0, y0, x0, ct, f0, sample_vox, n);
  if (p.stats) {
    __syncthreads();
    vs8_epilogue_stats<NF>(p, ty4, sred, tid, n, tzi, tyi, txi, ct);
  }
}

template <int NF, bool BST = false, bool RS = false>
int conv_launch_vs8d(const ConvParams& p0, hipStream_t st) {
  constexpr int lds = conv_vs8d_lds_bytes<NF>();
  auto kern = conv_igemm_vs8d_kernel<NF, BST, RS>;
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
