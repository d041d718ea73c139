// Loader-wave implicit-GEMM kernel on the 4 x 8 x 16 voxel tile (bf16, 3x3x3, dilation 1, Cout = 48 mod 96): the
// producer / consumer form of conv_igemm_vs8_kernel (networks/equiunet2020.py:19-25,424-437 and its input gradient).
//
// Why: the stamps of the one-tile kernel show a workgroup running its memory phases
// (halo loads -> wait -> LDS writes, twice per tile, then the epilogue: 27 k of 51 k cycles) and its MFMA phases strictly
// in sequence; two co-resident workgroups overlap them only by chance (25.8 k cycles per tile and CU against 16.1 k of
// matrix-pipe work).  A wave cannot prefetch its own next chunk: vmcnt is ONE in-order counter, so the wait for the
// weight fragments of the next macro-step would also wait for every halo load issued before them.
//
// Here a workgroup is 4 MFMA waves + NLW loader waves (320 / 384 threads, two workgroups per CU):
//   * the loader wave(s) fetch the halo tile of the NEXT 16-channel chunk by LDS-DMA (buffer_load ... lds, no staging
//     registers, no ds_write pass; pieces outside the volume use an out-of-range offset = zeros) into the other of two
//     LDS buffers (2 x 34 KB) while the MFMA waves work on the current one; their only VMEM traffic is that DMA, so
//     their vmcnt(0) before the chunk barrier means exactly "the next chunk has landed";
//   * the MFMA waves never wait for a halo load: weights stream from L2 as before (their own vmcnt), activation
//     fragments from LDS; one s_barrier per chunk hands the buffers over;
//   * three waves share one SIMD somewhere on the CU, so the kernel must fit 168 VGPRs: the MMA loop holds the eight
//     activation fragments of a macro-step in a ring of RB fragments (each is dead after its NF MFMAs) instead of
//     all eight at once.
// 16-channel chunks: unit = 8 channels, 2 units per tap, 54 units = 14 macro-steps (3.6 % padding); LDS voxel stride
// 32 B (ds_read_b128 conflict-free, conv_igemm.hpp ConvGeom::S).
#pragma once
#include "conv_igemm_vs8.hpp"

// (moved here with the kernel in round 4: the ring form of the MMA loop)
// One chunk of MFMA work with a ring of RB activation fragments: fragment j = (macro-step k = j / 8, x-row i = j % 8) is
// read RB - 1 fragments (3 (RB - 1) MFMAs) ahead of its use; the weight fragments of step k + WD are requested at the start
// of step k.  Registers: NF * 8 * 4 accumulators + (WD + 1) * NF * 4 weights + RB * 4 activations.
template <int NF, int WD, int RB, typename G>
DEVI void conv_mma_ring(const char* ldsb, int lane_b, int q, const void* wpk_chunk, int rows16, int f0, int lane,
                        f32x4 (&acc)[NF][8]) {
  constexpr int NB = 8, YB = 4, MS = G::MS, NJ = MS * NB;
  constexpr int FOZ = G::HY * G::HX * G::S;
  const bf16x8* wp0 = (const bf16x8*)wpk_chunk + (size_t)f0 * 64 + lane;
  bf16x8 a[WD + 1][NF];
  bf16x8 b[RB];
  auto load_a = [&](auto k_) {
    constexpr int k = k_;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#ifdef BRATS_ABL_NOWLOAD  // (ablation, see conv_mma_chunk)
      if constexpr (k <= WD) a[k % (WD + 1)][f] = wp0[((size_t)k * rows16 + f) * 64];
      else OPAQUE_V(a[k % (WD + 1)][f]);
#else
      a[k % (WD + 1)][f] = wp0[((size_t)k * rows16 + f) * 64];
#endif
    }
  };
  auto read_b = [&](auto j_) {
    constexpr int j = j_, k = j / NB, i = j % NB;
    constexpr int o0 = G::unitoff(4 * k), o1 = G::unitoff(4 * k + 1), o2 = G::unitoff(4 * k + 2), o3 = G::unitoff(4 * k + 3);
    int lb;
    if constexpr (o1 - o0 == G::UB && o2 - o0 == 2 * G::UB && o3 - o0 == 3 * G::UB) lb = lane_b + o0;
    else lb = lane_b + (q == 0 ? o0 : q == 1 ? o1 - G::UB : q == 2 ? o2 - 2 * G::UB : o3 - 3 * G::UB);
    b[j % RB] = *(const bf16x8*)(ldsb + lb + ((i / YB) * FOZ + (i % YB) * G::HX * G::S));
  };
  static_for<0, (WD < MS ? WD : MS)>([&](auto k_) { load_a(k_); });
  static_for<0, RB - 1>([&](auto j_) { read_b(j_); });
  static_for<0, NJ>([&](auto j_) {
    constexpr int j = j_, k = j / NB, i = j % NB;
    if constexpr (i == 0 && k + WD < MS) load_a(std::integral_constant<int, k + WD>{});
    if constexpr (j + RB - 1 < NJ) read_b(std::integral_constant<int, j + RB - 1>{});
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int f = 0; f < NF; ++f)
      acc[f][i] = MFMA16_16x16x32(a[k % (WD + 1)][f], b[j % RB], acc[f][i]);
    __builtin_amdgcn_sched_barrier(0);
  });
}



constexpr int LD_CK = 16;

template <int NF, int NLW> struct LdGeom {
  using G = ConvGeom<bf16_t, 3, LD_CK, 1, VS8_TY>;
  static_assert(G::S == 32 && G::PPV == 2, "lane-linear LDS image: voxel stride = 2 pieces");
  static constexpr int NPIECE = G::HVOX * G::PPV;       // 2160 16-byte pieces
  static constexpr int NDMA = (NPIECE + 63) / 64;       // 34 wave-instructions (the last one's spare lanes write zeros)
  static constexpr int BUF = NDMA * 1024;               // 34816
  static constexpr int SRED = 4 * NF * 16 * 2 * 4;
  static constexpr int LDS = 2 * BUF + SRED;
  static constexpr int DPW = (NDMA + NLW - 1) / NLW;    // DMA instructions per loader wave and chunk
};

// The loader wave(s): every chunk's halo tile as NDMA lane-linear LDS-DMA instructions; instruction j belongs to loader
// wave j % NLW.  Piece P = 64 j + lane -> voxel P / 2 of the 6 x 10 x 18 halo box, 16-byte half P % 2.
template <int NF, int NLW>
DEVI void conv_ld_loader(const ConvParams& p, char* lds, int lw, int lane, int n, int z0, int y0, int x0) {
  using L = LdGeom<NF, NLW>;
  using G = typename L::G;
  typedef bf16_t T;
  const size_t sample_vox = (size_t)n * p.D * p.H * p.W;
  // bit h of zm / ym / xm: halo plane / row / column h lies inside the volume; inter: the whole halo box does
  auto inside = [](int o, int size, int hn) {  // 0 <= o - 1 + h < size
    const int lo = o >= 1 ? 0 : 1 - o, hi = size - o + 1 < hn ? size - o + 1 : hn;
    return hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
  };
  const unsigned zm = inside(z0, p.D, G::HZ), ym = inside(y0, p.H, G::HY), xm = inside(x0, p.W, G::HX);
  const bool inter = zm == (1u << G::HZ) - 1 && ym == (1u << G::HY) - 1 && xm == (1u << G::HX) - 1;  // scalar
  const int org = ((z0 - 1) * p.H + (y0 - 1)) * p.W + (x0 - 1);  // voxel index of the halo corner (negative at the low faces)
  const int part16 = (lane & 1) * 16;
  int voff[L::DPW];   // voxel offset of the lane's piece from the halo corner
  int okm[L::DPW];    // 0 when the piece is inside the volume (and exists), -1 otherwise: OR-ed into the byte offset
  // (both are filled while chunk 0 is being issued: an LDS-DMA instruction holds its wave for 100+ cycles anyway, the
  //  ~20 VALU instructions of the next piece's decode hide behind it instead of delaying the first transfer)
  auto decode = [&](auto k_) {
    constexpr int k = k_;
    const int j = lw + NLW * k;
    const unsigned vox = 32u * j + (lane >> 1);
    const unsigned hx = vox % G::HX, t = vox / G::HX, hy = t % G::HY, hz = t / G::HY;
    const bool exists = j < L::NDMA && vox < (unsigned)G::HVOX;
    voff[k] = (hz * p.H + hy) * p.W + hx;
    const unsigned ok = exists ? (inter ? 1u : ((zm >> hz) & (ym >> hy) & (xm >> hx) & 1u)) : 0u;
    okm[k] = (int)ok - 1;
  };
  for (int chunk = 0; chunk < p.nchunks; ++chunk) {
    const int c0 = chunk * LD_CK;
    const T* src;
    int pitch;
    if (c0 < p.c1) { src = (const T*)p.x1 + c0; pitch = p.p1; }
    else { src = (const T*)p.x2 + (c0 - p.c1); pitch = p.p2; }
    const int pb = pitch * 2;
    const rsrc4_t rs = make_rsrc4(src + sample_vox * pitch, (unsigned)((size_t)p.D * p.H * p.W * pitch * 2));
    const int addend = org * pb + part16;
    char* dst = lds + (chunk & 1) * L::BUF;
    auto issue = [&](auto k_) {
      constexpr int k = k_;
      const int j = lw + NLW * k;  // scalar
      if (j < L::NDMA) lds_dma16_async(rs, dst + j * 1024, (voff[k] * pb + addend) | okm[k]);
    };
    if (chunk == 0) static_for<0, L::DPW>([&](auto k_) { decode(k_); issue(k_); });
    else static_for<0, L::DPW>([&](auto k_) { issue(k_); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the chunk have landed ...
    __builtin_amdgcn_s_barrier();                      // ... and behind the barrier everybody's have (and the MFMA waves are
                                                       // done with the buffer the next chunk goes to)
  }
}

template <int NF, int NLW, int WD, int RB>
__global__ __launch_bounds__(256 + 64 * NLW, 3) void conv_igemm_ld_kernel(const ConvParams p, int ty4 /* 4-row tiles in y */) {
  using L = LdGeom<NF, NLW>;
  using G = typename L::G;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int bt = blockIdx.x;
  const int txi = bt % p.tx; bt /= p.tx;
  const int tyi = bt % p.ty; bt /= p.ty;  // p.ty counts 8-row tiles here
  const int tzi = bt % p.tz;
  const int n = bt / p.tz;
  const int z0 = tzi * CONV_TZ, y0 = tyi * VS8_TY, x0 = txi * CONV_TX;

  if (wave >= 4) {  // ---- loader ----
    conv_ld_loader<NF, NLW>(p, lds, wave - 4, lane, n, z0, y0, x0);
    if (p.stats) __builtin_amdgcn_s_barrier();  // the statistics barrier of the epilogue
    return;
  }

  // ---- MFMA waves: the roles of conv_igemm_vs8_kernel ----
  const int wm = wave & 1, wn = wave >> 1;
  const int q = lane >> 4, v = lane & 15;
  const int ct = blockIdx.y;
  const int f0 = ct * NF;
  const size_t sample_vox = (size_t)n * p.D * p.H * p.W;
  f32x4 acc[NF][8];
  vs8_init_acc<NF>(p, acc, f0, q);
  const int lane_b = ((wm * 2) * G::HY * G::HX + wn * 4 * G::HX + v) * G::S + q * G::UB;
  const size_t chunk_stride = (size_t)G::MS * p.rows16 * 64 * 16;
  __syncthreads();  // chunk 0 has landed
  for (int chunk = 0; chunk < p.nchunks; ++chunk) {
    conv_mma_ring<NF, WD, RB, G>(lds + (chunk & 1) * L::BUF, lane_b, q, (const char*)p.wpk + chunk * chunk_stride, p.rows16,
                                 f0, lane, acc);
    if (chunk + 1 < p.nchunks) __syncthreads();  // next chunk landed; everybody is done reading this one
  }
  float* sred = (float*)(lds + 2 * L::BUF);
  vs8_epilogue_store<NF>(p, acc, sred, wm, wn, q, v, z0, y0, x0, ct, f0, sample_vox);
  if (p.stats) {
    __syncthreads();
    vs8_epilogue_stats<NF>(p, ty4, sred, tid, n, tzi, tyi, txi, ct);
  }
}

// conv_host.hip: 0 = the one-tile kernels, 1 = conv_igemm_vs8 (24-channel chunks), 2 = this kernel (16-channel chunks)
extern int g_conv_ld_variant;
static inline int conv_ld_variant() {
  if (g_conv_ld_variant >= 0) return g_conv_ld_variant;
  static int v = -1;
  if (v < 0) { const char* e = getenv("BRATS_CONV_LD_VARIANT"); v = e ? atoi(e) : 0; }
  return v;
}

template <int NF, int NLW, int WD, int RB>
int conv_launch_ld_one(const ConvParams& p0, hipStream_t st) {
  using L = LdGeom<NF, NLW>;
  auto kern = conv_igemm_ld_kernel<NF, NLW, WD, RB>;
  static std::atomic<uint64_t> attr_done{0};
  BRATS_ENSURE_LDS_ATTR(kern, L::LDS, attr_done);
  ConvParams p = p0;
  const int ty4 = p.ty;
  p.ty = ceil_div(p.H, VS8_TY);
  dim3 grid((unsigned)(p.N * p.tz * p.ty * p.tx), (unsigned)(p.rows16 / NF));
  hipLaunchKernelGGL(kern, grid, dim3(256 + 64 * NLW), L::LDS, st, p, ty4);
  BRATS_CHECK_LAUNCH();
  return 0;
}

template <int NF>
int conv_launch_ld(const ConvParams& p, hipStream_t st) {
  switch (conv_ld_variant()) {  // tuning variants (scripts/ab_ld.sh); 0 is the default
    case 1: return conv_launch_ld_one<NF, 2, 2, 5>(p, st);
    case 2: return conv_launch_ld_one<NF, 1, 1, 5>(p, st);
    case 3: return conv_launch_ld_one<NF, 1, 2, 6>(p, st);
  }
  return conv_launch_ld_one<NF, 1, 2, 5>(p, st);
}
