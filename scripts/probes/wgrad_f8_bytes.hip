// Probe (gfx950): what would the e4m3 all-taps weight gradient (conv_wgrad_alltaps_f8_kernel, csrc/conv_wgrad.hip) gain if its
// operands lived in HBM as e4m3 BYTES instead of 16-bit values quantised on stage?  Same roles, same
// MFMA phase (81 (tap, ci-fragment) pairs over 8 waves, ds_read_b64_tr_b8 operands, v_mfma_scale_f32_16x16x128_f8f6f4), but:
//   * the X halo tile (6 x 6 x 18 voxels x 48 B = 31 KB) and the dY tile (12 KB) arrive by LDS-DMA directly in the layout the
//     transposing reads want -- half the DMA instructions, no quantise pass, no second barrier;
//   * two tile buffers (2 x 48 KB): the next tile's DMA flies during the MFMA phase.
// Timing only (random bytes; the slab values are meaningless).  48 co x 48 ci block, 2 x 128^3 voxels, the product's grid.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -w scripts/probes/wgrad_f8_bytes.hip -o scripts/probes/wgrad_f8_bytes
#include "../../brats21_amd/csrc/common.hpp"
#include <stdio.h>

void brats_set_error(const char*, ...) {}

constexpr int TZ = 4, TY = 4, TX = 16, VOX = 256;
struct Params { const uint8_t* x; const uint8_t* dy; float* ws; int N, D, H, W, cin, cout, xp, yp, tz, ty, tx, ntiles, nlane; };

struct G {
  static constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2, HVOX = HZ * HY * HX;
  static constexpr int CO = 48, CI = 48, SX = CI, SY = CO, XPPV = CI / 16, YPPV = CO / 16;
  static constexpr int XPIECES = HVOX * XPPV, YPIECES = VOX * YPPV;          // 1944, 768 sixteen-byte pieces
  static constexpr int XI = (XPIECES + 511) / 512, YI = (YPIECES + 511) / 512;  // 4, 2 per thread
  static constexpr int XB = XI * 512 * 16, YB = YI * 512 * 16;                  // 32768, 16384
  static constexpr int LDS = 2 * (XB + YB);                                     // 98304
  static constexpr int PAIRS = 27 * 3, PPW = (PAIRS + 7) / 8;                   // 81, 11
};

template <int STRIDE> DEVI i32x8 tr8_frag(const char* base, int o0, int o1) {
  typedef __attribute__((ext_vector_type(2))) int v2i;
  typedef __attribute__((address_space(3))) v2i* lp;
  const v2i a0 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lp)(base + o0));
  const v2i a1 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lp)(base + o0 + 8 * STRIDE));
  const v2i a2 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lp)(base + o1));
  const v2i a3 = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lp)(base + o1 + 8 * STRIDE));
  return i32x8{a0[0], a0[1], a1[0], a1[1], a2[0], a2[1], a3[0], a3[1]};
}

__global__ __launch_bounds__(512, 1) void wgrad_f8_bytes_kernel(const Params p) {
  constexpr int COF = 3, CIF = 3;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kq = lane >> 4, L = lane & 15;
  const int lane8 = blockIdx.x % p.nlane, gsub = blockIdx.x / p.nlane, g8 = gridDim.x / p.nlane;
  const int split = blockIdx.x;
  const int tpx = (p.ntiles + p.nlane - 1) / p.nlane;
  const int tile_end = min(p.ntiles, (lane8 + 1) * tpx);
  const int xpb = p.xp, ypb = p.yp;  // bytes per voxel

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
  const unsigned xsample_bytes = (unsigned)p.D * p.H * p.W * xpb, ysample_bytes = (unsigned)p.D * p.H * p.W * ypb;
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

  struct TileLoads { rsrc4_t xrs, yrs; int xorg, yorg; unsigned zm, ym, xm, inter; };
  auto setup_loads = [&](int tile) {
    TileLoads T_;
    int bt = tile;
    const int x0 = (bt % p.tx) * TX; bt /= p.tx;
    const int y0 = (bt % p.ty) * TY; bt /= p.ty;
    const int z0 = (bt % p.tz) * TZ;
    const int n = bt / p.tz;
    const size_t sample_vox = (size_t)n * p.D * p.H * p.W;
    T_.xrs = make_rsrc4(p.x + sample_vox * xpb, xsample_bytes);
    T_.yrs = make_rsrc4(p.dy + sample_vox * ypb, ysample_bytes);
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
  auto issue = [&](const TileLoads& T_, int buf) {
    char* const xdst = lds + buf * (G::XB + G::YB) + wave * 1024;  // wave-uniform: the DMA adds lane * 16
    static_for<0, G::XI>([&](auto i_) {
      constexpr int i = i_;
      int c = xcode[i];
      OPAQUE_V(c);
      const unsigned ok = (unsigned)(c >> 14) & (T_.inter | ((T_.zm >> (c & 7)) & (T_.ym >> ((c >> 3) & 7)) & (T_.xm >> ((c >> 6) & 31)))) & 1u;
      lds_dma16_async(T_.xrs, xdst + i * 8192, (T_.xorg + xoffs[i]) | ((int)ok - 1));
    });
    static_for<0, G::YI>([&](auto i_) {
      constexpr int i = i_;
      int t_ = tid;
      OPAQUE_V(t_);
      const int P = t_ + 512 * i;
      const int vox = P / G::YPPV, part = P % G::YPPV;
      const int z = vox >> 6, y = (vox >> 4) & 3, x = vox & 15;
      const int yo = ((z * p.H + y) * p.W + x) * ypb + part * 16;
      const unsigned ok = (P < G::YPIECES ? 1u : 0u) & (T_.inter | ((T_.zm >> (z + 1)) & (T_.ym >> (y + 1)) & (T_.xm >> (x + 1)))) & 1u;
      lds_dma16_async(T_.yrs, xdst + G::XB + i * 8192, (T_.yorg + yo) | ((int)ok - 1));
    });
  };
  int xl[2], yl[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int r = 2 * kq + e;
    xl[e] = (((r >> 2) * G::HY + (r & 3)) * G::HX + (L >> 1)) * G::SX + (L & 1) * 8;
    yl[e] = (r * 16 + (L >> 1)) * G::SY + (L & 1) * 8;
  }
  const int tile_first = lane8 * tpx + gsub;
  int cur = 0;
  if (tile_first < tile_end) issue(setup_loads(tile_first), 0);
  for (int tile = tile_first; tile < tile_end; tile += g8, cur ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // this tile has landed for everybody; everybody is done reading the other buffer
    const bool more = tile + g8 < tile_end;
    TileLoads T_ = setup_loads(more ? tile + g8 : tile);
    if (!more) T_.zm = T_.inter = 0;
    __builtin_amdgcn_sched_barrier(0);
    issue(T_, cur ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    const char* ldx = lds + cur * (G::XB + G::YB);
    const char* ldy = ldx + G::XB;
    static_for<0, 2>([&](auto ks_) {
      constexpr int ks = ks_;
      const char* xk = ldx + ks * (2 * G::HY * G::HX * G::SX);
      i32x8 a[COF];
#pragma unroll
      for (int m = 0; m < COF; ++m) a[m] = tr8_frag<G::SY>(ldy + ks * (8 * 16 * G::SY) + m * 16, yl[0], yl[1]);
      i32x8 b[2];
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
#pragma unroll
  for (int jj = 0; jj < G::PPW; ++jj) {
    const int pid = wave + 8 * jj;
    if (pid < G::PAIRS) {
      const int t = pid / CIF, nn = pid % CIF;
      float* base = p.ws + ((size_t)split * 27 + t) * p.cout * p.cin;
      const int ci = nn * 16 + L;
#pragma unroll
      for (int m = 0; m < COF; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) base[(size_t)(m * 16 + 4 * kq + r) * p.cin + ci] = acc[jj][m][r];
    }
  }
}

int main(int argc, char** argv) {
  const bool dense = argc > 1 && argv[1][0] == 'd';  // "dense": no zeros among the X bytes (the product's timing script feeds randn data)
  Params p;
  p.N = 2; p.D = p.H = p.W = 128; p.cin = p.cout = 48; p.xp = p.yp = 48;
  p.tz = p.D / TZ; p.ty = p.H / TY; p.tx = p.W / TX; p.ntiles = p.N * p.tz * p.ty * p.tx; p.nlane = 8;
  const size_t bytes = (size_t)p.N * p.D * p.H * p.W * 48;
  uint8_t *x, *dy; float* ws;
  hipMalloc(&x, bytes); hipMalloc(&dy, bytes);
  const int grid = 256;
  hipMalloc(&ws, (size_t)grid * 27 * 48 * 48 * 4);
  // post-ReLU-like operand bytes: every second value zero, the rest small finite e4m3 values (0x28..0x47)
  uint8_t* h = (uint8_t*)malloc(bytes);
  unsigned s = 12345;
  for (size_t i = 0; i < bytes; ++i) { s = s * 1664525u + 1013904223u; h[i] = (!dense && (s >> 16 & 1)) ? 0 : (uint8_t)((0x28 + ((s >> 20) & 31)) | (dense ? (s >> 9) & 0x80 : 0)); }
  hipMemcpy(x, h, bytes, hipMemcpyHostToDevice);
  for (size_t i = 0; i < bytes; ++i) { s = s * 1664525u + 1013904223u; h[i] = (uint8_t)((0x28 + ((s >> 20) & 31)) | ((s >> 9) & 0x80)); }
  hipMemcpy(dy, h, bytes, hipMemcpyHostToDevice);
  p.x = x; p.dy = dy; p.ws = ws;
  hipFuncSetAttribute((const void*)wgrad_f8_bytes_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep) {
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(wgrad_f8_bytes_kernel, dim3(grid, 1, 1), dim3(512), G::LDS, 0, p);
    hipEventRecord(a);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(wgrad_f8_bytes_kernel, dim3(grid, 1, 1), dim3(512), G::LDS, 0, p);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 20;
    printf("%s X: e4m3-bytes all-taps wgrad 48->48 @2x128^3: %.3f ms  %.0f TF/s (kernel only, no slab reduce) err=%d\n", dense ? "dense" : "half-zero", ms, 2.0 * 27 * 48 * 48 * 2 * 128.0 * 128 * 128 / ms / 1e9, (int)hipGetLastError());
  }
  return 0;
}
