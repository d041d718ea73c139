// On-GPU inference drivers: sliding-window gather / weighted stitch (utils/inferers.py:103-162 of the
// reference, a MONAI-0.6 fork that stitches on the CPU) and test-time-augmentation index transforms
// (tta/transforms.py: OnAxes permute, flips, rot90 -- every composition is a signed permutation of the
// three spatial axes) with the sigmoid + running-sum of learning/engine.py:239-249 fused in.
// All tensors here are the reference's NCDHW f32.  Pure HBM-bound index kernels: one thread per
// output element, x-fastest so that stores (and, for un-permuted axes, loads) coalesce.
#include "common.hpp"

static inline int igrid(size_t total) {
  size_t b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b));
}

// dst[b][c][z][y][x] = src[n_b][c][z0_b + z - pad_z][...]; outside the source volume: cval (constant) or the voxel that
// F.pad's reflect / replicate / circular modes would put there (PytorchPadMode, inferers.py:109)
// (fuses the centred padding of inferers.py:103-109 with the window slicing of :126-130)
__device__ __forceinline__ int sw_pad_index(int i, int L, int mode) {
  if (mode == 1) { i = i < 0 ? -i : i; return i >= L ? 2 * (L - 1) - i : i; }  // reflect (no edge repeat; pad < L)
  if (mode == 2) return i < 0 ? 0 : (i >= L ? L - 1 : i);                      // replicate
  i %= L;                                                                      // circular
  return i < 0 ? i + L : i;
}
__global__ void sw_gather_kernel(const float* __restrict__ src, float* __restrict__ dst, const int* __restrict__ win,
                                 int B, int C, int D, int H, int W, int rd, int rh, int rw, int pz, int py, int px, float cval,
                                 int mode) {
  const size_t total = (size_t)B * C * rd * rh * rw;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t t = i;
    const int x = t % rw; t /= rw;
    const int y = t % rh; t /= rh;
    const int z = t % rd; t /= rd;
    const int c = t % C;
    const int b = (int)(t / C);
    const int n = win[b * 4];
    int sz = win[b * 4 + 1] + z - pz, sy = win[b * 4 + 2] + y - py, sx = win[b * 4 + 3] + x - px;
    if (mode != 0) { sz = sw_pad_index(sz, D, mode); sy = sw_pad_index(sy, H, mode); sx = sw_pad_index(sx, W, mode); }
    float v = cval;
    if (sz >= 0 && sz < D && sy >= 0 && sy < H && sx >= 0 && sx < W) v = src[((((size_t)n * C + c) * D + sz) * H + sy) * W + sx];
    dst[i] = v;
  }
}

// out[n][k][win] += imp * prob[b][k];  cnt[n][k][win] += imp      (one window per launch: no write races)
__global__ void sw_accumulate_kernel(const float* __restrict__ prob, const float* __restrict__ imp, float* __restrict__ out,
                                     float* __restrict__ cnt, int K, int Dp, int Hp, int Wp, int rd, int rh, int rw, int n,
                                     int z0, int y0, int x0) {
  const size_t total = (size_t)K * rd * rh * rw;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t t = i;
    const int x = t % rw; t /= rw;
    const int y = t % rh; t /= rh;
    const int z = t % rd;
    const int k = (int)(t / rd);
    const float w = imp[((size_t)z * rh + y) * rw + x];
    const size_t o = ((((size_t)n * K + k) * Dp + z0 + z) * Hp + y0 + y) * Wp + x0 + x;
    out[o] += w * prob[i];
    cnt[o] += w;
  }
}

// The same for ALL windows of a predictor batch in ONE launch (the per-window form needs one launch per window because
// overlapping windows would race): output-centric -- a thread owns one voxel of the padded image and adds the windows that
// cover it in window order, i.e. in exactly the order (and with exactly the operations) of B per-window launches.
__global__ void sw_accumulate_multi_kernel(const float* __restrict__ prob, const float* __restrict__ imp, float* __restrict__ out,
                                           float* __restrict__ cnt, const int* __restrict__ win, int B, int NB, int K, int Dp, int Hp,
                                           int Wp, int rd, int rh, int rw) {
  const size_t total = (size_t)NB * K * Dp * Hp * Wp;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t t = i;
    const int x = t % Wp; t /= Wp;
    const int y = t % Hp; t /= Hp;
    const int z = t % Dp; t /= Dp;
    const int k = t % K;
    const int n = (int)(t / K);
    float o = 0.f, c = 0.f;
    bool hit = false;
    for (int b = 0; b < B; ++b) {
      const int wz = z - win[b * 4 + 1], wy = y - win[b * 4 + 2], wx = x - win[b * 4 + 3];
      if (win[b * 4] != n || (unsigned)wz >= (unsigned)rd || (unsigned)wy >= (unsigned)rh || (unsigned)wx >= (unsigned)rw) continue;
      if (!hit) { o = out[i]; c = cnt[i]; hit = true; }
      const float w = imp[((size_t)wz * rh + wy) * rw + wx];
      o += w * prob[((((size_t)b * K + k) * rd + wz) * rh + wy) * rw + wx];
      c += w;
    }
    if (hit) { out[i] = o; cnt[i] = c; }
  }
}

// dst[n][k][z][y][x] = out[n][k][z+pz][y+py][x+px] / cnt[...]   (inferers.py:154-162)
__global__ void sw_finalize_kernel(const float* __restrict__ out, const float* __restrict__ cnt, float* __restrict__ dst,
                                   size_t NK, int Dp, int Hp, int Wp, int D, int H, int W, int pz, int py, int px) {
  const size_t total = NK * D * H * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t t = i;
    const int x = t % W; t /= W;
    const int y = t % H; t /= H;
    const int z = t % D;
    const size_t nk = t / D;
    const size_t o = ((nk * Dp + z + pz) * Hp + y + py) * Wp + x + px;
    dst[i] = out[o] / cnt[o];
  }
}

// dst[p][i0][i1][i2] (dims dd) = f(src[p][j0][j1][j2]) where source axis perm[a] runs along destination
// axis a, reversed when flip[a];  mode 0: dst = v, 1: dst += v, 2: dst += sigmoid(v)
__global__ void signed_perm_kernel(const float* __restrict__ src, float* __restrict__ dst, size_t planes, int d0, int d1,
                                   int d2, int s0, int s1, int s2, int p0, int p1, int p2, int f0, int f1, int f2, int mode) {
  const size_t total = planes * d0 * d1 * d2;
  const int sdim[3] = {s0, s1, s2};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t t = i;
    int di[3];
    di[2] = t % d2; t /= d2;
    di[1] = t % d1; t /= d1;
    di[0] = t % d0;
    const size_t pl = t / d0;
    int sj[3];
    sj[p0] = f0 ? d0 - 1 - di[0] : di[0];
    sj[p1] = f1 ? d1 - 1 - di[1] : di[1];
    sj[p2] = f2 ? d2 - 1 - di[2] : di[2];
    float v = src[((pl * sdim[0] + sj[0]) * sdim[1] + sj[1]) * sdim[2] + sj[2]];
    if (mode == 2) v = 1.f / (1.f + __expf(-v));
    dst[i] = mode == 0 ? v : dst[i] + v;
  }
}

extern "C" int brats_sw_gather(const float* src, float* dst, const int* windows, int B, int C, int D, int H, int W, int rd,
                               int rh, int rw, int pad_z, int pad_y, int pad_x, float cval, int pad_mode, brats_stream_t s) {
  if (!src || !dst || !windows || B <= 0 || pad_mode < 0 || pad_mode > 3) BRATS_FAIL(BRATS_E_ARG, "sw_gather: bad argument");
  if (pad_mode == 1 && (rd - D >= 2 * D - 1 || rh - H >= 2 * H - 1 || rw - W >= 2 * W - 1))  // F.pad: "padding size should be less than the input dimension"
    BRATS_FAIL(BRATS_E_ARG, "sw_gather: reflect padding must be smaller than the image");
  hipLaunchKernelGGL(sw_gather_kernel, dim3(igrid((size_t)B * C * rd * rh * rw)), dim3(256), 0, (hipStream_t)s, src, dst, windows, B,
                     C, D, H, W, rd, rh, rw, pad_z, pad_y, pad_x, cval, pad_mode);
  BRATS_CHECK_LAUNCH();
  return 0;
}

extern "C" int brats_sw_accumulate(const float* prob, const float* importance, float* out, float* count, int K, int Dp, int Hp,
                                   int Wp, int rd, int rh, int rw, int n, int z0, int y0, int x0, brats_stream_t s) {
  if (!prob || !importance || !out || !count || z0 < 0 || y0 < 0 || x0 < 0 || z0 + rd > Dp || y0 + rh > Hp || x0 + rw > Wp)
    BRATS_FAIL(BRATS_E_ARG, "sw_accumulate: window outside the (padded) image");
  hipLaunchKernelGGL(sw_accumulate_kernel, dim3(igrid((size_t)K * rd * rh * rw)), dim3(256), 0, (hipStream_t)s, prob, importance,
                     out, count, K, Dp, Hp, Wp, rd, rh, rw, n, z0, y0, x0);
  BRATS_CHECK_LAUNCH();
  return 0;
}

extern "C" int brats_sw_accumulate_multi(const float* prob, const float* importance, float* out, float* count, const int* windows,
                                         int B, int NB, int K, int Dp, int Hp, int Wp, int rd, int rh, int rw, brats_stream_t s) {
  if (!prob || !importance || !out || !count || !windows || B <= 0 || NB <= 0 || K <= 0 || rd > Dp || rh > Hp || rw > Wp)
    BRATS_FAIL(BRATS_E_ARG, "sw_accumulate_multi: bad argument");
  hipLaunchKernelGGL(sw_accumulate_multi_kernel, dim3(igrid((size_t)NB * K * Dp * Hp * Wp)), dim3(256), 0, (hipStream_t)s, prob,
                     importance, out, count, windows, B, NB, K, Dp, Hp, Wp, rd, rh, rw);
  BRATS_CHECK_LAUNCH();
  return 0;
}

extern "C" int brats_sw_finalize(const float* out, const float* count, float* dst, int NK, int Dp, int Hp, int Wp, int D, int H,
                                 int W, int pad_z, int pad_y, int pad_x, brats_stream_t s) {
  if (!out || !count || !dst) BRATS_FAIL(BRATS_E_ARG, "sw_finalize: null pointer");
  hipLaunchKernelGGL(sw_finalize_kernel, dim3(igrid((size_t)NK * D * H * W)), dim3(256), 0, (hipStream_t)s, out, count, dst,
                     (size_t)NK, Dp, Hp, Wp, D, H, W, pad_z, pad_y, pad_x);
  BRATS_CHECK_LAUNCH();
  return 0;
}

extern "C" int brats_spatial_signed_perm(const float* src, float* dst, int planes, int s0, int s1, int s2, int p0, int p1, int p2,
                                         int f0, int f1, int f2, int mode, brats_stream_t s) {
  if (!src || !dst || (1 << p0 | 1 << p1 | 1 << p2) != 7 || mode < 0 || mode > 2)
    BRATS_FAIL(BRATS_E_ARG, "spatial_signed_perm: perm must be a permutation of (0,1,2), mode in 0..2");
  const int sd[3] = {s0, s1, s2};
  const int d0 = sd[p0], d1 = sd[p1], d2 = sd[p2];
  hipLaunchKernelGGL(signed_perm_kernel, dim3(igrid((size_t)planes * d0 * d1 * d2)), dim3(256), 0, (hipStream_t)s, src, dst,
                     (size_t)planes, d0, d1, d2, s0, s1, s2, p0, p1, p2, f0, f1, f2, mode);
  BRATS_CHECK_LAUNCH();
  return 0;
}
