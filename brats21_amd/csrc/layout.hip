// NCDHW f32 (the reference's tensor layout, learning/engine.py:89-90) <-> NDHWC (library layout).
#include "common.hpp"

template <typename T>
__global__ void ncdhw_to_ndhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int C, int cpad, int pitch,
                                      size_t voxels) {
  const int n = blockIdx.y;
  const float* sb = src + (size_t)n * C * voxels;
  T* db = dst + (size_t)n * voxels * pitch;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < voxels; v += (size_t)gridDim.x * blockDim.x) {
    for (int c = 0; c < cpad; ++c) db[v * pitch + c] = from_f<T>(c < C ? sb[(size_t)c * voxels + v] : 0.f);
  }
}

template <typename T>
__global__ void ndhwc_to_ncdhw_kernel(const T* __restrict__ src, int pitch, float* __restrict__ dst, int C, size_t voxels) {
  const int n = blockIdx.y;
  const T* sb = src + (size_t)n * voxels * pitch;
  float* db = dst + (size_t)n * C * voxels;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < voxels; v += (size_t)gridDim.x * blockDim.x) {
    for (int c = 0; c < C; ++c) db[(size_t)c * voxels + v] = to_f<T>(sb[v * pitch + c]);
  }
}

static inline int lgrid(size_t total) {
  size_t b = (total + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

extern "C" int brats_ncdhw_to_ndhwc(const float* src, void* dst, int dtype, int N, int C, int cpad, int dst_pitch, int D,
                                    int H, int W, brats_stream_t s) {
  if (!src || !dst || cpad < C || dst_pitch < cpad) BRATS_FAIL(BRATS_E_ARG, "ncdhw_to_ndhwc: bad argument");
  const size_t vox = (size_t)D * H * W;
  dim3 grid(lgrid(vox), N);
  if (dtype == BRATS_BF16)
    hipLaunchKernelGGL(ncdhw_to_ndhwc_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)s, src, (bf16_t*)dst, C, cpad, dst_pitch, vox);
  else
    hipLaunchKernelGGL(ncdhw_to_ndhwc_kernel<float>, grid, dim3(256), 0, (hipStream_t)s, src, (float*)dst, C, cpad, dst_pitch, vox);
  BRATS_CHECK_LAUNCH();
  return 0;
}

extern "C" int brats_ndhwc_to_ncdhw(const void* src, int src_pitch, float* dst, int dtype, int N, int C, int D, int H, int W,
                                    brats_stream_t s) {
  if (!src || !dst || src_pitch < C) BRATS_FAIL(BRATS_E_ARG, "ndhwc_to_ncdhw: bad argument");
  const size_t vox = (size_t)D * H * W;
  dim3 grid(lgrid(vox), N);
  if (dtype == BRATS_BF16)
    hipLaunchKernelGGL(ndhwc_to_ncdhw_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)s, (const bf16_t*)src, src_pitch, dst, C, vox);
  else
    hipLaunchKernelGGL(ndhwc_to_ncdhw_kernel<float>, grid, dim3(256), 0, (hipStream_t)s, (const float*)src, src_pitch, dst, C, vox);
  BRATS_CHECK_LAUNCH();
  return 0;
}

// ---- im2col / col2im for the large-dilation ASPP branches (d = 4, 6 at the coarsest level,
// networks/equiunet2021.py:257-259): col[n][v][tap*C + c] = x[n][v + d*off(tap)][c] (0 outside).  The
// 3x3x3 dilated conv then is a 1x1x1 implicit GEMM over 27*C channels; the halo of such dilations
// (tile + 2*6 voxels per side) would not fit LDS, while the tensor itself (16^3 x 384) is L2-resident.
template <typename T>
__global__ void im2col3_kernel(const T* __restrict__ x, int xpitch, T* __restrict__ col, int C, int D, int H, int W, int dil) {
  constexpr int VW = 16 / sizeof(T);
  const int n = blockIdx.y, cv = C / VW;
  const size_t vox_n = (size_t)D * H * W, total = vox_n * 27 * cv;
  const T* xb = x + (size_t)n * vox_n * xpitch;
  T* cb = col + (size_t)n * vox_n * 27 * C;
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (size_t)gridDim.x * blockDim.x) {
    const int c0 = (int)(it % cv) * VW;
    size_t t = it / cv;
    const int tap = t % 27;
    const size_t v = t / 27;
    const int xx = v % W, yy = (v / W) % H, zz = (int)(v / ((size_t)W * H));
    const int sz = zz + (tap / 9 - 1) * dil, sy = yy + ((tap / 3) % 3 - 1) * dil, sx = xx + (tap % 3 - 1) * dil;
    u32x4 val = {0u, 0u, 0u, 0u};
    if (sz >= 0 && sz < D && sy >= 0 && sy < H && sx >= 0 && sx < W)
      val = *(const u32x4*)(xb + (((size_t)sz * H + sy) * W + sx) * xpitch + c0);
    *(u32x4*)(cb + (v * 27 + tap) * C + c0) = val;
  }
}

template <typename T>
__global__ void col2im3_kernel(const T* __restrict__ dcol, T* __restrict__ dx, int dxpitch, int C, int D, int H, int W, int dil) {
  constexpr int VW = 16 / sizeof(T);
  const int n = blockIdx.y, cv = C / VW;
  const size_t vox_n = (size_t)D * H * W, total = vox_n * cv;
  const T* cb = dcol + (size_t)n * vox_n * 27 * C;
  T* xb = dx + (size_t)n * vox_n * dxpitch;
  for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < total; it += (size_t)gridDim.x * blockDim.x) {
    const int c0 = (int)(it % cv) * VW;
    const size_t v = it / cv;
    const int xx = v % W, yy = (v / W) % H, zz = (int)(v / ((size_t)W * H));
    float acc[VW];
#pragma unroll
    for (int j = 0; j < VW; ++j) acc[j] = 0.f;
    for (int tap = 0; tap < 27; ++tap) {
      // x[v] was read by output voxel o = v - d*off(tap)
      const int oz = zz - (tap / 9 - 1) * dil, oy = yy - ((tap / 3) % 3 - 1) * dil, ox = xx - (tap % 3 - 1) * dil;
      if (oz >= 0 && oz < D && oy >= 0 && oy < H && ox >= 0 && ox < W) {
        float a[VW];
        Vec<T, VW>::load(cb + ((((size_t)oz * H + oy) * W + ox) * 27 + tap) * C + c0, a);
#pragma unroll
        for (int j = 0; j < VW; ++j) acc[j] += a[j];
      }
    }
    Vec<T, VW>::store(xb + v * dxpitch + c0, acc);
  }
}

extern "C" int brats_im2col3(const void* x, int xpitch, void* col, int dtype, int N, int C, int D, int H, int W, int dil,
                             brats_stream_t s) {
  const int vw = dtype == BRATS_BF16 ? 8 : 4;
  if (!x || !col || C % vw || xpitch % vw || dil < 1) BRATS_FAIL(BRATS_E_ARG, "im2col3: bad argument");
  dim3 grid(lgrid((size_t)D * H * W * 27 * (C / vw)), N);
  if (dtype == BRATS_BF16)
    hipLaunchKernelGGL(im2col3_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, xpitch, (bf16_t*)col, C, D, H, W, dil);
  else
    hipLaunchKernelGGL(im2col3_kernel<float>, grid, dim3(256), 0, (hipStream_t)s, (const float*)x, xpitch, (float*)col, C, D, H, W, dil);
  BRATS_CHECK_LAUNCH();
  return 0;
}

extern "C" int brats_col2im3(const void* dcol, void* dx, int dxpitch, int dtype, int N, int C, int D, int H, int W, int dil,
                             brats_stream_t s) {
  const int vw = dtype == BRATS_BF16 ? 8 : 4;
  if (!dcol || !dx || C % vw || dxpitch % vw || dil < 1) BRATS_FAIL(BRATS_E_ARG, "col2im3: bad argument");
  dim3 grid(lgrid((size_t)D * H * W * (C / vw)), N);
  if (dtype == BRATS_BF16)
    hipLaunchKernelGGL(col2im3_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)s, (const bf16_t*)dcol, (bf16_t*)dx, dxpitch, C, D, H, W, dil);
  else
    hipLaunchKernelGGL(col2im3_kernel<float>, grid, dim3(256), 0, (hipStream_t)s, (const float*)dcol, (float*)dx, dxpitch, C, D, H, W, dil);
  BRATS_CHECK_LAUNCH();
  return 0;
}
