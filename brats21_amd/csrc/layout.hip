// NCDHW f32 (the reference's tensor layout, learning/engine.py:89-90) <-> NDHWC (library layout).
#include "twin_begin.hpp"
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

// the network input: cpad * sizeof(T) == 16 (4 modalities -> 8 x 16-bit or 4 x f32 channels): the C plane loads of a voxel in
// flight together, ONE 16-byte store (the generic form above stores element by element behind one load each)
template <typename T>
__global__ void __launch_bounds__(256) ncdhw_to_ndhwc_vec_kernel(const float* __restrict__ src, T* __restrict__ dst, int C, int pitch,
                                                                 size_t voxels) {
  constexpr int VW = 16 / sizeof(T);
  const int n = blockIdx.y;
  const float* sb = src + (size_t)n * C * voxels;
  T* db = dst + (size_t)n * voxels * pitch;
  for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < voxels; v += (size_t)gridDim.x * blockDim.x) {
    float a[VW];
#pragma unroll
    for (int c = 0; c < VW; ++c) a[c] = c < C ? sb[(size_t)c * voxels + v] : 0.f;
    Vec<T, VW>::store(db + v * pitch, a);
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

extern "C" int BRATS_API(brats_ncdhw_to_ndhwc)(const float* src, void* dst, int dtype, int N, int C, int cpad, int dst_pitch, int D,
                                    int H, int W, brats_stream_t s) {
  if (!src || !dst || cpad < C || dst_pitch < cpad) BRATS_FAIL(BRATS_E_ARG, "ncdhw_to_ndhwc: bad argument");
  const size_t vox = (size_t)D * H * W;
  dim3 grid(lgrid(vox), N);
  const int vw = dtype == BRATS_BF16 ? 8 : 4;
  if (cpad == vw && dst_pitch % vw == 0 && ((uintptr_t)dst & 15) == 0) {
    if (dtype == BRATS_BF16)
      hipLaunchKernelGGL(ncdhw_to_ndhwc_vec_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)s, src, (bf16_t*)dst, C, dst_pitch, vox);
    else
      hipLaunchKernelGGL(ncdhw_to_ndhwc_vec_kernel<float>, grid, dim3(256), 0, (hipStream_t)s, src, (float*)dst, C, dst_pitch, vox);
    BRATS_CHECK_LAUNCH();
    return 0;
  }
  if (dtype == BRATS_BF16)
    hipLaunchKernelGGL(ncdhw_to_ndhwc_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)s, src, (bf16_t*)dst, C, cpad, dst_pitch, vox);
  else
    hipLaunchKernelGGL(ncdhw_to_ndhwc_kernel<float>, grid, dim3(256), 0, (hipStream_t)s, src, (float*)dst, C, cpad, dst_pitch, vox);
  BRATS_CHECK_LAUNCH();
  return 0;
}

extern "C" int BRATS_API(brats_ndhwc_to_ncdhw)(const void* src, int src_pitch, float* dst, int dtype, int N, int C, int D, int H, int W,
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
#include "twin_end.hpp"
