// nn.Dropout(p) behind the activation of a ConvBnRelu unit (networks/equiunet2020.py:62,72; --dropout of the unchanged CLI,
// src/arguments_train.py:52): out = x * keep / (1 - p), keep ~ Bernoulli(1 - p) per element.
// The mask is never stored: it is a pure function of (seed, step counter, unit id, logical element index) -- Philox4x32-10, one
// call per 4 consecutive channels of a voxel -- so the backward pass regenerates exactly the forward's mask by running the same
// kernel on the incoming gradient (d/dx = the same multiplier).  seed and step counter are read from DEVICE memory
// (state[0], state[1]): a step replayed from a hipGraph draws a new mask when the counter tensor is advanced inside the graph.
// The stream of random numbers is this library's, not torch's: parity with the reference is statistical (keep rate, scaling,
// mask shared by forward and backward) plus exactness given the mask (tests/test_equiunet_gpu.py).
#include <stdlib.h>
#include "twin_begin.hpp"
#include "common.hpp"

DEVI void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&r)[4]) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}

template <typename T>
__global__ void __launch_bounds__(256) dropout_kernel(const T* __restrict__ x, int xpitch, T* __restrict__ out, int opitch, size_t voxels,
                                                      int C, uint32_t thresh24, float scale, const unsigned long long* __restrict__ state,
                                                      uint32_t unit) {
  constexpr int VW = 16 / sizeof(T);
  const int cv = C / VW;
  const size_t total = voxels * cv;
  const unsigned long long seed = state[0], step = state[1];
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32) ^ (uint32_t)(step >> 32);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t vox = i / cv;
    const int c0 = (int)(i - vox * cv) * VW;
    float a[VW];
    Vec<T, VW>::load(x + vox * xpitch + c0, a);
    const unsigned long long e4 = (vox * (size_t)C + c0) >> 2;  // index of the element quad in the logical [voxels][C] tensor
#pragma unroll
    for (int h = 0; h < VW / 4; ++h) {
      uint32_t r[4];
      const unsigned long long ctr = e4 + h;
      philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), unit, (uint32_t)step, k0, k1, r);
#pragma unroll
      for (int j = 0; j < 4; ++j) a[4 * h + j] = (r[j] >> 8) >= thresh24 ? a[4 * h + j] * scale : 0.f;
    }
    Vec<T, VW>::store(out + vox * opitch + c0, a);
  }
}

extern "C" int BRATS_API(brats_dropout)(const void* x, int xpitch, void* out, int opitch, int dtype, size_t voxels, int C, float p,
                                        const void* state, int unit, brats_stream_t s) {
  if (!x || !out || !state || C <= 0) BRATS_FAIL(BRATS_E_ARG, "dropout: null pointer / bad size");
  if (!(p >= 0.f && p < 1.f)) BRATS_FAIL(BRATS_E_ARG, "dropout: p = %g must be in [0, 1)", (double)p);
  if (dtype != BRATS_BF16 && dtype != BRATS_F32) BRATS_FAIL(BRATS_E_UNSUPPORTED, "dropout: dtype %d", dtype);
  const int vw = dtype == BRATS_BF16 ? 8 : 4;
  if (C % vw || xpitch % vw || opitch % vw) BRATS_FAIL(BRATS_E_ARG, "dropout: C and pitches must be multiples of %d", vw);
  const size_t total = voxels * (size_t)(C / vw);
  if (!total) return 0;
  size_t nb = (total + 255) / 256;
  const unsigned blocks = (unsigned)(nb > 16384 ? 16384 : nb);
  const uint32_t thresh = (uint32_t)((double)p * 16777216.0);  // keep <=> 24 random bits >= p * 2^24
  const float scale = 1.f / (1.f - p);
  hipStream_t st = (hipStream_t)s;
  if (dtype == BRATS_BF16)
    hipLaunchKernelGGL(dropout_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)x, xpitch, (bf16_t*)out, opitch, voxels, C,
                       thresh, scale, (const unsigned long long*)state, (uint32_t)unit);
  else
    hipLaunchKernelGGL(dropout_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)x, xpitch, (float*)out, opitch, voxels, C,
                       thresh, scale, (const unsigned long long*)state, (uint32_t)unit);
  BRATS_CHECK_LAUNCH();
  return 0;
}
#include "twin_end.hpp"
