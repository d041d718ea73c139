// Probe (gfx950): lane <-> element map of ds_read_b64_tr_b8 (the 8-bit transposing LDS read).  Hypothesis by analogy with
// the 16-bit form: per 16-lane group a block of 8 rows x 16 byte-columns; lane 2r + h supplies the address of row r,
// columns 8h..8h+7; lane i receives column i, row r in byte r.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v2i __attribute__((ext_vector_type(2)));
__global__ void k(uint8_t* out, int stride) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = 0xEE;
  __syncthreads();
  // 4 blocks (one per 16-lane group g), block g at byte offset g * 1024, rows `stride` bytes apart: value = g*64... encode (row, col)
  for (int i = threadIdx.x; i < 4 * 8 * 16; i += 64) {
    const int g = i / 128, r = (i / 16) % 8, c = i % 16;
    lds[g * 1024 + r * stride + c] = (uint8_t)((g << 6) | (r << 3) | (c & 7) | ((c >> 3) << 7 & 0x80 ? 0 : 0));
  }
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, l = lane & 15;
  const uint8_t* addr = lds + g * 1024 + (l >> 1) * stride + (l & 1) * 8;
  v2i v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i*)addr);
  ((v2i*)out)[lane] = v;
}
int main() {
  uint8_t* d; hipMalloc(&d, 512);
  for (int stride : {16, 48}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride);
    uint8_t h[512]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("row stride %d B: value = g<<6 | row<<3 | (col & 7)   (col >= 8 and col < 8 share the low bits)\n", stride);
    for (int lane : {0, 1, 2, 7, 8, 9, 15, 16, 17, 40}) {
      printf("  lane %2d:", lane);
      for (int b = 0; b < 8; ++b) printf(" g%d r%d c%d", h[lane * 8 + b] >> 6, (h[lane * 8 + b] >> 3) & 7, h[lane * 8 + b] & 7);
      printf("\n");
    }
  }
  return 0;
}
