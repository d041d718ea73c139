#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
__global__ void k(const char* p, u32x4* out, int bytes) {
  __shared__ __attribute__((aligned(16))) char lds[2048];
  u32x4* l4 = (u32x4*)lds;
  l4[threadIdx.x] = u32x4{0xABABABABu, 0xABABABABu, 0xABABABABu, 0xABABABABu};
  l4[threadIdx.x + 64] = u32x4{0xCDCDCDCDu, 0xCDCDCDCDu, 0xCDCDCDCDu, 0xCDCDCDCDu};
  __syncthreads();
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p, (short)0, bytes, 0x00020000);
  int voff = threadIdx.x * 16;
  if (threadIdx.x % 4 == 1) voff = -1;          // OOB sentinel
  if (threadIdx.x % 4 == 2) voff = bytes + 64;  // beyond num_records
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  out[threadIdx.x] = l4[threadIdx.x];
  out[threadIdx.x + 64] = l4[threadIdx.x + 64];
}
int main() {
  char* d; u32x4* o; hipMalloc(&d, 4096); hipMalloc(&o, 2048);
  uint32_t h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 0x1000 + i;
  hipMemcpy(d, h, 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 512);
  uint32_t r[512]; hipMemcpy(r, o, 2048, hipMemcpyDeviceToHost);
  for (int i = 0; i < 12; ++i) printf("lane %d: %08x %08x %08x %08x\n", i, r[4*i], r[4*i+1], r[4*i+2], r[4*i+3]);
  printf("lane 40: %08x  second half[0]: %08x\n", r[160], r[256]);
  return 0;
}
