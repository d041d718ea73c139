// Shared device/host helpers for libbrats_hip.so (gfx950 only).
//
// 16-bit storage comes in two flavours built from the SAME sources: every translation unit that touches 16-bit activations
// is compiled twice -- once as is (bf16: BRATS_BF16) and once with -DBRATS_FP16 (IEEE half: BRATS_F16, the reference's own
// autocast dtype, learning/engine.py:304), where bf16_t / bf2f / f2bf / the MFMA macros below mean fp16 and everything lives
// in namespace brats_f16 with its extern "C" entry points renamed name##_f16 (twin_begin.hpp / twin_end.hpp).  The normal
// entry points forward dtype == BRATS_F16 calls to their twins (BRATS_F16_FORWARD).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include <type_traits>
#include "../../include/brats_hip.h"

typedef uint16_t bf16_t;  // raw 16-bit storage (bf16; fp16 in the -DBRATS_FP16 twin)
#ifdef BRATS_FP16
typedef __attribute__((ext_vector_type(8))) _Float16 bf16x8;  // (names kept: "the 16-bit MFMA fragment")
typedef __attribute__((ext_vector_type(4))) _Float16 bf16x4;
#else
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
#endif
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define DEVI __device__ __forceinline__

#ifdef BRATS_FP16
DEVI float bf2f(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
DEVI bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (_Float16)f); }  // v_cvt_f16_f32 (RNE; > 65504 -> inf)
DEVI void unpack2(uint32_t w, float& lo, float& hi) { lo = bf2f((bf16_t)(w & 0xffffu)); hi = bf2f((bf16_t)(w >> 16)); }
#define MFMA16_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#define MFMA16_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#else
DEVI float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
DEVI bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32 (RNE, NaN-preserving)
  return __builtin_bit_cast(bf16_t, b);
}
DEVI void unpack2(uint32_t w, float& lo, float& hi) { lo = __uint_as_float(w << 16); hi = __uint_as_float(w & 0xffff0000u); }
#define MFMA16_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define MFMA16_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif

// two floats -> one packed pair of the 16-bit type in ONE instruction (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32, RNE); the
// element-wise form (f2bf(lo) | f2bf(hi) << 16) costs three: a conversion each plus shift-and-or
typedef __attribute__((ext_vector_type(2))) float f32x2_;
#ifdef BRATS_FP16
typedef __attribute__((ext_vector_type(2))) _Float16 half2_;
#else
typedef __attribute__((ext_vector_type(2))) __bf16 half2_;
#endif
DEVI uint32_t pack2(float lo, float hi) {
  const f32x2_ v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, half2_));
}

template <typename T> DEVI float to_f(T v);
template <> DEVI float to_f<float>(float v) { return v; }
template <> DEVI float to_f<bf16_t>(bf16_t v) { return bf2f(v); }
template <typename T> DEVI T from_f(float v);
template <> DEVI float from_f<float>(float v) { return v; }
template <> DEVI bf16_t from_f<bf16_t>(float v) { return f2bf(v); }

// N contiguous elements <-> float registers (N*sizeof(T) must be 8 or 16 bytes, pointer aligned)
template <typename T, int N> struct Vec;
template <> struct Vec<float, 4> {
  static DEVI void load(const float* p, float* o) { f32x4 v = *(const f32x4*)p; o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3]; }
  static DEVI void store(float* p, const float* o) { f32x4 v = {o[0], o[1], o[2], o[3]}; *(f32x4*)p = v; }
  // (non-temporal forms: the f32 tensors of the split-precision mode at the 128^3 level are 805 MB each)
  static DEVI void load_nt(const float* p, float* o) { f32x4 v = __builtin_nontemporal_load((const f32x4*)p); o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3]; }
  static DEVI void store_nt(float* p, const float* o) { f32x4 v = {o[0], o[1], o[2], o[3]}; __builtin_nontemporal_store(v, (f32x4*)p); }
};
template <> struct Vec<bf16_t, 4> {
  static DEVI void load(const bf16_t* p, float* o) {
    u32x2 v = *(const u32x2*)p;
    unpack2(v[0], o[0], o[1]);
    unpack2(v[1], o[2], o[3]);
  }
  static DEVI void store(bf16_t* p, const float* o) {
    u32x2 v;
    v[0] = pack2(o[0], o[1]);
    v[1] = pack2(o[2], o[3]);
    *(u32x2*)p = v;
  }
};
template <> struct Vec<bf16_t, 8> {
  static DEVI void load(const bf16_t* p, float* o) {
    u32x4 v = *(const u32x4*)p;
#pragma unroll
    for (int i = 0; i < 4; ++i) unpack2(v[i], o[2 * i], o[2 * i + 1]);
  }
  static DEVI void store(bf16_t* p, const float* o) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack2(o[2 * i], o[2 * i + 1]);
    *(u32x4*)p = v;
  }
  // non-temporal forms for tensors far larger than L2 + Infinity Cache that are streamed once per pass: measured on a
  // 2 x 403 MB read-modify-write pass (scripts/probes/stream_rw.hip) 5.5 -> 6.2-6.7 TB/s with both hints
  static DEVI void load_nt(const bf16_t* p, float* o) {
    u32x4 v = __builtin_nontemporal_load((const u32x4*)p);
#pragma unroll
    for (int i = 0; i < 4; ++i) unpack2(v[i], o[2 * i], o[2 * i + 1]);
  }
  static DEVI void store_nt(bf16_t* p, const float* o) {
    u32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack2(o[2 * i], o[2 * i + 1]);
    __builtin_nontemporal_store(v, (u32x4*)p);
  }
};
// streaming access with a compile-time policy (NT exists for the 16-byte vectors: bf16 x 8, f32 x 4)
template <typename T, int N, bool NT> DEVI void vload(const T* p, float* o) {
  if constexpr (NT && ((std::is_same<T, bf16_t>::value && N == 8) || (std::is_same<T, float>::value && N == 4))) Vec<T, N>::load_nt(p, o);
  else Vec<T, N>::load(p, o);
}
template <typename T, int N, bool NT> DEVI void vstore(T* p, const float* o) {
  if constexpr (NT && ((std::is_same<T, bf16_t>::value && N == 8) || (std::is_same<T, float>::value && N == 4))) Vec<T, N>::store_nt(p, o);
  else Vec<T, N>::store(p, o);
}
// tensors from this size on are streamed with the non-temporal hints: they cannot stay in the 256 MB Infinity Cache
// anyway.  Smaller ones must NOT be: a 100 MB tensor that the producer kernel has just written is served from the cache
// (affine_act at 2 x 64^3 x 96: 0.031 ms with plain loads, 0.037 ms with the hints)
static inline bool stream_nt(size_t bytes) { return bytes >= ((size_t)256 << 20); }

// ---- e4m3 helpers shared by the fp8 convolution (conv_igemm_f8.hpp) and the fp8 weight gradient (conv_wgrad.hip) ----
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(2))) short s16x2;
#ifdef BRATS_FP16
typedef __attribute__((ext_vector_type(2))) _Float16 bf16x2;
#define CVT_SCALE_PK_FP8_H __builtin_amdgcn_cvt_scalef32_pk_fp8_f16
#else
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
#define CVT_SCALE_PK_FP8_H __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16
#endif
DEVI u32x2 f8_quant8(u32x4 v, float scale) {  // 8 bf16 -> 8 e4m3 (value / scale), channel order kept
  // the elements go through scalars first: __builtin_bit_cast applied to a vector-element lvalue (v[1]) reads element 0
  // with this hipcc
  const uint32_t e0 = v[0], e1 = v[1], e2 = v[2], e3 = v[3];
  s16x2 lo = {0, 0}, hi = {0, 0};
  lo = CVT_SCALE_PK_FP8_H(lo, __builtin_bit_cast(bf16x2, e0), scale, false);
  lo = CVT_SCALE_PK_FP8_H(lo, __builtin_bit_cast(bf16x2, e1), scale, true);
  hi = CVT_SCALE_PK_FP8_H(hi, __builtin_bit_cast(bf16x2, e2), scale, false);
  hi = CVT_SCALE_PK_FP8_H(hi, __builtin_bit_cast(bf16x2, e3), scale, true);
  return u32x2{__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi)};
}

// power-of-two scale that puts amax into [128, 256) after division (e4m3 tops out at 448)
DEVI float f8_scale_from_amax(float amax) {
  uint32_t e = (__float_as_uint(amax) >> 23) & 0xffu;
  e = e == 255u ? 127u : (e < 8u ? 1u : e - 7u);  // inf/NaN: scale 1 (they convert to NaN and propagate)
  return __uint_as_float(e << 23);
}


// ---- split precision (conv_igemm_x3.hpp, wgrad_x3): a tensor whose values lie far outside fp16's range (gradients) is
// multiplied by a power of two on its way into the fp16 hi / lo split so that its |max| lands in [2^14, 2^15); the result is
// multiplied by the inverse.  Both are exact.  amax = 0 / inf / NaN: scale 1 (non-finite values propagate as they are).
DEVI float x3_scale_from_amax(float amax) {
  const uint32_t e = (__float_as_uint(amax) >> 23) & 0xffu;
  if (e == 0u || e == 255u) return 1.f;
  uint32_t se = 268u - e;        // 2^(14 - (e - 127))
  se = se > 227u ? 227u : se;    // (|max| < 2^-86: 2^100 at most, so that 1 / scale stays a normal number)
  return __uint_as_float(se << 23);
}
DEVI float x3_inv_scale(float scale) { return __uint_as_float((254u << 23) - __float_as_uint(scale)); }  // scale = 2^k -> 2^-k
// x3 split of two f32 values: h = the packed 16-bit pair {rn16(x0), rn16(x1)}, l = {rn16(x0 - h.lo), rn16(x1 - h.hi)}.  x - rn16(x) is
// exact in f32 (|x - h| <= ulp16 / 2), so  fma(h, -1, x)  IS the difference: in the fp16 build v_fma_mix_f32 takes the half straight
// out of the packed register (one instruction instead of v_cvt_f32_f16 + v_sub_f32; 12 instead of 16 VALU instructions per four
// values in the staging paths of conv_igemm_x3.hpp / conv_wgrad_x3.hpp, bit-identical results).
DEVI void x3_split2(float x0, float x1, uint32_t& h, uint32_t& l) {
  h = pack2(x0, x1);
#if defined(BRATS_FP16) && defined(__HIP_DEVICE_COMPILE__)
  float d0, d1;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d0) : "v"(h), "v"(x0));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d1) : "v"(h), "v"(x1));
  l = pack2(d0, d1);
#else
  float h0, h1;
  unpack2(h, h0, h1);
  l = pack2(x0 - h0, x1 - h1);
#endif
}

// ---- LDS-DMA (buffer_load ... lds) helpers shared by the weight-gradient and the loader-wave convolution kernels ----
// a VGPR value the optimiser must treat as new: keeps per-tile address arithmetic from being hoisted out of the tile loop
// into registers the accumulators need.  (The host pass of hipcc instantiates the kernel template's generic lambdas too
// and silently drops the kernel stub when it meets a "v" constraint there, hence the device-pass guard.)
#if defined(__HIP_DEVICE_COMPILE__)
#define OPAQUE_V(x) asm volatile("" : "+v"(x))
#else
#define OPAQUE_V(x) do { } while (0)
#endif

// one LDS-DMA instruction: lane l's 16 bytes at buffer offset `off` (out of range: zeros) land at dst + 16 l; dst wave-uniform.
// (A function of its own: called directly inside the kernel TEMPLATE's generic lambdas the builtin makes hipcc's host pass
// drop the kernel stub without a diagnostic.)
DEVI void lds_dma16(__amdgpu_buffer_rsrc_t rs, char* dst, int off) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, off, 0, 0, 0);
}
// The same instruction as inline assembly.  Why: hipcc treats every LDS-DMA as a store to ALL of LDS and puts
// s_waitcnt vmcnt(<everything issued so far>) in front of the next LDS read -- the transfers of the NEXT tile, issued right
// before the MFMA phase, were waited for before its first operand read, i.e. never overlapped with it (found in the ISA:
// vmcnt(0) between the last buffer_load ... lds and the first ds_read_b64_tr).  An asm statement is opaque to that pass;
// the kernels wait themselves (s_waitcnt vmcnt(0) before the barrier that precedes the first read of the buffer).
// M0 (the LDS destination base) is compiler-reserved: it is saved, written and restored inside the ONE statement that
// reads it (cdna_hip_programming.md 5.7) -- no "m0" clobber, which hipcc flags as undefined behaviour.
// rs = buffer descriptor words {base lo, base hi (stride 0), bytes, 0x00020000}; dst wave-uniform.
typedef int rsrc4_t __attribute__((ext_vector_type(4)));
DEVI rsrc4_t make_rsrc4(const void* base, unsigned bytes) {
  const size_t p = (size_t)base;
  return rsrc4_t{__builtin_amdgcn_readfirstlane((int)(unsigned)p), __builtin_amdgcn_readfirstlane((int)((p >> 32) & 0xffffu)),
                 __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
}
DEVI void lds_dma16_async(rsrc4_t rs, char* dst, int off) {
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned m = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)dst);
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(m), "v"(off), "s"(rs) : "memory");
#endif
}

// compile-time loop with a constexpr index
template <int I, int N, typename F> DEVI void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// ---- host side ------------------------------------------------------------------------------
// entry points of the 16-bit translation units are DEFINED as BRATS_API(brats_x) = brats_x_bf16 in the normal build and
// brats_x_f16 in the -DBRATS_FP16 twin; the public brats_x (include/brats_hip.h) is generated into twin_dispatch.cpp by
// gen_twin_dispatch.py: it picks the build by its dtype argument (BRATS_F16 -> the twin, called with BRATS_BF16)
#ifdef BRATS_FP16
#define BRATS_API(name) name##_f16
#else
#define BRATS_API(name) name##_bf16
#endif
#ifdef BRATS_FP16
}  // shared, non-twinned host helpers are declared outside namespace brats_f16
#endif
void brats_set_error(const char* fmt, ...);  // abi.hip
// out[i] = sum_{b < nb} part[b * total + i] in a fixed order (bitwise reproducible reductions without float atomics);
// implemented in dice.hip
int brats_ordered_sum(const float* part, float* out, int nb, int total, hipStream_t st);
int brats_ordered_sum2(const float* part, float* out1, int n1, float* out2, int nb, int total, hipStream_t st);
#ifdef BRATS_FP16
namespace brats_f16 {
#endif
#define BRATS_FAIL(code, ...) do { brats_set_error(__VA_ARGS__); return (code); } while (0)
#define BRATS_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) \
  BRATS_FAIL(BRATS_E_HIP, "%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize once per (kernel, device): `flag` is a static std::atomic<uint64_t> of the
// launch site, one bit per device ordinal (a racing second call sets the same value again: harmless).  The entry points
// may be called from any host thread and on any device (include/brats_hip.h).
#define BRATS_ENSURE_LDS_ATTR(kern, bytes, flag) do { \
  int dev_ = 0; (void)hipGetDevice(&dev_); const uint64_t bit_ = 1ull << (dev_ & 63); \
  if (!((flag).load(std::memory_order_acquire) & bit_)) { \
    hipError_t e_ = hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes)); \
    if (e_ != hipSuccess) BRATS_FAIL(BRATS_E_HIP, "hipFuncSetAttribute(%d B LDS): %s", (int)(bytes), hipGetErrorString(e_)); \
    (flag).fetch_or(bit_, std::memory_order_release); } } while (0)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

