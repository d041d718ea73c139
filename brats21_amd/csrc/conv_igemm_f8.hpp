// fp8 (OCP e4m3) variant of the implicit-GEMM 3x3x3 convolution of conv_igemm.hpp (BASELINE.json configs[4]:
// "fp16 + fp8 MFMA conv path").  Same GEMM view, tile and wave roles; what changes:
//   * activations stay bf16 NDHWC in HBM and are quantised to e4m3 while they are STAGED into LDS
//     (v_cvt_scalef32_pk_fp8_bf16: x / 2^e, RNE; e from the tensor's |max| so that the largest value lands in
//     [128, 256) -- the instruction does not saturate, > 464 becomes NaN, scripts/probes/fp8.hip).  The LDS halo tile
//     is half the bf16 size (CK bytes per voxel);
//   * weights are packed as e4m3 with one power-of-two scale per GEMM row (output channel);
//   * the MMA is v_mfma_f32_16x16x128_f8f6f4 (the MX instruction with K = 128, block scales 2^0): 2x the bf16
//     MFMA rate.  Lane (q, v) feeds row/voxel v with K elements 32q..32q+31 = two 16-channel "units" (one
//     ds_read_b128 each); which (tap, channel) a K element means is free as long as A and B agree;
//   * epilogue: acc * (2^e * wscale[cout]) + bias, then exactly the bf16 epilogue (statistics, NDHWC store).
// Roles: cout-split (2*NF*16 couts per workgroup, 8 voxel fragments per wave) or y-split (VS).  No K-split.
#pragma once
#include "conv_igemm.hpp"


struct ConvF8Params {
  ConvParams c;
  const float* amax1;   // |max| of x1 / x2 (device scalars); NULL -> xscale is used as given
  const float* amax2;
  float xscale;         // static activation scale (a power of two): x is quantised as x / xscale
  const float* wscale;  // per-row weight scales written by brats_conv3d_f8_pack_weights
};

template <int CK, int DIL>
struct F8Geom {
  static constexpr int R = DIL;
  static constexpr int HZ = CONV_TZ + 2 * R, HY = CONV_TY + 2 * R, HX = CONV_TX + 2 * R;
  static constexpr int HVOX = HZ * HY * HX;
  static constexpr int PPV = CK / 8;  // 16-byte bf16 pieces per voxel in HBM = 8-byte e4m3 pieces in LDS
  // voxel stride in LDS: a fragment read is 16 lanes x 16 B at consecutive x; conflict-free iff S/16 is odd
  static constexpr int S = (CK / 16) % 2 ? CK : CK + 16;
  static constexpr int UPT = CK / 16;  // 16-channel units per tap
  static constexpr int UNITS = 27 * UPT;
  static constexpr int MS = (UNITS + 7) / 8;  // macro-steps (one K=128 MFMA = 8 units) per chunk
  static constexpr int LDS_TILE = HVOX * S;
  static_assert(CK % 16 == 0, "fp8 chunks are multiples of 16 channels");
  static constexpr int tapoff(int tap) { return ((((tap / 9) * DIL) * HY + ((tap / 3) % 3) * DIL) * HX + (tap % 3) * DIL) * S; }
  static constexpr int unitoff(int g) { return g < UNITS ? tapoff(g / UPT) + (g % UPT) * 16 : 0; }
};

template <int NF, bool VS> struct F8Tile {
  static constexpr int NFW = VS ? NF : 2 * NF;
  static constexpr int NB = VS ? 4 : 8;
  static constexpr int SRED_BYTES = (VS ? 4 : 2) * NFW * 16 * 2 * 4;
};

template <int CK, int DIL, int NF, bool VS> constexpr int conv_f8_lds_bytes() {
  return (F8Geom<CK, DIL>::LDS_TILE + 15) / 16 * 16 + F8Tile<NF, VS>::SRED_BYTES;
}

// One Cin chunk of MFMA work.  Order inside a macro-step: cout fragment f outermost, so that a[f] is dead after its NB
// MFMAs and is refilled for a later step right away (one weight buffer for NB = 8, two for NB = 4 where a step is only
// NF*4 MFMAs long); the activation fragments b[i] are re-read from LDS for the next step behind the MFMAs of the last f.
template <int CK, int DIL, int NF, int NB>
DEVI void conv_f8_mma_chunk(const char* ldsb, int lane_b, int q, __amdgpu_buffer_rsrc_t rsw, int wbase /* scalar */, int rows16,
                            int lane, f32x4 (&acc)[NF][NB]) {
  using G = F8Geom<CK, DIL>;
  constexpr int YB = NB / 2;
  constexpr int FOZ = G::HY * G::HX * G::S;
  constexpr int NSTEP = G::MS;
  constexpr int AB = NB == 8 ? 1 : 2;
  // fragment (ms, f): two 1-KB halves [h][lane][16 B]; the fragment origin is scalar (soffset), the lane part fixed
  const int wlane = lane * 16;
  i32x8 a[AB][NF];
  i32x8 b[NB];
  auto load_a = [&](auto k_, auto f_) {
    constexpr int k = k_, f = f_;
    const int so = wbase + (k * rows16 + f) * 2048;
    const u32x4 lo = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, wlane, so, 0));
    const u32x4 hi = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsw, wlane, so + 1024, 0));
    a[k % AB][f] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
  };
  int lb0, lb1;
  // opaque per call: otherwise the 2 x MS per-step LDS addresses are hoisted out of the chunk loop as live registers
  asm volatile("" : "+v"(q), "+v"(lane_b));
  const bool qlo = q & 1, qhi = q & 2;
  auto set_lb = [&](auto k_) {
    constexpr int k = k_;
    constexpr int o0 = G::unitoff(8 * k), o1 = G::unitoff(8 * k + 1), o2 = G::unitoff(8 * k + 2), o3 = G::unitoff(8 * k + 3),
                  o4 = G::unitoff(8 * k + 4), o5 = G::unitoff(8 * k + 5), o6 = G::unitoff(8 * k + 6), o7 = G::unitoff(8 * k + 7);
    // two-level selects on the bits of q (a 4-way ternary chain is lowered to a branchy switch here)
    const int e0 = qlo ? o2 : o0, e1 = qlo ? o6 : o4, d0 = qlo ? o3 : o1, d1 = qlo ? o7 : o5;
    lb0 = lane_b + (qhi ? e1 : e0);
    lb1 = lane_b + (qhi ? d1 : d0);
  };
  auto read_b = [&](auto i_) {
    constexpr int i = i_;
    constexpr int ro = (i / YB) * FOZ + (i % YB) * G::HX * G::S;
    const u32x4 lo = *(const u32x4*)(ldsb + lb0 + ro), hi = *(const u32x4*)(ldsb + lb1 + ro);
    b[i] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
  };
  static_for<0, (AB < NSTEP ? AB : NSTEP)>([&](auto k_) { static_for<0, NF>([&](auto f_) { load_a(k_, f_); }); });
  set_lb(std::integral_constant<int, 0>{});
  static_for<0, NB>([&](auto i_) { read_b(i_); });
  static_for<0, NSTEP>([&](auto k_) {
    constexpr int k = k_;
    static_for<0, NF>([&](auto f_) {
      constexpr int f = f_;
      if constexpr (f == NF - 1 && k + 1 < NSTEP) set_lb(std::integral_constant<int, k + 1>{});
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, NB>([&](auto i_) {
        constexpr int i = i_;
        // constant-zero scale operands select the unscaled v_mfma_f32_16x16x128_f8f6f4 (= block scales 2^0, checked in
        // scripts/probes/fp8.hip; 32.0 instead of 33.5 cycles per MFMA and no scale VGPR, scripts/probes/mfma_rate.hip)
        acc[f][i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[k % AB][f], b[i], acc[f][i], 0, 0, 0, 0, 0, 0);
        if constexpr (f == NF - 1 && k + 1 < NSTEP) {
          read_b(i_);
          __builtin_amdgcn_sched_barrier(0);
        }
      });
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (k + AB < NSTEP) load_a(std::integral_constant<int, k + AB>{}, f_);
      __builtin_amdgcn_sched_barrier(0);
    });
  });
}

template <int CK, int DIL, int NF, bool VS>
__global__ __launch_bounds__(256, 2) void conv_igemm_f8_kernel(const ConvF8Params pp) {
  using G = F8Geom<CK, DIL>;
  using TL = F8Tile<NF, VS>;
  using T = bf16_t;
  constexpr int NB = TL::NB, YB = NB / 2;
  const ConvParams& p = pp.c;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  const int q = lane >> 4, v = lane & 15;

  int bt = blockIdx.x;
  const int tile_in_sample = bt % (p.tz * p.ty * p.tx);
  const int txi = bt % p.tx; bt /= p.tx;
  const int tyi = bt % p.ty; bt /= p.ty;
  const int tzi = bt % p.tz;
  const int n = bt / p.tz;
  const int z0 = tzi * CONV_TZ, y0 = tyi * CONV_TY, x0 = txi * CONV_TX;
  const int ct = blockIdx.y;
  const int f0 = ct * TL::NFW + (VS ? 0 : wn * NF);
  const size_t sample_vox = (size_t)n * p.D * p.H * p.W;

  float xs = pp.xscale;
  if (pp.amax1) {
    float am = *pp.amax1;
    if (pp.amax2) am = fmaxf(am, *pp.amax2);
    xs = f8_scale_from_amax(am);
  }

  constexpr int LDS_MAIN = (G::LDS_TILE + 15) / 16 * 16;  // statistics scratch of the epilogue; staging's dummy slot
  constexpr int NROWS = G::HZ * G::HY;
  constexpr int PPR = G::HX * G::PPV;
  constexpr int IPR = (PPR + 63) / 64;
  constexpr int RPW = (NROWS + 3) / 4;
  int lds_off[IPR];
  int hx_part[IPR];
#pragma unroll
  for (int j = 0; j < IPR; ++j) {
    const int pc = lane + 64 * j;
    const int hx = pc / G::PPV, part = pc % G::PPV;
    const int gx = x0 - G::R + hx;
    const bool ok = pc < PPR && gx >= 0 && gx < p.W;
    hx_part[j] = ok ? (hx << 16) | part : -1;
    lds_off[j] = pc < PPR ? wave * (G::HX * G::S) + hx * G::S + part * 8 : -1;
  }

  f32x4 acc[NF][NB];
#pragma unroll
  for (int f = 0; f < NF; ++f)
#pragma unroll
    for (int i = 0; i < NB; ++i) acc[f][i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int lane_b = ((wm * 2) * G::HY * G::HX + (VS ? wn * 2 * G::HX : 0) + v) * G::S;
  const int chunk_stride = G::MS * p.rows16 * 2048;  // bytes of packed weights per chunk
  const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.wpk, (short)0, p.nchunks * chunk_stride, 0x00020000);

  for (int chunk = 0; chunk < p.nchunks; ++chunk) {
    const int c0 = chunk * CK;
    const T* src;
    int pitch;
    if (c0 < p.c1) { src = (const T*)p.x1 + c0; pitch = p.p1; }
    else { src = (const T*)p.x2 + (c0 - p.c1); pitch = p.p2; }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(src + sample_vox * pitch), (short)0, (int)((size_t)p.D * p.H * p.W * pitch * 2), 0x00020000);
    const int pb = pitch * 2;
    int goff[IPR];
#pragma unroll
    for (int j = 0; j < IPR; ++j) goff[j] = (hx_part[j] >> 16) * pb + (hx_part[j] & 0xffff) * 16;
    u32x4 r[RPW][IPR];
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
      const int row = wave + 4 * k;
      const int hz = row / G::HY, hy = row % G::HY;
      const int gz = z0 - G::R + hz, gy = y0 - G::R + hy;
      const bool row_ok = row < NROWS && gz >= 0 && gz < p.D && gy >= 0 && gy < p.H;
      const int rb = ((gz * p.H + gy) * p.W + (x0 - G::R)) * pb;
#pragma unroll
      for (int j = 0; j < IPR; ++j) {
        const int vo = (row_ok && hx_part[j] >= 0) ? rb + goff[j] : -1;
        r[k][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, 0, 0));
      }
    }
    if (chunk > 0) __syncthreads();
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
      if (wave + 4 * k < NROWS) {
#pragma unroll
        for (int j = 0; j < IPR; ++j)  // lanes without a piece write to a scratch slot (a select, not a branch around the cvt)
          *(u32x2*)(lds + (lds_off[j] >= 0 ? lds_off[j] + k * 4 * (G::HX * G::S) : LDS_MAIN)) = f8_quant8(r[k][j], xs);
      }
    }
    __syncthreads();
    conv_f8_mma_chunk<CK, DIL, NF, NB>(lds, lane_b, q, rsw, chunk * chunk_stride + f0 * 2048, p.rows16, lane, acc);
  }

  // --- epilogue: de-scale, bias, per-channel tile statistics, NDHWC bf16 store (as conv_igemm_kernel) ---
  float* sred = (float*)(lds + LDS_MAIN);
  {
    const bool x_ok = x0 + v < p.W;
    const bool second = p.y2 != nullptr && f0 * 16 >= p.ysplit;
    T* const ydst = second ? (T*)p.y2 : (T*)p.y;
    const int ypit = second ? p.y2pitch : p.ypitch;
    const int csub = second ? p.ysplit : 0;
    const int lane_o = (x0 + v) * ypit + 4 * q - csub;
    float bias[NF][4], mul[NF][4], s1[NF][4], s2[NF][4];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int cbase = (f0 + f) * 16 + 4 * q;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        bias[f][r] = (p.bias && cbase < p.cout) ? p.bias[cbase + r] : 0.f;
        mul[f][r] = xs * pp.wscale[cbase + r];  // wscale has rows16*16 entries
        s1[f][r] = 0.f;
        s2[f][r] = 0.f;
      }
    }
    const bool full = z0 + CONV_TZ <= p.D && y0 + CONV_TY <= p.H && x0 + CONV_TX <= p.W && (ct + 1) * TL::NFW * 16 <= p.cout;
    if (full) {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int z = z0 + 2 * wm + (i / YB), y = y0 + (VS ? 2 * wn : 0) + (i % YB);
        T* rowp = ydst + (sample_vox + (size_t)(z * p.H + y) * p.W) * ypit;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          float o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            o[r] = acc[f][i][r] * mul[f][r] + bias[f][r];
            s1[f][r] += o[r];
            s2[f][r] = __builtin_fmaf(o[r], o[r], s2[f][r]);
          }
          Vec<T, 4>::store(rowp + lane_o + (f0 + f) * 16, o);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int z = z0 + 2 * wm + (i / YB), y = y0 + (VS ? 2 * wn : 0) + (i % YB);
        const bool ok = z < p.D && y < p.H && x_ok;
        const float mk = ok ? 1.f : 0.f;
        T* rowp = ydst + (sample_vox + (size_t)(z * p.H + y) * p.W) * ypit;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
          const bool cok = (f0 + f) * 16 + 4 * q < p.cout;
          const float mf = cok ? mk : 0.f;
          float o[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            o[r] = acc[f][i][r] * mul[f][r] + bias[f][r];
            const float om = o[r] * mf;
            s1[f][r] += om;
            s2[f][r] = __builtin_fmaf(om, o[r], s2[f][r]);
          }
          if (ok && cok) Vec<T, 4>::store(rowp + lane_o + (f0 + f) * 16, o);
        }
      }
    }
    if (p.stats) {
#pragma unroll
      for (int f = 0; f < NF; ++f) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s1[f][r] = row16_sum(s1[f][r]);
          s2[f][r] = row16_sum(s2[f][r]);
        }
        if (v == 0) {
          const int cl = (f0 + f - ct * TL::NFW) * 16 + 4 * q;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            sred[(((VS ? wm + 2 * wn : wm) * TL::NFW * 16) + cl + r) * 2 + 0] = s1[f][r];
            sred[(((VS ? wm + 2 * wn : wm) * TL::NFW * 16) + cl + r) * 2 + 1] = s2[f][r];
          }
        }
      }
    }
  }
  if (p.stats) {
    __syncthreads();
    if (tid < TL::NFW * 16) {
      const int c = ct * TL::NFW * 16 + tid;
      if (c < p.cout) {
        const size_t tps = (size_t)p.tz * p.ty * p.tx;
        float* dst = p.stats + (((size_t)n * tps + tile_in_sample) * p.cout + c) * 2;
        float t1 = sred[tid * 2] + sred[(TL::NFW * 16 + tid) * 2];
        float t2 = sred[tid * 2 + 1] + sred[(TL::NFW * 16 + tid) * 2 + 1];
        if constexpr (VS) {
          t1 += sred[(2 * TL::NFW * 16 + tid) * 2] + sred[(3 * TL::NFW * 16 + tid) * 2];
          t2 += sred[(2 * TL::NFW * 16 + tid) * 2 + 1] + sred[(3 * TL::NFW * 16 + tid) * 2 + 1];
        }
        dst[0] = t1;
        dst[1] = t2;
      }
    }
  }
}

// ---- host-side dispatch -----------------------------------------------------------------------
template <int CK, int DIL, int NF, bool VS>
int conv_f8_launch_one(const ConvF8Params& p, hipStream_t st) {
  constexpr int lds = conv_f8_lds_bytes<CK, DIL, NF, VS>();
  auto kern = conv_igemm_f8_kernel<CK, DIL, NF, VS>;
  static std::atomic<uint64_t> attr_done{0};
  BRATS_ENSURE_LDS_ATTR(kern, lds, attr_done);
  dim3 grid((unsigned)(p.c.N * p.c.tz * p.c.ty * p.c.tx), (unsigned)(p.c.rows16 / F8Tile<NF, VS>::NFW));
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
  BRATS_CHECK_LAUNCH();
  return 0;
}

template <int CK, int DIL>
int conv_f8_launch_ck(const ConvF8Params& p, hipStream_t st) {
  const int r16 = p.c.rows16;
  if (r16 % 6 == 0) {
    if ((long)p.c.N * p.c.tz * p.c.ty * p.c.tx * (r16 / 6) < conv_small_grid_threshold()) return conv_f8_launch_one<CK, DIL, 3, true>(p, st);
    return conv_f8_launch_one<CK, DIL, 3, false>(p, st);
  }
  if (r16 % 3 == 0) return conv_f8_launch_one<CK, DIL, 3, true>(p, st);
  if (r16 % 4 == 0) return conv_f8_launch_one<CK, DIL, 2, false>(p, st);
  if (r16 % 2 == 0) return conv_f8_launch_one<CK, DIL, 2, true>(p, st);
  return conv_f8_launch_one<CK, DIL, 1, true>(p, st);
}

// implemented in conv_f8_k3_d<DIL>.hip
template <int DIL> int conv_f8_launch(const ConvF8Params& p, int ck, hipStream_t st);

#define CONV_F8_DEFINE_LAUNCH(DIL)                                                           \
  template <> int conv_f8_launch<DIL>(const ConvF8Params& p, int ck, hipStream_t st) {       \
    switch (ck) {                                                                            \
      case 48: return conv_f8_launch_ck<48, DIL>(p, st);                                     \
      case 32: return conv_f8_launch_ck<32, DIL>(p, st);                                     \
      case 16: return conv_f8_launch_ck<16, DIL>(p, st);                                     \
    }                                                                                        \
    BRATS_FAIL(BRATS_E_UNSUPPORTED, "conv fp8: unsupported channel chunk %d", ck);           \
  }
