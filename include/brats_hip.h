/* brats_hip.h -- C ABI of libbrats_hip.so: the MI355X (gfx950) drop-in for the device arithmetic of
 * the BraTS21 3D U-Net hot path.
 *
 * The reference has NO native / FFI boundary: its hot path is stock torch.nn called
 * from Python (`get_model(args) -> nn.Module`, src/definer.py:37-174).  Each entry point below
 * therefore cites the torch.nn call site of the reference it replaces.  The Python side
 * (brats21_amd/) binds these with ctypes and keeps the reference's module / factory API.
 *
 * Conventions
 *   - activations are NDHWC ("voxel-major, channel-minor"): element (n,d,h,w,c) of a tensor with
 *     channel pitch P lives at ((((n*D+d)*H+h)*W+w)*P + c).  A pitch larger than the channel count
 *     lets producers write straight into a channel slice of a concat buffer (torch.cat removed).
 *   - dtype: BRATS_F32 = 0 (exact-f32 MFMA, the parity mode), BRATS_BF16 = 1 (bf16 storage, f32
 *     accumulate, the throughput mode), BRATS_F16 = 2 (IEEE half storage, f32 accumulate: the
 *     reference's own autocast dtype, learning/engine.py:304 -- same kernels, same rates, three more
 *     mantissa bits and a 65504 range: train it under a GradScaler like the reference does).
 *     Statistics / gradients of parameters are always f32.  Every entry point with a `dtype`
 *     argument accepts all three; the few that move 16-bit data WITHOUT a dtype argument exist
 *     twice: brats_x (bf16) and brats_x_f16.
 *     BRATS_X3_F16 = 4 / BRATS_X3_BF16 = 3 ("split precision"; accepted by the 3x3x3 convolution entry points
 *     brats_conv3d_chunk / _packed_bytes / _pack_weights(_multi) / _fwd / _wgrad(_ws_bytes) only): activations are F32
 *     tensors in HBM, each operand is split on its way into the MFMA as x = hi + lo (two fp16 / bf16 values) and the
 *     product taken as hi*hi + lo*hi + hi*lo on the 16-bit MFMA with f32 accumulation -- f32-class results (fp16 split:
 *     per-product error 2^-22, operands must lie inside fp16's range: the forward pass; bf16 split: 2^-16 over f32's
 *     whole range: the gradients) at 3/16 of the exact-f32 MFMA's cost.  The 1e-3-logit parity configuration at usable
 *     speed (model.precision = "x3").
 *   - every pointer is a DEVICE pointer owned by the caller (incl. workspaces); the library never
 *     allocates, never synchronises, launches only on the stream passed in (graph-capturable).
 *   - return 0 on success; <0 = BRATS_E_* (message via brats_last_error()).  Never throws.
 *   - callable from any host thread; ordering is by stream.
 */
#ifndef BRATS_HIP_H
#define BRATS_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* brats_stream_t; /* hipStream_t */

enum { BRATS_F32 = 0, BRATS_BF16 = 1, BRATS_F16 = 2, BRATS_X3_BF16 = 3, BRATS_X3_F16 = 4 };
enum { BRATS_E_ARG = -1, BRATS_E_UNSUPPORTED = -2, BRATS_E_HIP = -3 };
/* --act of the reference (src/arguments_train.py:49-50; MONAI Act factory): relu, leakyrelu(slope), elu(alpha=1),
 * swish = x*sigmoid(x), mish = x*tanh(softplus(x)).  prelu = BRATS_ACT_LEAKY with the learnable slope read from device
 * memory (`slope_dev` of brats_affine_act / brats_gn_bwd_apply) + brats_prelu_slope_grad for its gradient. */
enum { BRATS_ACT_NONE = 0, BRATS_ACT_RELU = 1, BRATS_ACT_LEAKY = 2, BRATS_ACT_ELU = 3, BRATS_ACT_SWISH = 4, BRATS_ACT_MISH = 5 };

/* THE version: brats_abi_version() returns this define, the Python binding (brats21_amd/_lib.py) and tests/test_abi_cpu.py parse
 * it.  History: 2 since round 3 (a changed signature, brats_maxpool2_fwd); 3 in round 4 (additions only); 4 in round 4 (the block
 * table of brats_conv3d_pack_weights_multi changed meaning); 5 in round 5 (additions only: brats_conv3d_set_x3_wgrad_fused,
 * brats_dropout, brats_evonorm_bwd_tiles + its workspace query). */
#define BRATS_ABI_VERSION 6
int brats_abi_version(void);
const char* brats_last_error(void);

/* ---- dropout ---------------------------------------------------------------------------------
 * nn.Dropout(p) of a ConvBnRelu unit (networks/equiunet2020.py:62,72; --dropout, src/arguments_train.py:52):
 * out[v][c] = x[v][c] * keep / (1 - p) over `voxels` (N * D * H * W) voxels of C channels (NDHWC, pitches in elements; in place
 * allowed).  keep is a pure function of (state[0] = seed, state[1] = step counter, unit, v * C + c) -- Philox4x32-10 -- and
 * never stored: the backward pass calls the same function on the incoming gradient with the same arguments.  `state`: two
 * uint64 in DEVICE memory (read by the kernel, so a hipGraph replay sees an advanced counter).  dtype F32 | BF16 | F16. */
int brats_dropout(const void* x, int xpitch, void* out, int opitch, int dtype, size_t voxels, int C, float p,
                  const void* state, int unit, brats_stream_t s);

/* ---- layout ---------------------------------------------------------------------------------
 * NCDHW f32 (the reference's tensor layout, learning/engine.py:89) <-> NDHWC dtype.  `cpad` >= C
 * channels are written, the extra ones as zeros (first layer: 4 -> 8 so K is MFMA-friendly). */
int brats_ncdhw_to_ndhwc(const float* src, void* dst, int dtype, int N, int C, int cpad, int dst_pitch,
                         int D, int H, int W, brats_stream_t s);
int brats_ndhwc_to_ncdhw(const void* src, int src_pitch, float* dst, int dtype, int N, int C,
                         int D, int H, int W, brats_stream_t s);

/* ---- convolution 3x3x3 / 1x1x1, stride 1, "same" padding = dilation ---------------------------
 * Replaces nn.Conv3d at networks/equiunet2020.py:19-25 (conv3x3, dilation 1|2) and
 * networks/equiunet2021.py:197-206,169-171 (bias, dilation 1|2|4|6) by an implicit-GEMM MFMA
 * kernel; dgrad is the same kernel on weights packed with mode=BRATS_PACK_DGRAD.
 * The input is the virtual concat [x1 (c1 channels) | x2 (c2 channels, may be NULL/0)]
 * (torch.cat at equiunet2020.py:478-486 removed).  `stats` (may be NULL) receives per-tile,
 * per-channel sum / sum-of-squares of the (bias-added, f32) output:
 * [N][tiles_per_sample][cout][2], reduced by brats_gn_finalize / brats_evonorm_finalize. */
enum { BRATS_PACK_FWD = 0, BRATS_PACK_DGRAD = 1 };
/* bytes of the packed-weight buffer for a conv with `cin` (GEMM-K) and `cout` (GEMM-M) channels */
size_t brats_conv3d_packed_bytes(int dtype, int ksize, int cin, int cout, int ck);
/* channel chunk the kernel will use for an input made of c1 (+c2) channels and `cout` GEMM rows (output channels; for
 * the dgrad layout the input channels of the layer); 0 = unsupported */
int brats_conv3d_chunk(int dtype, int ksize, int dil, int c1, int c2, int cout);
/* w: [Cout_w][Cin_w][k][k][k] f32 (torch layout).  FWD: GEMM rows = Cout_w, K = Cin_w slice
 * [cin_off, cin_off+cin_cnt).  DGRAD: GEMM rows = Cin_w slice, K = Cout_w, taps flipped. */
int brats_conv3d_pack_weights(const float* w, void* packed, int dtype, int mode, int ksize,
                              int cout_w, int cin_w, int cin_off, int cin_cnt, int ck, brats_stream_t s);
/* All weight tensors of a network in ONE launch (a training step re-packs every layer twice, forward and dgrad layout:
 * 34 launches of ~9 us for EquiUnet).  jobs / blocks are device arrays the caller builds once: job j describes one
 * brats_conv3d_pack_weights call (cin_real < cin_w: input channels >= cin_real are zero padding that is not present in
 * w, whose channel pitch is cin_real); blocks[b] = {job, block index inside the job}; a job has
 * brats_conv3d_pack_blocks(kdim, ck, rows) blocks -- one per (K chunk, 16-row group, 8-row half): its torch-layout
 * weights are contiguous runs, staged coalesced through LDS (ABI 4; ABI <= 3: brats_conv3d_pack_block() output
 * elements per block, gathered from global memory). */
typedef struct {
  const float* w;
  void* out;
  int dtype, mode, taps, cin_w, cin_real, cin_off, rows, rows16, kdim, ck, ms_n, reserved;
  unsigned long long total; /* output elements of this job */
} brats_pack_job;
int brats_conv3d_pack_blocks(int kdim, int ck, int rows);
int brats_conv3d_pack_weights_multi(const brats_pack_job* jobs, const int* blocks /* [nblocks][2] */, int nblocks,
                                    brats_stream_t s);
int brats_conv3d_pack_weights_multi_f16(const brats_pack_job* jobs, const int* blocks /* [nblocks][2] */, int nblocks,
                                    brats_stream_t s);  /* the same on fp16 activations (job dtype BRATS_BF16 = 16-bit) */
int brats_conv3d_tiles_per_sample(int D, int H, int W);
/* ysplit of brats_conv3d_fwd must be a multiple of this (channels one wave owns for `cout` outputs) */
int brats_conv3d_split_granule(int cout);
/* bf16 3x3x3 dilation-1 layers with 48 (mod 96) output channels: 1 = 24-channel chunks + the 4x8x16-tile y-split kernel
 * (conv_igemm_vs8.hpp), 0 = 48-channel chunks + the 4x4x16-tile kernels, -1 = default (BRATS_CONV_VS8, on).  The setting
 * changes brats_conv3d_chunk(), i.e. the packed-weight layout: weights must be packed under the same setting they are
 * used with (the Python side offers ops.set_vs8(), which also drops its packed-weight caches).  Returns the previous setting.
 * (Round 3's mode 2, the loader-wave kernel, measured slower and left the library: scripts/probes/experiments/.) */
int brats_conv3d_set_vs8(int mode);
/* 16-bit 3x3x3 launches of at most one workgroup per CU (the 16^3 level: 32 tiles) with 48-channel chunks and the 48-cout
 * y-split roles: 1 = the 8-wave form (two K-parity teams of four waves per workgroup on one LDS tile and one output tile, team 1's
 * partial sums through LDS: conv_igemm.hpp "KP", round 6), 0 = the 4-wave workgroup, -1 = default (BRATS_CONV_KP, on).  Same
 * packed weights, same products; the K summation order differs (two partial sums per output).  Returns the previous setting. */
int brats_conv3d_set_kp(int mode);
/* y2 (may be NULL): second destination; output channels >= ysplit are written to y2 (channel index
 * minus ysplit, pitch y2pitch) -- the dgrad of a conv whose input was [x1 | x2] produces dx1 and dx2
 * as two dense tensors in one launch. */
int brats_conv3d_fwd(const void* x1, int c1, int pitch1, const void* x2, int c2, int pitch2,
                     const void* packed_w, const float* bias, void* y, int ypitch,
                     void* y2, int y2pitch, int ysplit, float* stats,
                     int dtype, int ksize, int dil, int N, int D, int H, int W, int cout,
                     brats_stream_t s);
/* "Normalise + activate on load" form of the 16-bit 3x3x3 convolution (inference): the inputs are RAW convolution outputs
 * of the producing layers; ss1 / ss2 ([N][c1][2] / [N][c2][2] f32 {scale, shift} per sample and channel, as written by
 * brats_gn_finalize; NULL = that source is read as it is) are applied as z = act(x * scale + shift) -- act BRATS_ACT_RELU or
 * BRATS_ACT_LEAKY(slope) -- while the halo tile is staged, bit-identical to brats_affine_act_fwd followed by
 * brats_conv3d_fwd; out-of-volume voxels enter as zeros.  The normalised activation is never stored: for a ConvBnRelu pair
 * inside a UBlock (networks/equiunet2020.py:105-123) under no_grad that removes one read + one write of the tensor.
 * brats_conv3d_pre_ok() = 1 where the form is built (Cout a multiple of 48, 48- or 24-channel chunks, dilation 1). */
int brats_conv3d_pre_ok(int dtype, int ksize, int dil, int c1, int c2, int cout);
int brats_conv3d_fwd_pre(const void* x1, int c1, int pitch1, const float* ss1, const void* x2, int c2, int pitch2,
                         const float* ss2, int act, float slope, const void* packed_w, const float* bias, void* y, int ypitch,
                         float* stats, int dtype, int dil, int N, int D, int H, int W, int cout, brats_stream_t s);
/* "Backward statistics" form of the 16-bit 3x3x3 convolution (training).  In a ConvBnRelu pair (networks/equiunet2020.py:
 * 105-123) the input gradient of the SECOND convolution is dz, the gradient of the first unit's activation
 * z = act(GN(fwd_y)); GroupNorm backward's first pass over (dz, fwd_y) -- per (sample, channel) sum u and sum u * xhat with
 * u = dz * act'(fwd_y * scale + shift) -- is taken here from the accumulators while the tile is stored: `tile_stats`
 * [N][tiles_per_sample][cout][2] receives sum u and sum u * fwd_y per 4x4x16 tile (the layout of brats_conv3d_fwd's `stats`),
 * and brats_gn_act_bwd_tiles() finishes from them.  fwd_y: the first unit's raw convolution output ([N][D][H][W][cout],
 * pitch fwd_pitch), scale_shift [N][cout][2] as written by brats_gn_finalize; act BRATS_ACT_RELU or BRATS_ACT_LEAKY(slope).
 * Saves one read of dz and of fwd_y per block (2 x 403 MB at the 128^3 level of EquiUnet-48).  Single source, no bias, no
 * second destination; brats_conv3d_bstats_ok() = 1 where the form is built (cout a multiple of 48; c1 = the K channels). */
int brats_conv3d_bstats_ok(int dtype, int ksize, int dil, int c1, int cout);
int brats_conv3d_fwd_bstats(const void* x1, int c1, int pitch1, const void* packed_w, void* y, int ypitch,
                            const void* fwd_y, int fwd_pitch, const float* scale_shift, int act, float slope,
                            float* tile_stats, int dtype, int dil, int N, int D, int H, int W, int cout, brats_stream_t s);
/* Split-precision form with an input scale (dtype BRATS_X3_F16 / BRATS_X3_BF16 only; 3x3x3): `xamax` (may be NULL = no
 * scaling) is a device scalar holding max|x| of the input -- written by the kernel that produced the tensor
 * (brats_gn_bwd_apply & co., brats_absmax).  The input is multiplied by the power of two that puts |max| into [2^14, 2^15)
 * before it is split into 16-bit pairs and the result by its inverse (both exact): this is how the INPUT GRADIENT
 * (weights packed BRATS_PACK_DGRAD, x = dY, values of 1e-6 and below) keeps f32-class accuracy on fp16 pairs, where the
 * reference needs a GradScaler (learning/engine.py:117-122).  Same nn.Conv3d call sites as brats_conv3d_fwd. */
int brats_conv3d_x3_fwd(const void* x1, int c1, int pitch1, const void* x2, int c2, int pitch2, const float* xamax,
                        const void* packed_w, const float* bias, void* y, int ypitch, void* y2, int y2pitch, int ysplit,
                        float* stats, int dtype, int dil, int N, int D, int H, int W, int cout, brats_stream_t s);
/* brats_conv3d_fwd_bstats for the split-precision kernels (dtype BRATS_X3_F16 / BRATS_X3_BF16; f32 tensors, fwd_y f32 and
 * 16-byte aligned): the input gradient of a block's second convolution in the parity mode, scaled through `xamax` like
 * brats_conv3d_x3_fwd, leaving the first-pass sums of the first unit's GroupNorm / EvoNorm backward per tile -- there the pass
 * it replaces reads 2 x 4 bytes per element.  brats_conv3d_bstats_ok(dtype = BRATS_X3_*) = 1 where the form is built (cout a
 * multiple of 48).  Autograd of networks/equiunet2020.py:105-123 / equiunet2021.py:197-206 under model.precision = "x3". */
int brats_conv3d_x3_fwd_bstats(const void* x1, int c1, int pitch1, const float* xamax, const void* packed_w, void* y, int ypitch,
                               const void* fwd_y, int fwd_pitch, const float* scale_shift, int act, float slope,
                               float* tile_stats, int dtype, int dil, int N, int D, int H, int W, int cout, brats_stream_t s);
/* ---- fp8 (OCP e4m3) variant of the 3x3x3 convolution (BASELINE.json configs[4], "fp8 MFMA conv path"): bf16 NDHWC
 * activations in and out; the input is quantised to e4m3 while it is staged (x / 2^e, round-to-nearest-even, e chosen
 * from the tensor's |max| so that it lands in [128, 256)), the weights are packed as e4m3 with one power-of-two scale
 * per output channel, the MMA is v_mfma_scale_f32_16x16x128_f8f6f4 with f32 accumulation.  Replaces the same
 * nn.Conv3d / dgrad as brats_conv3d_fwd.  Channel counts must be multiples of 16. */
int brats_conv3d_f8_chunk(int c1, int c2);                      /* 48, 32, 16 or 0 = unsupported */
size_t brats_conv3d_f8_packed_bytes(int cin, int cout, int ck); /* scales [ceil16(cout)] f32 + e4m3 fragments */
int brats_conv3d_f8_pack_weights(const float* w, void* packed, int mode, int cout_w, int cin_w, int cin_off,
                                 int cin_cnt, int ck, brats_stream_t s);
/* amax1 / amax2: device scalars holding max|x1| / max|x2| (written by brats_affine_act_fwd, brats_gn_act_bwd or
 * brats_absmax); both NULL -> the static power-of-two `xscale` is used (values beyond 448*xscale would become NaN) */
int brats_conv3d_f8_fwd(const void* x1, int c1, int pitch1, const float* amax1, const void* x2, int c2, int pitch2,
                        const float* amax2, float xscale, const void* packed_w, const float* bias, void* y, int ypitch,
                        void* y2, int y2pitch, int ysplit, float* stats, int dil, int N, int D, int H, int W, int cout,
                        brats_stream_t s);
int brats_conv3d_f8_fwd_f16(const void* x1, int c1, int pitch1, const float* amax1, const void* x2, int c2, int pitch2,
                        const float* amax2, float xscale, const void* packed_w, const float* bias, void* y, int ypitch,
                        void* y2, int y2pitch, int ysplit, float* stats, int dil, int N, int D, int H, int W, int cout,
                        brats_stream_t s);  /* the same on fp16 activations (job dtype BRATS_BF16 = 16-bit) */
/* out[0] = max |x| over `rows` voxels x C channels of an NDHWC tensor (channel pitch `pitch`) */
int brats_absmax(const void* x, int pitch, int dtype, size_t rows, int C, float* out, brats_stream_t s);
/* wgrad: dW[co][ci][tap] = sum_v dy[v][co] * x[v + off(tap)][ci]  (x = virtual concat as above).
 * `ws` = f32 workspace of brats_conv3d_wgrad_ws_bytes(); dw = [cout][c1+c2][k^3] f32, overwritten.
 * dbias (may be NULL): [cout] f32 = sum_v dy. */
size_t brats_conv3d_wgrad_ws_bytes(int dtype, int ksize, int N, int D, int H, int W, int c1, int c2, int cout);
/* Tuning / test knob: 1 = the all-taps kernel for 48x48 channel blocks of large bf16 dilation-1
 * layers (one 8-wave workgroup per CU owns all 27 taps of a tile), 0 = the tap-plane kernel
 * everywhere, -1 = default (1, or the BRATS_WGRAD_ALLTAPS environment variable).  Same values up
 * to f32 summation order; returns the previous setting. */
int brats_conv3d_set_wgrad_alltaps(int mode);
/* Tuning / test knob of the split-precision weight gradient (brats_conv3d_x3_wgrad): 1 = the fused kernel (f32 tiles staged
 * once, split into fp16 / bf16 pairs on their way into LDS, three MFMA products per staged tile: csrc/conv_wgrad_x3.hpp) for
 * dilation-1 layers with 48-channel blocks that have enough tiles, 2 = the fused kernel for ANY tile count (tests), 0 = the
 * round-4 form everywhere (one split pass to HBM + three launches of the 16-bit kernels), -1 = default (1, or the
 * BRATS_X3_WGRAD_FUSED environment variable).  Same values up to f32 summation order; returns the previous setting. */
int brats_conv3d_set_x3_wgrad_fused(int mode);
int brats_conv3d_wgrad(const void* x1, int c1, int pitch1, const void* x2, int c2, int pitch2,
                       const void* dy, int dypitch, float* ws, float* dw, float* dbias,
                       int dtype, int ksize, int dil, int N, int D, int H, int W, int cout,
                       brats_stream_t s);
/* Split-precision weight gradient with a scale for dY (dtype BRATS_X3_F16 / BRATS_X3_BF16; `amax_dy` = device scalar
 * max|dy|, may be NULL): dY * 2^k is split into 16-bit pairs, dw is multiplied by 2^-k in the slab reduction.  Workspace:
 * brats_conv3d_wgrad_ws_bytes(dtype, 3, ...). */
int brats_conv3d_x3_wgrad(const void* x1, int c1, int pitch1, const void* x2, int c2, int pitch2, const void* dy, int dypitch,
                          const float* amax_dy, float* ws, float* dw, float* dbias, int dtype, int dil, int N, int D, int H,
                          int W, int cout, brats_stream_t s);
/* e4m3 weight gradient (BASELINE.json configs[4]; model.conv_fp8 = "all"): X and dY (bf16 in HBM) are quantised to e4m3
 * while they are staged (power-of-two scales from their |max|, the device scalars amax*: as brats_conv3d_f8_fwd; one
 * scale for [x1 | x2]), the MMA is v_mfma_scale_f32_16x16x128_f8f6f4, dw is f32.  Built as the all-taps kernel only
 * (dilation 1; channel blocks 48 x 48 or 64 co x 32 ci; enough tiles): brats_conv3d_wgrad_f8_ws_bytes() returns the
 * workspace size, or 0 for a layer it is not built for (use brats_conv3d_wgrad there). */
size_t brats_conv3d_wgrad_f8_ws_bytes(int N, int D, int H, int W, int c1, int c2, int cout);
int brats_conv3d_wgrad_f8(const void* x1, int c1, int pitch1, const float* amax1, const void* x2, int c2, int pitch2,
                          const float* amax2, const void* dy, int dypitch, const float* amax_dy, float* ws, float* dw,
                          int N, int D, int H, int W, int cout, brats_stream_t s);
int brats_conv3d_wgrad_f8_f16(const void* x1, int c1, int pitch1, const float* amax1, const void* x2, int c2, int pitch2,
                          const float* amax2, const void* dy, int dypitch, const float* amax_dy, float* ws, float* dw,
                          int N, int D, int H, int W, int cout, brats_stream_t s);  /* the same on fp16 activations (job dtype BRATS_BF16 = 16-bit) */
/* Weight gradient in the "shifted-tap" form: ksize = 1 (ConvEvo / bridge / upconv / ASPP k1 convolutions,
 * networks/equiunet2021.py:212-222 -- a GEMM over the voxels) and ksize = 3 at ANY dilation >= 1 (the ASPP branches with
 * dilation 4 and 6, :121-189, whose halo does not fit LDS): one workgroup per (tap, channel block, voxel range), its X tile
 * is the dY tile's box shifted by the tap.  Same split-K slabs + fixed-order reduction as brats_conv3d_wgrad.
 * dw = [cout][cin][ksize^3] f32, overwritten; dbias (may be NULL) = sum_v dy. */
size_t brats_conv3d_wgrad_shift_ws_bytes(int dtype, int ksize, int N, int D, int H, int W, int cin, int cout);
int brats_conv3d_wgrad_shift(const void* x, int cin, int xpitch, const void* dy, int dypitch, float* ws, float* dw,
                             float* dbias, int dtype, int ksize, int dil, int N, int D, int H, int W, int cout,
                             brats_stream_t s);
/* the same in split precision (dtype BRATS_X3_F16 / BRATS_X3_BF16: f32 x and dy, three 16-bit MFMA products; amax_dy as in
 * brats_conv3d_x3_wgrad); brats_conv3d_wgrad_shift(_ws_bytes) accept the two codes as well (no scale) */
int brats_conv3d_x3_wgrad_shift(const void* x, int cin, int xpitch, const void* dy, int dypitch, const float* amax_dy,
                                float* ws, float* dw, float* dbias, int dtype, int ksize, int dil, int N, int D, int H, int W,
                                int cout, brats_stream_t s);

/* ---- GroupNorm(8) + activation (nn.GroupNorm networks/factory.py:179-182, get_act :195-200) ---
 * finalize: per-(n,channel) tile partials -> per-(n,group) mean / rstd (biased var, eps) and the
 * fused per-(n,channel) affine  z = act(y*scale + shift). */
/* f64 elements of the `chan_ws` workspace of brats_gn_finalize / brats_evonorm_finalize: [N][C][2] per-channel
 * totals (valid after the call) followed by the per-slice partial sums; no zeroing required. */
size_t brats_gn_ws_doubles(int N, int C);
int brats_gn_finalize(const float* stats, int tiles_per_sample, int N, int C, int groups,
                      double count_per_channel, float eps, const float* gamma, const float* beta,
                      float* mean_rstd /*[N][groups][2]*/, float* scale_shift /*[N][C][2]*/,
                      double* chan_ws /* f64 workspace of brats_gn_ws_doubles() elements */, brats_stream_t s);
/* amax (optional, device scalar that is ZERO before the call): receives max|z|, the scale source of the fp8
 * convolutions (brats_conv3d_f8_fwd) */
/* slope: the negative-side factor of BRATS_ACT_LEAKY; slope_dev (optional device scalar) overrides it -- nn.PReLU's
 * learnable weight (--act prelu, networks/factory.py:195-200) is read where it lives, without a host round trip */
int brats_affine_act_fwd(const void* y, int ypitch, const float* scale_shift, void* z, int zpitch,
                         int dtype, int act, float slope, const float* slope_dev, int N, int voxels, int C, float* amax,
                         brats_stream_t s);
/* The same pass for a layer that ends an encoder level (ConvBnRelu -> MaxPool3d(2, 2), equiunet2020.py:469-475): writes z
 * AND its 2x2x2 max pool (with_avg: MaxAvgPool's [max | mean], 2C channels) -- z is not read back by a pooling kernel.
 * Bit-identical to brats_affine_act_fwd + brats_maxpool2_fwd; relu / leakyrelu; amax receives max|z|. */
int brats_affine_act_pool_fwd(const void* y, int ypitch, const float* scale_shift, void* z, int zpitch, void* pooled,
                              int ppitch, unsigned char* argmax /* optional: [N][D/2][H/2][W/2][C] bytes, the window index
                              0..7 (d, h, w order) of torch's first arg-max: what brats_maxpool2_bwd_idx reads */,
                              int dtype, int act, float slope, const float* slope_dev, int N, int D, int H, int W, int C,
                              int with_avg, float* amax, brats_stream_t s);
/* backward of z = act(GN(y)): pass 1 reduces, per (n,channel), sum(u) and sum(u*xhat) with
 * u = dz * act'(.) into `red` (workspace of brats_gn_bwd_ws_floats() elements); pass 2 writes dy and finishes
 * dgamma/dbeta [C]. */
/* f32 elements of the `red` workspace of brats_gn_act_bwd: [N][C][2] totals followed by per-block partials (no zeroing
 * required; the partials are added in block order, so the result is bitwise reproducible) */
size_t brats_gn_bwd_ws_floats(int N, int C);
int brats_gn_act_bwd(const void* dz, int dzpitch, const void* y, int ypitch, const float* scale_shift,
                     const float* mean_rstd, const float* gamma, void* dy, int dypitch,
                     float* red /* workspace */, float* dgamma, float* dbeta,
                     int dtype, int act, float slope, const float* slope_dev, int N, int voxels, int C, int groups,
                     float* amax /* optional, zero before the call: receives max|dy| */, brats_stream_t s);
/* The same backward with pass 1 replaced by the tile sums brats_conv3d_fwd_bstats() left (sum u, sum u * y per tile and
 * channel): they are added in a fixed order in f64, sum u * xhat = rstd * (sum u*y - mean * sum u), then pass 2 as above.
 * 16-bit activations, relu / leakyrelu; `red`: the same workspace. */
int brats_gn_act_bwd_tiles(const float* tile_stats, int tiles_per_sample, const void* dz, int dzpitch, const void* y,
                           int ypitch, const float* scale_shift, const float* mean_rstd, const float* gamma, void* dy,
                           int dypitch, float* red /* workspace */, float* dgamma, float* dbeta, int dtype, int act,
                           float slope, int N, int voxels, int C, int groups,
                           float* amax /* optional, zero before the call: receives max|dy| */, brats_stream_t s);
/* The same backward for the layer whose output feeds ONLY a 1x1x1 head convolution with K = 3 logit planes -- the network's
 * last ConvBnRelu + outconv (networks/equiunet2020.py:488): dz[v][c] = sum_k dlogits[k][v] * hw[k][c] is computed inside both
 * passes from the 12 bytes of dlogits per voxel instead of being written (2 * C bytes per voxel) by brats_head_bwd and read
 * back twice, and the head's own gradients dhw [K][C] = sum_v dlogits[k][v] * z[v][c], dhb [K] = sum_v dlogits[k][v] come
 * out of pass 1 (z = act(GN(y)) is at hand there).  Replaces brats_head_bwd(scale 1) + brats_gn_act_bwd for that layer;
 * relu / leakyrelu only.  dlogits: f32 [N][K][voxels]; hws: brats_gn_bwd_head_ws_floats(N, C, K) floats. */
size_t brats_gn_bwd_head_ws_floats(int N, int C, int K);
int brats_gn_act_bwd_head(const float* dlogits, const float* hw, int K, const void* y, int ypitch, const float* scale_shift,
                          const float* mean_rstd, const float* gamma, void* dy, int dypitch, float* red /* as above */,
                          float* hws, float* dgamma, float* dbeta, float* dhw, float* dhb, int dtype, int act, float slope,
                          int N, int voxels, int C, int groups, float* amax, brats_stream_t s);
/* ... and for the layer that ends an encoder level (ConvBnRelu -> MaxPool3d(2, 2), its output also the skip connection,
 * equiunet2020.py:469-475): dz = dskip + maxpool-backward(dpool) is composed inside both passes from the skip gradient, the
 * pooled gradient and the arg-max bytes of brats_maxpool2_fwd / brats_affine_act_pool_fwd instead of being written by the
 * pooling backward and read back twice.  Replaces brats_maxpool2_bwd_idx + brats_gn_act_bwd; relu / leakyrelu. */
int brats_gn_act_bwd_pool(const void* dskip, int dskip_pitch, const void* dpool, int dpool_pitch, const unsigned char* argmax,
                          const void* y, int ypitch, const float* scale_shift, const float* mean_rstd, const float* gamma,
                          void* dy, int dypitch, float* red, float* dgamma, float* dbeta, int dtype, int act, float slope,
                          int N, int D, int H, int W, int C, int groups, float* amax, brats_stream_t s);
/* gradient of nn.PReLU's scalar slope: dslope[0] = sum dz * min(y*scale + shift, 0) over the whole tensor; ws = f32
 * workspace of brats_prelu_ws_floats(N) elements (block partials, added in block order) */
size_t brats_prelu_ws_floats(int N);
int brats_prelu_slope_grad(const void* dz, int dzpitch, const void* y, int ypitch, const float* scale_shift, float* ws,
                           float* dslope, int dtype, int N, int voxels, int C, brats_stream_t s);

/* ---- EvoNorm-S0 (EvoNorm3D networks/equiunet2021.py:55-118, group_std :48-52; groups = 8) ---------
 * z = x*sigmoid(x) * rstd_g * gamma_c + beta_c with the UNBIASED group variance.  `stats` are the
 * conv epilogue's tile partials; mean_rstd [N][groups][2].  chansum (may be NULL) [N][C] receives
 * sum_v z = the global-average-pool numerator of the following ResidualSELayer (:204-205). */
/* f32 elements of the reduction outputs `chansum` (vals = 1), `red` of brats_evonorm_bwd (vals = 3) and `out` of
 * brats_channel_dot (vals = 1): [N][C*vals] totals (valid after the call) followed by per-block partial sums that are
 * added in block order -- no zeroing required, no float atomics. */
size_t brats_chan_ws_floats(int N, int C, int vals);
int brats_evonorm_finalize(const float* stats, int tiles_per_sample, int N, int C, int groups,
                           double count_per_channel, float eps, float* mean_rstd, double* chan_ws,
                           brats_stream_t s);
int brats_evonorm_fwd(const void* x, int xpitch, const float* mean_rstd, const float* gamma,
                      const float* beta, void* z, int zpitch, float* chansum, int dtype, int N,
                      int voxels, int C, int groups,
                      float* amax /* optional, zero before the call: receives max|z| */, brats_stream_t s);
/* chan_sums = the f64 [N][C][2] per-channel sums brats_evonorm_finalize left in chan_ws; with it the
 * kernel also emits dconvbias[c] = sum_v dx (bias gradient of the producing conv) at no extra pass. */
int brats_evonorm_bwd(const void* dz, int dzpitch, const void* x, int xpitch, const float* mean_rstd,
                      const float* gamma, void* dx, int dxpitch, float* red /*[N][C][3]*/,
                      float* dgamma, float* dbeta, const double* chan_sums, float* dconvbias,
                      int dtype, int N, int voxels, int C, int groups,
                      float* amax /* optional, zero before the call: receives max|dx| */,
                      const float* gscale, const float* gadd /* optional [N][C]: dz is read as dz * gscale + gadd -- the
                      ResidualSELayer backward (out = z + z * gate) folded in instead of a separate channel_scale pass */,
                      brats_stream_t s);
/* EvoNorm backward whose first pass was taken by the producer of dz (the EvoNorm analogue of brats_gn_act_bwd_tiles): the
 * gradient dz of the first EvoNorm of a ConvEvoBlockCorrected (networks/equiunet2021.py:197-206) comes out of the second
 * convolution's input-gradient launch, run as brats_conv3d_fwd_bstats with by = z (that EvoNorm's stored output), act =
 * leakyrelu, slope 1 -- tile_stats [N][tiles][C][2] then holds sum dz and sum dz * z per tile and channel.  z is linear in
 * num(x) = x * sigmoid(x), so these give pass 2 everything pass 1 (brats_evonorm_bwd's first kernel: a read of dz and x and a
 * sigmoid per element) computed for it; the two sums that are not linear in z (for dgamma and dconvbias) are taken by pass 2
 * itself.  Same outputs as brats_evonorm_bwd (without gscale / gadd), 16-bit activations only.
 * red: brats_evonorm_bwd_tiles_ws_floats(N, C) floats. */
size_t brats_evonorm_bwd_tiles_ws_floats(int N, int C);
int brats_evonorm_bwd_tiles(const float* tile_stats, int tiles_per_sample, const void* dz, int dzpitch, const void* x,
                            int xpitch, const float* mean_rstd, const float* gamma, const float* beta, void* dx, int dxpitch,
                            float* red, float* dgamma, float* dbeta, const double* chan_sums, float* dconvbias, int dtype,
                            int N, int voxels, int C, int groups, float* amax, brats_stream_t s);
/* ---- squeeze-excite (MONAI ResidualSELayer(3, C, r = 2, relu, sigmoid), equiunet2021.py:204-205): per-(n,channel)
 * reductions over voxels, per-(n,channel) scale(+add) passes, and the gate itself (round 3: ONE launch forward, ONE launch
 * backward instead of ~20 ATen launches per block; up to 16 workgroups of 1024 threads, csrc/se.hip):
 *   gate1p[n][c] = 1 + sigmoid(b2 + W2 relu(b1 + W1 (chansum[n] * inv_vox)))      W1 [Ch][C], W2 [C][Ch] (nn.Linear layout)
 *   hidden[n][j] = the post-ReLU hidden vector (saved for the backward)
 * backward: dgate[n][c] = d loss / d gate (= brats_channel_dot(dout, z)) -> gadd[n][c] = (d loss / d gap) * inv_vox (the
 * `gadd` of brats_evonorm_bwd) and the four parameter gradients (samples added in order: bitwise reproducible).  N <= 8. */
int brats_se_fwd(const float* chansum, float inv_vox, const float* w1, const float* b1, const float* w2, const float* b2,
                 float* gate1p /*[N][C]*/, float* hidden /*[N][Ch]*/, int N, int C, int Ch, brats_stream_t s);
int brats_se_bwd(const float* dgate, const float* chansum, float inv_vox, const float* hidden, const float* gate1p,
                 const float* w1, const float* w2, float* gadd /*[N][C]*/, float* dw1, float* db1, float* dw2, float* db2,
                 int N, int C, int Ch, brats_stream_t s);
/* The EvoNorm backward of the layer a ResidualSELayer sits on (ConvEvoBlockCorrected, equiunet2021.py:192-209) with the SE
 * backward in the middle: pass 1 over (dout, x) collects five raw per-(n, channel) sums, from which brats_se_bwd's work is
 * done WITHOUT brats_channel_dot's pass (d loss / d gate = sum_v dout * z is linear in them), then pass 2 reads the
 * gradient as dout * gate1p + gadd.  Same results as brats_channel_dot + brats_se_bwd + brats_evonorm_bwd(gscale = gate1p,
 * gadd) up to f32 summation order.  ws: brats_chan_ws_floats(N, C, 5) + N * C * 3 floats; se_chansum = the `chansum` of
 * brats_evonorm_fwd ([N][C] = sum_v z); outputs as in brats_evonorm_bwd and brats_se_bwd. */
int brats_evonorm_se_bwd(const void* dout, int dopitch, const void* x, int xpitch, const float* mean_rstd, const float* gamma,
                         const float* beta, void* dx, int dxpitch, float* ws, float* dgamma, float* dbeta,
                         const double* chan_sums, float* dconvbias, const float* se_chansum, const float* hidden,
                         const float* gate1p, const float* w1, const float* w2, float* gadd, float* dw1, float* db1,
                         float* dw2, float* db2, int Ch,
                         const float* dlogits /* optional: f32 [N][K][voxels]; then dout may be NULL -- the block's output
                         feeds only the 1x1x1 output head, whose backward is folded in as in brats_gn_act_bwd_head (K = 3) */,
                         const float* hw /* [K][C] */, int K, float* hws /* brats_gn_bwd_head_ws_floats(N, C, K) floats */,
                         float* dhw /* [K][C] */, float* dhb /* [K] */,
                         int dtype, int N, int voxels, int C, int groups, float* amax, brats_stream_t s);
/* The same for a block that ends an encoder level (block -> MaxAvgPool, equiunet2021.py:261; its output is also the skip
 * connection): the block's output gradient dskip + pooling-backward(dpool) is composed inside both passes from the pieces and
 * the arg-max bytes of brats_maxpool2_fwd (with_avg: dpool holds [max | mean], 2C channels).  Replaces
 * brats_maxpool2_bwd_idx + brats_evonorm_se_bwd(dout): the block's output gradient is never written. */
int brats_evonorm_se_bwd_pool(const void* dskip, int dskip_pitch, const void* dpool, int dpool_pitch, const unsigned char* argmax,
                              int with_avg, int D, int H, int W, const void* x, int xpitch, const float* mean_rstd,
                              const float* gamma, const float* beta, void* dx, int dxpitch, float* ws, float* dgamma,
                              float* dbeta, const double* chan_sums, float* dconvbias, const float* se_chansum,
                              const float* hidden, const float* gate1p, const float* w1, const float* w2, float* gadd,
                              float* dw1, float* db1, float* dw2, float* db2, int Ch, int dtype, int N, int C, int groups,
                              float* amax, brats_stream_t s);
/* The forward counterpart: EvoNorm + ResidualSELayer without storing the EvoNorm output z.  Pass 1 reads x and sums
 * x*sigmoid(x) per (n, channel); sum_v z -- what the gate's global average pool reads -- is linear in those sums; pass 2
 * writes out = z * (1 + gate) directly (3 tensor passes instead of the 4 of brats_evonorm_fwd(chansum) + brats_se_fwd +
 * brats_channel_scale).  ws: brats_chan_ws_floats(N, C, 1) floats; chansum_out [N][C] = sum_v z, gate1p [N][C], hidden
 * [N][Ch]: kept for brats_evonorm_se_bwd.  out = NULL: pass 2 is skipped (the consumer recomputes it: brats_evonorm_head_fwd). */
int brats_evonorm_se_fwd(const void* x, int xpitch, const float* mean_rstd, const float* gamma, const float* beta,
                         const float* w1, const float* b1, const float* w2, const float* b2, void* out, int opitch,
                         float* ws, float* chansum_out, float* gate1p, float* hidden, int Ch, int dtype, int N, int voxels,
                         int C, int groups, float* amax /* optional, zero before the call: receives max|out| */,
                         brats_stream_t s);
int brats_channel_dot(const void* a, int apitch, const void* b /*may be NULL*/, int bpitch, float* out /*[N][C]*/,
                      int dtype, int N, int voxels, int C, brats_stream_t s);
int brats_channel_scale(const void* a, int apitch, const float* scale /*[N][C]*/, const float* add /*[N][C] or NULL*/,
                        void* dst, int dpitch, int dtype, int N, int voxels, int C,
                        float* amax /* optional, zero before the call: receives max|dst| */, brats_stream_t s);
/* ---- direct (gather) convolution of the ASPP head (SimpleASPPEVO, networks/equiunet2021.py:121-189: four parallel
 * nn.Conv3d 384 -> 96 with kernel 1 / 3, dilation 1 / 2 / 4 / 6, "same" padding, bias; torch.cat at :187; and their input
 * gradients).  The halo of dilation 4 / 6 does not fit LDS (it is larger than the 16^3 volume), but the volume lives in L2:
 * the MFMA B operand is gathered straight from global memory, taps outside the volume read zeros through the buffer
 * range check.  One launch = up to 4 jobs; a job's output [N*D*H*W][rows] (channel pitch ypitch, may be a channel slice
 * of a wider buffer) is bias + the sum of up to 4 terms, term = conv(x (cin channels at pitch xpitch), w) with
 * w = brats_dconv_pack_weights() output.  Forward of the ASPP head: 4 jobs x 1 term, each writing its slice of the concat
 * buffer; input gradient: 1 job x 4 terms (mode = BRATS_PACK_DGRAD weights), the four branch gradients summed in the
 * accumulators.  cin must be a multiple of 16 (bf16) / 8 (f32), rows a multiple of 4.  `jobs` is a HOST array. */
typedef struct { const void* x; const void* w; int xpitch, cin, ksize, dil; } brats_dconv_term;
typedef struct { brats_dconv_term term[4]; int nterms, rows; const float* bias; void* y; int ypitch, reserved; } brats_dconv_job;
size_t brats_dconv_packed_bytes(int dtype, int ksize, int kdim, int rows);
/* w: [Cout_w][Cin_w][k][k][k] f32 (torch layout); FWD: rows = Cout_w, K = the Cin_w slice [cin_off, cin_off + cin_cnt);
 * DGRAD: rows = the Cin_w slice, K = Cout_w, taps flipped (as brats_conv3d_pack_weights). */
int brats_dconv_pack_weights(const float* w, void* packed, int dtype, int mode, int ksize, int cout_w, int cin_w,
                             int cin_off, int cin_cnt, brats_stream_t s);
int brats_dconv_run(const brats_dconv_job* jobs, int njobs, int dtype, int N, int D, int H, int W, brats_stream_t s);

/* ---- pooling (nn.MaxPool3d(2,2) equiunet2020.py:433; MONAI MaxAvgPool equiunet2021.py:261) ---- */
int brats_maxpool2_fwd(const void* x, int xpitch, void* y, int ypitch,
                       unsigned char* argmax /* optional: [N][D/2][H/2][W/2][C] bytes = the window index 0..7 (d, h, w
                       order) of torch's first arg-max per pooled element, for brats_maxpool2_bwd_idx */,
                       int dtype, int N, int C, int D, int H, int W, int with_avg, brats_stream_t s);
/* dx = [dx_skip (may be NULL) +] maxpool^T(dy[..C]) [+ avgpool^T(dy[C..2C]) if with_avg] */
int brats_maxpool2_bwd(const void* x, int xpitch, const void* y, int ypitch, const void* dy, int dypitch,
                       const void* dx_skip, int dxskip_pitch, void* dx, int dxpitch, int dtype,
                       int N, int C, int D, int H, int W, int with_avg, brats_stream_t s);
/* The same backward from the arg-max bytes brats_affine_act_pool_fwd recorded: x is not read (8 window voxels per pooled
 * voxel: 403 MB at 2 x 48 x 128^3); bit-identical to brats_maxpool2_bwd. */
int brats_maxpool2_bwd_idx(const unsigned char* argmax, const void* dy, int dypitch, const void* dx_skip /*optional*/,
                           int dxskip_pitch, void* dx, int dxpitch, int dtype, int N, int C, int D, int H, int W,
                           int with_avg, brats_stream_t s);

/* ---- trilinear up-sampling, align_corners=True (nn.Upsample equiunet2020.py:439,446-458) ------
 * NDHWC -> NDHWC (scale 2, into a concat slice) ... */
int brats_upsample_fwd(const void* x, int xpitch, void* y, int ypitch, int dtype, int N, int C,
                       int D, int H, int W, int scale, brats_stream_t s);
/* adjoint; dx overwritten.  tmp: workspace of brats_upsample_bwd_ws_bytes() */
size_t brats_upsample_bwd_ws_bytes(int dtype, int N, int C, int D, int H, int W, int scale);
int brats_upsample_bwd(const void* dy, int dypitch, void* dx, int dxpitch, void* tmp, int dtype,
                       int N, int C, int D, int H, int W, int scale, brats_stream_t s);

/* ---- segmentation heads: 1x1x1 conv C -> K (K <= 4) + bias (conv1x1 equiunet2020.py:37-41,441)
 * followed by trilinear x`scale` up-sampling (deep heads :443-458); output NCDHW f32 logits. */
int brats_head_fwd(const void* x, int xpitch, const float* w /*[K][C]*/, const float* b, float* lowres /*[N][K][D][H][W] ws*/,
                   float* out /*[N][K][D*s][H*s][W*s]*/, int dtype, int N, int C, int K,
                   int D, int H, int W, int scale, brats_stream_t s);
/* The output head on the network's last layer WITHOUT storing that layer's activation: logits [N][K][voxels] =
 * conv1x1(act(y * scale + shift)) + b with y the raw convolution output and scale_shift [N][C][2] from brats_gn_finalize --
 * GroupNorm + relu / leakyrelu applied on load, rounded to the storage type exactly as the stored activation would be
 * (bit-identical logits).  Replaces brats_affine_act_fwd + brats_head_fwd(scale 1) for that layer; with
 * brats_gn_act_bwd_head the activation is not needed by the backward pass either. */
int brats_gn_head_fwd(const void* y, int ypitch, const float* scale_shift, int act, float slope, const float* w /*[K][C]*/,
                      const float* b, float* out, int dtype, int N, int C, int K, int voxels, brats_stream_t s);
/* ... and on EquiUnetASSPEvo's last block (EvoNorm + ResidualSELayer -> out_conv, equiunet2021.py:192-209): logits =
 * conv1x1(y * sigmoid(y) * scale + shift) + b, scale_shift [N][C][2] = { rstd_g * gamma_c * gate1p, beta_c * gate1p } with
 * gate1p from brats_evonorm_se_fwd called with out = NULL (statistics pass + gate only): the block's output is never stored. */
int brats_evonorm_head_fwd(const void* y, int ypitch, const float* scale_shift, const float* w /*[K][C]*/, const float* b,
                           float* out, int dtype, int N, int C, int K, int voxels, brats_stream_t s);
/* dout [N][K][Ds][Hs][Ws] f32 -> dx (NDHWC dtype, may be NULL), dw [K][C], db [K] (overwritten).
 * ws: f32 workspace of brats_head_bwd_ws_bytes() (up-sampling adjoint temporaries + per-block partial sums of dw / db,
 * added in a fixed order: no float atomics). */
size_t brats_head_bwd_ws_bytes(int N, int C, int K, int D, int H, int W, int scale);
int brats_head_bwd(const void* x, int xpitch, const float* w, const float* dout, float* ws,
                   void* dx, int dxpitch, float* dw, float* db, int dtype, int N, int C, int K,
                   int D, int H, int W, int scale, brats_stream_t s);

/* ---- sliding-window inference on the GPU (utils/inferers.py:103-162; the reference stitches on
 * the CPU, learning/engine.py:305-307).  NCDHW f32.  `windows` = device int32 [B][4] = (n, z, y, x)
 * window origins in the *padded* image; pad_* = leading constant padding (inferers.py:103-109). */
/* pad_mode = PytorchPadMode of inferers.py:34,109: 0 constant (cval), 1 reflect, 2 replicate, 3 circular */
int brats_sw_gather(const float* src, float* dst, const int* windows, int B, int C, int D, int H, int W,
                    int rd, int rh, int rw, int pad_z, int pad_y, int pad_x, float cval, int pad_mode, brats_stream_t s);
/* out[n][:, window] += importance * prob ; count[...] += importance (inferers.py:149-151) */
int brats_sw_accumulate(const float* prob, const float* importance, float* out, float* count, int K,
                        int Dp, int Hp, int Wp, int rd, int rh, int rw, int n, int z0, int y0, int x0,
                        brats_stream_t s);
/* the same for all B windows of one predictor batch (`windows` as in brats_sw_gather, prob = [B][K][rd][rh][rw]) in ONE launch:
 * a thread owns a voxel of the padded image [NB][K][Dp][Hp][Wp] and adds the windows covering it in window order --
 * bit-identical to B calls of brats_sw_accumulate in that order (overlapping windows cannot share a per-window launch) */
int brats_sw_accumulate_multi(const float* prob, const float* importance, float* out, float* count, const int* windows,
                              int B, int NB, int K, int Dp, int Hp, int Wp, int rd, int rh, int rw, brats_stream_t s);
/* dst = crop(out / count) (inferers.py:154-162) */
int brats_sw_finalize(const float* out, const float* count, float* dst, int NK, int Dp, int Hp, int Wp,
                      int D, int H, int W, int pad_z, int pad_y, int pad_x, brats_stream_t s);

/* ---- test-time augmentation (tta/transforms.py:16-74,149-173): any chain of OnAxes permute, flips
 * and rot90 is a signed permutation of the spatial axes.  dst axis a <- src axis p_a, reversed when
 * f_a.  mode 0: dst = v; 1: dst += v; 2: dst += sigmoid(v) (the on-GPU running sum of
 * learning/engine.py:239-249).  src dims (s0,s1,s2); `planes` = N*C. */
int brats_spatial_signed_perm(const float* src, float* dst, int planes, int s0, int s1, int s2,
                              int p0, int p1, int p2, int f0, int f1, int f2, int mode, brats_stream_t s);

/* ---- fused sigmoid-Dice / Jaccard passes (semantics of monai DiceLoss(sigmoid,
 * squared_pred, batch=True), src/definer.py:184-203).  NCDHW f32 logits / target [N][K][voxels].
 * stats: sums[k] = {sum t*p, sum p*p, sum t*t}; grad: dx = (coef[k][0]*t + coef[k][1]*2p) * p*(1-p). */
/* ws: f32 workspace of brats_dice_ws_floats(N, K) elements (per-block partial sums, added in a fixed order) */
size_t brats_dice_ws_floats(int N, int K);
int brats_dice_stats(const float* logits, const float* target, float* sums /*[K][3]*/, float* ws, int N, int K,
                     size_t voxels, brats_stream_t s);
int brats_dice_grad(const float* logits, const float* target, const float* coef /*[K][2]*/,
                    float* dlogits, int N, int K, size_t voxels, brats_stream_t s);

/* ---- post-forward chain of Engine.evaluate on the GPU.  NCDHW f32.
 * pad_crop: dst[p][z][y][x] = src[p][z-oz][y-oy][x-ox] inside the source box, `fill` outside; positive
 *   offsets = shape_to_divisible (utils/transforms.py:482-512, o = p_b), negative = shape_to_original
 *   (utils/transforms.py:515-533, o = -p_b).  planes = N*C.
 * post_threshold: seg = (prob*scale >= thresh) * any_c(img != 0): the mean over models x TTA passes
 *   (learning/engine.py:249, scale = 1/passes), AsDiscrete (src/definer.py:700-703) and
 *   remove_background_voxels (utils/transforms.py:536-550) in one pass; img may be NULL (no masking);
 *   labels (optional uint8 [N][voxels], K must be 3 = TC/WT/ET) = ConvertToBratsClassesBasedOnMultiChannel
 *   + ChangeLabel3To4 (utils/transforms.py:169-206).
 * overlap_counts: counts[nk] = {#(pred&target), #pred, #target} (uint64, exact) -- the sums behind the
 *   hard Dice / confusion metrics of utils/metrics.py:35-67. */
int brats_pad_crop(const float* src, float* dst, int planes, int sd, int sh, int sw, int dd, int dh,
                   int dw, int oz, int oy, int ox, float fill, brats_stream_t s);
int brats_post_threshold(const float* prob, const float* img, float* seg, uint8_t* labels, int N, int K,
                         int C, size_t voxels, float scale, float thresh, brats_stream_t s);
int brats_overlap_counts(const float* pred, const float* target, unsigned long long* counts, int NK,
                         size_t voxels, brats_stream_t s);

/* ---- multi-tensor Ranger2020 step (learning/optimizer.py:136-255: RAdam with the
 * N_sma threshold, gradient centralisation :11-20, lookahead :233-240).  All tensors f32, contiguous.
 * The host fills one record per parameter (device array `table`); per-step scalars that the reference
 * derives from state['step'] in Python (:198-214) are host-computed per tensor:
 *   neg_step = -step_size * lr ; flags bit0 = N_sma > threshold (adaptive branch), bit1 = lookahead
 *   step (step % k == 0); rowlen > 0 enables gradient centralisation over rows of `rowlen` elements
 *   (= numel / shape[0]) with the tensor's rows stored at row_means[row_base ...].
 * chunks: device int32 [nchunks][2] = (tensor index, chunk index), chunk = brats_ranger_chunk()
 * elements, tensor-major; rows: device int32 [nrows][2] = (tensor index, row) for every centralised row.
 * use_gcnorm (learning/optimizer.py:23-36,189-190; off by default): pass chunk_stats (f32 [nchunks][2] workspace) and
 * grad_std (f32 [ntensors] workspace) -- the centralised gradient of every tensor with more than two elements is
 * divided by its unbiased standard deviation + 1e-8; both NULL = off.  (normloss, :192-198, is not built: the
 * reference's own step() raises there.) */
typedef struct {
  void* param;
  const void* grad;
  void* exp_avg;
  void* exp_avg_sq;
  void* slow;
  long long numel;
  int rowlen;
  int row_base;
  float neg_step;
  float wd;
  int flags;
  int chunk_base; /* index of the tensor's first entry in `chunks` (its chunks are consecutive) */
} brats_ranger_tensor;
/* Graph-capturable stepping: `dyn` (device) holds the step counter and the two step-dependent scalars;
 * brats_ranger_advance increments the counter and recomputes them on the device (f64), and
 * brats_ranger_step with dyn != NULL reads them instead of the table's neg_step / flags -- a captured
 * hipGraph of (advance, step) replays without host-side changes.  dyn == NULL: host-computed scalars.
 * brats_ranger_advance(lr < 0) reads the learning rate from dyn->lr instead of the argument: the host rewrites those
 * 8 bytes between replays when an LR scheduler (the reference's --decay_type, learning/engine.py:151-155) changed it. */
typedef struct {
  int step;
  int flags;
  float neg_step;
  int reserved;
  double lr;
} brats_ranger_dyn;
int brats_ranger_chunk(void);
int brats_ranger_advance(brats_ranger_dyn* dyn, double lr, double beta1, double beta2, int k,
                         double nsma_threshold, brats_stream_t s);
int brats_ranger_step(const brats_ranger_tensor* table, int ntensors, const int* chunks, int nchunks,
                      const int* rows, int nrows, float* row_means, float* chunk_stats, float* grad_std,
                      const brats_ranger_dyn* dyn, float beta1, float beta2, float one_minus_beta1,
                      float one_minus_beta2, float eps, float alpha, brats_stream_t s);
/* The same step under torch.amp.GradScaler WITHOUT its host round trip (the reference's AMP loop, learning/engine.py:117-122, with
 * an optimizer that declares _step_supports_amp_scaling as torch's fused Adam does): grad_scale = device scalar holding the loss
 * scale the gradients still carry (a power of two; NULL = already unscaled) -- every gradient is read as g / grad_scale;
 * found_inf = device scalar, non-zero when a gradient overflowed (NULL = not checked) -- then every kernel of the step returns
 * at once and brats_ranger_advance_amp leaves the counter alone: the skipped step of GradScaler.step(), decided on the device. */
int brats_ranger_advance_amp(brats_ranger_dyn* dyn, double lr, double beta1, double beta2, int k,
                             double nsma_threshold, const float* found_inf, brats_stream_t s);
int brats_ranger_step_amp(const brats_ranger_tensor* table, int ntensors, const int* chunks, int nchunks,
                          const int* rows, int nrows, float* row_means, float* chunk_stats, float* grad_std,
                          const brats_ranger_dyn* dyn, float beta1, float beta2, float one_minus_beta1,
                          float one_minus_beta2, float eps, float alpha, const float* grad_scale,
                          const float* found_inf, brats_stream_t s);

/* ---- input pipeline on the GPU (the reference's CPU transform chain,
 * src/definer.py:449-467).  NCDHW f32.
 * crop_perm: dst axis a runs along source axis p_a over the crop box [c, c+e) of the source, reversed
 *   when f_a, then v*scale[plane] + shift[plane] (either may be NULL): RandSpatialCropd + RandRotate90d +
 *   RandFlipd + RandShiftIntensityd (/ scale) in one gather.  dst dims = (e[p0], e[p1], e[p2]).
 * label_to_channels: BraTS labels {0,1,2,4} -> 3 binary channels; order 0 = (TC, WT, ET) (MONAI
 *   ConvertToMultiChannelBasedOnBratsClassesd, src/definer.py:451), 1 = (WT, TC, ET)
 *   (utils/transforms.py:155-166).
 * zscore_normalize: NormalizeIntensity(nonzero, channel_wise=True[, remove_outliers -> clip])
 *   (utils/transforms.py:328-406): per plane, over the non-zero voxels, (x - mean) / population std
 *   (1 when 0), clip > 0 clamps to +-clip; zeros stay zero.  stats = f64 workspace [planes][3].
 * gamma_noise: MONAI AdjustContrast ((x - min)/(range + 1e-7))^gamma * range + min (gamma <= 0: skipped)
 *   followed by + noise (may be NULL): RandAdjustContrastd + RandGaussianNoised (:463-464). */
/* blur_axis: one axis of MONAI 0.6 GaussianSmooth / GaussianFilter (RandGaussianSmoothd, src/definer.py:464) on a tensor
 *   viewed as [outer][L][inner] f32: dst[o][i][v] = sum_k taps[k] * src[o][i + k - ntaps/2][v], zeros outside; taps = device
 *   f32 [ntaps], ntaps odd <= 63 (host-computed erf-integrated kernel, truncated at 4 sigma); out of place.
 * foreground_bbox: the box of MONAI CropForegroundd(source_key="img") (src/definer.py:452; select_fn x > 0 over any
 *   channel, margin 0) per sample of an NCDHW f32 batch: bbox[n] = {z0, y0, x0, z1, y1, x1}, ends exclusive;
 *   z0 = INT_MAX for a sample without foreground. */
int brats_blur_axis(const float* src, float* dst, size_t outer, int L, size_t inner, const float* taps, int ntaps,
                    brats_stream_t s);
int brats_foreground_bbox(const float* img, int N, int C, int D, int H, int W, int* bbox, brats_stream_t s);
int brats_crop_perm(const float* src, float* dst, int planes, int s0, int s1, int s2, int c0, int c1, int c2,
                    int e0, int e1, int e2, int p0, int p1, int p2, int f0, int f1, int f2,
                    const float* scale, const float* shift, brats_stream_t s);
int brats_label_to_channels(const float* label, float* out, int N, size_t voxels, int order, brats_stream_t s);
int brats_zscore_normalize(const float* x, float* y, double* stats, int planes, size_t voxels, int nonzero,
                           float clip, brats_stream_t s);
int brats_gamma_noise(const float* x, float* y, size_t total, float vmin, float vrange, float gamma,
                      const float* noise, brats_stream_t s);

/* ---- box calibration probes (bench.py's "box" record; no counterpart in the reference) -------------------------------
 * The boxes of one MI355X pool differ by several per cent on identical code, and the clock the chip holds under MFMA load
 * depends on power management: a roofline fraction against the nominal peak cannot tell a slower box from slower code.
 * The caller times these launches with HIP events on the launch stream.
 *   brats_probe_mfma: `blocks` workgroups of 4 waves, each wave issues `iters` x 8 independent v_mfma_f32_16x16x32_bf16 on
 *     pseudo-random operands -> FLOP = blocks * 4 * iters * 8 * 16384; `out` = blocks floats of scratch.  mode bit 0: every
 *     second operand value is zero (post-ReLU data); bit 1: 24 dependent VALU instructions between two groups of 8 MFMAs
 *     (~60 % matrix duty, the regime of the implicit-GEMM kernels) instead of the pure matrix loop (a power virus).
 *   brats_probe_stream: bf16 read + scale-shift-relu + write over `bytes` (16-byte aligned, multiple of 16) with the
 *     library's own streaming policy (non-temporal, 4 vectors in flight) -> 2 * bytes of HBM traffic. */
int brats_probe_mfma(float* out, int blocks, int iters, int mode, brats_stream_t s);
int brats_probe_stream(const void* src, void* dst, size_t bytes, brats_stream_t s);

#ifdef __cplusplus
}
#endif
#endif
