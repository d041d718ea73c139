// Internal interface between the EvoNorm backward (norm.hip, built twice: bf16 and fp16) and the squeeze-excite backward
// (se.hip, f32 only, built once): brats_evonorm_se_bwd runs the SE backward BETWEEN the two passes of the EvoNorm backward,
// brats_evonorm_se_fwd the gate between a statistics pass and the EvoNorm pass that applies it.
#pragma once
#ifdef BRATS_FP16
}  // (declared outside the twin namespace: one definition serves both builds)
#endif
// The ResidualSELayer sits on an EvoNorm output z = num(x) * rstd_g * gamma_c + beta_c, num = x * sigmoid(x); with the raw
// per-(n, channel) sums over voxels of the block's output gradient `do`
//     raw5[n][c] = { sum do, sum do * num, sum do * num', sum num, sum num' }
// the SE backward gets  d loss / d gate = sum_v do * z = rstd * gamma * raw5[1] + beta * raw5[0]  without a pass of its own,
// and writes the three sums the second EvoNorm pass needs for the gradient it reads, dz = do * gate1p + gadd (linear):
//     red3[n][c] = { gate1p * raw5[0] + gadd * V, gate1p * raw5[1] + gadd * raw5[3], gate1p * raw5[2] + gadd * raw5[4] }
struct SeFold {
  const float* raw5 = nullptr;       // null: dgate is given, nothing folded
  const float* mean_rstd = nullptr;  // [N][groups][2]
  const float* gamma = nullptr;
  const float* beta = nullptr;
  float* red3 = nullptr;             // [N][C][3]
  int groups = 0;
  float voxels = 0.f;
};
// Forward: with numsum[n][c] = sum_v num(x) (brats_evonorm_se_fwd's first pass) the pooled EvoNorm output the gate reads is
//     chansum[n][c] = sum_v z = rstd_g * gamma_c * numsum + beta_c * V
// without z ever being stored: the gate is applied inside the EvoNorm pass that follows (out = z * gate1p).
struct SeFwdFold {
  const float* numsum = nullptr;     // null: chansum is given
  const float* mean_rstd = nullptr;
  const float* gamma = nullptr;
  const float* beta = nullptr;
  float* chansum_out = nullptr;      // [N][C]: the reconstructed sum_v z (saved for the backward)
  int groups = 0;
  float voxels = 0.f;
};
int brats_se_fwd_launch(const float* chansum, const SeFwdFold& fold, float inv_vox, const float* w1, const float* b1, const float* w2,
                        const float* b2, float* gate1p, float* hidden, int N, int C, int Ch, hipStream_t st);
int brats_se_bwd_launch(const float* dgate, const SeFold& fold, const float* chansum, float inv_vox, const float* hidden,
                        const float* gate1p, const float* w1, const float* w2, float* gadd, float* dw1, float* db1, float* dw2,
                        float* db2, int N, int C, int Ch, hipStream_t st);
#ifdef BRATS_FP16
namespace brats_f16 {
#endif
