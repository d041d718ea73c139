// explicit instantiation unit: the "normalise + activate on load" forms of the 16-bit 3x3x3 kernels (inference only; see
// conv_pre_apply in conv_igemm.hpp) for the channel roles of the width-48 / 96 / ... networks: Cout a multiple of 48,
// 48-channel chunks (4x4x16 tile) or 24-channel chunks (4x8x16 tile, Cout = 48 mod 96), dilation 1.  Everything else keeps
// the two-pass path (brats_conv3d_pre_ok() tells the caller).
#include <stdlib.h>
#include "twin_begin.hpp"
#include "conv_igemm_vs8.hpp"

bool conv_pre_supported(int ck, int dil, int rows16) {
  if (rows16 % 3) return false;
  if (ck == 24) return dil == 1 && rows16 % 6 != 0;
  return ck == 48 && dil == 1;  // (dilation 2 -- the bottom block at 16^3 -- spills in its 96-cout form and moves 3 MB: not built)
}

template <int DIL>
static int conv_pre_launch_ck48(const ConvParams& p, hipStream_t st) {
  // conv_launch_ck's choice among the NF = 3 roles
  if (p.rows16 % 6 == 0) {
    if (conv_vsplit_enabled() && (long)p.N * p.tz * p.ty * p.tx * (p.rows16 / 6) < conv_small_grid_threshold())
      return conv_launch_nf3<bf16_t, 3, 48, DIL, true, true>(p, st);  // (the same 8-wave choice as the plain kernel: bit-identical)
    return conv_launch_one<bf16_t, 3, 48, DIL, 3, false, false, true>(p, st);
  }
  return conv_launch_nf3<bf16_t, 3, 48, DIL, true, true>(p, st);
}

int conv_pre_launch(const ConvParams& p, int ck, int dil, hipStream_t st) {
  if (!conv_pre_supported(ck, dil, p.rows16)) BRATS_FAIL(BRATS_E_UNSUPPORTED, "conv3d_fwd_pre: chunk %d, dilation %d, %d output rows", ck, dil, p.rows16 * 16);
  if (ck == 24) return conv_launch_vs8<24, 1, 3, true>(p, st);
  return conv_pre_launch_ck48<1>(p, st);
}
#include "twin_end.hpp"
