// explicit instantiation unit: the "backward statistics" forms of the 16-bit 3x3x3 kernels (training; ConvParams::by / bss in
// conv_igemm.hpp): the input gradient of a block's second convolution also leaves GroupNorm backward's first-pass sums of
// the block's first unit per tile -- for the channel roles of the width-48 / 96 / ... networks: rows (= the first unit's
// channels) a multiple of 48, 48-channel chunks (4x4x16 tile, dilation 1 / 2) or 24-channel chunks (4x8x16 tile, 48 mod 96
// rows, dilation 1).  Everything else keeps brats_gn_act_bwd's own first pass (brats_conv3d_bstats_ok() tells the caller).
#include <stdlib.h>
#include "twin_begin.hpp"
#include "conv_igemm_vs8.hpp"

bool conv_bst_supported(int ck, int dil, int rows16) {
  if (rows16 % 3 || (dil != 1 && dil != 2)) return false;
  if (ck == 24) return dil == 1 && rows16 % 6 != 0;
  return ck == 48;
}

template <int DIL>
static int conv_bst_launch_ck48(const ConvParams& p, hipStream_t st) {
  // conv_launch_ck's choice among the NF = 3 roles
  if (p.rows16 % 6 == 0 && !(conv_vsplit_enabled() && (long)p.N * p.tz * p.ty * p.tx * (p.rows16 / 6) < conv_small_grid_threshold()))
    return conv_launch_one<bf16_t, 3, 48, DIL, 3, false, false, false, true>(p, st);
  return conv_launch_nf3<bf16_t, 3, 48, DIL, true, false, true>(p, st);
}

int conv_bst_launch(const ConvParams& p, int ck, int dil, hipStream_t st) {
  if (!conv_bst_supported(ck, dil, p.rows16)) BRATS_FAIL(BRATS_E_UNSUPPORTED, "conv3d_fwd_bstats: chunk %d, dilation %d, %d output rows", ck, dil, p.rows16 * 16);
  if (ck == 24) return conv_launch_vs8<24, 1, 3, false, true>(p, st);
  return dil == 1 ? conv_bst_launch_ck48<1>(p, st) : conv_bst_launch_ck48<2>(p, st);
}
#include "twin_end.hpp"
