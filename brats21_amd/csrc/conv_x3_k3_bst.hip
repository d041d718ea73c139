// explicit instantiation unit: the "backward statistics" forms of the split-precision 3x3x3 kernels (training in the parity
// mode; ConvParams::by / bss, conv_igemm_x3.hpp): the input gradient of a block's second convolution also leaves GroupNorm /
// EvoNorm backward's first-pass sums of the block's first unit per tile.  Everything conv_x3_bst_supported() refuses keeps the
// normalisation's own first pass (brats_conv3d_bstats_ok() tells the caller).
#include <stdlib.h>
#include "twin_begin.hpp"
#include "conv_igemm_x3.hpp"

int conv_x3_bst_launch(const ConvParams& p, int ck, int dil, hipStream_t st) {
  if (!conv_x3_bst_supported(ck, dil, p.rows16))
    BRATS_FAIL(BRATS_E_UNSUPPORTED, "conv3d_x3_fwd_bstats: chunk %d, dilation %d, %d output rows", ck, dil, p.rows16 * 16);
  if (ck == 16) {  // conv_x3_launch_ck's choice for Cout = 48 (mod 96)
    if (conv_x3_ty8_enabled() && (long)p.N * p.tz * p.ty * p.tx >= 2048) return conv_x3_launch_one<3, 16, 1, 3, true, 8, true>(p, st);
    return conv_x3_launch_one<3, 16, 1, 3, true, CONV_TY, true>(p, st);
  }
  return dil == 1 ? conv_x3_bst_launch_ck24<1>(p, st) : conv_x3_bst_launch_ck24<2>(p, st);
}
#include "twin_end.hpp"
