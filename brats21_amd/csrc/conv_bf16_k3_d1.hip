// explicit instantiation unit: bf16, 3x3x3, dilation 1 (see conv_igemm.hpp)
#include <stdlib.h>
#include "twin_begin.hpp"
#include "conv_igemm_vs8.hpp"
#include "conv_igemm_first.hpp"
template <> int conv_launch<bf16_t, 3, 1>(const ConvParams& p, int ck, hipStream_t st) {
  if (ck == 24) return conv_launch_vs8<24, 1, 3>(p, st);  // brats_conv3d_chunk() hands out 24 only for the layers of that kernel
  // first layer (4 -> 8 padded input channels, K = 216: seven macro-steps per tile): the persistent, store-coalescing kernel
  // of conv_igemm_first.hpp (round 6) where the layer has its shape; the 4x8x16-tile kernel otherwise
  if (conv_first_ok(p, ck)) return conv_launch_first(p, st);
  if (ck == 8 && p.nchunks == 1 && p.rows16 % 3 == 0 && p.rows16 % 6 != 0 && conv_vsplit_enabled() && conv_vs8_mode() &&
      (long)p.N * p.tz * p.ty * p.tx >= 2048)
    return conv_launch_vs8<8, 1, 3>(p, st);
  switch (ck) {
    case 48: return conv_launch_ck<bf16_t, 3, 48, 1>(p, st);
    case 32: return conv_launch_ck<bf16_t, 3, 32, 1>(p, st);
    case 16: return conv_launch_ck<bf16_t, 3, 16, 1>(p, st);
    case 8: return conv_launch_ck<bf16_t, 3, 8, 1>(p, st);
  }
  BRATS_FAIL(BRATS_E_UNSUPPORTED, "conv bf16: unsupported channel chunk %d", ck);
}
#include "twin_end.hpp"
