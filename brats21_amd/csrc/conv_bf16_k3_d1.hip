// explicit instantiation unit: bf16, 3x3x3, dilation 1 (see conv_igemm.hpp)
#include <stdlib.h>
#include "twin_begin.hpp"
#include "conv_igemm_ld.hpp"
template <> int conv_launch<bf16_t, 3, 1>(const ConvParams& p, int ck, hipStream_t st) {
  // loader-wave kernel: brats_conv3d_chunk() hands out 16 for the Cout = 48 (mod 96) layers in mode 2 (>= 2 chunks: with a
  // single chunk there is nothing to prefetch)
  if (ck == 16 && conv_vs8_mode() == 2 && p.rows16 % 3 == 0 && p.rows16 % 6 != 0 && p.nchunks >= 2 && conv_vsplit_enabled())
    return conv_launch_ld<3>(p, st);
  if (ck == 24) {
    static int w3 = -1;  // experiment: three workgroups per CU (168-register build of the same kernel)
    if (w3 < 0) { const char* e = getenv("BRATS_CONV_VS8_W3"); w3 = e ? atoi(e) : 0; }
    if (w3) return conv_launch_vs8<24, 1, 3, true>(p, st);
    return conv_launch_vs8<24, 1, 3>(p, st);
  }  // brats_conv3d_chunk() hands out 24 only for the layers of that kernel
  // first layer (4 -> 8 padded input channels, K = 216: seven macro-steps per tile, all per-tile overhead): same kernel
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
