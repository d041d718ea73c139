// explicit instantiation unit: fp8 (e4m3) 3x3x3, dilation 1 (see conv_igemm_f8.hpp)
#include "twin_begin.hpp"
#include "conv_igemm_f8.hpp"
CONV_F8_DEFINE_LAUNCH(1)
#include "twin_end.hpp"
