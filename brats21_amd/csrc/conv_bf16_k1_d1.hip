// explicit instantiation unit: bf16, 1x1x1, dilation 1 (see conv_igemm.hpp)
#include "twin_begin.hpp"
#include "conv_igemm.hpp"
CONV_DEFINE_LAUNCH_BF16(1, 1)
#include "twin_end.hpp"
