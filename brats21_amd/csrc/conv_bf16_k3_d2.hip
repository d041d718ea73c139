// explicit instantiation unit: bf16, 3x3x3, dilation 2 (see conv_igemm.hpp)
#include "twin_begin.hpp"
#include "conv_igemm.hpp"
CONV_DEFINE_LAUNCH_BF16(3, 2)
#include "twin_end.hpp"
