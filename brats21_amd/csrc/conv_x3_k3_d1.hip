// explicit instantiation unit: split-precision (f32 storage, 3 x 16-bit MFMA), 3x3x3, dilation 1 (see conv_igemm_x3.hpp)
#include <stdlib.h>
#include "twin_begin.hpp"
#include "conv_igemm_x3.hpp"
CONV_DEFINE_LAUNCH_X3(1)
#include "twin_end.hpp"
