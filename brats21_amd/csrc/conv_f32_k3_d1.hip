// explicit instantiation unit: f32, 3x3x3, dilation 1 (see conv_igemm.hpp)
#include "conv_igemm.hpp"
CONV_DEFINE_LAUNCH_F32(3, 1)
