// explicit instantiation unit: f32, 1x1x1, dilation 1 (see conv_igemm.hpp)
#include "conv_igemm.hpp"
CONV_DEFINE_LAUNCH_F32(1, 1)
