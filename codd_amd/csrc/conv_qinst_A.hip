// Explicit instantiations of the quad-layout convolution kernel, group A (see conv_quad_kernel.h).
#include "conv_quad_kernel.h"

CONVQ_GROUP_A(CONVQ_DEFINE)
