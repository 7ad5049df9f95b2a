// Explicit instantiations of the quad-layout convolution kernel, group C (see conv_quad_kernel.h).
#include "conv_quad_kernel.h"

CONVQ_GROUP_C(CONVQ_DEFINE)
