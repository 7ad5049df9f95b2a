// Explicit instantiations of the quad-layout convolution kernel, group B (see conv_quad_kernel.h).
#include "conv_quad_kernel.h"

CONVQ_GROUP_B(CONVQ_DEFINE)
