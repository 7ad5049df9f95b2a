// Explicit instantiations of the multi-job quad-layout convolution kernel (see conv_quad_kernel.h).
#include "conv_quad_kernel.h"

CONVQ_MULTI(CONVQM_DEFINE)
