// Explicit instantiations of the persistent quad-layout convolution kernel (see conv_quad_persist.h).
#include "conv_quad_persist.h"

CONVQP_ALL(CONVQP_DEFINE)
