// Explicit instantiations of the convolution kernel, group 0 (see conv_kernel.h).
#include "conv_kernel.h"

CONV_GROUP_0(CONV_DEFINE)
