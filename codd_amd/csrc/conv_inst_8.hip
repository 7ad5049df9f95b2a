// Explicit instantiations of the convolution kernel, group 8 (see conv_kernel.h).
#include "conv_kernel.h"

CONV_GROUP_8(CONV_DEFINE)
