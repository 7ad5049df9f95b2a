// Explicit instantiations of the convolution kernel, group 2 (see conv_kernel.h).
#include "conv_kernel.h"

CONV_GROUP_2(CONV_DEFINE)
