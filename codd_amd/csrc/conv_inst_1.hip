// Explicit instantiations of the convolution kernel, group 1 (see conv_kernel.h).
#include "conv_kernel.h"

CONV_GROUP_1(CONV_DEFINE)
