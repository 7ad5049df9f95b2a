// Explicit instantiations of the convolution kernel, group 5 (see conv_kernel.h).
#include "conv_kernel.h"

CONV_GROUP_5(CONV_DEFINE)
