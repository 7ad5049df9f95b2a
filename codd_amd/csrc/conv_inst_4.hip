// Explicit instantiations of the convolution kernel, group 4 (see conv_kernel.h).
#include "conv_kernel.h"

CONV_GROUP_4(CONV_DEFINE)
