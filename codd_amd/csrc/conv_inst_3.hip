// Explicit instantiations of the convolution kernel, group 3 (see conv_kernel.h).
#include "conv_kernel.h"

CONV_GROUP_3(CONV_DEFINE)
