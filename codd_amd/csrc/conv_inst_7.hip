// Explicit instantiations of the convolution kernel, group 7 (see conv_kernel.h).
#include "conv_kernel.h"

CONV_GROUP_7(CONV_DEFINE)
