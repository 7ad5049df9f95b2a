// Explicit instantiations of the convolution kernel, group 9 (see conv_kernel.h).
#include "conv_kernel.h"

CONV_GROUP_9(CONV_DEFINE)
