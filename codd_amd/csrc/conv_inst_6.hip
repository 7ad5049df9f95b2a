// Explicit instantiations of the convolution kernel, group 6 (see conv_kernel.h).
#include "conv_kernel.h"

CONV_GROUP_6(CONV_DEFINE)
