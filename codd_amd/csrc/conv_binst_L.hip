// Explicit instantiations of the split-bf16 convolution kernel, group L (see conv_bf16_kernel.h).
#include "conv_bf16_kernel.h"

CONVB_GROUP_L(CONVB_DEFINE)
