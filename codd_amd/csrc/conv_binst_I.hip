// Explicit instantiations of the split-bf16 convolution kernel, group I (see conv_bf16_kernel.h).
#include "conv_bf16_kernel.h"

CONVB_GROUP_I(CONVB_DEFINE)
