// Explicit instantiations of the split-bf16 convolution kernel, group H (see conv_bf16_kernel.h).
#include "conv_bf16_kernel.h"

CONVB_GROUP_H(CONVB_DEFINE)
