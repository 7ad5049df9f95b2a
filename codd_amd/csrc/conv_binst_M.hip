// Explicit instantiations of the split-bf16 convolution kernel, group M: TERMS = 3 only (see conv_bf16_kernel.h).
#include "conv_bf16_kernel.h"

CONVB_GROUP_M(CONVB_DEFINE3)
