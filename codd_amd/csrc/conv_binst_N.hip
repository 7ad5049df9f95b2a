// Explicit instantiations of the split-bf16 convolution kernel, group N: TERMS = 3 only (see conv_bf16_kernel.h).
#include "conv_bf16_kernel.h"

CONVB_GROUP_N(CONVB_DEFINE3)
