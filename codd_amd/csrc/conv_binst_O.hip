// Explicit instantiations of the split-bf16 convolution kernel, group O: TERMS = 3 only (see conv_bf16_kernel.h).
#include "conv_bf16_kernel.h"

CONVB_GROUP_O(CONVB_DEFINE3)
