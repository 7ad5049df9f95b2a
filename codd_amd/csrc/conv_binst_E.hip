// Explicit instantiations of the split-bf16 convolution kernel, group E (see conv_bf16_kernel.h).
#include "conv_bf16_kernel.h"

CONVB_GROUP_E(CONVB_DEFINE)
