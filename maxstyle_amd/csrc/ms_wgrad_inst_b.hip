// Instantiations of the weight-gradient kernel: 1x1, 3x3 stride-2, ConvTranspose2d 2x2 stride-2.
#define MS_WGRAD_TU_B
#include "ms_wgrad.hip"
