// Instantiations of the weight-gradient kernel: 3x3 stride-1 (plain and nearest-upsampled input).
#define MS_WGRAD_TU_A
#include "ms_wgrad.hip"
