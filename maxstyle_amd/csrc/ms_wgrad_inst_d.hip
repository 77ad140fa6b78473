// Instantiations of the weight-gradient kernel: 3x3 stride-1 without prologues.
#define MS_WGRAD_TU_D
#include "ms_wgrad.hip"
