// Instantiations of the weight-gradient kernel: 3x3 stride-1 with a materialised input (BatchNorm-backward prologue on the gradient only).
#define MS_WGRAD_TU_C
#include "ms_wgrad.hip"
