// fp32 instantiation of the fused stage kernels
#define MI_T float
#define MI_SUFFIX f32
#include "mi_ode_launch.inc"

