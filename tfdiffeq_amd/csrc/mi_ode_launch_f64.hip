// fp64 instantiation of the fused stage kernels (config 4 headline path)
#define MI_T double
#define MI_SUFFIX f64
#include "mi_ode_launch.inc"
