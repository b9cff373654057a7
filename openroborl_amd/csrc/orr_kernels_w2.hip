// orr_kernels_w2.hip -- second translation unit of the env kernels: ONLY the two-waves-per-SIMD instantiation of the step kernel
// (orr_step_kernel<0, 2>) and its launcher, compiled WITHOUT the instruction-level-parallelism scheduler flag of the main unit
// (openroborl_amd/_lib.py: HIPCC_FLAGS_W2).  Why: see the top of orr_kernels.hip.
#define ORR_TU_STEP_W2 1
#include "orr_kernels.hip"
