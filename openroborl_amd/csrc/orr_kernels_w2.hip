// orr_kernels_w2.hip -- second translation unit of the env kernels: ONLY the two-waves-per-SIMD instantiation of the step kernel
// (orr_step_kernel<0, 2>) and its launcher, compiled with its OWN scheduler strategy (iterative-maxocc since round 4; the main unit's
// iterative-ilp costs this one 8 %: openroborl_amd/_lib.py: HIPCC_FLAGS_W2).  Why: see the top of orr_kernels.hip.
#define ORR_TU_STEP_W2 1
// start parity of the hand-written Gauss-Seidel loops (orr_device.h): with two waves sharing the instruction fetch the other parity wins
// (8192 robots: 0.3310 ms with the one-wave unit's 0x1000, 0.3259 ms with 0; the one-wave unit the other way round: 0.2344 vs 0.2395)
#ifndef ORR_PARITY
#define ORR_PARITY 0x0000
#endif
#include "orr_kernels.hip"
