// orr_kernels_anchor.hip -- third translation unit of the env kernels: ONLY the friction-anchor instantiations of the step kernel
// (orr_step_kernel<0, 1, true> = env step, <1, 1, true> = debug physics; Bullet's cached toe contact points, orr_model::friction_anchor,
// ABI v5) and their launchers, compiled with the main unit's flags.  Its own unit so that the default kernels' code generation does not
// depend on this optional feature being compiled next to them (see launch_step_anchor in orr_kernels.hip).
#define ORR_TU_STEP_ANCHOR 1
#include "orr_kernels.hip"
