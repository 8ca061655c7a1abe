// The bls12-381 G1 instantiations of the MSM kernels (msm.hip) as a translation unit of their own, so that they can be built
// with another instruction-scheduling strategy than the rest (Makefile: -mllvm -amdgpu-sched-strategy=max-ilp; see the head
// of msm.hip for the measurement).
#define NCG_MSM_TU_G1 1
#include "msm.hip"
