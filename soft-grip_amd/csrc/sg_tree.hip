// sg_tree.hip -- the TREE pipeline's kernel: one env per wavefront, the whole call (all substeps) in one launch, the env's state and
// per-chain matrices in LDS for its duration (sg_tree.h is the code; tests/emu compiles the same source for the host).
//
// north_star's layout -- "one env per wavefront with per-env state staged in LDS" -- is exactly this kernel; the fast kernels
// (sg_phase.hip, sg_rows.hip) left it for the two-finger class because 8 dofs per env leave a wavefront's lanes idle.  With 65 chain dofs, 64
// boxes and 17 016 candidate pairs per env the lanes have work: the pair walk, the mass-matrix entries, the M^-1 columns, the slider
// rows and a contact's chain block are all lane-parallel.  Bounds: fp64 VALU issue and LDS latency; HBM traffic is
// the contact rows' J / W blocks (<= 128 x 1.9 KB per env, L2-resident) and the 272 KB pair table shared by all envs.
#include <hip/hip_runtime.h>

#ifndef SGT_X_MONO
#define SG_HD_HEAVY __host__ __device__ inline __attribute__((noinline))   // (sg_math.h: the narrowphase routines are CALLED here)
#endif
#include "sg_tree.h"

// CHD: the capacity the per-chain loops are unrolled over (registers, not memory): 24 = SGT_CHD, 20 for the four-finger gripper's
// 16 / 17-dof chains (padded stride 20), 8 for short chains
template <int CHD>
__global__ __launch_bounds__(64) void sg_tree_kernel(sgt::TreeArgs a) {
  extern __shared__ double sg_tree_lds[];
  const int env = blockIdx.x;
  if (env >= a.nenv) return;
  // the step's stages are called functions (sg_tree.h tree_stage); they find the launch arguments through this word
  if (threadIdx.x == 0) *(unsigned long long*)sg_tree_lds = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
  __syncthreads();
  sgt::tree_env<CHD>(a, env, sg_tree_lds);
}

// ---- launchers (declared in sg_tree.h) ----
hipError_t sg_tree_prepare() {
  hipError_t e = hipFuncSetAttribute((const void*)sg_tree_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)sg_tree_kernel<20>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)sg_tree_kernel<SGT_CHD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  return e;
}
int sg_tree_occupancy(int CS, size_t lds_bytes) {
  int n = 0;
  hipError_t e = CS == 8    ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sg_tree_kernel<8>, 64, lds_bytes)
                 : CS == 20 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sg_tree_kernel<20>, 64, lds_bytes)
                            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, sg_tree_kernel<SGT_CHD>, 64, lds_bytes);
  return e == hipSuccess ? n : -1;
}
// the instantiation whose capacity IS the model's padded chain stride CS (sg_plan.cpp pads to 8, 20 or SGT_CHD)
hipError_t sg_launch_tree(const sgt::TreeArgs& a, int CS, size_t lds_bytes, hipStream_t s) {
  if (CS == 8) hipLaunchKernelGGL(sg_tree_kernel<8>, dim3(a.nenv), dim3(64), lds_bytes, s, a);
  else if (CS == 20) hipLaunchKernelGGL(sg_tree_kernel<20>, dim3(a.nenv), dim3(64), lds_bytes, s, a);
  else hipLaunchKernelGGL(sg_tree_kernel<SGT_CHD>, dim3(a.nenv), dim3(64), lds_bytes, s, a);
  return hipGetLastError();
}
