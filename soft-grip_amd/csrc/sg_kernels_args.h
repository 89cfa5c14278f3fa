// argument block of the fused kernel (sg_kernels.hip; test builds only, -DSG_LEGACY_PIPELINES)
#pragma once
#include "sg_plan.h"

struct SgKArgs {
  const SgPlanHeader* H;
  const double* elem;     // SGE_NFIELD x nelem
  double *qpos, *qvel, *warm, *act, *ctrl;  // [nenv][nv] / [nenv][nu]
  const double* kenv;     // [nenv] stiffness scalar (reference manenv.py:104)
  const int* kmask_jnt;   // [nv]  joints that take kenv (manenv.py:105-106)
  const int* kmask_ten;   // [ntendon]  (manenv.py:107-108)
  const unsigned char* mask;  // reset mask or null
  double* sens;
  long long sens_stride;
  int *flags, *touch, *ncon, *nefc, *iters;
  int nenv, nsub, mode;   // mode 0: step; 1: reset (+ forward + nsub steps)
};

