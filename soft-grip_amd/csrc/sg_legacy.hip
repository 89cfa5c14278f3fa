// sg_legacy.hip -- TEST BUILDS ONLY (-DSG_LEGACY_PIPELINES, `python soft-grip_amd/build_native.py --legacy` -> libsoftgrip_legacy.so):
// r01's two earlier pipelines, kept as independent cross-checks of the rows pipeline on the fix-rows-only models
// (tests/test_gpu_parity.py builds and loads this library itself; the product library does not contain them):
//   fused  sg_step_kernel (sg_kernels.hip): one wavefront per env, the whole call in one launch;
//   split  sg_chain_kernel + sg_phase_kernel (sg_phase.hip, record layout) + sg_pgs_kernel below: 8 lanes per env, 8 envs per
//          wavefront, each finger stream on its own lane.
#include "sg_kernels.hip"
#include "sg_work.h"

__global__ __launch_bounds__(64) void sg_pgs_kernel(SgPgsArgs a) {
  extern __shared__ double lds[];  // [8 envs][4 arrays][N] + invm[N] + coef[N] + limits
  const int lane = threadIdx.x, le = lane / SG_G, g = lane % SG_G;
  const int env = blockIdx.x * SG_EPW + le;
  const SG_CONSTAS SgPlanHeader& H = *(const SG_CONSTAS SgPlanHeader*)a.H;
  const int N = H.nelem;
  const size_t S = 2 * (size_t)a.nenv;
  const int nwb = (a.nenv + SG_EPW - 1) / SG_EPW;
  const SgWork& W = a.w;
  const double mu[2] = {H.con_mu[0], H.con_mu[1]}, pgs_scale = H.pgs_scale, tolerance = H.tolerance;
  const int max_iter = H.iterations;
  const bool valid = env < a.nenv && W.pending[env] != 0;
  if (!__ballot(valid)) return;
  double* Las = lds + (size_t)le * 4 * N;
  double *Lf = Las + N, *Lb = Lf + N, *LR = Lb + N;
  double* Linvm = lds + (size_t)SG_EPW * 4 * N;
  double* Lcoef = Linvm + N;
  double* Llim = Lcoef + N + (size_t)(le * 2) * 4 * SG_MAXLIM;  // per stream: sign, R, b, f x 8
  for (int j = lane; j < N; j += 64) { Linvm[j] = 1.0 / (a.elem[(size_t)SGE_MASS * N + j] + a.elem[(size_t)SGE_ARMATURE * N + j]); Lcoef[j] = a.elem[(size_t)SGE_COEF * N + j]; }
  if (valid)
    for (int j = g; j < N; j += SG_G) {
      size_t o = (size_t)env * N + j;
      Las[j] = W.as[o]; Lf[j] = W.eqf[o]; Lb[j] = W.eqb[o]; LR[j] = W.eqR[o];
    }
  const bool is_stream = valid && g < 2;
  const size_t st = 2 * (size_t)(env < a.nenv ? env : 0) + (g & 1);
  int ns = 0, lim_active = 0, shared = 0;
  double Minv[16], aF[SG_CD] = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 16; i++) Minv[i] = 0;
  double tb = 0, tR = 1, tA = 1, tf = 0;
  if (valid) {
    tb = W.envh[(size_t)0 * a.nenv + env]; tR = W.envh[(size_t)1 * a.nenv + env];
    tA = W.envh[(size_t)2 * a.nenv + env]; tf = W.envh[(size_t)3 * a.nenv + env];
    shared = W.shared[env];
  }
  double* mylim = Llim + (size_t)(g & 1) * 4 * SG_MAXLIM;
  if (is_stream) {
    ns = W.ns[st];
    lim_active = W.lim_active[st];
#pragma unroll
    for (int i = 0; i < 16; i++) Minv[i] = W.sMinv[(size_t)i * S + st];
#pragma unroll
    for (int d = 0; d < SG_CD; d++) aF[d] = W.saF[(size_t)d * S + st];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int k = 0; k < SG_MAXLIM; k++) mylim[q * SG_MAXLIM + k] = W.lim[((size_t)q * SG_MAXLIM + k) * S + st];
  }
  __syncthreads();
  int nsmax = ns;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(nsmax, o); nsmax = t > nsmax ? t : nsmax; }
  const unsigned long long any_lim = __ballot(lim_active != 0);

  bool running = valid;
  int iters = 0;
  // my record column inside the wave's block; idle lanes read their env's stream (same address as its stream lane)
  // and store to the dummy block
  double* const rec0 = W.crec + SG_REC_INDEX(0, blockIdx.x, 0, 2 * le + (g & 1), nwb);
  double* const rec0_store = (valid && g < 2) ? rec0 : W.crec + SG_REC_INDEX(0, nwb, 0, lane % SG_SPW, nwb);
  const size_t slot_stride = (size_t)(nwb + 1) * SG_RF * SG_SPW;

  for (int it = 0; it < max_iter; it++) {
    if (!__ballot(running)) break;
    double imp_acc = 0, tJap = 0;
    if (running) {
      for (int j = g; j < N; j += SG_G) {
        double ae = Las[j], f = Lf[j], old = f, im = Linvm[j];
        double Rr = LR[j];
        imp_acc -= scalar_update(f, Lb[j], ae, Rr, im + Rr, false);
        ae += im * (f - old);
        Lf[j] = f; Las[j] = ae;
        tJap += Lcoef[j] * ae;
      }
    }
    {  // tendon row: sum over the env's 8 lanes
      double Ja = tJap;
#pragma unroll
      for (int o = 1; o < SG_G; o <<= 1) Ja += __shfl_xor(Ja, o);
      if (running) {
        double old = tf, tfn = tf;
        double ch = scalar_update(tfn, tb, Ja, tR, tA, false);
        if (g == 0) imp_acc -= ch;
        tf = tfn;
        double dft = tf - old;
        for (int j = g; j < N; j += SG_G) Las[j] += Linvm[j] * Lcoef[j] * dft;
      }
    }
    __syncthreads();
    // limits + contacts.  pass 0: stream 0 everywhere and stream 1 where the streams share no slider; pass 1: the rest
    for (int pass = 0; pass < 2; pass++) {
      const bool mine = is_stream && running && ((g == 0 || !shared) ? pass == 0 : pass == 1);
      if (!__ballot(mine)) continue;
      if (any_lim) {
#pragma unroll
        for (int k = 0; k < SG_MAXLIM; k++) {
          if (mine && (lim_active >> k & 1)) {
            const int d = k / 2;
            double f = mylim[3 * SG_MAXLIM + k], old = f, sg = mylim[k], Rr = mylim[SG_MAXLIM + k];
            imp_acc -= scalar_update(f, mylim[2 * SG_MAXLIM + k], sg * aF[d], Rr, Minv[5 * d] + Rr, true);
            mylim[3 * SG_MAXLIM + k] = f;
            double df = sg * (f - old);
#pragma unroll
            for (int q = 0; q < SG_CD; q++) aF[q] += Minv[4 * q + d] * df;
          }
        }
      }
      // Software pipeline: the record of contact i+1 is requested before contact i is updated, so the L2 / Infinity
      // Cache latency overlaps the update arithmetic.  Every lane issues the same loads and stores unconditionally
      // (idle lanes read their env's stream -- same addresses as the active lane -- and write to the dummy block), so
      // the compiler can count vmcnt and only waits for the previous batch of loads.
      const int nsl = mine ? ns : 0;
      auto load_rec = [&](Contact& c, int i) {
        const double* rec = rec0 + (size_t)i * slot_stride;
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int d = 0; d < SG_CD; d++) c.Jf[r][d] = rec[(4 * r + d) * SG_SPW];
#pragma unroll
        for (int r = 0; r < 3; r++) c.Js[r] = rec[(12 + r) * SG_SPW];
#pragma unroll
        for (int q = 0; q < 6; q++) c.A[q] = rec[(15 + q) * SG_SPW];
#pragma unroll
        for (int r = 0; r < 3; r++) c.b[r] = rec[(21 + r) * SG_SPW];
        c.R = rec[24 * SG_SPW];
        c.invm = rec[25 * SG_SPW];
#pragma unroll
        for (int r = 0; r < 3; r++) c.f[r] = rec[(26 + r) * SG_SPW];
        c.sl = ((const int*)(rec + 29 * SG_SPW))[0];
      };
      auto update_rec = [&](Contact& c, int i) {
        if (i < nsl) {
          double as_ = c.sl >= 0 ? Las[c.sl] : 0.0, df[3];
          imp_acc -= contact_update(c, aF, as_, mu, df);
          if (c.sl >= 0) Las[c.sl] = as_ + c.invm * (c.Js[0] * df[0] + c.Js[1] * df[1] + c.Js[2] * df[2]);
          double gg[SG_CD];
#pragma unroll
          for (int d = 0; d < SG_CD; d++) gg[d] = c.Jf[0][d] * df[0] + c.Jf[1][d] * df[1] + c.Jf[2][d] * df[2];
#pragma unroll
          for (int q = 0; q < SG_CD; q++) aF[q] += (Minv[4 * q] * gg[0] + Minv[4 * q + 1] * gg[1]) + (Minv[4 * q + 2] * gg[2] + Minv[4 * q + 3] * gg[3]);
        }
        double* recs = rec0_store + (size_t)i * slot_stride;
#pragma unroll
        for (int r = 0; r < 3; r++) recs[(26 + r) * SG_SPW] = c.f[r];
      };
      // two contacts per trip with the buffers swapping roles (no register copies); requesting records two updates
      // ahead (three buffers) measured no faster
      Contact ca, cb;
      load_rec(ca, 0);
      for (int i = 0; i < nsmax; i += 2) {
        load_rec(cb, i + 1 < SG_CAP ? i + 1 : i);
        update_rec(ca, i);
        load_rec(ca, i + 2 < SG_CAP ? i + 2 : i);
        update_rec(cb, i + 1);
      }
      __syncthreads();
    }
    double imp = imp_acc;
#pragma unroll
    for (int o = 1; o < SG_G; o <<= 1) imp += __shfl_xor(imp, o);
    if (running) {
      iters = it + 1;
      if (imp * pgs_scale < tolerance) running = false;
    }
  }
  __syncthreads();
  // ---- fresh M^-1 J' f from the final forces (same as the fused kernel's recompute) ----
  if (valid)
    for (int j = g; j < N; j += SG_G) Las[j] = Linvm[j] * (Lf[j] + Lcoef[j] * tf);
  __syncthreads();
  double gF[SG_CD] = {0, 0, 0, 0};
  for (int pass = 0; pass < 2; pass++) {  // stream 0 then stream 1: deterministic when they share a slider
    if (is_stream && g == pass) {
#pragma unroll
      for (int k = 0; k < SG_MAXLIM; k++)
        if (lim_active >> k & 1) gF[k / 2] += mylim[k] * mylim[3 * SG_MAXLIM + k];
      for (int i = 0; i < ns; i++) {
        const double* rec = rec0 + (size_t)i * slot_stride;
        double f0 = rec[26 * SG_SPW], f1 = rec[27 * SG_SPW], f2 = rec[28 * SG_SPW];
        int sl = ((const int*)(rec + 29 * SG_SPW))[0];
        if (sl >= 0) Las[sl] += rec[25 * SG_SPW] * (rec[12 * SG_SPW] * f0 + rec[13 * SG_SPW] * f1 + rec[14 * SG_SPW] * f2);
#pragma unroll
        for (int d = 0; d < SG_CD; d++) gF[d] += rec[d * SG_SPW] * f0 + rec[(4 + d) * SG_SPW] * f1 + rec[(8 + d) * SG_SPW] * f2;
      }
    }
    __syncthreads();
  }
  if (is_stream) {
#pragma unroll
    for (int q = 0; q < SG_CD; q++) {
      double s = 0;
#pragma unroll
      for (int d = 0; d < SG_CD; d++) s += Minv[4 * q + d] * gF[d];
      W.saF[(size_t)q * S + st] = s;
    }
  }
  if (valid) {
    for (int j = g; j < N; j += SG_G) W.as[(size_t)env * N + j] = Las[j];
    if (g == 0) W.iters[env] = iters;
  }
}


hipError_t sg_legacy_prepare() { return hipFuncSetAttribute((const void*)sg_pgs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }
hipError_t sg_launch_pgs_split(const SgPgsArgs& a, int nenv, size_t lds_bytes, hipStream_t s) {
  hipLaunchKernelGGL(sg_pgs_kernel, dim3((nenv + SG_EPW - 1) / SG_EPW), dim3(64), lds_bytes, s, a);
  return hipGetLastError();
}
hipError_t sg_launch_fused(const SgKArgs& a, int rounds, int nenv, hipStream_t s) {
  dim3 grid(nenv), block(64);
  switch (rounds) {
    case 1: hipLaunchKernelGGL((sg_step_kernel<1, 2>), grid, block, 0, s, a); break;
    case 2: hipLaunchKernelGGL((sg_step_kernel<2, 2>), grid, block, 0, s, a); break;
    case 3: hipLaunchKernelGGL((sg_step_kernel<3, 2>), grid, block, 0, s, a); break;
    default: hipLaunchKernelGGL((sg_step_kernel<4, 2>), grid, block, 0, s, a); break;
  }
  return hipGetLastError();
}
