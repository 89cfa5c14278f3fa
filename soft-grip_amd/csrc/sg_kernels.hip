// sg_kernels.hip -- gfx950 step kernel of the batched soft-gripper simulator.
//
// One wavefront (64 lanes) integrates one env; a launch advances every env by n_substeps
// mj_step's (reference environment/manenv.py:48-49), keeping the whole state on chip
// between substeps.  Work split inside the wave:
//   * lanes 0 / 32 own finger chain 0 / 1: kinematics, 4x4 mass matrix, bias, tendon,
//     actuator, joint limits, accelerometer + gyro (sg_math.h chain_*); their state is
//     parked in LDS so that it does not occupy registers in the other 62 lanes;
//   * every lane owns R elements (e = r*64 + lane): slider dynamics, the joint-fix rows,
//     capsule-vs-finger-box narrowphase;
//   * contacts are compacted in MuJoCo's order into two streams (one per chain); the i-th
//     contact of stream s lives in the registers of lane 32*s + (i & 31), slot i >> 5;
//   * PGS runs its rows in MuJoCo's order.  Rows that do not interact are updated
//     together: all joint-fix rows (disjoint sliders) at once, then the tendon row via a
//     wave reduction, then per sweep position i the i-th contact of both streams (the
//     chains share no dof; if they share a slider the streams are run one after the other).
// HBM traffic per launch and env: state in, state + sensors out; model
// constants come from a ~20 KB plan that stays in L2/L1.
#include <hip/hip_runtime.h>

#include "../../include/softgrip.h"
#include "sg_math.h"

using namespace sgm;

#include "sg_kernels_args.h"

struct StageRec {
  double dist, pos[3], n[3];
  int sl, box;
};

struct ChainLds {  // everything only the chain lane needs between phases
  double q[SG_CD], v[SG_CD], w[SG_CD], k[SG_CD], act, ctrl, kten, act_dot;
  double qfrc_smooth[SG_CD], qacc_smooth[SG_CD], M[16], Minv[16];
  int lim_active, pad;
  double lim_sign[SG_MAXLIM], lim_R[SG_MAXLIM], lim_b[SG_MAXLIM], lim_f[SG_MAXLIM];
};

template <int R, int CPL>
struct Smem {
  ChainKin K[SG_MAXCH];
  ChainLds cs[SG_MAXCH];
  double boxp[SG_MAXCH * SG_CG][3], boxm[SG_MAXCH * SG_CG][9];
  double ve[R * 64], asme[R * 64], we[R * 64], as[R * 64];
  StageRec stage[SG_MAXCH][32 * CPL];
  int owner[R * 64];  // stream that touched slider e in this step (-1 none, 2 both)
};

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
  return x;
}
__device__ __forceinline__ int lanes_below(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
}
// value of x in lane `src` (low half) / lane `src + 32` (high half), src wave-uniform: two v_readlane pairs + a select
__device__ __forceinline__ double bcast_half(double x, int src, bool high) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  int lo0 = __builtin_amdgcn_readlane(lo, src), hi0 = __builtin_amdgcn_readlane(hi, src);
  int lo1 = __builtin_amdgcn_readlane(lo, src + 32), hi1 = __builtin_amdgcn_readlane(hi, src + 32);
  return __hiloint2double(high ? hi1 : hi0, high ? lo1 : lo0);
}

template <int R, int CPL>
__global__ __launch_bounds__(64, 2) void sg_step_kernel(SgKArgs a) {
  const int env = blockIdx.x, lane = threadIdx.x;
  if (env >= a.nenv) return;
  if (a.mode == 1 && a.mask && !a.mask[env]) return;
  const SgPlanHeader& H = *a.H;
  const int N = H.nelem, nv = H.nv, nu = H.nu, e0 = H.elem_dof0, nchain = H.nchain;
  const double h = H.timestep;
  __shared__ Smem<R, CPL> S;
  auto EL = [&](int f, int e) { return a.elem[(size_t)f * N + e]; };

  const int half = lane >> 5;
  const bool high = half != 0;
  const bool is_chain_lane = (lane & 31) == 0 && half < nchain;
  const SgChain& C = H.chain[half < nchain ? half : 0];
  ChainLds& CS = S.cs[half];

  // ---------------- load state ----------------
  double qe[R], ve[R], we[R], ke[R];
  const double kenv = a.kenv[env];
  const double kt0 = a.kmask_ten[H.t0_id] ? kenv : H.t0_k0;
  double* gq = a.qpos + (size_t)env * nv;
  double* gv = a.qvel + (size_t)env * nv;
  double* gw = a.warm + (size_t)env * nv;
  if (is_chain_lane) {
#pragma unroll
    for (int d = 0; d < SG_CD; d++) {
      int j = C.dof0 + d;
      if (a.mode == 1) { CS.q[d] = C.qpos0[d]; CS.v[d] = 0; CS.w[d] = 0; }
      else { CS.q[d] = gq[j]; CS.v[d] = gv[j]; CS.w[d] = gw[j]; }
      CS.k[d] = a.kmask_jnt[j] ? kenv : C.stiffness[d];
    }
    double act = 0, ctrl = 0;
    if (C.has_act) {
      if (a.mode == 1) a.ctrl[(size_t)env * nu + C.act_id] = 0;
      else { act = a.act[(size_t)env * nu + C.act_id]; ctrl = a.ctrl[(size_t)env * nu + C.act_id]; }
    }
    CS.act = act; CS.ctrl = ctrl; CS.act_dot = 0;
    CS.kten = C.has_ten ? (a.kmask_ten[C.ten_id] ? kenv : C.ten_k0) : 0.0;
  }
#pragma unroll
  for (int r = 0; r < R; r++) {
    int e = r * 64 + lane;
    qe[r] = ve[r] = we[r] = ke[r] = 0;
    if (e < N) {
      if (a.mode == 1) { qe[r] = EL(SGE_QPOS0, e); }
      else { qe[r] = gq[e0 + e]; ve[r] = gv[e0 + e]; we[r] = gw[e0 + e]; }
      ke[r] = a.kmask_jnt[e0 + e] ? kenv : EL(SGE_K0, e);
    }
  }

  int flags = 0, touch = 0, st_ncon = 0, st_nefc = 0, st_iters = 0;
  const int pre = a.mode == 1 ? 1 : 0;

  for (int step = -pre; step < a.nsub; step++) {
    const bool integrate = step >= 0;
    __syncthreads();
    // ---- mj_checkPos / mj_checkVel ----
    {
      int bad = 0;
#pragma unroll
      for (int r = 0; r < R; r++) bad |= (isbad(qe[r]) ? SG_FLAG_BADQPOS : 0) | (isbad(ve[r]) ? SG_FLAG_BADQVEL : 0);
      if (is_chain_lane) {
#pragma unroll
        for (int d = 0; d < SG_CD; d++) bad |= (isbad(CS.q[d]) ? SG_FLAG_BADQPOS : 0) | (isbad(CS.v[d]) ? SG_FLAG_BADQVEL : 0);
      }
      if (__ballot(bad != 0)) {
        flags |= (__ballot(bad & 1) ? 1 : 0) | (__ballot(bad & 2) ? 2 : 0);
        break;
      }
    }
    // ---- chains: kinematics + smooth dynamics + limit rows (lanes 0, 32) ----
    if (is_chain_lane) {
      double qc[SG_CD], vc[SG_CD], wc[SG_CD], kc[SG_CD];
#pragma unroll
      for (int d = 0; d < SG_CD; d++) { qc[d] = CS.q[d]; vc[d] = CS.v[d]; wc[d] = CS.w[d]; kc[d] = CS.k[d]; }
      ChainKin K;
      ChainDyn D;
      chain_kinematics(C, qc, K);
      chain_dynamics(C, K, qc, vc, CS.act, CS.ctrl, kc, CS.kten, H.gravity, D);
      S.K[half] = K;
#pragma unroll
      for (int i = 0; i < 16; i++) { CS.Minv[i] = D.Minv[i]; CS.M[i] = D.M[i]; }
#pragma unroll
      for (int d = 0; d < SG_CD; d++) { CS.qfrc_smooth[d] = D.qfrc_smooth[d]; CS.qacc_smooth[d] = D.qacc_smooth[d]; }
      CS.act_dot = D.act_dot;
#pragma unroll
      for (int g = 0; g < SG_CG; g++)
        if (g < C.ngeom) {
          double t[3], bp_[3], bm_[9], bm2[9];
          chain_body_pose(K, C.g_body[g], bp_, bm_);
          mulmat3(t, bm_, C.g_pos[g]);
          mulmat33(bm2, bm_, C.g_mat[g]);
#pragma unroll
          for (int k = 0; k < 3; k++) S.boxp[half * SG_CG + g][k] = bp_[k] + t[k];
#pragma unroll
          for (int k = 0; k < 9; k++) S.boxm[half * SG_CG + g][k] = bm2[k];
        }
      LimitRows L;
      limits_build(C, qc, vc, D.qacc_smooth, wc, L);
      CS.lim_active = L.active;
#pragma unroll
      for (int k = 0; k < SG_MAXLIM; k++) { CS.lim_sign[k] = L.sign[k]; CS.lim_R[k] = L.R[k]; CS.lim_b[k] = L.b[k]; CS.lim_f[k] = L.f[k]; }
    }
    // ---- elements: smooth dynamics ----
    double invm[R], fsm[R], asme[R], coef[R];
    double L0p = 0, Ldp = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
      int e = r * 64 + lane;
      coef[r] = e < N ? EL(SGE_COEF, e) : 0.0;
      L0p += coef[r] * qe[r]; Ldp += coef[r] * ve[r];
    }
    const double L0 = wave_sum(L0p), Ld = wave_sum(Ldp);
    const double frc_t0 = -kt0 * (L0 - H.t0_lspring) - H.t0_damping * Ld;
    int unsupported = 0;
    int ns0 = 0, ns1 = 0;  // contacts staged per stream
    {
      double cpos[R][3];
#pragma unroll
      for (int r = 0; r < R; r++) {
        int e = r * 64 + lane;
        invm[r] = fsm[r] = asme[r] = 0;
        cpos[r][0] = cpos[r][1] = cpos[r][2] = 1e30;
        if (e < N) {
          double ax[3] = {EL(SGE_AX, e), EL(SGE_AY, e), EL(SGE_AZ, e)}, m = EL(SGE_MASS, e);
          double bias = -m * dot3(H.gravity, ax);
          double f = -ke[r] * (qe[r] - EL(SGE_SPRINGREF, e)) - EL(SGE_DAMPING, e) * ve[r] + coef[r] * frc_t0 - bias;
          invm[r] = 1.0 / (m + EL(SGE_ARMATURE, e));
          fsm[r] = f; asme[r] = f * invm[r];
          double dq = qe[r] - EL(SGE_QPOS0, e);
          cpos[r][0] = EL(SGE_GX, e) + ax[0] * dq; cpos[r][1] = EL(SGE_GY, e) + ax[1] * dq; cpos[r][2] = EL(SGE_GZ, e) + ax[2] * dq;
          if (!(qe[r] > EL(SGE_QLO, e) && qe[r] < EL(SGE_QHI, e))) unsupported = 1;
          S.ve[e] = ve[r]; S.asme[e] = asme[r]; S.we[e] = we[r];
          S.owner[e] = -1;
        }
      }
      __syncthreads();
      // ---- collision: per chain, per box: centre sphere, then elements in index order ----
      int overflow = 0;
      touch = 0;
#pragma unroll
      for (int c = 0; c < SG_MAXCH; c++) {
        if (c >= nchain) break;
        const SgChain& Cc = H.chain[c];
        int nsc = 0;
#pragma unroll
        for (int g = 0; g < SG_CG; g++) {
          if (g >= Cc.ngeom) break;
          const int b = c * SG_CG + g;
          double bp[3], bm[9], sz[3];
#pragma unroll
          for (int k = 0; k < 3; k++) { bp[k] = S.boxp[b][k]; sz[k] = Cc.g_size[g][k]; }
#pragma unroll
          for (int k = 0; k < 9; k++) bm[k] = S.boxm[b][k];
          const double rb = Cc.g_rbound[g];
          if (H.has_center) {  // uniform: every lane computes the same test
            double dif[3] = {bp[0] - H.center_pos[0], bp[1] - H.center_pos[1], bp[2] - H.center_pos[2]}, bound = H.center_radius + rb + H.con_margin;
            ConRec rc;
            if (dot3(dif, dif) <= bound * bound && sphere_box(H.center_pos, H.center_radius, bp, bm, sz, H.con_margin, rc) && rc.dist < H.con_margin) {
              int slot = nsc;
              if (slot < 32 * CPL) {
                if (lane == 0) {
                  StageRec& s = S.stage[c][slot];
                  s.dist = rc.dist; s.sl = -1; s.box = g;
                  for (int k = 0; k < 3; k++) { s.pos[k] = rc.pos[k]; s.n[k] = rc.n[k]; }
                }
                nsc = slot + 1;
                touch |= 1 << b;
              } else overflow = 1;
            }
          }
#pragma unroll
          for (int r = 0; r < R; r++) {
            int e = r * 64 + lane, n = 0;
            ConRec r0, r1;
            bool v0 = false, v1 = false;
            double dif[3] = {bp[0] - cpos[r][0], bp[1] - cpos[r][1], bp[2] - cpos[r][2]}, bound = H.cap_rbound + rb + H.con_margin;
            if (e < N && dot3(dif, dif) <= bound * bound) {
              double cax[3] = {EL(SGE_CX, e), EL(SGE_CY, e), EL(SGE_CZ, e)};
              int mk = capsule_box(cpos[r], cax, H.cap_radius, H.cap_hl, bp, bm, sz, H.con_margin, r0, r1);
              v0 = (mk & 1) && r0.dist < H.con_margin;
              v1 = (mk & 2) && r1.dist < H.con_margin;
              n = (int)v0 + (int)v1;
            }
            unsigned long long m1 = __ballot(n >= 1), m2 = __ballot(n >= 2);
            int base = nsc + lanes_below(m1) + lanes_below(m2);
            int total = __popcll(m1) + __popcll(m2);
            if (v0 && base < 32 * CPL) {
              StageRec& s = S.stage[c][base];
              s.dist = r0.dist; s.sl = e; s.box = g;
              for (int q = 0; q < 3; q++) { s.pos[q] = r0.pos[q]; s.n[q] = r0.n[q]; }
            }
            if (v1 && base + (int)v0 < 32 * CPL) {
              StageRec& s = S.stage[c][base + (int)v0];
              s.dist = r1.dist; s.sl = e; s.box = g;
              for (int q = 0; q < 3; q++) { s.pos[q] = r1.pos[q]; s.n[q] = r1.n[q]; }
            }
            if (n > 0) {  // which streams touch this slider (streams may only run together when they share none)
              int o = S.owner[e];
              S.owner[e] = (o < 0 || o == c) ? c : 2;
            }
            if (total) touch |= 1 << b;
            nsc += total;
            if (nsc > 32 * CPL) { nsc = 32 * CPL; overflow = 1; }
          }
        }
        if (c == 0) ns0 = nsc; else ns1 = nsc;
      }
      if (overflow) flags |= SG_FLAG_CONTACTFULL;
    }
    // envelope checks: pairs that are legal in the model but outside the supported class
    {
      int nb = nchain * SG_CG, npairs = nb * H.nstatic;
      if (lane < npairs) {
        int b = lane / H.nstatic, s = lane % H.nstatic, c = b / SG_CG, g = b % SG_CG;
        if (g < H.chain[c].ngeom) {
          const SgChain& Cc = H.chain[c];
          double dif[3] = {S.boxp[b][0] - H.st_pos[s][0], S.boxp[b][1] - H.st_pos[s][1], S.boxp[b][2] - H.st_pos[s][2]}, bd = Cc.g_rbound[g] + H.st_rbound[s];
          if (dot3(dif, dif) <= bd * bd && box_box_overlap(S.boxp[b], S.boxm[b], Cc.g_size[g], H.st_pos[s], H.st_mat[s], H.st_size[s], 0)) unsupported = 1;
        }
      } else if (lane >= 32 && lane < 32 + SG_CG * SG_CG && nchain == 2) {
        int g = (lane - 32) / SG_CG, g2 = (lane - 32) % SG_CG;
        if (g < H.chain[0].ngeom && g2 < H.chain[1].ngeom) {
          int b = g, b2 = SG_CG + g2;
          double dif[3] = {S.boxp[b][0] - S.boxp[b2][0], S.boxp[b][1] - S.boxp[b2][1], S.boxp[b][2] - S.boxp[b2][2]}, bd = H.chain[0].g_rbound[g] + H.chain[1].g_rbound[g2];
          if (dot3(dif, dif) <= bd * bd && box_box_overlap(S.boxp[b], S.boxm[b], H.chain[0].g_size[g], S.boxp[b2], S.boxm[b2], H.chain[1].g_size[g2], 0)) unsupported = 1;
        }
      } else if (lane >= 48 && lane < 48 + SG_MAXCH * SG_CG && H.has_plane) {
        int b = lane - 48, c = b / SG_CG, g = b % SG_CG;
        if (c < nchain && g < H.chain[c].ngeom) {
          double dif[3] = {S.boxp[b][0] - H.plane_pos[0], S.boxp[b][1] - H.plane_pos[1], S.boxp[b][2] - H.plane_pos[2]}, ext = 0;
          for (int k = 0; k < 3; k++)
            ext += H.chain[c].g_size[g][k] * fabs(H.plane_normal[0] * S.boxm[b][k] + H.plane_normal[1] * S.boxm[b][3 + k] + H.plane_normal[2] * S.boxm[b][6 + k]);
          if (dot3(dif, H.plane_normal) - ext <= 0) unsupported = 1;
        }
      }
      if (__ballot(unsupported)) flags |= SG_FLAG_UNSUPPORTED_PAIR;
    }
    __syncthreads();
    // do the two streams share a slider?  (then they are swept one after the other)
    int shared_slider = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
      int e = r * 64 + lane;
      if (e < N && S.owner[e] == 2) shared_slider = 1;
    }
    shared_slider = __ballot(shared_slider) != 0;

    // ---- contact rows: owner lanes build their contacts from the staged geometry ----
    Contact ct[CPL];
    const int myn = high ? ns1 : ns0;
#pragma unroll
    for (int k = 0; k < CPL; k++) {
      int i = (lane & 31) + 32 * k;
      ct[k].sl = -1; ct[k].invm = 0; ct[k].R = 1;
#pragma unroll
      for (int r = 0; r < 3; r++) {
        ct[k].f[r] = ct[k].b[r] = ct[k].Js[r] = 0;
#pragma unroll
        for (int d = 0; d < SG_CD; d++) ct[k].Jf[r][d] = 0;
      }
#pragma unroll
      for (int q = 0; q < 6; q++) ct[k].A[q] = (q == 0 || q == 3 || q == 5) ? 1.0 : 0.0;
      if (i < myn) {
        const StageRec& s = S.stage[half][i];
        ConRec rec;
        rec.dist = s.dist;
        for (int q = 0; q < 3; q++) { rec.pos[q] = s.pos[q]; rec.n[q] = s.n[q]; }
        int sl = s.sl, g = s.box, bi = C.g_body[g], nd = chain_ndof_of_body(bi);
        double ax[3] = {0, 0, 0}, ve_ = 0, as_ = 0, we_ = 0, im = 0, bw = 0;
        if (sl >= 0) {
          ax[0] = EL(SGE_AX, sl); ax[1] = EL(SGE_AY, sl); ax[2] = EL(SGE_AZ, sl);
          ve_ = S.ve[sl]; as_ = S.asme[sl]; we_ = S.we[sl];
          im = 1.0 / (EL(SGE_MASS, sl) + EL(SGE_ARMATURE, sl)); bw = EL(SGE_BINVW, sl);
        }
        contact_build(ct[k], rec, S.K[half], nd, CS.Minv, CS.v, CS.qacc_smooth, CS.w, C.b_invw_tran[bi], sl, ax, ve_, as_, we_, im, bw, H);
      }
    }
    // ---- equality rows ----
    double eqR[R], eqb[R], eqf[R];
    double tbp = 0, tjp = 0, tAp = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
      int e = r * 64 + lane;
      eqR[r] = 1; eqb[r] = 0; eqf[r] = 0;
      if (e < N) {
        double pos = qe[r] - EL(SGE_QPOS0, e), imp = impedance(H.eqj_solimp, pos, 0);
        eqR[r] = fmax(SG_MINVAL, (1 - imp) / imp * EL(SGE_INVW, e));
        double aref = -H.eqj_B * ve[r] - H.eqj_K * imp * pos;
        eqb[r] = asme[r] - aref;
        eqf[r] = -(we[r] - aref) / eqR[r];
        tbp += coef[r] * asme[r]; tjp += coef[r] * we[r]; tAp += coef[r] * coef[r] * invm[r];
      }
    }
    const double tpos = L0 - H.t0_L0, timp = impedance(H.eqt_solimp, tpos, 0), tR = fmax(SG_MINVAL, (1 - timp) / timp * H.eqt_invw);
    const double taref = -H.eqt_B * Ld - H.eqt_K * timp * tpos;
    const double tb = wave_sum(tbp) - taref, tjar = wave_sum(tjp) - taref, tA = wave_sum(tAp) + tR;
    double tf = -tjar / tR;

    const int limact = S.cs[0].lim_active | (nchain > 1 ? S.cs[1].lim_active : 0);
    st_ncon = ns0 + ns1;
    st_nefc = N + 1 + 3 * st_ncon + __popc(S.cs[0].lim_active) + (nchain > 1 ? __popc(S.cs[1].lim_active) : 0);
    const int nmaxs = ns0 > ns1 ? ns0 : ns1;

    // ---- M^-1 J' f from scratch: sliders into S.as, chains into aF (replicated per half) ----
    double aF[SG_CD];
    auto recompute_a = [&]() {
#pragma unroll
      for (int r = 0; r < R; r++) {
        int e = r * 64 + lane;
        if (e < N) S.as[e] = invm[r] * (eqf[r] + coef[r] * tf);
      }
      __syncthreads();
      // contact contributions to the sliders, in contact order (deterministic)
      double g[SG_CD] = {0, 0, 0, 0};
      for (int pass = 0; pass < (shared_slider ? 2 : 1); pass++)
#pragma unroll
        for (int k = 0; k < CPL; k++)
          for (int ii = 0; ii < 32; ii++) {
            int i = 32 * k + ii;
            if (i >= nmaxs) break;
            bool mine = (lane & 31) == ii && i < myn && (!shared_slider || half == pass);
            if (mine && ct[k].sl >= 0) S.as[ct[k].sl] += ct[k].invm * (ct[k].Js[0] * ct[k].f[0] + ct[k].Js[1] * ct[k].f[1] + ct[k].Js[2] * ct[k].f[2]);
          }
#pragma unroll
      for (int k = 0; k < CPL; k++)
        if ((lane & 31) + 32 * k < myn)
#pragma unroll
          for (int d = 0; d < SG_CD; d++) g[d] += ct[k].Jf[0][d] * ct[k].f[0] + ct[k].Jf[1][d] * ct[k].f[1] + ct[k].Jf[2][d] * ct[k].f[2];
      if (is_chain_lane) {
        const int la = CS.lim_active;
#pragma unroll
        for (int k = 0; k < SG_MAXLIM; k++)
          if (la >> k & 1) g[k / 2] += CS.lim_sign[k] * CS.lim_f[k];
      }
      // sum g over the half (32 lanes), then aF = Minv g
#pragma unroll
      for (int d = 0; d < SG_CD; d++) {
        double x = g[d];
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) x += __shfl_xor(x, o);
        g[d] = x;
      }
#pragma unroll
      for (int a2 = 0; a2 < SG_CD; a2++) {
        double s = 0;
#pragma unroll
        for (int b2 = 0; b2 < SG_CD; b2++) s += CS.Minv[4 * a2 + b2] * g[b2];
        aF[a2] = s;
      }
      __syncthreads();
    };
    recompute_a();
    // ---- warmstart cost 0.5 f'(A+R)f + f'b; keep the warmstart only if it beats f = 0 ----
    {
      double cp = 0, tJap = 0;
#pragma unroll
      for (int r = 0; r < R; r++) {
        int e = r * 64 + lane;
        if (e < N) { double ae = S.as[e]; cp += eqf[r] * (0.5 * (ae + eqR[r] * eqf[r]) + eqb[r]); tJap += coef[r] * ae; }
      }
      double tJa = wave_sum(tJap);
      if (lane == 0) cp += tf * (0.5 * (tJa + tR * tf) + tb);
      if (is_chain_lane) {
        const int la = CS.lim_active;
#pragma unroll
        for (int k = 0; k < SG_MAXLIM; k++)
          if (la >> k & 1) cp += CS.lim_f[k] * (0.5 * (CS.lim_sign[k] * aF[k / 2] + CS.lim_R[k] * CS.lim_f[k]) + CS.lim_b[k]);
      }
#pragma unroll
      for (int k = 0; k < CPL; k++)
        if ((lane & 31) + 32 * k < myn) {
          double as_ = ct[k].sl >= 0 ? S.as[ct[k].sl] : 0.0;
#pragma unroll
          for (int r = 0; r < 3; r++) {
            double Ja = ct[k].Js[r] * as_;
#pragma unroll
            for (int d = 0; d < SG_CD; d++) Ja += ct[k].Jf[r][d] * aF[d];
            cp += ct[k].f[r] * (0.5 * (Ja + ct[k].R * ct[k].f[r]) + ct[k].b[r]);
          }
        }
      double cost = wave_sum(cp);
      if (cost > 0) {
#pragma unroll
        for (int r = 0; r < R; r++) eqf[r] = 0;
        tf = 0;
        if (is_chain_lane) {
#pragma unroll
          for (int k = 0; k < SG_MAXLIM; k++) CS.lim_f[k] = 0;
        }
#pragma unroll
        for (int k = 0; k < CPL; k++) ct[k].f[0] = ct[k].f[1] = ct[k].f[2] = 0;
        __syncthreads();
        recompute_a();
      }
    }
    // ---- PGS sweeps ----
    st_iters = 0;
    for (int it = 0; it < H.iterations; it++) {
      double imp_acc = 0;
      // joint-fix rows (mutually independent) then the tendon row
      double tJap = 0;
      double ael[R];
#pragma unroll
      for (int r = 0; r < R; r++) {
        int e = r * 64 + lane;
        ael[r] = 0;
        if (e < N) {
          double ae = S.as[e], old = eqf[r];
          imp_acc -= scalar_update(eqf[r], eqb[r], ae, eqR[r], invm[r] + eqR[r], false);
          ae += invm[r] * (eqf[r] - old);
          ael[r] = ae;
          tJap += coef[r] * ae;
        }
      }
      {
        double Ja = wave_sum(tJap), old = tf, tfn = tf;
        double ch = scalar_update(tfn, tb, Ja, tR, tA, false);
        if (lane == 0) imp_acc -= ch;
        tf = tfn;
        double dft = tf - old;
#pragma unroll
        for (int r = 0; r < R; r++) {
          int e = r * 64 + lane;
          if (e < N) S.as[e] = ael[r] + invm[r] * coef[r] * dft;
        }
      }
      __syncthreads();
      // limits then contacts; stream 0 on the low half, stream 1 on the high half
      for (int pass = 0; pass < (shared_slider ? 2 : 1); pass++) {
        const bool act_half = !shared_slider || half == pass;
#pragma unroll
        for (int k = 0; k < SG_MAXLIM; k++) {
          if (!(limact >> k & 1)) continue;  // uniform
          const int d = k / 2;
          double dA[SG_CD] = {0, 0, 0, 0};
          if (is_chain_lane && act_half && (CS.lim_active >> k & 1)) {
            double f = CS.lim_f[k], old = f, sg = CS.lim_sign[k], Rr = CS.lim_R[k];
            imp_acc -= scalar_update(f, CS.lim_b[k], sg * aF[d], Rr, CS.Minv[5 * d] + Rr, true);
            CS.lim_f[k] = f;
            double df = sg * (f - old);
#pragma unroll
            for (int q = 0; q < SG_CD; q++) dA[q] = CS.Minv[4 * q + d] * df;
          }
#pragma unroll
          for (int q = 0; q < SG_CD; q++) aF[q] += bcast_half(dA[q], 0, high);
        }
#pragma unroll
        for (int k = 0; k < CPL; k++)
          for (int ii = 0; ii < 32; ii++) {
            const int i = 32 * k + ii;
            if (i >= nmaxs) break;
            double dA[SG_CD] = {0, 0, 0, 0};
            if ((lane & 31) == ii && i < myn && act_half) {
              const int sl = ct[k].sl;
              double as_ = sl >= 0 ? S.as[sl] : 0.0, df[3];
              imp_acc -= contact_update(ct[k], aF, as_, H.con_mu, df);
              if (sl >= 0) S.as[sl] = as_ + ct[k].invm * (ct[k].Js[0] * df[0] + ct[k].Js[1] * df[1] + ct[k].Js[2] * df[2]);
              double g[SG_CD];
#pragma unroll
              for (int d = 0; d < SG_CD; d++) g[d] = ct[k].Jf[0][d] * df[0] + ct[k].Jf[1][d] * df[1] + ct[k].Jf[2][d] * df[2];
#pragma unroll
              for (int q = 0; q < SG_CD; q++) {
                double s = 0;
#pragma unroll
                for (int d = 0; d < SG_CD; d++) s += CS.Minv[4 * q + d] * g[d];
                dA[q] = s;
              }
            }
#pragma unroll
            for (int q = 0; q < SG_CD; q++) aF[q] += bcast_half(dA[q], ii, high);
          }
        __syncthreads();
      }
      st_iters = it + 1;
      double improvement = wave_sum(imp_acc) * H.pgs_scale;
      if (improvement < H.tolerance) break;
    }
    recompute_a();
    // ---- accelerations, sensors (stage 11), warmstart for the next solve ----
    double qacc_e[R];
    int badacc = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
      int e = r * 64 + lane;
      qacc_e[r] = 0;
      if (e < N) { qacc_e[r] = asme[r] + S.as[e]; if (isbad(qacc_e[r])) badacc = 1; }
    }
    if (is_chain_lane) {
#pragma unroll
      for (int d = 0; d < SG_CD; d++)
        if (isbad(CS.qacc_smooth[d] + aF[d])) badacc = 1;
    }
    // mj_checkAcc: a bad acceleration anywhere stops the env before anything is integrated (uniform through the ballot)
    const bool anybadacc = __ballot(badacc) != 0;
    if (is_chain_lane) {
      double qacc_c[SG_CD], vc[SG_CD];
#pragma unroll
      for (int d = 0; d < SG_CD; d++) {
        qacc_c[d] = CS.qacc_smooth[d] + aF[d];
        vc[d] = CS.v[d];
      }
      if (a.sens) {
        ChainMotion Mo;
        chain_motion(C, S.K[half], vc, qacc_c, H.gravity, Mo);
        double* so = a.sens + (size_t)env * a.sens_stride;
        for (int s = 0; s < C.nsite; s++) {
          int bi = C.s_body[s];
          double r3[3], sm[9], t[3], t2[3], acc[3], out[3], sbp[3], sbm[9], bw[3], bal[3];
          chain_body_pose(S.K[half], bi, sbp, sbm);
          mulmat3(r3, sbm, C.s_pos[s]);
          mulmat33(sm, sbm, C.s_mat[s]);
          for (int k = 0; k < 3; k++) { bw[k] = bi == 0 ? Mo.w[0][k] : Mo.w[SG_CB - 1][k]; bal[k] = bi == 0 ? Mo.al[0][k] : Mo.al[SG_CB - 1][k]; }
          if (C.s_gyro_adr[s] >= 0) {
            mulmatT3(out, sm, bw);
            for (int k = 0; k < 3; k++) so[C.s_gyro_adr[s] + k] = out[k];
          }
          if (C.s_acc_adr[s] >= 0) {
            for (int k = 0; k < 3; k++) acc[k] = bi == 0 ? Mo.a[0][k] : Mo.a[SG_CB - 1][k];
            cross3(t, bal, r3); addscl3(acc, t, 1);
            cross3(t, bw, r3); cross3(t2, bw, t); addscl3(acc, t2, 1);
            mulmatT3(out, sm, acc);
            for (int k = 0; k < 3; k++) so[C.s_acc_adr[s] + k] = out[k];
          }
        }
      }
      if (!anybadacc) {
#pragma unroll
        for (int d = 0; d < SG_CD; d++) CS.w[d] = qacc_c[d];
        if (integrate) {  // Euler with implicit joint damping (stage 12), chain part
          bool damp = false;
#pragma unroll
          for (int d = 0; d < SG_CD; d++) damp |= C.damping[d] > 0;
          double qa[SG_CD];
          if (damp) {
            double MhB[16], MhBinv[16], rhs[SG_CD];
#pragma unroll
            for (int i = 0; i < 16; i++) MhB[i] = CS.M[i];
#pragma unroll
            for (int d = 0; d < SG_CD; d++) MhB[5 * d] += h * C.damping[d];
            spd_inverse4(MhB, MhBinv);
#pragma unroll
            for (int a2 = 0; a2 < SG_CD; a2++) {
              double s = CS.qfrc_smooth[a2];
#pragma unroll
              for (int b2 = 0; b2 < SG_CD; b2++) s += CS.M[4 * a2 + b2] * aF[b2];
              rhs[a2] = s;
            }
#pragma unroll
            for (int a2 = 0; a2 < SG_CD; a2++) {
              double s = 0;
#pragma unroll
              for (int b2 = 0; b2 < SG_CD; b2++) s += MhBinv[4 * a2 + b2] * rhs[b2];
              qa[a2] = s;
            }
          } else {
#pragma unroll
            for (int d = 0; d < SG_CD; d++) qa[d] = qacc_c[d];
          }
          CS.act += h * CS.act_dot;
#pragma unroll
          for (int d = 0; d < SG_CD; d++) { double vn = CS.v[d] + h * qa[d]; CS.v[d] = vn; CS.q[d] += h * vn; }
        }
      }
    }
    if (anybadacc) { flags |= SG_FLAG_BADQACC; break; }
#pragma unroll
    for (int r = 0; r < R; r++) we[r] = qacc_e[r];
    if (!integrate) continue;
    double qa_e[R], yc[R], Sp = 0;
#pragma unroll
    for (int r = 0; r < R; r++) {
      int e = r * 64 + lane;
      qa_e[r] = yc[r] = 0;
      if (e < N) {
        double m = 1.0 / invm[r];
        double den = m + h * EL(SGE_DAMPING, e);
        qa_e[r] = (fsm[r] + m * S.as[e]) / den;
        if (H.t0_implicit) { yc[r] = coef[r] / den; Sp += coef[r] * qa_e[r]; }
      }
    }
    if (H.t0_implicit) {  // deviation D5, as in sg_phase_kernel's FINISH
      const double kk = h * H.t0_damping * wave_sum(Sp) / (1.0 + H.t0_hcT);
#pragma unroll
      for (int r = 0; r < R; r++) qa_e[r] -= yc[r] * kk;
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
      int e = r * 64 + lane;
      if (e < N) {
        ve[r] += h * qa_e[r];
        qe[r] += h * ve[r];
      }
    }
  }

  // ---------------- store state ----------------
  __syncthreads();
  if (is_chain_lane) {
#pragma unroll
    for (int d = 0; d < SG_CD; d++) { int j = C.dof0 + d; gq[j] = CS.q[d]; gv[j] = CS.v[d]; gw[j] = CS.w[d]; }
    if (C.has_act) a.act[(size_t)env * nu + C.act_id] = CS.act;
  }
#pragma unroll
  for (int r = 0; r < R; r++) {
    int e = r * 64 + lane;
    if (e < N) { gq[e0 + e] = qe[r]; gv[e0 + e] = ve[r]; gw[e0 + e] = we[r]; }
  }
  if (lane == 0) {
    if (a.flags) a.flags[env] = flags;
    if (a.touch) a.touch[env] = touch;
    if (a.ncon) a.ncon[env] = st_ncon;
    if (a.nefc) a.nefc[env] = st_nefc;
    if (a.iters) a.iters[env] = st_iters;
  }
}
