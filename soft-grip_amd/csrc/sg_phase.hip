// sg_phase.hip -- the per-env stages around the solver (rows pipeline; the split pipeline of test builds shares them):
//
//   sg_chain_kernel         one LANE per finger chain: FINISH of the previous substep for the chain dofs (qacc, accelerometer / gyro,
//                           warm start, Euler) and BEGIN (kinematics, 4 x 4 M and M^-1, RNE bias, tendon + actuator, limit rows).
//   sg_phase_kernel<R,CPL,NB,GEN>  one WAVEFRONT per env, lane = shell element.  FINISH part: takes the solver's result for the
//                           previous substep (slider and chain constraint accelerations), produces qacc, the warmstart and
//                           integrates.  BEGIN part: smooth dynamics, collision, constraint rows, warmstart test, and EXPORTS the
//                           constraint problem to the work space in the solver's layouts (sg_work.h).
//
// What they replace: mj_step's stages around mj_fwdConstraint for reference environment/manenv.py:48-49.
#include "sg_work.h"

// ------------------------------------------------------------------------------------------------
// General contact path in the phase kernel (sg_general.h): an env in which a pair outside the fast path's two kinds is within reach
// builds, for this substep, ONE ordered contact list over the plan's candidate pairs and exports generic rows (W.gcon); the solver
// sweeps them as one serial stream.  Rare by construction (never in the reference's scenes), so it is kept OUT of the kernel's
// straight-line code: __noinline__ functions working on the kernel's LDS block through pointers.  In this mode the LDS regions of the
// fast path's contact staging are re-used (layout at the call site in sg_phase_kernel).
// ------------------------------------------------------------------------------------------------
struct GenLds {
  StageRec2* stage;            // [SG_GEN_MAXCON]
  double* gas;                 // [R * 64] per element: invm * sum of Js' f over the general contacts on its slider (inside the staging area)
  double* gg;                  // [SG_MAXCH][SG_CD]: sum over the general contacts of Jf[c]' f
  double* tmp;                 // 96 doubles (box - box clipping)
  const double (*boxp)[3];
  const double (*boxm)[9];
  const ChainKin* K;
  const ChainLds2* cs;
  const double *qe, *ve, *asme, *we;   // per element (LDS): slider position, velocity, smooth acceleration, warmstart
};

// geometry of a capsule of the pair table
__device__ __forceinline__ void sg_gen_capsule(const SgPhaseArgs& a, const GenLds& L, int e, double* cp, double* cax) {
  const int N = a.nelem;
  auto EL = [&](int f, int k) { return a.elem[(size_t)f * N + k]; };
  const double dq = L.qe[e] - EL(SGE_QPOS0, e);
  cp[0] = EL(SGE_GX, e) + EL(SGE_AX, e) * dq; cp[1] = EL(SGE_GY, e) + EL(SGE_AY, e) * dq; cp[2] = EL(SGE_GZ, e) + EL(SGE_AZ, e) * dq;
  cax[0] = EL(SGE_CX, e); cax[1] = EL(SGE_CY, e); cax[2] = EL(SGE_CZ, e);
}

// collision + rows.  Returns the number of contacts; *flags gets CONTACTFULL / UNSUPPORTED_PAIR bits, *touch the finger-box bits.
__device__ __noinline__ int sg_gen_phase(const SgPhaseArgs& a, const SgPlanHeader& H, const int env, const GenLds L, int* flags, int* touch) {
  const int lane = threadIdx.x, N = a.nelem;
  auto EL = [&](int f, int k) { return a.elem[(size_t)f * N + k]; };
  const SgWork& W = a.w;
  int ng = 0, fl = 0;
  // ---- the pair table, 64 pairs per pass: lane = pair.  Pairs with at most two contacts (capsule / sphere against a box, plane against a
  //      capsule) are evaluated by their lanes and appended in order with ballots; a box against a box or the plane (up to 8 / 4
  //      contacts) within reach is evaluated by lane 0 at its place in the order
#pragma unroll 1
  for (int p0 = 0; p0 < H.ngpair; p0 += 64) {
    const int pi = p0 + lane;
    const bool have = pi < H.ngpair;
    SgGenPair gp;
    gp.kind = SGP_UNSUPPORTED; gp.g1 = gp.g2 = 0; gp.pad = 0;
    if (have) gp = a.gpairs[pi];
    ConRec r0, r1;
    int n = 0;
    bool big = false;
    if (have) {
      double cp[3] = {0, 0, 0}, cax[3] = {0, 0, 1};
      if (gp.kind == SGP_PLANE_CAP) {
        sg_gen_capsule(a, L, sgg_index(gp.g2), cp, cax);
        const double dif[3] = {cp[0] - H.plane_pos[0], cp[1] - H.plane_pos[1], cp[2] - H.plane_pos[2]};
        if (!(dot3(dif, H.plane_normal) > H.con_margin + H.cap_rbound))
          n = gen_plane_capsule(H.plane_pos, H.plane_normal, cp, cax, H.cap_radius, H.cap_hl, H.con_margin, r0, r1);
      } else if (gp.kind == SGP_PLANE_BOX) {
        const double *p2, *R2, *s2;
        double rb2;
        gen_box_of(gp.g2, H, L.boxp, L.boxm, p2, R2, s2, rb2);
        const double dif[3] = {p2[0] - H.plane_pos[0], p2[1] - H.plane_pos[1], p2[2] - H.plane_pos[2]};
        big = !(dot3(dif, H.plane_normal) > H.con_margin + rb2);
      } else {
        const double *p1 = nullptr, *R1 = nullptr, *s1 = nullptr, *p2, *R2, *s2;
        double rb1 = 0, rb2;
        const bool boxes = sgg_kind(gp.g1) == SGG_BOX || sgg_kind(gp.g1) == SGG_STATIC;
        if (gp.kind == SGP_SPH_BOX) { cp[0] = H.center_pos[0]; cp[1] = H.center_pos[1]; cp[2] = H.center_pos[2]; rb1 = H.center_radius; }
        else if (gp.kind == SGP_CAP_BOX) { sg_gen_capsule(a, L, sgg_index(gp.g1), cp, cax); rb1 = H.cap_rbound; }
        else if (boxes) { gen_box_of(gp.g1, H, L.boxp, L.boxm, p1, R1, s1, rb1); cp[0] = p1[0]; cp[1] = p1[1]; cp[2] = p1[2]; }
        if (gp.kind == SGP_UNSUPPORTED && !boxes) fl |= SG_FLAG_UNSUPPORTED_PAIR;   // cannot even be tested: flagged whenever this path runs
        else {
          gen_box_of(gp.g2, H, L.boxp, L.boxm, p2, R2, s2, rb2);
          const double dif[3] = {p2[0] - cp[0], p2[1] - cp[1], p2[2] - cp[2]}, bound = rb1 + rb2 + H.con_margin;
          if (dot3(dif, dif) <= bound * bound) {
            if (gp.kind == SGP_SPH_BOX) n = sphere_box(cp, H.center_radius, p2, R2, s2, H.con_margin, r0);
            else if (gp.kind == SGP_CAP_BOX) {
              const int mk = capsule_box(cp, cax, H.cap_radius, H.cap_hl, p2, R2, s2, H.con_margin, r0, r1);
              if ((mk & 2) && !(mk & 1)) r0 = r1;
              n = (mk & 1) + ((mk >> 1) & 1);
            } else big = true;
          }
        }
      }
      // only contacts inside the margin become constraints (and count, as on the fast path)
      if (n == 2 && !(r1.dist < H.con_margin)) n = 1;
      if (n >= 1 && !(r0.dist < H.con_margin)) { r0 = r1; n--; }
    }
    // ordered append, split at the big pairs
    unsigned long long todo = __ballot(n > 0 || big);
#pragma unroll 1
    while (todo) {
      const unsigned long long bigm = __ballot(big) & todo;
      const int Lb = bigm ? __ffsll((long long)bigm) - 1 : 64;
      const unsigned long long seg = Lb == 64 ? todo : (todo & ((1ull << Lb) - 1ull));
      const bool mine = (seg >> lane) & 1ull;
      const unsigned long long m1 = __ballot(mine && n >= 1), m2 = __ballot(mine && n >= 2);
      const int base = ng + lanes_below2(m1) + lanes_below2(m2);
      auto put = [&](int at, const ConRec& r) {
        StageRec2& o = L.stage[at];
        o.dist = r.dist; o.sl = pi; o.box = 0;
        for (int q = 0; q < 3; q++) { o.pos[q] = r.pos[q]; o.n[q] = r.n[q]; }
      };
      if (mine && n >= 1 && base < SG_GEN_MAXCON) put(base, r0);
      if (mine && n >= 2 && base + 1 < SG_GEN_MAXCON) put(base + 1, r1);
      ng += __popcll(m1) + __popcll(m2);
      todo &= ~seg;
      if (Lb < 64) {
        const int pb = p0 + Lb;            // uniform
        const SgGenPair gb = a.gpairs[pb];
        int cnt = 0;
        if (lane == 0) {
          const int room = SG_GEN_MAXCON - (ng < SG_GEN_MAXCON ? ng : SG_GEN_MAXCON);
          StageRec2* out = L.stage + (ng < SG_GEN_MAXCON ? ng : 0);
          const double *p2, *R2, *s2;
          double rb2;
          gen_box_of(gb.g2, H, L.boxp, L.boxm, p2, R2, s2, rb2);
          int nb = 0;
          if (room >= 8) {   // the routines write up to 8 records: without room for them the list is full
            if (gb.kind == SGP_PLANE_BOX) nb = gen_plane_box(H.plane_pos, H.plane_normal, p2, R2, s2, H.con_margin, out);
            else {
              const double *p1, *R1, *s1;
              double rb1;
              gen_box_of(gb.g1, H, L.boxp, L.boxm, p1, R1, s1, rb1);
              nb = gen_box_box(p1, R1, s1, p2, R2, s2, H.con_margin, out, (double (*)[3])L.tmp, (double (*)[3])(L.tmp + 48));
            }
            for (int q = 0; q < nb; q++)      // keep the contacts inside the margin, in place
              if (out[q].dist < H.con_margin) { if (cnt != q) out[cnt] = out[q]; out[cnt].sl = pb; cnt++; }
            if (gb.kind == SGP_UNSUPPORTED && cnt > 0) { cnt = 0; fl |= SG_FLAG_UNSUPPORTED_PAIR; }
          } else {
            fl |= SG_FLAG_CONTACTFULL;
          }
        }
        ng += __shfl(cnt, 0);
        todo &= ~(1ull << Lb);
        if (lane == Lb) big = false;
      }
    }
  }
  if (ng > SG_GEN_MAXCON) { ng = SG_GEN_MAXCON; fl |= SG_FLAG_CONTACTFULL; }
  __syncthreads();
  // ---- rows: lane = contact, SG_GEN_ROUNDS rounds.  Exported to W.gcon; what the warmstart test and the slider accelerations need
  //      (slider index and push per contact) goes back into the staging area once every lane has read its record
  double Minv2[SG_MAXCH][16], vc2[SG_MAXCH][SG_CD], asm2[SG_MAXCH][SG_CD], warm2[SG_MAXCH][SG_CD];
#pragma unroll
  for (int c = 0; c < SG_MAXCH; c++) {
#pragma unroll
    for (int i = 0; i < 16; i++) Minv2[c][i] = L.cs[c].Minv[i];
#pragma unroll
    for (int d = 0; d < SG_CD; d++) { vc2[c][d] = L.cs[c].v[d]; asm2[c][d] = L.cs[c].qacc_smooth[d]; warm2[c][d] = L.cs[c].w[d]; }
  }
  double gsum[SG_MAXCH][SG_CD] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, push[SG_GEN_ROUNDS];
  int slk[SG_GEN_ROUNDS], tch = 0;
  StageRec2 rec[SG_GEN_ROUNDS];
#pragma unroll
  for (int k = 0; k < SG_GEN_ROUNDS; k++)
    if (lane + 64 * k < ng) rec[k] = L.stage[lane + 64 * k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < SG_GEN_ROUNDS; k++) {
    const int i = lane + 64 * k;
    push[k] = 0; slk[k] = -1;
    if (i < ng) {
      const SgGenPair gp = a.gpairs[rec[k].sl];
      const GenSide S1 = gen_side_of(gp.g1, H, a.elem + (size_t)SGE_BINVW * N), S2 = gen_side_of(gp.g2, H, a.elem + (size_t)SGE_BINVW * N);
      const int sl = S1.sl >= 0 ? S1.sl : S2.sl;
      double ax[3] = {0, 0, 0}, hint[3] = {0, 0, 0}, ve_ = 0, as_ = 0, we_ = 0, im = 0;
      if (sl >= 0) {
        ax[0] = EL(SGE_AX, sl); ax[1] = EL(SGE_AY, sl); ax[2] = EL(SGE_AZ, sl);
        ve_ = L.ve[sl]; as_ = L.asme[sl]; we_ = L.we[sl];
        im = 1.0 / (EL(SGE_MASS, sl) + EL(SGE_ARMATURE, sl));
      }
      if (gp.kind == SGP_PLANE_CAP) { hint[0] = EL(SGE_CX, sl); hint[1] = EL(SGE_CY, sl); hint[2] = EL(SGE_CZ, sl); }   // first tangent along the capsule
      GenContact c;
      gen_contact_build(c, rec[k], gp.kind == SGP_PLANE_CAP ? hint : nullptr, S1, S2, L.K, Minv2, vc2, asm2, warm2, ax, ve_, as_, we_, im, H);
      gen_contact_store(W.gcon + ((size_t)env * SG_GEN_MAXCON + i) * SG_GEN_W, c);
#pragma unroll
      for (int ch = 0; ch < SG_MAXCH; ch++)
#pragma unroll
        for (int d = 0; d < SG_CD; d++) gsum[ch][d] += c.Jf[ch][0][d] * c.f[0] + c.Jf[ch][1][d] * c.f[1] + c.Jf[ch][2][d] * c.f[2];
      slk[k] = sl;
      push[k] = c.invm * (c.Js[0] * c.f[0] + c.Js[1] * c.f[1] + c.Js[2] * c.f[2]);
      // touch bits: a finger box against an object geom (capsule or centre sphere)
      if ((gp.kind == SGP_CAP_BOX || gp.kind == SGP_SPH_BOX) && sgg_kind(gp.g2) == SGG_BOX) tch |= 1 << sgg_index(gp.g2);
    }
  }
  double* const gpush = (double*)L.stage;                    // [SG_GEN_MAXCON]
  int* const gsl = (int*)(gpush + SG_GEN_MAXCON);            // [SG_GEN_MAXCON]
#pragma unroll
  for (int k = 0; k < SG_GEN_ROUNDS; k++)
    if (lane + 64 * k < SG_GEN_MAXCON) { gpush[lane + 64 * k] = push[k]; gsl[lane + 64 * k] = slk[k]; }
#pragma unroll
  for (int ch = 0; ch < SG_MAXCH; ch++)
#pragma unroll
    for (int d = 0; d < SG_CD; d++) {
      const double x = wave_sum2(gsum[ch][d]);
      if (lane == 0) L.gg[ch * SG_CD + d] = x;
    }
  __syncthreads();
  for (int e = lane; e < N; e += 64) {   // per element: the pushes of its general contacts, in list order
    double acc = 0;
    for (int i = 0; i < ng; i++)
      if (gsl[i] == e) acc += gpush[i];
    L.gas[e] = acc;
  }
  int t = 0;
#pragma unroll
  for (int bb = 0; bb < SG_MAXCH * SG_CG; bb++)
    if (__ballot((tch >> bb) & 1)) t |= 1 << bb;
  *touch = t;
  int f2 = 0;
#pragma unroll
  for (int bit = 0; bit < 6; bit++)
    if (__ballot((fl >> bit) & 1)) f2 |= 1 << bit;
  *flags = f2;
  __syncthreads();
  return ng;
}

// warmstart cost of the general contacts (per-lane partial sums) for the current accelerations; zero != 0: set their forces to 0 instead
__device__ __noinline__ double sg_gen_cost(const SgPhaseArgs& a, const int env, const int ng, const double (*aF)[SG_CD], const double* as_lds,
                                           const int zero) {
  const int lane = threadIdx.x;
  double cp = 0;
  for (int i = lane; i < ng; i += 64) {
    double* rec = a.w.gcon + ((size_t)env * SG_GEN_MAXCON + i) * SG_GEN_W;
    if (zero) { rec[SG_GEN_F_OFF] = rec[SG_GEN_F_OFF + 1] = rec[SG_GEN_F_OFF + 2] = 0.0; continue; }
    GenContact c;
    gen_contact_load(c, rec);
    const double as_ = c.sl >= 0 ? as_lds[c.sl] : 0.0;
#pragma unroll
    for (int r = 0; r < 3; r++) {
      double Ja = c.Js[r] * as_;
#pragma unroll
      for (int ch = 0; ch < SG_MAXCH; ch++)
#pragma unroll
        for (int d = 0; d < SG_CD; d++) Ja += c.Jf[ch][r][d] * aF[ch][d];
      cp += c.f[r] * (0.5 * (Ja + c.R * c.f[r]) + c.b[r]);
    }
  }
  return cp;
}

// ------------------------------------------------------------------------------------------------
// phase kernel: [finish previous substep] [begin next substep]
// ------------------------------------------------------------------------------------------------
// GEN = false: the kernel every env runs.  An env in which a collision pair outside the fast path's two kinds is within reach is put on
// W.gen_list; the GEN = true instantiation -- launched after it with a small grid whose blocks stride over that list (normally empty: every block returns at once) -- then redoes
// the BEGIN part of exactly those envs on the general contact path (sg_gen_phase) and overwrites their exports.  The general path
// needs a stack (scratch memory) and every register; compiled into the main instantiation it doubled that kernel's time.
template <int R, int CPL, bool NB, bool GEN>  // NB: the model has neighbour equality rows (H.nnb > 0)
__device__ __forceinline__ void sg_phase_env(const SgPhaseArgs& a, const int env) {
  const int lane = threadIdx.x;
  if (env >= a.nenv) return;
  if (a.mask && !a.mask[env]) return;
  SG_T0();
  // the plan tables are read-only for the kernel's lifetime: through the constant address space a uniform index is a scalar load the
  // compiler may hoist and keep, not a vector load behind a full wait after every store
  const SG_CONSTAS SgPlanHeader& H = *(const SG_CONSTAS SgPlanHeader*)a.H;
  const int N = a.nelem, nv = a.nv, nu = a.nu, e0 = a.elem_dof0, nchain = a.nchain;
  const size_t S = 2 * (size_t)a.nenv;
  const double h = a.timestep;
  __shared__ Smem2<R, CPL, NB> Sm;
  const SG_CONSTAS double* const elemc = (const SG_CONSTAS double*)a.elem;
  [[maybe_unused]] const SG_CONSTAS int* const nbtabc = (const SG_CONSTAS int*)a.nbtab;
  auto EL = [&](int f, int e) { return elemc[(size_t)f * N + e]; };
  const int half = lane >> 5;
  const bool high = half != 0;
  const bool is_chain_lane = (lane & 31) == 0 && half < nchain;
  // the chains' model constants are read straight from the plan header (a few cached loads per wavefront); the chain stage itself,
  // which read them hundreds of times, runs in sg_chain_kernel
  const SG_CONSTAS SgChain& C = H.chain[half < nchain ? half : 0];
  ChainLds2& CS = Sm.cs[half];
  const SgWork& W = a.w;
  // the pair list and the per-slot slider pushes: LDS, or -- four element rounds, SG_PHASE_SLIM -- the env's slices of the work space (written
  // and read by this wavefront only, a workgroup barrier's fence between the two)
  [[maybe_unused]] unsigned short* const gpairs_env = W.gpairs16 + (size_t)env * SG_PAIRS_CAP(R);   // (uniform: scalar registers)
  [[maybe_unused]] double* const gcval_env = W.gcval + (size_t)env * SG_MAXCH * (32 * CPL);
  auto pair_at = [&](int i) -> unsigned short& {
    if constexpr (SG_PHASE_SLIM(R)) return gpairs_env[i];
    else return Sm.pairs[i];
  };
  auto cval_at = [&](int cc, int i) -> double& {
    if constexpr (SG_PHASE_SLIM(R)) return gcval_env[cc * (32 * CPL) + i];
    else return Sm.cval[cc * (32 * CPL) + i];
  };

  // status and pending are LOADED here and TESTED below, after the state loads have been issued: an early return on them would put
  // one memory round trip in front of every other load of the kernel (a wavefront lives ~30 us, a round trip costs 1 - 2)
  int status = W.status[env];  // sg_chain_kernel, which runs first, resets it at the start of a call
  const int pend = W.pending[env];

  double* gq = a.qpos + (size_t)env * nv;
  double* gv = a.qvel + (size_t)env * nv;
  double* gw = a.warm + (size_t)env * nv;
  const double kenv = a.kenv[env];
  const int kt0_masked = a.kmask_ten[a.t0_id];

  // ---------------- load state ----------------
  double qe[R], ve[R], we[R], ke[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    int e = r * 64 + lane;
    qe[r] = ve[r] = we[r] = ke[r] = 0;
    if (e < N) {
      if (a.do_reset) { qe[r] = EL(SGE_QPOS0, e); }
      else { qe[r] = gq[e0 + e]; ve[r] = gv[e0 + e]; we[r] = gw[e0 + e]; }
      ke[r] = a.kmask_jnt[e0 + e] ? kenv : EL(SGE_K0, e);
    }
  }
  // the loads of FINISH (solver result, smooth acceleration and force of the previous substep) and the import of the chain
  // hand-off record are issued here, together with the state: one memory latency instead of three in a row (the kernel waits for
  // memory two thirds of its time, profiles/r02)
  double ase[R], asme_p[R], fsm_p[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int e = r * 64 + lane;
    ase[r] = asme_p[r] = fsm_p[r] = 0;
    if (a.do_finish && e < N) {  // whether a substep is pending is tested below: the workspace words exist either way
      ase[r] = W.as[(size_t)env * N + e]; asme_p[r] = W.asme[(size_t)env * N + e];
      if (a.finish_integrate) fsm_p[r] = W.fsm[(size_t)env * N + e];
    }
  }
  if (a.do_begin && half < nchain) {  // the chain stage ran in sg_chain_kernel: import its hand-off record, the 32 lanes of a half sharing the loads
    const double* ch = W.chh + ((size_t)env * 2 + half) * SG_CHW;
    double* const kd = (double*)&Sm.K[half];
    double* const box = &Sm.boxp[half * SG_CG][0];
    double* const boxm = &Sm.boxm[half * SG_CG][0];
    double* const lim = CS.lim_sign;  // lim_sign, lim_R, lim_b, lim_f are contiguous, as SGH_LIMSIGN .. SGH_LIMF are
#pragma unroll
    for (int j0 = 0; j0 < SG_CHW; j0 += 32) {
      const int j = j0 + (lane & 31);
      const double v = j < SG_CHW ? ch[j] : 0.0;
      if (j >= SGH_QSM && j < SGH_QSM + SG_CD) CS.qacc_smooth[j - SGH_QSM] = v;
      else if (j >= SGH_K && j < SGH_K + 48) kd[j - SGH_K] = v;
      else if (j >= SGH_MINV && j < SGH_MINV + 16) CS.Minv[j - SGH_MINV] = v;
      else if (j >= SGH_V && j < SGH_V + SG_CD) CS.v[j - SGH_V] = v;
      else if (j >= SGH_W && j < SGH_W + SG_CD) CS.w[j - SGH_W] = v;
      else if (j >= SGH_BOX && j < SGH_BOX + 12 * SG_CG) {
        const int g = (j - SGH_BOX) / 12, k = (j - SGH_BOX) % 12;
        if (k < 3) box[3 * g + k] = v; else boxm[9 * g + k - 3] = v;
      }
      else if (j == SGH_LIMACT) CS.lim_active = (int)v;
      else if (j >= SGH_LIMSIGN && j < SGH_LIMSIGN + 4 * SG_MAXLIM) lim[j - SGH_LIMSIGN] = v;
    }
  }
  const bool dead = (status & (SG_FLAG_BADQPOS | SG_FLAG_BADQVEL | SG_FLAG_BADQACC)) != 0;
  if (dead) return;  // the env stopped integrating earlier in this call
  const bool fin = a.do_finish && pend;
  const double kt0 = kt0_masked ? kenv : H.t0_k0;
  __syncthreads();

  SG_T(0);
  // =============================== FINISH the previous substep ===============================
  if (fin) {
    int badacc = 0;
    double qacc_e[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      int e = r * 64 + lane;
      qacc_e[r] = 0;
      if (e < N) {
        qacc_e[r] = asme_p[r] + ase[r];
        if (isbad(qacc_e[r])) badacc = 1;
      }
    }
    const bool anybadacc = __ballot(badacc) != 0;
    if (anybadacc) {
      status |= SG_FLAG_BADQACC;
    } else {
      double qa[R], yc[R];  // yc = coef / (m + h d): the tendon's column of (M + h B)^-1 J'
      double Sp = 0;
#pragma unroll
      for (int r = 0; r < R; r++) {
        int e = r * 64 + lane;
        we[r] = qacc_e[r];
        qa[r] = yc[r] = 0;
        if (e < N && a.finish_integrate) {
          double m = EL(SGE_MASS, e) + EL(SGE_ARMATURE, e);
          double den = m + h * EL(SGE_DAMPING, e);
          qa[r] = (fsm_p[r] + m * ase[r]) / den;
          if (H.t0_implicit) { const double cf = EL(SGE_COEF, e); yc[r] = cf / den; Sp += cf * qa[r]; }
        }
      }
      if (H.t0_implicit && a.finish_integrate) {
        // deviation D5: (M + h B + h c J'J) qacc = f by Sherman-Morrison, qacc = x - y (h c J x) / (1 + h c J y), x = (M + h B)^-1 f,
        // y = (M + h B)^-1 J' (the sliders' block of M is diagonal, J is zero on the fingers); oracle sgo_step
        const double kk = h * H.t0_damping * wave_sum2(Sp) / (1.0 + H.t0_hcT);
#pragma unroll
        for (int r = 0; r < R; r++) qa[r] -= yc[r] * kk;
      }
#pragma unroll
      for (int r = 0; r < R; r++) {
        int e = r * 64 + lane;
        if (e < N && a.finish_integrate) {
          ve[r] += h * qa[r];
          qe[r] += h * ve[r];
        }
      }
    }
    if (lane == 0) W.pending[env] = 0;
  }
  __syncthreads();

  SG_T(1);
  // =============================== BEGIN the next substep ===============================
  int flags = 0;
  if (a.do_begin && !(status & SG_FLAG_BADQACC)) {
    int bad = 0;
#pragma unroll
    for (int r = 0; r < R; r++) bad |= (isbad(qe[r]) ? SG_FLAG_BADQPOS : 0) | (isbad(ve[r]) ? SG_FLAG_BADQVEL : 0);
    if (__ballot(bad != 0)) {
      flags |= (__ballot(bad & 1) ? 1 : 0) | (__ballot(bad & 2) ? 2 : 0);
    } else {
      // ---- chains: imported at the top of the kernel ----
      // ---- elements ----
      double invm[R], asme[R], coef[R];
      double L0p = 0, Ldp = 0;
#pragma unroll
      for (int r = 0; r < R; r++) {
        int e = r * 64 + lane;
        coef[r] = e < N ? EL(SGE_COEF, e) : 0.0;
        L0p += coef[r] * qe[r]; Ldp += coef[r] * ve[r];
      }
      const double L0 = wave_sum2(L0p), Ld = wave_sum2(Ldp);
      const double frc_t0 = -kt0 * (L0 - H.t0_lspring) - H.t0_damping * Ld;
      int unsupported = 0, ns0 = 0, ns1 = 0, touch = 0;
      bool special = false;
      {
        double cpos[R][3];
#pragma unroll
        for (int r = 0; r < R; r++) {
          int e = r * 64 + lane;
          invm[r] = asme[r] = 0;
          cpos[r][0] = cpos[r][1] = cpos[r][2] = 1e30;
          if (e < N) {
            double ax[3] = {EL(SGE_AX, e), EL(SGE_AY, e), EL(SGE_AZ, e)}, m = EL(SGE_MASS, e);
            double bias = -m * dot3(H.gravity, ax);
            double f = -ke[r] * (qe[r] - EL(SGE_SPRINGREF, e)) - EL(SGE_DAMPING, e) * ve[r] + coef[r] * frc_t0 - bias;
            invm[r] = 1.0 / (m + EL(SGE_ARMATURE, e));
            asme[r] = f * invm[r];
            W.fsm[(size_t)env * N + e] = f; W.asme[(size_t)env * N + e] = asme[r];
            double dq = qe[r] - EL(SGE_QPOS0, e);
            cpos[r][0] = EL(SGE_GX, e) + ax[0] * dq; cpos[r][1] = EL(SGE_GY, e) + ax[1] * dq; cpos[r][2] = EL(SGE_GZ, e) + ax[2] * dq;
            if (!(qe[r] > EL(SGE_QLO, e) && qe[r] < EL(SGE_QHI, e))) unsupported = 1;
            Sm.ve[e] = ve[r]; Sm.asme[e] = asme[r]; Sm.we[e] = we[r];
            *(unsigned int*)&Sm.eslot[e][0] = 0u;
            Sm.as[e] = qe[r];  // scratch until recompute_a: the dense narrowphase below reads other lanes' slider positions
          }
        }
        SG_T(2);
        __syncthreads();
        int overflow = 0;
        // ---- broadphase: (box, element) pairs that pass MuJoCo's bounding-sphere filter and the grown-box test, listed in
        //      contact order (chain, box, element; the object's centre sphere precedes the box's capsules).  Real loops: unrolled,
        //      the inlined narrowphase copies push the kernel far beyond the instruction cache
        int np = 0;
#pragma unroll 1
        for (int b = 0; b < nchain * SG_CG; b++) {
          const int c = b / SG_CG, g = b % SG_CG;
          const SgChain& Cc = H.chain[c];
          if (g >= Cc.ngeom) continue;
          double bp[3], bm[9], sz[3];
#pragma unroll
          for (int k = 0; k < 3; k++) { bp[k] = Sm.boxp[b][k]; sz[k] = Cc.g_size[g][k]; }
#pragma unroll
          for (int k = 0; k < 9; k++) bm[k] = Sm.boxm[b][k];
          const double rb = Cc.g_rbound[g];
          if (H.has_center) {
            double dif[3] = {bp[0] - H.center_pos[0], bp[1] - H.center_pos[1], bp[2] - H.center_pos[2]}, bound = H.center_radius + rb + H.con_margin;
            if (dot3(dif, dif) <= bound * bound) {
              if (lane == 0) pair_at(np) = (unsigned short)((b << 12) | SG_PAIR_CENTER);
              np++;
            }
          }
#pragma unroll
          for (int r = 0; r < R; r++) {
            const int e = r * 64 + lane;
            // bounding spheres, then the capsule centre against the box grown by the capsule's bounding radius in the box frame --
            // still conservative, so the contact set is unchanged
            double dif[3] = {cpos[r][0] - bp[0], cpos[r][1] - bp[1], cpos[r][2] - bp[2]}, bound = H.cap_rbound + rb + H.con_margin;
            bool near = e < N && dot3(dif, dif) <= bound * bound;
            if (near) {
              double lc[3], grow = H.cap_rbound + H.con_margin;
              mulmatT3(lc, bm, dif);
              near = fabs(lc[0]) <= sz[0] + grow && fabs(lc[1]) <= sz[1] + grow && fabs(lc[2]) <= sz[2] + grow;
              if (near) {  // third filter: the capsule's own extent along the box axes (|half segment| + radius) instead of its bounding radius
                const double cax[3] = {EL(SGE_CX, e), EL(SGE_CY, e), EL(SGE_CZ, e)}, rm = H.cap_radius + H.con_margin;
                double hb[3];
                mulmatT3(hb, bm, cax);
                near = fabs(lc[0]) <= sz[0] + rm + H.cap_hl * fabs(hb[0]) && fabs(lc[1]) <= sz[1] + rm + H.cap_hl * fabs(hb[1]) &&
                       fabs(lc[2]) <= sz[2] + rm + H.cap_hl * fabs(hb[2]);
              }
            }
            const unsigned long long m = __ballot(near);
            if (near) pair_at(np + lanes_below2(m)) = (unsigned short)((b << 12) | e);
            np += __popcll(m);
          }
        }
        __syncthreads();
        // ---- narrowphase over the dense pair list, 64 pairs per pass (every lane works; the per-box loop ran 8 passes at 10-30 %
        //      lane occupancy), then ordered compaction into the two finger streams
        int nsc[SG_MAXCH] = {0, 0};
#pragma unroll 1
        for (int p0 = 0; p0 < np; p0 += 64) {
          const bool have = p0 + lane < np;
          const int code = have ? (int)pair_at(p0 + lane) : 0, b = code >> 12, e = code & 0xFFF, c = b / SG_CG, g = b % SG_CG;
          const bool is_center = have && e == SG_PAIR_CENTER;
          ConRec r0, r1;
          bool v0 = false, v1 = false;
          double bp[3], bm[9], sz[3];
#pragma unroll
          for (int k = 0; k < 3; k++) { bp[k] = Sm.boxp[b][k]; sz[k] = H.chain[c].g_size[g][k]; }
#pragma unroll
          for (int k = 0; k < 9; k++) bm[k] = Sm.boxm[b][k];
          if (__ballot(is_center)) {
            if (is_center) v0 = sphere_box(H.center_pos, H.center_radius, bp, bm, sz, H.con_margin, r0) && r0.dist < H.con_margin;
          }
          if (have && !is_center) {
            const double dq = Sm.as[e] - EL(SGE_QPOS0, e);
            double cp[3] = {EL(SGE_GX, e) + EL(SGE_AX, e) * dq, EL(SGE_GY, e) + EL(SGE_AY, e) * dq, EL(SGE_GZ, e) + EL(SGE_AZ, e) * dq};
            double cax[3] = {EL(SGE_CX, e), EL(SGE_CY, e), EL(SGE_CZ, e)};
            int mk = capsule_box(cp, cax, H.cap_radius, H.cap_hl, bp, bm, sz, H.con_margin, r0, r1);
            v0 = (mk & 1) && r0.dist < H.con_margin;
            v1 = (mk & 2) && r1.dist < H.con_margin;
          }
          const int n = (int)v0 + (int)v1;
#pragma unroll
          for (int cc = 0; cc < SG_MAXCH; cc++) {
            const bool mine = have && c == cc;
            const unsigned long long m1 = __ballot(mine && n >= 1), m2 = __ballot(mine && n >= 2);
            const int base = nsc[cc] + lanes_below2(m1) + lanes_below2(m2);
            if (mine && v0 && base < 32 * CPL) {
              StageRec2& s = Sm.stage[cc][base];
              s.dist = r0.dist; s.sl = is_center ? -1 : e; s.box = g;
              for (int q = 0; q < 3; q++) { s.pos[q] = r0.pos[q]; s.n[q] = r0.n[q]; }
            }
            if (mine && v1 && base + (int)v0 < 32 * CPL) {
              StageRec2& s = Sm.stage[cc][base + (int)v0];
              s.dist = r1.dist; s.sl = e; s.box = g;
              for (int q = 0; q < 3; q++) { s.pos[q] = r1.pos[q]; s.n[q] = r1.n[q]; }
            }
            if (mine && n > 0 && !is_center) {
              const int room = 32 * CPL - base, nst = n < room ? n : (room > 0 ? room : 0);
              static_assert(32 * CPL <= 64, "a contact slot index must fit 6 bits");
              Sm.eslot[e][b] = nst ? (unsigned char)(base | (nst << 6)) : (unsigned char)0;
            }
            nsc[cc] += __popcll(m1) + __popcll(m2);
            if (nsc[cc] > 32 * CPL) { nsc[cc] = 32 * CPL; overflow = 1; }
          }
#pragma unroll
          for (int bb = 0; bb < SG_MAXCH * SG_CG; bb++)
            if (__ballot(n > 0 && b == bb)) touch |= 1 << bb;
        }
        ns0 = nsc[0]; ns1 = nsc[1];
#ifdef SG_SECTION_PROF
        if (lane == 0) { atomicAdd(&a.w.secprof[30], (unsigned long long)np); atomicAdd(&a.w.secprof[31], (unsigned long long)((np + 63) / 64)); }
#endif
        if (overflow) flags |= SG_FLAG_CONTACTFULL;
      }
      SG_T(3);
      {  // envelope checks (same pairs as the fused kernel).  Finger box against static box (lanes 0 .. npairs-1) and finger box
         // against finger box (lanes 32 .. 35) go through ONE separating-axis test: each lane sets up its pair, then all of
         // them run the 15 axes together (two copies of the test, one per kind of pair, ran one after the other before)
        int nb = nchain * SG_CG, npairs = nb * H.nstatic;
        const double *p1 = nullptr, *R1 = nullptr, *s1 = nullptr, *p2 = nullptr, *R2 = nullptr, *s2 = nullptr;
        double bd = 0;
        bool pair = false;
        if (lane < npairs) {
          int b = lane / H.nstatic, s = lane % H.nstatic, c = b / SG_CG, g = b % SG_CG;
          if (g < H.chain[c].ngeom) {
            pair = true;
            p1 = Sm.boxp[b]; R1 = Sm.boxm[b]; s1 = H.chain[c].g_size[g];
            p2 = H.st_pos[s]; R2 = H.st_mat[s]; s2 = H.st_size[s];
            bd = H.chain[c].g_rbound[g] + H.st_rbound[s];
          }
        } else if (lane >= 32 && lane < 32 + SG_CG * SG_CG && nchain == 2) {
          int g = (lane - 32) / SG_CG, g2 = (lane - 32) % SG_CG;
          if (g < H.chain[0].ngeom && g2 < H.chain[1].ngeom) {
            int b = g, b2 = SG_CG + g2;
            pair = true;
            p1 = Sm.boxp[b]; R1 = Sm.boxm[b]; s1 = H.chain[0].g_size[g];
            p2 = Sm.boxp[b2]; R2 = Sm.boxm[b2]; s2 = H.chain[1].g_size[g2];
            bd = H.chain[0].g_rbound[g] + H.chain[1].g_rbound[g2];
          }
        }
        if (pair) {
          double dif[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
          if (dot3(dif, dif) <= bd * bd && box_box_overlap(p1, R1, s1, p2, R2, s2, 0)) unsupported = 1;
        }
        if (lane >= 48 && lane < 48 + SG_MAXCH * SG_CG && H.has_plane) {
          int b = lane - 48, c = b / SG_CG, g = b % SG_CG;
          if (c < nchain && g < H.chain[c].ngeom) {
            double dif[3] = {Sm.boxp[b][0] - H.plane_pos[0], Sm.boxp[b][1] - H.plane_pos[1], Sm.boxp[b][2] - H.plane_pos[2]}, ext = 0;
            for (int k = 0; k < 3; k++)
              ext += H.chain[c].g_size[g][k] * fabs(H.plane_normal[0] * Sm.boxm[b][k] + H.plane_normal[1] * Sm.boxm[b][3 + k] + H.plane_normal[2] * Sm.boxm[b][6 + k]);
            if (dot3(dif, H.plane_normal) - ext <= 0) unsupported = 1;
          }
        }
        special = __ballot(unsupported) != 0;
      }
      SG_T(4);
      __syncthreads();
      // a pair outside the fast path's two kinds is within reach: this substep's contacts are rebuilt as ONE ordered list over all
      // candidate pairs (sg_gen_phase; the rows pipeline's solver sweeps it as one stream).  The other pipelines flag the env.
      int ngen = 0;
      [[maybe_unused]] GenLds GL;
      if (special) {
        if (!a.rowlayout) flags |= SG_FLAG_UNSUPPORTED_PAIR;
        else if constexpr (!GEN) {
          // hand the env to the general pass (sg_phase_kernel<.., true>).  The list has room for the whole batch (until r04: 256 envs,
          // the rest flagged -- which 256 depended on the order of arrival): every env that needs the path gets it, whatever the order
          if (lane == 0) a.w.gen_list[atomicAdd(a.w.gen_count, 1)] = env;
        } else {
          // the contact staging area in this mode: [0, 8 M) doubles the contact list (M = SG_GEN_MAXCON records of 8 doubles), then 96
          // doubles of box - box work space and the two chains' J' f sums; once the rows are built the list's place is taken by
          // the per-contact pushes [M], slider indices [M ints] and the per-element sums [R * 64]
          static_assert(sizeof(StageRec2) == 64 && sizeof(Sm.stage) >= 8 * (8 * SG_GEN_MAXCON + 96 + SG_MAXCH * SG_CD) &&
                        8 * SG_GEN_MAXCON >= SG_GEN_MAXCON + SG_GEN_MAXCON / 2 + R * 64, "the general path's lists live in the contact staging area");
          GL.stage = &Sm.stage[0][0];
          GL.gas = (double*)&Sm.stage[0][0] + SG_GEN_MAXCON + SG_GEN_MAXCON / 2;
          GL.tmp = (double*)&Sm.stage[0][0] + 8 * SG_GEN_MAXCON; GL.gg = GL.tmp + 96;
          GL.boxp = Sm.boxp; GL.boxm = Sm.boxm; GL.K = Sm.K; GL.cs = Sm.cs;
          GL.qe = Sm.as; GL.ve = Sm.ve; GL.asme = Sm.asme; GL.we = Sm.we;
          int gfl = 0, gtouch = 0;
          ngen = sg_gen_phase(a, *a.H, env, GL, &gfl, &gtouch);
          flags |= gfl;
          touch = gtouch;
          ns0 = ns1 = 0;                       // no contact stays on the per-finger streams
#pragma unroll
          for (int r = 0; r < R; r++) {
            const int e = r * 64 + lane;
            if (e < N) *(unsigned int*)&Sm.eslot[e][0] = 0u;
          }
          __syncthreads();
        }
      }
      int shared_slider = 0;
#pragma unroll
      for (int r = 0; r < R; r++) {
        int e = r * 64 + lane;
        // (a slider under both fingers: both chains hold a contact slot of this element -- eslot[e][2 c .. 2 c + 1], SG_CG = 2 boxes a chain)
        static_assert(SG_MAXCH == 2 && SG_CG == 2, "the shared-slider test reads an element's four slot bytes as two halves");
        if (e < N && ((const unsigned short*)&Sm.eslot[e][0])[0] != 0 && ((const unsigned short*)&Sm.eslot[e][0])[1] != 0) shared_slider = 1;
      }
      shared_slider = __ballot(shared_slider) != 0;

      // ---- contact rows: built one slot at a time and exported at once; only a 7-double summary per slot stays
      //      in registers for the warmstart test (g = Jf' f, Js.f, invm, f.(R f/2 + b), slider index)
      const int myn = high ? ns1 : ns0;
      const size_t st = 2 * (size_t)env + half;
      const int nwb = (a.nenv + SG_EPW - 1) / SG_EPW;
      double cg[CPL][SG_CD], cjsf[CPL], cinvm[CPL], ccost0[CPL];
      int csl_[CPL];
#pragma unroll
      for (int k = 0; k < CPL; k++) {
        int i = (lane & 31) + 32 * k;
        csl_[k] = -1; cjsf[k] = cinvm[k] = ccost0[k] = 0;
#pragma unroll
        for (int d = 0; d < SG_CD; d++) cg[k][d] = 0;
        double2 fp[4][SG_RK / 2];  // my contact's three rows + the quad's fourth lane in the solver's field pairs (row layout only)
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
          for (int pr = 0; pr < SG_RK / 2; pr++) fp[r][pr] = make_double2(0.0, 0.0);
        if (i < myn) {
          Contact c;
          const StageRec2& sr = Sm.stage[half][i];
          ConRec rec;
          rec.dist = sr.dist;
          for (int q = 0; q < 3; q++) { rec.pos[q] = sr.pos[q]; rec.n[q] = sr.n[q]; }
          int sl = sr.sl, g = sr.box, bi = C.g_body[g], nd = chain_ndof_of_body(bi);
          double ax[3] = {0, 0, 0}, ve_ = 0, as_ = 0, we_ = 0, im = 0, bw = 0;
          if (sl >= 0) {
            ax[0] = EL(SGE_AX, sl); ax[1] = EL(SGE_AY, sl); ax[2] = EL(SGE_AZ, sl);
            ve_ = Sm.ve[sl]; as_ = Sm.asme[sl]; we_ = Sm.we[sl];
            im = 1.0 / (EL(SGE_MASS, sl) + EL(SGE_ARMATURE, sl)); bw = EL(SGE_BINVW, sl);
          }
          contact_build(c, rec, Sm.K[half], nd, CS.Minv, CS.v, CS.qacc_smooth, CS.w, C.b_invw_tran[bi], sl, ax, ve_, as_, we_, im, bw, *a.H);
          csl_[k] = sl; cinvm[k] = c.invm;
          cjsf[k] = c.Js[0] * c.f[0] + c.Js[1] * c.f[1] + c.Js[2] * c.f[2];
          cval_at(half, i) = c.invm * cjsf[k];
          ccost0[k] = c.f[0] * (0.5 * c.R * c.f[0] + c.b[0]) + c.f[1] * (0.5 * c.R * c.f[1] + c.b[1]) + c.f[2] * (0.5 * c.R * c.f[2] + c.b[2]);
#pragma unroll
          for (int d = 0; d < SG_CD; d++) cg[k][d] = c.Jf[0][d] * c.f[0] + c.Jf[1][d] * c.f[1] + c.Jf[2][d] * c.f[2];
          SG_T(24);
          if (a.rowlayout) {
            const double S11 = c.A[3] * H.con_mu[0] * H.con_mu[0], S22 = c.A[5] * H.con_mu[1] * H.con_mu[1], S12 = c.A[4] * H.con_mu[0] * H.con_mu[1];
            const double det = S11 * S22 - S12 * S12, di = det < 1e-10 ? 0.0 : sg_div(1.0, det);
            const double P11 = S22 * di, P22 = S11 * di, P12 = -S12 * di;
            // eigen-decomposition of the (friction-scaled) block S = Q diag(e1, e2) Q', Q = [[cs, sn], [-sn, cs]] (one Jacobi rotation):
            // constant over the solve, so mju_QCQP2's Newton iteration in the solver runs in these coordinates (sg_pgs_rows_kernel)
            double ecs = 1.0, esn = 0.0, ee1 = S11, ee2 = S22;
            if (fabs(S12) > 1e-300) {
              const double tau = (S22 - S11) / (2.0 * S12), tt = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
              ecs = 1.0 / sqrt(1.0 + tt * tt); esn = tt * ecs;
              ee1 = S11 - tt * S12; ee2 = S22 + tt * S12;
            }
            const double Afull[3][3] = {{c.A[0], c.A[1], c.A[2]}, {c.A[1], c.A[3], c.A[4]}, {c.A[2], c.A[4], c.A[5]}};
            double Wm[3][SG_CD];  // W_r = M^-1 J_F[r]': lane q of the quad keeps (W_0[q], W_1[q], W_2[q]), its column of the finger update
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
              for (int d = 0; d < SG_CD; d++) {
                double sw = 0;
#pragma unroll
                for (int e = 0; e < SG_CD; e++) sw += c.Jf[r][e] * CS.Minv[4 * e + d];
                Wm[r][d] = sw;
              }
#pragma unroll
            for (int r = 0; r < 4; r++) {
              double fld[SG_RK];
#pragma unroll
              for (int q = 0; q < SG_RK; q++) fld[q] = 0.0;
              if (r < 3) {
#pragma unroll
                for (int d = 0; d < SG_CD; d++) fld[d] = c.Jf[r][d];
                fld[4] = c.Js[r]; fld[5] = c.b[r]; fld[6] = c.f[r];
                fld[7] = Afull[r][0] * c.f[0] + Afull[r][1] * c.f[1] + Afull[r][2] * c.f[2];
                fld[8] = Afull[r][0]; fld[9] = Afull[r][1]; fld[10] = Afull[r][2];
                fld[11] = c.invm * c.Js[r];
                fld[15] = c.R;
              } else {  // the fourth lane carries what the three rows share (R is replicated on the row lanes: no broadcast)
                fld[0] = P11; fld[1] = P12; fld[2] = P22; fld[4] = __hiloint2double(0, sl);
                fld[8] = ee1; fld[9] = ee2; fld[10] = ecs; fld[11] = esn;
              }
              fld[12] = Wm[0][r]; fld[13] = Wm[1][r]; fld[14] = Wm[2][r];
#pragma unroll
              for (int pr = 0; pr < SG_RK / 2; pr++) fp[r][pr] = make_double2(fld[2 * pr], fld[2 * pr + 1]);
            }
          }
          if (!a.rowlayout) {
          double* ro = W.crec + SG_REC_INDEX(i, env / SG_EPW, 0, 2 * (env % SG_EPW) + half, nwb);
#pragma unroll
          for (int r = 0; r < 3; r++)
#pragma unroll
            for (int d = 0; d < SG_CD; d++) ro[(4 * r + d) * SG_SPW] = c.Jf[r][d];
#pragma unroll
          for (int r = 0; r < 3; r++) ro[(12 + r) * SG_SPW] = c.Js[r];
#pragma unroll
          for (int q = 0; q < 6; q++) ro[(15 + q) * SG_SPW] = c.A[q];
#pragma unroll
          for (int r = 0; r < 3; r++) ro[(21 + r) * SG_SPW] = c.b[r];
          ro[24 * SG_SPW] = c.R;
          ro[25 * SG_SPW] = c.invm;
#pragma unroll
          for (int r = 0; r < 3; r++) ro[(26 + r) * SG_SPW] = c.f[r];
          ((int*)(ro + 29 * SG_SPW))[0] = sl;
          }
        }
        // ---- export in row layout, transposed through LDS so that every store instruction writes whole 128-byte lines: the line
        //      (slot, pair) of this env is the 8 lanes (finger, row) x 16 B, held by two contact lanes (fingers 0 and 1 of slot i).
        //      The staging area is the part of Sm.stage this pass has consumed (slots 32 k .. 32 k + 31 of both fingers: 2 x 2 KB).
        const int nk = (ns0 > ns1 ? ns0 : ns1) - 32 * k;  // slots of this pass that any finger uses
        if (a.rowlayout && nk > 0) {
          const int nwb8 = (a.nenv + 7) / 8;
          const int li = lane & 31;
          auto X = [&](int sl_, int L) -> double2* {                                   // entry (slot, lane-of-8) for the reads
            return (double2*)&Sm.stage[sl_ >> 4][32 * k] + ((sl_ & 15) * 8 + L);
          };
          __syncthreads();  // every lane has copied its stage record
          // my entries: (slot li, lane-of-8 4 half + r); slot li lives in piece li >> 4
          double2* const e0 = (double2*)&Sm.stage[li >> 4][32 * k] + ((li & 15) * 8 + 4 * half);
#pragma unroll
          for (int pr = 0; pr < SG_RK / 2; pr++) {
            e0[0] = fp[0][pr]; e0[1] = fp[1][pr]; e0[2] = fp[2][pr]; e0[3] = fp[3][pr];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; j++) {
              if (8 * j >= nk) break;  // 8 slots per store instruction
              const int e = j * 64 + lane, s_ = e >> 3, L = e & 7;
              if (s_ < nk) *(double2*)(W.crow + SG_ROW_INDEX(32 * k + s_, env >> 3, 2 * pr, 8 * (env & 7) + L, nwb8)) = *X(s_, L);
            }
            __syncthreads();
          }
        }
      }
      SG_T(5);
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
      // ---- neighbour rows (slider e = slider e2, J = +1 / -1): built by the lane of their first element, up to three each.
      //      b and R go straight to the workspace; the warmstart force lives in LDS (Sm.nbf, by row id) for the gathers below
      const int nnb = NB ? H.nnb : 0;
      constexpr int ND = NB ? 3 : 0;  // neighbour rows per element (loops over d vanish without them)
      int nbe2[R][3], nbid[R][3];
      double nbc0[R][3];  // f (R f / 2 + b)
      // the rows' warmstart forces: in registers on the lane that builds them, and in the work space (W.nbf, where the solver takes them
      // from) for the one cross-lane use -- an element gathering the rows that have it as SECOND joint.  Until r04 an LDS array by
      // row id (6 KB for the ball): without it the kernel's LDS block is 23 KB instead of 29 (ball) / 20 instead of 25 (cylinder):
      // 7 / 8 workgroups per CU instead of 5 / 6, the 4096 wavefronts in 2.3 / 2 rounds instead of 3.2 / 2.7
      [[maybe_unused]] double nbff[R][3];
      [[maybe_unused]] double* const gnbf = W.nbf + (size_t)env * 3 * N;
#pragma unroll
      for (int r = 0; r < R; r++)
#pragma unroll
        for (int d = 0; d < 3; d++) { nbe2[r][d] = -1; nbid[r][d] = -1; nbc0[r][d] = 0; nbff[r][d] = 0; }
      if constexpr (NB) {
        // The solver's step factors c = (1/m) / (A + R) of the block [fix_e, e's neighbour rows] (A = 1/m for the fix row, 2/m for a
        // neighbour row: equal masses, sg_plan_build) go out in the SOLVER's order: four consecutive doubles at the place of the
        // block's lane pair in its round of the wavefront's stream (W.cst) -- the solver reads them as one coalesced 16-byte load per
        // lane and round instead of keeping them in LDS, which is what lets four of its workgroups share a CU for the ball and the
        // cylinder (218 / 192 elements).  Rows that do not exist hold 0: their update is a no-op.
        const double im0 = 1.0 / (EL(SGE_MASS, 0) + EL(SGE_ARMATURE, 0));
        const SG_CONSTAS int* const cposc = (const SG_CONSTAS int*)a.cpos;
        const bool wantc = a.rowlayout && a.cst_rounds;   // (uniform; a model whose solver keeps the factors in LDS computes them there)
#pragma unroll
        for (int r = 0; r < R; r++) {
          const int e = r * 64 + lane;
          if (e < N) {
            double c4[4] = {0.0, 0.0, 0.0, 0.0};
            if (wantc) c4[0] = sg_div(im0, im0 + eqR[r]);
#pragma unroll
            for (int d = 0; d < 3; d++) {
              const int e2 = nbtabc[d * N + e];
              if (e2 >= 0) {
                const int id = nbtabc[(3 + d) * N + e];
                const double pos = (qe[r] - EL(SGE_QPOS0, e)) - (Sm.as[e2] - EL(SGE_QPOS0, e2)), imp = impedance(H.eqj_solimp, pos, 0);
                const double Rr = fmax(SG_MINVAL, (1 - imp) / imp * (EL(SGE_INVW, e) + EL(SGE_INVW, e2)));
                const double aref = -H.eqj_B * (ve[r] - Sm.ve[e2]) - H.eqj_K * imp * pos;
                const double bb = (asme[r] - Sm.asme[e2]) - aref, ff = -((we[r] - Sm.we[e2]) - aref) / Rr;
                nbe2[r][d] = e2; nbid[r][d] = id; nbc0[r][d] = ff * (0.5 * Rr * ff + bb);
                nbff[r][d] = ff; gnbf[id] = ff;
                W.nbb[(size_t)env * 3 * N + id] = bb; W.nbR[(size_t)env * 3 * N + id] = Rr;
                if (wantc) c4[1 + d] = sg_div(im0, 2.0 * im0 + Rr);
              }
            }
            if (wantc) {
              double* const dst = W.cst + SG_CST_INDEX(env >> 2, 0, 16 * (env & 3), a.cst_rounds) + cposc[e];
              *(double2*)dst = make_double2(c4[0], c4[1]);
              *(double2*)(dst + 2) = make_double2(c4[2], c4[3]);
            }
          }
        }
      }
      const double tpos = L0 - H.t0_L0, timp = impedance(H.eqt_solimp, tpos, 0), tR = fmax(SG_MINVAL, (1 - timp) / timp * H.eqt_invw);
      const double taref = -H.eqt_B * Ld - H.eqt_K * timp * tpos;
      const double tb = wave_sum2(tbp) - taref, tjar = wave_sum2(tjp) - taref, tA = wave_sum2(tAp) + tR;
      double tf = -tjar / tR;
      const int nmaxs = ns0 > ns1 ? ns0 : ns1;

      double aF[SG_CD];
      auto recompute_a = [&]() {
        __syncthreads();
        // slider accelerations M^-1 J' f: every element lane adds the pushes of its own contacts, finger 0's slots then finger 1's,
        // ascending -- the order of the solver's stream sweep (a serial loop over all contact slots used to do this)
#pragma unroll
        for (int r = 0; r < R; r++) {
          int e = r * 64 + lane;
          if (e < N) {
            double fe = eqf[r] + coef[r] * tf;
            if constexpr (NB) {  // + its own neighbour rows (J = +1), - the rows that have it as second joint (J = -1), in row order per side
#pragma unroll
              for (int d = 0; d < 3; d++) if (nbid[r][d] >= 0) fe += nbff[r][d];
              double fin[3];   // (the three loads go out together: one round trip to L2, behind the barrier that made the stores visible)
#pragma unroll
              for (int d = 0; d < 3; d++) { const int ii = nbtabc[(6 + d) * N + e]; fin[d] = ii >= 0 ? gnbf[ii] : 0.0; }
#pragma unroll
              for (int d = 0; d < 3; d++) fe -= fin[d];
            }
            double as_ = invm[r] * fe;
#pragma unroll
            for (int cb = 0; cb < SG_MAXCH * SG_CG; cb++) {
              const int u = Sm.eslot[e][cb], i0 = u & 0x3F, nst = u >> 6;
              if (nst >= 1) as_ += cval_at(cb / SG_CG, i0);
              if (nst >= 2) as_ += cval_at(cb / SG_CG, i0 + 1);
            }
            if constexpr (GEN)
              if (ngen) as_ += GL.gas[e];   // general contact path: the pushes of the env's one contact list on this slider
            Sm.as[e] = as_;
          }
        }
        __syncthreads();
        double g[SG_CD] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < CPL; k++)
#pragma unroll
          for (int d = 0; d < SG_CD; d++) g[d] += cg[k][d];
        if (is_chain_lane) {
          const int la = CS.lim_active;
#pragma unroll
          for (int k = 0; k < SG_MAXLIM; k++)
            if (la >> k & 1) g[k / 2] += CS.lim_sign[k] * CS.lim_f[k];
          if constexpr (GEN)
            if (ngen) {
#pragma unroll
              for (int d = 0; d < SG_CD; d++) g[d] += GL.gg[half * SG_CD + d];
            }
        }
#pragma unroll
        for (int d = 0; d < SG_CD; d++) {
          double x = g[d];
#pragma unroll
          for (int o = 16; o > 0; o >>= 1) x += __shfl_xor(x, o);
          g[d] = x;
        }
#pragma unroll
        for (int a2 = 0; a2 < SG_CD; a2++) {
          double s2 = 0;
#pragma unroll
          for (int b2 = 0; b2 < SG_CD; b2++) s2 += CS.Minv[4 * a2 + b2] * g[b2];
          aF[a2] = s2;
        }
        __syncthreads();
      };
      recompute_a();
      SG_T(6);
      {
        double cp = 0, tJap = 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
          int e = r * 64 + lane;
          if (e < N) {
            double ae = Sm.as[e]; cp += eqf[r] * (0.5 * (ae + eqR[r] * eqf[r]) + eqb[r]); tJap += coef[r] * ae;
#pragma unroll
            for (int d = 0; d < ND; d++)
              if (nbid[r][d] >= 0) cp += 0.5 * nbff[r][d] * (ae - Sm.as[nbe2[r][d]]) + nbc0[r][d];
          }
        }
        double tJa = wave_sum2(tJap);
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
            // sum_r f_r (J_r a / 2 + R f_r / 2 + b_r) = (g.aF + (Js.f) a_s) / 2 + f.(R f / 2 + b)
            double as_ = csl_[k] >= 0 ? Sm.as[csl_[k]] : 0.0, ga = 0;
#pragma unroll
            for (int d = 0; d < SG_CD; d++) ga += cg[k][d] * aF[d];
            cp += 0.5 * (ga + cjsf[k] * as_) + ccost0[k];
          }
        [[maybe_unused]] double aF2[SG_MAXCH][SG_CD];   // both chains' accelerations on every lane (general contact path only)
        if constexpr (GEN)
          if (ngen) {
#pragma unroll
            for (int d = 0; d < SG_CD; d++) {
              const double other = __shfl_xor(aF[d], 32);
              aF2[0][d] = high ? other : aF[d]; aF2[1][d] = high ? aF[d] : other;
            }
            cp += sg_gen_cost(a, env, ngen, aF2, Sm.as, 0);
          }
        double cost = wave_sum2(cp);
        if (cost > 0) {
          if constexpr (GEN)
            if (ngen) {
              sg_gen_cost(a, env, ngen, aF2, Sm.as, 1);
              for (int e = lane; e < N; e += 64) GL.gas[e] = 0.0;
              if (lane < SG_MAXCH * SG_CD) GL.gg[lane] = 0.0;
            }
#pragma unroll
          for (int r = 0; r < R; r++) {
            eqf[r] = 0;
#pragma unroll
            for (int d = 0; d < ND; d++) if (nbid[r][d] >= 0) { nbff[r][d] = 0.0; gnbf[nbid[r][d]] = 0.0; }
          }
          tf = 0;
          if (is_chain_lane) {
#pragma unroll
            for (int k = 0; k < SG_MAXLIM; k++) CS.lim_f[k] = 0;
          }
#pragma unroll
          for (int k = 0; k < CPL; k++) {
            int i = (lane & 31) + 32 * k;
            cjsf[k] = 0;
#pragma unroll
            for (int d = 0; d < SG_CD; d++) cg[k][d] = 0;
            if (i < myn) {
              cval_at(half, i) = 0.0;
              if (!a.rowlayout) {
              double* ro = W.crec + SG_REC_INDEX(i, env / SG_EPW, 0, 2 * (env % SG_EPW) + half, nwb);
#pragma unroll
              for (int r = 0; r < 3; r++) ro[(26 + r) * SG_SPW] = 0.0;
              }
              if (a.rowlayout) {
#pragma unroll
                for (int r = 0; r < 3; r++)  // f and A f (one pair)
                  *(double2*)(W.crow + SG_ROW_INDEX(i, env >> 3, 6, 8 * (env & 7) + 4 * half + r, (a.nenv + 7) / 8)) = make_double2(0.0, 0.0);
              }
            }
          }
          __syncthreads();
          recompute_a();
        }
      }
      SG_T(7);
      // ---- export the rest of the constraint problem ----
#pragma unroll
      for (int r = 0; r < R; r++) {
        int e = r * 64 + lane;
        if (e < N) {
          size_t o = (size_t)env * N + e;
          W.as[o] = Sm.as[e]; W.eqf[o] = eqf[r]; W.eqb[o] = eqb[r]; W.eqR[o] = eqR[r];
          // (W.nbf holds the neighbour rows' forces already: written where they were built, zeroed with the rest when the warmstart lost)
        }
      }
      if (is_chain_lane) {
        W.ns[st] = myn < SG_CAP ? myn : SG_CAP;
        W.lim_active[st] = CS.lim_active;
      }
      if (half < nchain) {  // the 32 lanes of a half write its chain's 4 x SG_MAXLIM limit values (contiguous in CS) and M^-1 J' f
        static_assert(4 * SG_MAXLIM == 32, "one limit value per lane of a half");
        const int i = lane & 31;
        W.lim[(size_t)i * S + st] = (&CS.lim_sign[0])[i];
        if (i < SG_CD) W.saF[(size_t)i * S + st] = i == 0 ? aF[0] : (i == 1 ? aF[1] : (i == 2 ? aF[2] : aF[3]));
      }
      if (lane == 0) {
        W.envh[(size_t)0 * a.nenv + env] = tb; W.envh[(size_t)1 * a.nenv + env] = tR;
        W.envh[(size_t)2 * a.nenv + env] = tA; W.envh[(size_t)3 * a.nenv + env] = tf;
        W.shared[env] = shared_slider;
        W.pending[env] = 1;
        W.ncon[env] = ns0 + ns1 + ngen;
        W.gen[env] = ngen;
        W.nefc[env] = N + nnb + 1 + 3 * (ns0 + ns1 + ngen) + __popc(Sm.cs[0].lim_active) + (nchain > 1 ? __popc(Sm.cs[1].lim_active) : 0);
        W.touch[env] = touch;
      }
    }
  }

  SG_T(8);
  // ---------------- store state ----------------
  __syncthreads();
#pragma unroll
  for (int r = 0; r < R; r++) {
    int e = r * 64 + lane;
    if (e < N) { gq[e0 + e] = qe[r]; gv[e0 + e] = ve[r]; gw[e0 + e] = we[r]; }
  }
  if (lane == 0) {
    if (status | flags) atomicOr(&W.status[env], status | flags);
  }
  SG_T(9);
  SG_TEND();
}

#ifndef SG_X_PHASE_OCC
#define SG_X_PHASE_OCC 2
#endif
#ifndef SG_X_CPL
#define SG_X_CPL 2
#endif
template <int R, int CPL, bool NB, bool GEN = false>
__global__ __launch_bounds__(64, SG_X_PHASE_OCC) void sg_phase_kernel(SgPhaseArgs a) {
  if constexpr (GEN) {
    // the general pass: a SMALL grid (SG_GEN_GRID blocks; the kernel needs scratch memory and every register, and an empty block of it
    // is not free: one block per env of the batch cost the default benchmark 5 %) whose blocks stride over the list -- normally empty --
    // so that the pass takes however many envs the main pass has listed, the whole batch if need be
    const int cnt = a.w.gen_count[0];            // uniform
    a.do_finish = 0; a.do_reset = 0; a.sens = nullptr;   // BEGIN only: the main pass has finished the previous substep and stored the state
    for (int i = blockIdx.x; i < cnt && i < a.nenv; i += gridDim.x) {
      sg_phase_env<R, CPL, NB, true>(a, a.w.gen_list[i]);
      __syncthreads();   // (the next env re-uses the block's LDS)
    }
  } else {
    sg_phase_env<R, CPL, NB, false>(a, blockIdx.x);
  }
}

// ------------------------------------------------------------------------------------------------
// chain kernel: ONE LANE PER FINGER CHAIN (64 chains per wavefront, all with the same chain index).  The chain stage is a few thousand strictly serial
// instructions; inside the phase kernel it ran on 2 of 64 lanes of every env's wavefront, here 64 chains share one
// instruction stream.  FINISH: qacc of the chain, its sensors, warmstart, integration.  BEGIN: kinematics, mass matrix,
// bias, tendon/actuator, limit rows, box poses -> hand-off record (enum SGH_*) for the phase and PGS kernels.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void sg_chain_kernel(SgPhaseArgs a) {
  // wavefront 2 b + c holds chain c of envs 64 b .. 64 b + 63: the chain index is uniform over the wavefront, so the chain's model
  // constants (SgChain, ~230 doubles read all over the stage) are scalar loads / SGPR operands instead of per-lane vector loads
  const int lane = threadIdx.x, c = blockIdx.x & 1, env = (int)(blockIdx.x >> 1) * 64 + lane;
  if (blockIdx.x == 0 && lane == 0) a.w.gen_count[0] = 0;   // the phase kernel of this substep refills the general pass's list
  SG_T0();
  const SG_CONSTAS SgPlanHeader& H = *(const SG_CONSTAS SgPlanHeader*)a.H;
  const int nv = a.nv, nu = a.nu;
  const size_t S = 2 * (size_t)a.nenv;
  if (env >= a.nenv) return;
  const size_t st = 2 * (size_t)env + c;
  if (a.mask && !a.mask[env]) return;
  SgWork& W = a.w;
  if (a.first && c == 0) { W.status[env] = 0; if (!a.do_finish) W.pending[env] = 0; }
  if (c >= a.nchain) return;
  // loaded here, tested after the state loads have been issued (an early return would put a memory round trip in front of them)
  const int status = a.first ? 0 : W.status[env];
  const int pend = a.do_finish ? W.pending[env] : 0;
  const SgChain& C = H.chain[c];
  const double h = a.timestep;
  double* gq = a.qpos + (size_t)env * nv;
  double* gv = a.qvel + (size_t)env * nv;
  double* gw = a.warm + (size_t)env * nv;
  const double kenv = a.kenv[env];
  double q[SG_CD], v[SG_CD], w[SG_CD], kk[SG_CD], act = 0, ctrl = 0;
#pragma unroll
  for (int d = 0; d < SG_CD; d++) {
    int j = C.dof0 + d;
    if (a.do_reset) { q[d] = C.qpos0[d]; v[d] = 0; w[d] = 0; }
    else { q[d] = gq[j]; v[d] = gv[j]; w[d] = gw[j]; }
    kk[d] = a.kmask_jnt[j] ? kenv : C.stiffness[d];
  }
  if (C.has_act) {
    if (a.do_reset) a.ctrl[(size_t)env * nu + C.act_id] = 0;
    else { act = a.act[(size_t)env * nu + C.act_id]; ctrl = a.ctrl[(size_t)env * nu + C.act_id]; }
  }
  const double kten = C.has_ten ? (a.kmask_ten[C.ten_id] ? kenv : C.ten_k0) : 0.0;
  double* ch = W.chh + st * SG_CHW;
  bool bad_acc = false;
  if (status & (SG_FLAG_BADQPOS | SG_FLAG_BADQVEL | SG_FLAG_BADQACC)) return;

  SG_T(17);
  if (a.do_finish && pend) {
    double aF[SG_CD], qacc_c[SG_CD];
#pragma unroll
    for (int d = 0; d < SG_CD; d++) {
      aF[d] = W.saF[(size_t)d * S + st];
      qacc_c[d] = ch[SGH_QSM + d] + aF[d];
      if (isbad(qacc_c[d])) bad_acc = true;
    }
    ChainKin K;
    {
      double* kd = (double*)&K;
#pragma unroll
      for (int i = 0; i < 48; i++) kd[i] = ch[SGH_K + i];
    }
    if (a.sens) {
      ChainMotion Mo;
      chain_motion(C, K, v, qacc_c, H.gravity, Mo);
      double* so = a.sens + (size_t)env * a.sens_stride;
      for (int s = 0; s < C.nsite; s++) {
        int bi = C.s_body[s];
        double r3[3], sm[9], t[3], t2[3], acc[3], out[3], sbp[3], sbm[9], bw[3], bal[3];
        chain_body_pose(K, bi, sbp, sbm);
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
    if (bad_acc) {
      atomicOr(&W.status[env], SG_FLAG_BADQACC);
    } else {
#pragma unroll
      for (int d = 0; d < SG_CD; d++) w[d] = qacc_c[d];
      if (a.finish_integrate) {
        bool damp = false;
#pragma unroll
        for (int d = 0; d < SG_CD; d++) damp |= C.damping[d] > 0;
        double qa[SG_CD];
        if (damp) {
          double MhB[16], MhBinv[16], rhs[SG_CD], Mm[16];
#pragma unroll
          for (int i = 0; i < 16; i++) { Mm[i] = ch[SGH_M + i]; MhB[i] = Mm[i]; }
#pragma unroll
          for (int d = 0; d < SG_CD; d++) MhB[5 * d] += h * C.damping[d];
          spd_inverse4(MhB, MhBinv);
#pragma unroll
          for (int a2 = 0; a2 < SG_CD; a2++) {
            double s2 = ch[SGH_QFRC + a2];
#pragma unroll
            for (int b2 = 0; b2 < SG_CD; b2++) s2 += Mm[4 * a2 + b2] * aF[b2];
            rhs[a2] = s2;
          }
#pragma unroll
          for (int a2 = 0; a2 < SG_CD; a2++) {
            double s2 = 0;
#pragma unroll
            for (int b2 = 0; b2 < SG_CD; b2++) s2 += MhBinv[4 * a2 + b2] * rhs[b2];
            qa[a2] = s2;
          }
        } else {
#pragma unroll
          for (int d = 0; d < SG_CD; d++) qa[d] = qacc_c[d];
        }
        act += h * ch[SGH_ACTDOT];
#pragma unroll
        for (int d = 0; d < SG_CD; d++) { v[d] += h * qa[d]; q[d] += h * v[d]; }
      }
    }
  }

  SG_T(18);
  if (a.do_begin && !bad_acc) {
    int bad = 0;
#pragma unroll
    for (int d = 0; d < SG_CD; d++) bad |= (isbad(q[d]) ? SG_FLAG_BADQPOS : 0) | (isbad(v[d]) ? SG_FLAG_BADQVEL : 0);
    if (bad) {
      atomicOr(&W.status[env], bad);
    } else {
      ChainKin K;
      ChainDyn D;
      chain_kinematics(C, q, K);
      SG_T(19);
      chain_dynamics(C, K, q, v, act, ctrl, kk, kten, H.gravity, D);
      SG_T(20);
      // hand-off record: assembled in registers and written as 16-byte stores (a lane's record is 1280 contiguous bytes; every store
      // instruction touches 64 different lines, so their number is what counts)
      double rec[SG_CHW];
#pragma unroll
      for (int i = 0; i < SG_CHW; i++) rec[i] = 0.0;
#pragma unroll
      for (int d = 0; d < SG_CD; d++) { rec[SGH_QSM + d] = D.qacc_smooth[d]; rec[SGH_QFRC + d] = D.qfrc_smooth[d]; rec[SGH_V + d] = v[d]; rec[SGH_W + d] = w[d]; }
      rec[SGH_ACTDOT] = D.act_dot;
#pragma unroll
      for (int i = 0; i < 16; i++) { rec[SGH_M + i] = D.M[i]; rec[SGH_MINV + i] = D.Minv[i]; W.sMinv[(size_t)i * S + st] = D.Minv[i]; }
      {
        const double* kd = (const double*)&K;
#pragma unroll
        for (int i = 0; i < 48; i++) rec[SGH_K + i] = kd[i];
      }
#pragma unroll
      for (int g = 0; g < SG_CG; g++) {
        double t[3] = {0, 0, 0}, bp_[3] = {0, 0, 0}, bm_[9], bm2[9];
#pragma unroll
        for (int k = 0; k < 9; k++) bm2[k] = 0;
        if (g < C.ngeom) {
          chain_body_pose(K, C.g_body[g], bp_, bm_);
          mulmat3(t, bm_, C.g_pos[g]);
          mulmat33(bm2, bm_, C.g_mat[g]);
        }
#pragma unroll
        for (int k = 0; k < 3; k++) rec[SGH_BOX + 12 * g + k] = bp_[k] + t[k];
#pragma unroll
        for (int k = 0; k < 9; k++) rec[SGH_BOX + 12 * g + 3 + k] = bm2[k];
      }
      SG_T(21);
      LimitRows L;
      limits_build(C, q, v, D.qacc_smooth, w, L);
      rec[SGH_LIMACT] = (double)L.active;
#pragma unroll
      for (int k = 0; k < SG_MAXLIM; k++) { rec[SGH_LIMSIGN + k] = L.sign[k]; rec[SGH_LIMR + k] = L.R[k]; rec[SGH_LIMB + k] = L.b[k]; rec[SGH_LIMF + k] = L.f[k]; }
      static_assert(SG_CHW % 2 == 0 && SGH_LIMF + SG_MAXLIM <= SG_CHW, "hand-off record layout");
#pragma unroll
      for (int i = 0; i < (SGH_LIMF + SG_MAXLIM + 1) / 2; i++) ((double2*)ch)[i] = make_double2(rec[2 * i], rec[2 * i + 1]);
    }
  }
  SG_T(22);
  // store the chain's state
#pragma unroll
  for (int d = 0; d < SG_CD; d++) { int j = C.dof0 + d; gq[j] = q[d]; gv[j] = v[d]; gw[j] = w[d]; }
  if (C.has_act) a.act[(size_t)env * nu + C.act_id] = act;
  SG_T(23);
  SG_TEND();
}


// ---- launchers ----
hipError_t sg_launch_chain(const SgPhaseArgs& p, int nenv, hipStream_t s) {
  hipLaunchKernelGGL(sg_chain_kernel, dim3(2 * ((nenv + 63) / 64)), dim3(64), 0, s, p);
  return hipGetLastError();
}
hipError_t sg_launch_phase(const SgPhaseArgs& p, int rounds, bool nb, bool genpass, int nenv, hipStream_t s) {
#define SG_PHASE(r)                                                                                   \
  if (nb) {                                                                                           \
    hipLaunchKernelGGL((sg_phase_kernel<r, SG_X_CPL, true>), dim3(nenv), dim3(64), 0, s, p);                 \
    if (genpass && p.do_begin) hipLaunchKernelGGL((sg_phase_kernel<r, 2, true, true>), dim3(nenv < SG_GEN_GRID ? nenv : SG_GEN_GRID), dim3(64), 0, s, p); \
  } else {                                                                                            \
    hipLaunchKernelGGL((sg_phase_kernel<r, SG_X_CPL, false>), dim3(nenv), dim3(64), 0, s, p);                \
    if (genpass && p.do_begin) hipLaunchKernelGGL((sg_phase_kernel<r, 2, false, true>), dim3(nenv < SG_GEN_GRID ? nenv : SG_GEN_GRID), dim3(64), 0, s, p); \
  }
  switch (rounds) {
    case 1: SG_PHASE(1); break;
    case 2: SG_PHASE(2); break;
    case 3: SG_PHASE(3); break;
    default: SG_PHASE(4); break;
  }
#undef SG_PHASE
  return hipGetLastError();
}
