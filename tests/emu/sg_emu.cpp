// sg_emu.cpp -- host-side, lane-serial driver of the kernels' per-lane math (csrc/sg_math.h).
//
// TEST HARNESS ONLY (built and used by tests/test_emu_vs_oracle.py): it executes the same
// matrix-free, structure-exploiting algorithm the HIP kernels run -- chain blocks, 1x1
// slider blocks, incremental M^-1 J' f, per-chain Gauss-Seidel streams -- but one lane at a
// time on the CPU, so the restructuring can be checked against the general-purpose oracle
// without a GPU.  It is not reachable from the product package.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../soft-grip_amd/csrc/sg_math.h"
#include "../../soft-grip_amd/csrc/sg_general.h"

using namespace sgm;

struct Emu {
  SgPlan P;
  int N;
  // state
  double qc[SG_MAXCH][SG_CD], vc[SG_MAXCH][SG_CD], wc[SG_MAXCH][SG_CD], act[SG_MAXCH], ctrl[SG_MAXCH];
  std::vector<double> qe, ve, we;
  // per-env parameters
  double kc[SG_MAXCH][SG_CD], kten[SG_MAXCH], kt0;
  std::vector<double> ke;
  std::vector<double> sens;
  int ncon, nefc, iters, flags;
  std::vector<Contact> dbg_con[SG_MAXCH];
  int general;  // the last substep ran on the general contact path (sg_general.h)
  std::vector<GenContact> dbg_gcon;
};

extern "C" {

// The plan's equality-row schedule of a neighbour-row model (SgPlan::sched), flattened for tests/test_plan_schedule.py:
// out[4 * i .. 4 * i + 3] = (round, e1, e2, row) of slot i; returns the number of slots (rounds * 8) or -1.
int emu_plan_schedule(const void* blob, size_t n, int* out, int cap, int* nelem, int* nnb, char* err, size_t errlen) {
  SgPlan P;
  std::string e;
  if (!sg_plan_build(blob, n, &P, &e)) {
    snprintf(err, errlen, "%s", e.c_str());
    return -1;
  }
  *nelem = P.h.nelem; *nnb = P.h.nnb;
  const int ns = (int)P.sched.size();
  if (ns != SG_EQ_SLOTS * P.h.eq_rounds || ns > cap) return -1;
  for (int i = 0; i < ns; i++) {
    out[4 * i] = P.sched[i].e; out[4 * i + 1] = P.sched[i].p[0]; out[4 * i + 2] = P.sched[i].p[1]; out[4 * i + 3] = P.sched[i].p[2];
  }
  return ns;
}

Emu* emu_new(const void* blob, size_t n, char* err, size_t errlen) {
  Emu* E = new Emu();
  std::string e;
  if (!sg_plan_build(blob, n, &E->P, &e)) {
    snprintf(err, errlen, "%s", e.c_str());
    delete E;
    return nullptr;
  }
  const SgPlanHeader& H = E->P.h;
  E->N = H.nelem;
  E->qe.resize(E->N); E->ve.resize(E->N); E->we.resize(E->N); E->ke.resize(E->N);
  E->sens.assign(H.nsensordata, 0.0);
  for (int c = 0; c < SG_MAXCH; c++) {
    for (int d = 0; d < SG_CD; d++) E->kc[c][d] = H.chain[c].stiffness[d];
    E->kten[c] = H.chain[c].ten_k0;
  }
  E->kt0 = H.t0_k0;
  for (int e2 = 0; e2 < E->N; e2++) E->ke[e2] = E->P.elem[(size_t)SGE_K0 * E->N + e2];
  return E;
}
void emu_free(Emu* E) { delete E; }
int emu_nelem(Emu* E) { return E->N; }

// stiffness by global joint / tendon id (reference environment/manenv.py:105-108)
void emu_set_jnt_stiffness(Emu* E, int j, double k) {
  const SgPlanHeader& H = E->P.h;
  if (j >= H.elem_dof0) { E->ke[j - H.elem_dof0] = k; return; }
  for (int c = 0; c < H.nchain; c++)
    if (j >= H.chain[c].dof0 && j < H.chain[c].dof0 + H.chain[c].ndof) E->kc[c][j - H.chain[c].dof0] = k;
}
void emu_set_tendon_stiffness(Emu* E, int t, double k) {
  const SgPlanHeader& H = E->P.h;
  if (t == H.t0_id) E->kt0 = k;
  for (int c = 0; c < H.nchain; c++)
    if (H.chain[c].has_ten && H.chain[c].ten_id == t) E->kten[c] = k;
}
void emu_set_ctrl(Emu* E, int u, double v) {
  const SgPlanHeader& H = E->P.h;
  for (int c = 0; c < H.nchain; c++)
    if (H.chain[c].has_act && H.chain[c].act_id == u) E->ctrl[c] = v;
}

void emu_reset(Emu* E) {
  const SgPlanHeader& H = E->P.h;
  for (int c = 0; c < SG_MAXCH; c++) {
    for (int d = 0; d < SG_CD; d++) { E->qc[c][d] = H.chain[c].qpos0[d]; E->vc[c][d] = E->wc[c][d] = 0; }
    E->act[c] = E->ctrl[c] = 0;
  }
  for (int e = 0; e < E->N; e++) { E->qe[e] = E->P.elem[(size_t)SGE_QPOS0 * E->N + e]; E->ve[e] = E->we[e] = 0; }
  std::fill(E->sens.begin(), E->sens.end(), 0.0);
  E->flags = 0;
}

void emu_get_state(Emu* E, double* qpos, double* qvel, double* warm, double* act) {
  const SgPlanHeader& H = E->P.h;
  for (int c = 0; c < H.nchain; c++)
    for (int d = 0; d < H.chain[c].ndof; d++) {
      int j = H.chain[c].dof0 + d;
      qpos[j] = E->qc[c][d]; qvel[j] = E->vc[c][d]; warm[j] = E->wc[c][d];
    }
  for (int e = 0; e < E->N; e++) { qpos[H.elem_dof0 + e] = E->qe[e]; qvel[H.elem_dof0 + e] = E->ve[e]; warm[H.elem_dof0 + e] = E->we[e]; }
  for (int c = 0; c < H.nchain; c++)
    if (H.chain[c].has_act) act[H.chain[c].act_id] = E->act[c];
}
void emu_set_state(Emu* E, const double* qpos, const double* qvel, const double* warm, const double* act) {  // re-seating (tests)
  const SgPlanHeader& H = E->P.h;
  for (int c = 0; c < H.nchain; c++)
    for (int d = 0; d < H.chain[c].ndof; d++) {
      int j = H.chain[c].dof0 + d;
      E->qc[c][d] = qpos[j]; E->vc[c][d] = qvel[j]; E->wc[c][d] = warm[j];
    }
  for (int e = 0; e < E->N; e++) { E->qe[e] = qpos[H.elem_dof0 + e]; E->ve[e] = qvel[H.elem_dof0 + e]; E->we[e] = warm[H.elem_dof0 + e]; }
  for (int c = 0; c < H.nchain; c++)
    if (H.chain[c].has_act) E->act[c] = act[H.chain[c].act_id];
}
const double* emu_sensordata(Emu* E) { return E->sens.data(); }
int emu_ncon(Emu* E) { return E->ncon; }
int emu_contact(Emu* E, int i, int* chain, int* sl, double* R, double* b3) {
  for (int c = 0; c < SG_MAXCH; c++) {
    if (i < (int)E->dbg_con[c].size()) { *chain = c; *sl = E->dbg_con[c][i].sl; *R = E->dbg_con[c][i].R; memcpy(b3, E->dbg_con[c][i].b, 24); return 1; }
    i -= (int)E->dbg_con[c].size();
  }
  return 0;
}
int emu_nefc(Emu* E) { return E->nefc; }
int emu_general(Emu* E) { return E->general; }
int emu_gcontact(Emu* E, int i, double* out) {  // the i-th general contact as built (before the solve), SG_GEN_W doubles
  if (i < 0 || i >= (int)E->dbg_gcon.size()) return 0;
  gen_contact_store(out, E->dbg_gcon[i]);
  return 1;
}
int emu_iters(Emu* E) { return E->iters; }

// one mj_forward (+ Euler when integrate != 0)
int emu_substep(Emu* E, int integrate) {
  const SgPlan& P = E->P;
  const SgPlanHeader& H = P.h;
  const int N = E->N;
  const double h = H.timestep;
  auto EL = [&](int f, int e) { return P.elem[(size_t)f * N + e]; };
  int flags = 0;
  for (int e = 0; e < N; e++)
    if (isbad(E->qe[e]) || isbad(E->ve[e])) flags |= 1;
  for (int c = 0; c < H.nchain; c++)
    for (int d = 0; d < H.chain[c].ndof; d++)
      if (isbad(E->qc[c][d]) || isbad(E->vc[c][d])) flags |= 1;
  if (flags) { E->flags |= flags; return flags; }

  // ---- chains ----
  ChainKin K[SG_MAXCH];
  ChainDyn D[SG_MAXCH];
  double boxp[SG_MAXCH][SG_CG][3], boxm[SG_MAXCH][SG_CG][9];
  for (int c = 0; c < H.nchain; c++) {
    const SgChain& C = H.chain[c];
    chain_kinematics(C, E->qc[c], K[c]);
    chain_dynamics(C, K[c], E->qc[c], E->vc[c], E->act[c], E->ctrl[c], E->kc[c], E->kten[c], H.gravity, D[c]);
    for (int g = 0; g < C.ngeom; g++) {
      double t[3], bp_[3], bm_[9];
      chain_body_pose(K[c], C.g_body[g], bp_, bm_);
      mulmat3(t, bm_, C.g_pos[g]);
      for (int k = 0; k < 3; k++) boxp[c][g][k] = bp_[k] + t[k];
      mulmat33(boxm[c][g], bm_, C.g_mat[g]);
    }
  }
  // ---- elements: smooth dynamics ----
  std::vector<double> invm(N), fsm(N), asm_e(N), cpos(3 * N);
  double L0 = 0, Ld = 0;
  for (int e = 0; e < N; e++) { L0 += EL(SGE_COEF, e) * E->qe[e]; Ld += EL(SGE_COEF, e) * E->ve[e]; }
  double frc_t0 = -E->kt0 * (L0 - H.t0_lspring) - H.t0_damping * Ld;
  for (int e = 0; e < N; e++) {
    double ax[3] = {EL(SGE_AX, e), EL(SGE_AY, e), EL(SGE_AZ, e)}, m = EL(SGE_MASS, e);
    double bias = -m * dot3(H.gravity, ax);
    double f = -E->ke[e] * (E->qe[e] - EL(SGE_SPRINGREF, e)) - EL(SGE_DAMPING, e) * E->ve[e] + EL(SGE_COEF, e) * frc_t0 - bias;
    invm[e] = 1.0 / (m + EL(SGE_ARMATURE, e));
    fsm[e] = f; asm_e[e] = f * invm[e];
    double dq = E->qe[e] - EL(SGE_QPOS0, e);
    cpos[3 * e] = EL(SGE_GX, e) + ax[0] * dq; cpos[3 * e + 1] = EL(SGE_GY, e) + ax[1] * dq; cpos[3 * e + 2] = EL(SGE_GZ, e) + ax[2] * dq;
  }
  // ---- collision: per chain, per box: centre sphere first, then elements in order ----
  std::vector<Contact> con[SG_MAXCH];
  for (int c = 0; c < H.nchain; c++) {
    const SgChain& C = H.chain[c];
    for (int g = 0; g < C.ngeom; g++) {
      int nd = chain_ndof_of_body(C.g_body[g]);
      double binvw = C.b_invw_tran[C.g_body[g]];
      auto emit = [&](const ConRec& r, int sl) {
        Contact ct;
        double ax[3] = {0, 0, 0};
        if (sl >= 0) { ax[0] = EL(SGE_AX, sl); ax[1] = EL(SGE_AY, sl); ax[2] = EL(SGE_AZ, sl); }
        contact_build(ct, r, K[c], nd, D[c].Minv, E->vc[c], D[c].qacc_smooth, E->wc[c], binvw, sl, ax, sl >= 0 ? E->ve[sl] : 0.0,
                      sl >= 0 ? asm_e[sl] : 0.0, sl >= 0 ? E->we[sl] : 0.0, sl >= 0 ? invm[sl] : 0.0, sl >= 0 ? EL(SGE_BINVW, sl) : 0.0, H);
        con[c].push_back(ct);
      };
      if (H.has_center) {
        double dif[3] = {boxp[c][g][0] - H.center_pos[0], boxp[c][g][1] - H.center_pos[1], boxp[c][g][2] - H.center_pos[2]};
        double bound = H.center_radius + C.g_rbound[g] + H.con_margin;
        ConRec r;
        if (dot3(dif, dif) <= bound * bound && sphere_box(H.center_pos, H.center_radius, boxp[c][g], boxm[c][g], C.g_size[g], H.con_margin, r))
          emit(r, -1);
      }
      for (int e = 0; e < N; e++) {
        double dif[3] = {boxp[c][g][0] - cpos[3 * e], boxp[c][g][1] - cpos[3 * e + 1], boxp[c][g][2] - cpos[3 * e + 2]};
        double bound = H.cap_rbound + C.g_rbound[g] + H.con_margin;
        if (dot3(dif, dif) > bound * bound) continue;
        double cax[3] = {EL(SGE_CX, e), EL(SGE_CY, e), EL(SGE_CZ, e)};
        ConRec r0, r1;
        int mk = capsule_box(&cpos[3 * e], cax, H.cap_radius, H.cap_hl, boxp[c][g], boxm[c][g], C.g_size[g], H.con_margin, r0, r1);
        if ((mk & 1) && r0.dist < H.con_margin) emit(r0, e);
        if ((mk & 2) && r1.dist < H.con_margin) emit(r1, e);
      }
    }
  }
  // envelope checks: a pair outside the fast path's two kinds within reach -> this substep runs on the general contact path
  bool special = false;
  for (int c = 0; c < H.nchain; c++)
    for (int g = 0; g < H.chain[c].ngeom; g++) {
      const SgChain& C = H.chain[c];
      for (int s = 0; s < H.nstatic; s++) {
        double dif[3] = {boxp[c][g][0] - H.st_pos[s][0], boxp[c][g][1] - H.st_pos[s][1], boxp[c][g][2] - H.st_pos[s][2]}, bd = C.g_rbound[g] + H.st_rbound[s];
        if (dot3(dif, dif) <= bd * bd && box_box_overlap(boxp[c][g], boxm[c][g], C.g_size[g], H.st_pos[s], H.st_mat[s], H.st_size[s], 0)) special = true;
      }
      for (int c2 = c + 1; c2 < H.nchain; c2++)
        for (int g2 = 0; g2 < H.chain[c2].ngeom; g2++) {
          double dif[3] = {boxp[c][g][0] - boxp[c2][g2][0], boxp[c][g][1] - boxp[c2][g2][1], boxp[c][g][2] - boxp[c2][g2][2]}, bd = C.g_rbound[g] + H.chain[c2].g_rbound[g2];
          if (dot3(dif, dif) <= bd * bd && box_box_overlap(boxp[c][g], boxm[c][g], C.g_size[g], boxp[c2][g2], boxm[c2][g2], H.chain[c2].g_size[g2], 0)) special = true;
        }
      if (H.has_plane) {
        double dif[3] = {boxp[c][g][0] - H.plane_pos[0], boxp[c][g][1] - H.plane_pos[1], boxp[c][g][2] - H.plane_pos[2]}, ext = 0;
        for (int k = 0; k < 3; k++) ext += C.g_size[g][k] * fabs(H.plane_normal[0] * boxm[c][g][k] + H.plane_normal[1] * boxm[c][g][3 + k] + H.plane_normal[2] * boxm[c][g][6 + k]);
        if (dot3(dif, H.plane_normal) - ext <= 0) special = true;
      }
    }
  for (int e = 0; e < N; e++)
    if (!(E->qe[e] > EL(SGE_QLO, e) && E->qe[e] < EL(SGE_QHI, e))) special = true;
  // ---- general contact path (sg_general.h): ONE ordered list over the plan's candidate pairs, generic rows ----
  std::vector<GenContact> gcon;
  E->general = special;
  if (special) {
    for (int c = 0; c < H.nchain; c++) con[c].clear();
    double bxp[SG_MAXCH * SG_CG][3], bxm[SG_MAXCH * SG_CG][9];
    for (int c = 0; c < SG_MAXCH; c++)
      for (int g = 0; g < SG_CG; g++) { memcpy(bxp[c * SG_CG + g], boxp[c][g], 24); memcpy(bxm[c * SG_CG + g], boxm[c][g], 72); }
    std::vector<double> ebinvw(N);
    for (int e = 0; e < N; e++) ebinvw[e] = EL(SGE_BINVW, e);
    double Minv2[SG_MAXCH][16], vc2[SG_MAXCH][SG_CD], asm2[SG_MAXCH][SG_CD], warm2[SG_MAXCH][SG_CD];
    for (int c = 0; c < SG_MAXCH; c++) {
      for (int i = 0; i < 16; i++) Minv2[c][i] = c < H.nchain ? D[c].Minv[i] : 0.0;
      for (int d = 0; d < SG_CD; d++) { vc2[c][d] = c < H.nchain ? E->vc[c][d] : 0.0; asm2[c][d] = c < H.nchain ? D[c].qacc_smooth[d] : 0.0; warm2[c][d] = c < H.nchain ? E->wc[c][d] : 0.0; }
    }
    for (const SgGenPair& gp : P.gpairs) {
      ConRec rec[8];
      int n = 0;
      const double* hint = nullptr;
      double cax[3], cp[3];
      auto capsule_of = [&](int ref) {
        const int e = sgg_index(ref);
        cp[0] = cpos[3 * e]; cp[1] = cpos[3 * e + 1]; cp[2] = cpos[3 * e + 2];
        cax[0] = EL(SGE_CX, e); cax[1] = EL(SGE_CY, e); cax[2] = EL(SGE_CZ, e);
      };
      const double *p2 = nullptr, *R2 = nullptr, *s2 = nullptr, *p1 = nullptr, *R1 = nullptr, *s1 = nullptr;
      double rb1 = 0, rb2 = 0;
      if (gp.kind == SGP_PLANE_CAP) {
        capsule_of(gp.g2);
        double dif[3] = {cp[0] - H.plane_pos[0], cp[1] - H.plane_pos[1], cp[2] - H.plane_pos[2]};
        if (dot3(dif, H.plane_normal) > H.con_margin + H.cap_rbound) continue;
        n = gen_plane_capsule(H.plane_pos, H.plane_normal, cp, cax, H.cap_radius, H.cap_hl, H.con_margin, rec[0], rec[1]);
        hint = cax;
      } else if (gp.kind == SGP_PLANE_BOX) {
        gen_box_of(gp.g2, H, bxp, bxm, p2, R2, s2, rb2);
        double dif[3] = {p2[0] - H.plane_pos[0], p2[1] - H.plane_pos[1], p2[2] - H.plane_pos[2]};
        if (dot3(dif, H.plane_normal) > H.con_margin + rb2) continue;
        n = gen_plane_box(H.plane_pos, H.plane_normal, p2, R2, s2, H.con_margin, rec);
      } else {
        if (gp.kind == SGP_SPH_BOX) { cp[0] = H.center_pos[0]; cp[1] = H.center_pos[1]; cp[2] = H.center_pos[2]; rb1 = H.center_radius; }
        else if (gp.kind == SGP_CAP_BOX) { capsule_of(gp.g1); rb1 = H.cap_rbound; }
        else if (gp.kind == SGP_BOX_BOX || gp.kind == SGP_UNSUPPORTED) {
          if (sgg_kind(gp.g1) != SGG_BOX && sgg_kind(gp.g1) != SGG_STATIC) { flags |= 32; continue; }  // an unsupported pair that is not box - box: flag whenever checked
          gen_box_of(gp.g1, H, bxp, bxm, p1, R1, s1, rb1);
          cp[0] = p1[0]; cp[1] = p1[1]; cp[2] = p1[2];
        }
        gen_box_of(gp.g2, H, bxp, bxm, p2, R2, s2, rb2);
        double dif[3] = {p2[0] - cp[0], p2[1] - cp[1], p2[2] - cp[2]}, bound = rb1 + rb2 + H.con_margin;
        if (dot3(dif, dif) > bound * bound) continue;
        if (gp.kind == SGP_SPH_BOX) n = sphere_box(cp, H.center_radius, p2, R2, s2, H.con_margin, rec[0]);
        else if (gp.kind == SGP_CAP_BOX) {
          int mk = capsule_box(cp, cax, H.cap_radius, H.cap_hl, p2, R2, s2, H.con_margin, rec[0], rec[1]);
          if ((mk & 2) && !(mk & 1)) rec[0] = rec[1];
          n = (mk & 1) + ((mk >> 1) & 1);
        } else {
          double poly[16][3], tmp[16][3];
          n = gen_box_box(p1, R1, s1, p2, R2, s2, H.con_margin, rec, poly, tmp);
          if (gp.kind == SGP_UNSUPPORTED) { if (n > 0) flags |= 32; continue; }
        }
      }
      const GenSide S1 = gen_side_of(gp.g1, H, ebinvw.data()), S2 = gen_side_of(gp.g2, H, ebinvw.data());
      const int sl = S1.sl >= 0 ? S1.sl : S2.sl;
      for (int q = 0; q < n; q++) {
        if (!(rec[q].dist < H.con_margin)) continue;
        if ((int)gcon.size() == SG_GEN_MAXCON) { flags |= 8; break; }
        GenContact gc;
        double ax[3] = {0, 0, 0};
        if (sl >= 0) { ax[0] = EL(SGE_AX, sl); ax[1] = EL(SGE_AY, sl); ax[2] = EL(SGE_AZ, sl); }
        gen_contact_build(gc, rec[q], hint, S1, S2, K, Minv2, vc2, asm2, warm2, ax, sl >= 0 ? E->ve[sl] : 0.0, sl >= 0 ? asm_e[sl] : 0.0,
                          sl >= 0 ? E->we[sl] : 0.0, sl >= 0 ? invm[sl] : 0.0, H);
        gcon.push_back(gc);
      }
    }
  }
  // ---- equality rows ----
  std::vector<double> eqR(N), eqb(N), eqf(N);
  for (int e = 0; e < N; e++) {
    double pos = E->qe[e] - EL(SGE_QPOS0, e), imp = impedance(H.eqj_solimp, pos, 0);
    eqR[e] = fmax(SG_MINVAL, (1 - imp) / imp * EL(SGE_INVW, e));
    double aref = -H.eqj_B * E->ve[e] - H.eqj_K * imp * pos;
    eqb[e] = asm_e[e] - aref;
    eqf[e] = -(E->we[e] - aref) / eqR[e];
  }
  double tpos = L0 - H.t0_L0, timp = impedance(H.eqt_solimp, tpos, 0), tR = fmax(SG_MINVAL, (1 - timp) / timp * H.eqt_invw);
  double taref = -H.eqt_B * Ld - H.eqt_K * timp * tpos, tb = -taref, tjar = -taref, tA = tR;
  for (int e = 0; e < N; e++) {
    double cf = EL(SGE_COEF, e);
    tb += cf * asm_e[e]; tjar += cf * E->we[e]; tA += cf * cf * invm[e];
  }
  double tf = -tjar / tR;
  // ---- neighbour equality rows (models with the composite's neighbour equalities: slider e = slider e2, J = +1 / -1), as the phase
  //      kernel builds them: by workspace slot d * N + e (the d-th row registered for element e, MuJoCo's order)
  const int nnb = H.nnb;
  const int* nbt = P.nbtab.data();
  std::vector<double> nbf(3 * N + 1, 0.0), nbb(3 * N + 1, 0.0), nbR(3 * N + 1, 1.0);
  std::vector<int> nb_e2(3 * N + 1, -1), nb_e1(3 * N + 1, -1);
  if (nnb > 0)
    for (int e = 0; e < N; e++)
      for (int d = 0; d < 3; d++) {
        const int e2 = nbt[d * N + e];
        if (e2 < 0) continue;
        const int id = nbt[(3 + d) * N + e];
        const double pos = (E->qe[e] - EL(SGE_QPOS0, e)) - (E->qe[e2] - EL(SGE_QPOS0, e2)), imp = impedance(H.eqj_solimp, pos, 0);
        const double Rr = fmax(SG_MINVAL, (1 - imp) / imp * (EL(SGE_INVW, e) + EL(SGE_INVW, e2)));
        const double aref = -H.eqj_B * (E->ve[e] - E->ve[e2]) - H.eqj_K * imp * pos;
        nbb[id] = (asm_e[e] - asm_e[e2]) - aref; nbR[id] = Rr; nbf[id] = -((E->we[e] - E->we[e2]) - aref) / Rr;
        nb_e1[id] = e; nb_e2[id] = e2;
      }
  LimitRows Lm[SG_MAXCH];
  for (int c = 0; c < H.nchain; c++) limits_build(H.chain[c], E->qc[c], E->vc[c], D[c].qacc_smooth, E->wc[c], Lm[c]);
  E->ncon = 0; E->nefc = N + 1 + nnb;
  for (int c = 0; c < H.nchain; c++) { E->ncon += (int)con[c].size(); E->nefc += __builtin_popcount(Lm[c].active) + 3 * (int)con[c].size(); }
  E->ncon += (int)gcon.size(); E->nefc += 3 * (int)gcon.size();

  for (int c = 0; c < H.nchain; c++) E->dbg_con[c] = con[c];
  E->dbg_gcon = gcon;
  // ---- M^-1 J' f from scratch ----
  std::vector<double> ae(N);
  double aF[SG_MAXCH][SG_CD];
  auto recompute_a = [&]() {
    for (int e = 0; e < N; e++) {
      double fe = eqf[e] + EL(SGE_COEF, e) * tf;
      if (nnb > 0) {   // + its own neighbour rows (J = +1), - the rows that have it as second joint (J = -1), in the kernel's order
        for (int d = 0; d < 3; d++) if (nbt[d * N + e] >= 0) fe += nbf[nbt[(3 + d) * N + e]];
        for (int d = 0; d < 3; d++) { const int ii = nbt[(6 + d) * N + e]; if (ii >= 0) fe -= nbf[ii]; }
      }
      ae[e] = invm[e] * fe;
    }
    for (int c = 0; c < H.nchain; c++) {
      double g[SG_CD] = {0, 0, 0, 0};
      for (int k = 0; k < SG_MAXLIM; k++)
        if (Lm[c].active >> k & 1) g[k / 2] += Lm[c].sign[k] * Lm[c].f[k];
      for (auto& ct : con[c]) {
        for (int d = 0; d < SG_CD; d++) g[d] += ct.Jf[0][d] * ct.f[0] + ct.Jf[1][d] * ct.f[1] + ct.Jf[2][d] * ct.f[2];
        if (ct.sl >= 0) ae[ct.sl] += ct.invm * (ct.Js[0] * ct.f[0] + ct.Js[1] * ct.f[1] + ct.Js[2] * ct.f[2]);
      }
      for (auto& gc : gcon)
        for (int d = 0; d < SG_CD; d++) g[d] += gc.Jf[c][0][d] * gc.f[0] + gc.Jf[c][1][d] * gc.f[1] + gc.Jf[c][2][d] * gc.f[2];
      for (int a = 0; a < SG_CD; a++) {
        double s = 0;
        for (int b = 0; b < SG_CD; b++) s += D[c].Minv[4 * a + b] * g[b];
        aF[c][a] = s;
      }
    }
    for (auto& gc : gcon)
      if (gc.sl >= 0) ae[gc.sl] += gc.invm * (gc.Js[0] * gc.f[0] + gc.Js[1] * gc.f[1] + gc.Js[2] * gc.f[2]);
  };
  recompute_a();
  // ---- warmstart cost 0.5 f'(A+R)f + f'b ----
  double cost = 0, tJa = 0;
  for (int e = 0; e < N; e++) { cost += eqf[e] * (0.5 * (ae[e] + eqR[e] * eqf[e]) + eqb[e]); tJa += EL(SGE_COEF, e) * ae[e]; }
  cost += tf * (0.5 * (tJa + tR * tf) + tb);
  for (int id = 0; id < 3 * N && nnb > 0; id++)
    if (nb_e1[id] >= 0) cost += nbf[id] * (0.5 * ((ae[nb_e1[id]] - ae[nb_e2[id]]) + nbR[id] * nbf[id]) + nbb[id]);
  for (int c = 0; c < H.nchain; c++) {
    for (int k = 0; k < SG_MAXLIM; k++)
      if (Lm[c].active >> k & 1) cost += Lm[c].f[k] * (0.5 * (Lm[c].sign[k] * aF[c][k / 2] + Lm[c].R[k] * Lm[c].f[k]) + Lm[c].b[k]);
    for (auto& ct : con[c])
      for (int r = 0; r < 3; r++) {
        double Ja = ct.sl >= 0 ? ct.Js[r] * ae[ct.sl] : 0.0;
        for (int d = 0; d < SG_CD; d++) Ja += ct.Jf[r][d] * aF[c][d];
        cost += ct.f[r] * (0.5 * (Ja + ct.R * ct.f[r]) + ct.b[r]);
      }
  }
  for (auto& gc : gcon)
    for (int r = 0; r < 3; r++) {
      double Ja = gc.sl >= 0 ? gc.Js[r] * ae[gc.sl] : 0.0;
      for (int c = 0; c < SG_MAXCH; c++)
        for (int d = 0; d < SG_CD; d++) Ja += gc.Jf[c][r][d] * aF[c][d];
      cost += gc.f[r] * (0.5 * (Ja + gc.R * gc.f[r]) + gc.b[r]);
    }
  if (cost > 0) {
    for (auto& gc : gcon) gc.f[0] = gc.f[1] = gc.f[2] = 0;
    std::fill(nbf.begin(), nbf.end(), 0.0);
    std::fill(eqf.begin(), eqf.end(), 0.0);
    tf = 0;
    for (int c = 0; c < H.nchain; c++) {
      for (int k = 0; k < SG_MAXLIM; k++) Lm[c].f[k] = 0;
      for (auto& ct : con[c]) ct.f[0] = ct.f[1] = ct.f[2] = 0;
    }
    recompute_a();
  }
  // ---- PGS ----
  E->iters = 0;
  // Neighbour-row models: the solver kernel's formulation of the equality block (sg_pgs_rows_kernel<.., NB = true>), lane-serially --
  // the plan's block schedule (one block = element e's fix row and its up to three neighbour rows, all on slider e), per row the state
  // g = b + R f and the step factor c = (1/m) / (A + R), slider accelerations kept MINUS the env's offset aoff (the tendon row's
  // push, the same for every slider), the tendon row's J a tracked instead of summed
  const double im0 = nnb > 0 ? invm[0] : 0.0;
  std::vector<double> Ae(N + 1, 0.0), recg(4 * (N + 1), 0.0), recc(4 * (N + 1), 0.0);
  double aoff = 0.0, Ssum = 0.0, dS = 0.0;
  if (nnb > 0) {
    for (int e = 0; e < N; e++) {
      Ae[e] = ae[e]; Ssum += ae[e];
      recg[4 * e] = eqb[e] + eqR[e] * eqf[e]; recc[4 * e] = sg_div(im0, im0 + eqR[e]);
      for (int d = 0; d < 3; d++)
        if (nbt[d * N + e] >= 0) {
          const int id = nbt[(3 + d) * N + e];
          recg[4 * e + 1 + d] = nbb[id] + nbR[id] * nbf[id]; recc[4 * e + 1 + d] = sg_div(im0, 2.0 * im0 + nbR[id]);
        }
    }
  }
  for (int it = 0; it < H.iterations; it++) {
    double improvement = 0;
    if (nnb > 0) {
      double qc = 0, sc = 0;
      for (size_t si = 0; si < P.sched.size(); si++) {
        const SgEqSlot& sl = P.sched[si];
        const int e = sl.e;
        if (e >= N) continue;   // idle slot
        double ek = Ae[e];
        for (int k = 0; k < 4; k++) {
          const int pk = k == 0 ? -1 : sl.p[k - 1];
          if (k > 0 && pk >= N) continue;                      // no such row (its record is (0, 0): w = 0)
          const double Pk = k == 0 ? -aoff : Ae[pk], g = recg[4 * e + k], c = recc[4 * e + k];
          const double dk = g - Pk, sk = dk + ek, wk = c * sk, en = ek - wk, Pn = k == 0 ? Pk : Pk + wk;
          if (k > 0) Ae[pk] = Pn;
          recg[4 * e + k] = Pn - en;
          qc += sk * wk;
          if (k == 0) sc += wk;
          ek = en;
        }
        Ae[e] = ek;
      }
      improvement += 0.5 * qc * (1.0 / im0);
      dS -= sc;
      const double Ja = Ssum + dS, old = tf;
      improvement -= scalar_update(tf, tb, Ja, tR, tA, false);
      const double dft = tf - old;
      aoff += im0 * dft;
      Ssum = Ja + (tA - tR) * dft;
      dS = 0.0;
      for (int e = 0; e < N; e++) ae[e] = Ae[e] + aoff;       // the contact rows below read and push the true accelerations
    } else {
    for (int e = 0; e < N; e++) {
      double old = eqf[e];
      improvement -= scalar_update(eqf[e], eqb[e], ae[e], eqR[e], invm[e] + eqR[e], false);
      ae[e] += invm[e] * (eqf[e] - old);
    }
    {
      double Ja = 0, old = tf;
      for (int e = 0; e < N; e++) Ja += EL(SGE_COEF, e) * ae[e];
      improvement -= scalar_update(tf, tb, Ja, tR, tA, false);
      for (int e = 0; e < N; e++) ae[e] += invm[e] * EL(SGE_COEF, e) * (tf - old);
    }
    }
    const std::vector<double> ae_before = ae;   // (NB) what the contact rows add to the sliders goes into the tracked sum
    for (int c = 0; c < H.nchain; c++) {
      for (int k = 0; k < SG_MAXLIM; k++) {
        if (!(Lm[c].active >> k & 1)) continue;
        int d = k / 2;
        double old = Lm[c].f[k];
        improvement -= scalar_update(Lm[c].f[k], Lm[c].b[k], Lm[c].sign[k] * aF[c][d], Lm[c].R[k], D[c].Minv[5 * d] + Lm[c].R[k], true);
        double df = Lm[c].sign[k] * (Lm[c].f[k] - old);
        for (int a = 0; a < SG_CD; a++) aF[c][a] += D[c].Minv[4 * a + d] * df;
      }
    }
    for (int c = 0; c < H.nchain; c++)
      for (auto& ct : con[c]) {
        double df[3], as_ = ct.sl >= 0 ? ae[ct.sl] : 0.0;
        improvement -= contact_update(ct, aF[c], as_, H.con_mu, df);
        double g[SG_CD];
        for (int d = 0; d < SG_CD; d++) g[d] = ct.Jf[0][d] * df[0] + ct.Jf[1][d] * df[1] + ct.Jf[2][d] * df[2];
        for (int a = 0; a < SG_CD; a++)
          for (int b = 0; b < SG_CD; b++) aF[c][a] += D[c].Minv[4 * a + b] * g[b];
        if (ct.sl >= 0) ae[ct.sl] += ct.invm * (ct.Js[0] * df[0] + ct.Js[1] * df[1] + ct.Js[2] * df[2]);
      }
    for (auto& gc : gcon) {   // the general path's single stream, in mj_collision's order (after both chains' limit rows)
      double df[3], as_ = gc.sl >= 0 ? ae[gc.sl] : 0.0;
      improvement -= gen_contact_update(gc, aF, as_, H.con_mu, df);
      for (int c = 0; c < H.nchain; c++) {
        double g[SG_CD];
        for (int d = 0; d < SG_CD; d++) g[d] = gc.Jf[c][0][d] * df[0] + gc.Jf[c][1][d] * df[1] + gc.Jf[c][2][d] * df[2];
        for (int a = 0; a < SG_CD; a++)
          for (int b = 0; b < SG_CD; b++) aF[c][a] += D[c].Minv[4 * a + b] * g[b];
      }
      if (gc.sl >= 0) ae[gc.sl] += gc.invm * (gc.Js[0] * df[0] + gc.Js[1] * df[1] + gc.Js[2] * df[2]);
    }
    if (nnb > 0)
      for (int e = 0; e < N; e++) { dS += ae[e] - ae_before[e]; Ae[e] = ae[e] - aoff; }
    E->iters = it + 1;
    if (improvement * H.pgs_scale < H.tolerance) break;
  }
  if (nnb == 0) recompute_a();   // (NB: the block formulation keeps states g, not forces -- the accelerations it maintains ARE the result, as in the kernel)
  // ---- qacc, sensors, warmstart ----
  double qaccc[SG_MAXCH][SG_CD];
  for (int c = 0; c < H.nchain; c++) {
    const SgChain& C = H.chain[c];
    for (int d = 0; d < SG_CD; d++) {
      qaccc[c][d] = D[c].qacc_smooth[d] + aF[c][d];
      if (d < C.ndof && isbad(qaccc[c][d])) flags |= 4;
    }
    ChainMotion Mo;
    chain_motion(C, K[c], E->vc[c], qaccc[c], H.gravity, Mo);
    for (int s = 0; s < C.nsite; s++) {
      int bi = C.s_body[s];
      double r[3], sm[9], t[3], t2[3], a[3], sbp[3], sbm[9];
      chain_body_pose(K[c], bi, sbp, sbm);
      mulmat3(r, sbm, C.s_pos[s]);
      mulmat33(sm, sbm, C.s_mat[s]);
      if (C.s_gyro_adr[s] >= 0) mulmatT3(&E->sens[C.s_gyro_adr[s]], sm, Mo.w[bi]);
      if (C.s_acc_adr[s] >= 0) {
        for (int k = 0; k < 3; k++) a[k] = Mo.a[bi][k];
        cross3(t, Mo.al[bi], r); addscl3(a, t, 1);
        cross3(t, Mo.w[bi], r); cross3(t2, Mo.w[bi], t); addscl3(a, t2, 1);
        mulmatT3(&E->sens[C.s_acc_adr[s]], sm, a);
      }
    }
  }
  for (int e = 0; e < N; e++)
    if (isbad(asm_e[e] + ae[e])) flags |= 4;
  E->flags |= flags;
  if (flags & 4) return flags;
  for (int c = 0; c < H.nchain; c++)
    for (int d = 0; d < SG_CD; d++) E->wc[c][d] = qaccc[c][d];
  for (int e = 0; e < N; e++) E->we[e] = asm_e[e] + ae[e];
  if (!integrate) return flags;
  // ---- Euler with implicit joint damping ----
  for (int c = 0; c < H.nchain; c++) {
    const SgChain& C = H.chain[c];
    bool damp = false;
    for (int d = 0; d < C.ndof; d++) damp |= C.damping[d] > 0;
    double qa[SG_CD];
    if (damp) {
      double MhB[16], MhBinv[16], rhs[SG_CD];
      for (int i = 0; i < 16; i++) MhB[i] = D[c].M[i];
      for (int d = 0; d < C.ndof; d++) MhB[5 * d] += h * C.damping[d];
      spd_inverse4(MhB, MhBinv);
      for (int a = 0; a < SG_CD; a++) {
        double s = D[c].qfrc_smooth[a];
        for (int b = 0; b < SG_CD; b++) s += D[c].M[4 * a + b] * aF[c][b];
        rhs[a] = s;
      }
      for (int a = 0; a < SG_CD; a++) {
        double s = 0;
        for (int b = 0; b < SG_CD; b++) s += MhBinv[4 * a + b] * rhs[b];
        qa[a] = s;
      }
    } else {
      for (int d = 0; d < SG_CD; d++) qa[d] = qaccc[c][d];
    }
    E->act[c] += h * D[c].act_dot;
    for (int d = 0; d < C.ndof; d++) { E->vc[c][d] += h * qa[d]; E->qc[c][d] += h * E->vc[c][d]; }
  }
  double kk = 0;
  if (H.t0_implicit) {  // deviation D5 (sg_split.hip FINISH)
    double Ssum = 0;
    for (int e = 0; e < N; e++) {
      double m = 1.0 / invm[e];
      Ssum += EL(SGE_COEF, e) * (fsm[e] + m * ae[e]) / (m + h * EL(SGE_DAMPING, e));
    }
    kk = h * H.t0_damping * Ssum / (1.0 + H.t0_hcT);
  }
  for (int e = 0; e < N; e++) {
    double m = 1.0 / invm[e];
    double den = m + h * EL(SGE_DAMPING, e);
    double qa = (fsm[e] + m * ae[e]) / den - (H.t0_implicit ? EL(SGE_COEF, e) / den * kk : 0.0);
    E->ve[e] += h * qa;
    E->qe[e] += h * E->ve[e];
  }
  return flags;
}

}  // extern "C"
