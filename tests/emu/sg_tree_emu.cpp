// sg_tree_emu.cpp -- host-side driver of the TREE pipeline's source (csrc/sg_tree.h).
//
// TEST HARNESS ONLY (tests/test_tree_emu.py): sg_tree.h is bulk-synchronous code -- parallel loops, single-lane sections, barriers --
// and compiles for the host with a parallel loop as a serial loop and a wavefront sum as the identity.  This file gives that build
// the few entry points a test needs (state, stiffness, ctrl, reset, step), one env at a time, so that the restructured algorithm
// (dense per-chain blocks, matrix-free rows, pair-table collision) is checked against the general-purpose oracle without a GPU, and
// under ASan / UBSan.  It is not reachable from the product package.
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../soft-grip_amd/csrc/sg_tree.h"

struct TreeEmu {
  SgPlan P;
  SgTreeDev T;
  std::vector<double> qpos, qvel, warm, act, ctrl, sens, cws, lds;
  std::vector<int> kmask_jnt, kmask_ten;
  double kenv;
  int flags, touch, touchw[2], ncon, nefc, iters;
#ifdef SGT_EMU_SEPARATE
  sgt::SepPool pool;   // the checking build: one heap block per array (sg_tree.h lds_carve)
  ~TreeEmu() {
    for (auto& b : pool.lds) free(b.first);
    for (auto& b : pool.glob) free(b.first);
    for (double* q : pool.part) free(q);
  }
#endif
};

extern "C" {

TreeEmu* temu_new(const void* blob, size_t n, char* err, size_t errlen) {
  TreeEmu* E = new TreeEmu();
  std::string e;
  if (!sg_tree_plan_build(blob, n, &E->P, &E->T, &e)) {
    snprintf(err, errlen, "%s", e.c_str());
    delete E;
    return nullptr;
  }
  const SgPlanHeader& H = E->P.h;
  const int nu = H.nu > 0 ? H.nu : 1;
  E->qpos.assign(H.nq, 0.0); E->qvel.assign(H.nv, 0.0); E->warm.assign(H.nv, 0.0); E->act.assign(nu, 0.0); E->ctrl.assign(nu, 0.0);
  E->sens.assign(H.nsensordata, 0.0);
  E->kmask_jnt.assign(H.njnt, 0); E->kmask_ten.assign(H.ntendon, 0);
  E->kenv = 0;
  E->cws.assign((size_t)sgt::cws_doubles(E->T, H.nelem, H.has_free, H.nnb), 0.0);
  E->lds.assign(sgt::lds_bytes(E->T, H.nelem, H.has_free, H.nnb) / 8 + 8, 0.0);
  for (int d = 0; d < E->T.ND; d++) E->qpos[E->T.d_gid[d]] = E->T.d_qpos0[d];
  for (int e2 = 0; e2 < H.nelem; e2++) E->qpos[H.elem_qpos0 + e2] = E->P.elem[(size_t)SGE_QPOS0 * H.nelem + e2];
  if (H.has_free)
    for (int c = 0; c < 7; c++) E->qpos[H.free_qadr + c] = H.free_q0[c];
  E->flags = E->touch = E->ncon = E->nefc = E->iters = 0;
  E->touchw[0] = E->touchw[1] = 0;
  return E;
}
void temu_free(TreeEmu* E) { delete E; }
int temu_nv(TreeEmu* E) { return E->P.h.nv; }
int temu_nq(TreeEmu* E) { return E->P.h.nq; }
int temu_nu(TreeEmu* E) { return E->P.h.nu; }
int temu_nsens(TreeEmu* E) { return E->P.h.nsensordata; }
size_t temu_lds_bytes(TreeEmu* E) { return sgt::lds_bytes(E->T, E->P.h.nelem, E->P.h.has_free, E->P.h.nnb); }
// (debugging: the env's work space and where its arrays sit in it -- scripts/dev/work_diff.py holds the GPU's against it)
double* temu_cws(TreeEmu* E) { return E->cws.data(); }
long long temu_cws_doubles(TreeEmu* E) { return (long long)E->cws.size(); }
int temu_layout(TreeEmu* E, char* buf, int cap) {
  const SgPlanHeader& H = E->P.h;
  sgt::Lds S;
  double* const cw = E->cws.data();
  const long long CW = sgt::cws_row_doubles(E->T.CS);
  double* const g0 = cw + (size_t)SGT_MAXHIT * SGT_HITREC * SGT_RECW + (size_t)SGT_MAXCON * CW + E->T.NMAT;
  sgt::lds_carve(S, E->lds.data(), E->T, H.nelem, H.has_free, g0, nullptr, H.nnb);
  int n = snprintf(buf, cap, "stage 0\ncrow %lld\nMg %lld\n", (long long)SGT_MAXHIT * SGT_HITREC * SGT_RECW, (long long)SGT_MAXHIT * SGT_HITREC * SGT_RECW + (long long)SGT_MAXCON * CW);
#define LAY(f) do { const long long o = (long long)(S.f - cw); if (o >= 0 && o < (long long)E->cws.size()) n += snprintf(buf + n, cap - n, #f " %lld\n", o); } while (0)
  LAY(fs); LAY(fc); LAY(bias); LAY(tenJ); LAY(kd); LAY(qacc); LAY(xpos); LAY(xmat); LAY(xipos); LAY(ximat); LAY(bw); LAY(bal); LAY(ba); LAY(bf); LAY(bn);
  LAY(anchor); LAY(axis); LAY(spos); LAY(L); LAY(Minv); LAY(tmpP); LAY(qe); LAY(ve); LAY(we); LAY(asme); LAY(fse); LAY(bfix); LAY(Rfix); LAY(blim); LAY(Rlim);
  LAY(ke); LAY(einvm); LAY(ecoef); LAY(ecen); LAY(Ifix); LAY(Ilim); LAY(seg); LAY(chs); LAY(Afix); LAY(nbf); LAY(nbb); LAY(nbR); LAY(nbI); LAY(nbA);
#undef LAY
#ifdef SG_DEBUG_WORK
  {
    const long long nl = (long long)(sgt::lds_bytes(E->T, H.nelem, H.has_free, H.nnb) / sizeof(double)), at = (long long)E->cws.size() - nl;
#define LAYL(f) n += snprintf(buf + n, cap - n, "lds." #f " %lld\n", at + (long long)((double*)S.f - E->lds.data()))
    n += snprintf(buf + n, cap - n, "lds.hdr %lld\n", at);
    LAYL(q); LAYL(v); LAYL(warm); LAYL(asm_); LAYL(aF); LAYL(gpos); LAYL(gmat); LAYL(gsz); LAYL(ae); LAYL(ffix); LAYL(flim); LAYL(lrow); LAYL(cf); LAYL(red); LAYL(swc);
    LAYL(of); LAYL(Be); LAYL(Ce); LAYL(frow); LAYL(hit_pair); LAYL(hit_sorted); LAYL(hit_cnt); LAYL(hit_off); LAYL(con_src); LAYL(con_chain); LAYL(icnt); LAYL(csc);
    if (H.has_free) LAYL(einvm);
#undef LAYL
  }
#endif
  n += snprintf(buf + n, cap - n, "end %lld\nCS %d\nCW %lld\n", (long long)E->cws.size(), E->T.CS, CW);
  return n;
}
size_t temu_lds_used(TreeEmu* E) { return sgt::lds_used_bytes(E->T, E->P.h.nelem, E->P.h.has_free, E->P.h.nnb); }
double* temu_qpos(TreeEmu* E) { return E->qpos.data(); }
double* temu_qvel(TreeEmu* E) { return E->qvel.data(); }
double* temu_warm(TreeEmu* E) { return E->warm.data(); }
double* temu_act(TreeEmu* E) { return E->act.data(); }
double* temu_ctrl(TreeEmu* E) { return E->ctrl.data(); }
double* temu_sens(TreeEmu* E) { return E->sens.data(); }
int temu_flags(TreeEmu* E) { return E->flags; }
int temu_ncon(TreeEmu* E) { return E->ncon; }
int temu_nefc(TreeEmu* E) { return E->nefc; }
int temu_iters(TreeEmu* E) { return E->iters; }
int temu_touch_word(TreeEmu* E, int w) { return E->touchw[w & 1]; }
// stiffness by joint / tendon id sets, as sg_set_stiffness
void temu_set_stiffness(TreeEmu* E, double k, const int* jnt, int nj, const int* ten, int nt) {
  std::fill(E->kmask_jnt.begin(), E->kmask_jnt.end(), 0);
  std::fill(E->kmask_ten.begin(), E->kmask_ten.end(), 0);
  for (int i = 0; i < nj; i++) E->kmask_jnt[jnt[i]] = 1;
  for (int i = 0; i < nt; i++) E->kmask_ten[ten[i]] = 1;
  E->kenv = k;
}
// mode 1: reset + forward + nsub steps; mode 0: nsub steps
void temu_run(TreeEmu* E, int mode, int nsub) {
  sgt::TreeArgs A;
  A.H = &E->P.h; A.T = &E->T; A.elem = E->P.elem.data(); A.gpairs = E->P.gpairs.data();
  A.sched = E->P.sched.data(); A.nbtab = E->P.nbtab.data();
  A.qpos = E->qpos.data(); A.qvel = E->qvel.data(); A.warm = E->warm.data(); A.act = E->act.data(); A.ctrl = E->ctrl.data();
  A.kenv = &E->kenv; A.kmask_jnt = E->kmask_jnt.data(); A.kmask_ten = E->kmask_ten.data();
  A.mask = nullptr; A.sens = E->sens.data(); A.sens_stride = E->P.h.nsensordata;
  A.flags = &E->flags; A.touch = &E->touch; A.touch_words = E->touchw; A.ncon = &E->ncon; A.nefc = &E->nefc; A.iters = &E->iters;
  A.cws = E->cws.data(); A.cws_stride = (long long)E->cws.size();
  A.nenv = 1; A.nsub = nsub; A.mode = mode; A.secprof = nullptr;
  // a workgroup's LDS block holds whatever the previous workgroup on that CU left there: every launch starts from poison here
  // (SGT_EMU_POISON = the byte, default 0xff: NaN as a double, -1 as an int), so a word read before it is written shows
  static const int poison = getenv("SGT_EMU_POISON") ? (int)strtol(getenv("SGT_EMU_POISON"), nullptr, 0) : 0xff;
  memset(E->lds.data(), poison, E->lds.size() * sizeof(double));
#ifdef SGT_EMU_SEPARATE
  sgt::sep_pool() = &E->pool;
  for (auto& b : E->pool.lds) memset(b.first, poison, b.second);
#ifdef SGT_EMU_MSAN   // (scripts/sanitize/tree_msan_main.cpp: the LDS block is UNINITIALISED at a launch, and MSan knows it)
  for (auto& b : E->pool.lds) __msan_poison(b.first, b.second);
#endif
#endif
  // the same choice of instantiation as the library's launch (sg_api.hip launch_tree)
  if (E->T.CS == 8) sgt::tree_env<8>(A, 0, E->lds.data());
  else if (E->T.CS == 20) sgt::tree_env<20>(A, 0, E->lds.data());
  else sgt::tree_env<SGT_CHD>(A, 0, E->lds.data());
}

}  // extern "C"

// ---- the two forms of the contact block update on random blocks (tests/test_tree_emu.py): the generic one (mju_QCQP2 as written,
// the oracle's formula) and the one with precomputed constants that the tree sweep runs (sg_math.h contact_block_update_pre)
extern "C" int temu_block_update_check(int n, unsigned seed, double* max_df, double* max_rel, int* n_slide, int* n_rejected) {
  unsigned long long st = seed * 6364136223846793005ULL + 1442695040888963407ULL;
  auto rnd = [&]() { st = st * 6364136223846793005ULL + 1442695040888963407ULL; return (double)((st >> 11) & ((1ULL << 53) - 1)) / (double)(1ULL << 53) * 2.0 - 1.0; };
  int bad = 0;
  *max_df = *max_rel = 0; *n_slide = *n_rejected = 0;
  for (int t = 0; t < n; t++) {
    double J[3][5], A[6], res[3], f0[3], mu[2] = {1.0, 1.0};
    if (t % 3 == 1) { mu[0] = 0.7; mu[1] = 0.4; }
    for (auto& r : J) for (double& x : r) x = rnd();
    const double R = 0.02 + 0.5 * (rnd() + 1), scale = (t % 5 == 0) ? 1e-3 : 1.0;
    int k = 0;
    for (int i = 0; i < 3; i++)
      for (int j = i; j < 3; j++) { double s = 0; for (int q = 0; q < 5; q++) s += J[i][q] * J[j][q]; A[k++] = scale * s + (i == j ? R : 0.0); }
    const double fn = (t % 7 == 0) ? 0.0 : 2.0 * (rnd() + 1);   // some contacts without a normal force yet
    f0[0] = fn; f0[1] = 0.6 * fn * rnd() * mu[0]; f0[2] = 0.6 * fn * rnd() * mu[1];
    for (double& x : res) x = 3.0 * rnd();
    double fa[3] = {f0[0], f0[1], f0[2]}, fb[3] = {f0[0], f0[1], f0[2]}, da[3], db[3], Pe[7];
    const double ca = sgm::contact_block_update(A, res, fa, mu, da);
    sgm::contact_block_constants(A, mu, Pe);
    const double cb = sgm::contact_block_update_pre(A, Pe, res, fb, mu, db);
    const double mag = 1.0 + fabs(fa[0]) + fabs(fa[1]) + fabs(fa[2]);
    for (int i = 0; i < 3; i++) {
      const double d = fabs(fa[i] - fb[i]);
      if (d > *max_df) *max_df = d;
      if (d / mag > *max_rel) *max_rel = d / mag;
      if (d > 1e-9 * mag) bad++;
    }
    if (fabs(ca - cb) > 1e-9 * (1 + fabs(ca))) bad++;
    const double s2 = fa[1] * fa[1] / (mu[0] * mu[0]) + fa[2] * fa[2] / (mu[1] * mu[1]);
    if (fa[0] > 0 && s2 > 0.999999 * fa[0] * fa[0]) (*n_slide)++;
    if (da[0] == 0 && da[1] == 0 && da[2] == 0) (*n_rejected)++;
  }
  return bad;
}
