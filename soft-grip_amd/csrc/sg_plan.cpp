// sg_plan.cpp -- builds the kernel plan from a generic model blob (host side, once per model).
// Counterpart of the part of mujoco_py.load_model_from_path (reference
// environment/manenv.py:27) that turns a parsed model into solver-ready constants.
#include "sg_plan.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>

#include "../../include/softgrip_model.h"

namespace {

struct Blob {
  const char* base;
  size_t n;
  const void* find(const char* name, int dtype, long long* cnt) const {
    const sg_blob_header* h = (const sg_blob_header*)base;
    const char* p = base + sizeof(sg_blob_header);
    for (uint32_t r = 0; r < h->nrec; r++) {
      const sg_blob_record* rec = (const sg_blob_record*)p;
      size_t es = rec->dtype == SG_DT_F64 ? 8 : rec->dtype == SG_DT_I32 ? 4 : 1;
      size_t nb = (size_t)rec->count * es;
      nb += (8 - nb % 8) % 8;
      if (strncmp(rec->name, name, 24) == 0 && (int)rec->dtype == dtype) {
        if (cnt) *cnt = rec->count;
        return p + sizeof(sg_blob_record);
      }
      p += sizeof(sg_blob_record) + nb;
    }
    return nullptr;
  }
};

void quat2mat(double* M, const double* q) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  M[0] = w * w + x * x - y * y - z * z; M[1] = 2 * (x * y - w * z); M[2] = 2 * (x * z + w * y);
  M[3] = 2 * (x * y + w * z); M[4] = w * w - x * x + y * y - z * z; M[5] = 2 * (y * z - w * x);
  M[6] = 2 * (x * z - w * y); M[7] = 2 * (y * z + w * x); M[8] = w * w - x * x - y * y + z * z;
}
void mulmat33(double* r, const double* A, const double* B) {
  double t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  memcpy(r, t, sizeof t);
}
void mulmat3(double* r, const double* M, const double* v) {
  double x = M[0] * v[0] + M[1] * v[1] + M[2] * v[2], y = M[3] * v[0] + M[4] * v[1] + M[5] * v[2], z = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}

}  // namespace

#define FAIL(msg)                  \
  do {                             \
    if (err) *err = (msg);         \
    return false;                  \
  } while (0)
#define NEEDF(var, name)                                                         \
  const double* var = (const double*)B.find(name, SG_DT_F64, &cnt);              \
  if (!var) FAIL(std::string("model blob lacks ") + name)
#define NEEDI(var, name)                                                         \
  const int* var = (const int*)B.find(name, SG_DT_I32, &cnt);                    \
  if (!var) FAIL(std::string("model blob lacks ") + name)

static bool plan_build(const void* blob, size_t nbytes, SgPlan* out, SgTreeDev* tree, std::string* err);
bool sg_plan_build(const void* blob, size_t nbytes, SgPlan* out, std::string* err) { return plan_build(blob, nbytes, out, nullptr, err); }
bool sg_tree_plan_build(const void* blob, size_t nbytes, SgPlan* out, SgTreeDev* tree, std::string* err) {
  if (!tree) {
    if (err) *err = "sg_tree_plan_build: no tree table";
    return false;
  }
  return plan_build(blob, nbytes, out, tree, err);
}

// tree == nullptr: the two-finger class of sg_plan.h (SgPlanHeader::chain); else the chains go into *tree (sg_tree_plan.h)
static bool plan_build(const void* blob, size_t nbytes, SgPlan* out, SgTreeDev* tree, std::string* err) {
  const sg_blob_header* hd = (const sg_blob_header*)blob;
  if (nbytes < sizeof *hd || hd->magic != SG_BLOB_MAGIC || hd->version != SG_BLOB_VERSION || (size_t)hd->total_bytes != nbytes)
    FAIL("not a softgrip model blob");
  Blob B{(const char*)blob, nbytes};
  long long cnt = 0;
  NEEDF(opt_d, "opt_d");
  NEEDI(opt_i, "opt_i");
  const long long n_opt_i = cnt;
  NEEDF(body_pos, "body_pos");
  const int nbody = (int)(cnt / 3);
  NEEDF(body_quat, "body_quat"); NEEDF(body_ipos, "body_ipos"); NEEDF(body_imat, "body_imat"); NEEDF(body_mass, "body_mass");
  NEEDF(body_invweight0, "body_invweight0");
  int free_jnt = -1;   // tree plans: the joint id of the object's free joint (soft_experiments_softball.xml:8), else -1
  NEEDF(jnt_pos, "jnt_pos");
  const int nv = (int)(cnt / 3);   // joints (= dofs = positions unless there is a free joint: 7 positions, 6 dofs)
  {  // a free joint (7 positions, 6 dofs: mjcf.py compiles it, the oracle runs it) makes joint, position and dof indices differ
    long long nj = 0;
    const int* jt = (const int*)B.find("jnt_type", SG_DT_I32, &nj);
    for (long long j = 0; jt && j < nj; j++)
      if (jt[j] == SG_JNT_FREE) {
        if (!tree) FAIL("the model has a free joint: the two-finger kernels do not run it (the tree pipeline's object block does: DESIGN.md 4.8)");
        if (free_jnt >= 0) FAIL("more than one free joint");
        free_jnt = (int)j;
      }
  }
  NEEDF(jnt_axis, "jnt_axis"); NEEDF(jnt_range, "jnt_range"); NEEDF(jnt_stiffness, "jnt_stiffness"); NEEDF(jnt_margin, "jnt_margin");
  NEEDF(jnt_solref, "jnt_solref"); NEEDF(jnt_solimp, "jnt_solimp"); NEEDF(qpos0, "qpos0"); NEEDF(qpos_spring, "qpos_spring");
  const long long nq_model = cnt;   // (count of qpos_spring)
  NEEDF(dof_damping, "dof_damping");
  const long long nv_model = cnt;
  NEEDF(dof_armature, "dof_armature"); NEEDF(dof_invweight0, "dof_invweight0");
  // joint -> first position / first dof: one free joint shifts everything behind it by 6 / 5
  auto JQ = [&](int j) { return (free_jnt >= 0 && j > free_jnt) ? j + 6 : j; };
  auto JD = [&](int j) { return (free_jnt >= 0 && j > free_jnt) ? j + 5 : j; };
  if (nq_model != nv + (free_jnt >= 0 ? 6 : 0) || nv_model != nv + (free_jnt >= 0 ? 5 : 0)) FAIL("position / dof counts do not match the joints");
  NEEDF(geom_size, "geom_size");
  const int ngeom = (int)(cnt / 3);
  NEEDF(geom_pos, "geom_pos"); NEEDF(geom_quat, "geom_quat"); NEEDF(geom_friction, "geom_friction"); NEEDF(geom_solref, "geom_solref");
  NEEDF(geom_solimp, "geom_solimp"); NEEDF(geom_solmix, "geom_solmix"); NEEDF(geom_margin, "geom_margin"); NEEDF(geom_gap, "geom_gap");
  NEEDF(geom_rbound, "geom_rbound");
  NEEDF(site_pos, "site_pos");
  const int nsite = (int)(cnt / 3);
  NEEDF(site_quat, "site_quat");
  NEEDF(tendon_stiffness, "tendon_stiffness");
  const int ntendon = (int)cnt;
  NEEDF(tendon_damping, "tendon_damping"); NEEDF(tendon_lengthspring, "tendon_lengthspring"); NEEDF(tendon_length0, "tendon_length0");
  NEEDF(tendon_invweight0, "tendon_invweight0"); NEEDF(wrap_prm, "wrap_prm");
  NEEDF(eq_solref, "eq_solref");
  const int neq = (int)(cnt / 2);
  NEEDF(eq_solimp, "eq_solimp"); NEEDF(eq_data, "eq_data");
  NEEDF(actuator_timeconst, "actuator_timeconst");
  const int nu = (int)cnt;
  NEEDF(actuator_gain, "actuator_gain"); NEEDF(actuator_bias, "actuator_bias"); NEEDF(actuator_gear, "actuator_gear");
  NEEDI(body_parentid, "body_parentid"); NEEDI(body_weldid, "body_weldid"); NEEDI(body_jntadr, "body_jntadr"); NEEDI(body_jntnum, "body_jntnum");
  NEEDI(body_geomadr, "body_geomadr"); NEEDI(body_geomnum, "body_geomnum"); NEEDI(jnt_type, "jnt_type"); NEEDI(jnt_limited, "jnt_limited");
  NEEDI(geom_type, "geom_type"); NEEDI(geom_bodyid, "geom_bodyid"); NEEDI(geom_contype, "geom_contype"); NEEDI(geom_conaffinity, "geom_conaffinity");
  NEEDI(geom_condim, "geom_condim"); NEEDI(geom_priority, "geom_priority"); NEEDI(site_bodyid, "site_bodyid");
  NEEDI(tendon_adr, "tendon_adr"); NEEDI(tendon_num, "tendon_num"); NEEDI(wrap_type, "wrap_type"); NEEDI(wrap_objid, "wrap_objid");
  NEEDI(eq_type, "eq_type"); NEEDI(eq_obj1id, "eq_obj1id"); NEEDI(eq_obj2id, "eq_obj2id"); NEEDI(actuator_trnid, "actuator_trnid");
  NEEDI(sensor_type, "sensor_type");
  const int nsensor = (int)cnt;
  NEEDI(sensor_objid, "sensor_objid"); NEEDI(sensor_adr, "sensor_adr");
  (void)geom_quat; (void)geom_priority;
  if (tree) memset(tree, 0, sizeof *tree);

  SgPlan& P = *out;
  P = SgPlan();
  SgPlanHeader& H = P.h;
  memset(&H, 0, sizeof H);
  H.nv = (int)nv_model; H.nq = (int)nq_model; H.njnt = nv; H.nu = nu; H.nsensordata = 3 * nsensor; H.ntendon = ntendon;
  H.timestep = opt_d[0]; memcpy(H.gravity, opt_d + 1, 24); H.tolerance = opt_d[4]; H.impratio = opt_d[5]; H.meaninertia = opt_d[6];
  H.iterations = opt_i[0];
  H.pgs_scale = 1.0 / (H.meaninertia * (H.nv > 1 ? H.nv : 1));
  if (H.impratio != 1.0) FAIL("impratio != 1 is not supported by the kernels");

  // world poses of world-welded (static) bodies
  std::vector<double> wpos(3 * nbody, 0.0), wmat(9 * nbody, 0.0), wquat(4 * nbody, 0.0);
  wquat[0] = 1; wmat[0] = wmat[4] = wmat[8] = 1;
  auto qmul = [](double* r, const double* a, const double* b) {
    double t[4] = {a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                   a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]};
    memcpy(r, t, sizeof t);
  };
  for (int b = 1; b < nbody; b++) {
    if (body_weldid[b] != 0) continue;
    int p = body_parentid[b];
    double t[3];
    mulmat3(t, &wmat[9 * p], body_pos + 3 * b);
    for (int c = 0; c < 3; c++) wpos[3 * b + c] = wpos[3 * p + c] + t[c];
    qmul(&wquat[4 * b], &wquat[4 * p], body_quat + 4 * b);
    quat2mat(&wmat[9 * b], &wquat[4 * b]);
  }
  std::vector<int> nchild(nbody, 0);
  for (int b = 1; b < nbody; b++) nchild[body_parentid[b]]++;

  // ---- classify moving bodies: elements vs chain bodies ----
  std::vector<int> is_elem(nbody, 0), chain_of(nbody, -1), cbidx(nbody, -1);
  int first_elem = -1, nelem = 0, free_body = -1;
  for (int b = 1; b < nbody && free_jnt >= 0; b++)
    if (body_jntnum[b] > 0 && body_jntadr[b] == free_jnt) free_body = b;
  if (free_jnt >= 0 && (free_body < 0 || body_jntnum[free_body] != 1 || body_parentid[free_body] != 0)) FAIL("a free joint must be the only joint of a child of the world");
  for (int b = 1; b < nbody; b++) {
    if (body_weldid[b] == 0) continue;
    bool slider = body_jntnum[b] == 1 && jnt_type[body_jntadr[b]] == SG_JNT_SLIDE;
    if (slider) {
      if ((body_weldid[body_parentid[b]] != 0 && body_parentid[b] != free_body) || nchild[b] != 0)
        FAIL("slide joints are only supported on leaf bodies with a static parent (tree plans: or the free body)");
      if (body_geomnum[b] != 1 || geom_type[body_geomadr[b]] != SG_GEOM_CAPSULE) FAIL("element bodies must carry exactly one capsule");
      if (first_elem < 0) first_elem = b;
      if (b != first_elem + nelem) FAIL("element bodies must be contiguous");
      if (body_jntadr[b] != body_jntadr[first_elem] + nelem) FAIL("element dofs must be contiguous");
      is_elem[b] = 1;
      nelem++;
    }
  }
  if (nelem == 0) FAIL("model has no composite elements");
  H.nelem = nelem;
  const int elem_jnt0 = body_jntadr[first_elem];   // joint id of the first element (kmask_jnt, tendon wraps and equalities speak joint ids)
  H.elem_jnt0 = elem_jnt0; H.elem_dof0 = JD(elem_jnt0); H.elem_qpos0 = JQ(elem_jnt0);
  if (elem_jnt0 + nelem != nv) FAIL("element dofs must be the last dofs of the model");
  H.has_free = free_jnt >= 0;
  if (H.has_free) {
    for (int e = 0; e < nelem; e++)
      if (body_parentid[first_elem + e] != free_body) FAIL("with a free joint every composite element must hang off the free body");
    if (free_jnt != elem_jnt0 - 1 || free_body != first_elem - 1) FAIL("the free body must come right before the composite elements");
    H.free_jnt = free_jnt; H.free_qadr = free_jnt; H.free_dadr = free_jnt;   // (every joint before it is scalar)
    memcpy(H.free_q0, qpos0 + free_jnt, 56);
    H.free_mass = body_mass[free_body];
    memcpy(H.free_com, body_ipos + 3 * free_body, 24); memcpy(H.free_inertia, body_imat + 9 * free_body, 72);
    H.free_binvw = body_invweight0[2 * free_body];
  }

  int nchain = 0;
  auto lim_kb = [&](int j, double* K, double* Bd) {
    const double *sr = jnt_solref + 2 * j, *si = jnt_solimp + 5 * j;
    double dmax = fmin(0.9999, fmax(1e-4, si[1]));
    if (sr[0] > 0 && sr[1] > 0) {
      double tc = fmax(sr[0], 2 * H.timestep);
      *K = 1 / fmax(1e-15, dmax * dmax * tc * tc * sr[1] * sr[1]);
      *Bd = 2 / fmax(1e-15, dmax * tc);
    } else {
      *K = -sr[0] / fmax(1e-15, dmax * dmax);
      *Bd = -sr[1] / fmax(1e-15, dmax);
    }
  };
  std::vector<int> tb_of(nbody, -1);   // tree mode: flat chain-body index of a model body
  if (tree) {
    SgTreeDev& T = *tree;
    for (int b = 1; b < nbody; b++) {
      if (body_weldid[b] == 0 || is_elem[b] || b == free_body) continue;
      if (b >= first_elem) FAIL("chain bodies must precede the composite elements");
      if (free_jnt >= 0 && body_jntadr[b] + body_jntnum[b] > free_jnt) FAIL("finger joints must precede the free joint");
      if (body_jntnum[b] < 1) FAIL("a moving chain body without a joint");
      for (int k = 0; k < body_jntnum[b]; k++)
        if (jnt_type[body_jntadr[b] + k] != SG_JNT_HINGE) FAIL("chain bodies may only have hinge joints");
      const int p = body_parentid[b];
      int c;
      if (body_weldid[p] == 0) {  // new chain
        if (T.K == SGT_MAXCH) FAIL("more finger chains than the tree pipeline holds");
        c = T.K++;
        T.c_body0[c] = T.NB; T.c_nbody[c] = 0; T.c_dof0[c] = T.ND; T.c_ndof[c] = 0;
        memcpy(T.c_root_pos[c], &wpos[3 * p], 24); memcpy(T.c_root_quat[c], &wquat[4 * p], 32);
      } else {
        c = chain_of[p];
        if (c < 0 || c != T.K - 1 || tb_of[p] != T.NB - 1) FAIL("finger chains must be serial (no branching) and listed one after the other");
      }
      if (T.NB == SGT_MAXB) FAIL("more chain bodies than the tree pipeline holds");
      const int tb = T.NB++;
      chain_of[b] = c; cbidx[b] = T.c_nbody[c]++; tb_of[b] = tb;
      T.b_chain[tb] = c; T.b_njnt[tb] = body_jntnum[b]; T.b_dof0[tb] = T.ND;
      memcpy(T.b_pos[tb], body_pos + 3 * b, 24); memcpy(T.b_quat[tb], body_quat + 4 * b, 32);
      memcpy(T.b_ipos[tb], body_ipos + 3 * b, 24); memcpy(T.b_imat[tb], body_imat + 9 * b, 72);
      T.b_mass[tb] = body_mass[b]; T.b_invw[tb] = body_invweight0[2 * b];
      for (int k = 0; k < body_jntnum[b]; k++) {
        const int j = body_jntadr[b] + k;
        if (T.ND == SGT_MAXD) FAIL("more chain dofs than the tree pipeline holds");
        if (T.c_ndof[c] == SGT_CHD) FAIL("a finger chain has more dofs than the tree pipeline holds");
        const int d = T.ND++;
        T.c_ndof[c]++;
        if (d > 0 && T.d_chain[d - 1] == c && j != T.d_gid[d - 1] + 1) FAIL("chain dofs must be contiguous");
        T.d_body[d] = tb; T.d_chain[d] = c; T.d_limited[d] = jnt_limited[j]; T.d_gid[d] = j;
        memcpy(T.d_axis[d], jnt_axis + 3 * j, 24); memcpy(T.d_pos[d], jnt_pos + 3 * j, 24);
        T.d_qpos0[d] = qpos0[j]; T.d_range[d][0] = jnt_range[2 * j]; T.d_range[d][1] = jnt_range[2 * j + 1]; T.d_margin[d] = jnt_margin[j];
        T.d_damping[d] = dof_damping[j]; T.d_armature[d] = dof_armature[j]; T.d_stiffness[d] = jnt_stiffness[j]; T.d_springref[d] = qpos_spring[j];
        T.d_invw[d] = dof_invweight0[j];
        lim_kb(j, &T.d_limK[d], &T.d_limB[d]);
        memcpy(T.d_solimp[d], jnt_solimp + 5 * j, 40);
      }
      T.b_nabove[tb] = T.c_ndof[c];
      for (int k = 0; k < body_geomnum[b]; k++) {
        const int g = body_geomadr[b] + k;
        if (geom_type[g] != SG_GEOM_BOX) FAIL("chain bodies may only carry box geoms");
        if (T.NG == SGT_MAXG) FAIL("more finger boxes than the tree pipeline holds");
        const int gi = T.NG++;
        T.g_body[gi] = tb; T.g_id[gi] = g;
        memcpy(T.g_pos[gi], geom_pos + 3 * g, 24);
        quat2mat(T.g_mat[gi], geom_quat + 4 * g);
        memcpy(T.g_size[gi], geom_size + 3 * g, 24);
        T.g_rbound[gi] = geom_rbound[g];
      }
    }
    if (T.K == 0) FAIL("model has no finger chain");
    if (T.NG > 64) FAIL("more than 64 finger boxes (the contact read-out has 64 bits)");
    T.CS = 0;
    for (int c = 0; c < T.K; c++) T.CS = std::max(T.CS, T.c_ndof[c]);
    // every chain's vectors and matrix blocks are padded to ONE stride, and the stride is one of the kernel's three instantiations
    // (sg_tree.hip: 8, 20, SGT_CHD = 24): the kernel's CS is then a compile-time constant -- index arithmetic folds into immediates,
    // the per-chain loops unroll without guards (r04; until then a multiple of four, the kernel carrying it as a run-time value)
    T.CS = T.CS <= 8 ? 8 : (T.CS <= 20 ? 20 : SGT_CHD);
    for (int c = 0; c < T.K; c++) T.c_mat0[c] = c * T.CS * T.CS;
    T.NMAT = T.K * T.CS * T.CS;
    if (T.ND + nelem + (free_jnt >= 0 ? 1 : 0) != nv) FAIL("the model has dofs that belong neither to a finger chain nor to a composite element");
  }
  for (int b = 1; b < nbody && !tree; b++) {
    if (body_weldid[b] == 0 || is_elem[b]) continue;
    if (b >= first_elem) FAIL("chain bodies must precede the composite elements");
    for (int k = 0; k < body_jntnum[b]; k++)
      if (jnt_type[body_jntadr[b] + k] != SG_JNT_HINGE) FAIL("chain bodies may only have hinge joints");
    if (body_jntnum[b] < 1 || body_jntnum[b] > 2) FAIL("chain bodies need 1 or 2 hinge joints");
    int p = body_parentid[b];
    if (body_weldid[p] == 0) {  // new chain
      if (nchain == SG_MAXCH) FAIL("more than 2 finger chains");
      chain_of[b] = nchain; cbidx[b] = 0;
      SgChain& C = H.chain[nchain++];
      C.nbody = 1;
      memcpy(C.root_pos, &wpos[3 * p], 24);
      memcpy(C.root_mat, &wmat[9 * p], 72);
      C.dof0 = body_jntadr[b];
    } else {
      int c = chain_of[p];
      if (c < 0 || cbidx[p] != H.chain[c].nbody - 1) FAIL("finger chains must be serial (no branching)");
      if (H.chain[c].nbody == SG_CB) FAIL("finger chain longer than 2 bodies");
      chain_of[b] = c; cbidx[b] = H.chain[c].nbody++;
    }
    SgChain& C = H.chain[chain_of[b]];
    int bi = cbidx[b];
    memcpy(C.b_pos[bi], body_pos + 3 * b, 24); memcpy(C.b_quat[bi], body_quat + 4 * b, 32);
    memcpy(C.b_ipos[bi], body_ipos + 3 * b, 24); memcpy(C.b_imat[bi], body_imat + 9 * b, 72);
    C.b_mass[bi] = body_mass[b]; C.b_invw_tran[bi] = body_invweight0[2 * b];
    C.b_njnt[bi] = body_jntnum[b]; C.b_dof0[bi] = C.ndof;
    for (int k = 0; k < body_jntnum[b]; k++) {
      int j = body_jntadr[b] + k, d = C.ndof++;
      if (d >= SG_CD) FAIL("finger chain has more than 4 dofs");
      if (j != C.dof0 + d) FAIL("chain dofs must be contiguous");
      memcpy(C.j_axis[d], jnt_axis + 3 * j, 24); memcpy(C.j_pos[d], jnt_pos + 3 * j, 24);
      C.qpos0[d] = qpos0[j]; C.range[d][0] = jnt_range[2 * j]; C.range[d][1] = jnt_range[2 * j + 1]; C.jmargin[d] = jnt_margin[j];
      C.damping[d] = dof_damping[j]; C.armature[d] = dof_armature[j]; C.stiffness[d] = jnt_stiffness[j]; C.springref[d] = qpos_spring[j];
      C.invw[d] = dof_invweight0[j]; C.limited[d] = jnt_limited[j]; C.d_body[d] = bi;
      const double *sr = jnt_solref + 2 * j, *si = jnt_solimp + 5 * j;
      double dmax = fmin(0.9999, fmax(1e-4, si[1]));
      if (sr[0] > 0 && sr[1] > 0) {
        double tc = fmax(sr[0], 2 * H.timestep);
        C.lim_K[d] = 1 / fmax(1e-15, dmax * dmax * tc * tc * sr[1] * sr[1]);
        C.lim_B[d] = 2 / fmax(1e-15, dmax * tc);
      } else {
        C.lim_K[d] = -sr[0] / fmax(1e-15, dmax * dmax);
        C.lim_B[d] = -sr[1] / fmax(1e-15, dmax);
      }
      memcpy(C.lim_solimp[d], si, 40);
    }
    for (int k = 0; k < body_geomnum[b]; k++) {
      int g = body_geomadr[b] + k;
      if (geom_type[g] != SG_GEOM_BOX) FAIL("chain bodies may only carry box geoms");
      if (C.ngeom == SG_CG) FAIL("more than 2 box geoms on a finger chain");
      int gi = C.ngeom++;
      C.g_body[gi] = bi; C.g_id[gi] = g;
      memcpy(C.g_pos[gi], geom_pos + 3 * g, 24);
      quat2mat(C.g_mat[gi], geom_quat + 4 * g);
      memcpy(C.g_size[gi], geom_size + 3 * g, 24);
      C.g_rbound[gi] = geom_rbound[g];
    }
  }
  H.nchain = nchain;
  if (nchain == 0 && !tree) FAIL("model has no finger chain");
  // the kernels are compiled for one chain topology: SG_CB bodies with SG_CJ hinges each
  for (int c = 0; c < nchain; c++) {
    if (H.chain[c].nbody != SG_CB) FAIL("finger chains must have exactly 2 moving bodies");
    // (the contact read-out numbers the finger boxes 2 * chain + box: with another count per chain that is not the geom-id order the
    //  C ABI documents for touch_out / sg_get_touch_words -- such a gripper runs in the tree pipeline, which numbers the boxes flat)
    if (H.chain[c].ngeom != SG_CG) FAIL("finger chains must carry exactly 2 box geoms");
    for (int bi = 0; bi < SG_CB; bi++)
      if (H.chain[c].b_njnt[bi] != SG_CJ) FAIL("finger chain bodies must have exactly 2 hinge joints");
  }

  // ---- elements ----
  P.elem.assign((size_t)SGE_NFIELD * nelem, 0.0);
  P.elem_geom.resize(nelem);
  P.elem_dofmap.resize(nelem);
  auto E = [&](int f, int e) -> double& { return P.elem[(size_t)f * nelem + e]; };
  for (int e = 0; e < nelem; e++) {
    int b = first_elem + e, j = elem_jnt0 + e, jd = JD(j), jq = JQ(j), g = body_geomadr[b], p = body_parentid[b];
    P.elem_geom[e] = g; P.elem_dofmap[e] = jd;
    // body frame in the world at q = qpos0 -- in the frame of the FREE body when the elements hang off one (H.has_free: every
    // "world" field below is then local to that body, and the kernel turns it with the body's pose)
    double bp[3], bq[4], bm[9], t[3];
    static const double I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Z3[3] = {0, 0, 0}, Q1[4] = {1, 0, 0, 0};
    const double *pm = p == free_body ? I9 : &wmat[9 * p], *pp = p == free_body ? Z3 : &wpos[3 * p], *pq = p == free_body ? Q1 : &wquat[4 * p];
    mulmat3(t, pm, body_pos + 3 * b);
    for (int c = 0; c < 3; c++) bp[c] = pp[c] + t[c];
    qmul(bq, pq, body_quat + 4 * b);
    quat2mat(bm, bq);
    {  // the element's centre of mass at q = qpos0 and its inertia about it, same frame (the free object's mass matrix and bias)
      double kc[3], RI[9], Rt[9], Iw[9];
      mulmat3(kc, bm, body_ipos + 3 * b);
      mulmat33(RI, bm, body_imat + 9 * b);
      for (int a = 0; a < 3; a++)
        for (int c = 0; c < 3; c++) Rt[3 * a + c] = bm[3 * c + a];
      mulmat33(Iw, RI, Rt);
      E(SGE_KX, e) = bp[0] + kc[0]; E(SGE_KY, e) = bp[1] + kc[1]; E(SGE_KZ, e) = bp[2] + kc[2];
      E(SGE_I00, e) = Iw[0]; E(SGE_I01, e) = Iw[1]; E(SGE_I02, e) = Iw[2]; E(SGE_I11, e) = Iw[4]; E(SGE_I12, e) = Iw[5]; E(SGE_I22, e) = Iw[8];
    }
    double ax[3], gp[3], gm[9], gl[9];
    mulmat3(ax, bm, jnt_axis + 3 * j);
    mulmat3(t, bm, geom_pos + 3 * g);
    for (int c = 0; c < 3; c++) gp[c] = bp[c] + t[c];
    quat2mat(gl, geom_quat + 4 * g);
    mulmat33(gm, bm, gl);
    E(SGE_AX, e) = ax[0]; E(SGE_AY, e) = ax[1]; E(SGE_AZ, e) = ax[2];
    E(SGE_GX, e) = gp[0]; E(SGE_GY, e) = gp[1]; E(SGE_GZ, e) = gp[2];
    E(SGE_CX, e) = gm[2]; E(SGE_CY, e) = gm[5]; E(SGE_CZ, e) = gm[8];
    E(SGE_MASS, e) = body_mass[b]; E(SGE_ARMATURE, e) = dof_armature[jd]; E(SGE_DAMPING, e) = dof_damping[jd];
    E(SGE_K0, e) = jnt_stiffness[j]; E(SGE_SPRINGREF, e) = qpos_spring[jq]; E(SGE_QPOS0, e) = qpos0[jq];
    E(SGE_INVW, e) = dof_invweight0[jd]; E(SGE_BINVW, e) = body_invweight0[2 * b];
    if (jnt_limited[j]) {
      if (!tree) FAIL("limited element sliders are not supported");
      double K, Bd;
      lim_kb(j, &K, &Bd);
      if (H.nlimited_elem == 0) { H.lime_K = K; H.lime_B = Bd; memcpy(H.lime_solimp, jnt_solimp + 5 * j, 40); H.lime_margin = jnt_margin[j]; }
      else if (K != H.lime_K || Bd != H.lime_B || memcmp(H.lime_solimp, jnt_solimp + 5 * j, 40) || H.lime_margin != jnt_margin[j])
        FAIL("limited element sliders must share solref / solimp / margin");
      H.nlimited_elem++;
      E(SGE_LIMITED, e) = 1; E(SGE_RLO, e) = jnt_range[2 * j]; E(SGE_RHI, e) = jnt_range[2 * j + 1];
    }
    if (e == 0) { H.cap_radius = geom_size[3 * g]; H.cap_hl = geom_size[3 * g + 1]; H.cap_rbound = geom_rbound[g]; }
    else if (geom_size[3 * g] != H.cap_radius || geom_size[3 * g + 1] != H.cap_hl) FAIL("element capsules must share one size");
    if (body_mass[b] <= 0) FAIL("element without mass");
  }

  // ---- tendons ----
  H.t0_id = -1;
  std::vector<int> tree_site_id;   // tree mode: model site id of tree site q
  for (int t = 0; t < ntendon; t++) {
    int a = tendon_adr[t], n = tendon_num[t];
    if (wrap_type[a] == SG_WRAP_JOINT) {
      if (H.t0_id >= 0) FAIL("more than one fixed tendon");
      if (n != nelem) FAIL("the fixed tendon must wrap every element slider exactly once");
      for (int w = 0; w < n; w++) {
        if (wrap_objid[a + w] != elem_jnt0 + w) FAIL("the fixed tendon must list the element sliders in order");
        E(SGE_COEF, w) = wrap_prm[a + w];
      }
      H.t0_id = t; H.t0_k0 = tendon_stiffness[t]; H.t0_damping = tendon_damping[t]; H.t0_lspring = tendon_lengthspring[t];
      H.t0_L0 = tendon_length0[t]; H.eqt_invw = tendon_invweight0[t];
    } else if (tree) {
      SgTreeDev& T = *tree;
      if (n < 2 || n > SGT_MAXTS) FAIL("a spatial tendon needs 2 .. 16 sites");
      int c = -1;
      for (int w = 0; w < n; w++) {
        if (wrap_type[a + w] != SG_WRAP_SITE) FAIL("spatial tendons may only wrap sites");
        const int bs = site_bodyid[wrap_objid[a + w]];
        if (body_weldid[bs] == 0) continue;
        if (chain_of[bs] < 0) FAIL("a tendon site sits on a body that is no finger body");
        if (c >= 0 && chain_of[bs] != c) FAIL("a spatial tendon runs over more than one finger chain");
        c = chain_of[bs];
      }
      if (c < 0) FAIL("a spatial tendon without a moving site");
      if (T.t_has[c]) FAIL("more than one spatial tendon on a finger chain");
      T.t_has[c] = 1; T.t_id[c] = t; T.t_nsite[c] = n;
      T.t_k0[c] = tendon_stiffness[t]; T.t_damping[c] = tendon_damping[t]; T.t_lspring[c] = tendon_lengthspring[t];
      for (int w = 0; w < n; w++) {
        const int sid = wrap_objid[a + w], bs = site_bodyid[sid];
        if (body_weldid[bs] == 0) {
          T.t_site[c][w] = -1;
          double tt[3];
          mulmat3(tt, &wmat[9 * bs], site_pos + 3 * sid);
          for (int q = 0; q < 3; q++) T.t_fixed[c][w][q] = wpos[3 * bs + q] + tt[q];
        } else {
          int found = -1;
          for (int q = 0; q < T.NS; q++)
            if (tree_site_id[q] == sid) found = q;
          if (found < 0) {
            if (T.NS == SGT_MAXS) FAIL("more chain sites than the tree pipeline holds");
            found = T.NS++;
            tree_site_id.push_back(sid);
            T.s_body[found] = tb_of[bs];
            memcpy(T.s_pos[found], site_pos + 3 * sid, 24);
            quat2mat(T.s_mat[found], site_quat + 4 * sid);
          }
          T.t_site[c][w] = found;
        }
      }
    } else {
      if (n != 2) FAIL("spatial tendons must have exactly two sites");
      int s0 = wrap_objid[a], s1 = wrap_objid[a + 1], b0 = site_bodyid[s0], b1 = site_bodyid[s1];
      int sfix = -1, smov = -1;
      if (body_weldid[b0] == 0 && chain_of[b1] >= 0) { sfix = s0; smov = s1; }
      else if (body_weldid[b1] == 0 && chain_of[b0] >= 0) { sfix = s1; smov = s0; }
      else FAIL("spatial tendons must join a static site and a finger site");
      int bm_ = site_bodyid[smov], bf = site_bodyid[sfix];
      SgChain& C = H.chain[chain_of[bm_]];
      if (C.has_ten) FAIL("more than one spatial tendon on a finger chain");
      C.has_ten = 1; C.ten_id = t; C.ten_body = cbidx[bm_];
      memcpy(C.ten_site, site_pos + 3 * smov, 24);
      double tt[3];
      mulmat3(tt, &wmat[9 * bf], site_pos + 3 * sfix);
      for (int c = 0; c < 3; c++) C.ten_fixed[c] = wpos[3 * bf + c] + tt[c];
      C.ten_k0 = tendon_stiffness[t]; C.ten_damping = tendon_damping[t]; C.ten_lspring = tendon_lengthspring[t];
    }
  }
  if (H.t0_id < 0) FAIL("model has no fixed tendon over the elements");
  H.t0_implicit = n_opt_i > 3 && opt_i[3] != 0;
  H.t0_hcT = 0;
  if (H.t0_implicit) {
    for (int c = 0; c < H.nchain; c++)
      if (H.chain[c].has_ten && H.chain[c].ten_damping != 0) FAIL("implicit tendon damping covers the elements' fixed tendon only: a finger tendon has a damper");
    for (int c = 0; tree && c < tree->K; c++)
      if (tree->t_has[c] && tree->t_damping[c] != 0) FAIL("implicit tendon damping covers the elements' fixed tendon only: a finger tendon has a damper");
    double T = 0;
    for (int e = 0; e < nelem; e++) T += E(SGE_COEF, e) * E(SGE_COEF, e) / (E(SGE_MASS, e) + E(SGE_ARMATURE, e) + H.timestep * E(SGE_DAMPING, e));
    H.t0_hcT = H.timestep * H.t0_damping * T;
  }

  // ---- equality rows: [joint-fix of element e, then its neighbour rows (e -> e')] for e in order, [tendon-fix] ----
  auto kb = [&](const double* sr, const double* si, double* K, double* Bd) {
    double dmax = fmin(0.9999, fmax(1e-4, si[1]));
    if (sr[0] > 0 && sr[1] > 0) {
      double tc = fmax(sr[0], 2 * H.timestep);
      *K = 1 / fmax(1e-15, dmax * dmax * tc * tc * sr[1] * sr[1]);
      *Bd = 2 / fmax(1e-15, dmax * tc);
    } else {
      *K = -sr[0] / fmax(1e-15, dmax * dmax);
      *Bd = -sr[1] / fmax(1e-15, dmax);
    }
  };
  if (neq < nelem + 1) FAIL("expected one joint equality per element plus one tendon equality");
  std::vector<int> row_e1, row_e2;  // every joint equality in id order; e2 = -1 for a fix row
  {
    int nfix = 0;
    for (int q = 0; q < neq - 1; q++) {
      if (eq_type[q] != SG_EQ_JOINT) FAIL("only the last equality may be a tendon equality");
      if (memcmp(eq_solref + 2 * q, eq_solref, 16) || memcmp(eq_solimp + 5 * q, eq_solimp, 40)) FAIL("joint equalities must share solref/solimp");
      const int e1 = eq_obj1id[q] - elem_jnt0, e2 = eq_obj2id[q] < 0 ? -1 : eq_obj2id[q] - elem_jnt0;
      if (e1 < 0 || e1 >= nelem) FAIL("joint equalities must act on element sliders");
      if (eq_obj2id[q] < 0) {
        if (e1 != nfix) FAIL("the joint-fix equalities must come in element order");
        if (eq_data[5 * q] != 0) FAIL("joint equality offsets are not supported");
        nfix++;
      } else {
        if (e2 < 0 || e2 >= nelem || e2 == e1) FAIL("neighbour equalities must couple two different element sliders");
        if (e1 != nfix - 1) FAIL("an element's neighbour equalities must follow its joint-fix equality");
        const double* pc = eq_data + 5 * q;
        if (pc[0] != 0 || pc[1] != 1 || pc[2] != 0 || pc[3] != 0 || pc[4] != 0) FAIL("neighbour equalities must be q1 = q2 (polycoef 0 1 0 0 0)");
      }
      row_e1.push_back(e1); row_e2.push_back(e2);
    }
    if (nfix != nelem) FAIL("expected one joint-fix equality per element");
  }
  H.nnb = neq - 1 - nelem;
  H.eq_rounds = 0;
  const int eq_slots = tree ? 64 : SG_EQ_SLOTS;   // blocks per round: a lane each in the tree pipeline, a lane pair of the 16-lane group in the solver
  H.eq_slots = eq_slots;
  if (H.nnb > 0) {
    const int N = nelem, nnb = H.nnb;
    P.nbtab.assign((size_t)9 * N + 3 * nnb, -1);
    int* out_e2 = P.nbtab.data(); int* out_id = out_e2 + 3 * N; int* in_id = out_id + 3 * N; int* r_e1 = in_id + 3 * N; int* r_e2 = r_e1 + nnb;
    int* r_slot = r_e2 + nnb;
    std::vector<int> nout(N, 0), nin(N, 0);
    int k = 0;
    for (size_t q = 0; q < row_e1.size(); q++) {
      const int e1 = row_e1[q], e2 = row_e2[q];
      if (e2 < 0) continue;
      if (nout[e1] >= 3 || nin[e2] >= 3) FAIL("more than three neighbour equalities on one side of an element");
      const int slot = nout[e1] * N + e1;  // workspace slot of the row: (its rank among e1's rows, e1) -- lanes = elements store coalesced
      out_e2[slot] = e2; out_id[slot] = slot; nout[e1]++;
      in_id[nin[e2] * N + e2] = slot; nin[e2]++;
      r_e1[k] = e1; r_e2[k] = e2; r_slot[k] = slot;
      k++;
    }
    // The equality rows in MuJoCo's order are [fix_e, e's neighbour rows (up to three: slots d = 0, 1, 2)] for e = 0, 1, ...: one BLOCK
    // per element, all of whose rows act on slider e.  Blocks that share no slider commute exactly; blocks that share one must
    // keep their order.  List scheduling of the blocks (longest remaining dependency chain first) into rounds of SG_EQ_SLOTS (one block per
    // lane quad of an env's group in the solver): every block sits in a later round than the blocks it depends on, so the rounds in
    // order, all slots of a round at once, ARE the sequential sweep.  A lane runs its block's rows one after the other with slider
    // e's acceleration in a register (until r02 the unit was the row: 53 rounds of one LDS round trip each for softbox; now 24).
    std::vector<int> part((size_t)3 * N, N);  // partner slider of block e's d-th neighbour row, N (the zero word) = no such row
    for (int d = 0; d < 3; d++)
      for (int e = 0; e < N; e++)
        if (out_e2[d * N + e] >= 0) part[(size_t)d * N + e] = out_e2[d * N + e];
    std::vector<std::vector<int>> pred(N), succ(N);
    {
      std::vector<int> last(N, -1);
      for (int e = 0; e < N; e++) {
        const int sl[4] = {e, part[e], part[N + e], part[2 * N + e]};
        for (int t = 0; t < 4; t++) {
          if (sl[t] >= N) continue;
          const int p = last[sl[t]];
          if (p >= 0 && p != e) {
            bool dup = false;
            for (int x : pred[e]) dup = dup || x == p;
            if (!dup) { pred[e].push_back(p); succ[p].push_back(e); }
          }
          last[sl[t]] = e;
        }
      }
    }
    std::vector<int> height(N, 1), round_of(N, -1);
    for (int e = N - 1; e >= 0; e--)
      for (int t : succ[e]) height[e] = height[e] > 1 + height[t] ? height[e] : 1 + height[t];
    int left = N, rnd = 0;
    auto invm = [&](int e) { return 1.0 / (P.elem[(size_t)SGE_MASS * N + e] + P.elem[(size_t)SGE_ARMATURE * N + e]); };
    while (left > 0) {
      std::vector<int> ready;
      for (int e = 0; e < N; e++) {
        if (round_of[e] >= 0) continue;
        bool ok = true;
        for (int p : pred[e]) ok = ok && round_of[p] >= 0 && round_of[p] < rnd;
        if (ok) ready.push_back(e);
      }
      std::stable_sort(ready.begin(), ready.end(), [&](int a, int b) { return height[a] > height[b]; });
      for (int g = 0; g < eq_slots; g++) {
        SgEqSlot sl;
        sl.e = sl.p[0] = sl.p[1] = sl.p[2] = N;  // idle slot: the zero word and the dummy records
        if (g < (int)ready.size()) {
          const int e = ready[g];
          round_of[e] = rnd; left--;
          sl.e = e;
          for (int d = 0; d < 3; d++) sl.p[d] = part[(size_t)d * N + e];
        }
        P.sched.push_back(sl);
      }
      rnd++;
    }
    H.eq_rounds = rnd;
    for (int e = 1; e < N && !tree; e++)
      if (invm(e) != invm(0)) FAIL("neighbour-row models need elements of equal mass (the solver keeps 1 / m as a constant)");
    for (int e = 0; e < N && !tree; e++)
      if (P.elem[(size_t)SGE_COEF * N + e] != 1.0) FAIL("neighbour-row models need the element tendon with coefficients 1 (the solver tracks the sum of the slider accelerations)");
  }
  kb(eq_solref, eq_solimp, &H.eqj_K, &H.eqj_B);
  memcpy(H.eqj_solimp, eq_solimp, 40);
  {
    const int qt = neq - 1;
    if (eq_type[qt] != SG_EQ_TENDON || eq_obj1id[qt] != H.t0_id || eq_data[5 * qt] != 0) FAIL("the last equality must fix the element tendon");
    kb(eq_solref + 2 * qt, eq_solimp + 5 * qt, &H.eqt_K, &H.eqt_B);
    memcpy(H.eqt_solimp, eq_solimp + 5 * qt, 40);
  }

  // ---- actuators and sensors ----
  for (int u = 0; tree && u < nu; u++) {
    SgTreeDev& T = *tree;
    int t = actuator_trnid[u], found = -1;
    for (int c = 0; c < T.K; c++)
      if (T.t_has[c] && T.t_id[c] == t) found = c;
    if (found < 0) FAIL("actuators must act on a finger tendon");
    if (T.a_has[found]) FAIL("more than one actuator on a finger tendon");
    T.a_has[found] = 1; T.a_id[found] = u; T.a_gain[found] = actuator_gain[u]; T.a_tc[found] = actuator_timeconst[u]; T.a_gear[found] = actuator_gear[u];
    memcpy(T.a_bias[found], actuator_bias + 3 * u, 24);
  }
  for (int s = 0; tree && s < nsensor; s++) {
    SgTreeDev& T = *tree;
    const int sid = sensor_objid[s], bs = site_bodyid[sid];
    if (sid < 0 || sid >= nsite || chain_of[bs] < 0) FAIL("sensors must sit on finger bodies");
    if (sensor_type[s] != SG_SENS_ACCELEROMETER && sensor_type[s] != SG_SENS_GYRO) FAIL("unsupported sensor type");
    if (T.NSENS == SGT_MAXSENS) FAIL("more sensors than the tree pipeline holds");
    int found = -1;
    for (int q = 0; q < T.NS; q++)
      if (tree_site_id[q] == sid) found = q;
    if (found < 0) {
      if (T.NS == SGT_MAXS) FAIL("more chain sites than the tree pipeline holds");
      found = T.NS++;
      tree_site_id.push_back(sid);
      T.s_body[found] = tb_of[bs];
      memcpy(T.s_pos[found], site_pos + 3 * sid, 24);
      quat2mat(T.s_mat[found], site_quat + 4 * sid);
    }
    const int k = T.NSENS++;
    T.sn_type[k] = sensor_type[s]; T.sn_site[k] = found; T.sn_adr[k] = sensor_adr[s];
  }
  for (int u = 0; !tree && u < nu; u++) {
    int t = actuator_trnid[u], found = -1;
    for (int c = 0; c < nchain; c++)
      if (H.chain[c].has_ten && H.chain[c].ten_id == t) found = c;
    if (found < 0) FAIL("actuators must act on a finger tendon");
    SgChain& C = H.chain[found];
    if (C.has_act) FAIL("more than one actuator on a finger tendon");
    C.has_act = 1; C.act_id = u; C.act_gain = actuator_gain[u]; C.act_tc = actuator_timeconst[u]; C.act_gear = actuator_gear[u];
    memcpy(C.act_bias, actuator_bias + 3 * u, 24);
  }
  std::map<int, std::pair<int, int>> site_slot;  // site id -> (chain, slot)
  for (int s = 0; !tree && s < nsensor; s++) {
    int site = sensor_objid[s], b = site_bodyid[site];
    if (chain_of[b] < 0) FAIL("sensors must sit on finger bodies");
    auto it = site_slot.find(site);
    if (it == site_slot.end()) {
      SgChain& C = H.chain[chain_of[b]];
      if (C.nsite == SG_CS) FAIL("more than 2 sensor sites on a finger chain");
      int sl = C.nsite++;
      C.s_body[sl] = cbidx[b]; C.s_acc_adr[sl] = -1; C.s_gyro_adr[sl] = -1;
      memcpy(C.s_pos[sl], site_pos + 3 * site, 24);
      quat2mat(C.s_mat[sl], site_quat + 4 * site);
      it = site_slot.emplace(site, std::make_pair(chain_of[b], sl)).first;
    }
    SgChain& C = H.chain[it->second.first];
    if (sensor_type[s] == SG_SENS_ACCELEROMETER) C.s_acc_adr[it->second.second] = sensor_adr[s];
    else if (sensor_type[s] == SG_SENS_GYRO) C.s_gyro_adr[it->second.second] = sensor_adr[s];
    else FAIL("unsupported sensor type");
  }

  // ---- static geoms and contact-parameter uniformity ----
  H.center_geom = H.plane_geom = -1;
  int ref_g1 = -1, ref_g2 = -1;
  auto allowed = [&](int g1, int g2) {
    return (geom_contype[g1] & geom_conaffinity[g2]) || (geom_contype[g2] & geom_conaffinity[g1]);
  };
  auto check_pair = [&](int g1, int g2) -> bool {  // same mixed parameters as the reference pair?
    if (ref_g1 < 0) { ref_g1 = g1; ref_g2 = g2; return true; }
    auto mx = [&](const double* a, int k, int x, int y) { return fmax(a[k * x], a[k * y]); };
    (void)mx;
    for (int k = 0; k < 3; k++)
      if (fmax(geom_friction[3 * g1 + k], geom_friction[3 * g2 + k]) != fmax(geom_friction[3 * ref_g1 + k], geom_friction[3 * ref_g2 + k])) return false;
    for (int g : {g1, g2})
      if (memcmp(geom_solref + 2 * g, geom_solref + 2 * ref_g1, 16) || memcmp(geom_solimp + 5 * g, geom_solimp + 5 * ref_g1, 40) ||
          geom_solmix[g] != geom_solmix[ref_g1] || geom_margin[g] != 0 || geom_gap[g] != 0)
        return false;
    return std::max(geom_condim[g1], geom_condim[g2]) == 3;
  };
  std::vector<int> chain_geoms;
  for (int c = 0; c < nchain; c++)
    for (int k = 0; k < H.chain[c].ngeom; k++) chain_geoms.push_back(H.chain[c].g_id[k]);
  for (int g = 0; tree && g < tree->NG; g++) chain_geoms.push_back(tree->g_id[g]);
  if (chain_geoms.empty()) FAIL("finger chains carry no geoms");
  for (int g : chain_geoms) {
    if (!allowed(g, P.elem_geom[0])) FAIL("finger boxes must be able to collide with the element capsules");
    if (!check_pair(P.elem_geom[0], g)) FAIL("contact parameters must be uniform over all finger/object pairs");
  }
  if (std::max(geom_condim[ref_g1], geom_condim[ref_g2]) != 3) FAIL("only condim 3 contacts are supported");
  for (int e = 1; e < nelem; e++) {
    int g = P.elem_geom[e], g0 = P.elem_geom[0];
    if (geom_contype[g] != geom_contype[g0] || geom_conaffinity[g] != geom_conaffinity[g0] || geom_condim[g] != geom_condim[g0] ||
        memcmp(geom_friction + 3 * g, geom_friction + 3 * g0, 24) || memcmp(geom_solref + 2 * g, geom_solref + 2 * g0, 16) ||
        memcmp(geom_solimp + 5 * g, geom_solimp + 5 * g0, 40) || geom_solmix[g] != geom_solmix[g0] || geom_margin[g] != 0 || geom_gap[g] != 0)
      FAIL("element capsules must share contact parameters");
    if (allowed(g, g0)) FAIL("element capsules must not collide with each other");
  }
  for (int g = 0; g < ngeom; g++) {
    int b = geom_bodyid[g];
    if (body_weldid[b] != 0) continue;
    bool hits_chain = false, hits_elem = allowed(g, P.elem_geom[0]);
    for (int cg : chain_geoms) hits_chain |= allowed(g, cg);
    if (!hits_chain && !hits_elem) continue;
    double gp[3], gm[9], gl[9], t[3];
    mulmat3(t, &wmat[9 * b], geom_pos + 3 * g);
    for (int c = 0; c < 3; c++) gp[c] = wpos[3 * b + c] + t[c];
    quat2mat(gl, geom_quat + 4 * g);
    mulmat33(gm, &wmat[9 * b], gl);
    if (geom_type[g] == SG_GEOM_PLANE) {
      if (H.has_plane) FAIL("more than one static plane");
      H.has_plane = 1; H.plane_geom = g;
      memcpy(H.plane_pos, gp, 24);
      H.plane_normal[0] = gm[2]; H.plane_normal[1] = gm[5]; H.plane_normal[2] = gm[8];
    } else if (geom_type[g] == SG_GEOM_SPHERE) {
      if (hits_elem) FAIL("static spheres colliding with elements are not supported");
      if (H.has_center) FAIL("more than one static sphere");
      if (!check_pair(g, chain_geoms[0])) FAIL("contact parameters must be uniform over all finger/object pairs");
      H.has_center = 1; H.center_geom = g; H.center_radius = geom_size[3 * g];
      memcpy(H.center_pos, gp, 24);
    } else if (geom_type[g] == SG_GEOM_BOX) {
      if (H.nstatic == SG_MAXSTATIC) FAIL("too many static boxes");
      int k = H.nstatic++;
      memcpy(H.st_pos[k], gp, 24); memcpy(H.st_mat[k], gm, 72); memcpy(H.st_size[k], geom_size + 3 * g, 24);
      H.st_rbound[k] = geom_rbound[g];
    } else {
      FAIL("unsupported static geom type");
    }
  }
  if (H.has_free && H.nlimited_elem > 0) FAIL("limited sliders on a free object are not built (the object block sweeps joint-fix rows only)");
  if (H.has_free) {   // the free body's own geoms: the composite's centre sphere (its position stays LOCAL to the body)
    for (int k = 0; k < body_geomnum[free_body]; k++) {
      const int g = body_geomadr[free_body] + k;
      bool hits_chain = false;
      for (int cg : chain_geoms) hits_chain |= allowed(g, cg);
      if (geom_type[g] != SG_GEOM_SPHERE) FAIL("the free body may only carry a sphere");
      if (H.has_center) FAIL("more than one centre sphere");
      if (hits_chain && !check_pair(g, chain_geoms[0])) FAIL("contact parameters must be uniform over all finger/object pairs");
      H.has_center = 1; H.center_on_free = 1; H.center_geom = g; H.center_radius = geom_size[3 * g];
      memcpy(H.center_pos, geom_pos + 3 * g, 24);
    }
    // constants of the object's mass matrix (header comment)
    const int N = nelem;
    for (int k = 0; k < 21; k++) H.obj_BBD[k] = H.obj_BBDh[k] = 0;
    for (int k = 0; k < 6; k++) H.obj_tenB[k] = H.obj_tenBh[k] = 0;
    H.obj_msum = H.free_mass;
    for (int c = 0; c < 3; c++) H.obj_mk0[c] = H.free_mass * H.free_com[c];
    for (int e = 0; e < N; e++) {
      const double m = E(SGE_MASS, e), a[3] = {E(SGE_AX, e), E(SGE_AY, e), E(SGE_AZ, e)}, k0[3] = {E(SGE_KX, e), E(SGE_KY, e), E(SGE_KZ, e)};
      const double Bv[6] = {m * a[0], m * a[1], m * a[2], m * (k0[1] * a[2] - k0[2] * a[1]), m * (k0[2] * a[0] - k0[0] * a[2]), m * (k0[0] * a[1] - k0[1] * a[0])};
      const double D = m + E(SGE_ARMATURE, e), Dh = D + H.timestep * E(SGE_DAMPING, e), co = E(SGE_COEF, e);
      int q = 0;
      for (int r = 0; r < 6; r++)
        for (int c = r; c < 6; c++) { H.obj_BBD[q] += Bv[r] * Bv[c] / D; H.obj_BBDh[q] += Bv[r] * Bv[c] / Dh; q++; }
      for (int r = 0; r < 6; r++) { H.obj_tenB[r] += co * Bv[r] / D; H.obj_tenBh[r] += co * Bv[r] / Dh; }
      H.obj_msum += m;
    }
  }
  // Safe slider range per element: while q stays inside (QLO, QHI) the capsule cannot touch the plane (exact: the lower of its two
  // end spheres) and its bounding sphere cannot touch a static box, so those pairs (legal in the model, never active in the reference
  // scenes) need no narrowphase on the fast path; outside it the env's substep runs on the general contact path (sg_general.h).
  // The distance to a convex static geom is convex in q, so the unsafe set is one interval.
  for (int e = 0; e < nelem && !H.has_free; e++) {   // (a free object's element fields are local: no fixed range exists)
    double ax[3] = {E(SGE_AX, e), E(SGE_AY, e), E(SGE_AZ, e)}, g0[3] = {E(SGE_GX, e), E(SGE_GY, e), E(SGE_GZ, e)}, q0 = E(SGE_QPOS0, e);
    double qlo = -1e30, qhi = 1e30;
    auto clearance = [&](int kind, int k, double q) {  // >0: bounding sphere clear of static geom k
      double c[3] = {g0[0] + ax[0] * (q - q0), g0[1] + ax[1] * (q - q0), g0[2] + ax[2] * (q - q0)};
      if (kind == 0) {  // capsule against the plane: exact (a bounding sphere leaves the bottom corner elements of a box shell ~1 cm of travel)
        const double cax[3] = {E(SGE_CX, e), E(SGE_CY, e), E(SGE_CZ, e)};
        const double dc = (c[0] - H.plane_pos[0]) * H.plane_normal[0] + (c[1] - H.plane_pos[1]) * H.plane_normal[1] + (c[2] - H.plane_pos[2]) * H.plane_normal[2];
        const double da = cax[0] * H.plane_normal[0] + cax[1] * H.plane_normal[1] + cax[2] * H.plane_normal[2];
        return dc - H.cap_hl * fabs(da) - H.cap_radius;
      }
      double t[3] = {c[0] - H.st_pos[k][0], c[1] - H.st_pos[k][1], c[2] - H.st_pos[k][2]}, o2 = 0, in = -1e300;
      for (int a = 0; a < 3; a++) {
        double l = t[0] * H.st_mat[k][a] + t[1] * H.st_mat[k][3 + a] + t[2] * H.st_mat[k][6 + a], ex = fabs(l) - H.st_size[k][a];
        if (ex > 0) o2 += ex * ex;
        if (ex > in) in = ex;
      }
      return (o2 > 0 ? sqrt(o2) : in) - H.cap_rbound;
    };
    for (int kind = 0; kind < 2; kind++)
      for (int k = 0; k < (kind == 0 ? H.has_plane : H.nstatic); k++) {
        if (clearance(kind, k, q0) <= 0) {   // within reach of a static geom already in the reference pose (an object resting on the
          qlo = qhi = q0;                     // ground): no safe range, the env always runs on the general contact path
          continue;
        }
        for (int dir = -1; dir <= 1; dir += 2) {  // march outwards from q0 to bracket the first unsafe q, then bisect
          double lo = q0, hi = q0, step = 0.01;
          bool found = false;
          for (int it = 0; it < 40 && fabs(hi - q0) < 100; it++) {
            hi = q0 + dir * step;
            if (clearance(kind, k, hi) <= 0) { found = true; break; }
            lo = hi; step *= 2;
          }
          if (!found) continue;
          for (int it = 0; it < 60; it++) {
            double mid = 0.5 * (lo + hi);
            if (clearance(kind, k, mid) <= 0) hi = mid; else lo = mid;
          }
          if (dir < 0) qlo = fmax(qlo, lo); else qhi = fmin(qhi, lo);
        }
      }
    E(SGE_QLO, e) = qlo; E(SGE_QHI, e) = qhi;
    if (getenv("SG_PLAN_DEBUG")) fprintf(stderr, "elem %d safe slider range (%g, %g)\n", e, qlo - q0, qhi - q0);
  }
  // ---- the general contact path's candidate pairs (sg_general.h): mj_collision's pair list for this model, as the oracle builds it --
  //      body pairs (b1 < b2) ascending, geoms of b1 outer, geoms of b2 inner; filtered by contype / conaffinity, weld group and the
  //      parent-child rule; geoms of a pair ordered by type.  A pair whose mixed parameters differ from the finger / object pairs'
  //      keeps its place in the list as SGP_UNSUPPORTED: within reach it raises the unsupported-pair flag instead of a contact.
  {
    std::vector<int> ref_of(ngeom, 0);
    int nst = 0;
    for (int g = 0; g < ngeom; g++) {
      const int b = geom_bodyid[g];
      if (body_weldid[b] != 0) continue;
      bool hits_chain = false, hits_elem = allowed(g, P.elem_geom[0]);
      for (int cg : chain_geoms) hits_chain |= allowed(g, cg);
      if (!hits_chain && !hits_elem) continue;
      if (geom_type[g] == 0 /* plane */) ref_of[g] = 1 << 16;
      else if (geom_type[g] == SG_GEOM_SPHERE) ref_of[g] = 3 << 16;
      else if (geom_type[g] == SG_GEOM_BOX) ref_of[g] = (2 << 16) | nst++;   // same order as H.st_* above
    }
    for (int c = 0; c < nchain; c++)
      for (int k = 0; k < H.chain[c].ngeom; k++) ref_of[H.chain[c].g_id[k]] = (4 << 16) | (c * SG_CG + k);
    for (int g = 0; tree && g < tree->NG; g++) ref_of[tree->g_id[g]] = (4 << 16) | g;
    for (int e = 0; e < nelem; e++) ref_of[P.elem_geom[e]] = (5 << 16) | e;
    if (H.has_free && H.has_center) ref_of[H.center_geom] = 3 << 16;
    auto pair_allowed = [&](int g1, int g2) {
      const int b1 = geom_bodyid[g1], b2 = geom_bodyid[g2];
      if (!allowed(g1, g2)) return false;
      const int w1 = body_weldid[b1], w2 = body_weldid[b2];
      if (w1 == w2) return false;
      const int wp1 = body_weldid[body_parentid[w1]], wp2 = body_weldid[body_parentid[w2]];
      if (w1 != 0 && w2 != 0 && (w1 == wp2 || w2 == wp1)) return false;
      return true;
    };
    for (int b1 = 0; b1 < nbody; b1++)
      for (int b2 = b1 + 1; b2 < nbody; b2++)
        for (int i = 0; i < body_geomnum[b1]; i++)
          for (int j = 0; j < body_geomnum[b2]; j++) {
            int g1 = body_geomadr[b1] + i, g2 = body_geomadr[b2] + j;
            if (!pair_allowed(g1, g2)) continue;
            if (geom_type[g1] > geom_type[g2]) std::swap(g1, g2);
            const int t1 = geom_type[g1], t2 = geom_type[g2];
            SgGenPair gp;
            gp.g1 = ref_of[g1]; gp.g2 = ref_of[g2]; gp.pad = 0;
            if (t1 == 0 && t2 == SG_GEOM_CAPSULE) gp.kind = 0;
            else if (t1 == 0 && t2 == SG_GEOM_BOX) gp.kind = 1;
            else if (t1 == SG_GEOM_SPHERE && t2 == SG_GEOM_BOX) gp.kind = 2;
            else if (t1 == SG_GEOM_CAPSULE && t2 == SG_GEOM_BOX) gp.kind = 3;
            else if (t1 == SG_GEOM_BOX && t2 == SG_GEOM_BOX) gp.kind = 4;
            else if (t1 == 0 && t2 == SG_GEOM_SPHERE && tree) gp.kind = 6;   // SGP_PLANE_SPH: the free object's centre sphere on the ground
            else FAIL("unsupported collision pair types");
            if (gp.g1 == 0 || gp.g2 == 0) FAIL("a collision pair involves a geom outside the plan class");
            if (!check_pair(g1, g2)) gp.kind = 5;   // SGP_UNSUPPORTED
            {  // the pair's bounding distance as a float rounded UP (in `pad`): plane pairs margin + rbound(geom2), else the sum of the
               // bounding radii + margin.  A filter on it passes every pair the exact test passes (the tree pipeline's pair walk)
              const double mg = fmax(geom_margin[g1], geom_margin[g2]);
              const double bd = (t1 == 0 ? 0.0 : geom_rbound[g1]) + geom_rbound[g2] + mg;
              float bf = (float)bd;
              if ((double)bf < bd) bf = nextafterf(bf, INFINITY);
              memcpy(&gp.pad, &bf, 4);
            }
            P.gpairs.push_back(gp);
          }
    if (P.gpairs.size() > 60000) FAIL("too many candidate collision pairs");
    H.ngpair = (int)P.gpairs.size();
    if (tree) {
      // block descriptors of the tree pipeline's pair walk, appended to the table at [ngpair + 1 + block]: a block is 64 consecutive
      // pairs (one trip of the wavefront).  kind != 0: every pair of the block is (element capsule | centre sphere | finger box) x
      // finger box -- both geoms' poses are in LDS, the walk runs them in its short loop; kind = 1 - 4: no box - box pair among them
      // and that many distinct finger boxes, g1 = their flat indices a byte each -- such a block is skipped while all its boxes are
      // out of reach of the object's bounding box; kind = -1: not skippable.  kind = 0: the general loop (plane, static-box pairs)
      SgGenPair z;
      z.kind = 0; z.g1 = z.g2 = z.pad = 0;
      P.gpairs.push_back(z);
      for (int p0 = 0; p0 < H.ngpair; p0 += 64) {
        SgGenPair d = z;
        int boxes[4], nb = 0;
        bool pure = true, skippable = true, any = false;
        for (int p = p0; p < H.ngpair && p < p0 + 64 && pure; p++) {
          const SgGenPair& gp = P.gpairs[p];
          if (gp.kind == 5) { pure = false; break; }   // SGP_UNSUPPORTED: the general loop tests its bounding distance and flags the env (never skipped)
          const int k1 = gp.g1 >> 16, k2 = gp.g2 >> 16;
          if (!((gp.kind == 2 || gp.kind == 3 || gp.kind == 4) && k2 == 4 && (k1 == 3 || k1 == 5 || k1 == 4))) { pure = false; break; }
          any = true;
          if (k1 == 4) skippable = false;
          const int b = gp.g2 & 0xFFFF;
          int j = 0;
          while (j < nb && boxes[j] != b) j++;
          if (j == nb) { if (nb == 4 || b > 255) skippable = false; else boxes[nb++] = b; }
        }
        if (pure && any) {
          d.kind = skippable ? nb : -1;
          for (int j = 0; j < nb && skippable; j++) d.g1 |= boxes[j] << (8 * j);
        }
        P.gpairs.push_back(d);
      }
      if (getenv("SG_PLAN_DEBUG")) {
        int nsk = 0, npu = 0, nb = 0;
        for (size_t i = H.ngpair + 1; i < P.gpairs.size(); i++, nb++) { nsk += P.gpairs[i].kind > 0; npu += P.gpairs[i].kind != 0; }
        fprintf(stderr, "pair table: %d pairs, %d blocks (%d in the short loop, %d of them skippable)\n", H.ngpair, nb, npu, nsk);
      }
    }
  }
  // mixed contact parameters of the reference pair
  {
    int g1 = ref_g1, g2 = ref_g2;
    double fr0 = fmax(geom_friction[3 * g1], geom_friction[3 * g2]);
    H.con_mu[0] = H.con_mu[1] = fr0;
    double sr[2], si[5];
    for (int k = 0; k < 2; k++) sr[k] = 0.5 * geom_solref[2 * g1 + k] + 0.5 * geom_solref[2 * g2 + k];
    for (int k = 0; k < 5; k++) si[k] = 0.5 * geom_solimp[5 * g1 + k] + 0.5 * geom_solimp[5 * g2 + k];
    if (!(geom_solref[2 * g1] > 0 && geom_solref[2 * g2] > 0)) FAIL("direct-format contact solref is not supported");
    kb(sr, si, &H.con_K, &H.con_B);
    memcpy(H.con_solimp, si, 40);
    H.con_margin = 0;
  }
  return true;
}
