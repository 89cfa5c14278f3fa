/* sg_oracle.c -- CPU restatement of MuJoCo's mj_step / mj_forward / mj_resetData for the
 * soft-gripper model class.  See sg_oracle.h: TEST INFRASTRUCTURE ONLY, PARITY UNPINNED.
 *
 * What the reference calls (the only physics entry points on its hot path):
 *   sim.step()    -> mj_step        reference environment/manenv.py:49
 *   sim.reset()   -> mj_resetData   reference environment/manenv.py:57
 *   sim.forward() -> mj_forward     reference environment/manenv.py:58
 * MuJoCo itself is a third-party dependency absent from /root/reference (unpinned; the MJCF
 * syntax implies MuJoCo 2.0/2.1).  Every stage below restates MuJoCo's
 * published computation pipeline (stage numbers quoted in the comments)
 * with *general* algorithms -- kinematic tree walk, chain Jacobians, full mass matrix with
 * sparse L'DL, explicit constraint Jacobian, explicit A = J M^-1 J' + R and a Gauss-Seidel
 * sweep over it -- deliberately not sharing the structure exploitation of the HIP kernels,
 * so that agreement between the two is evidence and not tautology.
 *
 * Known, documented deviations from MuJoCo (DESIGN.md "Deviations"):
 *   D1 capsule-box narrowphase is the exact segment/box closest-point construction, not
 *      MuJoCo's mjc_CapsuleBox case analysis (identical for a single nearest-point contact,
 *      may differ in the choice of a second contact for near-parallel configurations);
 *   D2 box-box and plane-box contacts are generated (since r02) by a separating-axis + face-clipping construction and from the
 *      box corners within the margin -- written from the published description of such routines, not a transcription of
 *      mjc_BoxBox / mjc_PlaneBox (their source is not available here); plane-capsule and static-box-capsule likewise.
 *      SGO_WARN_UNSUPPORTED_PAIR is left for geometry outside the model class.
 */
#include "sg_oracle.h"
#include "../include/softgrip_model.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MINVAL 1e-15
#define MAXVAL 1e10
#define MINIMP 1e-4
#define MAXIMP 0.9999
#define MAXCHAIN 24
#define MAXCON 512

/* Sensitivity switches (scripts/d5_sensitivity.py; 0 in every test and everywhere else): each bit flips ONE of the places where this
 * restatement had to resolve something the documentation leaves open, to see whether a different
 * resolution changes a qualitative result (does the explicit volume-tendon damper survive the ball scene's start?).
 *   1  the warmstart saved for the next step is the implicit-damping acceleration the Euler step integrates, not the solver's qacc
 *   2  direct-format solref (-stiffness, -damping) taken literally: K = -solref[0], B = -solref[1], no division by dmax^2 / dmax
 *   4  equality rows use the impedance at zero violation (d0) instead of d(|pos|)
 *   8  the tendon equality's diagApprox is the sum of its dofs' invweights instead of tendon_invweight0
 *  16  no warmstart at all (every solve starts from f = 0) */
static int g_variant = 0;
void sgo_set_variant(int bits) { g_variant = bits; }

enum { EFC_EQUALITY = 0, EFC_LIMIT = 3, EFC_CONTACT_FRICTIONLESS = 5, EFC_CONTACT_ELLIPTIC = 7 };

typedef struct {
  double dist, pos[3], frame[9], friction[5], solref[2], solimp[5], mu, includemargin;
  int dim, geom1, geom2, efc_address;
} contact_t;

struct sgo_model {
  void* blob;
  int nbody, nv, nq, njnt, ngeom, nsite, ntendon, nwrap, neq, nu, nsensor;   /* nq = nv = njnt unless the model has a free joint (7 positions, 6 dofs) */
  double timestep, gravity[3], tolerance, impratio, meaninertia;
  int iterations, nconmax, njmax;
  /* views into blob */
  const double *body_pos, *body_quat, *body_ipos, *body_imat, *body_mass, *body_invweight0;
  const double *jnt_pos, *jnt_axis, *jnt_range, *jnt_stiffness, *jnt_margin, *jnt_solref, *jnt_solimp;
  const double *qpos0, *qpos_spring, *dof_damping, *dof_armature, *dof_invweight0;
  const double *geom_size, *geom_pos, *geom_quat, *geom_friction, *geom_solref, *geom_solimp, *geom_solmix;
  const double *geom_margin, *geom_gap, *geom_rbound;
  const double *site_pos, *site_quat;
  const double *tendon_stiffness, *tendon_damping, *tendon_lengthspring, *tendon_length0, *tendon_invweight0;
  const double *wrap_prm, *eq_solref, *eq_solimp, *eq_data;
  const double *actuator_timeconst, *actuator_gain, *actuator_bias, *actuator_gear;
  const int *body_parentid, *body_weldid, *body_jntadr, *body_jntnum, *body_geomadr, *body_geomnum;
  const int *jnt_type, *jnt_bodyid, *jnt_limited, *dof_parentid;
  const int *geom_type, *geom_bodyid, *geom_contype, *geom_conaffinity, *geom_condim, *geom_priority;
  const int *site_bodyid, *tendon_adr, *tendon_num, *wrap_type, *wrap_objid;
  const int *eq_type, *eq_obj1id, *eq_obj2id, *actuator_trnid, *sensor_type, *sensor_objid, *sensor_adr;
  /* derived at load */
  double* geom_lmat; /* ngeom x 9 local rotation */
  double* site_lmat;
  int npair;
  int* pair; /* npair x 2 candidate geom pairs in MuJoCo's body-pair order */
  /* joint -> first position / first dof, dof -> joint (identity unless the blob brings them: free joints), and per dof / body:
   * rotational? (hinge, free rotation: jp = axis x r, jr = axis; else slide / free translation: jp = axis), the body's dof range */
  int *jnt_qposadr, *jnt_dofadr, *dof_jntid, *dof_rot, *body_dofadr, *body_dofnum;
  int dof_damping_any;
  int implicit_tendon_damping; /* model flag opt_i[3] (0 when the blob has only three): see sgo_step */
};

struct sgo_data {
  const sgo_model* m;
  double time;
  double *qpos, *qvel, *act, *ctrl, *qacc_warmstart, *jnt_stiffness, *tendon_stiffness;
  double *xpos, *xquat, *xmat, *xipos, *ximat, *xanchor, *xaxis, *geom_xpos, *geom_xmat, *site_xpos, *site_xmat;
  double *ten_length, *ten_J, *ten_velocity;
  double *qM, *qLD;
  double *bw, *bv, *bal, *ba; /* per body: ang vel, lin vel of origin, ang acc, lin acc of origin */
  double *qfrc_passive, *qfrc_bias, *qfrc_actuator, *qfrc_smooth, *qacc_smooth, *qfrc_constraint, *qacc;
  double *act_dot, *actuator_force, *sensordata, *tmpv, *tmpv2, *tmpv3;
  contact_t* contact;
  int ncon;
  /* constraints */
  int nefc, efc_cap, jnnz_cap;
  int *efc_type, *efc_id, *J_rowadr, *J_col;
  double *J_val, *efc_pos, *efc_margin, *efc_diagApprox, *efc_R, *efc_D, *efc_KBIP, *efc_vel, *efc_aref, *efc_b,
      *efc_force, *efc_jar;
  double* AR; /* dense nefc x nefc */
  size_t AR_cap;
  int *AR_rownnz, *AR_rowadr, *AR_col, AR_colcap; /* nonzero pattern of AR rows */
  int *dofrow_adr, *dofrow, dofrow_cap, *stamp;
  int *scr_cols, *scr_fill, *scr_rowJk, scr_rowJk_cap;
  double* scr_vals;
  int solver_iter, warnings;
};

/* ------------------------------------------------------------------ small vector helpers */
static inline double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void cross3(double* r, const double* a, const double* b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline void addscl3(double* r, const double* a, double s) { r[0] += a[0] * s; r[1] += a[1] * s; r[2] += a[2] * s; }
static inline void mulmat3(double* r, const double* M, const double* v) {
  double x = M[0] * v[0] + M[1] * v[1] + M[2] * v[2], y = M[3] * v[0] + M[4] * v[1] + M[5] * v[2],
         z = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static inline void mulmatT3(double* r, const double* M, const double* v) {
  double x = M[0] * v[0] + M[3] * v[1] + M[6] * v[2], y = M[1] * v[0] + M[4] * v[1] + M[7] * v[2],
         z = M[2] * v[0] + M[5] * v[1] + M[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
static void mulmat33(double* r, const double* A, const double* B) {
  double t[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  memcpy(r, t, sizeof t);
}
static void quat2mat(double* M, const double* q) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  M[0] = w * w + x * x - y * y - z * z; M[1] = 2 * (x * y - w * z); M[2] = 2 * (x * z + w * y);
  M[3] = 2 * (x * y + w * z); M[4] = w * w - x * x + y * y - z * z; M[5] = 2 * (y * z - w * x);
  M[6] = 2 * (x * z - w * y); M[7] = 2 * (y * z + w * x); M[8] = w * w - x * x - y * y + z * z;
}
static void quatmul(double* r, const double* a, const double* b) {
  double t[4] = {a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                 a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]};
  memcpy(r, t, sizeof t);
}
static int isbad(double x) { return (x != x) || x > MAXVAL || x < -MAXVAL; }

/* ------------------------------------------------------------------ blob access */
static const void* blob_find(const void* blob, const char* name, int dtype, long long* count) {
  const sg_blob_header* h = (const sg_blob_header*)blob;
  const char* p = (const char*)blob + sizeof(sg_blob_header);
  for (uint32_t r = 0; r < h->nrec; r++) {
    const sg_blob_record* rec = (const sg_blob_record*)p;
    size_t es = rec->dtype == SG_DT_F64 ? 8 : rec->dtype == SG_DT_I32 ? 4 : 1;
    size_t nb = (size_t)rec->count * es;
    nb += (8 - nb % 8) % 8;
    if (strncmp(rec->name, name, 24) == 0 && (int)rec->dtype == dtype) {
      if (count) *count = rec->count;
      return p + sizeof(sg_blob_record);
    }
    p += sizeof(sg_blob_record) + nb;
  }
  return NULL;
}

#define GETF(field)                                                              \
  do {                                                                           \
    m->field = (const double*)blob_find(m->blob, #field, SG_DT_F64, &cnt);       \
    if (!m->field) { snprintf(err, errlen, "blob lacks %s", #field); goto fail; } \
  } while (0)
#define GETI(field)                                                              \
  do {                                                                           \
    m->field = (const int*)blob_find(m->blob, #field, SG_DT_I32, &cnt);          \
    if (!m->field) { snprintf(err, errlen, "blob lacks %s", #field); goto fail; } \
  } while (0)

static int pair_allowed(const sgo_model* m, int g1, int g2) {
  int b1 = m->geom_bodyid[g1], b2 = m->geom_bodyid[g2];
  if (!((m->geom_contype[g1] & m->geom_conaffinity[g2]) || (m->geom_contype[g2] & m->geom_conaffinity[g1]))) return 0;
  int w1 = m->body_weldid[b1], w2 = m->body_weldid[b2];
  if (w1 == w2) return 0; /* same weld group (includes both static) */
  int wp1 = m->body_weldid[m->body_parentid[w1]], wp2 = m->body_weldid[m->body_parentid[w2]];
  if (w1 != 0 && w2 != 0 && (w1 == wp2 || w2 == wp1)) return 0; /* parent-child filter */
  return 1;
}

sgo_model* sgo_model_load(const void* blob, size_t nbytes, char* err, size_t errlen) {
  char dummy[8];
  if (!err) { err = dummy; errlen = sizeof dummy; }
  const sg_blob_header* h = (const sg_blob_header*)blob;
  if (nbytes < sizeof *h || h->magic != SG_BLOB_MAGIC || h->version != SG_BLOB_VERSION || (size_t)h->total_bytes != nbytes) {
    snprintf(err, errlen, "not a softgrip model blob");
    return NULL;
  }
  sgo_model* m = (sgo_model*)calloc(1, sizeof *m);
  m->blob = malloc(nbytes);
  memcpy(m->blob, blob, nbytes);
  long long cnt = 0;
  const double* od = (const double*)blob_find(m->blob, "opt_d", SG_DT_F64, &cnt);
  const int* oi = (const int*)blob_find(m->blob, "opt_i", SG_DT_I32, &cnt);
  if (!od || !oi) { snprintf(err, errlen, "blob lacks opt"); goto fail; }
  m->timestep = od[0]; memcpy(m->gravity, od + 1, 24); m->tolerance = od[4]; m->impratio = od[5]; m->meaninertia = od[6];
  m->iterations = oi[0]; m->nconmax = oi[1]; m->njmax = oi[2];
  m->implicit_tendon_damping = cnt > 3 ? oi[3] : 0;
  GETF(body_pos); m->nbody = (int)(cnt / 3);
  GETF(body_quat); GETF(body_ipos); GETF(body_imat); GETF(body_mass); GETF(body_invweight0);
  GETF(jnt_pos); m->njnt = (int)(cnt / 3);
  GETF(jnt_axis); GETF(jnt_range); GETF(jnt_stiffness); GETF(jnt_margin); GETF(jnt_solref); GETF(jnt_solimp);
  GETF(qpos0); m->nq = (int)cnt;
  GETF(qpos_spring);
  GETF(dof_damping); m->nv = (int)cnt;
  GETF(dof_armature); GETF(dof_invweight0);
  GETF(geom_size); m->ngeom = (int)(cnt / 3);
  GETF(geom_pos); GETF(geom_quat); GETF(geom_friction); GETF(geom_solref); GETF(geom_solimp); GETF(geom_solmix);
  GETF(geom_margin); GETF(geom_gap); GETF(geom_rbound);
  GETF(site_pos); m->nsite = (int)(cnt / 3);
  GETF(site_quat);
  GETF(tendon_stiffness); m->ntendon = (int)cnt;
  GETF(tendon_damping); GETF(tendon_lengthspring); GETF(tendon_length0); GETF(tendon_invweight0);
  GETF(wrap_prm); m->nwrap = (int)cnt;
  GETF(eq_solref); m->neq = (int)(cnt / 2);
  GETF(eq_solimp); GETF(eq_data);
  GETF(actuator_timeconst); m->nu = (int)cnt;
  GETF(actuator_gain); GETF(actuator_bias); GETF(actuator_gear);
  GETI(body_parentid); GETI(body_weldid); GETI(body_jntadr); GETI(body_jntnum); GETI(body_geomadr); GETI(body_geomnum);
  GETI(jnt_type); GETI(jnt_bodyid); GETI(jnt_limited); GETI(dof_parentid);
  GETI(geom_type); GETI(geom_bodyid); GETI(geom_contype); GETI(geom_conaffinity); GETI(geom_condim); GETI(geom_priority);
  GETI(site_bodyid); GETI(tendon_adr); GETI(tendon_num); GETI(wrap_type); GETI(wrap_objid);
  GETI(eq_type); GETI(eq_obj1id); GETI(eq_obj2id); GETI(actuator_trnid);
  GETI(sensor_type); m->nsensor = (int)cnt;
  GETI(sensor_objid); GETI(sensor_adr);

  { /* index maps: the blob carries them only when a free joint makes joints, positions and dofs differ */
    const int* qa = (const int*)blob_find(m->blob, "jnt_qposadr", SG_DT_I32, &cnt);
    const int* da = (const int*)blob_find(m->blob, "jnt_dofadr", SG_DT_I32, &cnt);
    const int* dj = (const int*)blob_find(m->blob, "dof_jntid", SG_DT_I32, &cnt);
    if ((!qa || !da || !dj) && (m->nq != m->njnt || m->nv != m->njnt)) { snprintf(err, errlen, "blob lacks the joint address maps"); goto fail; }
    m->jnt_qposadr = (int*)malloc(sizeof(int) * (m->njnt + 1)); m->jnt_dofadr = (int*)malloc(sizeof(int) * (m->njnt + 1));
    m->dof_jntid = (int*)malloc(sizeof(int) * (m->nv + 1)); m->dof_rot = (int*)malloc(sizeof(int) * (m->nv + 1));
    m->body_dofadr = (int*)malloc(sizeof(int) * m->nbody); m->body_dofnum = (int*)calloc(m->nbody, sizeof(int));
    for (int j = 0; j < m->njnt; j++) { m->jnt_qposadr[j] = qa ? qa[j] : j; m->jnt_dofadr[j] = da ? da[j] : j; }
    for (int i = 0; i < m->nv; i++) m->dof_jntid[i] = dj ? dj[i] : i;
    for (int i = 0; i < m->nv; i++) {
      int j = m->dof_jntid[i], ty = m->jnt_type[j];
      if (ty != SG_JNT_HINGE && ty != SG_JNT_SLIDE && ty != SG_JNT_FREE) { snprintf(err, errlen, "unsupported joint type %d", ty); goto fail; }
      m->dof_rot[i] = ty == SG_JNT_HINGE || (ty == SG_JNT_FREE && i - m->jnt_dofadr[j] >= 3);
    }
    for (int b = 0; b < m->nbody; b++) {
      m->body_dofadr[b] = m->body_jntnum[b] > 0 ? m->jnt_dofadr[m->body_jntadr[b]] : -1;
      for (int k = 0; k < m->body_jntnum[b]; k++) {
        int j = m->body_jntadr[b] + k;
        m->body_dofnum[b] += m->jnt_type[j] == SG_JNT_FREE ? 6 : 1;
        if (m->jnt_type[j] == SG_JNT_FREE && (m->body_parentid[b] != 0 || m->body_jntnum[b] != 1)) { snprintf(err, errlen, "a free joint must be the only joint of a child of the world"); goto fail; }
      }
    }
  }
  m->geom_lmat = (double*)malloc(sizeof(double) * 9 * (m->ngeom + 1));
  for (int g = 0; g < m->ngeom; g++) quat2mat(m->geom_lmat + 9 * g, m->geom_quat + 4 * g);
  m->site_lmat = (double*)malloc(sizeof(double) * 9 * (m->nsite + 1));
  for (int s = 0; s < m->nsite; s++) quat2mat(m->site_lmat + 9 * s, m->site_quat + 4 * s);
  for (int i = 0; i < m->nv; i++)
    if (m->dof_damping[i] > 0) m->dof_damping_any = 1;
  /* chain depth check */
  for (int b = 1; b < m->nbody; b++) {
    int n = 0;
    for (int a = b; a > 0; a = m->body_parentid[a]) n += m->body_dofnum[a];
    if (n > MAXCHAIN) { snprintf(err, errlen, "kinematic chain too deep"); goto fail; }
  }
  /* candidate geom pairs: body pairs (b1<b2) ascending, geoms of b1 outer, geoms of b2 inner */
  {
    int cap = 1024;
    m->pair = (int*)malloc(sizeof(int) * 2 * cap);
    for (int b1 = 0; b1 < m->nbody; b1++)
      for (int b2 = b1 + 1; b2 < m->nbody; b2++)
        for (int i = 0; i < m->body_geomnum[b1]; i++)
          for (int j = 0; j < m->body_geomnum[b2]; j++) {
            int g1 = m->body_geomadr[b1] + i, g2 = m->body_geomadr[b2] + j;
            if (!pair_allowed(m, g1, g2)) continue;
            int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
            if (t1 > t2) { int t = g1; g1 = g2; g2 = t; t = t1; t1 = t2; t2 = t; } /* order by type, as mj_collideGeoms */
            int ok = (t1 == SG_GEOM_PLANE && (t2 == SG_GEOM_SPHERE || t2 == SG_GEOM_CAPSULE || t2 == SG_GEOM_BOX)) ||
                     ((t1 == SG_GEOM_SPHERE || t1 == SG_GEOM_CAPSULE || t1 == SG_GEOM_BOX) && t2 == SG_GEOM_BOX);
            if (!ok) { snprintf(err, errlen, "unsupported collision pair types %d-%d", t1, t2); goto fail; }
            int cd = m->geom_condim[g1] > m->geom_condim[g2] ? m->geom_condim[g1] : m->geom_condim[g2];
            if (m->geom_priority[g1] != m->geom_priority[g2]) { snprintf(err, errlen, "geom priority unsupported"); goto fail; }
            if (cd != 1 && cd != 3) { snprintf(err, errlen, "condim %d unsupported", cd); goto fail; }
            if (m->npair == cap) { cap *= 2; m->pair = (int*)realloc(m->pair, sizeof(int) * 2 * cap); }
            m->pair[2 * m->npair] = g1; m->pair[2 * m->npair + 1] = g2; m->npair++;
          }
  }
  return m;
fail:
  sgo_model_free(m);
  return NULL;
}

void sgo_model_free(sgo_model* m) {
  if (!m) return;
  free(m->blob); free(m->geom_lmat); free(m->site_lmat); free(m->pair);
  free(m->jnt_qposadr); free(m->jnt_dofadr); free(m->dof_jntid); free(m->dof_rot); free(m->body_dofadr); free(m->body_dofnum);
  free(m);
}
int sgo_nv(const sgo_model* m) { return m->nv; }
int sgo_nq(const sgo_model* m) { return m->nq; }
int sgo_njnt(const sgo_model* m) { return m->njnt; }
int sgo_nu(const sgo_model* m) { return m->nu; }
int sgo_nsensordata(const sgo_model* m) { return 3 * m->nsensor; }
int sgo_ntendon(const sgo_model* m) { return m->ntendon; }

static double* dalloc(size_t n) { return (double*)calloc(n ? n : 1, sizeof(double)); }
static int* ialloc(size_t n) { return (int*)calloc(n ? n : 1, sizeof(int)); }

static void efc_reserve(sgo_data* d, int rows, int nnz) {
  if (rows > d->efc_cap) {
    int c = d->efc_cap ? d->efc_cap : 256;
    while (c < rows) c *= 2;
#define GROWD(p, k) d->p = (double*)realloc(d->p, sizeof(double) * (size_t)(k) * c)
#define GROWI(p, k) d->p = (int*)realloc(d->p, sizeof(int) * (size_t)(k) * c)
    GROWI(efc_type, 1); GROWI(efc_id, 1); d->J_rowadr = (int*)realloc(d->J_rowadr, sizeof(int) * (c + 1));
    GROWD(efc_pos, 1); GROWD(efc_margin, 1); GROWD(efc_diagApprox, 1); GROWD(efc_R, 1); GROWD(efc_D, 1); GROWD(efc_KBIP, 4);
    GROWD(efc_vel, 1); GROWD(efc_aref, 1); GROWD(efc_b, 1); GROWD(efc_force, 1); GROWD(efc_jar, 1);
    GROWI(AR_rownnz, 1); GROWI(AR_rowadr, 1); GROWI(stamp, 1);
    d->efc_cap = c;
  }
  if (nnz > d->jnnz_cap) {
    int c = d->jnnz_cap ? d->jnnz_cap : 4096;
    while (c < nnz) c *= 2;
    d->J_col = (int*)realloc(d->J_col, sizeof(int) * c);
    d->J_val = (double*)realloc(d->J_val, sizeof(double) * c);
    d->jnnz_cap = c;
  }
}

sgo_data* sgo_data_new(const sgo_model* m) {
  sgo_data* d = (sgo_data*)calloc(1, sizeof *d);
  int nv = m->nv, nb = m->nbody;
  d->m = m;
  d->qpos = dalloc(m->nq); d->qvel = dalloc(nv); d->act = dalloc(m->nu); d->ctrl = dalloc(m->nu); d->qacc_warmstart = dalloc(nv);
  d->jnt_stiffness = dalloc(m->njnt); d->tendon_stiffness = dalloc(m->ntendon);
  memcpy(d->jnt_stiffness, m->jnt_stiffness, sizeof(double) * m->njnt);
  memcpy(d->tendon_stiffness, m->tendon_stiffness, sizeof(double) * m->ntendon);
  d->xpos = dalloc(3 * nb); d->xquat = dalloc(4 * nb); d->xmat = dalloc(9 * nb); d->xipos = dalloc(3 * nb); d->ximat = dalloc(9 * nb);
  d->xanchor = dalloc(3 * nv); d->xaxis = dalloc(3 * nv);
  d->geom_xpos = dalloc(3 * m->ngeom); d->geom_xmat = dalloc(9 * m->ngeom);
  d->site_xpos = dalloc(3 * m->nsite); d->site_xmat = dalloc(9 * m->nsite);
  d->ten_length = dalloc(m->ntendon); d->ten_J = dalloc((size_t)m->ntendon * nv); d->ten_velocity = dalloc(m->ntendon);
  d->qM = dalloc((size_t)nv * nv); d->qLD = dalloc((size_t)nv * nv);
  d->bw = dalloc(3 * nb); d->bv = dalloc(3 * nb); d->bal = dalloc(3 * nb); d->ba = dalloc(3 * nb);
  d->qfrc_passive = dalloc(nv); d->qfrc_bias = dalloc(nv); d->qfrc_actuator = dalloc(nv); d->qfrc_smooth = dalloc(nv);
  d->qacc_smooth = dalloc(nv); d->qfrc_constraint = dalloc(nv); d->qacc = dalloc(nv); d->tmpv = dalloc(nv); d->tmpv2 = dalloc(nv); d->tmpv3 = dalloc(nv);
  d->act_dot = dalloc(m->nu); d->actuator_force = dalloc(m->nu); d->sensordata = dalloc(3 * m->nsensor);
  d->contact = (contact_t*)calloc(MAXCON, sizeof(contact_t));
  d->dofrow_adr = ialloc(nv + 1);
  d->scr_cols = ialloc(nv + 2 * MAXCHAIN); d->scr_vals = dalloc(nv + 2 * MAXCHAIN); d->scr_fill = ialloc(nv);
  sgo_reset(m, d);
  return d;
}

void sgo_data_free(sgo_data* d) {
  if (!d) return;
  double* ds[] = {d->qpos, d->qvel, d->act, d->ctrl, d->qacc_warmstart, d->jnt_stiffness, d->tendon_stiffness, d->xpos, d->xquat,
                  d->xmat, d->xipos, d->ximat, d->xanchor, d->xaxis, d->geom_xpos, d->geom_xmat, d->site_xpos, d->site_xmat,
                  d->ten_length, d->ten_J, d->ten_velocity, d->qM, d->qLD, d->bw, d->bv, d->bal, d->ba, d->qfrc_passive,
                  d->qfrc_bias, d->qfrc_actuator, d->qfrc_smooth, d->qacc_smooth, d->qfrc_constraint, d->qacc, d->tmpv, d->tmpv2, d->tmpv3,
                  d->act_dot, d->actuator_force, d->sensordata, d->J_val, d->efc_pos, d->efc_margin, d->efc_diagApprox, d->efc_R,
                  d->efc_D, d->efc_KBIP, d->efc_vel, d->efc_aref, d->efc_b, d->efc_force, d->efc_jar, d->AR};
  for (size_t i = 0; i < sizeof ds / sizeof ds[0]; i++) free(ds[i]);
  int* is[] = {d->efc_type, d->efc_id, d->J_rowadr, d->J_col, d->AR_rownnz, d->AR_rowadr, d->AR_col, d->dofrow_adr, d->dofrow, d->stamp,
               d->scr_cols, d->scr_fill, d->scr_rowJk};
  free(d->scr_vals);
  for (size_t i = 0; i < sizeof is / sizeof is[0]; i++) free(is[i]);
  free(d->contact);
  free(d);
}

double* sgo_qpos(sgo_data* d) { return d->qpos; }
double* sgo_qvel(sgo_data* d) { return d->qvel; }
double* sgo_act(sgo_data* d) { return d->act; }
double* sgo_ctrl(sgo_data* d) { return d->ctrl; }
double* sgo_qacc(sgo_data* d) { return d->qacc; }
double* sgo_qacc_warmstart(sgo_data* d) { return d->qacc_warmstart; }
double* sgo_sensordata(sgo_data* d) { return d->sensordata; }
double* sgo_jnt_stiffness(sgo_data* d) { return d->jnt_stiffness; }
double* sgo_tendon_stiffness(sgo_data* d) { return d->tendon_stiffness; }
double* sgo_ten_length(sgo_data* d) { return d->ten_length; }
double* sgo_qfrc_bias(sgo_data* d) { return d->qfrc_bias; }
double* sgo_qM(sgo_data* d) { return d->qM; }
double* sgo_site_xpos(sgo_data* d) { return d->site_xpos; }
double* sgo_efc_force(sgo_data* d) { return d->efc_force; }
/* the constraint problem of the last forward pass (tests/test_oracle_kat.py checks the PGS fixed point against its KKT
 * conditions): dense A + R (nefc x nefc, row-major), b with res = (A + R) f + b, row types (0 equality, 3 limit, 7 elliptic
 * contact: three consecutive rows), contact id of a row, and a contact's friction coefficients */
double* sgo_efc_AR(sgo_data* d) { return d->AR; }
double* sgo_efc_b(sgo_data* d) { return d->efc_b; }
int* sgo_efc_type(sgo_data* d) { return d->efc_type; }
int* sgo_efc_id(sgo_data* d) { return d->efc_id; }
void sgo_contact_friction(const sgo_data* d, int i, double* mu5) { for (int k = 0; k < 5; k++) mu5[k] = d->contact[i].friction[k]; }
int sgo_ncon(const sgo_data* d) { return d->ncon; }
int sgo_nefc(const sgo_data* d) { return d->nefc; }
int sgo_solver_iter(const sgo_data* d) { return d->solver_iter; }
void sgo_contact(const sgo_data* d, int i, int* g1, int* g2, double* dist, double* pos3, double* frame9) {
  const contact_t* c = d->contact + i;
  if (g1) *g1 = c->geom1;
  if (g2) *g2 = c->geom2;
  if (dist) *dist = c->dist;
  if (pos3) memcpy(pos3, c->pos, 24);
  if (frame9) memcpy(frame9, c->frame, 72);
}

/* mj_resetData (manenv.py:57): qpos=qpos0, everything else zero; model parameters untouched */
void sgo_reset(const sgo_model* m, sgo_data* d) {
  memcpy(d->qpos, m->qpos0, sizeof(double) * m->nq);
  memset(d->qvel, 0, sizeof(double) * m->nv);
  memset(d->qacc_warmstart, 0, sizeof(double) * m->nv);
  memset(d->qacc, 0, sizeof(double) * m->nv);
  memset(d->act, 0, sizeof(double) * m->nu);
  memset(d->ctrl, 0, sizeof(double) * m->nu);
  memset(d->sensordata, 0, sizeof(double) * 3 * m->nsensor);
  d->time = 0; d->ncon = 0; d->nefc = 0; d->warnings = 0; d->solver_iter = 0;
}

/* ------------------------------------------------------------------ stage 1: kinematics */
static void kinematics(const sgo_model* m, sgo_data* d) {
  static const double I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  d->xpos[0] = d->xpos[1] = d->xpos[2] = 0;
  d->xquat[0] = 1; d->xquat[1] = d->xquat[2] = d->xquat[3] = 0;
  memcpy(d->xmat, I9, sizeof I9);
  for (int i = 1; i < m->nbody; i++) {
    int p = m->body_parentid[i];
    double pos[3], quat[4], mat[9], t[3];
    mulmat3(t, d->xmat + 9 * p, m->body_pos + 3 * i);
    for (int k = 0; k < 3; k++) pos[k] = d->xpos[3 * p + k] + t[k];
    quatmul(quat, d->xquat + 4 * p, m->body_quat + 4 * i);
    for (int k = 0; k < m->body_jntnum[i]; k++) { /* xanchor / xaxis are indexed by DOF */
      int j = m->body_jntadr[i] + k, qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
      if (m->jnt_type[j] == SG_JNT_FREE) { /* the 7 positions ARE the body's pose (a child of the world) */
        memcpy(pos, d->qpos + qa, 24); memcpy(quat, d->qpos + qa + 3, 32);
        double nq = sqrt(quat[0] * quat[0] + quat[1] * quat[1] + quat[2] * quat[2] + quat[3] * quat[3]);
        for (int c = 0; c < 4; c++) quat[c] /= nq;
        quat2mat(mat, quat);
        for (int c = 0; c < 3; c++) /* dofs 0-2: translation along the world axes; 3-5: rotation about the body's own axes */
          for (int r = 0; r < 3; r++) {
            d->xanchor[3 * (da + c) + r] = d->xanchor[3 * (da + 3 + c) + r] = pos[r];
            d->xaxis[3 * (da + c) + r] = r == c ? 1.0 : 0.0;
            d->xaxis[3 * (da + 3 + c) + r] = mat[3 * r + c];
          }
        continue;
      }
      quat2mat(mat, quat);
      mulmat3(t, mat, m->jnt_pos + 3 * j);
      for (int c = 0; c < 3; c++) d->xanchor[3 * da + c] = pos[c] + t[c];
      mulmat3(d->xaxis + 3 * da, mat, m->jnt_axis + 3 * j);
      double dq = d->qpos[qa] - m->qpos0[qa];
      if (m->jnt_type[j] == SG_JNT_SLIDE) {
        addscl3(pos, d->xaxis + 3 * da, dq);
      } else {
        double s = sin(0.5 * dq), ql[4] = {cos(0.5 * dq), m->jnt_axis[3 * j] * s, m->jnt_axis[3 * j + 1] * s, m->jnt_axis[3 * j + 2] * s};
        quatmul(quat, quat, ql);
        quat2mat(mat, quat);
        mulmat3(t, mat, m->jnt_pos + 3 * j);
        for (int c = 0; c < 3; c++) pos[c] = d->xanchor[3 * da + c] - t[c];
      }
    }
    double n = sqrt(quat[0] * quat[0] + quat[1] * quat[1] + quat[2] * quat[2] + quat[3] * quat[3]);
    for (int c = 0; c < 4; c++) d->xquat[4 * i + c] = quat[c] / n;
    memcpy(d->xpos + 3 * i, pos, 24);
    quat2mat(d->xmat + 9 * i, d->xquat + 4 * i);
    mulmat3(t, d->xmat + 9 * i, m->body_ipos + 3 * i);
    for (int c = 0; c < 3; c++) d->xipos[3 * i + c] = pos[c] + t[c];
    /* world-frame inertia about the COM: R I R' */
    double RI[9], Rt[9];
    mulmat33(RI, d->xmat + 9 * i, m->body_imat + 9 * i);
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) Rt[3 * a + b] = d->xmat[9 * i + 3 * b + a];
    mulmat33(d->ximat + 9 * i, RI, Rt);
  }
  for (int g = 0; g < m->ngeom; g++) {
    int b = m->geom_bodyid[g];
    double t[3];
    mulmat3(t, d->xmat + 9 * b, m->geom_pos + 3 * g);
    for (int c = 0; c < 3; c++) d->geom_xpos[3 * g + c] = d->xpos[3 * b + c] + t[c];
    mulmat33(d->geom_xmat + 9 * g, d->xmat + 9 * b, m->geom_lmat + 9 * g);
  }
  for (int s = 0; s < m->nsite; s++) {
    int b = m->site_bodyid[s];
    double t[3];
    mulmat3(t, d->xmat + 9 * b, m->site_pos + 3 * s);
    for (int c = 0; c < 3; c++) d->site_xpos[3 * s + c] = d->xpos[3 * b + c] + t[c];
    mulmat33(d->site_xmat + 9 * s, d->xmat + 9 * b, m->site_lmat + 9 * s);
  }
}

/* Jacobian of a world point rigidly attached to `body`, restricted to the dofs of its chain */
typedef struct { int n; int dof[MAXCHAIN]; double jp[MAXCHAIN][3], jr[MAXCHAIN][3]; } chainjac;
static void jac_chain(const sgo_model* m, const sgo_data* d, int body, const double* point, chainjac* cj) {
  cj->n = 0;
  for (int b = body; b > 0; b = m->body_parentid[b])
    for (int k = m->body_dofnum[b] - 1; k >= 0; k--) {
      int j = m->body_dofadr[b] + k, n = cj->n++;   /* j: a dof */
      const double* ax = d->xaxis + 3 * j;
      cj->dof[n] = j;
      if (!m->dof_rot[j]) {
        memcpy(cj->jp[n], ax, 24);
        cj->jr[n][0] = cj->jr[n][1] = cj->jr[n][2] = 0;
      } else {
        double r[3] = {point[0] - d->xanchor[3 * j], point[1] - d->xanchor[3 * j + 1], point[2] - d->xanchor[3 * j + 2]};
        cross3(cj->jp[n], ax, r);
        memcpy(cj->jr[n], ax, 24);
      }
    }
}

/* ------------------------------------------------------------------ stage 3: tendons (+transmission) */
static void tendons(const sgo_model* m, sgo_data* d) {
  int nv = m->nv;
  memset(d->ten_J, 0, sizeof(double) * (size_t)m->ntendon * nv);
  for (int t = 0; t < m->ntendon; t++) {
    int a = m->tendon_adr[t], n = m->tendon_num[t];
    double L = 0, *J = d->ten_J + (size_t)t * nv;
    if (m->wrap_type[a] == SG_WRAP_JOINT) {
      for (int w = a; w < a + n; w++) { /* wrap_objid: a joint id */
        L += m->wrap_prm[w] * d->qpos[m->jnt_qposadr[m->wrap_objid[w]]];
        J[m->jnt_dofadr[m->wrap_objid[w]]] = m->wrap_prm[w];
      }
    } else {
      for (int w = a; w < a + n - 1; w++) {
        int s0 = m->wrap_objid[w], s1 = m->wrap_objid[w + 1];
        const double *p0 = d->site_xpos + 3 * s0, *p1 = d->site_xpos + 3 * s1;
        double dif[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]}, len = sqrt(dot3(dif, dif));
        L += len;
        if (len < MINVAL) continue;
        for (int c = 0; c < 3; c++) dif[c] /= len;
        chainjac c1, c0;
        jac_chain(m, d, m->site_bodyid[s1], p1, &c1);
        jac_chain(m, d, m->site_bodyid[s0], p0, &c0);
        for (int k = 0; k < c1.n; k++) J[c1.dof[k]] += dot3(dif, c1.jp[k]);
        for (int k = 0; k < c0.n; k++) J[c0.dof[k]] -= dot3(dif, c0.jp[k]);
      }
    }
    d->ten_length[t] = L;
  }
}

/* ------------------------------------------------------------------ stage 4: mass matrix + L'DL */
static void mass_matrix(const sgo_model* m, sgo_data* d) {
  int nv = m->nv;
  memset(d->qM, 0, sizeof(double) * (size_t)nv * nv);
  for (int i = 0; i < nv; i++) d->qM[(size_t)i * nv + i] = m->dof_armature[i];
  for (int b = 1; b < m->nbody; b++) {
    if (m->body_weldid[b] == 0 || m->body_mass[b] <= 0) continue;
    chainjac cj;
    jac_chain(m, d, b, d->xipos + 3 * b, &cj);
    double mass = m->body_mass[b];
    for (int a = 0; a < cj.n; a++) {
      double Ir[3];
      mulmat3(Ir, d->ximat + 9 * b, cj.jr[a]);
      for (int c = 0; c < cj.n; c++)
        d->qM[(size_t)cj.dof[a] * nv + cj.dof[c]] += mass * dot3(cj.jp[a], cj.jp[c]) + dot3(Ir, cj.jr[c]);
    }
  }
}
/* Featherstone sparse L'DL using dof_parentid; L stored in place of the lower triangle (unit diag implied), D on the diag */
static void factor(const sgo_model* m, const double* M, double* L) {
  int nv = m->nv;
  if (L != M) memcpy(L, M, sizeof(double) * (size_t)nv * nv);
  for (int k = nv - 1; k >= 0; k--) {
    int i = m->dof_parentid[k];
    while (i >= 0) {
      double a = L[(size_t)k * nv + i] / L[(size_t)k * nv + k];
      for (int j = i; j >= 0; j = m->dof_parentid[j]) L[(size_t)i * nv + j] -= a * L[(size_t)k * nv + j];
      L[(size_t)k * nv + i] = a;
      i = m->dof_parentid[i];
    }
  }
}
static void solve_ld(const sgo_model* m, const double* L, double* x) {
  int nv = m->nv;
  for (int k = nv - 1; k >= 0; k--)
    for (int i = m->dof_parentid[k]; i >= 0; i = m->dof_parentid[i]) x[i] -= L[(size_t)k * nv + i] * x[k];
  for (int k = 0; k < nv; k++) x[k] /= L[(size_t)k * nv + k];
  for (int k = 0; k < nv; k++)
    for (int i = m->dof_parentid[k]; i >= 0; i = m->dof_parentid[i]) x[k] -= L[(size_t)k * nv + i] * x[i];
}

/* ------------------------------------------------------------------ stage 5: collision */
static void make_frame(double* f) { /* mju_makeFrame: f[0..2] normal given, f[3..5] zero -> complete right-handed frame */
  double n = sqrt(dot3(f, f));
  f[0] /= n; f[1] /= n; f[2] /= n;
  if (sqrt(dot3(f + 3, f + 3)) < 0.5) {
    f[3] = f[4] = f[5] = 0;
    if (f[1] < 0.5 && f[1] > -0.5) f[4] = 1; else f[5] = 1;
  }
  double t = dot3(f, f + 3);
  addscl3(f + 3, f, -t);
  n = sqrt(dot3(f + 3, f + 3));
  f[3] /= n; f[4] /= n; f[5] /= n;
  cross3(f + 6, f, f + 3);
}

static contact_t* add_contact(const sgo_model* m, sgo_data* d, int g1, int g2, double dist, const double* pos, const double* normal,
                              const double* tangent_hint) {
  if (d->ncon >= MAXCON || (m->nconmax > 0 && d->ncon >= m->nconmax)) { d->warnings |= SGO_WARN_CONTACTFULL; return NULL; }
  contact_t* c = d->contact + d->ncon++;
  c->geom1 = g1; c->geom2 = g2; c->dist = dist;
  memcpy(c->pos, pos, 24);
  memcpy(c->frame, normal, 24);
  if (tangent_hint) memcpy(c->frame + 3, tangent_hint, 24); else c->frame[3] = c->frame[4] = c->frame[5] = 0;
  make_frame(c->frame);
  /* contact parameter mixing (equal priority): condim max, friction max, solref/solimp weighted by solmix, margin/gap max */
  c->dim = m->geom_condim[g1] > m->geom_condim[g2] ? m->geom_condim[g1] : m->geom_condim[g2];
  const double *f1 = m->geom_friction + 3 * g1, *f2 = m->geom_friction + 3 * g2;
  double fr[3];
  for (int k = 0; k < 3; k++) fr[k] = f1[k] > f2[k] ? f1[k] : f2[k];
  c->friction[0] = c->friction[1] = fr[0]; c->friction[2] = fr[1]; c->friction[3] = c->friction[4] = fr[2];
  double mix, s1 = m->geom_solmix[g1], s2 = m->geom_solmix[g2];
  if (s1 >= MINVAL && s2 >= MINVAL) mix = s1 / (s1 + s2);
  else if (s1 < MINVAL && s2 < MINVAL) mix = 0.5;
  else mix = s1 < MINVAL ? 0.0 : 1.0;
  const double *r1 = m->geom_solref + 2 * g1, *r2 = m->geom_solref + 2 * g2;
  if (r1[0] > 0 && r2[0] > 0) for (int k = 0; k < 2; k++) c->solref[k] = mix * r1[k] + (1 - mix) * r2[k];
  else for (int k = 0; k < 2; k++) c->solref[k] = r1[k] < r2[k] ? r1[k] : r2[k];
  for (int k = 0; k < 5; k++) c->solimp[k] = mix * m->geom_solimp[5 * g1 + k] + (1 - mix) * m->geom_solimp[5 * g2 + k];
  double mg = fmax(m->geom_margin[g1], m->geom_margin[g2]), gp = fmax(m->geom_gap[g1], m->geom_gap[g2]);
  c->includemargin = mg - gp;
  c->mu = c->friction[0];
  c->efc_address = -1;
  return c;
}

/* sphere (radius r at world centre c) against box g2; geom1 = g1.  Normal points from geom1 into the box. */
static int sphere_box(const sgo_model* m, sgo_data* d, int g1, int g2, const double* c, double r, double margin) {
  const double *bp = d->geom_xpos + 3 * g2, *bm = d->geom_xmat + 9 * g2, *sz = m->geom_size + 3 * g2;
  double t[3] = {c[0] - bp[0], c[1] - bp[1], c[2] - bp[2]}, cen[3], cl[3], dif[3];
  mulmatT3(cen, bm, t);
  for (int k = 0; k < 3; k++) { cl[k] = cen[k] < -sz[k] ? -sz[k] : cen[k] > sz[k] ? sz[k] : cen[k]; dif[k] = cl[k] - cen[k]; }
  double dist = sqrt(dot3(dif, dif));
  if (dist - r > margin) return 0;
  double nl[3], pl[3], cd;
  if (dist <= MINVAL) { /* centre inside the box: leave through the nearest face */
    double closest = 1e300; int ka = 0; double sg = 1;
    for (int k = 0; k < 3; k++) {
      if (sz[k] - cen[k] < closest - 1e-12) { closest = sz[k] - cen[k]; ka = k; sg = 1; }
      if (sz[k] + cen[k] < closest - 1e-12) { closest = sz[k] + cen[k]; ka = k; sg = -1; }
    }
    nl[0] = nl[1] = nl[2] = 0; nl[ka] = -sg;
    for (int k = 0; k < 3; k++) pl[k] = cen[k] + nl[k] * (r - closest) * 0.5;
    cd = -closest - r;
  } else {
    for (int k = 0; k < 3; k++) { nl[k] = dif[k] / dist; pl[k] = cen[k] + nl[k] * (r + dist) * 0.5; }
    cd = dist - r;
  }
  double nw[3], pw[3];
  mulmat3(nw, bm, nl);
  mulmat3(pw, bm, pl);
  for (int k = 0; k < 3; k++) pw[k] += bp[k];
  return add_contact(m, d, g1, g2, cd, pw, nw, NULL) != NULL;
}

/* Signed distance from point q (box frame) to the solid box: >0 outside (Euclidean), <=0 inside (-depth to nearest face) */
static double box_sdist(const double* q, const double* sz) {
  double o2 = 0, in = -1e300;
  for (int k = 0; k < 3; k++) {
    double e = fabs(q[k]) - sz[k];
    if (e > 0) o2 += e * e;
    if (e > in) in = e;
  }
  return o2 > 0 ? sqrt(o2) : in;
}

/* Deviation D1: closest point of the capsule segment p + t*h (t in [-1,1], box frame) to the solid box.
 * Outside: the squared distance is convex piecewise quadratic in t; its derivative g is piecewise linear and
 * monotone, so the minimiser is bracketed by the breakpoints (where a coordinate crosses a face) and found by
 * one linear interpolation.  If the segment touches the box (distance 0) the parameter of deepest penetration
 * minimises max_k(|q_k| - s_k), a max of 6 lines: examine the endpoints and pairwise intersections. */
static double seg_box_param(const double* p, const double* h, const double* sz) {
  double tk[8], gk[8];
  int n = 0;
  tk[n++] = -1; tk[n++] = 1;
  for (int k = 0; k < 3; k++)
    if (fabs(h[k]) > MINVAL)
      for (int s = -1; s <= 1; s += 2) {
        double t = (s * sz[k] - p[k]) / h[k];
        if (t > -1 && t < 1) tk[n++] = t;
      }
  for (int i = 0; i < n; i++) {
    double g = 0;
    for (int k = 0; k < 3; k++) {
      double q = p[k] + tk[i] * h[k];
      if (q > sz[k]) g += (q - sz[k]) * h[k]; else if (q < -sz[k]) g += (q + sz[k]) * h[k];
    }
    gk[i] = g;
  }
  double tbest;
  if (gk[0] >= 0) tbest = -1;
  else if (gk[1] <= 0) tbest = 1;
  else {
    double tlo = -1, glo = gk[0], thi = 1, ghi = gk[1];
    for (int i = 2; i < n; i++) {
      if (gk[i] <= 0 && tk[i] > tlo) { tlo = tk[i]; glo = gk[i]; }
      if (gk[i] >= 0 && tk[i] < thi) { thi = tk[i]; ghi = gk[i]; }
    }
    tbest = (ghi - glo > MINVAL && thi > tlo) ? tlo + (thi - tlo) * (-glo) / (ghi - glo) : tlo;
  }
  double q[3] = {p[0] + tbest * h[0], p[1] + tbest * h[1], p[2] + tbest * h[2]};
  if (box_sdist(q, sz) > 0) return tbest;
  /* segment reaches into the box: minimise the max of the 6 lines l(t) = +-(p_k + t h_k) - s_k over [-1,1] */
  double a[6], b[6];
  for (int k = 0; k < 3; k++) { a[2 * k] = h[k]; b[2 * k] = p[k] - sz[k]; a[2 * k + 1] = -h[k]; b[2 * k + 1] = -p[k] - sz[k]; }
  double cand[17];
  int nc = 0;
  cand[nc++] = -1; cand[nc++] = 1;
  for (int i = 0; i < 6; i++)
    for (int j = i + 1; j < 6; j++)
      if (fabs(a[i] - a[j]) > MINVAL) {
        double t = (b[j] - b[i]) / (a[i] - a[j]);
        if (t > -1 && t < 1) cand[nc++] = t;
      }
  double best = 1e300, tb = -1;
  for (int i = 0; i < nc; i++) {
    double f = -1e300;
    for (int k = 0; k < 6; k++) { double v = a[k] * cand[i] + b[k]; if (v > f) f = v; }
    if (f < best - 1e-12) { best = f; tb = cand[i]; } /* ties: first candidate wins in every implementation */
  }
  return tb;
}

static int capsule_box(const sgo_model* m, sgo_data* d, int g1, int g2, double margin) {
  const double *cp = d->geom_xpos + 3 * g1, *cm = d->geom_xmat + 9 * g1;
  const double *bp = d->geom_xpos + 3 * g2, *bm = d->geom_xmat + 9 * g2, *sz = m->geom_size + 3 * g2;
  double r = m->geom_size[3 * g1], hl = m->geom_size[3 * g1 + 1];
  double axw[3] = {cm[2], cm[5], cm[8]}, t[3] = {cp[0] - bp[0], cp[1] - bp[1], cp[2] - bp[2]}, p[3], h[3];
  mulmatT3(p, bm, t);
  mulmatT3(h, bm, axw);
  for (int k = 0; k < 3; k++) h[k] *= hl;
  double t1 = seg_box_param(p, h, sz);
  double c1[3] = {cp[0] + axw[0] * hl * t1, cp[1] + axw[1] * hl * t1, cp[2] + axw[2] * hl * t1};
  int n = sphere_box(m, d, g1, g2, c1, r, margin);
  /* second contact at the far end cap when it is also within the margin */
  double t2 = t1 >= 0 ? -1.0 : 1.0;
  if (fabs(t2 - t1) * hl > 1e-6) {
    double c2[3] = {cp[0] + axw[0] * hl * t2, cp[1] + axw[1] * hl * t2, cp[2] + axw[2] * hl * t2};
    n += sphere_box(m, d, g1, g2, c2, r, margin);
  }
  return n;
}

/* Box-box narrowphase (deviation D2': NOT a transcription of MuJoCo's mjc_BoxBox, which is not available here; the classic
 * separating-axis + face-clipping construction that box-box routines of this family share):
 *  1. 15 candidate axes (3 face normals of each box, 9 edge x edge cross products); separation s = |T.L| - (rA + rB); any
 *     s > margin -> no contact.  The axis of least penetration wins; an edge axis must beat the best face axis by more than
 *     BB_FUDGE (relative) to be chosen, so near-ties go to faces (stable manifolds).
 *  2. face axis: the reference box owns it; of the other ("incident") box take the face most anti-parallel to the normal, clip
 *     its 4 corners against the 4 side planes of the reference face (Sutherland-Hodgman), keep the points whose distance to
 *     the reference face is <= margin: up to 8 contacts, in clip order; position = midway between point and reference face.
 *  3. edge axis: one contact midway between the closest points of the two edges.
 * Normal: from geom1 towards geom2; dist < 0 = penetration. */
#define BB_FUDGE 1.05
static int box_box(const sgo_model* m, sgo_data* d, int g1, int g2, double margin) {
  const double *p1 = d->geom_xpos + 3 * g1, *R1 = d->geom_xmat + 9 * g1, *s1 = m->geom_size + 3 * g1;
  const double *p2 = d->geom_xpos + 3 * g2, *R2 = d->geom_xmat + 9 * g2, *s2 = m->geom_size + 3 * g2;
  double T[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]}, A[3][3], B[3][3];
  for (int k = 0; k < 3; k++) { A[k][0] = R1[k]; A[k][1] = R1[3 + k]; A[k][2] = R1[6 + k]; B[k][0] = R2[k]; B[k][1] = R2[3 + k]; B[k][2] = R2[6 + k]; }
  double C[3][3], Q[3][3]; /* C = A' B, Q = |C| */
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { C[i][j] = dot3(A[i], B[j]); Q[i][j] = fabs(C[i][j]); }
  double best = -1e300, bn[3] = {0, 0, 0};
  int code = -1; /* 0..2 face of box 1, 3..5 face of box 2, 6 + 3 i + j edge i x edge j */
  for (int k = 0; k < 3; k++) { /* faces of box 1 */
    double t = dot3(T, A[k]), sep = fabs(t) - (s1[k] + s2[0] * Q[k][0] + s2[1] * Q[k][1] + s2[2] * Q[k][2]);
    if (sep > margin) return 0;
    if (sep > best) { best = sep; code = k; double sg = t < 0 ? -1 : 1; for (int c = 0; c < 3; c++) bn[c] = sg * A[k][c]; }
  }
  for (int k = 0; k < 3; k++) { /* faces of box 2 */
    double t = dot3(T, B[k]), sep = fabs(t) - (s2[k] + s1[0] * Q[0][k] + s1[1] * Q[1][k] + s1[2] * Q[2][k]);
    if (sep > margin) return 0;
    if (sep > best) { best = sep; code = 3 + k; double sg = t < 0 ? -1 : 1; for (int c = 0; c < 3; c++) bn[c] = sg * B[k][c]; }
  }
  double ebest = -1e300, en[3] = {0, 0, 0};
  int ecode = -1;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double L[3];
      cross3(L, A[i], B[j]);
      double n = sqrt(dot3(L, L));
      if (n < 1e-6) continue; /* parallel edges: covered by the face axes */
      for (int c = 0; c < 3; c++) L[c] /= n;
      double ra = 0, rb = 0;
      for (int k = 0; k < 3; k++) { ra += s1[k] * fabs(dot3(L, A[k])); rb += s2[k] * fabs(dot3(L, B[k])); }
      double t = dot3(T, L), sep = fabs(t) - (ra + rb);
      if (sep > margin) return 0;
      if (sep > ebest) { ebest = sep; ecode = 6 + 3 * i + j; double sg = t < 0 ? -1 : 1; for (int c = 0; c < 3; c++) en[c] = sg * L[c]; }
    }
  if (ecode >= 0 && ebest > best + (BB_FUDGE - 1.0) * fabs(best) + 1e-9) { best = ebest; code = ecode; memcpy(bn, en, 24); }
  if (code >= 6) { /* edge-edge: closest points of the two supporting edges */
    int i = (code - 6) / 3, j = (code - 6) % 3;
    double pa[3], pb[3];
    for (int c = 0; c < 3; c++) { pa[c] = p1[c]; pb[c] = p2[c]; }
    for (int k = 0; k < 3; k++) {
      if (k != i) { double sg = dot3(bn, A[k]) > 0 ? 1 : -1; addscl3(pa, A[k], sg * s1[k]); }
      if (k != j) { double sg = dot3(bn, B[k]) > 0 ? -1 : 1; addscl3(pb, B[k], sg * s2[k]); }
    }
    /* lines pa + a A[i], pb + b B[j] */
    double r[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]}, uv = C[i][j], den = 1 - uv * uv;
    double ta = 0, tb = 0;
    if (den > 1e-12) { double q1 = dot3(A[i], r), q2 = dot3(B[j], r); ta = (q1 - uv * q2) / den; tb = (uv * q1 - q2) / den; }
    ta = ta > s1[i] ? s1[i] : ta < -s1[i] ? -s1[i] : ta;
    tb = tb > s2[j] ? s2[j] : tb < -s2[j] ? -s2[j] : tb;
    double pos[3];
    for (int c = 0; c < 3; c++) pos[c] = 0.5 * ((pa[c] + ta * A[i][c]) + (pb[c] + tb * B[j][c]));
    return add_contact(m, d, g1, g2, best, pos, bn, NULL) != NULL;
  }
  /* face contact: reference box owns the axis */
  const int ref1 = code < 3, ka = ref1 ? code : code - 3;
  const double (*Ra)[3] = ref1 ? A : B, (*Rb)[3] = ref1 ? B : A;
  const double *pa = ref1 ? p1 : p2, *pb = ref1 ? p2 : p1, *sa = ref1 ? s1 : s2, *sb = ref1 ? s2 : s1;
  double nrm[3]; /* outward normal of the reference face (towards the incident box) */
  for (int c = 0; c < 3; c++) nrm[c] = ref1 ? bn[c] : -bn[c];
  /* incident face: axis of the incident box most aligned with nrm; its face on the side facing the reference box */
  int kb = 0; double mx = -1;
  for (int k = 0; k < 3; k++) { double v = fabs(dot3(nrm, Rb[k])); if (v > mx + 1e-12) { mx = v; kb = k; } }
  double sgb = dot3(nrm, Rb[kb]) > 0 ? -1 : 1, cen[3];
  for (int c = 0; c < 3; c++) cen[c] = pb[c] + sgb * sb[kb] * Rb[kb][c];
  int u = (kb + 1) % 3, v = (kb + 2) % 3, ua = (ka + 1) % 3, va = (ka + 2) % 3;
  double poly[16][3], tmp[16][3];
  int np = 4;
  const double cs[4][2] = {{1, 1}, {-1, 1}, {-1, -1}, {1, -1}};
  for (int q = 0; q < 4; q++)
    for (int c = 0; c < 3; c++) poly[q][c] = cen[c] + cs[q][0] * sb[u] * Rb[u][c] + cs[q][1] * sb[v] * Rb[v][c];
  /* clip against the four side planes of the reference face: |(x - pa).Ra[ua]| <= sa[ua], same for va */
  for (int pl = 0; pl < 4 && np > 0; pl++) {
    const int ax = pl < 2 ? ua : va; const double sg = (pl & 1) ? -1 : 1, lim = sa[ax];
    int nq = 0;
    for (int q = 0; q < np; q++) {
      const double *x0 = poly[q], *x1 = poly[(q + 1) % np];
      double e0[3] = {x0[0] - pa[0], x0[1] - pa[1], x0[2] - pa[2]}, e1[3] = {x1[0] - pa[0], x1[1] - pa[1], x1[2] - pa[2]};
      double d0 = sg * dot3(e0, Ra[ax]) - lim, d1 = sg * dot3(e1, Ra[ax]) - lim;
      if (d0 <= 0) { memcpy(tmp[nq++], x0, 24); }
      if ((d0 <= 0) != (d1 <= 0)) { double w = d0 / (d0 - d1); for (int c = 0; c < 3; c++) tmp[nq][c] = x0[c] + w * (x1[c] - x0[c]); nq++; }
    }
    np = nq;
    memcpy(poly, tmp, sizeof(double) * 3 * np);
  }
  int n = 0;
  for (int q = 0; q < np && n < 8; q++) {
    double e[3] = {poly[q][0] - pa[0], poly[q][1] - pa[1], poly[q][2] - pa[2]}, dist = dot3(e, nrm) - sa[ka], pos[3];
    if (dist > margin) continue;
    for (int c = 0; c < 3; c++) pos[c] = poly[q][c] - 0.5 * dist * nrm[c];
    if (add_contact(m, d, g1, g2, dist, pos, bn, NULL)) n++;
  }
  return n;
}

/* plane (geom g1) against box g2: the box corners within the margin of the plane, in corner order (x fastest), at most 4 */
static int plane_box(const sgo_model* m, sgo_data* d, int g1, int g2, double margin) {
  const double *p1 = d->geom_xpos + 3 * g1, *M1 = d->geom_xmat + 9 * g1;
  const double *p2 = d->geom_xpos + 3 * g2, *M2 = d->geom_xmat + 9 * g2, *sz = m->geom_size + 3 * g2;
  double nrm[3] = {M1[2], M1[5], M1[8]};
  int n = 0;
  for (int q = 0; q < 8 && n < 4; q++) {
    double loc[3] = {(q & 1 ? 1 : -1) * sz[0], (q & 2 ? 1 : -1) * sz[1], (q & 4 ? 1 : -1) * sz[2]}, w[3], pos[3];
    mulmat3(w, M2, loc);
    for (int c = 0; c < 3; c++) w[c] += p2[c] - p1[c];
    double dist = dot3(w, nrm);
    if (dist > margin) continue;
    for (int c = 0; c < 3; c++) pos[c] = w[c] + p1[c] - 0.5 * dist * nrm[c];
    if (add_contact(m, d, g1, g2, dist, pos, nrm, NULL)) n++;
  }
  return n;
}

static void collision(const sgo_model* m, sgo_data* d) {
  d->ncon = 0;
  for (int k = 0; k < m->npair; k++) {
    int g1 = m->pair[2 * k], g2 = m->pair[2 * k + 1], t1 = m->geom_type[g1], t2 = m->geom_type[g2];
    double margin = fmax(m->geom_margin[g1], m->geom_margin[g2]);
    const double *p1 = d->geom_xpos + 3 * g1, *p2 = d->geom_xpos + 3 * g2;
    if (t1 == SG_GEOM_PLANE) {
      const double* M1 = d->geom_xmat + 9 * g1;
      double nrm[3] = {M1[2], M1[5], M1[8]}, dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
      if (dot3(dif, nrm) > margin + m->geom_rbound[g2]) continue; /* bounding-sphere vs plane */
      if (t2 == SG_GEOM_SPHERE) {
        double r = m->geom_size[3 * g2], dist = dot3(dif, nrm) - r, pos[3];
        if (dist > margin) continue;
        for (int c = 0; c < 3; c++) pos[c] = p2[c] - nrm[c] * (r + 0.5 * dist);
        add_contact(m, d, g1, g2, dist, pos, nrm, NULL);
      } else if (t2 == SG_GEOM_CAPSULE) {
        const double* M2 = d->geom_xmat + 9 * g2;
        double ax[3] = {M2[2], M2[5], M2[8]}, r = m->geom_size[3 * g2], hl = m->geom_size[3 * g2 + 1];
        for (int s = -1; s <= 1; s += 2) { /* one sphere test per end cap; tangent aligned with the capsule axis */
          double c[3] = {p2[0] + s * hl * ax[0], p2[1] + s * hl * ax[1], p2[2] + s * hl * ax[2]};
          double e[3] = {c[0] - p1[0], c[1] - p1[1], c[2] - p1[2]}, dist = dot3(e, nrm) - r, pos[3];
          if (dist > margin) continue;
          for (int q = 0; q < 3; q++) pos[q] = c[q] - nrm[q] * (r + 0.5 * dist);
          add_contact(m, d, g1, g2, dist, pos, nrm, ax);
        }
      } else {
        plane_box(m, d, g1, g2, margin);
      }
      continue;
    }
    double dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]}, bound = m->geom_rbound[g1] + m->geom_rbound[g2] + margin;
    if (dot3(dif, dif) > bound * bound) continue; /* bounding spheres */
    if (t1 == SG_GEOM_SPHERE) sphere_box(m, d, g1, g2, p1, m->geom_size[3 * g1], margin);
    else if (t1 == SG_GEOM_CAPSULE) capsule_box(m, d, g1, g2, margin);
    else {
      int nb = box_box(m, d, g1, g2, margin);
      if (nb > 0 && getenv("SGO_DEBUG_PAIR")) fprintf(stderr, "box-box %d %d: %d contacts t=%g\n", g1, g2, nb, d->time);
    }
  }
}

/* ------------------------------------------------------------------ stage 6: constraint assembly */
static void impedance(const double* solimp_in, double pos, double margin, double* imp) { /* App. B.5 */
  double s[5];
  memcpy(s, solimp_in, sizeof s);
  s[0] = fmin(MAXIMP, fmax(MINIMP, s[0])); s[1] = fmin(MAXIMP, fmax(MINIMP, s[1]));
  s[2] = fmax(0, s[2]); s[3] = fmin(MAXIMP, fmax(MINIMP, s[3])); s[4] = fmax(1, s[4]);
  if (s[0] == s[1] || s[2] <= MINVAL) { *imp = 0.5 * (s[0] + s[1]); return; }
  double x = fabs((pos - margin) / s[2]);
  if (x >= 1 || x <= 0) { *imp = x >= 1 ? s[1] : s[0]; return; }
  double y;
  if (s[4] == 1) y = x;
  else if (x <= s[3]) y = pow(x, s[4]) / pow(s[3], s[4] - 1);
  else y = 1 - pow(1 - x, s[4]) / pow(1 - s[3], s[4] - 1);
  *imp = s[0] + y * (s[1] - s[0]);
}

static void add_row(sgo_data* d, int type, int id, double pos, double margin, int nnz, const int* col, const double* val) {
  int r = d->nefc, a = d->J_rowadr[r];
  efc_reserve(d, r + 2, a + nnz + 1);
  d->efc_type[r] = type; d->efc_id[r] = id; d->efc_pos[r] = pos; d->efc_margin[r] = margin;
  int k = 0;
  for (int i = 0; i < nnz; i++)
    if (val[i] != 0.0) {
      int j = 0; /* a contact between two bodies of one chain names their common dofs twice: one entry per column (mj_jacDifPair) */
      while (j < k && d->J_col[a + j] != col[i]) j++;
      if (j < k) { d->J_val[a + j] += val[i]; continue; }
      d->J_col[a + k] = col[i]; d->J_val[a + k] = val[i]; k++;
    }
  d->J_rowadr[r + 1] = a + k;
  d->nefc = r + 1;
}

static void make_constraint(const sgo_model* m, sgo_data* d) {
  int nv = m->nv;
  efc_reserve(d, 8, 8);
  d->nefc = 0; d->J_rowadr[0] = 0;
  int* cols = d->scr_cols;
  double* vals = d->scr_vals;
  /* equality rows, by id */
  for (int e = 0; e < m->neq; e++) {
    if (m->eq_type[e] == SG_EQ_JOINT) {
      int j = m->eq_obj1id[e], j2 = m->eq_obj2id[e]; /* joint ids (scalar joints) */
      int q1 = m->jnt_qposadr[j], d1 = m->jnt_dofadr[j];
      const double* pc = m->eq_data + 5 * e; /* polycoef */
      if (j2 < 0) {
        double one = 1.0;
        add_row(d, EFC_EQUALITY, e, d->qpos[q1] - m->qpos0[q1] - pc[0], 0, 1, &d1, &one);
      } else { /* two joints: (q1 - q1_0) = poly(q2 - q2_0), J = (+1 on joint 1, -poly' on joint 2) */
        int q2 = m->jnt_qposadr[j2];
        double dif = d->qpos[q2] - m->qpos0[q2];
        double poly = pc[0] + dif * (pc[1] + dif * (pc[2] + dif * (pc[3] + dif * pc[4])));
        double der = pc[1] + dif * (2 * pc[2] + dif * (3 * pc[3] + dif * 4 * pc[4]));
        int cc[2] = {d1, m->jnt_dofadr[j2]};
        double vv[2] = {1.0, -der};
        add_row(d, EFC_EQUALITY, e, d->qpos[q1] - m->qpos0[q1] - poly, 0, 2, cc, vv);
      }
    } else {
      int t = m->eq_obj1id[e], n = 0;
      for (int i = 0; i < nv; i++)
        if (d->ten_J[(size_t)t * nv + i] != 0) { cols[n] = i; vals[n++] = d->ten_J[(size_t)t * nv + i]; }
      add_row(d, EFC_EQUALITY, e, d->ten_length[t] - m->tendon_length0[t] - m->eq_data[5 * e], 0, n, cols, vals);
    }
  }
  /* joint limits, by joint id, lower side first */
  for (int j = 0; j < m->njnt; j++) {
    if (!m->jnt_limited[j] || m->jnt_type[j] == SG_JNT_FREE) continue;
    int dj = m->jnt_dofadr[j];
    for (int side = -1; side <= 1; side += 2) {
      double dist = side * (m->jnt_range[2 * j + (side + 1) / 2] - d->qpos[m->jnt_qposadr[j]]);
      if (dist < m->jnt_margin[j]) { double v = -side; add_row(d, EFC_LIMIT, j, dist, m->jnt_margin[j], 1, &dj, &v); }
    }
  }
  /* contacts, in detection order */
  for (int c = 0; c < d->ncon; c++) {
    contact_t* con = d->contact + c;
    con->efc_address = -1;
    if (con->dist >= con->includemargin) continue;
    if (m->njmax > 0 && d->nefc + con->dim > m->njmax) { d->warnings |= SGO_WARN_CNSTRFULL; break; }
    chainjac c1, c2;
    jac_chain(m, d, m->geom_bodyid[con->geom1], con->pos, &c1);
    jac_chain(m, d, m->geom_bodyid[con->geom2], con->pos, &c2);
    con->efc_address = d->nefc;
    for (int r = 0; r < con->dim; r++) {
      int n = 0;
      for (int k = c2.n - 1; k >= 0; k--) { cols[n] = c2.dof[k]; vals[n++] = dot3(con->frame + 3 * r, c2.jp[k]); }
      for (int k = c1.n - 1; k >= 0; k--) { cols[n] = c1.dof[k]; vals[n++] = -dot3(con->frame + 3 * r, c1.jp[k]); }
      add_row(d, con->dim == 1 ? EFC_CONTACT_FRICTIONLESS : EFC_CONTACT_ELLIPTIC, c, r == 0 ? con->dist : 0.0,
              r == 0 ? con->includemargin : 0.0, n, cols, vals);
    }
  }
  /* diagApprox, impedance, R, D, KBIP (mj_makeImpedance) */
  for (int i = 0; i < d->nefc; i++) {
    const double *solref, *solimp;
    double dA;
    int id = d->efc_id[i];
    if (d->efc_type[i] == EFC_EQUALITY) {
      solref = m->eq_solref + 2 * id; solimp = m->eq_solimp + 5 * id;
      if (m->eq_type[id] == SG_EQ_JOINT) /* joint equality: the invweights of its one or two dofs */
        dA = m->dof_invweight0[m->jnt_dofadr[m->eq_obj1id[id]]] + (m->eq_obj2id[id] >= 0 ? m->dof_invweight0[m->jnt_dofadr[m->eq_obj2id[id]]] : 0.0);
      else {
        dA = m->tendon_invweight0[m->eq_obj1id[id]];
        if (g_variant & 8) {
          dA = 0;
          for (int k = d->J_rowadr[i]; k < d->J_rowadr[i + 1]; k++) dA += d->J_val[k] * d->J_val[k] * m->dof_invweight0[d->J_col[k]];
        }
      }
    } else if (d->efc_type[i] == EFC_LIMIT) {
      solref = m->jnt_solref + 2 * id; solimp = m->jnt_solimp + 5 * id; dA = m->dof_invweight0[m->jnt_dofadr[id]];
    } else {
      const contact_t* con = d->contact + id;
      solref = con->solref; solimp = con->solimp;
      dA = m->body_invweight0[2 * m->geom_bodyid[con->geom1]] + m->body_invweight0[2 * m->geom_bodyid[con->geom2]];
    }
    d->efc_diagApprox[i] = dA;
    double imp, dmax = fmin(MAXIMP, fmax(MINIMP, solimp[1])), K, B;
    impedance(solimp, ((g_variant & 4) && d->efc_type[i] == EFC_EQUALITY) ? 0.0 : d->efc_pos[i], d->efc_margin[i], &imp);
    d->efc_R[i] = fmax(MINVAL, (1 - imp) / imp * dA);
    if (solref[0] > 0 && solref[1] > 0) { /* (timeconst, dampratio); timeconst >= 2h (refsafe) */
      double tc = fmax(solref[0], 2 * m->timestep);
      K = 1 / fmax(MINVAL, dmax * dmax * tc * tc * solref[1] * solref[1]);
      B = 2 / fmax(MINVAL, dmax * tc);
    } else { /* direct (-stiffness, -damping) */
      K = -solref[0] / fmax(MINVAL, dmax * dmax);
      B = -solref[1] / fmax(MINVAL, dmax);
      if (g_variant & 2) { K = -solref[0]; B = -solref[1]; }
    }
    d->efc_KBIP[4 * i] = K; d->efc_KBIP[4 * i + 1] = B; d->efc_KBIP[4 * i + 2] = imp; d->efc_KBIP[4 * i + 3] = 0;
  }
  /* elliptic cones: friction-row R from the normal row via impratio, regularised mu */
  for (int i = 0; i < d->nefc; i++)
    if (d->efc_type[i] == EFC_CONTACT_ELLIPTIC) {
      contact_t* con = d->contact + d->efc_id[i];
      d->efc_R[i + 1] = d->efc_R[i] / fmax(MINVAL, m->impratio);
      con->mu = con->friction[0] * sqrt(d->efc_R[i + 1] / d->efc_R[i]);
      for (int j = 2; j < con->dim; j++)
        d->efc_R[i + j] = d->efc_R[i + 1] * con->friction[0] * con->friction[0] / (con->friction[j - 1] * con->friction[j - 1]);
      i += con->dim - 1;
    }
  for (int i = 0; i < d->nefc; i++) d->efc_D[i] = 1 / d->efc_R[i];
}

static double jdot(const sgo_data* d, int row, const double* v) {
  double s = 0;
  for (int k = d->J_rowadr[row]; k < d->J_rowadr[row + 1]; k++) s += d->J_val[k] * v[d->J_col[k]];
  return s;
}

/* A = J M^-1 J' + diag(R), stored dense with a per-row nonzero pattern (mj_projectConstraint) */
static void project_constraint(const sgo_model* m, sgo_data* d) {
  int nv = m->nv, ne = d->nefc;
  if (!ne) return;
  if ((size_t)ne * ne > d->AR_cap) {
    d->AR_cap = (size_t)ne * ne * 2;
    free(d->AR);
    d->AR = (double*)malloc(sizeof(double) * d->AR_cap);
  }
  memset(d->AR, 0, sizeof(double) * (size_t)ne * ne);
  /* rows touching each dof */
  int nnz = d->J_rowadr[ne];
  if (nnz > d->dofrow_cap) { d->dofrow_cap = nnz * 2; d->dofrow = (int*)realloc(d->dofrow, sizeof(int) * d->dofrow_cap); }
  memset(d->dofrow_adr, 0, sizeof(int) * (nv + 1));
  for (int k = 0; k < nnz; k++) d->dofrow_adr[d->J_col[k] + 1]++;
  for (int i = 0; i < nv; i++) d->dofrow_adr[i + 1] += d->dofrow_adr[i];
  int* fill = d->scr_fill;
  memset(fill, 0, sizeof(int) * nv);
  if (nnz > d->scr_rowJk_cap) { d->scr_rowJk_cap = nnz * 2; d->scr_rowJk = (int*)realloc(d->scr_rowJk, sizeof(int) * d->scr_rowJk_cap); }
  int* rowJk = d->scr_rowJk; /* index into J_val for (dof,row) */
  for (int r = 0; r < ne; r++)
    for (int k = d->J_rowadr[r]; k < d->J_rowadr[r + 1]; k++) {
      int c = d->J_col[k], slot = d->dofrow_adr[c] + fill[c]++;
      d->dofrow[slot] = r; rowJk[slot] = k;
    }
  /* column j of A: w = M^-1 J_j', then scatter J_i . w over the rows i that touch supp(w) */
  size_t colcap = 0;
  int* colcount = d->AR_rownnz;
  memset(colcount, 0, sizeof(int) * ne);
  double* w = d->tmpv;
  /* pass 1 fills the dense matrix, pass 2 builds the pattern from it */
  for (int j = 0; j < ne; j++) {
    memset(w, 0, sizeof(double) * nv);
    for (int k = d->J_rowadr[j]; k < d->J_rowadr[j + 1]; k++) w[d->J_col[k]] = d->J_val[k];
    solve_ld(m, d->qLD, w);
    for (int c = 0; c < nv; c++) {
      if (w[c] == 0.0) continue;
      for (int s = d->dofrow_adr[c]; s < d->dofrow_adr[c + 1]; s++) d->AR[(size_t)d->dofrow[s] * ne + j] += d->J_val[rowJk[s]] * w[c];
    }
  }
  for (int i = 0; i < ne; i++) d->AR[(size_t)i * ne + i] += d->efc_R[i];
  for (int i = 0; i < ne; i++) {
    int n = 0;
    for (int j = 0; j < ne; j++) n += d->AR[(size_t)i * ne + j] != 0.0;
    colcount[i] = n; colcap += n;
  }
  if ((int)colcap > d->AR_colcap) { d->AR_colcap = (int)colcap * 2; d->AR_col = (int*)realloc(d->AR_col, sizeof(int) * d->AR_colcap); }
  int adr = 0;
  for (int i = 0; i < ne; i++) {
    d->AR_rowadr[i] = adr;
    for (int j = 0; j < ne; j++)
      if (d->AR[(size_t)i * ne + j] != 0.0) d->AR_col[adr++] = j;
  }
}

/* ------------------------------------------------------------------ velocity / acceleration tree walks */
/* Walk the tree carrying (w, v) resp. (w, al, a) of the running frame at the running reference point P.
 * qacc == NULL: velocities only (fills bw, bv).  Otherwise also bal, ba with the world accelerating at -gravity
 * (so that gravity enters as a fictitious force, as in mj_rne / mj_rnePostConstraint). */
static void tree_motion(const sgo_model* m, sgo_data* d, const double* qacc, int with_acc) {
  for (int c = 0; c < 3; c++) { d->bw[c] = d->bv[c] = d->bal[c] = 0; d->ba[c] = -m->gravity[c]; }
  for (int i = 1; i < m->nbody; i++) {
    int p = m->body_parentid[i];
    double w[3], v[3], al[3], a[3], P[3], r[3], t[3], t2[3];
    memcpy(w, d->bw + 3 * p, 24); memcpy(v, d->bv + 3 * p, 24); memcpy(al, d->bal + 3 * p, 24); memcpy(a, d->ba + 3 * p, 24);
    memcpy(P, d->xpos + 3 * p, 24);
    for (int k = 0; k <= m->body_jntnum[i]; k++) {
      int last = k == m->body_jntnum[i], jn = last ? -1 : m->body_jntadr[i] + k, j = last ? -1 : m->jnt_dofadr[jn]; /* j: the joint's (first) dof */
      int ty = last ? -1 : m->jnt_type[jn];
      if (last || ty == SG_JNT_HINGE || ty == SG_JNT_FREE) { /* move the reference point (to the anchor, or finally the body origin) */
        const double* Q = last ? d->xpos + 3 * i : d->xanchor + 3 * j;
        for (int c = 0; c < 3; c++) r[c] = Q[c] - P[c];
        cross3(t, w, r);
        if (with_acc) { cross3(t2, al, r); addscl3(a, t2, 1); cross3(t2, w, t); addscl3(a, t2, 1); }
        addscl3(v, t, 1);
        memcpy(P, Q, 24);
      }
      if (last) break;
      if (ty == SG_JNT_FREE) {
        /* dofs 0-2: the origin's velocity along the world axes (constant axes: no velocity-product term, mj_comVel zeroes
         * cdof_dot for them); dofs 3-5: angular velocity about the body's own axes -- all three axes turn with the body, so each
         * one's rate of change is taken with the angular velocity BEFORE the joint (the three terms of the joint's own velocity
         * cancel: w_joint x w_joint = 0), as mj_comVel does for ball / free joints */
        double w0[3] = {w[0], w[1], w[2]};
        for (int c = 0; c < 3; c++) {
          const double* u = d->xaxis + 3 * (j + c);
          double qd = d->qvel[j + c], qdd = qacc ? qacc[j + c] : 0.0;
          cross3(t, w0, u);
          if (with_acc) { addscl3(a, u, qdd); addscl3(a, t, 2 * qd); }
          addscl3(v, u, qd);
        }
        for (int c = 0; c < 3; c++) {
          const double* u = d->xaxis + 3 * (j + 3 + c);
          double qd = d->qvel[j + 3 + c], qdd = qacc ? qacc[j + 3 + c] : 0.0;
          cross3(t, w0, u);
          if (with_acc) { addscl3(al, u, qdd); addscl3(al, t, qd); }
          addscl3(w, u, qd);
        }
        continue;
      }
      const double* u = d->xaxis + 3 * j;
      double qd = d->qvel[j], qdd = qacc ? qacc[j] : 0.0;
      cross3(t, w, u); /* du/dt */
      if (ty == SG_JNT_HINGE) {
        if (with_acc) { addscl3(al, u, qdd); addscl3(al, t, qd); }
        addscl3(w, u, qd);
      } else {
        if (with_acc) { addscl3(a, u, qdd); addscl3(a, t, 2 * qd); }
        addscl3(v, u, qd);
      }
    }
    memcpy(d->bw + 3 * i, w, 24); memcpy(d->bv + 3 * i, v, 24); memcpy(d->bal + 3 * i, al, 24); memcpy(d->ba + 3 * i, a, 24);
  }
}

/* recursive Newton-Euler with qacc = 0: Coriolis/centrifugal + gravity (mj_rne) */
static void rne_bias(const sgo_model* m, sgo_data* d) {
  memset(d->qfrc_bias, 0, sizeof(double) * m->nv);
  tree_motion(m, d, NULL, 1);
  for (int b = 1; b < m->nbody; b++) {
    if (m->body_weldid[b] == 0 || m->body_mass[b] <= 0) continue;
    const double *w = d->bw + 3 * b, *al = d->bal + 3 * b;
    double c[3] = {d->xipos[3 * b] - d->xpos[3 * b], d->xipos[3 * b + 1] - d->xpos[3 * b + 1], d->xipos[3 * b + 2] - d->xpos[3 * b + 2]};
    double t[3], t2[3], f[3], n[3], Iw[3];
    memcpy(f, d->ba + 3 * b, 24);
    cross3(t, al, c); addscl3(f, t, 1);
    cross3(t, w, c); cross3(t2, w, t); addscl3(f, t2, 1);
    for (int k = 0; k < 3; k++) f[k] *= m->body_mass[b];
    mulmat3(n, d->ximat + 9 * b, al);
    mulmat3(Iw, d->ximat + 9 * b, w);
    cross3(t, w, Iw); addscl3(n, t, 1);
    chainjac cj;
    jac_chain(m, d, b, d->xipos + 3 * b, &cj);
    for (int k = 0; k < cj.n; k++) d->qfrc_bias[cj.dof[k]] += dot3(cj.jp[k], f) + dot3(cj.jr[k], n);
  }
}


/* ------------------------------------------------------------------ PGS (mj_solPGS) */
long long sgo_dbg_counters[8];
long long sgo_dbg_qcqp_hist[24]; /* debug: calls of qcqp2 by the number of Newton evaluations they ran (index 21: singular, 22: ran out of its 20) */
static int qcqp2(double* res, const double* Ain, const double* bin, const double* dd, double r) {
  double b1 = bin[0] * dd[0], b2 = bin[1] * dd[1];
  double A11 = Ain[0] * dd[0] * dd[0], A22 = Ain[3] * dd[1] * dd[1], A12 = Ain[1] * dd[0] * dd[1];
  double la = 0, v1 = 0, v2 = 0;
  sgo_dbg_counters[0]++;
  for (int it = 0; it < 20; it++) {
    sgo_dbg_counters[1]++;
    double det = (A11 + la) * (A22 + la) - A12 * A12;
    if (det < 1e-10) { res[0] = res[1] = 0; sgo_dbg_qcqp_hist[21]++; return 0; }
    double di = 1 / det, P11 = (A22 + la) * di, P22 = (A11 + la) * di, P12 = -A12 * di;
    v1 = -P11 * b1 - P12 * b2; v2 = -P12 * b1 - P22 * b2;
    double val = v1 * v1 + v2 * v2 - r * r;
    if (val < 1e-10) { sgo_dbg_qcqp_hist[it + 1]++; break; }
    double deriv = -2 * (P11 * v1 * v1 + 2 * P12 * v1 * v2 + P22 * v2 * v2), delta = -val / deriv;
    if (delta < 1e-10) { sgo_dbg_qcqp_hist[it + 1]++; break; }
    la += delta;
    if (it == 19) sgo_dbg_qcqp_hist[22]++;
  }
  res[0] = v1 * dd[0]; res[1] = v2 * dd[1];
  return la != 0;
}

static void residual(const sgo_data* d, double* res, int i, int dim) {
  int ne = d->nefc;
  for (int j = 0; j < dim; j++) {
    double s = d->efc_b[i + j];
    const double* row = d->AR + (size_t)(i + j) * ne;
    for (int k = d->AR_rowadr[i + j], e = k + d->AR_rownnz[i + j]; k < e; k++) s += row[d->AR_col[k]] * d->efc_force[d->AR_col[k]];
    res[j] = s;
  }
}

static double cost_change(const double* A, double* force, const double* old, const double* res, int dim) {
  double delta[3], change = 0;
  for (int j = 0; j < dim; j++) delta[j] = force[j] - old[j];
  for (int j = 0; j < dim; j++) {
    double s = 0;
    for (int k = 0; k < dim; k++) s += A[j * dim + k] * delta[k];
    change += 0.5 * delta[j] * s + delta[j] * res[j];
  }
  if (change > 1e-10) { memcpy(force, old, sizeof(double) * dim); change = 0; }
  return change;
}

static void sol_pgs(const sgo_model* m, sgo_data* d) {
  int ne = d->nefc;
  double* f = d->efc_force;
  double scale = 1 / (m->meaninertia * (m->nv > 1 ? m->nv : 1));
  d->solver_iter = 0;
  for (int iter = 0; iter < m->iterations; iter++) {
    double improvement = 0;
    for (int i = 0; i < ne;) {
      int dim = d->efc_type[i] == EFC_CONTACT_ELLIPTIC ? d->contact[d->efc_id[i]].dim : 1;
      double res[3], old[3], Athis[9];
      residual(d, res, i, dim);
      memcpy(old, f + i, sizeof(double) * dim);
      if (dim == 1) {
        Athis[0] = d->AR[(size_t)i * ne + i];
        f[i] -= res[0] / Athis[0];
        if (d->efc_type[i] != EFC_EQUALITY && f[i] < 0) f[i] = 0;
      } else {
        const contact_t* con = d->contact + d->efc_id[i];
        for (int j = 0; j < 3; j++)
          for (int k = 0; k < 3; k++) Athis[3 * j + k] = d->AR[(size_t)(i + j) * ne + i + k];
        /* normal or ray update */
        sgo_dbg_counters[2]++;
        if (f[i] < MINVAL) {
          sgo_dbg_counters[3]++;
          f[i] -= res[0] / Athis[0];
          if (f[i] < 0) f[i] = 0;
          f[i + 1] = f[i + 2] = 0;
        } else {
          double v[3] = {f[i], f[i + 1], f[i + 2]}, v1[3], denom = 0;
          for (int j = 0; j < 3; j++) { v1[j] = Athis[3 * j] * v[0] + Athis[3 * j + 1] * v[1] + Athis[3 * j + 2] * v[2]; denom += v[j] * v1[j]; }
          if (denom >= MINVAL) {
            double x = -(v[0] * res[0] + v[1] * res[1] + v[2] * res[2]) / denom;
            if (f[i] + x * v[0] < 0) x = -f[i] / v[0];
            for (int j = 0; j < 3; j++) f[i + j] += x * v[j];
          }
        }
        /* friction update with the normal force fixed */
        if (f[i] < MINVAL) {
          f[i + 1] = f[i + 2] = 0;
        } else {
          double Ac[4] = {Athis[4], Athis[5], Athis[7], Athis[8]}, bc[2], v[2];
          for (int j = 0; j < 2; j++) {
            bc[j] = res[j + 1];
            for (int k = 0; k < 2; k++) bc[j] -= Ac[2 * j + k] * old[1 + k];
            bc[j] += Athis[3 * (j + 1)] * (f[i] - old[0]);
          }
          int active = qcqp2(v, Ac, bc, con->friction, f[i]);
          if (active) {
            double s = v[0] * v[0] / (con->friction[0] * con->friction[0]) + v[1] * v[1] / (con->friction[1] * con->friction[1]);
            s = sqrt(f[i] * f[i] / fmax(MINVAL, s));
            v[0] *= s; v[1] *= s;
          }
          f[i + 1] = v[0]; f[i + 2] = v[1];
        }
      }
      improvement -= cost_change(Athis, f + i, old, res, dim);
      i += dim;
    }
    improvement *= scale;
    d->solver_iter = iter + 1;
    if (improvement < m->tolerance) break;
  }
}

/* primal-to-force map used for the PGS warmstart (mj_constraintUpdate) */
static void constraint_update(const sgo_data* d, const double* jar, double* f) {
  for (int i = 0; i < d->nefc; i++) {
    int ty = d->efc_type[i];
    if (ty == EFC_EQUALITY) f[i] = -d->efc_D[i] * jar[i];
    else if (ty == EFC_LIMIT || ty == EFC_CONTACT_FRICTIONLESS) f[i] = jar[i] < 0 ? -d->efc_D[i] * jar[i] : 0;
    else {
      const contact_t* con = d->contact + d->efc_id[i];
      double mu = con->mu, U[3] = {jar[i] * mu, jar[i + 1] * con->friction[0], jar[i + 2] * con->friction[1]};
      double N = U[0], T = sqrt(U[1] * U[1] + U[2] * U[2]);
      if (N >= mu * T || (T <= 0 && N >= 0)) { f[i] = f[i + 1] = f[i + 2] = 0; }
      else if (mu * N + T <= 0 || (T <= 0 && N < 0)) { for (int j = 0; j < 3; j++) f[i + j] = -d->efc_D[i + j] * jar[i + j]; }
      else {
        double Dm = d->efc_D[i] / (mu * mu * (1 + mu * mu)), NmT = N - mu * T;
        f[i] = -Dm * NmT * mu;
        f[i + 1] = -f[i] / T * U[1] * con->friction[0];
        f[i + 2] = -f[i] / T * U[2] * con->friction[1];
      }
      i += 2;
    }
  }
}

/* ------------------------------------------------------------------ mj_forward */
int sgo_forward(const sgo_model* m, sgo_data* d) {
  int nv = m->nv;
  double h = m->timestep;
  (void)h;
  /* --- position stage --- */
  kinematics(m, d);        /* 1 */
  tendons(m, d);           /* 3 (+ transmission: actuator length = gear * tendon length) */
  mass_matrix(m, d);       /* 4 */
  factor(m, d->qM, d->qLD);
  collision(m, d);         /* 5 */
  make_constraint(m, d);   /* 6 */
  project_constraint(m, d);
  /* --- velocity stage (7) --- */
  for (int t = 0; t < m->ntendon; t++) {
    double s = 0;
    for (int i = 0; i < nv; i++) s += d->ten_J[(size_t)t * nv + i] * d->qvel[i];
    d->ten_velocity[t] = s;
  }
  for (int i = 0; i < nv; i++) { /* joint springs (scalar joints; a free joint has none) and dampers */
    int j = m->dof_jntid[i], qa = m->jnt_qposadr[j];
    double spring = m->jnt_type[j] == SG_JNT_FREE ? 0.0 : -d->jnt_stiffness[j] * (d->qpos[qa] - m->qpos_spring[qa]);
    d->qfrc_passive[i] = spring - m->dof_damping[i] * d->qvel[i];
  }
  for (int t = 0; t < m->ntendon; t++) { /* tendon springs and dampers */
    double frc = -d->tendon_stiffness[t] * (d->ten_length[t] - m->tendon_lengthspring[t]) - m->tendon_damping[t] * d->ten_velocity[t];
    if (frc != 0)
      for (int i = 0; i < nv; i++) d->qfrc_passive[i] += d->ten_J[(size_t)t * nv + i] * frc;
  }
  for (int i = 0; i < d->nefc; i++) { /* reference acceleration */
    d->efc_vel[i] = jdot(d, i, d->qvel);
    d->efc_aref[i] = -d->efc_KBIP[4 * i + 1] * d->efc_vel[i] - d->efc_KBIP[4 * i] * d->efc_KBIP[4 * i + 2] * (d->efc_pos[i] - d->efc_margin[i]);
  }
  rne_bias(m, d); /* also leaves body velocities in bw/bv */
  /* gyro (velocity-stage sensor): site-frame angular velocity */
  for (int s = 0; s < m->nsensor; s++)
    if (m->sensor_type[s] == SG_SENS_GYRO) {
      int site = m->sensor_objid[s];
      mulmatT3(d->sensordata + m->sensor_adr[s], d->site_xmat + 9 * site, d->bw + 3 * m->site_bodyid[site]);
    }
  /* --- actuation (8): first-order filter state, force = gain*act + bias --- */
  memset(d->qfrc_actuator, 0, sizeof(double) * nv);
  for (int u = 0; u < m->nu; u++) {
    int t = m->actuator_trnid[u];
    double g = m->actuator_gear[u], len = g * d->ten_length[t], vel = g * d->ten_velocity[t];
    d->act_dot[u] = (d->ctrl[u] - d->act[u]) / fmax(MINVAL, m->actuator_timeconst[u]);
    double frc = m->actuator_gain[u] * d->act[u] + m->actuator_bias[3 * u] + m->actuator_bias[3 * u + 1] * len + m->actuator_bias[3 * u + 2] * vel;
    d->actuator_force[u] = frc;
    for (int i = 0; i < nv; i++) d->qfrc_actuator[i] += g * d->ten_J[(size_t)t * nv + i] * frc;
  }
  /* --- smooth acceleration (9) --- */
  for (int i = 0; i < nv; i++) d->qacc_smooth[i] = d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_actuator[i];
  solve_ld(m, d->qLD, d->qacc_smooth);
  /* --- constraint solve (10) --- */
  if (!d->nefc) {
    memcpy(d->qacc, d->qacc_smooth, sizeof(double) * nv);
    memset(d->qfrc_constraint, 0, sizeof(double) * nv);
    d->solver_iter = 0;
  } else {
    int ne = d->nefc;
    for (int i = 0; i < ne; i++) d->efc_b[i] = jdot(d, i, d->qacc_smooth) - d->efc_aref[i];
    /* warmstart: forces implied by last step's qacc, kept only if they beat f = 0 */
    for (int i = 0; i < ne; i++) d->efc_jar[i] = jdot(d, i, d->qacc_warmstart) - d->efc_aref[i];
    constraint_update(d, d->efc_jar, d->efc_force);
    if (g_variant & 16) memset(d->efc_force, 0, sizeof(double) * ne);
    double cost = 0;
    for (int i = 0; i < ne; i++) {
      double s = 0;
      const double* row = d->AR + (size_t)i * ne;
      for (int k = d->AR_rowadr[i], e = k + d->AR_rownnz[i]; k < e; k++) s += row[d->AR_col[k]] * d->efc_force[d->AR_col[k]];
      cost += d->efc_force[i] * (0.5 * s + d->efc_b[i]);
    }
    if (cost > 0) memset(d->efc_force, 0, sizeof(double) * ne);
    sol_pgs(m, d);
    memset(d->qfrc_constraint, 0, sizeof(double) * nv);
    for (int i = 0; i < ne; i++)
      for (int k = d->J_rowadr[i]; k < d->J_rowadr[i + 1]; k++) d->qfrc_constraint[d->J_col[k]] += d->J_val[k] * d->efc_force[i];
    memcpy(d->qacc, d->qfrc_constraint, sizeof(double) * nv);
    solve_ld(m, d->qLD, d->qacc);
    for (int i = 0; i < nv; i++) d->qacc[i] += d->qacc_smooth[i];
  }
  /* mj_fwdConstraint leaves this step's solution as the next solve's warmstart (both branches) */
  memcpy(d->qacc_warmstart, d->qacc, sizeof(double) * nv);
  /* --- acceleration-stage sensors (11): accelerometer = site-frame (point acceleration - gravity) --- */
  tree_motion(m, d, d->qacc, 1);
  for (int s = 0; s < m->nsensor; s++)
    if (m->sensor_type[s] == SG_SENS_ACCELEROMETER) {
      int site = m->sensor_objid[s], b = m->site_bodyid[site];
      const double *w = d->bw + 3 * b, *al = d->bal + 3 * b;
      double r[3] = {d->site_xpos[3 * site] - d->xpos[3 * b], d->site_xpos[3 * site + 1] - d->xpos[3 * b + 1], d->site_xpos[3 * site + 2] - d->xpos[3 * b + 2]};
      double a[3], t[3], t2[3];
      memcpy(a, d->ba + 3 * b, 24);
      cross3(t, al, r); addscl3(a, t, 1);
      cross3(t, w, r); cross3(t2, w, t); addscl3(a, t2, 1);
      mulmatT3(d->sensordata + m->sensor_adr[s], d->site_xmat + 9 * site, a);
    }
  return d->warnings;
}

/* mj_step = checks; mj_forward; mj_checkAcc; Euler with implicit joint damping (12) */
int sgo_step(const sgo_model* m, sgo_data* d) {
  int nv = m->nv;
  double h = m->timestep;
  d->warnings = 0;
  for (int i = 0; i < m->nq; i++)
    if (isbad(d->qpos[i])) d->warnings |= SGO_WARN_BADQPOS;
  for (int i = 0; i < nv; i++)
    if (isbad(d->qvel[i])) d->warnings |= SGO_WARN_BADQVEL;
  if (d->warnings) return d->warnings;
  sgo_forward(m, d);
  for (int i = 0; i < nv; i++)
    if (isbad(d->qacc[i])) d->warnings |= SGO_WARN_BADQACC;
  if (d->warnings & SGO_WARN_BADQACC) return d->warnings;
  double* qacc = d->qacc;
  if (m->dof_damping_any || m->implicit_tendon_damping) { /* (M + h*diag(damping)) qacc' = qfrc_smooth + qfrc_constraint */
    double* MhB = d->qLD; /* qLD is recomputed at the next forward */
    memcpy(MhB, d->qM, sizeof(double) * (size_t)nv * nv);
    for (int i = 0; i < nv; i++) MhB[(size_t)i * nv + i] += h * m->dof_damping[i];
    factor(m, MhB, MhB);
    for (int i = 0; i < nv; i++) d->tmpv2[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i];
    solve_ld(m, MhB, d->tmpv2);
    qacc = d->tmpv2;
    if (m->implicit_tendon_damping) {
      /* EXTENSION (not MuJoCo's Euler; model flag): the damper of a FIXED tendon (constant J) joins the implicit
       * term, (M + h B + h c J'J) qacc' = f, by Sherman-Morrison on the solve above -- one tendon after the other (exact for one,
       * which is all the composites have) */
      for (int t = 0; t < m->ntendon; t++) {
        if (m->wrap_type[m->tendon_adr[t]] != SG_WRAP_JOINT || !(m->tendon_damping[t] > 0)) continue;
        const double *J = d->ten_J + (size_t)t * nv, c = m->tendon_damping[t];
        double* y = d->tmpv3;
        memcpy(y, J, sizeof(double) * nv);
        solve_ld(m, MhB, y);
        double Jy = 0, Jx = 0;
        for (int i = 0; i < nv; i++) { Jy += J[i] * y[i]; Jx += J[i] * qacc[i]; }
        const double k = h * c * Jx / (1 + h * c * Jy);
        for (int i = 0; i < nv; i++) qacc[i] -= k * y[i];
      }
    }
  }
  if (g_variant & 1) memcpy(d->qacc_warmstart, qacc, sizeof(double) * nv);
  for (int u = 0; u < m->nu; u++) d->act[u] += h * d->act_dot[u];
  for (int i = 0; i < nv; i++) d->qvel[i] += h * qacc[i];
  for (int j = 0; j < m->njnt; j++) { /* mj_integratePos */
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (m->jnt_type[j] != SG_JNT_FREE) { d->qpos[qa] += h * d->qvel[da]; continue; }
    for (int c = 0; c < 3; c++) d->qpos[qa + c] += h * d->qvel[da + c];
    /* mju_quatIntegrate: turn by |w| h about w / |w|, w = the angular velocity in the BODY frame: q <- q * (cos, axis sin) */
    const double* wl = d->qvel + da + 3;
    double nw = sqrt(dot3(wl, wl)), ang = h * nw;
    if (nw > MINVAL) {
      double sn = sin(0.5 * ang), qr[4] = {cos(0.5 * ang), wl[0] / nw * sn, wl[1] / nw * sn, wl[2] / nw * sn}, *q = d->qpos + qa + 3;
      quatmul(q, q, qr);
      double nq = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
      for (int c = 0; c < 4; c++) q[c] /= nq;
    }
  }
  d->time += h;
  return d->warnings;
}

int sgo_step_many(const sgo_model* m, sgo_data** envs, int n, int nsteps, int nthreads) {
  int warn = 0;
  (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1) reduction(| : warn)
#endif
  for (int e = 0; e < n; e++)
    for (int s = 0; s < nsteps; s++) warn |= sgo_step(m, envs[e]);
  return warn;
}
