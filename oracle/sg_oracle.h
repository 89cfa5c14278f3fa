/* sg_oracle.h -- CPU restatement (plain C, fp64, one env) of MuJoCo's mj_step for the
 * soft-gripper model class.  TEST INFRASTRUCTURE ONLY: this is the checker for the HIP
 * path (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).  Nothing in the
 * product package may import, link or call it.
 *
 * PARITY UNPINNED: the reference (mbed92/soft-grip) delegates this arithmetic to the
 * third-party MuJoCo library through mujoco_py (reference environment/manenv.py:2,27-28,
 * 49,57-58); neither is vendored, version-pinned or installable in the build container,
 * and the reference ships no golden vectors.  The algorithm below restates MuJoCo's
 * documented pipeline; it is pinned only by analytic known-answer
 * tests (tests/test_oracle_kat.py) and harness fixtures (tests/golden/).
 */
#ifndef SG_ORACLE_H
#define SG_ORACLE_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct sgo_model sgo_model;
typedef struct sgo_data sgo_data;

/* warning bits returned by sgo_step / sgo_forward (mujoco_py turns any MuJoCo warning into
 * MujocoException, reference environment/manenv.py:50) */
enum {
  SGO_WARN_BADQPOS = 1, SGO_WARN_BADQVEL = 2, SGO_WARN_BADQACC = 4,
  SGO_WARN_CONTACTFULL = 8, SGO_WARN_CNSTRFULL = 16, SGO_WARN_UNSUPPORTED_PAIR = 32
};

sgo_model* sgo_model_load(const void* blob, size_t nbytes, char* err, size_t errlen);
void sgo_model_free(sgo_model*);
sgo_data* sgo_data_new(const sgo_model*);
void sgo_data_free(sgo_data*);

int sgo_nv(const sgo_model*);
int sgo_nq(const sgo_model*);    /* = nv unless the model has a free joint (7 positions, 6 dofs) */
int sgo_njnt(const sgo_model*);
int sgo_nu(const sgo_model*);
int sgo_nsensordata(const sgo_model*);
int sgo_ntendon(const sgo_model*);

void sgo_reset(const sgo_model*, sgo_data*);   /* mj_resetData  (manenv.py:57) */
int sgo_forward(const sgo_model*, sgo_data*);  /* mj_forward    (manenv.py:58) */
int sgo_step(const sgo_model*, sgo_data*);     /* mj_step       (manenv.py:49) */

/* state / parameter access (pointers stay valid for the life of the data) */
double* sgo_qpos(sgo_data*);
double* sgo_qvel(sgo_data*);
double* sgo_act(sgo_data*);
double* sgo_ctrl(sgo_data*);
double* sgo_qacc(sgo_data*);
double* sgo_qacc_warmstart(sgo_data*);
double* sgo_sensordata(sgo_data*);
double* sgo_jnt_stiffness(sgo_data*);    /* per-env copy of model.jnt_stiffness    (manenv.py:106) */
double* sgo_tendon_stiffness(sgo_data*); /* per-env copy of model.tendon_stiffness (manenv.py:108) */
double* sgo_ten_length(sgo_data*);
double* sgo_qfrc_bias(sgo_data*);
double* sgo_qM(sgo_data*);               /* dense nv x nv */
double* sgo_site_xpos(sgo_data*);
double* sgo_efc_force(sgo_data*);
double* sgo_efc_AR(sgo_data*);   /* dense (A + R), nefc x nefc, of the last forward pass */
double* sgo_efc_b(sgo_data*);    /* res = (A + R) f + b */
int* sgo_efc_type(sgo_data*);    /* 0 equality, 3 limit, 7 elliptic contact (three consecutive rows) */
int* sgo_efc_id(sgo_data*);
void sgo_contact_friction(const sgo_data*, int i, double* mu5);
int sgo_ncon(const sgo_data*);
int sgo_nefc(const sgo_data*);
int sgo_solver_iter(const sgo_data*);
/* contact i: geom ids, dist, pos[3], frame[9] (normal first) */
void sgo_contact(const sgo_data*, int i, int* geom1, int* geom2, double* dist, double* pos3, double* frame9);

/* sensitivity switches for scripts/d5_sensitivity.py (bit mask, 0 = the restatement as tested); see sg_oracle.c */
void sgo_set_variant(int bits);

/* batched helper for the CPU baseline: run `nsteps` mj_step on each of n independent envs */
int sgo_step_many(const sgo_model*, sgo_data** envs, int n, int nsteps, int nthreads);

#ifdef __cplusplus
}
#endif
#endif
