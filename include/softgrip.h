/* softgrip.h -- C ABI of the MI355X-native batched soft-gripper simulator.
 *
 * This is the boundary the reference crosses through mujoco_py; every
 * entry point names the mujoco_py call it replaces in reference environment/manenv.py.
 * All functions return 0 on success or a negative sg_status; sg_last_error() gives the
 * message of the last failure on the calling thread.  Per-env numeric failure is DATA
 * (the `flags` output), not an error code.
 *
 * Memory: the library owns models and batch state (device memory).  Callers own every
 * output buffer and pass raw pointers (device pointers unless stated otherwise; with
 * PyTorch-ROCm: tensor.data_ptr()).  `stream` is a hipStream_t (NULL = default stream);
 * no function synchronises the host unless it says so.
 *
 * A batch is externally synchronised (one caller thread at a time); different batches,
 * e.g. one per GPU, are fully independent.  No CPU fallback exists: creating a batch on
 * a machine without a HIP device fails with SG_ERR_NO_DEVICE.
 */
#ifndef SOFTGRIP_H
#define SOFTGRIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sg_model sg_model;
typedef struct sg_batch sg_batch;

typedef enum sg_status {
  SG_OK = 0,
  SG_ERR_INVALID = -1,     /* bad argument */
  SG_ERR_MODEL = -2,       /* blob malformed or model outside the supported class */
  SG_ERR_NO_DEVICE = -3,   /* no HIP device / device index out of range */
  SG_ERR_HIP = -4,         /* a HIP runtime call failed */
  SG_ERR_NOMEM = -5
} sg_status;

/* per-env flag bits written by sg_step/sg_reset (any of 1|2|4|8|16|32 is what mujoco_py
 * would have turned into MujocoException, reference environment/manenv.py:50) */
enum {
  SG_FLAG_BADQPOS = 1, SG_FLAG_BADQVEL = 2, SG_FLAG_BADQACC = 4,
  SG_FLAG_CONTACTFULL = 8, SG_FLAG_CNSTRFULL = 16, SG_FLAG_UNSUPPORTED_PAIR = 32
};

const char* sg_last_error(void);
const char* sg_version(void);

/* ---- model: replaces mujoco_py.load_model_from_path (manenv.py:27,36).
 * sg_model_compile: MJCF file -> model: the native compiler of the MJCF subset the soft-gripper scenes use
 *   (csrc/sg_mjcf.cpp; <include>s resolved relative to the file's directory), then sg_model_create.  flags: SG_COMPILE_*.
 * sg_mjcf_compile: the same compiler, returning the blob of softgrip_model.h (malloc'd: release with sg_blob_free) -- e.g. to
 *   store a compiled scene.  The Python host has its own implementation of the same compiler (soft-grip_amd/mjcf.py);
 *   tests/test_mjcf.py holds the two against each other.
 * sg_model_create: validates a blob and derives the kernel plan.  Host pointers throughout. */
enum {
  SG_COMPILE_NO_NEIGHBORS = 1,            /* leave out the composite's neighbour equalities (models/<scene>_fix) */
  SG_COMPILE_IMPLICIT_TENDON_DAMPER = 2   /* opt_i[3] = 1: implicit volume-tendon damper */
};
int sg_model_compile(const char* xml_path, int flags, sg_model** out);
int sg_mjcf_compile(const char* xml_path, int flags, void** blob, size_t* nbytes);
void sg_blob_free(void* blob);
int sg_model_create(const void* blob, size_t nbytes, sg_model** out);
void sg_model_destroy(sg_model* m);
int sg_model_nq(const sg_model* m);           /* positions; == nv unless the model has a free joint (7 positions, 6 dofs) */
int sg_model_nv(const sg_model* m);           /* dofs: the width of qvel / qacc_warmstart */
int sg_model_njnt(const sg_model* m);         /* joints: what the ids of sg_set_stiffness count */
int sg_model_nu(const sg_model* m);           /* == na */
int sg_model_nsensordata(const sg_model* m);
int sg_model_ntendon(const sg_model* m);
int sg_model_nelem(const sg_model* m);

/* ---- batch of n_envs independent simulations: replaces mujoco_py.MjSim(model)
 * (manenv.py:28,37), one MjSim per env.  State starts as after mj_resetData. */
int sg_batch_create(const sg_model* m, int n_envs, int device, sg_batch** out);
void sg_batch_destroy(sg_batch* b);
int sg_batch_nenvs(const sg_batch* b);
int sg_batch_device(const sg_batch* b);

/* replaces `model.jnt_stiffness[i] = k` / `model.tendon_stiffness[i] = k`
 * (manenv.py:105-108): env e uses k[e] on the listed joint and tendon ids and the model's
 * own stiffness everywhere else.  k: device or host pointer to n_envs doubles
 * (k_on_host selects); ids: host pointers.  The id sets replace those of earlier calls.
 * SYNCHRONISES THE HOST with `stream` when k is a host pointer or when the id sets differ
 * from the previous call's (once per scene; the step path never calls it): with a device k
 * and unchanged id sets it enqueues one copy on `stream` and returns. */
int sg_set_stiffness(sg_batch* b, const double* k, int k_on_host, const int* jnt_ids, int nj, const int* ten_ids, int nt,
                     void* stream);

/* replaces `data.ctrl[i] = v` (manenv.py:95,100).  ctrl: HOST pointer to nu doubles when
 * broadcast != 0 (same control for every env), else device pointer to [n_envs][nu]. */
int sg_set_ctrl(sg_batch* b, const double* ctrl, int broadcast, void* stream);

/* replaces sim.reset(); sim.forward(); then `sim_start` x sim.step()  (manenv.py:57-61)
 * for the envs whose mask byte is non-zero (mask == NULL: all).  mask: device pointer to
 * n_envs bytes.  ctrl of the reset envs is zeroed (mj_resetData).  Outputs as sg_step. */
int sg_reset(sg_batch* b, const uint8_t* mask, int sim_start, double* sens_out, int32_t* flags_out, int32_t* touch_out,
             void* stream);

/* replaces n_substeps x sim.step() followed by reading data.sensordata / data.contact
 * (manenv.py:48-49,65-85).  Device pointers, any may be NULL:
 *   sens_out  [n_envs][nsensordata] f64, the sensordata after the last substep
 *   flags_out [n_envs] int32, OR of SG_FLAG_* raised during the call; an env that raised
 *             BADQPOS/BADQVEL/BADQACC stops integrating for the rest of the call
 *   touch_out [n_envs] int32, bit (2*chain + box) set when that finger box is in contact
 *             with an object geom in the final contact list (data.contact at read time)
 * sens_stride: element stride between envs in sens_out (0 = nsensordata); lets the caller
 * write step t of a [n_envs][T][nsensordata] block directly. */
int sg_step(sg_batch* b, int n_substeps, double* sens_out, long long sens_stride, int32_t* flags_out, int32_t* touch_out,
            void* stream);

/* state access for tests and checkpointing: [n_envs][nq] (qpos), [n_envs][nv] (qvel, qacc_warmstart) and
 * [n_envs][nu] (act, ctrl); device pointers, any may be NULL.  A free joint's seven positions are the body's world position and
 * quaternion, its six velocities the linear velocity in the world frame and the angular velocity in the body frame (MuJoCo's layout). */
int sg_get_state(sg_batch* b, double* qpos, double* qvel, double* act, double* qacc_warmstart, double* ctrl, void* stream);
int sg_set_state(sg_batch* b, const double* qpos, const double* qvel, const double* act, const double* qacc_warmstart,
                 const double* ctrl, void* stream);

/* diagnostics of the last sg_step/sg_reset: [n_envs] int32 each, device pointers, may be NULL:
 * number of contacts, constraint rows and PGS sweeps of the final substep */
int sg_get_solver_stats(sg_batch* b, int32_t* ncon, int32_t* nefc, int32_t* iters, void* stream);

/* kernel pipeline (same results to round-off, all parity-tested): 3 = tree (csrc/sg_tree.h: one env per wavefront, state in LDS;
 * the only pipeline for grippers outside the two-finger class -- any number of serial finger chains up to 24 dofs, multi-site
 * tendons, limited sliders: the reference's soft_grip_four_fingers.xml -- and for an object on a free joint: the reference's
 * soft_experiments_softball.xml; selectable for fix-rows-only two-finger models),
 * 2 = rows (default for the two-finger class: chain / phase / row-parallel PGS
 * kernels, a lane quad per finger stream in the solver), 1 = split (same chain with one lane per finger stream),
 * 0 = fused (one kernel per call, everything on chip).  The env var SG_PIPELINE=fused|split|rows sets the default of
 * new batches.  A model compiled with the composite's neighbour equalities (two-joint equality rows, eq_obj2id >= 0;
 * softgrip_model.h, mjcf.py composite_neighbors=True) runs in the rows pipeline only: its batches start there and
 * sg_set_pipeline(b, 0 or 1) returns SG_ERR_MODEL. */
int sg_set_pipeline(sg_batch* b, int pipeline);

/* Contact read-out for grippers with more than 32 finger boxes (the four-finger gripper has 64; replaces the loop over
 * data.contact of manenv.py:65-85): out [n_envs][nwords] int32, bit g of an env's words (word g / 32, bit g % 32) set when moving
 * finger box g -- the model's box geoms on moving bodies in geom-id order -- touches an object geom in the contact list of the
 * last sg_step / sg_reset.  Word 0 equals touch_out for models with up to 32 boxes.  (Fused pipeline: valid when the call passed
 * touch_out = NULL.)  sg_model_nboxes: the number of such boxes. */
int sg_get_touch_words(sg_batch* b, int32_t* out, int nwords, void* stream);
int sg_model_nboxes(const sg_model* m);

/* envs per wavefront of the rows pipeline's solver kernel: 8 fills the wavefront (fix-rows-only models, chosen automatically from
 * 8185 envs on, where 8 per wavefront still give every SIMD of the chip a wavefront), 4 spreads a smaller batch over twice as many
 * wavefronts.  Same results either way (parity-tested in both).  epw: 0 = automatic, 4, 8; SG_ERR_MODEL for 8 on a model with
 * neighbour equality rows (always 4).  sg_solver_envs_per_wavefront reports what the next sg_step will use. */
int sg_set_solver_envs_per_wavefront(sg_batch* b, int epw);
int sg_solver_envs_per_wavefront(const sg_batch* b);

/* tree pipeline: how many workgroups (= envs, one wavefront each) of this batch's kernel the runtime places on one CU with the LDS the
 * launch asks for (hipOccupancyMaxActiveBlocksPerMultiprocessor) -- what bench.py reports as config.tree_workgroups_per_cu.
 * 0 for a batch that does not run the tree pipeline; negative: an error code. */
int sg_tree_workgroups_per_cu(const sg_batch* b);

/* kernel timing hook for bench.py: average device time (ms) of one sg_step/sg_reset call's kernels over the
 * calls since the last call with reset != 0, measured with HIP events on the launch
 * stream.  Synchronises the host. */
int sg_profile_enable(sg_batch* b, int enable);
int sg_profile_read(sg_batch* b, int reset, double* avg_ms, long long* launches);
/* the same for the dominant kernel alone: average device time (ms) of one solver-kernel launch (sg_pgs_rows_kernel / sg_pgs_kernel;
 * one per substep) over the calls since the last reset; the split and rows pipelines only (0 launches otherwise) */
int sg_profile_read_solver(sg_batch* b, int reset, double* avg_ms, long long* launches);

#ifdef __cplusplus
}
#endif
#endif /* SOFTGRIP_H */
