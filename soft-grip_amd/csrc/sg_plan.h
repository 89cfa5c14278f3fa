// sg_plan.h -- the "plan": model constants re-laid-out for the CDNA4 kernels.
//
// The generic model blob (include/softgrip_model.h) describes a kinematic tree.  The
// kernels exploit the structure every soft-gripper scene has:
//   * <=2 independent finger CHAINS: serial hinge chains (<=2 bodies, <=4 dofs) hanging
//     off world-welded bodies, carrying the box geoms, the actuated spatial tendon and
//     the accelerometer/gyro sites;
//   * N ELEMENTS: leaf bodies of a <composite> shell -- one slide joint each, static
//     parent, one capsule -- so their mass-matrix blocks are 1x1 and constant;
//   * constraints: one joint-fix equality per element, optionally followed by the
//     composite's neighbour equalities (slider e = slider e', towards the next shell
//     elements along +x, +y, +z), one tendon-fix equality over all elements, hinge
//     limits, box-capsule / box-sphere contacts.
// sg_plan_build() verifies a model has exactly this structure and refuses it otherwise.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "sg_tree_plan.h"

#define SG_MAXCH 2  // chains
#define SG_CB 2     // bodies per chain (exactly)
#define SG_CJ 2     // hinge joints per chain body (exactly)
#define SG_CD 4     // dofs per chain = SG_CB * SG_CJ
#define SG_CG 2     // box geoms per chain
#define SG_CS 2     // sensor sites per chain
#define SG_MAXSTATIC 8
#define SG_EQ_SLOTS 8   // blocks per round of the equality schedule: one per lane pair of an env's 16-lane group in the solver

struct SgChain {
  int nbody, ndof, ngeom, nsite, dof0, pad0[3];
  double root_pos[3], root_mat[9];
  // bodies (pose relative to previous chain body / chain root)
  double b_pos[SG_CB][3], b_quat[SG_CB][4], b_ipos[SG_CB][3], b_imat[SG_CB][9], b_mass[SG_CB], b_invw_tran[SG_CB];
  int b_njnt[SG_CB], b_dof0[SG_CB];
  // dofs (hinges), local index
  double j_axis[SG_CD][3], j_pos[SG_CD][3], qpos0[SG_CD], range[SG_CD][2], jmargin[SG_CD], damping[SG_CD], armature[SG_CD],
      stiffness[SG_CD], springref[SG_CD], invw[SG_CD], lim_K[SG_CD], lim_B[SG_CD], lim_solimp[SG_CD][5];
  int limited[SG_CD], d_body[SG_CD];
  // box geoms
  int g_body[SG_CG], g_id[SG_CG];
  double g_pos[SG_CG][3], g_mat[SG_CG][9], g_size[SG_CG][3], g_rbound[SG_CG];
  // spatial tendon (static site <-> chain site) and its cylinder actuator
  int has_ten, ten_id, ten_body, has_act, act_id, pad1[3];
  double ten_site[3], ten_fixed[3], ten_k0, ten_damping, ten_lspring;
  double act_gain, act_tc, act_bias[3], act_gear;
  // sensor sites
  int s_body[SG_CS], s_acc_adr[SG_CS], s_gyro_adr[SG_CS];
  double s_pos[SG_CS][3], s_mat[SG_CS][9];
};

// per-element SoA field indices into SgPlan::elem (each field is nelem doubles)
enum {
  SGE_AX = 0, SGE_AY, SGE_AZ,        // slider axis (world, constant because the parent is static)
  SGE_GX, SGE_GY, SGE_GZ,            // capsule centre at q = qpos0
  SGE_CX, SGE_CY, SGE_CZ,            // capsule axis (world)
  SGE_MASS, SGE_ARMATURE, SGE_DAMPING, SGE_K0, SGE_SPRINGREF, SGE_QPOS0,
  SGE_INVW,                          // dof_invweight0
  SGE_BINVW,                         // body_invweight0 (translational)
  SGE_COEF,                          // coefficient in the fixed tendon
  SGE_QLO, SGE_QHI,                  // slider range inside which the capsule cannot reach any static geom
  SGE_LIMITED, SGE_RLO, SGE_RHI,     // limited slider (tree plans only: sg_tree_plan.h) and its range
  SGE_KX, SGE_KY, SGE_KZ,            // centre of mass at q = qpos0 (tree plans)
  SGE_I00, SGE_I01, SGE_I02, SGE_I11, SGE_I12, SGE_I22,   // inertia about it, same frame (tree plans)
  SGE_NFIELD
};

struct SgPlanHeader {
  int nv, nu, nsensordata, ntendon, nchain, nelem, elem_dof0, iterations, nstatic, has_center, has_plane, center_geom, plane_geom;
  int ngpair;     // entries of SgPlan::gpairs (the general contact path's candidate pairs)
  int nnb;        // neighbour equality rows (0: the model has none)
  int eq_rounds;  // rounds of the equality-row schedule (SgPlan::sched), 0 when nnb == 0
  int eq_slots;   // blocks per round of that schedule: SG_EQ_SLOTS (solver kernel) or 64 (tree plans: a lane each)
  double timestep, gravity[3], tolerance, impratio, meaninertia, pgs_scale;
  // element-uniform parameters
  double cap_radius, cap_hl, cap_rbound;
  double eqj_K, eqj_B, eqj_solimp[5];            // joint-fix equality rows
  double eqt_K, eqt_B, eqt_solimp[5], eqt_invw;  // tendon-fix equality row
  double t0_k0, t0_damping, t0_lspring, t0_L0;   // the fixed tendon's own spring/damper
  // the object's FREE body (tree plans only; soft_experiments_softball.xml:8): the composite's elements hang off a body with a free
  // joint -- 7 positions (world position + quaternion), 6 dofs (world-frame linear, body-frame angular velocity).  Every element
  // field that is "world" for a static parent (axis, capsule centre / axis, centre of mass, inertia) and center_pos are then LOCAL to
  // that body.  obj_*: constants of the object's arrow-shaped mass matrix [[M_ff, B], [B', D]] (B_e = m_e (a_e ; k_e x a_e) in the
  // body frame, D_e = m_e + armature_e): sum_e B_e B_e' / D_e (upper triangle, row-major), sum_e coef_e B_e / D_e, and both with
  // D_e + h damping_e (the Euler step's M + h B)
  int nq, njnt, elem_jnt0, elem_qpos0;
  int has_free, free_jnt, free_qadr, free_dadr, center_on_free, pad_free;
  double free_q0[7], free_mass, free_com[3], free_inertia[9], free_binvw;
  double obj_BBD[21], obj_BBDh[21], obj_tenB[6], obj_tenBh[6], obj_msum, obj_mk0[3];
  // limit rows of the element sliders (tree plans only; uniform over the limited sliders, checked at build)
  double lime_K, lime_B, lime_solimp[5], lime_margin;
  int nlimited_elem;
  int t0_id;
  int t0_implicit;   // model flag opt_i[3]: FINISH integrates the fixed tendon's damper implicitly (Sherman-Morrison)
  double t0_hcT;     // h c sum_e coef_e^2 / (m_e + armature_e + h d_e), the rank-one term's denominator is 1 + t0_hcT
  // contact parameters (identical for every candidate pair, checked at build)
  double con_K, con_B, con_solimp[5], con_mu[2], con_margin;
  // static geoms
  double center_pos[3], center_radius;
  double plane_pos[3], plane_normal[3];
  double st_pos[SG_MAXSTATIC][3], st_mat[SG_MAXSTATIC][9], st_size[SG_MAXSTATIC][3], st_rbound[SG_MAXSTATIC];
  SgChain chain[SG_MAXCH];
};

// One slot of the equality-row schedule (neighbour-row models).  MuJoCo sweeps the equality rows in id order
// [fix_0, nb_0.., fix_1, nb_1.., ...]: one block of rows per element e, all acting on slider e.  Two blocks that share a
// slider must keep that order, blocks that share none commute exactly.  The plan list-schedules the BLOCKS into rounds of SG_EQ_SLOTS
// (one per lane quad of an env's group in the PGS kernel): every block sits in a later round than the blocks it depends on, so
// executing the rounds in order, all slots of a round at once, IS the sequential sweep.
struct SgEqSlot {
  int e;     // the block's element (slider); nelem (a per-env word that stays 0, with all-zero records) for an idle slot
  int p[3];  // partner slider of the block's d-th neighbour row (workspace slot d * nelem + e); nelem = no such row
};

// one candidate pair of the general contact path (sg_general.h): narrowphase routine, geometry references (kind << 16 | index) of
// geom1 / geom2 in mj_collideGeoms' order, the pair's bounding radii summed (for the sphere filter)
struct SgGenPair {
  int kind, g1, g2;
  int pad;   // the bits of a float: the pair's bounding distance rounded up (sg_plan.cpp)
};

struct SgPlan {
  SgPlanHeader h;
  // every geom pair mj_collision would look at, in its order (body pairs ascending, geoms of the first body outer): what an env on
  // the general contact path walks (sg_general.h).  The fast path's pairs (finger box x capsule / centre sphere) are part of it.
  std::vector<SgGenPair> gpairs;
  // neighbour rows: ints [9 * nelem + 3 * nnb] = out_e2[3][nelem] | out_slot[3][nelem] | in_slot[3][nelem] | row_e1[nnb] | row_e2[nnb] | row_slot[nnb]
  //   a row's workspace SLOT is d * nelem + e1 (d = its rank among the rows registered for element e1, MuJoCo's order): per env the
  //   workspace arrays nbf / nbb / nbR have 3 * nelem slots, so the phase kernel's lanes (= elements) store them coalesced
  //   out_*: the (up to 3) neighbour rows of element e: partner element / slot, -1 = none; in_slot: the (up to 3) rows that have e as second joint
  std::vector<int> nbtab;
  std::vector<SgEqSlot> sched;     // eq_rounds x eq_slots
  std::vector<double> elem;        // SGE_NFIELD x nelem
  std::vector<int> elem_geom;      // geom id of each element's capsule
  std::vector<int> elem_dofmap;    // (informational) global dof of element e = elem_dof0 + e
};

// Parses a model blob and fills the plan.  Returns false and sets err if the model is
// outside the supported class.
bool sg_plan_build(const void* blob, size_t nbytes, SgPlan* out, std::string* err);
// The same with the finger chains described by a tree table (sg_tree_plan.h) instead of SgPlanHeader::chain (h.nchain stays 0):
// any number of serial hinge chains within the SGT_* capacities.  The box references of `gpairs` are flat box indices then.
bool sg_tree_plan_build(const void* blob, size_t nbytes, SgPlan* out, SgTreeDev* tree, std::string* err);
