// sg_tree_plan.h -- plan tables of the TREE pipeline (sg_tree.h): grippers outside the two-finger class of sg_plan.h.
//
// the reference's four-finger gripper (data/gripper/soft_grip_four_fingers.xml:50-363; its ids are the
// comments in environment/manenv.py:11,16) has 4 finger chains of 8 links / 16 - 17 hinge dofs, 2 boxes per link, one spatial tendon
// through 8 sites per finger.  The fast kernels (sg_phase.hip, sg_rows.hip) are compiled for 2 chains x 2 bodies x 2 hinges with closed-form 4 x 4
// algebra; this plan describes the general member of the same family:
//   * K <= 8 serial hinge CHAINS hanging off static bodies, <= 24 dofs each (dense mass-matrix block per chain, L'DL in LDS),
//     any number of hinges per body, box geoms, sites;
//   * per chain at most one spatial tendon through <= 16 sites (static sites allowed anywhere along it) with at most one
//     actuator on it;
//   * accelerometers / gyros on chain sites;
//   * the composite ELEMENTS, their equalities, the static geoms and the candidate-pair table exactly as in SgPlan (sg_plan.h) --
//     the tree plan is an SgPlan whose header has nchain = 0 plus this table.  Element sliders may be limited here
//     (the four-finger file's <joint limited="true"/> default reaches the composite's sliders), the composite may sit on a body
//     with a free joint (SgPlanHeader::has_free: soft_experiments_softball.xml), and the neighbour equalities are scheduled 64 blocks
//     a round (SgPlanHeader::eq_slots).
// Everything is a fixed-size POD so that one hipMemcpy puts it on the device.
#pragma once

#define SGT_MAXCH 8     // chains
#define SGT_MAXB 64     // chain bodies, all chains
#define SGT_MAXD 128    // chain dofs, all chains
#ifndef SGT_CHD
#define SGT_CHD 24      // dofs of one chain
#endif
#define SGT_MAXG 128    // finger boxes
#define SGT_MAXS 64     // sites on chain bodies (tendon / sensor sites)
#define SGT_MAXTS 16    // sites of one spatial tendon
#define SGT_MAXSENS 32  // sensors

struct SgTreeDev {
  int K, NB, ND, NG, NS, NSENS;
  int CS;      // padded stride of a chain: max ndof over the chains rounded up to a multiple of 4.  Chain vectors are [K][CS], matrix
               // blocks [K][CS][CS] (identity / zero padding), a contact's J / W rows [3][CS]: the hot loops run to CS on every lane
  int NMAT;    // K * CS * CS: doubles of one set of per-chain blocks
  // ---- chains
  int c_body0[SGT_MAXCH], c_nbody[SGT_MAXCH], c_dof0[SGT_MAXCH], c_ndof[SGT_MAXCH], c_mat0[SGT_MAXCH];
  double c_root_pos[SGT_MAXCH][3], c_root_quat[SGT_MAXCH][4];
  // spatial tendon of chain c: sites in order; t_site >= 0: chain site (index into s_*), < 0: static, world position t_fixed
  int t_has[SGT_MAXCH], t_id[SGT_MAXCH], t_nsite[SGT_MAXCH], t_site[SGT_MAXCH][SGT_MAXTS];
  double t_fixed[SGT_MAXCH][SGT_MAXTS][3], t_k0[SGT_MAXCH], t_damping[SGT_MAXCH], t_lspring[SGT_MAXCH];
  int a_has[SGT_MAXCH], a_id[SGT_MAXCH];
  double a_gain[SGT_MAXCH], a_tc[SGT_MAXCH], a_bias[SGT_MAXCH][3], a_gear[SGT_MAXCH];
  // ---- chain bodies (flat over the chains, a chain's bodies contiguous and in kinematic order)
  int b_chain[SGT_MAXB], b_njnt[SGT_MAXB], b_dof0[SGT_MAXB];   // b_dof0: flat index of the body's first dof
  int b_nabove[SGT_MAXB];                                       // dofs of the chain that move the body (chain-local count, own ones included)
  double b_pos[SGT_MAXB][3], b_quat[SGT_MAXB][4], b_ipos[SGT_MAXB][3], b_imat[SGT_MAXB][9], b_mass[SGT_MAXB], b_invw[SGT_MAXB];
  // ---- chain dofs (flat, a chain's dofs contiguous = chain-local index + c_dof0)
  int d_body[SGT_MAXD], d_chain[SGT_MAXD], d_limited[SGT_MAXD], d_gid[SGT_MAXD];   // d_gid: dof id in the model (qpos / qvel column)
  double d_axis[SGT_MAXD][3], d_pos[SGT_MAXD][3], d_qpos0[SGT_MAXD], d_range[SGT_MAXD][2], d_margin[SGT_MAXD], d_damping[SGT_MAXD],
      d_armature[SGT_MAXD], d_stiffness[SGT_MAXD], d_springref[SGT_MAXD], d_invw[SGT_MAXD], d_limK[SGT_MAXD], d_limB[SGT_MAXD],
      d_solimp[SGT_MAXD][5];
  // ---- finger boxes (flat, in geom id order = (chain, body, geom) order)
  int g_body[SGT_MAXG], g_id[SGT_MAXG];
  double g_pos[SGT_MAXG][3], g_mat[SGT_MAXG][9], g_size[SGT_MAXG][3], g_rbound[SGT_MAXG];
  // ---- sites on chain bodies
  int s_body[SGT_MAXS];
  double s_pos[SGT_MAXS][3], s_mat[SGT_MAXS][9];
  // ---- sensors: type (SG_SENS_*), chain site, address in sensordata
  int sn_type[SGT_MAXSENS], sn_site[SGT_MAXSENS], sn_adr[SGT_MAXSENS];
};
