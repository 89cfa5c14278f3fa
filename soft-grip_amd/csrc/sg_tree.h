// sg_tree.h -- the TREE pipeline: mj_step for grippers outside the two-finger class (sg_tree_plan.h), one env per wavefront.
//
// What it replaces: the same calls as the fast kernels -- sim.step() / sim.reset() / sim.forward() of reference
// environment/manenv.py:48-61 -- for the reference's four-finger gripper (data/gripper/soft_grip_four_fingers.xml; SURVEY.md 8(f)
// rank 4).  The stages follow the oracle's restatement of mj_forward / mj_step (oracle/sg_oracle.c: kinematics, tendons,
// mass_matrix + factor, collision, make_constraint, rne_bias, the actuation and smooth-acceleration stages, sol_pgs, the sensor
// stage, Euler with implicit joint damping), restructured for the model class:
//   * a dense mass-matrix block per finger chain (<= 24 x 24) with MuJoCo's L'DL, its inverse by columns; the composite's sliders
//     are 1 x 1 blocks;
//   * A = J M^-1 J' is never formed: a constraint row keeps J and W = M^-1 J', the sweep keeps the accelerations a = M^-1 J' f;
//   * the joint-fix / limit rows of the sliders commute among themselves (each touches its own slider): one row per lane; with the
//     composite's neighbour equalities the equality BLOCKS [fix_e, e's neighbour rows] run by the plan's list schedule, 64 blocks a
//     round; a chain's limit rows run serially on one lane per chain; contacts run as one stream per chain (they commute across
//     chains unless they share a slider) or, when they do not, serially in mj_collision's order with the lanes of the wavefront
//     spread over the dofs of the contact's chain block(s);
//   * collision walks the plan's candidate-pair table (SgPlan::gpairs = mj_collision's pair order) 64 pairs at a time:
//     bounding tests per lane, hits ranked by pair index, one lane per hit in the narrowphase (sg_math.h / sg_general.h);
//   * a FREE OBJECT (reference data/gripper/soft_experiments_softball.xml:8: the composite on a body with a free joint) is an object
//     block with an arrow-shaped mass matrix, solved through a 6 x 6 Schur complement in the body's frame; its joint-fix rows run one
//     after the other (every one moves the body), its contacts carry six object columns.
//
// The code is BULK-SYNCHRONOUS: parallel loops over work items (SGT_PAR), single-lane sections (SGT_ONE) and barriers (SGT_SYNC)
// between them; lanes talk through the env's LDS block and its global work space only.  That is what lets tests/emu run the very
// same source on the host (a parallel loop becomes a serial loop, a wavefront sum the identity) against the oracle -- and the
// sanitizers over it.  On the device the env's state lives in LDS for the whole call (all substeps in one launch).
#pragma once
#include <stdint.h>
#include <string.h>

#include "../../include/softgrip.h"
#include "../../include/softgrip_model.h"
#include "sg_general.h"
#include "sg_plan.h"
#if defined(SGT_EMU_SEPARATE)
#include <stdlib.h>

#include <utility>
#include <vector>
#endif

#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define SGT_DEVICE 1
#else
#define SGT_DEVICE 0
#endif

namespace sgt {
using namespace sgm;

#ifndef SGT_DIET
#define SGT_DIET 0x1ff   // which groups of arrays live in the env's work space instead of LDS (lds_carve): bits 0 - 5 the build-only groups,
                         // 6 M^-1, 7 the sliders' 1 / m (scenes without a free object), 8 the capsule centres of a free object's scene.
                         // All set: 37 KB of LDS for the four-finger scene, FOUR workgroups per CU (129 against 94 k env-steps/s with
                         // M^-1 and 1 / m in LDS, r04v profile)
#endif
#define SGT_MAXCON 128   // contacts of an env
#define SGT_MAXHIT 256   // candidate pairs that pass the bounding tests
#define SGT_HITREC 8     // contacts one pair can produce (box - box)
#define SGT_RECW 10      // doubles of a staged narrowphase record: dist, pos[3], n[3], tangent hint[3]
#define SGT_LDS_HEADER 2   // doubles at the head of the env's LDS block (lds_carve)
#define SGT_CSC 56       // scalar doubles of a contact record in the work space
#define SGT_LROW 6       // doubles of a chain limit row: dof, sign, R, b, f, 1 / (A + R)
#ifndef SGT_LROW_AHEAD
#define SGT_LROW_AHEAD 4 // register sets of the sweep's chain-limit-row pass (lookahead + 1).  8 measured: four-finger 196.6 -> 193.0 k, free ball 51.7 -> 49.8 k (r05 t9)
#endif

struct TreeArgs {
  const SgPlanHeader* H;
  const SgTreeDev* T;
  const double* elem;        // SgPlan::elem (SoA over elements)
  const SgGenPair* gpairs;
  const SgEqSlot* sched;     // neighbour-row models: SgPlan::sched (eq_rounds x 64 blocks) and SgPlan::nbtab (out_e2 | out_slot | in_slot, [3][N] each)
  const int* nbtab;
  double *qpos, *qvel, *warm, *act, *ctrl;   // [n][nq] / [n][nv], [n][nu]
  const double* kenv;
  const int *kmask_jnt, *kmask_ten;
  const unsigned char* mask;  // mode 1: envs to reset (nullptr = all)
  double* sens;
  long long sens_stride;
  int *flags, *touch, *touch_words, *ncon, *nefc, *iters;   // touch_words: [n][2], bit g = finger box g touches an object geom
  double* cws;               // per-env work space, cws_stride doubles each
  long long cws_stride;
  int nenv, nsub, mode;      // mode 1: reset + one forward without integration, then nsub steps
  unsigned long long* secprof;   // profiling build (-DSG_SECTION_PROF) only: cycle sums per section, else unused
};

// work-space layout (doubles): staged narrowphase records | contact rows | the chains' mass-matrix blocks
SG_HD long long cws_row_doubles(int CS) { return 12LL * CS + SGT_CSC; }   // 2 blocks x (J, W) x 3 rows x CS + scalars


// scalar part of a contact record
enum { CS_A = 0, CS_B = 6, CS_F0 = 9, CS_R = 12, CS_INVM = 13, CS_JS = 14, CS_SL = 17, CS_C1 = 18, CS_N1 = 19, CS_C2 = 20, CS_N2 = 21,
       CS_ROWS = 22, CS_TOUCH = 23, CS_OBJ = 24 /* the contact touches the free object */, CS_JO = 25 /* [3][6]: its rows on the object's free dofs, body frame */,
       CS_TMP = 43 /* [12]: between the phases of the rows' build: J v, J a_smooth, J a_warm, body invweights, blocks, distance */,
       CS_PE = 43 /* [7], once the rows are built (the temporaries are done with): the friction block's inverse and eigen-decomposition, contact_block_constants */ };
// the free object's block in LDS (S.of[..]); body frame unless said otherwise
enum { OF_P = 0, OF_Q = 3, OF_R = 7, OF_VW = 16 /* world */, OF_VL = 19, OF_WL = 22 /* (v, w) contiguous */, OF_WARM = 25, OF_ASM = 31, OF_AF = 37, OF_GF = 43, OF_SINV = 49,
       OF_CEN = 85 /* world */, OF_GL = 88, OF_CTEN = 91, OF_BIAS = 97, OF_X = 103, OF_Y = 109, OF_WB = 115 /* OF_WARM: dof coordinates (world translations), kept
       across substeps; OF_WB: the same in the body frame of this substep */, OF_MFF = 121 /* upper triangle of M_ff, 21 */, OF_TMP = 142, OF_N = 160 };

struct Lds {
  double *q, *v, *warm, *asm_, *aF, *fs, *fc, *bias, *tenJ, *kd, *qacc;
  double *xpos, *xmat, *xipos, *ximat, *bw, *bal, *ba, *bf, *bn;
  double *anchor, *axis, *gpos, *gmat, *gsz, *spos;
  double *L, *Minv, *tmpP;
  double *qe, *ve, *we, *asme, *ae, *fse, *ffix, *bfix, *Rfix, *flim, *blim, *Rlim, *ke;
  double *einvm, *ecoef, *ecen, *Ifix, *Ilim;   // 1 / (m + armature), tendon coefficient, capsule centres [3][N], 1 / (A + R) of the fix / limit rows
  double *lrow, *seg, *chs, *cf, *red, *swc, *ctx;   // swc: the step's scalars for the sweep function (SWC_*); ctx: those the stage functions hand on (CTX_*)
  double *nbf, *nbb, *nbR, *nbI, *nbA;   // neighbour equality rows by slot d * N + e (the d-th row registered for element e): force, b, R, 1 / (A + R); free object: A + R
  double *nbq, *fixq;   // grippers with neighbour rows (no free object): the rows' sweep constants PACKED for the pipelined equality rounds of tree_sweep -- nbq[4 k] = R, b, 1 / (A + R), 1 / m of the partner; fixq[4 e] = b, R, 1 / (A + R), 1 / m of the fix row
  double *frow;   // free object: the joint-fix rows' constants for the serial sweep, [N][5]: b, R, A + R, 1 / (A + R), 1 / D
  double *of, *Be, *Ce, *Afix;   // free object (plans with has_free): scalars (OF_*), B_e [N][6], C_e = -S^-1 B_e / D_e [N][6], the fix rows' diagonals A + R
  int *hit_pair, *hit_sorted, *hit_cnt, *hit_off, *con_src, *con_chain, *icnt;
  double* csc;   // LDS copies of the first `ncache` contacts' scalar records (the sweeps read them 30 times; the rest stay in the work space)
  int ncache;
};
enum { IC_NHIT = 0, IC_NCON, IC_SERIAL, IC_NLIVE, IC_NPURE, IC_NLIM0 /* + chain */, IC_NLEV = IC_NLIM0 + SGT_MAXCH /* levels of the contact schedule */, IC_N };
static_assert(IC_N <= 32, "S.icnt holds 32 counters");
// per-chain scalars in LDS (chs[c * CHS_N + ..])
enum { CHS_TLEN = 0, CHS_TVEL, CHS_TFRC, CHS_AFRC, CHS_ACTDOT, CHS_ACT, CHS_CTRL, CHS_KT, CHS_N };

// The env's arrays.  base: its LDS block; gbase: the part of its work space that backs the arrays only their own lane (or a later
// phase behind a barrier) touches -- the L'DL blocks, the sliders' sweep constants and build-only state: 35 KB of the four-finger
// scene's 113 KB, which is what lets two workgroups share a CU's LDS (the loads are coalesced and L2-resident).  Returns the LDS
// bytes; *gdoubles the doubles taken from gbase.
#if !SGT_DEVICE && defined(SGT_EMU_SEPARATE)
// Host emulation, checking build (tests/emu, `make sep`): every array of the carve is a heap block of its own, EXACT in size, so that
// AddressSanitizer sees an access one element past ANY array -- inside the env's one LDS block / work space such an access lands in the
// neighbouring array and shows, if at all, as a wrong number on some other layout.  The driver owns the pool (blocks are handed out in
// the carve's order, the same on every call) and poisons the LDS-class blocks before a launch.
struct SepPool { std::vector<std::pair<void*, size_t>> lds, glob; size_t il = 0, ig = 0; double* part[3] = {nullptr, nullptr, nullptr}; };   // part: staged records, contact rows, mass-matrix blocks
inline SepPool*& sep_pool() { static SepPool* p = nullptr; return p; }
inline double* sep_part(int k, size_t n) { SepPool* sp = sep_pool(); if (!sp->part[k]) sp->part[k] = (double*)calloc(n ? n : 1, sizeof(double)); return sp->part[k]; }
inline void* sep_take(std::vector<std::pair<void*, size_t>>& v, size_t& i, size_t bytes) {
  if (i == v.size()) v.push_back({calloc(bytes ? bytes : 1, 1), bytes});
  if (v[i].second != bytes) abort();   // (the carve's order and sizes are a function of the model alone)
  return v[i++].first;
}
#endif
SG_HD size_t lds_carve(Lds& L, double* base, const SgTreeDev& T, int N, int has_free, double* gbase, size_t* gdoubles, int nnb = 0, size_t* used_out = nullptr) {
  L = Lds();   // (every pointer null until assigned: an array the carve forgets faults on the host emulation instead of reading the stack's leftovers)
  double *p = base + SGT_LDS_HEADER, *g = gbase;   // (the block's first words: the launch's argument segment for the called stages, tree_stage)
#if !SGT_DEVICE && defined(SGT_EMU_SEPARATE)
  SepPool* const sp = (base != reinterpret_cast<double*>((uintptr_t)4096)) ? sep_pool() : nullptr;   // (the sizing calls carve from address 4096)
  if (sp) sp->il = sp->ig = 0;
  auto take = [&](size_t n) { double* r = p; p += (n + 1) & ~(size_t)1; return sp ? (double*)sep_take(sp->lds, sp->il, n * sizeof(double)) : r; };
  auto takeg = [&](size_t n) { double* r = g; g += (n + 1) & ~(size_t)1; return sp ? (double*)sep_take(sp->glob, sp->ig, n * sizeof(double)) : r; };
#else
  auto take = [&](size_t n) { double* r = p; p += (n + 1) & ~(size_t)1; return r; };
  auto takeg = [&](size_t n) { double* r = g; g += (n + 1) & ~(size_t)1; return r; };
#endif
  const int ND = T.ND, NB = T.NB;
  // r04: LDS holds what the SWEEP touches (accelerations, forces, limit rows) and what the PAIR WALK touches (capsule centres, box poses);
  // everything only the once-per-substep build stages read or write -- kinematics, body poses and RNE temporaries, the sliders' state,
  // tendon segments, M^-1 (whose rows the sweep prefetches) -- sits in the env's work space (coalesced, L2 / Infinity-Cache resident):
  // 37.5 KB instead of 76 for the four-finger scene, i.e. FOUR workgroups per CU (one per SIMD) instead of two
  auto tk = [&](int bit, size_t n) { return (SGT_DIET >> bit) & 1 ? takeg(n) : take(n); };   // (SGT_DIET: which groups live in the work space)
  L.q = take(ND); L.v = take(ND); L.warm = take(ND); L.asm_ = take(ND); L.aF = take(T.K * T.CS);   // (read joint by joint by the one-lane-per-chain stages: LDS)
  L.fs = tk(0, ND); L.fc = tk(0, ND); L.bias = tk(0, ND); L.tenJ = tk(0, ND); L.kd = tk(0, ND); L.qacc = tk(0, ND);
  L.xpos = tk(1, 3 * NB); L.xmat = tk(1, 9 * NB); L.xipos = tk(1, 3 * NB); L.ximat = tk(1, 9 * NB); L.bw = tk(1, 3 * NB);
  L.bal = tk(1, 3 * NB); L.ba = tk(1, 3 * NB); L.bf = tk(1, 3 * NB); L.bn = tk(1, 3 * NB);
  L.anchor = tk(2, 3 * ND); L.axis = tk(2, 3 * ND); L.gpos = take(3 * T.NG); L.gmat = take(9 * T.NG); L.gsz = take(3 * T.NG); L.spos = tk(2, 3 * T.NS);
  L.L = takeg(T.NMAT); L.Minv = tk(6, T.NMAT); L.tmpP = tk(3, T.K * T.CS);   // (M^-1: the sweep's limit rows prefetch its rows, W = J M^-1 reads it lane = word)
  L.qe = takeg(N); L.ve = tk(4, N); L.we = tk(4, N); L.asme = tk(4, N); L.ae = take(N); L.fse = takeg(N); L.ffix = take(N);
  L.bfix = takeg(N); L.Rfix = takeg(N); L.flim = take(2 * N); L.blim = takeg(2 * N); L.Rlim = takeg(2 * N); L.ke = takeg(N);
  // 1 / m: in registers for the sweeps (the free object's serial rows read it per contact: LDS there).  The capsule centres: LDS for the pair
  // walk -- but a free object's scene has few candidate pairs and its own LDS arrays (B_e, C_e, the rows' constants): work space there
  L.einvm = has_free ? take(N) : tk(7, N); L.ecoef = takeg(N);
  L.ecen = (has_free && ((SGT_DIET >> 8) & 1)) ? takeg(3 * N) : take(3 * N);
  L.Ifix = takeg(N); L.Ilim = takeg(2 * N);
  L.lrow = take(SGT_LROW * 2 * ND); L.seg = tk(5, 4 * T.K * SGT_MAXTS); L.chs = tk(5, CHS_N * SGT_MAXCH); L.cf = take(3 * SGT_MAXCON);
  L.red = take(16); L.swc = take(16); L.ctx = take(16);
  L.of = take(has_free ? OF_N : 0); L.Be = take(has_free ? 6 * N : 0); L.Ce = take((has_free && nnb) ? 6 * N : 0); L.Afix = takeg(has_free ? N : 0);   // (C_e: kept for the neighbour-row blocks only -- the plain rows recompute it, free_fix_rows)
  L.frow = take(has_free ? 4 * N : 0);
  L.nbf = takeg(nnb ? 3 * N : 0); L.nbb = takeg(nnb ? 3 * N : 0); L.nbR = takeg(nnb ? 3 * N : 0); L.nbI = takeg(nnb ? 3 * N : 0);
  L.nbA = takeg((nnb && has_free) ? 3 * N : 0);
  L.nbq = takeg((nnb && !has_free) ? 12 * N : 0); L.fixq = takeg((nnb && !has_free) ? 4 * N : 0);
  if (gdoubles) *gdoubles = (size_t)(g - gbase);
  int* ip = (int*)p;
  L.hit_pair = ip; ip += SGT_MAXHIT;
  L.hit_sorted = ip; ip += SGT_MAXHIT;
  L.hit_cnt = ip; ip += SGT_MAXHIT;
  L.hit_off = ip; ip += SGT_MAXHIT;
  L.con_src = ip; ip += SGT_MAXCON;
  L.con_chain = ip; ip += SGT_MAXCON;
  L.icnt = ip; ip += 32;
#if !SGT_DEVICE && defined(SGT_EMU_SEPARATE)
  if (sp) {
    auto takei = [&](size_t n) { return (int*)sep_take(sp->lds, sp->il, n * sizeof(int)); };
    L.hit_pair = takei(SGT_MAXHIT); L.hit_sorted = takei(2 * SGT_MAXHIT); L.hit_cnt = L.hit_sorted + SGT_MAXHIT; L.hit_off = takei(SGT_MAXHIT);   // (hit_sorted + hit_cnt: ONE array to the pair walk's block lists)
    L.con_src = takei(SGT_MAXCON); L.con_chain = takei(SGT_MAXCON); L.icnt = takei(32);
  }
#endif
  // what is left of the LDS up to the next occupancy step (160 KB / k workgroups per CU) caches contact scalars
  // (the hardware hands LDS out in granules -- a workgroup's request is rounded up -- so a share is taken a granule short of 160 KB / k:
  //  r04 measured 53 920 B per workgroup, 3 x which is under 160 KB, still running TWO per CU)
  const size_t used = (size_t)((char*)ip - (char*)base), total = 160 * 1024;
  if (used_out) *used_out = used;
  const size_t kper = used + 2560 < total ? total / (used + 2560) : 1, share = (total / (kper ? kper : 1)) / 2560 * 2560 - 2560;
  const size_t room = share > used ? share - used : 0;
  size_t nc = room / (SGT_CSC * sizeof(double));
  if (nc > SGT_MAXCON) nc = SGT_MAXCON;
  L.ncache = (int)nc;
  L.csc = (double*)ip;
#if !SGT_DEVICE && defined(SGT_EMU_SEPARATE)
  if (sp) L.csc = (double*)sep_take(sp->lds, sp->il, nc * SGT_CSC * sizeof(double));
#endif
  return used + nc * SGT_CSC * sizeof(double);
}
SG_HD size_t lds_bytes(const SgTreeDev& T, int N, int has_free = 0, int nnb = 0) {
  Lds L;
  return lds_carve(L, reinterpret_cast<double*>((uintptr_t)4096), T, N, has_free, reinterpret_cast<double*>((uintptr_t)4096), nullptr, nnb);
}
SG_HD size_t lds_used_bytes(const SgTreeDev& T, int N, int has_free = 0, int nnb = 0) {   // without the contact-scalar cache that fills the share
  Lds L;
  size_t u = 0;
  lds_carve(L, reinterpret_cast<double*>((uintptr_t)4096), T, N, has_free, reinterpret_cast<double*>((uintptr_t)4096), nullptr, nnb, &u);
  return u;
}
SG_HD size_t gws_doubles(const SgTreeDev& T, int N, int has_free, int nnb);
SG_HD long long cws_doubles(const SgTreeDev& T, int N, int has_free, int nnb = 0) {
  long long n = (long long)SGT_MAXHIT * SGT_HITREC * SGT_RECW + (long long)SGT_MAXCON * cws_row_doubles(T.CS) + T.NMAT + (long long)gws_doubles(T, N, has_free, nnb);
#ifdef SG_DEBUG_WORK   // (debugging build: the env's LDS block is copied behind its work space when a launch ends, scripts/dev/work_diff.py)
  n += (long long)(lds_bytes(T, N, has_free, nnb) / sizeof(double));
#endif
  return n;
}
SG_HD size_t gws_doubles(const SgTreeDev& T, int N, int has_free, int nnb) {   // the work-space doubles behind the global-backed arrays
  Lds L;
  size_t n = 0;
  lds_carve(L, reinterpret_cast<double*>((uintptr_t)4096), T, N, has_free, reinterpret_cast<double*>((uintptr_t)4096), &n, nnb);
  return n;
}

// section stamps (profiling build only: build_native.py --prof, scripts/tree_section_profile.py): lane 0 adds the cycles since the
// previous stamp to secprof[k]
#if defined(SG_SECTION_PROF) && defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define SGT_STAMP(k)                                                                  \
  do {                                                                                \
    if (threadIdx.x == 0) {                                                           \
      const long long t_ = clock64();                                                 \
      atomicAdd(&A.secprof[k], (unsigned long long)(t_ - sgt_t_last));                \
      sgt_t_last = t_;                                                                \
    }                                                                                 \
  } while (0)
#define SGT_STAMP_INIT() long long sgt_t_last = clock64()
#define SGT_STAMP_RESET() sgt_t_last = clock64()   /* after a called stage that kept its own stamps */
#else
#define SGT_STAMP_RESET() ((void)0)
#define SGT_STAMP(k) ((void)0)
#define SGT_STAMP_INIT() ((void)0)
#endif

// a parallel loop over the composite's elements whose per-item constants live in a small per-lane array across loops (the sweep's row
// constants: read from the work space ONCE, not once per sweep): e the element, t its slot in the lane's array (N <= 256: four a lane)
#define SGT_NSLOT (SGT_DEVICE ? 4 : 256)
#if SGT_DEVICE
#define SGT_PAR_SLOT(e, t, n) _Pragma("unroll") for (int t = 0, e = (int)threadIdx.x; t < 4; t++, e += 64) if (e < (n))
#elif defined(SGT_EMU_REVERSE)
#define SGT_PAR_SLOT(e, t, n) for (int e = (n) - 1, t = e; e >= 0; e--, t = e)
#else
#define SGT_PAR_SLOT(e, t, n) for (int e = 0, t = 0; e < (n); e++, t = e)
#endif
#if SGT_DEVICE
#define SGT_FIRST ((int)threadIdx.x)
#define SGT_STRIDE 64
#define SGT_PAR(i, n) for (int i = (int)threadIdx.x; i < (n); i += 64)
#define SGT_ONE if (threadIdx.x == 0)
#if defined(SGT_X_ROWS1LANE)
#define SGT_ROW_LANES if (threadIdx.x == 0)
#else
#define SGT_ROW_LANES if (threadIdx.x < 8)   // free_fix_rows: the free body's serial joint-fix rows on eight lanes
#endif
#define SGT_SYNC() __syncthreads()
// cross-lane moves without LDS: DPP on the two halves of a double (row_ror:n = 0x120 + n, rotation inside a row of 16 lanes)
template <int CTRL>
__device__ __forceinline__ double dpp64(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// sum over the 16 lanes of a row (a LANE GROUP: one finger chain's lanes in the sweep), result in all 16: a butterfly of rotations
__device__ __forceinline__ double rowsum16(double x) {
  x += dpp64<0x128>(x);
  x += dpp64<0x124>(x);
  x += dpp64<0x122>(x);
  x += dpp64<0x121>(x);
  return x;
}
__device__ __forceinline__ double readlane64(double x, int l) {   // l uniform
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
// sum over the wavefront, result in all lanes: the rows by rotations, the four rows by scalar reads (was six ds_bpermute round trips)
__device__ __forceinline__ double wsum(double x) {
  x = rowsum16(x);
  return ((readlane64(x, 0) + readlane64(x, 16)) + readlane64(x, 32)) + readlane64(x, 48);
}
__device__ __forceinline__ double wmax(double x) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) x = fmax(x, __shfl_xor(x, m, 64));
  return x;
}
__device__ __forceinline__ int lds_inc(int* p) { return atomicAdd(p, 1); }
#else
#define SGT_FIRST 0
#define SGT_STRIDE 1
#if defined(SGT_EMU_REVERSE)
// host emulation, order-checking build (tests/emu `make rev`): every parallel loop runs its items in DESCENDING order.  A parallel loop's
// items must not depend on one another (on the device they run in lockstep, 64 at a time); the ascending serial loop of the normal
// emulation satisfies any dependency of item i on an item j < i by accident.  Same results in both orders = no such dependency.
#define SGT_PAR(i, n) for (int i = (n) - 1; i >= 0; i--)
#else
#define SGT_PAR(i, n) for (int i = 0; i < (n); i++)
#endif
#define SGT_ONE if (true)
#define SGT_ROW_LANES if (true)
#define SGT_SYNC() ((void)0)
inline double wsum(double x) { return x; }
inline double wmax(double x) { return x; }
inline int lds_inc(int* p) { return (*p)++; }
#endif

// solve (L'DL) x = x for one chain block (MuJoCo's mj_solveLD on a serial chain: dof_parentid[k] = k - 1); Lc: nd x nd, row-major,
// L[k][i] (i < k) below the diagonal, D on it
SG_HD void chain_solve(const double* Lc, int nd, double* x) {
  for (int k = nd - 1; k >= 1; k--) {
    const double xk = x[k];
    for (int i = k - 1; i >= 0; i--) x[i] -= Lc[k * nd + i] * xk;
  }
  for (int k = 0; k < nd; k++) x[k] /= Lc[k * nd + k];
  for (int k = 1; k < nd; k++) {
    double s = x[k];
    for (int i = k - 1; i >= 0; i--) s -= Lc[k * nd + i] * x[i];
    x[k] = s;
  }
}
// the same solve with x in registers: fully unrolled over the SGT_CHD capacity, guarded by nd (LDS reads of L only, no dependent
// read-modify-write chain through LDS: 9 k instead of 100 k cycles for the M^-1 columns).  Same operations in the same order.
// Lc: a padded block [P][P] (identity beyond the chain's dofs), xmem: a padded vector [P]; P is the same on every lane, so the guards
// are scalar branches around straight-line blocks
template <int CHD>
SG_HD void chain_solve_reg(const double* Lc, int P, double* xmem) {
  double x[CHD];
#pragma unroll
  for (int k = 0; k < CHD; k += 4)
    if (k < P) { x[k] = xmem[k]; x[k + 1] = xmem[k + 1]; x[k + 2] = xmem[k + 2]; x[k + 3] = xmem[k + 3]; }
#pragma unroll
  for (int k = CHD - 1; k >= 1; k--)
    if (k < P) {
      const double xk = x[k];
#pragma unroll
      for (int i = k - 1; i >= 0; i--) x[i] -= Lc[k * P + i] * xk;
    }
#pragma unroll
  for (int k = 0; k < CHD; k += 4)
    if (k < P) { x[k] /= Lc[k * P + k]; x[k + 1] /= Lc[(k + 1) * P + k + 1]; x[k + 2] /= Lc[(k + 2) * P + k + 2]; x[k + 3] /= Lc[(k + 3) * P + k + 3]; }
#pragma unroll
  for (int k = 1; k < CHD; k++)
    if (k < P) {
      double s = x[k];
#pragma unroll
      for (int i = k - 1; i >= 0; i--) s -= Lc[k * P + i] * x[i];
      x[k] = s;
    }
#pragma unroll
  for (int k = 0; k < CHD; k += 4)
    if (k < P) { xmem[k] = x[k]; xmem[k + 1] = x[k + 1]; xmem[k + 2] = x[k + 2]; xmem[k + 3] = x[k + 3]; }
}
// scalar row update with the reciprocal of the row's diagonal A + R precomputed (as the fast kernels' equality rows)
SG_HD double scalar_update_rcp(double& f, double b, double Ja, double R, double Adiag, double Ainv, bool inequality) {
  const double res = b + Ja + R * f, old = f;
  double fn = f - res * Ainv;
  if (inequality && fn < 0) fn = 0;
  const double d = fn - old;
  double change = 0.5 * d * d * Adiag + d * res;
  if (change > 1e-10) { fn = old; change = 0; }
  f = fn;
  return change;
}
// inverse of a symmetric positive definite 6 x 6 matrix (Gauss-Jordan without pivoting: the free object's Schur complement)
SG_HD_HEAVY void spd_inverse6(const double* Sm, double* Si) {
  double a[6][12];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) { a[i][j] = Sm[6 * i + j]; a[i][6 + j] = i == j ? 1.0 : 0.0; }
  for (int k = 0; k < 6; k++) {
    const double pv = 1.0 / a[k][k];
    for (int j = 0; j < 12; j++) a[k][j] *= pv;
    for (int i = 0; i < 6; i++) {
      if (i == k) continue;
      const double f = a[i][k];
      for (int j = 0; j < 12; j++) a[i][j] -= f * a[k][j];
    }
  }
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) Si[6 * i + j] = a[i][6 + j];
}
SG_HD double dot6(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5]; }
SG_HD void mat6vec(double* r, const double* M, const double* v) {
  for (int i = 0; i < 6; i++) r[i] = dot6(M + 6 * i, v);
}
#if defined(__HIPCC__)
#define SGT_NOINLINE __host__ __device__ __attribute__((noinline))
#else
#define SGT_NOINLINE __attribute__((noinline))
#endif
// pointers into the env's LDS block, typed as such for an out-of-line function: through generic pointers the loads are FLAT, whose
// completion the compiler can only wait for all at once -- which turns a prefetch into a stall
#if SGT_DEVICE
#define SGT_LDSP __attribute__((address_space(3)))
#define SGT_CONST __attribute__((address_space(4)))
#define SGT_GLOBP __attribute__((address_space(1)))
#else
#define SGT_LDSP
#define SGT_CONST
#define SGT_GLOBP
#endif
// the address space of the arrays whose home SGT_DIET decides (lds_carve): M^-1 (bit 6), the sliders' 1 / m without a free object (bit 7)
#if (SGT_DIET >> 6) & 1
#define SGT_MINV_AS SGT_GLOBP
#else
#define SGT_MINV_AS SGT_LDSP
#endif
#if (SGT_DIET >> 7) & 1
#define SGT_EINVM_AS SGT_GLOBP
#else
#define SGT_EINVM_AS SGT_LDSP
#endif
// The free object's joint-fix rows, one after the other (one lane).  A function of its own ON PURPOSE: inlined into the step kernel --
// 256 + 256 registers and spilling -- the loop's 40 live values went to scratch memory and a row cost 600 cycles; called, it gets a
// register allocation of its own.  The next row's 19 words are loaded before this row's dependent arithmetic.
static SGT_NOINLINE double free_fix_rows(const SGT_LDSP double* frow, const SGT_LDSP double* Be, const SGT_LDSP double* einvm, const SGT_LDSP double* Sinv,
                                         SGT_LDSP double* ffix, SGT_LDSP double* ae, SGT_LDSP double* af, int N) {
#if SGT_DEVICE && !defined(SGT_X_ROWS1LANE)
  // EIGHT LANES (r05; the caller enters with lanes 0 .. 7): lane q < 6 owns component q of the body's acceleration a_f, of B_e and of
  // C_e = -S^-1 B_e / D_e.  A row on one lane cost ~66 instructions -- 6 for B_e . a_f, 36 for C_e (recomputed per row since r04: the
  // array would not fit the LDS share of four workgroups per CU), 6 for a_f += C_e df -- and a wavefront alone on its SIMD pays ~7
  // cycles for each, whatever it is and however few lanes it feeds: 500 cycles a row, 40 % of a free-ball substep.  Here the dot product
  // is one multiply and an 8-lane DPP sum, C_e six multiply-adds per lane (row q of S^-1 in registers), the push one: ~40 instructions.
  // The scalar part of a row (residual, force, cost) runs on all eight lanes alike.  The next row's words are read one row ahead.
  const int q = (int)threadIdx.x & 7;
  const bool own = q < 6;
  const int qq = own ? q : 0;
  double afq = own ? af[qq] : 0.0, imp = 0;
  double Sq[6];
#pragma unroll
  for (int k = 0; k < 6; k++) Sq[k] = Sinv[6 * qq + k];
  double nr[4], nB[6], nBq = Be[qq], nf = ffix[0], na = ae[0], ni = einvm[0];
#pragma unroll
  for (int k = 0; k < 4; k++) nr[k] = frow[k];
#pragma unroll
  for (int k = 0; k < 6; k++) nB[k] = Be[k];
  for (int e = 0; e < N; e++) {
    double r4[4], B6[6];
    const double Bq = nBq, f = nf, ael = na, invm = ni;
#pragma unroll
    for (int k = 0; k < 4; k++) r4[k] = nr[k];
#pragma unroll
    for (int k = 0; k < 6; k++) B6[k] = nB[k];
    {  // (the row behind the last one is read too: the arrays are followed by other words of the LDS block, and the values are dropped)
      const int en = e + 1;
#pragma unroll
      for (int k = 0; k < 4; k++) nr[k] = frow[4 * en + k];
#pragma unroll
      for (int k = 0; k < 6; k++) nB[k] = Be[6 * en + k];
      nBq = Be[6 * en + qq]; nf = ffix[en]; na = ae[en]; ni = einvm[en];
    }
    double dot = own ? Bq * afq : 0.0;          // B_e . a_f over the six owner lanes (lanes 6, 7 add nothing)
    dot += dpp64<0xB1>(dot);                    // quad_perm [1,0,3,2]
    dot += dpp64<0x4E>(dot);                    // quad_perm [2,3,0,1]
    dot += dpp64<0x141>(dot);                   // row_half_mirror: the other quad of the eight
    double fn = f;
    imp -= scalar_update_rcp(fn, r4[0], ael - dot * invm, r4[1], r4[2], r4[3], false);
    const double dfl = fn - f;
    const double Cq = -(((Sq[0] * B6[0] + Sq[1] * B6[1]) + (Sq[2] * B6[2] + Sq[3] * B6[3])) + (Sq[4] * B6[4] + Sq[5] * B6[5])) * invm;
    afq += Cq * dfl;                            // (lanes 6, 7 carry a dummy: never stored)
    if (q == 0) { ffix[e] = fn; ae[e] = ael + invm * dfl; }
  }
  if (own) af[qq] = afq;
  return imp;
#else
  // C_e = -S^-1 B_e / D_e is RECOMPUTED per row (r04: 36 multiply-adds that do not depend on the previous row -- they run in the shadow
  // of its dependent chain) instead of read from a [N][6] array: without that array and the rows' copy of 1 / D the free ball's
  // LDS block is 37.8 KB instead of 50 -- four workgroups per CU instead of three.  Same expressions as the rows' build (tree_stage).
  double af6[6], imp = 0, Si[36];
  for (int q = 0; q < 6; q++) af6[q] = af[q];
  for (int q = 0; q < 36; q++) Si[q] = Sinv[q];
  double nr[4], nB[6], nC[6], nf = ffix[0], na = ae[0], ni = einvm[0];
  for (int q = 0; q < 4; q++) nr[q] = frow[q];
  for (int q = 0; q < 6; q++) nB[q] = Be[q];
  {
    double Bs[6];
    mat6vec(Bs, Si, nB);
    for (int q = 0; q < 6; q++) nC[q] = -Bs[q] * ni;
  }
#pragma unroll 2
  for (int e = 0; e < N; e++) {
    double r4[4], B6[6], C6[6], f = nf;
    const double ael = na, invm = ni;
    for (int q = 0; q < 4; q++) r4[q] = nr[q];
    for (int q = 0; q < 6; q++) { B6[q] = nB[q]; C6[q] = nC[q]; }
    {  // (on the device the row behind the last one is read too: the arrays are followed by other words of the LDS block, and the values
       //  are dropped; the host build reads the last row again -- its checking layout gives every array a heap block of its own)
      const int en = (SGT_DEVICE || e + 1 < N) ? e + 1 : e;
      for (int q = 0; q < 4; q++) nr[q] = frow[4 * en + q];
      for (int q = 0; q < 6; q++) nB[q] = Be[6 * en + q];
      nf = ffix[en]; na = ae[en]; ni = einvm[en];
      double Bs[6];
      mat6vec(Bs, Si, nB);
      for (int q = 0; q < 6; q++) nC[q] = -Bs[q] * ni;
    }
    const double old = f;
    imp -= scalar_update_rcp(f, r4[0], ael - dot6(B6, af6) * invm, r4[1], r4[2], r4[3], false);
    const double dfl = f - old;
    ffix[e] = f;
    ae[e] = ael + invm * dfl;
    for (int q = 0; q < 6; q++) af6[q] += C6[q] * dfl;
  }
  for (int q = 0; q < 6; q++) af[q] = af6[q];
  return imp;
#endif
}
#if SGT_DEVICE && !defined(SGT_X_ROWS8LANE) && !defined(SGT_X_ROWS1LANE)
#define SGT_FIXROWS_BLOCKED 1
// The free object's joint-fix rows IN BLOCKS (r05).  A row's update is affine in the body's acceleration a_f -- with u = B_e . a_f its force
// step is d = alpha_e + beta_e u (alpha_e = -(b + a_e + R f) / (A + R) from the row's own state, beta_e = (1 / D_e) / (A + R)) and
// a_f' = a_f + C_e d = (I + beta_e C_e B_e') a_f + alpha_e C_e -- and an equality row is never clamped or reverted (its step always lowers the
// cost: scalar_update_rcp's test cannot fire), so the sweep over the N rows is a chain of N affine maps of a 6-vector.  One after the other
// on eight lanes it was 500 cycles a row, 218 rows, 30 sweeps: 40 - 48 % of a free-ball substep (profiles/r05_tree_sections_freeball.txt).
// Here lane b < 32 owns a BLOCK of L consecutive rows (L = ceil(N / 32), made odd: the lanes' LDS addresses then fall into different banks):
//   1. every lane runs its block from a_f = 0 (-> c_b) and, beside it, the six unit vectors without the rows' alpha (-> M_b, 6 x 6):
//      the block as ONE affine map a_f -> M_b a_f + c_b.  M_b is constant over a substep's sweeps, but 36 values per lane have nowhere to
//      stay between two calls (the env's LDS block is full), so they are rebuilt: 78 instructions a row;
//   2. the scan: a_f at the start of block b + 1 = M_b (a_f at the start of block b) + c_b, block after block, the running a_f in scalar
//      registers (one matrix-vector product on every lane, lane b's result read back: ~57 instructions a block);
//   3. every lane runs its block again from its true start, now as the serial code does -- residual, force, cost, the slider's local part.
// ~3 300 instructions a sweep instead of 218 x 70.  Same mathematics as mj_solPGS's row-after-row sweep; the rounding differs (a block's
// successors see M_b a + c_b, not the sum its own rows accumulate: relative 1e-16 per block), as it already did between the oracle's serial
// dot product and the eight-lane tree sum.  Host builds (the emulation) keep the serial loop above.
// (INLINED into the sweep: as a called function -- 248 registers -- it saved 44 callee-saved registers to scratch memory on every call, 30
//  calls a substep: the free ball's fabric traffic went from 13.2 to 27.5 GB per sg_step call, profiles/r05_freeball_fix_hbm_traffic.json)
#if defined(SGT_X_BLOCKED_CALL)
#define SGT_BLOCKED_ATTR SGT_NOINLINE
#else
#define SGT_BLOCKED_ATTR __device__ __forceinline__
#endif
static SGT_BLOCKED_ATTR double free_fix_rows_blocked(const SGT_LDSP double* frow, const SGT_LDSP double* Be, const SGT_LDSP double* einvm, const SGT_LDSP double* Sinv,
                                                 SGT_LDSP double* ffix, SGT_LDSP double* ae, SGT_LDSP double* af, int N) {
  const int lane = (int)threadIdx.x;
  if (N <= 0) return 0.0;   // (a free body without sliders: no rows, a_f stays; uniform)
  const int L = ((N + 31) >> 5) | 1, nblk = (N + L - 1) / L;      // (uniform)
  const bool act = lane < nblk;
  const int e0 = act ? lane * L : 0;
  double Si[36];
#pragma unroll
  for (int k = 0; k < 36; k++) Si[k] = Sinv[k];
  struct Row { double b, R, A, I, B[6], C[6], f, al, im, z; int e; };
  auto load = [&](Row& w, int r) {
    const int e = e0 + r;
    const bool ok = act && e < N;
    const int ec = ok ? e : 0;
    w.e = ec; w.z = ok ? 1.0 : 0.0;
    w.b = frow[4 * ec]; w.R = frow[4 * ec + 1]; w.A = frow[4 * ec + 2]; w.I = frow[4 * ec + 3];
#pragma unroll
    for (int k = 0; k < 6; k++) w.B[k] = Be[6 * ec + k];
    w.f = ffix[ec]; w.al = ae[ec]; w.im = einvm[ec];
  };
  auto cvec = [&](Row& w) {   // C_e = -S^-1 B_e / D_e (same expressions as the rows' build and the serial loop)
#pragma unroll
    for (int q = 0; q < 6; q++)
      w.C[q] = -(((Si[6 * q] * w.B[0] + Si[6 * q + 1] * w.B[1]) + (Si[6 * q + 2] * w.B[2] + Si[6 * q + 3] * w.B[3])) + (Si[6 * q + 4] * w.B[4] + Si[6 * q + 5] * w.B[5])) * w.im;
  };
  // ---- 1. my block as an affine map
  double c[6] = {0, 0, 0, 0, 0, 0}, M[36];
#pragma unroll
  for (int k = 0; k < 36; k++) M[k] = (k % 7 == 0) ? 1.0 : 0.0;
  {
    Row w, wn;
    load(wn, 0);
    for (int r = 0; r < L; r++) {
      w = wn;
      load(wn, r + 1 < L ? r + 1 : r);
      cvec(w);
      const double beta = w.z * w.I * w.im, alpha = -(w.z * w.I) * ((w.b + w.al) + w.R * w.f);
      const double d = alpha + beta * dot6(w.B, c);
#pragma unroll
      for (int q = 0; q < 6; q++) c[q] += w.C[q] * d;
#pragma unroll
      for (int k = 0; k < 6; k++) {   // column k of M
        const double uk = beta * (((w.B[0] * M[k] + w.B[1] * M[6 + k]) + (w.B[2] * M[12 + k] + w.B[3] * M[18 + k])) + (w.B[4] * M[24 + k] + w.B[5] * M[30 + k]));
#pragma unroll
        for (int q = 0; q < 6; q++) M[6 * q + k] += w.C[q] * uk;
      }
    }
  }
  // ---- 2. the scan over the blocks: the running a_f is uniform (scalar registers), lane b keeps the value it had in front of block b
  // (lanes below b sit the step out: lane b's t is then final -- the a_f behind ITS block -- and block b + 1 starts from its left neighbour's t)
  double au[6], ain[6], t[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int q = 0; q < 6; q++) au[q] = af[q];
  for (int b = 0; b < nblk; b++) {
    if (lane >= b) {
#pragma unroll
      for (int q = 0; q < 6; q++)
        t[q] = ((c[q] + M[6 * q] * au[0]) + (M[6 * q + 1] * au[1] + M[6 * q + 2] * au[2])) + ((M[6 * q + 3] * au[3] + M[6 * q + 4] * au[4]) + M[6 * q + 5] * au[5]);
    }
#pragma unroll
    for (int q = 0; q < 6; q++) au[q] = readlane64(t[q], b);
  }
#pragma unroll
  for (int q = 0; q < 6; q++) {
    const double left = dpp64<0x138>(t[q]);   // wave_shr:1 -- lane l receives lane l - 1's word
    ain[q] = lane == 0 ? af[q] : left;
  }
  // ---- 3. my block's rows from their true start, as the serial sweep runs them
  double imp = 0;
  {
    Row w, wn;
    load(wn, 0);
    for (int r = 0; r < L; r++) {
      w = wn;
      load(wn, r + 1 < L ? r + 1 : r);
      cvec(w);
      const double Ja = w.al - dot6(w.B, ain) * w.im;
      const double res = w.b + Ja + w.R * w.f, fn = w.f - res * w.I, d = w.z * (fn - w.f);
      imp -= 0.5 * d * d * w.A + d * res;
      if (w.z != 0.0) { ffix[w.e] = fn; ae[w.e] = w.al + w.im * d; }
#pragma unroll
      for (int q = 0; q < 6; q++) ain[q] += w.C[q] * d;
    }
  }
  // the body's acceleration behind the last block
#pragma unroll
  for (int q = 0; q < 6; q++) {
    const double last = readlane64(ain[q], nblk - 1);
    if (lane == 0) af[q] = last;
  }
  return imp;
}
#endif
// The same with the composite's neighbour equalities: the equality BLOCKS [fix_e, e's neighbour rows (partner p: J = +1 on e, -1 on p)] in
// mj_solPGS's order.  A neighbour row moves two sliders and, through both, the body: a_f += (C_e - C_p) df.  (The neighbour rows' words
// sit in the work space: generic pointers.)
static SGT_NOINLINE double free_eq_blocks(const SGT_LDSP double* frow, const SGT_LDSP double* Be, const SGT_LDSP double* Ce, const SGT_LDSP double* einvm, SGT_LDSP double* ffix, SGT_LDSP double* ae,
                                          SGT_LDSP double* af, int N, const int* nbtab, double* nbf, const double* nbb, const double* nbR, const double* nbA,
                                          const double* nbI) {
  double af6[6], imp = 0;
  for (int q = 0; q < 6; q++) af6[q] = af[q];
  for (int e = 0; e < N; e++) {
    double B6[6], C6[6];
    for (int q = 0; q < 6; q++) { B6[q] = Be[6 * e + q]; C6[q] = Ce[6 * e + q]; }
    const double invm = einvm[e];
    double f = ffix[e], old = f, ael = ae[e];
    imp -= scalar_update_rcp(f, frow[4 * e], ael - dot6(B6, af6) * invm, frow[4 * e + 1], frow[4 * e + 2], frow[4 * e + 3], false);
    double dfl = f - old;
    ffix[e] = f;
    ael += invm * dfl;
    for (int q = 0; q < 6; q++) af6[q] += C6[q] * dfl;
    for (int d = 0; d < 3; d++) {
      const int k = d * N + e, pe = nbtab[k];
      if (pe < 0) continue;
      double Bp[6], Cp[6];
      for (int q = 0; q < 6; q++) { Bp[q] = Be[6 * pe + q]; Cp[q] = Ce[6 * pe + q]; }
      const double ipm = einvm[pe], apl = ae[pe];
      f = nbf[k]; old = f;
      imp -= scalar_update_rcp(f, nbb[k], (ael - dot6(B6, af6) * invm) - (apl - dot6(Bp, af6) * ipm), nbR[k], nbA[k], nbI[k], false);
      dfl = f - old;
      nbf[k] = f;
      ael += invm * dfl;
      ae[pe] = apl - ipm * dfl;
      for (int q = 0; q < 6; q++) af6[q] += (C6[q] - Cp[q]) * dfl;
    }
    ae[e] = ael;
  }
  for (int q = 0; q < 6; q++) af[q] = af6[q];
  return imp;
}
// in-place L'DL of a chain block (mj_factorM restricted to a serial chain)
SG_HD void chain_factor(double* Lc, int nd) {
  for (int k = nd - 1; k >= 1; k--) {
    const double dk = Lc[k * nd + k];
    for (int i = k - 1; i >= 0; i--) {
      const double a = Lc[k * nd + i] / dk;
      for (int j = i; j >= 0; j--) Lc[i * nd + j] -= a * Lc[k * nd + j];
      Lc[k * nd + i] = a;
    }
  }
}

// scalars the sweep takes from / hands back to the step (S.swc, in LDS: uniform reads)
enum { SWC_TEN_R = 0, SWC_TEN_B, SWC_TEN_F, SWC_TJ_A, SWC_TEN_I, SWC_CTEN, SWC_NCON = SWC_CTEN + 6, SWC_SERIAL, SWC_ITERS, SWC_N = 16 };

// ---------------------------------------------------------------- stage 10b: the PGS sweeps (mj_solPGS) of one forward pass, one env
// Everything the rows need was laid out by tree_env: the sliders' rows (S.ffix, S.flim, constants in the work space), the chains' limit
// rows (S.lrow), the contacts' J / W rows and scalars (work space), the accelerations a = M^-1 J' f of the current forces (S.aF, S.ae).
// A function of its own ON PURPOSE (see the call site).  Pointers come typed by address space, uniform values are made scalar again.
template <int CHD, bool FRT, bool NBT>   // FRT: the scene has a free object, NBT: the composite's neighbour rows -- compile-time facts of the instantiation, so that a sweep carries only its own scene class's code (r04: 355 -> 190 spill instructions for the four-finger gripper's)
static SGT_NOINLINE void tree_sweep(const SGT_CONST SgPlanHeader* Hp, const SGT_CONST SgTreeDev* Tp, const SGT_CONST int* nbtab, const SGT_CONST SgEqSlot* sched,
                                    const int* nbtab_generic, SGT_GLOBP double* cw_, SGT_LDSP double* lds_, unsigned long long* secprof) {
#if SGT_DEVICE
  // (arguments of a called function arrive in vector registers: back to scalar ones, so that the plan tables are scalar loads again)
  auto uni = [](auto* q) { return (decltype(q))(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)q >> 32)) << 32) |
                                                 (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned long long)q)); };
  Hp = uni(Hp); Tp = uni(Tp); nbtab = uni(nbtab); sched = uni(sched); nbtab_generic = uni(nbtab_generic); secprof = uni(secprof);
  cw_ = (SGT_GLOBP double*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)cw_ >> 32)) << 32) |
                            (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned long long)cw_));
  lds_ = (SGT_LDSP double*)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)lds_);
#endif
  const SGT_CONST SgPlanHeader& H = *Hp;
  const SGT_CONST SgTreeDev& T = *Tp;
  double* const cw = (double*)cw_;
  const int N = H.nelem, K = T.K;
  constexpr int CS = CHD;   // (= T.CS: the plan pads the chains' stride to the instantiation's capacity)
  constexpr bool FR = FRT, NB = NBT;
  Lds S;
  lds_carve(S, (double*)lds_, T, N, H.has_free, cw + (size_t)SGT_MAXHIT * SGT_HITREC * SGT_RECW + (size_t)SGT_MAXCON * cws_row_doubles(T.CS) + T.NMAT, nullptr, H.nnb);
  const long long CW = cws_row_doubles(CS);
#if !SGT_DEVICE && defined(SGT_EMU_SEPARATE)
  double* const crow0 = sep_pool() ? sep_part(1, (size_t)SGT_MAXCON * CW) : cw + (size_t)SGT_MAXHIT * SGT_HITREC * SGT_RECW;
#else
  double* const crow0 = cw + (size_t)SGT_MAXHIT * SGT_HITREC * SGT_RECW;
#endif
  auto crow = [&](int c) { return crow0 + (size_t)c * CW; };
  auto cscr = [&](int c) -> const double* { return c < S.ncache ? S.csc + (size_t)c * SGT_CSC : crow0 + (size_t)c * CW + 12 * CS; };
  // The sweeps' arrays once more, TYPED BY ADDRESS SPACE (r05).  Through the carve's generic pointers every access is a FLAT instruction,
  // which counts on both memory counters: the wait for an LDS word (a contact's force, a slider's acceleration) then also waits for
  // every global load in flight -- the NEXT contact's record, requested one update ahead precisely so that its latency is hidden.  The
  // r04 ISA had `flat_load_dwordx4 (S.cf)` + `s_waitcnt vmcnt(0)` in the middle of every update: ~8 k cycles an update, two exposed
  // round trips.  Typed, the LDS words are ds_read / ds_write (lgkmcnt only) and the prefetch stays in flight.
  SGT_LDSP double* const aeL = (SGT_LDSP double*)S.ae;
  SGT_LDSP double* const aFL = (SGT_LDSP double*)S.aF;
  SGT_LDSP double* const cfL = (SGT_LDSP double*)S.cf;
  SGT_LDSP double* const ffixL = (SGT_LDSP double*)S.ffix;
  SGT_LDSP double* const flimL = (SGT_LDSP double*)S.flim;
  SGT_LDSP double* const lrowL = (SGT_LDSP double*)S.lrow;
  SGT_LDSP double* const ofL = (SGT_LDSP double*)S.of;
  SGT_LDSP double* const BeL = (SGT_LDSP double*)S.Be;
  const SGT_LDSP int* const icntL = (const SGT_LDSP int*)S.icnt;
  const SGT_LDSP int* const hitpairL = (const SGT_LDSP int*)S.hit_pair;
  const SGT_LDSP int* const hitcntL = (const SGT_LDSP int*)S.hit_cnt;
  const SGT_MINV_AS double* const MinvT = (const SGT_MINV_AS double*)S.Minv;
  const SGT_GLOBP double* const bfixG = (const SGT_GLOBP double*)S.bfix;
  const SGT_GLOBP double* const RfixG = (const SGT_GLOBP double*)S.Rfix;
  const SGT_GLOBP double* const IfixG = (const SGT_GLOBP double*)S.Ifix;
  const SGT_GLOBP double* const nbbG = (const SGT_GLOBP double*)S.nbb;
  const SGT_GLOBP double* const nbRG = (const SGT_GLOBP double*)S.nbR;
  const SGT_GLOBP double* const nbIG = (const SGT_GLOBP double*)S.nbI;
  SGT_GLOBP double* const nbfG = (SGT_GLOBP double*)S.nbf;
  [[maybe_unused]] const SGT_GLOBP double* const nbqG = (const SGT_GLOBP double*)S.nbq;
  [[maybe_unused]] const SGT_GLOBP double* const fixqG = (const SGT_GLOBP double*)S.fixq;
  const SGT_EINVM_AS double* const einvmNF = (const SGT_EINVM_AS double*)S.einvm;   // scenes WITHOUT a free object only (with one: LDS, lds_carve)
  (void)aeL; (void)aFL; (void)cfL; (void)ffixL; (void)flimL; (void)lrowL; (void)ofL; (void)BeL; (void)icntL; (void)hitpairL; (void)hitcntL; (void)MinvT;
  (void)bfixG; (void)RfixG; (void)IfixG; (void)nbbG; (void)nbRG; (void)nbIG; (void)nbfG; (void)einvmNF;
  struct { unsigned long long* secprof; const int* nbtab; } A = {secprof, nbtab_generic};   // (what SGT_STAMP and the free object's row functions name)
  (void)A;
  SGT_STAMP_INIT();
  const double con_mu[2] = {H.con_mu[0], H.con_mu[1]};
  const int ncon = (int)S.swc[SWC_NCON];
  const bool serial_contacts = S.swc[SWC_SERIAL] != 0.0;
  const double ten_R = S.swc[SWC_TEN_R], ten_b = S.swc[SWC_TEN_B], tj_A = S.swc[SWC_TJ_A], ten_I = S.swc[SWC_TEN_I];
  double ten_f = S.swc[SWC_TEN_F];
  const double cten[6] = {S.swc[SWC_CTEN], S.swc[SWC_CTEN + 1], S.swc[SWC_CTEN + 2], S.swc[SWC_CTEN + 3], S.swc[SWC_CTEN + 4], S.swc[SWC_CTEN + 5]};
  auto slider_acc = [&](int e) {   // a slider's constraint acceleration: with a free object its local part minus the body's share
    return FR ? S.ae[e] - dot6(S.Be + 6 * e, S.of + OF_AF) * S.einvm[e] : S.ae[e];
  };
  int iters = 0;
  // the sliders' row constants, per lane for the whole solve (r04): b, R, 1 / (A + R) of the joint-fix row, R, b, 1 / (A + R) of the two
  // limit rows, 1 / m and the tendon coefficient.  They sit in the work space (not in LDS); every sweep used to fetch
  // them again -- four dependent trips of the wavefront to L2 per pass, 6 % + 3 % of a substep at the squeeze
  double kfb[SGT_NSLOT], kfR[SGT_NSLOT], kfI[SGT_NSLOT], kim[SGT_NSLOT], kco[SGT_NSLOT], klR[SGT_NSLOT][2], klb[SGT_NSLOT][2], klI[SGT_NSLOT][2];
  SGT_PAR_SLOT(e, t, N) {
    kfb[t] = S.bfix[e]; kfR[t] = S.Rfix[e]; kfI[t] = S.Ifix[e]; kim[t] = S.einvm[e]; kco[t] = S.ecoef[e];
    for (int sd = 0; sd < 2; sd++) { klR[t][sd] = S.Rlim[2 * e + sd]; klb[t][sd] = S.blim[2 * e + sd]; klI[t][sd] = S.Ilim[2 * e + sd]; }
  }
  for (int it = 0; it < H.iterations; it++) {
    double imp_par = 0, imp_uni = 0;
    // joint-fix rows: each on its own slider
    double S_ae = 0;
    if (FR) {
      // with a free object a joint-fix row moves the body and through it every slider: the rows run one after the other (mj_solPGS's
      // order), the body's acceleration a_f in registers, a row's own slider from its local part and a_f
      if (NB) SGT_ONE {
        S.red[0] = free_eq_blocks((const SGT_LDSP double*)S.frow, (const SGT_LDSP double*)S.Be, (const SGT_LDSP double*)S.Ce, (const SGT_LDSP double*)S.einvm, (SGT_LDSP double*)S.ffix,
                                  (SGT_LDSP double*)S.ae, (SGT_LDSP double*)(S.of + OF_AF), N, A.nbtab, S.nbf, S.nbb, S.nbR, S.nbA, S.nbI);
      }
#ifdef SGT_FIXROWS_BLOCKED
      if (!NB) {   // (device: every lane a block of the rows, free_fix_rows_blocked; each lane's share of the cost change goes into the wavefront sum)
        imp_par += free_fix_rows_blocked((const SGT_LDSP double*)S.frow, (const SGT_LDSP double*)S.Be, (const SGT_LDSP double*)S.einvm, (const SGT_LDSP double*)(S.of + OF_SINV),
                                         (SGT_LDSP double*)S.ffix, (SGT_LDSP double*)S.ae, (SGT_LDSP double*)(S.of + OF_AF), N);
        SGT_ONE { S.red[0] = 0.0; }
      }
#else
      if (!NB) SGT_ROW_LANES {   // (device: lanes 0 .. 7, the six components of the body's acceleration a lane each)
        const double r_ = free_fix_rows((const SGT_LDSP double*)S.frow, (const SGT_LDSP double*)S.Be, (const SGT_LDSP double*)S.einvm, (const SGT_LDSP double*)(S.of + OF_SINV),
                                        (SGT_LDSP double*)S.ffix, (SGT_LDSP double*)S.ae, (SGT_LDSP double*)(S.of + OF_AF), N);
        SGT_ONE { S.red[0] = r_; }
      }
#endif
      SGT_SYNC();
      imp_uni += S.red[0];
      SGT_PAR(e, N) S_ae += S.ecoef[e] * S.ae[e];
      S_ae = wsum(S_ae) - dot6(H.obj_tenB, S.of + OF_AF);   // sum coef_e a_e, a_e = local part - B_e . a_f / D_e
    } else if (NB) {
      // equality BLOCKS [fix_e, e's neighbour rows] in the plan's list schedule: the blocks of a round share no slider (they
      // commute exactly), every block sits in a later round than the blocks it depends on -- the rounds in order ARE mj_solPGS's
      // sequential sweep (sg_plan.h); 64 blocks per round, a lane each
#if SGT_DEVICE && !defined(SGT_X_EQSYNC)
      // (r05s) The rounds PIPELINED: a round's table words (its slot) and its rows' constants and forces -- two dependent trips to the work
      // space -- do not depend on the rounds before it, only the sliders' accelerations (LDS) do.  As written for the emulation below, every
      // round paid both trips and then a barrier that drains the rows' force stores: ~2.5 us a round, 55 us a sweep, three quarters of a
      // substep of the four-finger gripper's default model.  Here a round's constants are requested two rounds ahead and its slot five (three
      // register sets), and nothing in the loop waits at a barrier: the wavefront's LDS instructions execute in order, so a lane's read of a
      // slider sees the write another lane made a round earlier.  Same rows, same order, same arithmetic: same bits.
      {
        const int lane = (int)threadIdx.x;
        const int nr = H.eq_rounds;
        struct ERec { int e, pe[3]; double invm, Rf, bf, If, ipm[3], R[3], b[3], I[3], f[3]; };
        auto load_rec = [&](ERec& q, const SgEqSlot slot) {
          q.e = slot.e;
          const int e = slot.e < N ? slot.e : 0;
          // (the rows' constants packed by the build stage, S.fixq / S.nbq: one 32-byte record per row -- two loads behind ONE address
          //  instead of four or five words from as many arrays behind as many 64-bit address computations)
          {
            const SGT_GLOBP double2* const fq = (const SGT_GLOBP double2*)(fixqG + 4 * e);
            const double2 u = fq[0], w = fq[1];
            q.bf = u.x; q.Rf = u.y; q.If = w.x; q.invm = w.y;
          }
#pragma unroll
          for (int d = 0; d < 3; d++) {
            const int pe = slot.e < N ? slot.p[d] : N;
            const int k = d * N + e;
            q.pe[d] = pe;
            const SGT_GLOBP double2* const nq = (const SGT_GLOBP double2*)(nbqG + 4 * k);   // (row k's words exist whether the block has the row or not)
            const double2 u = nq[0], w = nq[1];
            q.R[d] = u.x; q.b[d] = u.y; q.I[d] = w.x; q.ipm[d] = w.y;
            q.f[d] = nbfG[k];
          }
        };
        // A block straight through: the slider's and its partners' accelerations are read TOGETHER at the top (one LDS latency, not four in
        // a row behind each other's stores), the four rows run in registers, the stores follow.  A row the block does not have is a no-op
        // by its record (R = b = 1 / (A + R) = 1 / m_p = f = 0: the step is exactly 0) on the block's own slider as stand-in partner, so no
        // lane branches inside a block; the own slider's store comes last.
        auto run = [&](const ERec& q) {
          const int e = q.e;
          if (e < N) {
            const double invm = q.invm;
            const int pc0 = q.pe[0] < N ? q.pe[0] : e, pc1 = q.pe[1] < N ? q.pe[1] : e, pc2 = q.pe[2] < N ? q.pe[2] : e;
            double ae_ = aeL[e], f0 = ffixL[e];
            const double ap0 = aeL[pc0], ap1 = aeL[pc1], ap2 = aeL[pc2];
            double old = f0;
            imp_par -= scalar_update_rcp(f0, q.bf, ae_, q.Rf, invm + q.Rf, q.If, false);
            ae_ += invm * (f0 - old);
            double f1 = q.f[0];
            old = f1;
            imp_par -= scalar_update_rcp(f1, q.b[0], ae_ - ap0, q.R[0], invm + q.ipm[0] + q.R[0], q.I[0], false);
            const double d1 = f1 - old;
            ae_ += invm * d1;
            double f2 = q.f[1];
            old = f2;
            imp_par -= scalar_update_rcp(f2, q.b[1], ae_ - ap1, q.R[1], invm + q.ipm[1] + q.R[1], q.I[1], false);
            const double d2 = f2 - old;
            ae_ += invm * d2;
            double f3 = q.f[2];
            old = f3;
            imp_par -= scalar_update_rcp(f3, q.b[2], ae_ - ap2, q.R[2], invm + q.ipm[2] + q.R[2], q.I[2], false);
            const double d3 = f3 - old;
            ae_ += invm * d3;
            ffixL[e] = f0;
            nbfG[e] = f1; nbfG[N + e] = f2; nbfG[2 * N + e] = f3;
            aeL[pc0] = ap0 - q.ipm[0] * d1;
            aeL[pc1] = ap1 - q.ipm[1] * d2;
            aeL[pc2] = ap2 - q.ipm[2] * d3;
            aeL[e] = ae_;
          }
        };
        if (nr > 0) {
          // three register sets: a round's record is requested TWO rounds before it runs, its slot three rounds before that
          auto slot_of = [&](int r) { return sched[(r < nr ? r : 0) * 64 + lane]; };   // (past the end: round 0's words, read and not used)
          ERec q0, q1, q2;
          SgEqSlot t0 = slot_of(0), t1 = slot_of(1), t2 = slot_of(2);
          load_rec(q0, t0); load_rec(q1, t1);
          t0 = slot_of(3); t1 = slot_of(4);
          for (int r = 0; r < nr; r += 3) {
            load_rec(q2, t2); t2 = slot_of(r + 5);
            run(q0);
            __builtin_amdgcn_wave_barrier();
            load_rec(q0, t0); t0 = slot_of(r + 6);
            if (r + 1 < nr) run(q1);
            __builtin_amdgcn_wave_barrier();
            load_rec(q1, t1); t1 = slot_of(r + 7);
            if (r + 2 < nr) run(q2);
            __builtin_amdgcn_wave_barrier();
          }
        }
        SGT_SYNC();
      }
#else
      for (int r = 0; r < H.eq_rounds; r++) {
        SGT_PAR(sl, 64) {
          const SgEqSlot slot = sched[r * 64 + sl];
          const int e = slot.e;
          if (e < N) {
            const double invm = einvmNF[e];
            double ae_ = aeL[e], f = ffixL[e];
            double old = f;
            const double Rf = RfixG[e];
            imp_par -= scalar_update_rcp(f, bfixG[e], ae_, Rf, invm + Rf, IfixG[e], false);
            ffixL[e] = f;
            ae_ += invm * (f - old);
            for (int d = 0; d < 3; d++) {
              const int pe = slot.p[d];
              if (pe >= N) continue;
              const int k = d * N + e;
              const double ap = aeL[pe], ipm = einvmNF[pe], R = nbRG[k];
              f = nbfG[k]; old = f;
              imp_par -= scalar_update_rcp(f, nbbG[k], ae_ - ap, R, invm + ipm + R, nbIG[k], false);
              nbfG[k] = f;
              ae_ += invm * (f - old);
              aeL[pe] = ap - ipm * (f - old);
            }
            aeL[e] = ae_;
          }
        }
        SGT_SYNC();
      }
#endif
      SGT_PAR_SLOT(e, t, N) S_ae += kco[t] * aeL[e];
      S_ae = wsum(S_ae);
    } else {
      SGT_PAR_SLOT(e, t, N) {
        const double invm = kim[t];
        double f = ffixL[e];
        const double old = f, ael = aeL[e];
        imp_par -= scalar_update_rcp(f, kfb[t], ael, kfR[t], invm + kfR[t], kfI[t], false);
        ffixL[e] = f;
        const double an = ael + invm * (f - old);
        aeL[e] = an;
        S_ae += kco[t] * an;
      }
      S_ae = wsum(S_ae);
    }
    SGT_STAMP(17);
    {  // the tendon-fix row over all sliders
      const double old = ten_f;
      imp_uni -= scalar_update_rcp(ten_f, ten_b, S_ae, ten_R, tj_A + ten_R, ten_I, false);
      const double dfl = ten_f - old;
      SGT_PAR_SLOT(e, t, N) aeL[e] += kco[t] * dfl * kim[t];
      if (FR) {
        SGT_SYNC();
        SGT_ONE { for (int q = 0; q < 6; q++) S.of[OF_AF + q] += cten[q] * dfl; }
        SGT_SYNC();
      }
    }
    SGT_STAMP(18);
    // chain limit rows: serial within a chain, the chains side by side
#if SGT_DEVICE && !defined(SGT_X_NOLG)
    // A LANE GROUP per chain (r04): the 16 lanes of a DPP row hold the chain's accelerations -- lane l dofs l and l + 16 -- in registers
    // for the whole pass; a row's J a = +-a[dof] is a masked row sum (rotations, no LDS), its scalar update runs on all 16 lanes
    // alike, its push a += M^-1[dof][.] df is one multiply-add per lane and word.  (One lane per chain -- 4 of 64 -- read and wrote
    // all CS words through LDS per row: 1 900 cycles a row, 13 % of a substep at the squeeze.)  Four chains per pass.
    {
      const int grp = (int)threadIdx.x >> 4, l = (int)threadIdx.x & 15;
      const bool lo_w = l < CS, hi_w = l + 16 < CS;   // (short chains: CS < 16 -- the lanes beyond the stride hold no word)
      const int ll = lo_w ? l : 0;
      for (int c0 = 0; c0 < K; c0 += 4) {
        const int c = c0 + grp, cc = c < K ? c : 0;
        SGT_LDSP double* rows = lrowL + SGT_LROW * 2 * T.c_dof0[cc];
        const SGT_MINV_AS double* Mi = MinvT + cc * CS * CS;
        double a0 = lo_w ? aFL[cc * CS + l] : 0.0, a1 = hi_w ? aFL[cc * CS + l + 16] : 0.0;
        const int nrow = c < K ? icntL[IC_NLIM0 + cc] : 0;
        int nmax = __builtin_amdgcn_readlane(nrow, 0);
        for (int g2 = 16; g2 < 64; g2 += 16) { const int o = __builtin_amdgcn_readlane(nrow, g2); nmax = o > nmax ? o : nmax; }
#if defined(SG_SECTION_PROF)
        if (threadIdx.x == 0) { atomicAdd(&A.secprof[44], (unsigned long long)nmax); atomicAdd(&A.secprof[45], 1ull); }   // chain limit rows: row slots per pass
#endif
        // (two register sets: the next row's record and its row of M^-1 -- two dependent LDS round trips -- are on their way during a row's update)
        struct LRec { double sg, R, b, f, Ainv, mdd, m0, m1; int dl; };
        auto load_row = [&](LRec& q, int i) {
          // (a group past its own list -- or without one -- reads row 0's words and dof 0's row of M^-1: both exist, nothing is applied.
          //  The dof index MUST be a valid one: M^-1 sits in the work space, and a stale LDS word as an index into it is a memory fault)
          const bool have = i < nrow;
          const SGT_LDSP double* r = rows + SGT_LROW * (have ? i : 0);
          int dl = have ? (int)r[0] : 0;
          dl = dl < 0 ? 0 : (dl >= CS ? CS - 1 : dl);
          q.dl = dl; q.sg = r[1]; q.R = r[2]; q.b = r[3]; q.f = r[4]; q.Ainv = r[5];
          q.mdd = Mi[dl * CS + dl]; q.m0 = Mi[dl * CS + ll]; q.m1 = Mi[dl * CS + (hi_w ? l + 16 : ll)];
        };
        auto update_row = [&](const LRec& q, int i) {
          const bool act = i < nrow;
          double f = q.f;
          const double adl = rowsum16(l == (q.dl & 15) ? (q.dl < 16 ? a0 : a1) : 0.0);
          const double ch = scalar_update_rcp(f, q.b, q.sg * adl, q.R, q.mdd + q.R, q.Ainv, true);
          const double dfl = act ? q.sg * (f - q.f) : 0.0;
          if (lo_w) a0 += q.m0 * dfl;
          if (hi_w) a1 += q.m1 * dfl;
          if (act && l == 0) { imp_par -= ch; rows[SGT_LROW * i + 4] = f; }
        };
        // SGT_LROW_AHEAD register sets: a row's words and its row of M^-1 -- two dependent round trips away: the row's dof index from LDS, then
        // the work space -- are requested SGT_LROW_AHEAD - 1 rows ahead.  Seven ahead instead of three measured SLOWER (r05): the wait is
        // not the loads' latency
        constexpr int AH = SGT_LROW_AHEAD;
        LRec q[AH];
        if (nmax > 0) {
#pragma unroll
          for (int k = 0; k < AH - 1; k++) load_row(q[k], k);
        }
        for (int i = 0; i < nmax; i += AH) {
#pragma unroll
          for (int k = 0; k < AH; k++) {
            load_row(q[(k + AH - 1) % AH], i + k + AH - 1);
            update_row(q[k], i + k);
          }
        }
        if (c < K) {
          if (lo_w) aFL[cc * CS + l] = a0;
          if (hi_w) aFL[cc * CS + l + 16] = a1;
        }
      }
    }
#else
    SGT_PAR(c, K) {
      double* rows = S.lrow + SGT_LROW * 2 * T.c_dof0[c];
      const double* Mi = S.Minv + c * CS * CS;
      double* aFc = S.aF + c * CS;
      const int nrow = S.icnt[IC_NLIM0 + c];
      for (int i = 0; i < nrow; i++) {
        double* r = rows + SGT_LROW * i;
        const int dl = (int)r[0];
        double f = r[4];
        const double old = f;
        imp_par -= scalar_update_rcp(f, r[3], r[1] * aFc[dl], r[2], Mi[dl * CS + dl] + r[2], r[5], true);
        r[4] = f;
        const double dfl = r[1] * (f - old);
        // every load before the first store (a load-store chain through LDS costs a round trip per element): unrolled over the
        // capacity, the loads unguarded (beyond the padded stride CS they hit other LDS words and are dropped), the stores behind
        // scalar branches on CS, which is the same on every lane
        double an[CHD];
#pragma unroll
        for (int k = 0; k < CHD; k++) an[k] = aFc[k] + Mi[dl * CS + k] * dfl;
#pragma unroll
        for (int k = 0; k < CHD; k += 4)
          if (k < CS) { aFc[k] = an[k]; aFc[k + 1] = an[k + 1]; aFc[k + 2] = an[k + 2]; aFc[k + 3] = an[k + 3]; }
      }
    }
#endif
    SGT_STAMP(19);
    // slider limit rows
    SGT_PAR_SLOT(e, t, N) {
      const double invm = kim[t];
#pragma unroll
      for (int sd = 0; sd < 2; sd++) {
        const double R = klR[t][sd];
        if (R == 0.0) continue;
        const double sg = sd ? -1.0 : 1.0;
        double f = flimL[2 * e + sd];
        const double old = f;
        imp_par -= scalar_update_rcp(f, klb[t][sd], sg * aeL[e], R, invm + R, klI[t][sd], true);
        flimL[2 * e + sd] = f;
        aeL[e] += invm * sg * (f - old);
      }
    }
    SGT_SYNC();
    SGT_STAMP(12);
    // contacts: one stream per chain ...
    if (!serial_contacts) {
#if SGT_DEVICE && !defined(SGT_X_NOSTREAM)
      // One STREAM PER CHAIN on a lane group (r04): the chain's accelerations in registers as in the limit-row pass (lane l: dofs l,
      // l + 16), a contact's J and W rows read one word per lane and row (coalesced 128-byte pieces from the work space, the NEXT
      // contact's on their way during this one's update), J a as three row sums, the 3 x 3 block update on all 16 lanes alike, the
      // push a += W' df as three multiply-adds per lane and word.  The streams' contact lists (S.hit_pair: contact ids by chain,
      // offsets behind them) are built with the rows.  One lane per chain cost ~14 k cycles an update: 120 loads and the whole
      // chain vector through LDS per contact, 47 % of a substep at the squeeze.
      {
        const int grp = (int)threadIdx.x >> 4, l = (int)threadIdx.x & 15;
        const bool lo_w = l < CS, hi_w = l + 16 < CS;   // (short chains: CS < 16 -- the lanes beyond the stride hold no word)
        const int ll = lo_w ? l : 0;
        const SGT_LDSP int* const lvl = hitpairL;       // [nlev][K]: the contact of chain c in level L, or -1 (tree_stage 3)
        const int nlev = icntL[IC_NLEV], nb = (K + 3) >> 2, nslot = nlev * nb;   // a slot = (level, batch of four chains): one update per lane group
        // a contact as the sweep needs it: J and W rows, word l (j, w) and word l + 16 (k, x), and the scalars of its record -- all from the
        // work space (one address space: the loads of the NEXT contact, requested before this one's update, stay in flight across it;
        // through a pointer that may be LDS or global every use waited for every load issued before it)
        struct CRec { double j0, j1, j2, k0, k1, k2, w0, w1, w2, x0, x1, x2, A[6], Pe[7], B[3], R, invm, Js[3], slf; int ci; };
        auto load_rec = [&](CRec& q, int ci) {
          const double* J = crow(ci);
          const double* W = J + 3 * CS;
          const double* sc = J + 12 * CS;
          const int lh = hi_w ? l + 16 : ll;   // (a lane without a word reads word 0 / word ll again: its product is zeroed below)
          q.ci = ci;
          q.j0 = J[ll]; q.j1 = J[CS + ll]; q.j2 = J[2 * CS + ll]; q.k0 = J[lh]; q.k1 = J[CS + lh]; q.k2 = J[2 * CS + lh];
          q.w0 = W[ll]; q.w1 = W[CS + ll]; q.w2 = W[2 * CS + ll]; q.x0 = W[lh]; q.x1 = W[CS + lh]; q.x2 = W[2 * CS + lh];
#pragma unroll
          for (int k = 0; k < 6; k++) q.A[k] = sc[CS_A + k];
#pragma unroll
          for (int k = 0; k < 3; k++) { q.B[k] = sc[CS_B + k]; q.Js[k] = sc[CS_JS + k]; }
          q.R = sc[CS_R]; q.invm = sc[CS_INVM]; q.slf = sc[CS_SL];
#pragma unroll
          for (int k = 0; k < 7; k++) q.Pe[k] = sc[CS_PE + k];
        };
        if (nslot > 0) {
#if defined(SG_SECTION_PROF)
          if (threadIdx.x == 0) { atomicAdd(&A.secprof[40], (unsigned long long)nslot); atomicAdd(&A.secprof[41], 1ull); }   // update slots per pass
#endif
          const bool one_batch = nb == 1;   // (K <= 4: a group keeps ITS chain's accelerations in registers over the whole pass)
          int ci_safe = 0;   // (level 0 holds a contact: what a group without one in a slot reads; nothing of it is applied)
          for (int c = K - 1; c >= 0; c--) { const int x = lvl[c]; ci_safe = x >= 0 ? x : ci_safe; }
          // (K <= 4, every reference scene: slot = level, the group's chain is fixed -- no divisions by the batch count in the loop)
          auto chain_of = [&](int sl_) { return one_batch ? grp : 4 * (sl_ % nb) + grp; };
          auto contact_of = [&](int sl_) {
            if (one_batch) return (sl_ < nslot && grp < K) ? lvl[sl_ * K + grp] : -1;
            const int c = chain_of(sl_);
            return (sl_ < nslot && c < K) ? lvl[(sl_ / nb) * K + c] : -1;
          };
          int cc = grp < K ? grp : 0;
          double a0 = lo_w ? aFL[cc * CS + l] : 0.0, a1 = hi_w ? aFL[cc * CS + l + 16] : 0.0;   // (0 on a lane without a word: its J a terms vanish)
#if defined(SG_SECTION_PROF)
          long long tpa = 0, tpb = 0, tpc = 0, tpn = 0;   // cycles of an update's three parts (registers; added up once per pass, below)
#define SGT_TP(x) const long long x = clock64()
#else
#define SGT_TP(x) ((void)0)
#endif
          auto update = [&](const CRec& q, const bool act) {
            SGT_TP(t0_);
            const int ci = q.ci, sl = (int)q.slf;
            const double p0 = rowsum16(q.j0 * a0 + q.k0 * a1), p1 = rowsum16(q.j1 * a0 + q.k1 * a1), p2 = rowsum16(q.j2 * a0 + q.k2 * a1);
            const double as_ = sl >= 0 ? aeL[sl] : 0.0;
            double f[3] = {cfL[3 * ci], cfL[3 * ci + 1], cfL[3 * ci + 2]}, df[3];
            const double res[3] = {q.B[0] + q.Js[0] * as_ + p0 + q.R * f[0], q.B[1] + q.Js[1] * as_ + p1 + q.R * f[1], q.B[2] + q.Js[2] * as_ + p2 + q.R * f[2]};
#if defined(SG_SECTION_PROF)
            asm volatile("" :: "v"(res[0]), "v"(res[1]), "v"(res[2]));
#endif
            SGT_TP(t1_);
            const double ch = contact_block_update_pre(q.A, q.Pe, res, f, con_mu, df);
#if defined(SG_SECTION_PROF)
            asm volatile("" :: "v"(df[0]), "v"(df[1]), "v"(df[2]), "v"(ch));
#endif
            SGT_TP(t2_);
            if (act) {
              if (lo_w) a0 += q.w0 * df[0] + q.w1 * df[1] + q.w2 * df[2];
              if (hi_w) a1 += q.x0 * df[0] + q.x1 * df[1] + q.x2 * df[2];
              if (l == 0) {
                imp_par -= ch;
                cfL[3 * ci] = f[0]; cfL[3 * ci + 1] = f[1]; cfL[3 * ci + 2] = f[2];
                if (sl >= 0) aeL[sl] += q.invm * (q.Js[0] * df[0] + q.Js[1] * df[1] + q.Js[2] * df[2]);
              }
            }
#if defined(SG_SECTION_PROF)
            asm volatile("" :: "v"(a0), "v"(a1));
            { const long long t3_ = clock64(); tpa += t1_ - t0_; tpb += t2_ - t1_; tpc += t3_ - t2_; tpn++; }
#endif
          };
          // one slot: more than four chains -> the group's chain changes from slot to slot, its accelerations go through LDS
          auto slot = [&](const CRec& q, int sl_, int ci) {
            if (!one_batch) {
              const int c = chain_of(sl_);
              cc = c < K ? c : 0;
              a0 = lo_w ? aFL[cc * CS + l] : 0.0; a1 = hi_w ? aFL[cc * CS + l + 16] : 0.0;
            }
            update(q, ci >= 0);
            if (!one_batch && ci >= 0) {
              if (lo_w) aFL[cc * CS + l] = a0;
              if (hi_w) aFL[cc * CS + l + 16] = a1;
            }
          };
          CRec ra, rb;   // two register sets: no copies, the other set's loads in flight during an update
          int cia = contact_of(0), cib;
          load_rec(ra, cia >= 0 ? cia : ci_safe);
          for (int j = 0; j < nslot; j += 2) {
            cib = contact_of(j + 1);
            load_rec(rb, cib >= 0 ? cib : ci_safe);
            slot(ra, j, cia);
            cia = contact_of(j + 2);
            load_rec(ra, cia >= 0 ? cia : ci_safe);
            slot(rb, j + 1, cib);
          }
          if (one_batch && grp < K) {
            if (lo_w) aFL[cc * CS + l] = a0;
            if (hi_w) aFL[cc * CS + l + 16] = a1;
          }
#if defined(SG_SECTION_PROF)
          if (threadIdx.x == 0) { atomicAdd(&A.secprof[42], (unsigned long long)tpa); atomicAdd(&A.secprof[43], (unsigned long long)tpn); atomicAdd(&A.secprof[46], (unsigned long long)tpb); atomicAdd(&A.secprof[47], (unsigned long long)tpc); }
#endif
#undef SGT_TP
        }
      }
#else
      for (int Lv = 0; Lv < S.icnt[IC_NLEV]; Lv++) {   // the levels in sequence, a level's contacts (one per chain at most) side by side
        SGT_PAR(c, K) {
          double* aFc = S.aF + c * CS;
          const int ci = S.hit_pair[Lv * K + c];
          if (ci < 0) continue;
          const double* sc = cscr(ci);
          const double* J = crow(ci);
          const double* W = J + 3 * CS;
          const int sl = (int)sc[CS_SL];
          double p0 = 0, p1 = 0, p2 = 0;
          // whole padded rows (J is zero beyond the body's dofs), unrolled over the capacity with every load issued up front: the rows
          // sit in global memory (L2), and a loop would pay that latency once per trip.  Beyond the padded stride CS (uniform) the
          // loads hit the record's other words (finite), against a zero.
#pragma unroll
          for (int k = 0; k < CHD; k++) {   // (one select, not three)
            const double a = k < CS ? aFc[k] : 0.0, j0 = J[k], j1 = J[CS + k], j2 = J[2 * CS + k];
            p0 += j0 * a; p1 += j1 * a; p2 += j2 * a;
          }
          double w0[CHD], w1[CHD], w2[CHD];   // the W rows are on their way while the block update runs
#pragma unroll
          for (int k = 0; k < CHD; k++) { w0[k] = W[k]; w1[k] = W[CS + k]; w2[k] = W[2 * CS + k]; }
          const double as_ = sl >= 0 ? S.ae[sl] : 0.0;
          double f[3] = {S.cf[3 * ci], S.cf[3 * ci + 1], S.cf[3 * ci + 2]}, df[3];
          const double res[3] = {sc[CS_B] + sc[CS_JS] * as_ + p0 + sc[CS_R] * f[0], sc[CS_B + 1] + sc[CS_JS + 1] * as_ + p1 + sc[CS_R] * f[1],
                                 sc[CS_B + 2] + sc[CS_JS + 2] * as_ + p2 + sc[CS_R] * f[2]};
          imp_par -= contact_block_update_pre(sc + CS_A, sc + CS_PE, res, f, con_mu, df);
          double an[CHD];
#pragma unroll
          for (int k = 0; k < CHD; k++) an[k] = aFc[k] + (w0[k] * df[0] + w1[k] * df[1] + w2[k] * df[2]);
#pragma unroll
          for (int k = 0; k < CHD; k += 4)
            if (k < CS) { aFc[k] = an[k]; aFc[k + 1] = an[k + 1]; aFc[k + 2] = an[k + 2]; aFc[k + 3] = an[k + 3]; }
          S.cf[3 * ci] = f[0]; S.cf[3 * ci + 1] = f[1]; S.cf[3 * ci + 2] = f[2];
          if (sl >= 0) S.ae[sl] += sc[CS_INVM] * (sc[CS_JS] * df[0] + sc[CS_JS + 1] * df[1] + sc[CS_JS + 2] * df[2]);
        }
        SGT_SYNC();
      }
#endif
      SGT_SYNC();
    }
    // ... or one serial list, the lanes spread over the dofs of a contact's chain block(s)
#if SGT_DEVICE
    // (r04) WAVE-SYNCHRONOUS when all chain words fit the wavefront (K CS <= 64: the free ball's two-finger gripper): lane c CS + d
    // keeps chain word d of chain c in a register for the whole pass, the free body's acceleration and S^-1 sit in registers on every
    // lane alike, a contact's J / W words, scalars and object columns come from the work space one contact AHEAD (two register
    // sets), J a is three wavefront sums (DPP), and nothing in the loop waits at a barrier.  Per contact the bulk-synchronous
    // version below pays two barriers -- each draining every outstanding load -- and two exposed round trips to the work space:
    // 7.6 k cycles an update, 70 % of a free-ball substep.
#ifdef SGT_X_NOSF
    const bool serial_fast = false;
#else
    const bool serial_fast = serial_contacts && K * CS <= 64;
#endif
    if (serial_fast) {
      const int lane = (int)threadIdx.x;
      const bool dofl = lane < K * CS;
      const int mc = dofl ? lane / CS : -1, mdl = dofl ? lane % CS : 0;
      // (with a free object its LDS arrays -- B_e, 1 / m, C_e -- are read through typed pointers too: FR is a fact of the instantiation)
      const SGT_LDSP double* const einvmL = (const SGT_LDSP double*)S.einvm;
      const SGT_LDSP double* const CeL = (const SGT_LDSP double*)S.Ce;
      double a = dofl ? aFL[lane] : 0.0;
      // (r05) The body's push a_f += S^-1 w is spread over six lanes: lane q < 6 keeps ROW q of S^-1 and computes component q, six scalar
      // reads hand the result to every lane.  All 36 words on every lane -- parked in accumulation registers and fetched back for each
      // product -- were 108 instructions per contact, and the slider's share C_sl dg_e = -S^-1 B_sl dg_e / D_sl a second such product:
      // now ONE product, S^-1 (J_o' df - B_sl dg_e / D_sl).  ~900 instructions per contact before, a wavefront alone on its SIMD pays ~7
      // cycles for each.
      double af[6] = {0, 0, 0, 0, 0, 0}, gf[6] = {0, 0, 0, 0, 0, 0}, Siq[6] = {0, 0, 0, 0, 0, 0};
      if (FR) {
        const int qr = (lane & 7) < 6 ? (lane & 7) : 5;
#pragma unroll
        for (int q = 0; q < 6; q++) { af[q] = ofL[OF_AF + q]; gf[q] = ofL[OF_GF + q]; Siq[q] = ofL[OF_SINV + 6 * qr + q]; }
      }
      // J a: sums over the lanes that hold chain words, the first K CS of the wavefront -- 8 for a two-finger gripper: a butterfly inside
      // every group of eight lanes (the three rows' sums side by side: each step's DPP moves wait two cycles for the add in front of them),
      // then one scalar read per group in use (a loop over a scalar count: a branch the compiler cannot turn into "do all and select")
      const int ngrp = __builtin_amdgcn_readfirstlane((K * CS + 7) >> 3);
      auto chain_sums = [&](double& x0, double& x1, double& x2) {
        { const double t0 = dpp64<0xB1>(x0), t1 = dpp64<0xB1>(x1), t2 = dpp64<0xB1>(x2); x0 += t0; x1 += t1; x2 += t2; }
        { const double t0 = dpp64<0x4E>(x0), t1 = dpp64<0x4E>(x1), t2 = dpp64<0x4E>(x2); x0 += t0; x1 += t1; x2 += t2; }
        { const double t0 = dpp64<0x141>(x0), t1 = dpp64<0x141>(x1), t2 = dpp64<0x141>(x2); x0 += t0; x1 += t1; x2 += t2; }
        double s0 = readlane64(x0, 0), s1 = readlane64(x1, 0), s2 = readlane64(x2, 0);
        for (int gq = 1; gq < ngrp; gq++) { s0 += readlane64(x0, 8 * gq); s1 += readlane64(x1, 8 * gq); s2 += readlane64(x2, 8 * gq); }
        x0 = s0; x1 = s1; x2 = s2;
      };
      struct SRec { double j0, j1, j2, w0, w1, w2, A[6], Pe[7], B[3], R, invm, Js[3], slf, rowsf, objf, Jo[18]; int ci; };
      auto load_srec = [&](SRec& q, int ci) {
        const double* J = crow(ci);
        const double* sc = J + 12 * CS;
        const int cc12 = hitcntL[ci];   // (c1 + 1) | (c2 + 1) << 8, packed with the rows
        const int c1 = (cc12 & 0xff) - 1, c2 = ((cc12 >> 8) & 0xff) - 1;
        const int blk = (dofl && mc == c1) ? 0 : ((dofl && mc == c2) ? 1 : -1);
        const double* Jb = J + (blk == 1 ? 6 * CS : 0) + (blk >= 0 ? mdl : 0);
        const double z = blk >= 0 ? 1.0 : 0.0;
        q.ci = ci;
        q.j0 = z * Jb[0]; q.j1 = z * Jb[CS]; q.j2 = z * Jb[2 * CS];
        q.w0 = z * Jb[3 * CS]; q.w1 = z * Jb[4 * CS]; q.w2 = z * Jb[5 * CS];
#pragma unroll
        for (int k = 0; k < 6; k++) q.A[k] = sc[CS_A + k];
#pragma unroll
        for (int k = 0; k < 3; k++) { q.B[k] = sc[CS_B + k]; q.Js[k] = sc[CS_JS + k]; }
        q.R = sc[CS_R]; q.invm = sc[CS_INVM]; q.slf = sc[CS_SL]; q.rowsf = sc[CS_ROWS]; q.objf = sc[CS_OBJ];
#pragma unroll
        for (int k = 0; k < 7; k++) q.Pe[k] = sc[CS_PE + k];
        if (FR) {
#pragma unroll
          for (int k = 0; k < 18; k++) q.Jo[k] = sc[CS_JO + k];
        }
      };
      auto update = [&](const SRec& q, const bool have) {
        // (the record's words are the same on every lane: as scalars, the branches on them are real branches, not masked regions)
        const int ci = q.ci, sl = __builtin_amdgcn_readfirstlane((int)q.slf);
        const bool act = have && __builtin_amdgcn_readfirstlane((int)(q.rowsf != 0.0)) != 0, ob = FR && __builtin_amdgcn_readfirstlane((int)(q.objf != 0.0)) != 0;
        double p0 = q.j0 * a, p1 = q.j1 * a, p2 = q.j2 * a;
        chain_sums(p0, p1, p2);
        if (ob) { p0 += dot6(q.Jo, af); p1 += dot6(q.Jo + 6, af); p2 += dot6(q.Jo + 12, af); }
        double as_ = 0.0;
        // (r05) the slider's B_sl and 1 / D_sl are read ONCE, for the slider's acceleration here and for its share of the body's push below:
        // read again there (the stores in between may alias them, as far as the compiler knows) they were six more LDS reads and a wait
        // in every update's dependency chain -- free ball +1.5 %
        double Bs_[6] = {0, 0, 0, 0, 0, 0}, eim = 0.0;
        if (sl >= 0) {
          if (FR) {
#pragma unroll
            for (int k = 0; k < 6; k++) Bs_[k] = BeL[6 * sl + k];
            eim = einvmL[sl];
            as_ = aeL[sl] - (Bs_[0] * af[0] + Bs_[1] * af[1] + Bs_[2] * af[2] + Bs_[3] * af[3] + Bs_[4] * af[4] + Bs_[5] * af[5]) * eim;
          } else as_ = aeL[sl];
        }
        double f[3] = {cfL[3 * ci], cfL[3 * ci + 1], cfL[3 * ci + 2]}, df[3];
        const double res[3] = {q.B[0] + q.Js[0] * as_ + p0 + q.R * f[0], q.B[1] + q.Js[1] * as_ + p1 + q.R * f[1], q.B[2] + q.Js[2] * as_ + p2 + q.R * f[2]};
        const double ch = contact_block_update_pre(q.A, q.Pe, res, f, con_mu, df);
        if (act) {
          imp_uni -= ch;
          a += q.w0 * df[0] + q.w1 * df[1] + q.w2 * df[2];
          const double dge = sl >= 0 ? q.Js[0] * df[0] + q.Js[1] * df[1] + q.Js[2] * df[2] : 0.0;
          if (lane == 0) {
            cfL[3 * ci] = f[0]; cfL[3 * ci + 1] = f[1]; cfL[3 * ci + 2] = f[2];
            if (sl >= 0) aeL[sl] += q.invm * dge;
          }
          if (ob) {   // the push on the body: g_f += J_o' df; a_f += S^-1 J_o' df + C_sl dg_e, C_sl = -S^-1 B_sl / D_sl
            double dg[6], w6[6];
#pragma unroll
            for (int k = 0; k < 6; k++) { dg[k] = q.Jo[k] * df[0] + q.Jo[6 + k] * df[1] + q.Jo[12 + k] * df[2]; w6[k] = dg[k]; }
            double Cs[6] = {0, 0, 0, 0, 0, 0};   // the neighbour-row models keep C_sl (S.Ce); the others fold the slider's share into the one product
            if (sl >= 0) {
              if (NB) {
#pragma unroll
                for (int k = 0; k < 6; k++) Cs[k] = CeL[6 * sl + k];
              } else {
                const double sh = dge * eim;
#pragma unroll
                for (int k = 0; k < 6; k++) w6[k] -= Bs_[k] * sh;
              }
            }
            const double daq = dot6(Siq, w6);   // lane q < 6: component q of S^-1 w
#pragma unroll
            for (int k = 0; k < 6; k++) { gf[k] += dg[k]; af[k] += NB ? readlane64(daq, k) + Cs[k] * dge : readlane64(daq, k); }
          }
        }
      };
      if (ncon > 0) {
        SRec ra, rb;
        load_srec(ra, 0);
        for (int j = 0; j < ncon; j += 2) {
          load_srec(rb, j + 1 < ncon ? j + 1 : 0);
          update(ra, true);
          load_srec(ra, j + 2 < ncon ? j + 2 : 0);
          update(rb, j + 1 < ncon);
        }
      }
      if (dofl) aFL[lane] = a;
      if (FR && lane == 0) {
#pragma unroll
        for (int q = 0; q < 6; q++) { ofL[OF_AF + q] = af[q]; ofL[OF_GF + q] = gf[q]; }
      }
      SGT_SYNC();
    }
    for (int ci = 0; serial_contacts && !serial_fast && ci < ncon; ci++) {
#else
    for (int ci = 0; serial_contacts && ci < ncon; ci++) {
#endif
      const double* sc = cscr(ci);
      if (sc[CS_ROWS] == 0.0) continue;
      const int c1 = (int)sc[CS_C1], c2 = (int)sc[CS_C2], n1 = (int)sc[CS_N1], n2 = (int)sc[CS_N2], sl = (int)sc[CS_SL];
      double p0 = 0, p1 = 0, p2 = 0;
      SGT_PAR(i, n1 + n2) {
        const bool second = i >= n1;
        const int dl = second ? i - n1 : i;
        const double* J = crow(ci) + (second ? 6 * CS : 0);
        const double a = S.aF[(second ? c2 : c1) * CS + dl];
        p0 += J[dl] * a; p1 += J[CS + dl] * a; p2 += J[2 * CS + dl] * a;
      }
      p0 = wsum(p0); p1 = wsum(p1); p2 = wsum(p2);
      const bool ob = FR && sc[CS_OBJ] != 0.0;
      if (ob) { p0 += dot6(sc + CS_JO, S.of + OF_AF); p1 += dot6(sc + CS_JO + 6, S.of + OF_AF); p2 += dot6(sc + CS_JO + 12, S.of + OF_AF); }
      const double as_ = sl >= 0 ? slider_acc(sl) : 0.0;
      double f[3] = {S.cf[3 * ci], S.cf[3 * ci + 1], S.cf[3 * ci + 2]}, df[3];
      const double res[3] = {sc[CS_B] + sc[CS_JS] * as_ + p0 + sc[CS_R] * f[0], sc[CS_B + 1] + sc[CS_JS + 1] * as_ + p1 + sc[CS_R] * f[1],
                             sc[CS_B + 2] + sc[CS_JS + 2] * as_ + p2 + sc[CS_R] * f[2]};
      imp_uni -= contact_block_update_pre(sc + CS_A, sc + CS_PE, res, f, con_mu, df);
      SGT_SYNC();   // every lane has read the old forces and accelerations
      const int n1c = c1 >= 0 ? CS : 0, n2c = c2 >= 0 ? CS : 0;
      SGT_PAR(i, n1c + n2c) {
        const bool second = i >= n1c;
        const int dl = second ? i - n1c : i;
        const double* W = crow(ci) + (second ? 9 * CS : 3 * CS);
        S.aF[(second ? c2 : c1) * CS + dl] += W[dl] * df[0] + W[CS + dl] * df[1] + W[2 * CS + dl] * df[2];
      }
      SGT_ONE {
        S.cf[3 * ci] = f[0]; S.cf[3 * ci + 1] = f[1]; S.cf[3 * ci + 2] = f[2];
        const double dge = sl >= 0 ? sc[CS_JS] * df[0] + sc[CS_JS + 1] * df[1] + sc[CS_JS + 2] * df[2] : 0.0;
        if (sl >= 0) S.ae[sl] += sc[CS_INVM] * dge;
        if (ob) {   // the push on the body: g_f += J_o' df; a_f += S^-1 J_o' df + C_sl dg_e
          double dg[6], da[6];
          for (int q = 0; q < 6; q++) dg[q] = sc[CS_JO + q] * df[0] + sc[CS_JO + 6 + q] * df[1] + sc[CS_JO + 12 + q] * df[2];
          mat6vec(da, S.of + OF_SINV, dg);
          double Cs[6] = {0, 0, 0, 0, 0, 0};
          if (sl >= 0) {
            if (NB) { for (int q = 0; q < 6; q++) Cs[q] = S.Ce[6 * sl + q]; }
            else {
              double Bs[6];
              mat6vec(Bs, S.of + OF_SINV, S.Be + 6 * sl);
              for (int q = 0; q < 6; q++) Cs[q] = -Bs[q] * S.einvm[sl];
            }
          }
          for (int q = 0; q < 6; q++) { S.of[OF_GF + q] += dg[q]; S.of[OF_AF + q] += da[q] + Cs[q] * dge; }
        }
      }
      SGT_SYNC();
    }
    SGT_STAMP(13);
    const double improvement = (wsum(imp_par) + imp_uni) * H.pgs_scale;
    iters = it + 1;
    if (improvement < H.tolerance) break;
  }

  SGT_ONE { S.swc[SWC_ITERS] = iters; S.swc[SWC_TEN_F] = ten_f; }
}

// ---------------------------------------------------------------- the step's stages, each a function of its own
// One env's whole step used to be ONE function: every stage below pasted into the kernel, ~60 array pointers, the plan's tables and
// every stage's temporaries competing for one register allocation -- 850 scalar and 550 - 1 650 vector registers spilled (r04
// profile), the scalar ones into lanes of vector registers that were themselves parked in accumulation registers.  Builds of that
// function that differed only in unrelated places (a profiling stamp, a debugging copy at the end) then disagreed about single
// stores of the contact rows' build -- a word of a contact's record keeping its old value -- which is how a fuzz scene went
// to NaN on one build and not on the next.  Now: the step is a sequence of CALLED functions, one per group of
// stages (PH), each with its own registers; what they hand each other lives in the env's LDS block and work space anyway, and the
// step's few scalars (flags, counts, the touch bits) travel in S.ctx.  On the device a stage finds the launch arguments in the
// kernel-argument segment (uniform: scalar loads) and its env in the workgroup id; on the host they are passed.
enum { CTX_FLAGS = 0, CTX_NCON, CTX_NEFC, CTX_ITERS, CTX_TLO, CTX_THI, CTX_STOP, CTX_LAST, CTX_INTEGRATE, CTX_SUB, CTX_N = 16 };
#if SGT_DEVICE
#define SGT_STAGE_PARAMS SGT_LDSP double* lds_
#define SGT_STAGE_CALL(PH) tree_stage<CHD, PH>((SGT_LDSP double*)lds_base)
#else
#define SGT_STAGE_PARAMS const TreeArgs& A, const int env, double* lds_base
#define SGT_STAGE_CALL(PH) tree_stage<CHD, PH>(A, env, lds_base)
#endif
// PH 1: checks, kinematics, tendons, mass matrix, L'DL + M^-1, bias and smooth accelerations (chains, sliders, free object)
// PH 2: collision -- block culling, the pair walks, rank, narrowphase
// PH 3: constraint rows (equality, limits, contacts), warmstart, the PGS sweeps (tree_sweep)
// PH 4: qacc, sensors, Euler with implicit damping
// (SGT_X_MONO: the r04 layout that produced the dropped stores -- every stage pasted into the kernel, one register allocation for the
//  whole step -- kept buildable for scripts/repro/tree_mono: `build_native.py --ko mono -DSGT_X_MONO`; never the product)
#if defined(SGT_X_MONO) && defined(__HIPCC__)
#define SGT_STAGE_ATTR __host__ __device__ __forceinline__
#else
#define SGT_STAGE_ATTR SGT_NOINLINE
#endif
template <int CHD, int PH>
static SGT_STAGE_ATTR void tree_stage(SGT_STAGE_PARAMS) {
#if SGT_DEVICE
  // (the launch arguments: the kernel left the address of its argument segment in the first word of the LDS block -- a called function
  //  has no register for it -- and the segment is read through the constant address space: uniform, scalar loads)
  lds_ = (SGT_LDSP double*)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)lds_);
  const unsigned long long ka_ = *(const SGT_LDSP unsigned long long*)lds_;
  const SGT_CONST TreeArgs& A = *(const SGT_CONST TreeArgs*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ka_ >> 32)) << 32) |
                                                               (unsigned)__builtin_amdgcn_readfirstlane((int)ka_));
  const int env = (int)blockIdx.x;
  double* const lds_base = (double*)lds_;
#endif
  // The plan tables are read-only for the kernel's lifetime: read through the constant address space, a uniform index is a scalar load
  // (K$) that the compiler may hoist and keep, not a vector load behind a full vmcnt wait after every store
  const SGT_CONST SgPlanHeader& H = *(const SGT_CONST SgPlanHeader*)A.H;
  const SGT_CONST SgTreeDev& T = *(const SGT_CONST SgTreeDev*)A.T;
  const int N = H.nelem, ND = T.ND, NB = T.NB, K = T.K, nv = H.nv, nu = H.nu;
  constexpr int CS = CHD;   // (= T.CS: the plan pads the chains' stride to the instantiation's capacity, sg_plan.cpp)
  const double h = H.timestep;
  Lds S;
  lds_carve(S, lds_base, T, N, H.has_free, A.cws + (size_t)env * A.cws_stride + (size_t)SGT_MAXHIT * SGT_HITREC * SGT_RECW + (size_t)SGT_MAXCON * cws_row_doubles(T.CS) + T.NMAT, nullptr, H.nnb);
  const SGT_CONST double* const elemc = (const SGT_CONST double*)A.elem;
  const SGT_CONST SgGenPair* const gpairs = (const SGT_CONST SgGenPair*)A.gpairs;
  const SGT_CONST int* const nbtab = (const SGT_CONST int*)A.nbtab;
  const SGT_CONST SgEqSlot* const sched = (const SGT_CONST SgEqSlot*)A.sched;
  auto E = [&](int f, int e) { return elemc[(size_t)f * N + e]; };
  double* const cw = A.cws + (size_t)env * A.cws_stride;
  const long long CW = cws_row_doubles(CS);
#if !SGT_DEVICE && defined(SGT_EMU_SEPARATE)
  double* const stage = sep_pool() ? sep_part(0, (size_t)SGT_MAXHIT * SGT_HITREC * SGT_RECW) : cw;
  double* const crow0 = sep_pool() ? sep_part(1, (size_t)SGT_MAXCON * CW) : cw + (size_t)SGT_MAXHIT * SGT_HITREC * SGT_RECW;
#else
  double* const stage = cw;
  double* const crow0 = cw + (size_t)SGT_MAXHIT * SGT_HITREC * SGT_RECW;
#endif
  auto crow = [&](int c) { return crow0 + (size_t)c * CW; };                 // J1[3][CS] | W1[3][CS] | J2[3][CS] | W2[3][CS] | scalars
  auto cscal = [&](int c) { return crow0 + (size_t)c * CW + 12 * CS; };
  auto cscr = [&](int c) -> const double* { return c < S.ncache ? S.csc + (size_t)c * SGT_CSC : crow0 + (size_t)c * CW + 12 * CS; };   // for the sweeps: the LDS copy when there is one
#if !SGT_DEVICE && defined(SGT_EMU_SEPARATE)
  double* const Mg = sep_pool() ? sep_part(2, (size_t)T.NMAT) : crow0 + (size_t)SGT_MAXCON * CW;
#else
  double* const Mg = crow0 + (size_t)SGT_MAXCON * CW;    // the chains' mass-matrix blocks [K][CS][CS], identity-padded
#endif
  auto pidx = [&](int d) { const int c = T.d_chain[d]; return c * CS + d - T.c_dof0[c]; };   // flat chain dof -> index in a padded [K][CS] vector

  const bool FR = H.has_free != 0;   // the composite's elements hang off a free body (6 dofs): the "object block" below
  double* const gq = A.qpos + (size_t)env * H.nq;
  double* const gv = A.qvel + (size_t)env * nv;
  double* const gw = A.warm + (size_t)env * nv;
  double* const gact = A.act + (size_t)env * (nu > 0 ? nu : 1);
  double* const gctrl = A.ctrl + (size_t)env * (nu > 0 ? nu : 1);

  const double kenv = A.kenv[env];
  const double kt0 = A.kmask_ten[H.t0_id] ? kenv : H.t0_k0;
  (void)kenv; (void)gq; (void)gv; (void)gw; (void)gact; (void)gctrl; (void)kt0; (void)stage; (void)Mg; (void)nu; (void)sched; (void)nbtab; (void)gpairs;
  // L'DL of every chain block in S.L at once (mj_factorM on serial chains): step s eliminates dof k = nd - 1 - s of each chain, one lane
  // per row i < k: L[i][j] -= (L[k][i] / L[k][k]) L[k][j] for j <= i, then row k is scaled.  Same operations as chain_factor.
  // The blocks are padded to [CS][CS]; a row's update runs over the whole row (the entries right of the diagonal are never read), so
  // that every lane's loop has the same count.
  int maxnd = 0;
  for (int c = 0; c < K; c++) maxnd = T.c_ndof[c] > maxnd ? T.c_ndof[c] : maxnd;
  auto factor_all = [&]() {
#if SGT_DEVICE && !defined(SGT_X_NOREGLDL)
    // L'DL IN REGISTERS (r04), chains of up to 17 dofs: a lane group per chain, lane i holds row i of the (symmetric) block.  Pivot k
    // (from the last dof down, mj_factorM's order): every lane i < k needs a = M[i][k] / D_k -- its own word and one broadcast -- and
    // row k's words M[k][j] = M[j][k], j < k: the SAME register of the lanes j, k shuffles; then M[i][j] -= a M[k][j] in registers
    // (both triangles are kept, so that the words a lane needs of row k are the column words of the other lanes) and U[i][k] = a is
    // L[k][i].  Row 16 -- the 17th dof of the four-finger gripper's long chains, one more than a group has lanes -- is eliminated
    // first and is never updated: its words are read by every lane of the group.  ~400 shuffles + 140 multiply-adds for all chains
    // at once; step by step through the work space with two barriers a pivot it took 100 k cycles, and it runs twice a substep
    // (M and M + h B: 40 % of a contact-free substep).
    constexpr int NC = CHD < 17 ? CHD : 17;
    if (maxnd <= NC && CS <= 20) {
      const int grp = (int)threadIdx.x >> 4, l = (int)threadIdx.x & 15, gb = (int)threadIdx.x & 48;
      for (int c0 = 0; c0 < K; c0 += 4) {
        const int c = c0 + grp, cc = c < K ? c : 0;
        double* Lc = S.L + cc * CS * CS;
        const bool row = c < K && l < CS;   // the lane holds a row of a chain (else a virtual identity row: every step a no-op)
        double m[NC];
#pragma unroll
        for (int j = 0; j < NC; j++) m[j] = (row && j < CS) ? Lc[l * CS + j] : (j == l ? 1.0 : 0.0);
        if constexpr (NC == 17) {
          if (CS > 16) {   // (uniform)
            const double a = m[16] / Lc[16 * CS + 16];
#pragma unroll
            for (int j = 0; j < 16; j++) m[j] -= a * Lc[16 * CS + j];
            m[16] = a;
          }
        }
#pragma unroll
        for (int k = (NC == 17 ? 15 : NC - 1); k >= 1; k--) {
          const double Dk = __shfl(m[k], gb + k, 64);
          double v[16];
#pragma unroll
          for (int j = 0; j < k; j++) v[j] = __shfl(m[k], gb + j, 64);
          if (l < k) {
            const double a = m[k] / Dk;
#pragma unroll
            for (int j = 0; j < k; j++) m[j] -= a * v[j];
            m[k] = a;
          }
        }
        // back to the work space in chain_solve's layout: D on the diagonal, L[k][i] (i < k) below it -- lane i writes column i
        if (row) {
          double dd = m[0];
#pragma unroll
          for (int j = 1; j < NC; j++) dd = j == l ? m[j] : dd;
          Lc[l * CS + l] = dd;
#pragma unroll
          for (int k = 1; k < NC; k++)
            if (l < k && k < CS) Lc[k * CS + l] = m[k];
        }
      }
      SGT_SYNC();
      return;
    }
#endif
    for (int st = 0; st + 1 < CS; st++) {
      SGT_PAR(idx, K * CS) {
        const int c = idx / CS, i = idx % CS, k = T.c_ndof[c] - 1 - st;
        if (k >= 1 && i < k) {
          double* Lc = S.L + c * CS * CS;
          const double a = Lc[k * CS + i] / Lc[k * CS + k];
          for (int j = 0; j < CS; j += 4) {
            const double l0 = Lc[k * CS + j], l1 = Lc[k * CS + j + 1], l2 = Lc[k * CS + j + 2], l3 = Lc[k * CS + j + 3];
            const double r0 = Lc[i * CS + j], r1 = Lc[i * CS + j + 1], r2 = Lc[i * CS + j + 2], r3 = Lc[i * CS + j + 3];
            Lc[i * CS + j] = r0 - a * l0; Lc[i * CS + j + 1] = r1 - a * l1; Lc[i * CS + j + 2] = r2 - a * l2; Lc[i * CS + j + 3] = r3 - a * l3;
          }
        }
      }
      SGT_SYNC();
      SGT_PAR(idx, K * CS) {
        const int c = idx / CS, i = idx % CS, k = T.c_ndof[c] - 1 - st;
        if (k >= 1 && i < k) {
          double* Lc = S.L + c * CS * CS;
          Lc[k * CS + i] = Lc[k * CS + i] / Lc[k * CS + k];
        }
      }
      SGT_SYNC();
    }
  };
  auto tree_motion = [&](const double* qacc) {
    SGT_PAR(c, K) {
      double w[3] = {0, 0, 0}, al[3] = {0, 0, 0}, a[3] = {-H.gravity[0], -H.gravity[1], -H.gravity[2]}, P[3], r[3], t[3], t2[3];
      for (int k = 0; k < 3; k++) P[k] = T.c_root_pos[c][k];
      for (int bi = 0; bi < T.c_nbody[c]; bi++) {
        const int tb = T.c_body0[c] + bi;
        for (int kj = 0; kj <= T.b_njnt[tb]; kj++) {
          const bool lastj = kj == T.b_njnt[tb];
          const int d = T.b_dof0[tb] + kj;
          const double* Q = lastj ? S.xpos + 3 * tb : S.anchor + 3 * d;
          for (int k = 0; k < 3; k++) r[k] = Q[k] - P[k];
          cross3(t, w, r);
          cross3(t2, al, r); addscl3(a, t2, 1);
          cross3(t2, w, t); addscl3(a, t2, 1);
          for (int k = 0; k < 3; k++) P[k] = Q[k];
          if (lastj) break;
          const double* u = S.axis + 3 * d;
          const double qd = S.v[d], qdd = qacc ? qacc[d] : 0.0;
          cross3(t, w, u);
          addscl3(al, u, qdd); addscl3(al, t, qd);
          addscl3(w, u, qd);
        }
        for (int k = 0; k < 3; k++) { S.bw[3 * tb + k] = w[k]; S.bal[3 * tb + k] = al[k]; S.ba[3 * tb + k] = a[k]; }
      }
    }
  };
  auto slider_acc = [&](int e) {   // a slider's constraint acceleration: with a free object its local part minus the body's share
    return FR ? S.ae[e] - dot6(S.Be + 6 * e, S.of + OF_AF) * S.einvm[e] : S.ae[e];
  };
  int flags = (int)S.ctx[CTX_FLAGS], ncon = (int)S.ctx[CTX_NCON], nefc = (int)S.ctx[CTX_NEFC], iters = (int)S.ctx[CTX_ITERS], stop = 0;
  unsigned touch_lo = (unsigned)S.ctx[CTX_TLO], touch_hi = (unsigned)S.ctx[CTX_THI];
  const bool last = S.ctx[CTX_LAST] != 0.0, integrate = S.ctx[CTX_INTEGRATE] != 0.0;
  const int sub = (int)S.ctx[CTX_SUB];
  (void)last; (void)integrate; (void)sub; (void)nefc; (void)iters; (void)touch_lo; (void)touch_hi; (void)ncon;
  SGT_SYNC();   // (every lane has the step's scalars before lane 0 writes them back)
  SGT_STAMP_INIT();
  {
    if constexpr (PH == 1) {
    // ---------------------------------------------------------------- mj_checkPos / mj_checkVel
    {
      double bad = 0;
      SGT_PAR(d, ND) bad += (isbad(S.q[d]) ? 1.0 : 0.0) + (isbad(S.v[d]) ? 1024.0 : 0.0);
      SGT_PAR(e, N) bad += (isbad(S.qe[e]) ? 1.0 : 0.0) + (isbad(S.ve[e]) ? 1024.0 : 0.0);
      if (FR) SGT_PAR(c, 7) bad += (isbad(S.of[OF_P + c]) ? 1.0 : 0.0) + ((c < 3 && (isbad(S.of[OF_VW + c]) || isbad(S.of[OF_WL + c]))) ? 1024.0 : 0.0);
      bad = wsum(bad);
      if (bad > 0) {
        const int nb = (int)bad;
        if (nb % 1024) flags |= SG_FLAG_BADQPOS;
        if (nb / 1024) flags |= SG_FLAG_BADQVEL;
        { stop = 1; goto stage_done; }   // uniform: the env stops integrating for the rest of the call
      }
    }
    SGT_STAMP(0);
    // ---------------------------------------------------------------- stage 1: kinematics, one lane per chain
    SGT_PAR(c, K) {
      double pos[3], quat[4], mat[9], ppos[3], pquat[4], pmat[9], t[3];
      for (int k = 0; k < 3; k++) ppos[k] = T.c_root_pos[c][k];
      for (int k = 0; k < 4; k++) pquat[k] = T.c_root_quat[c][k];
      quat2mat(pmat, pquat);
      for (int bi = 0; bi < T.c_nbody[c]; bi++) {
        const int tb = T.c_body0[c] + bi;
        mulmat3(t, pmat, T.b_pos[tb]);
        for (int k = 0; k < 3; k++) pos[k] = ppos[k] + t[k];
        quatmul(quat, pquat, T.b_quat[tb]);
        for (int kj = 0; kj < T.b_njnt[tb]; kj++) {
          const int d = T.b_dof0[tb] + kj;
          quat2mat(mat, quat);
          mulmat3(t, mat, T.d_pos[d]);
          for (int k = 0; k < 3; k++) S.anchor[3 * d + k] = pos[k] + t[k];
          mulmat3(S.axis + 3 * d, mat, T.d_axis[d]);
          const double dq = S.q[d] - T.d_qpos0[d], sn = sin(0.5 * dq);
          const double ql[4] = {cos(0.5 * dq), T.d_axis[d][0] * sn, T.d_axis[d][1] * sn, T.d_axis[d][2] * sn};
          quatmul(quat, quat, ql);
          quat2mat(mat, quat);
          mulmat3(t, mat, T.d_pos[d]);
          for (int k = 0; k < 3; k++) pos[k] = S.anchor[3 * d + k] - t[k];
        }
        const double nq = sqrt(quat[0] * quat[0] + quat[1] * quat[1] + quat[2] * quat[2] + quat[3] * quat[3]);
        for (int k = 0; k < 4; k++) quat[k] /= nq;
        quat2mat(mat, quat);
        for (int k = 0; k < 3; k++) S.xpos[3 * tb + k] = pos[k];
        for (int k = 0; k < 9; k++) S.xmat[9 * tb + k] = mat[k];
        mulmat3(t, mat, T.b_ipos[tb]);
        for (int k = 0; k < 3; k++) S.xipos[3 * tb + k] = pos[k] + t[k];
        double RI[9], Rt[9];
        mulmat33(RI, mat, T.b_imat[tb]);
        for (int a = 0; a < 3; a++)
          for (int b = 0; b < 3; b++) Rt[3 * a + b] = mat[3 * b + a];
        mulmat33(S.ximat + 9 * tb, RI, Rt);
        for (int k = 0; k < 3; k++) ppos[k] = pos[k];
        for (int k = 0; k < 4; k++) pquat[k] = quat[k];
        for (int k = 0; k < 9; k++) pmat[k] = mat[k];
      }
    }
    if (FR) SGT_ONE {   // the free body's pose IS its 7 positions; velocities, gravity and the warmstart in its frame
      double* o = S.of;
      const double nq = sqrt(o[OF_Q] * o[OF_Q] + o[OF_Q + 1] * o[OF_Q + 1] + o[OF_Q + 2] * o[OF_Q + 2] + o[OF_Q + 3] * o[OF_Q + 3]);
      double qn[4] = {o[OF_Q] / nq, o[OF_Q + 1] / nq, o[OF_Q + 2] / nq, o[OF_Q + 3] / nq};
      quat2mat(o + OF_R, qn);
      mulmatT3(o + OF_VL, o + OF_R, o + OF_VW);
      mulmatT3(o + OF_GL, o + OF_R, H.gravity);
      mulmatT3(o + OF_WB, o + OF_R, o + OF_WARM);
      for (int c = 0; c < 3; c++) o[OF_WB + 3 + c] = o[OF_WARM + 3 + c];
      double t[3];
      mulmat3(t, o + OF_R, H.center_pos);
      for (int c = 0; c < 3; c++) o[OF_CEN + c] = o[OF_P + c] + t[c];
    }
    SGT_SYNC();
    SGT_PAR(g, T.NG) {
      const int tb = T.g_body[g];
      double t[3];
      mulmat3(t, S.xmat + 9 * tb, T.g_pos[g]);
      for (int k = 0; k < 3; k++) S.gpos[3 * g + k] = S.xpos[3 * tb + k] + t[k];
      mulmat33(S.gmat + 9 * g, S.xmat + 9 * tb, T.g_mat[g]);
    }
    SGT_PAR(s, T.NS) {
      const int tb = T.s_body[s];
      double t[3];
      mulmat3(t, S.xmat + 9 * tb, T.s_pos[s]);
      for (int k = 0; k < 3; k++) S.spos[3 * s + k] = S.xpos[3 * tb + k] + t[k];
    }
    SGT_SYNC();
    SGT_STAMP(1);
    // ---------------------------------------------------------------- stage 3: tendons.  segments, then one lane per dof
    SGT_PAR(i, K * SGT_MAXTS) {
      const int c = i / SGT_MAXTS, w = i % SGT_MAXTS;
      if (T.t_has[c] && w + 1 < T.t_nsite[c]) {
        const int s0 = T.t_site[c][w], s1 = T.t_site[c][w + 1];
        const double* p0 = s0 >= 0 ? S.spos + 3 * s0 : T.t_fixed[c][w];
        const double* p1 = s1 >= 0 ? S.spos + 3 * s1 : T.t_fixed[c][w + 1];
        double dif[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
        const double len = sqrt(dot3(dif, dif));
        double* sg = S.seg + 4 * i;
        sg[3] = len;
        for (int k = 0; k < 3; k++) sg[k] = len < SG_MINVAL ? 0.0 : dif[k] / len;
      }
    }
    SGT_SYNC();
    SGT_PAR(d, ND) {
      const int c = T.d_chain[d], dl = d - T.c_dof0[c];
      double J = 0;
      if (T.t_has[c])
        for (int w = 0; w + 1 < T.t_nsite[c]; w++) {
          const double* sg = S.seg + 4 * (c * SGT_MAXTS + w);
          if (sg[3] < SG_MINVAL) continue;
          const int s0 = T.t_site[c][w], s1 = T.t_site[c][w + 1];
          double r[3], jp[3];
          if (s1 >= 0 && T.b_nabove[T.s_body[s1]] > dl) {
            for (int k = 0; k < 3; k++) r[k] = S.spos[3 * s1 + k] - S.anchor[3 * d + k];
            cross3(jp, S.axis + 3 * d, r);
            J += dot3(sg, jp);
          }
          if (s0 >= 0 && T.b_nabove[T.s_body[s0]] > dl) {
            for (int k = 0; k < 3; k++) r[k] = S.spos[3 * s0 + k] - S.anchor[3 * d + k];
            cross3(jp, S.axis + 3 * d, r);
            J -= dot3(sg, jp);
          }
        }
      S.tenJ[d] = J;
    }
    SGT_SYNC();
    // tendon length / velocity, spring-damper and actuator force (stages 3, 7, 8), one lane per chain
    SGT_PAR(c, K) {
      double* cs = S.chs + c * CHS_N;
      double Lt = 0, vel = 0;
      if (T.t_has[c]) {
        for (int w = 0; w + 1 < T.t_nsite[c]; w++) Lt += S.seg[4 * (c * SGT_MAXTS + w) + 3];
        for (int dl = 0; dl < T.c_ndof[c]; dl++) vel += S.tenJ[T.c_dof0[c] + dl] * S.v[T.c_dof0[c] + dl];
      }
      cs[CHS_TLEN] = Lt; cs[CHS_TVEL] = vel;
      cs[CHS_TFRC] = T.t_has[c] ? -cs[CHS_KT] * (Lt - T.t_lspring[c]) - T.t_damping[c] * vel : 0.0;
      double afrc = 0, adot = 0;
      if (T.a_has[c]) {
        const double g = T.a_gear[c];
        adot = (cs[CHS_CTRL] - cs[CHS_ACT]) / fmax(SG_MINVAL, T.a_tc[c]);
        afrc = T.a_gain[c] * cs[CHS_ACT] + T.a_bias[c][0] + T.a_bias[c][1] * (g * Lt) + T.a_bias[c][2] * (g * vel);
        afrc *= g;   // qfrc_actuator = gear * J * force
      }
      cs[CHS_AFRC] = afrc; cs[CHS_ACTDOT] = adot;
    }
    SGT_STAMP(2);
    // ---------------------------------------------------------------- stage 4: mass matrix, one lane per entry of the lower triangles
    // (the identity padding of the blocks never changes: written by the first forward pass of a call)
    if (sub == 0) SGT_PAR(i, T.NMAT) {
      const int c = i / (CS * CS), a = (i % (CS * CS)) / CS, b = i % CS, nd = T.c_ndof[c];
      if (a >= nd || b >= nd) Mg[i] = a == b ? 1.0 : 0.0;
    }
    // one lane per entry of the LOWER TRIANGLES only (r04: 840 items for the four-finger gripper's four 20 x 20 blocks instead of 1 600,
    // of which the upper ones idled through the trips of their wavefront): item = (chain, triangular index)
    const int TRI = CS * (CS + 1) / 2;
    SGT_PAR(i, K * TRI) {
      const int c = i / TRI, tt = i % TRI, nd = T.c_ndof[c];
      int a = (int)((sqrt(8.0 * tt + 1.0) - 1.0) * 0.5);
      while (a * (a + 1) / 2 > tt) a--;
      while ((a + 1) * (a + 2) / 2 <= tt) a++;
      const int b = tt - a * (a + 1) / 2;
      if (a < nd) {
        const int da = T.c_dof0[c] + a, db = T.c_dof0[c] + b;
        double s = a == b ? T.d_armature[da] : 0.0;
        for (int tb = T.d_body[da]; tb < T.c_body0[c] + T.c_nbody[c]; tb++) {
          const double mass = T.b_mass[tb];
          if (mass <= 0) continue;
          double ra[3], rb[3], ja[3], jb[3], Ir[3];
          for (int k = 0; k < 3; k++) { ra[k] = S.xipos[3 * tb + k] - S.anchor[3 * da + k]; rb[k] = S.xipos[3 * tb + k] - S.anchor[3 * db + k]; }
          cross3(ja, S.axis + 3 * da, ra);
          cross3(jb, S.axis + 3 * db, rb);
          mulmat3(Ir, S.ximat + 9 * tb, S.axis + 3 * da);
          s += mass * dot3(ja, jb) + dot3(Ir, S.axis + 3 * db);
        }
        Mg[c * CS * CS + a * CS + b] = s;
        Mg[c * CS * CS + b * CS + a] = s;
      }
    }
    SGT_SYNC();
    SGT_STAMP(3);
    SGT_PAR(i, T.NMAT) S.L[i] = Mg[i];
    SGT_SYNC();
    factor_all();
    SGT_PAR(idx, K * CS) {   // M^-1 by columns (= rows): solve for the unit vectors (the padding rows come out as unit vectors too)
      const int c = idx / CS, dl = idx % CS;
      double* x = S.Minv + c * CS * CS + dl * CS;
      for (int k = 0; k < CS; k++) x[k] = k == dl ? 1.0 : 0.0;
      chain_solve_reg<CHD>(S.L + c * CS * CS, CS, x);
    }
    SGT_STAMP(4);
    // ---------------------------------------------------------------- stage 7: bias forces (RNE with qacc = 0), body velocities
    tree_motion(nullptr);
    SGT_SYNC();
    SGT_PAR(tb, NB) {
      const double *w = S.bw + 3 * tb, *al = S.bal + 3 * tb;
      double c[3], t[3], t2[3], f[3], n[3], Iw[3];
      for (int k = 0; k < 3; k++) { c[k] = S.xipos[3 * tb + k] - S.xpos[3 * tb + k]; f[k] = S.ba[3 * tb + k]; }
      cross3(t, al, c); addscl3(f, t, 1);
      cross3(t, w, c); cross3(t2, w, t); addscl3(f, t2, 1);
      for (int k = 0; k < 3; k++) f[k] *= T.b_mass[tb];
      mulmat3(n, S.ximat + 9 * tb, al);
      mulmat3(Iw, S.ximat + 9 * tb, w);
      cross3(t, w, Iw); addscl3(n, t, 1);
      for (int k = 0; k < 3; k++) { S.bf[3 * tb + k] = f[k]; S.bn[3 * tb + k] = n[k]; }
    }
    SGT_SYNC();
    SGT_PAR(d, ND) {
      const int c = T.d_chain[d];
      double s = 0;
      for (int tb = T.d_body[d]; tb < T.c_body0[c] + T.c_nbody[c]; tb++) {
        if (T.b_mass[tb] <= 0) continue;
        double r[3], jp[3];
        for (int k = 0; k < 3; k++) r[k] = S.xipos[3 * tb + k] - S.anchor[3 * d + k];
        cross3(jp, S.axis + 3 * d, r);
        s += dot3(jp, S.bf + 3 * tb) + dot3(S.axis + 3 * d, S.bn + 3 * tb);
      }
      S.bias[d] = s;
      // passive (joint spring / damper, tendon spring / damper) - bias + actuator
      const double* cs = S.chs + c * CHS_N;
      const double pas = -S.kd[d] * (S.q[d] - T.d_springref[d]) - T.d_damping[d] * S.v[d] + S.tenJ[d] * cs[CHS_TFRC];
      S.fs[d] = pas - s + S.tenJ[d] * cs[CHS_AFRC];
    }
    SGT_PAR(i, K * CS) S.tmpP[i] = 0;
    SGT_SYNC();
    SGT_PAR(d, ND) S.tmpP[pidx(d)] = S.fs[d];
    SGT_SYNC();
    SGT_PAR(c, K) chain_solve_reg<CHD>(S.L + c * CS * CS, CS, S.tmpP + c * CS);
    SGT_SYNC();
    SGT_PAR(d, ND) S.asm_[d] = S.tmpP[pidx(d)];
    SGT_STAMP(5);
    // ---------------------------------------------------------------- the composite's sliders: smooth forces (stages 7 - 9)
    double t0_len = 0, t0_vel = 0;
    SGT_PAR(e, N) { t0_len += E(SGE_COEF, e) * S.qe[e]; t0_vel += E(SGE_COEF, e) * S.ve[e]; }
    t0_len = wsum(t0_len); t0_vel = wsum(t0_vel);
    const double t0_frc = -kt0 * (t0_len - H.t0_lspring) - H.t0_damping * t0_vel;
    if (FR) {
      // ---- the free object, everything in the body's frame.  Dofs: (v, w) of the body -- v turned into its frame --
      // and the sliders.  Mass matrix [[M_ff, B], [B', D]]: M_ff from the total mass, first moment and inertia about the body's
      // origin (they move with the sliders: three reductions), B_e constant, D diagonal.  Bias: RNE over a star -- the body
      // and its leaves (oracle tree_motion / rne_bias): a leaf's centre of mass accelerates with -g + 2 (w x a_e) s'_e + w x (w x k_e).
      const double* o = S.of;
      const double w[3] = {o[OF_WL], o[OF_WL + 1], o[OF_WL + 2]}, gl[3] = {o[OF_GL], o[OF_GL + 1], o[OF_GL + 2]};
      double acc[15];   // force (3), torque about the origin (3), first moment (3), inertia about the origin (6: 00 01 02 11 12 22)
      for (int k = 0; k < 15; k++) acc[k] = 0;
      SGT_PAR(e, N) {
        const double m = E(SGE_MASS, e), sd = S.qe[e] - E(SGE_QPOS0, e), a[3] = {E(SGE_AX, e), E(SGE_AY, e), E(SGE_AZ, e)};
        const double k[3] = {E(SGE_KX, e) + a[0] * sd, E(SGE_KY, e) + a[1] * sd, E(SGE_KZ, e) + a[2] * sd};
        const double Ie[9] = {E(SGE_I00, e), E(SGE_I01, e), E(SGE_I02, e), E(SGE_I01, e), E(SGE_I11, e), E(SGE_I12, e), E(SGE_I02, e), E(SGE_I12, e), E(SGE_I22, e)};
        double t[3], t2[3], f[3], n[3], Iw[3], kxf[3];
        cross3(t, w, a);
        cross3(t2, w, k); cross3(f, w, t2);
        for (int c = 0; c < 3; c++) f[c] = m * (f[c] - gl[c] + 2 * t[c] * S.ve[e]);
        mulmat3(Iw, Ie, w);
        cross3(n, w, Iw);
        cross3(kxf, k, f);
        const double pas = -S.ke[e] * (S.qe[e] - E(SGE_SPRINGREF, e)) - E(SGE_DAMPING, e) * S.ve[e] + E(SGE_COEF, e) * t0_frc;
        S.fse[e] = pas - dot3(a, f);
        const double kk = dot3(k, k);
        for (int c = 0; c < 3; c++) { acc[c] += f[c]; acc[3 + c] += kxf[c] + n[c]; acc[6 + c] += m * k[c]; }
        acc[9] += Ie[0] + m * (kk - k[0] * k[0]); acc[10] += Ie[1] - m * k[0] * k[1]; acc[11] += Ie[2] - m * k[0] * k[2];
        acc[12] += Ie[4] + m * (kk - k[1] * k[1]); acc[13] += Ie[5] - m * k[1] * k[2]; acc[14] += Ie[8] + m * (kk - k[2] * k[2]);
        double cl[3] = {E(SGE_GX, e) + a[0] * sd, E(SGE_GY, e) + a[1] * sd, E(SGE_GZ, e) + a[2] * sd}, cw[3];   // the capsule's centre, world
        mulmat3(cw, o + OF_R, cl);
        for (int c = 0; c < 3; c++) S.ecen[3 * e + c] = o[OF_P + c] + cw[c];
      }
      for (int k = 0; k < 15; k++) acc[k] = wsum(acc[k]);
      SGT_SYNC();
      SGT_ONE {
        double* ow = S.of;
        const double mF = H.free_mass, *c = H.free_com;
        double t2[3], f[3], n[3], Iw[3], cxf[3];
        cross3(t2, w, c); cross3(f, w, t2);
        for (int q = 0; q < 3; q++) f[q] = mF * (f[q] - gl[q]);
        mulmat3(Iw, H.free_inertia, w);
        cross3(n, w, Iw);
        cross3(cxf, c, f);
        const double cc = dot3(c, c);
        double F6[6], mk[3], I6[6];
        for (int q = 0; q < 3; q++) { F6[q] = acc[q] + f[q]; F6[3 + q] = acc[3 + q] + cxf[q] + n[q]; mk[q] = acc[6 + q] + mF * c[q]; }
        I6[0] = acc[9] + H.free_inertia[0] + mF * (cc - c[0] * c[0]); I6[1] = acc[10] + H.free_inertia[1] - mF * c[0] * c[1];
        I6[2] = acc[11] + H.free_inertia[2] - mF * c[0] * c[2]; I6[3] = acc[12] + H.free_inertia[4] + mF * (cc - c[1] * c[1]);
        I6[4] = acc[13] + H.free_inertia[5] - mF * c[1] * c[2]; I6[5] = acc[14] + H.free_inertia[8] + mF * (cc - c[2] * c[2]);
        for (int q = 0; q < 6; q++) ow[OF_BIAS + q] = F6[q];
        // M_ff = [[m I, -[mk]x], [[mk]x, I_o]]; kept (21 numbers) for the Euler step's S' = M_ff - sum B B' / (D + h d)
        double Mff[36];
        for (int q = 0; q < 36; q++) Mff[q] = 0;
        Mff[0] = Mff[7] = Mff[14] = H.obj_msum;
        Mff[0 * 6 + 4] = mk[2]; Mff[0 * 6 + 5] = -mk[1]; Mff[1 * 6 + 3] = -mk[2]; Mff[1 * 6 + 5] = mk[0]; Mff[2 * 6 + 3] = mk[1]; Mff[2 * 6 + 4] = -mk[0];
        Mff[21] = I6[0]; Mff[22] = I6[1]; Mff[23] = I6[2]; Mff[28] = I6[3]; Mff[29] = I6[4]; Mff[35] = I6[5];
        for (int r = 0; r < 6; r++)
          for (int q = 0; q < r; q++) Mff[6 * r + q] = Mff[6 * q + r];
        double Sm[36];
        int qq = 0;
        for (int r = 0; r < 6; r++)
          for (int q = r; q < 6; q++) { Sm[6 * r + q] = Sm[6 * q + r] = Mff[6 * r + q] - H.obj_BBD[qq]; ow[OF_MFF + qq] = Mff[6 * r + q]; qq++; }
        spd_inverse6(Sm, ow + OF_SINV);
      }
      SGT_SYNC();
      double red[6] = {0, 0, 0, 0, 0, 0};
      SGT_PAR(e, N)
        for (int q = 0; q < 6; q++) red[q] += S.Be[6 * e + q] * S.fse[e] * S.einvm[e];
      for (int q = 0; q < 6; q++) red[q] = wsum(red[q]);
      double rhs[6], af6[6];
      for (int q = 0; q < 6; q++) rhs[q] = -o[OF_BIAS + q] - red[q];
      mat6vec(af6, o + OF_SINV, rhs);
      SGT_SYNC();
      SGT_ONE { for (int q = 0; q < 6; q++) S.of[OF_ASM + q] = af6[q]; }
      SGT_PAR(e, N) S.asme[e] = (S.fse[e] - dot6(S.Be + 6 * e, af6)) * S.einvm[e];
    }
    if (!FR) SGT_PAR(e, N) {
      const double m = E(SGE_MASS, e), ga = H.gravity[0] * E(SGE_AX, e) + H.gravity[1] * E(SGE_AY, e) + H.gravity[2] * E(SGE_AZ, e);
      const double pas = -S.ke[e] * (S.qe[e] - E(SGE_SPRINGREF, e)) - E(SGE_DAMPING, e) * S.ve[e] + E(SGE_COEF, e) * t0_frc;
      S.fse[e] = pas + m * ga;   // - bias, bias = -m g . axis
      S.asme[e] = S.fse[e] / (m + E(SGE_ARMATURE, e));
      const double dq = S.qe[e] - E(SGE_QPOS0, e);   // the capsule's centre (the pair walk reads it from LDS)
      S.ecen[3 * e] = E(SGE_GX, e) + E(SGE_AX, e) * dq; S.ecen[3 * e + 1] = E(SGE_GY, e) + E(SGE_AY, e) * dq; S.ecen[3 * e + 2] = E(SGE_GZ, e) + E(SGE_AZ, e) * dq;
    }
    SGT_SYNC();

    SGT_STAMP(6);
    }
    if constexpr (PH == 2) {
    // ---------------------------------------------------------------- stage 5: collision over the candidate-pair table
    SGT_ONE { S.icnt[IC_NHIT] = 0; S.icnt[IC_NLIVE] = 0; S.icnt[IC_NPURE] = 0; }
    SGT_SYNC();
    auto elem_center = [&](int e, double* c) { c[0] = S.ecen[3 * e]; c[1] = S.ecen[3 * e + 1]; c[2] = S.ecen[3 * e + 2]; };
    // The table is walked a BLOCK (64 consecutive pairs: one trip of the wavefront) at a time.  A block of (capsule | centre sphere) x
    // finger-box pairs only -- most of the table: a finger body's boxes against 32 elements -- is skipped while every box in it is out
    // of reach of the bounding box of the object (element centres and the centre sphere): no pair of it can pass its own bounding
    // test, let alone produce a contact.  The plan lists a block's boxes behind the table (sg_plan.cpp); the live blocks are
    // gathered in parallel (in any order: the hits are ranked by pair index afterwards), then walked.
    const int ngpair = H.ngpair, nblk = (ngpair + 63) / 64;
    const bool cull = nblk <= 2 * SGT_MAXHIT && T.NG <= SGT_MAXHIT;   // the list lives in hit_sorted + hit_cnt, the boxes' flags in hit_off
    int* const live = S.hit_sorted;
    if (cull) {
      double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
      SGT_PAR(e, N)
        for (int k = 0; k < 3; k++) { lo[k] = fmin(lo[k], S.ecen[3 * e + k]); hi[k] = fmax(hi[k], S.ecen[3 * e + k]); }
      for (int k = 0; k < 3; k++) { lo[k] = -wmax(-lo[k]); hi[k] = wmax(hi[k]); }
      if (H.has_center) {
        const double* cenw = FR ? S.of + OF_CEN : H.center_pos;
        const double ex = fmax(0.0, H.center_radius - H.cap_rbound);
        for (int k = 0; k < 3; k++) { lo[k] = fmin(lo[k], cenw[k] - ex); hi[k] = fmax(hi[k], cenw[k] + ex); }
      }
      SGT_PAR(g, T.NG) {
        double d2 = 0;
        for (int k = 0; k < 3; k++) {
          const double x = S.gpos[3 * g + k], d = fmax(fmax(lo[k] - x, x - hi[k]), 0.0);
          d2 += d * d;
        }
        const double reach = (T.g_rbound[g] + H.cap_rbound + H.con_margin) * 1.000001 + 1e-9;   // (the pairs' own bounds are floats rounded up)
        S.hit_off[g] = d2 > reach * reach ? 1 : 0;
      }
      SGT_SYNC();
      SGT_PAR(b, nblk) {
        const SgGenPair d = gpairs[ngpair + 1 + b];
        bool far = d.kind > 0;
        for (int j = 0; j < d.kind; j++) far = far && S.hit_off[(d.g1 >> (8 * j)) & 0xFF] != 0;
        if (far) continue;
        if (d.kind != 0) live[2 * SGT_MAXHIT - 1 - lds_inc(&S.icnt[IC_NPURE])] = b;   // blocks of the common kind: their own list, from the far end
        else live[lds_inc(&S.icnt[IC_NLIVE])] = b;
      }
      SGT_SYNC();
    }
    SGT_STAMP(20);
    const int nlive = cull ? S.icnt[IC_NLIVE] : nblk;
    {  // blocks of (capsule | centre sphere) x finger box pairs: no kinds to tell apart, everything in LDS -- a trip is ~40 instructions,
       // a fraction of the latency of its table words, so the words of 8 trips are fetched together
      const int npure = cull ? S.icnt[IC_NPURE] : 0;
      const double* const cen0 = FR ? S.of + OF_CEN : H.center_pos;
      const double cen[3] = {cen0[0], cen0[1], cen0[2]};
      constexpr int G = SGT_DEVICE ? 8 : 1;
      const double reach2 = (H.cap_rbound + H.con_margin) * (H.cap_rbound + H.con_margin) * (1.0 + 1e-12);
      for (int b0 = 0; b0 < npure; b0 += G) {
        SgGenPair buf[G];
        int blks[G];
#pragma unroll
        for (int g = 0; g < G; g++) {
          buf[g].kind = SGP_UNSUPPORTED; buf[g].g1 = buf[g].g2 = buf[g].pad = 0;
          blks[g] = b0 + g < npure ? live[2 * SGT_MAXHIT - 1 - (b0 + g)] : -1;
          const int p = blks[g] * 64 + SGT_FIRST;
          // (unconditional loads, all eight in flight together: a trip beyond the list or the table reads the sentinel entry at [ngpair])
          if (SGT_DEVICE) buf[g] = gpairs[blks[g] >= 0 && p < ngpair ? p : ngpair];
        }
#pragma unroll
        for (int g = 0; g < G; g++) {
          if (blks[g] < 0) continue;
          SGT_PAR(j, 64) {
            const int p = blks[g] * 64 + j;
            if (p >= ngpair) continue;
            if (!SGT_DEVICE) buf[g] = gpairs[p];
            const SgGenPair gp = buf[g];
            if (gp.kind == SGP_UNSUPPORTED) continue;
            const int i2 = sgg_index(gp.g2), i1 = sgg_index(gp.g1), k1 = sgg_kind(gp.g1);
            const bool el = k1 == SGG_ELEM, ctr = k1 == SGG_CENTER;
            const double* const c1 = (el ? S.ecen : S.gpos) + (ctr ? 0 : 3 * i1);   // (both in LDS)
            float bf;
            memcpy(&bf, &gp.pad, 4);
            const double bound = (double)bf;
            const double dif[3] = {S.gpos[3 * i2] - (ctr ? cen[0] : c1[0]), S.gpos[3 * i2 + 1] - (ctr ? cen[1] : c1[1]), S.gpos[3 * i2 + 2] - (ctr ? cen[2] : c1[2])};
            if (dot3(dif, dif) > bound * bound) continue;
            if (el) {   // tighter: the capsule's bounding sphere against the box itself
              const double t[3] = {-dif[0], -dif[1], -dif[2]};
              double loc[3];
              mulmatT3(loc, S.gmat + 9 * i2, t);
              // (box_sdist(loc, size) - cap_rbound > margin, without the root: the distance to the box squared against the reach
              //  squared, a hair more permissive -- a filter; the narrowphase decides)
              double q = 0;
              for (int k = 0; k < 3; k++) { const double d = fmax(fabs(loc[k]) - S.gsz[3 * i2 + k], 0.0); q += d * d; }
              if (q > reach2) continue;
            }
            const int idx = lds_inc(&S.icnt[IC_NHIT]);
            if (idx < SGT_MAXHIT) S.hit_pair[idx] = p;
          }
        }
      }
    }
    SGT_STAMP(21);
    {  // 64 pairs at a time; the next trip's table words are fetched before this trip's tests.  The bounding distance of a pair
       // (sum of the bounding radii + margin; plane pairs: rbound + margin) travels in the table as a float rounded up: a filter
       // that passes every pair the exact test passes -- the narrowphase decides
      SgGenPair nxt;
      nxt.kind = SGP_PLANE_CAP; nxt.g1 = nxt.g2 = 0; nxt.pad = (int)0xff800000u;   // (a pair that never hits: bound = -inf)
      if (SGT_DEVICE && nlive > 0) {
        const int p0 = (cull ? live[0] : 0) * 64 + SGT_FIRST;
        nxt = gpairs[p0 < ngpair ? p0 : ngpair];
      }
      for (int bi = 0; bi < nlive; bi++) {
        const int blk = cull ? live[bi] : bi;
        SgGenPair cur = nxt;
        if (SGT_DEVICE) {
          const int p1 = (bi + 1 < nlive ? (cull ? live[bi + 1] : bi + 1) : nblk) * 64 + SGT_FIRST;
          nxt = gpairs[p1 < ngpair ? p1 : ngpair];
        }
        SGT_PAR(j, 64) {
        const int p = blk * 64 + j;
        if (p >= ngpair) continue;
        if (!SGT_DEVICE) cur = gpairs[p];
        const SgGenPair gp = cur;
        const int i1 = sgg_index(gp.g1), i2 = sgg_index(gp.g2), k2 = sgg_kind(gp.g2);
        // (a pair with other contact parameters than the plan's one set, SGP_UNSUPPORTED, is walked like any pair of its geometry:
        //  the narrowphase below turns what would be its contact into the unsupported-pair flag)
        const int gk = gp.kind == SGP_UNSUPPORTED ? sgp_geometry(gp.g1, gp.g2) : gp.kind;
        float bf;
        memcpy(&bf, &gp.pad, 4);
        const double bound = (double)bf;
        bool hit = false;
        const double* cenw = FR ? S.of + OF_CEN : H.center_pos;   // the centre sphere (on the free body when there is one)
        if (gk == SGP_PLANE_CAP || gk == SGP_PLANE_BOX || gk == SGP_PLANE_SPH) {
          const double* c = gk == SGP_PLANE_SPH ? cenw : gk == SGP_PLANE_CAP ? S.ecen + 3 * i2 : (k2 == SGG_BOX ? S.gpos + 3 * i2 : H.st_pos[i2]);
          const double dif[3] = {c[0] - H.plane_pos[0], c[1] - H.plane_pos[1], c[2] - H.plane_pos[2]};
          hit = !(dot3(dif, H.plane_normal) > bound);
        } else {
          // geom2 is a box (finger or static); geom1 the centre sphere, an element capsule or a box
          const double* bp = k2 == SGG_BOX ? S.gpos + 3 * i2 : H.st_pos[i2];
          const int k1 = sgg_kind(gp.g1);
          const double* c = k1 == SGG_CENTER ? cenw : (k1 == SGG_ELEM ? S.ecen + 3 * i1 : (k1 == SGG_BOX ? S.gpos + 3 * i1 : H.st_pos[i1]));
          const double dif[3] = {bp[0] - c[0], bp[1] - c[1], bp[2] - c[2]};
          hit = !(dot3(dif, dif) > bound * bound);
          if (hit && k1 == SGG_ELEM) {   // tighter: the capsule's bounding sphere against the box itself
            const double* bm = k2 == SGG_BOX ? S.gmat + 9 * i2 : H.st_mat[i2];
            const double* sz = k2 == SGG_BOX ? T.g_size[i2] : H.st_size[i2];
            const double t[3] = {-dif[0], -dif[1], -dif[2]};
            double loc[3];
            mulmatT3(loc, bm, t);
            hit = !(box_sdist(loc, sz) - H.cap_rbound > H.con_margin);
          }
        }
        if (hit) {
          const int idx = lds_inc(&S.icnt[IC_NHIT]);
          if (idx < SGT_MAXHIT) S.hit_pair[idx] = p;
        }
        }
      }
    }
    SGT_SYNC();
    SGT_STAMP(7);
    int nhit = S.icnt[IC_NHIT];
    if (nhit > SGT_MAXHIT) { nhit = SGT_MAXHIT; flags |= SG_FLAG_CONTACTFULL; }
    SGT_PAR(i, nhit) {   // rank by pair index = mj_collision's order
      const int p = S.hit_pair[i];
      int r = 0;
      for (int j = 0; j < nhit; j++) r += S.hit_pair[j] < p ? 1 : 0;
      S.hit_sorted[r] = p;
    }
    SGT_SYNC();
    int unsup = 0;
    SGT_PAR(i, nhit) {   // narrowphase, one lane per hit
      SgGenPair gp = gpairs[S.hit_sorted[i]];
      const bool unsupported = gp.kind == SGP_UNSUPPORTED;
      if (unsupported) gp.kind = sgp_geometry(gp.g1, gp.g2);
      const int i1 = sgg_index(gp.g1), i2 = sgg_index(gp.g2), k1 = sgg_kind(gp.g1), k2 = sgg_kind(gp.g2);
      double* out = stage + (size_t)i * SGT_HITREC * SGT_RECW;
      int n = 0;
      auto put = [&](const ConRec& r, const double* hint) {
        double* o = out + n * SGT_RECW;
        o[0] = r.dist;
        for (int k = 0; k < 3; k++) { o[1 + k] = r.pos[k]; o[4 + k] = r.n[k]; o[7 + k] = hint ? hint[k] : 0.0; }
        n++;
      };
      const int st2 = k2 == SGG_STATIC ? i2 : 0;   // (a plane pair's geom2 is an element or a box: no static geom is named, none is read)
      const double* bp = k2 == SGG_BOX ? S.gpos + 3 * i2 : H.st_pos[st2];
      const double* bm = k2 == SGG_BOX ? S.gmat + 9 * i2 : H.st_mat[st2];
      const double* sz = k2 == SGG_BOX ? T.g_size[i2] : H.st_size[st2];
      const double* cenw = FR ? S.of + OF_CEN : H.center_pos;
      auto elem_axis = [&](int e, double* cax) {   // the capsule's axis in the world (it turns with a free body)
        const double cl[3] = {E(SGE_CX, e), E(SGE_CY, e), E(SGE_CZ, e)};
        if (FR) mulmat3(cax, S.of + OF_R, cl);
        else { cax[0] = cl[0]; cax[1] = cl[1]; cax[2] = cl[2]; }
      };
      if (gp.kind == SGP_PLANE_SPH) {   // oracle collision(), plane - sphere branch
        const double e3[3] = {cenw[0] - H.plane_pos[0], cenw[1] - H.plane_pos[1], cenw[2] - H.plane_pos[2]};
        const double dist = dot3(e3, H.plane_normal) - H.center_radius;
        if (!(dist > H.con_margin)) {
          ConRec r0;
          r0.dist = dist;
          for (int k = 0; k < 3; k++) { r0.pos[k] = cenw[k] - H.plane_normal[k] * (H.center_radius + 0.5 * dist); r0.n[k] = H.plane_normal[k]; }
          put(r0, nullptr);
        }
      } else if (gp.kind == SGP_PLANE_CAP) {
        double c[3], cax[3];
        elem_center(i2, c);
        elem_axis(i2, cax);
        ConRec r0, r1;
        const int m = gen_plane_capsule(H.plane_pos, H.plane_normal, c, cax, H.cap_radius, H.cap_hl, H.con_margin, r0, r1);
        if (m > 0) put(r0, cax);
        if (m > 1) put(r1, cax);
      } else if (gp.kind == SGP_PLANE_BOX) {
        ConRec r[4];
        const int m = gen_plane_box(H.plane_pos, H.plane_normal, bp, bm, sz, H.con_margin, r);
        for (int k = 0; k < m; k++) put(r[k], nullptr);
      } else if (gp.kind == SGP_SPH_BOX) {
        ConRec r0;
        if (sphere_box(cenw, H.center_radius, bp, bm, sz, H.con_margin, r0)) put(r0, nullptr);
      } else if (gp.kind == SGP_CAP_BOX) {
        double c[3], cax[3];
        elem_center(i1, c);
        elem_axis(i1, cax);
        ConRec r0, r1;
        const int m = capsule_box(c, cax, H.cap_radius, H.cap_hl, bp, bm, sz, H.con_margin, r0, r1);
        if (m & 1) put(r0, nullptr);
        if (m & 2) put(r1, nullptr);
      } else if (gp.kind == SGP_BOX_BOX) {
        const double* p1 = k1 == SGG_BOX ? S.gpos + 3 * i1 : H.st_pos[i1];
        const double* R1 = k1 == SGG_BOX ? S.gmat + 9 * i1 : H.st_mat[i1];
        const double* s1 = k1 == SGG_BOX ? T.g_size[i1] : H.st_size[i1];
        ConRec r[8];
        double poly[16][3], tmp[16][3];
        const int m = gen_box_box(p1, R1, s1, bp, bm, sz, H.con_margin, r, poly, tmp);
        for (int k = 0; k < m; k++) put(r[k], nullptr);
      }
      if (unsupported) {
        // A pair whose mixed contact parameters differ from the finger / object pairs' (the plan keeps ONE set) cannot become rows.
        // It must not vanish either: what would be its contact raises the unsupported-pair flag -- data, as on the rows pipeline's
        // general path (sg_phase.hip sg_gen_phase) -- and the host resets the env, instead of a finger passing through the geom.
        for (int k = 0; k < n; k++)
          if (out[k * SGT_RECW] < H.con_margin) unsup = 1;
        n = 0;
      }
      S.hit_cnt[i] = n;
    }
    if (wmax((double)unsup) > 0) flags |= SG_FLAG_UNSUPPORTED_PAIR;
    SGT_SYNC();
    SGT_ONE {
      int off = 0;
      for (int i = 0; i < nhit; i++) { S.hit_off[i] = off; off += S.hit_cnt[i]; }
      S.icnt[IC_NCON] = off;
    }
    SGT_SYNC();
    ncon = S.icnt[IC_NCON];
    if (ncon > SGT_MAXCON) { ncon = SGT_MAXCON; flags |= SG_FLAG_CONTACTFULL; }
    SGT_PAR(i, nhit)
      for (int k = 0; k < S.hit_cnt[i]; k++)
        if (S.hit_off[i] + k < SGT_MAXCON) S.con_src[S.hit_off[i] + k] = i * SGT_HITREC + k;
    SGT_SYNC();

    SGT_STAMP(8);
    }
    if constexpr (PH == 3) {
    // ---------------------------------------------------------------- stage 6: constraint rows
    // (a) equality rows: one joint-fix row per element, the tendon-fix row over all sliders
    double tj_pos = 0, tj_vel = 0, tj_asm = 0, tj_warm = 0, tj_A = 0;
    SGT_PAR(e, N) {
      const double pos = S.qe[e] - E(SGE_QPOS0, e), imp = impedance(H.eqj_solimp, pos, 0.0);
      const double R = fmax(SG_MINVAL, (1 - imp) / imp * E(SGE_INVW, e));
      const double aref = -H.eqj_B * S.ve[e] - H.eqj_K * imp * pos;
      S.Rfix[e] = R; S.bfix[e] = S.asme[e] - aref;
      S.ffix[e] = -(S.we[e] - aref) / R;
      const double co = S.ecoef[e], invm = S.einvm[e];
      double Aee = invm;
      if (FR) {   // the row reaches every slider through the body: [M^-1]_ee = 1/D + B' S^-1 B / D^2; C_e = -S^-1 B_e / D
        double Bs[6];
        mat6vec(Bs, S.of + OF_SINV, S.Be + 6 * e);
        Aee = invm + dot6(S.Be + 6 * e, Bs) * invm * invm;
        if (H.nnb > 0) for (int q = 0; q < 6; q++) S.Ce[6 * e + q] = -Bs[q] * invm;   // (kept for the neighbour-row blocks only: lds_carve)
        S.Afix[e] = Aee + R;
        double* fr4 = S.frow + 4 * e;
        fr4[0] = S.bfix[e]; fr4[1] = R; fr4[2] = Aee + R; fr4[3] = 1.0 / (Aee + R);
      }
      S.Ifix[e] = 1.0 / (Aee + R);
      if (!FR && H.nnb > 0) { double* const fq = S.fixq + 4 * e; fq[0] = S.bfix[e]; fq[1] = R; fq[2] = S.Ifix[e]; fq[3] = invm; }
      tj_pos += co * S.qe[e]; tj_vel += co * S.ve[e]; tj_asm += co * S.asme[e]; tj_warm += co * S.we[e]; tj_A += co * co * invm;
      // (d) limit rows of the slider: slot 0 lower side, slot 1 upper side (MuJoCo's order)
      for (int sd = 0; sd < 2; sd++) {
        const int side = 2 * sd - 1;
        double Rl = 0, bl = 0, fl = 0;   // R = 0 marks an inactive slot
        if (E(SGE_LIMITED, e) != 0.0) {
          const double dist = side * ((sd ? E(SGE_RHI, e) : E(SGE_RLO, e)) - S.qe[e]);
          if (dist < H.lime_margin) {
            const double sg = -side, impl = impedance(H.lime_solimp, dist, H.lime_margin);
            Rl = fmax(SG_MINVAL, (1 - impl) / impl * E(SGE_INVW, e));
            const double arefl = -H.lime_B * sg * S.ve[e] - H.lime_K * impl * (dist - H.lime_margin);
            const double jar = sg * S.we[e] - arefl;
            bl = sg * S.asme[e] - arefl;
            fl = jar < 0 ? -jar / Rl : 0.0;
          }
        }
        S.Rlim[2 * e + sd] = Rl; S.blim[2 * e + sd] = bl; S.flim[2 * e + sd] = fl;
        S.Ilim[2 * e + sd] = 1.0 / (invm + Rl);
      }
    }
    tj_pos = wsum(tj_pos); tj_vel = wsum(tj_vel); tj_asm = wsum(tj_asm); tj_warm = wsum(tj_warm); tj_A = wsum(tj_A);
    double cten[6] = {0, 0, 0, 0, 0, 0};   // free object: the tendon row's push on the body, C_ten = -S^-1 sum_e coef_e B_e / D_e
    if (FR) {
      double Bs[6];
      mat6vec(Bs, S.of + OF_SINV, H.obj_tenB);
      tj_A += dot6(H.obj_tenB, Bs);
      for (int q = 0; q < 6; q++) cten[q] = -Bs[q];
    }
    double ten_R, ten_b, ten_f;
    {
      const double pos = tj_pos - H.t0_L0, imp = impedance(H.eqt_solimp, pos, 0.0);
      ten_R = fmax(SG_MINVAL, (1 - imp) / imp * H.eqt_invw);
      const double aref = -H.eqt_B * tj_vel - H.eqt_K * imp * pos;
      ten_b = tj_asm - aref;
      ten_f = -(tj_warm - aref) / ten_R;
    }
    const double ten_I = 1.0 / (tj_A + ten_R);
#ifdef SGT_X_TAP   // (scripts/repro/tree_mono: the tendon row's intermediates into spare words of S.red, for a word-by-word comparison with the emulation)
    SGT_ONE { S.red[8] = tj_pos; S.red[9] = tj_vel; S.red[10] = tj_asm; S.red[11] = tj_warm; S.red[12] = tj_A; S.red[13] = ten_R; S.red[14] = ten_b; S.red[15] = H.t0_L0; }
#endif
    // (a') the composite's neighbour equalities q_e1 - q0_e1 = q_e2 - q0_e2 (MuJoCo's documented composite): slot
    //      d * N + e = the d-th row registered for element e (its partner: nbtab's out_e2); J = +1 on e, -1 on the partner
    const bool NB = H.nnb > 0;
    if (NB) {
      SGT_SYNC();   // (asme / we of other lanes' elements)
      SGT_PAR(k, 3 * N) {
        const int e = k % N, pe = nbtab[k];
        double R = 0, b = 0, f = 0, I = 0;   // R = 0 marks an empty slot
        if (pe >= 0) {
          const double pos = (S.qe[e] - E(SGE_QPOS0, e)) - (S.qe[pe] - E(SGE_QPOS0, pe)), imp = impedance(H.eqj_solimp, pos, 0.0);
          R = fmax(SG_MINVAL, (1 - imp) / imp * (E(SGE_INVW, e) + E(SGE_INVW, pe)));
          const double aref = -H.eqj_B * (S.ve[e] - S.ve[pe]) - H.eqj_K * imp * pos;
          b = (S.asme[e] - S.asme[pe]) - aref;
          f = -((S.we[e] - S.we[pe]) - aref) / R;
          double Arow = S.einvm[e] + S.einvm[pe];
          if (FR) {   // through the body too: [M^-1]_ee + [M^-1]_pp - 2 [M^-1]_ep, [M^-1]_xy = delta_xy / D_x + B_x' S^-1 B_y / (D_x D_y)
            double Se[6], Sp[6];
            mat6vec(Se, S.of + OF_SINV, S.Be + 6 * e);
            mat6vec(Sp, S.of + OF_SINV, S.Be + 6 * pe);
            const double ie = S.einvm[e], ip = S.einvm[pe];
            Arow += dot6(S.Be + 6 * e, Se) * ie * ie + dot6(S.Be + 6 * pe, Sp) * ip * ip - 2 * dot6(S.Be + 6 * e, Sp) * ie * ip;
            S.nbA[k] = Arow + R;
          }
          I = 1.0 / (Arow + R);
        }
        S.nbR[k] = R; S.nbb[k] = b; S.nbf[k] = f; S.nbI[k] = I;
        if (!FR) { double* const nq = S.nbq + 4 * k; nq[0] = R; nq[1] = b; nq[2] = I; nq[3] = pe >= 0 ? S.einvm[pe] : 0.0; }
      }
    }
    // (c) limit rows of the chain dofs, one lane per chain: compact list in dof order, lower side first
    SGT_PAR(c, K) {
      int n = 0;
      double* rows = S.lrow + SGT_LROW * 2 * T.c_dof0[c];
      for (int dl = 0; dl < T.c_ndof[c]; dl++) {
        const int d = T.c_dof0[c] + dl;
        if (!T.d_limited[d]) continue;
        for (int sd = 0; sd < 2; sd++) {
          const int side = 2 * sd - 1;
          const double dist = side * (T.d_range[d][sd] - S.q[d]);
          if (!(dist < T.d_margin[d])) continue;
          const double sg = -side, imp = impedance(T.d_solimp[d], dist, T.d_margin[d]);
          const double R = fmax(SG_MINVAL, (1 - imp) / imp * T.d_invw[d]);
          const double aref = -T.d_limB[d] * sg * S.v[d] - T.d_limK[d] * imp * (dist - T.d_margin[d]);
          const double jar = sg * S.warm[d] - aref;
          double* r = rows + SGT_LROW * n++;
          r[0] = dl; r[1] = sg; r[2] = R; r[3] = sg * S.asm_[d] - aref; r[4] = jar < 0 ? -jar / R : 0.0;
          r[5] = 1.0 / (S.Minv[c * CS * CS + dl * CS + dl] + R);
        }
      }
      S.icnt[IC_NLIM0 + c] = n;
    }
    SGT_STAMP(9);
    // (e) contact rows, one lane per contact
    SGT_PAR(ci, ncon) {
      const int src = S.con_src[ci], hi = src / SGT_HITREC;
      const double* rec = stage + (size_t)src * SGT_RECW;
      const SgGenPair gp = gpairs[S.hit_sorted[hi]];
      double* J1 = crow(ci);
      double *W1 = J1 + 3 * CS, *J2 = J1 + 6 * CS, *W2 = J1 + 9 * CS, *sc = cscal(ci);
      double fr[9];
      const double hint[3] = {rec[7], rec[8], rec[9]};
      make_frame_hint(rec + 4, (hint[0] != 0 || hint[1] != 0 || hint[2] != 0) ? hint : nullptr, fr);
      // the two sides: geom1's body enters the row with -, geom2's with +
      int ch[2] = {-1, -1}, nd[2] = {0, 0}, sl = -1, touchbit = -1;
      double binvw = 0, Js[3] = {0, 0, 0}, invm = 0;
      bool obj = false, onfree = false;
      int nblk = 0;
      double Jo[3][6] = {{0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}}, fl[9], xl[3];   // free object: frame rows and contact point in the body's frame
      if (FR) {
        const double d3[3] = {rec[1] - S.of[OF_P], rec[2] - S.of[OF_P + 1], rec[3] - S.of[OF_P + 2]};
        mulmatT3(xl, S.of + OF_R, d3);
        for (int rr = 0; rr < 3; rr++) mulmatT3(fl + 3 * rr, S.of + OF_R, fr + 3 * rr);
      }
      auto object_side = [&](double sg) {   // a point of the free body (or of one of its leaves): translation n, rotation x x n
        onfree = true;
        for (int rr = 0; rr < 3; rr++) {
          double xn[3];
          cross3(xn, xl, fl + 3 * rr);
          for (int c = 0; c < 3; c++) { Jo[rr][c] += sg * fl[3 * rr + c]; Jo[rr][3 + c] += sg * xn[c]; }
        }
      };
      for (int side = 0; side < 2; side++) {
        const int ref = side ? gp.g2 : gp.g1, kind = sgg_kind(ref), idx = sgg_index(ref);
        const double sg = side ? 1.0 : -1.0;
        if (kind == SGG_BOX) {
          const int tb = T.g_body[idx], c = T.b_chain[tb], n = T.b_nabove[tb], d0 = T.c_dof0[c];
          binvw += T.b_invw[tb];
          touchbit = idx;
          int blk = -1;
          for (int b = 0; b < nblk; b++)
            if (ch[b] == c) blk = b;
          const bool fresh = blk < 0;
          if (fresh) { blk = nblk++; ch[blk] = c; nd[blk] = 0; }
          double* J = blk ? J2 : J1;
          const int nold = nd[blk];
          if (fresh)
            for (int k = 0; k < 3 * CS; k++) J[k] = 0.0;   // the sweep reads whole padded rows
          for (int dl = 0; dl < (n > nold ? n : nold); dl++) {
            double jp[3] = {0, 0, 0};
            if (dl < n) {
              double r[3];
              for (int k = 0; k < 3; k++) r[k] = rec[1 + k] - S.anchor[3 * (d0 + dl) + k];
              cross3(jp, S.axis + 3 * (d0 + dl), r);
            }
            for (int rr = 0; rr < 3; rr++) {
              const double add = sg * dot3(fr + 3 * rr, jp);
              J[rr * CS + dl] = (dl < nold ? J[rr * CS + dl] : 0.0) + add;
            }
          }
          if (n > nold) nd[blk] = n;
        } else if (kind == SGG_ELEM) {
          sl = idx; obj = true;
          binvw += E(SGE_BINVW, idx);
          invm = 1.0 / (E(SGE_MASS, idx) + E(SGE_ARMATURE, idx));
          const double ax[3] = {E(SGE_AX, idx), E(SGE_AY, idx), E(SGE_AZ, idx)};   // (local to the free body when there is one)
          for (int rr = 0; rr < 3; rr++) Js[rr] += sg * dot3((FR ? fl : fr) + 3 * rr, ax);
          if (FR) object_side(sg);
        } else if (kind == SGG_CENTER) {
          obj = true;
          if (FR) { object_side(sg); binvw += H.free_binvw; }
        }
      }
      // J v, J a_smooth, J a_warmstart (the rows' reference accelerations need them); W = J M^-1 and A = J M^-1 J' + R follow in the next two
      // phases.  What this lane knows about the contact travels in its scalar record (the final A, b, f overwrite the temporaries).
      double vel[3], js[3], jw[3];
      for (int rr = 0; rr < 3; rr++) {
        vel[rr] = sl >= 0 ? Js[rr] * S.ve[sl] : 0.0;
        js[rr] = sl >= 0 ? Js[rr] * S.asme[sl] : 0.0;
        jw[rr] = sl >= 0 ? Js[rr] * S.we[sl] : 0.0;
        if (onfree) { vel[rr] += dot6(Jo[rr], S.of + OF_VL); js[rr] += dot6(Jo[rr], S.of + OF_ASM); jw[rr] += dot6(Jo[rr], S.of + OF_WB); }
      }
      for (int b = 0; b < nblk; b++) {
        const int c = ch[b], d0 = T.c_dof0[c], n = nd[b];
        const double* J = b ? J2 : J1;
        for (int rr = 0; rr < 3; rr++)
          for (int dl = 0; dl < n; dl++) {
            vel[rr] += J[rr * CS + dl] * S.v[d0 + dl];
            js[rr] += J[rr * CS + dl] * S.asm_[d0 + dl];
            jw[rr] += J[rr * CS + dl] * S.warm[d0 + dl];
          }
      }
      for (int k = 0; k < 3; k++) { sc[CS_TMP + k] = vel[k]; sc[CS_TMP + 3 + k] = js[k]; sc[CS_TMP + 6 + k] = jw[k]; sc[CS_JS + k] = Js[k]; }
      sc[CS_TMP + 9] = binvw; sc[CS_TMP + 10] = nblk; sc[CS_TMP + 11] = rec[0];
      sc[CS_INVM] = invm; sc[CS_SL] = sl;
      sc[CS_C1] = ch[0]; sc[CS_N1] = nd[0]; sc[CS_C2] = ch[1]; sc[CS_N2] = nd[1];
      sc[CS_TOUCH] = (obj && touchbit >= 0) ? touchbit : -1;
      sc[CS_OBJ] = onfree ? 1.0 : 0.0;
      for (int rr = 0; rr < 3; rr++)
        for (int q = 0; q < 6; q++) sc[CS_JO + 6 * rr + q] = Jo[rr][q];
    }
    SGT_SYNC();
    // (e2) W = J M^-1, one lane per WORD of a W row (r04: contact, chain block, row, dof -- 780 items for 13 contacts; a contact's own lane
    //      used to run the 3 x CS x n products alone, out of an LDS copy of M^-1: with M^-1 in the work space the lanes of a row read
    //      consecutive words of its rows)
    SGT_PAR(i, ncon * 6 * CS) {
      const int ci = i / (6 * CS), rem = i % (6 * CS), b = rem / (3 * CS), rr = (rem / CS) % 3, dl = rem % CS;
      const double* sc = cscal(ci);
      const int c = (int)sc[b ? CS_C2 : CS_C1];
      if (c >= 0) {
        const double* J = crow(ci) + (b ? 6 * CS : 0) + rr * CS;
        const double* Mi = S.Minv + c * CS * CS;
        double s = 0;
#pragma unroll
        for (int e2 = 0; e2 < CS; e2++) s += J[e2] * Mi[e2 * CS + dl];   // (J is zero beyond the body's dofs: the same sum as over them)
        crow(ci)[(b ? 9 * CS : 3 * CS) + rr * CS + dl] = s;
      }
    }
    SGT_SYNC();
    // (e3) A, the reference accelerations, the warmstart force: the contact's lane again
    SGT_PAR(ci, ncon) {
      double* J1 = crow(ci);
      double *W1 = J1 + 3 * CS, *J2 = J1 + 6 * CS, *W2 = J1 + 9 * CS, *sc = cscal(ci);
      int ch[2] = {(int)sc[CS_C1], (int)sc[CS_C2]}, nd[2] = {(int)sc[CS_N1], (int)sc[CS_N2]};
      const int sl = (int)sc[CS_SL], nblk = (int)sc[CS_TMP + 10];
      const bool onfree = sc[CS_OBJ] != 0.0;
      const double invm = sc[CS_INVM], binvw = sc[CS_TMP + 9];
      double Js[3], vel[3], js[3], jw[3], Jo[3][6];
      for (int k = 0; k < 3; k++) { Js[k] = sc[CS_JS + k]; vel[k] = sc[CS_TMP + k]; js[k] = sc[CS_TMP + 3 + k]; jw[k] = sc[CS_TMP + 6 + k]; }
      for (int rr = 0; rr < 3; rr++)
        for (int q = 0; q < 6; q++) Jo[rr][q] = sc[CS_JO + 6 * rr + q];
      const double rec0 = sc[CS_TMP + 11];
      const double* rec = &rec0;
      double Am[6] = {0, 0, 0, 0, 0, 0};
      for (int b = 0; b < nblk; b++) {
        const int n = nd[b];
        const double* J = b ? J2 : J1;
        const double* W = b ? W2 : W1;
        int k = 0;
        for (int rr = 0; rr < 3; rr++)
          for (int s2 = rr; s2 < 3; s2++) {
            double s = 0;
            for (int dl = 0; dl < n; dl++) s += W[rr * CS + dl] * J[s2 * CS + dl];
            Am[k++] += s;
          }
      }
      const double dist = rec[0], imp = impedance(H.con_solimp, dist, H.con_margin);
      const double R = fmax(SG_MINVAL, (1 - imp) / imp * binvw), D = 1 / R;
      if (onfree) {   // the object's share of J M^-1 J' through the arrow matrix: y = M^-1 J_r' = (y_f ; y_e)
        double yf[3][6], ye[3];
        for (int rr = 0; rr < 3; rr++) {
          double t6[6];
          for (int q = 0; q < 6; q++) t6[q] = Jo[rr][q] - (sl >= 0 ? S.Be[6 * sl + q] * Js[rr] * invm : 0.0);
          mat6vec(yf[rr], S.of + OF_SINV, t6);
          ye[rr] = sl >= 0 ? (Js[rr] - dot6(S.Be + 6 * sl, yf[rr])) * invm : 0.0;
        }
        int k = 0;
        for (int rr = 0; rr < 3; rr++)
          for (int s2 = rr; s2 < 3; s2++) { Am[k] += dot6(Jo[s2], yf[rr]) + Js[s2] * ye[rr] + (rr == s2 ? R : 0.0); k++; }
      } else {
        int k = 0;
        for (int rr = 0; rr < 3; rr++)
          for (int s2 = rr; s2 < 3; s2++) { Am[k] += Js[rr] * Js[s2] * invm + (rr == s2 ? R : 0.0); k++; }
      }
      double bb[3], jar[3];
      for (int rr = 0; rr < 3; rr++) {
        const double aref = -H.con_B * vel[rr] - (rr == 0 ? H.con_K * imp * (dist - H.con_margin) : 0.0);
        bb[rr] = js[rr] - aref;
        jar[rr] = jw[rr] - aref;
      }
      double f[3];
      {  // warmstart force: primal -> dual map of the elliptic cone (mj_constraintUpdate)
        const double mu = H.con_mu[0], U0 = jar[0] * mu, U1 = jar[1] * H.con_mu[0], U2 = jar[2] * H.con_mu[1];
        const double Nn = U0, Tt = sqrt(U1 * U1 + U2 * U2);
        if (Nn >= mu * Tt || (Tt <= 0 && Nn >= 0)) { f[0] = f[1] = f[2] = 0; }
        else if (mu * Nn + Tt <= 0 || (Tt <= 0 && Nn < 0)) { for (int rr = 0; rr < 3; rr++) f[rr] = -D * jar[rr]; }
        else {
          const double Dm = D / (mu * mu * (1 + mu * mu)), NmT = Nn - mu * Tt;
          f[0] = -Dm * NmT * mu;
          f[1] = -f[0] / Tt * U1 * H.con_mu[0];
          f[2] = -f[0] / Tt * U2 * H.con_mu[1];
        }
      }
      const bool rows = dist < H.con_margin;   // mj_makeConstraint: a contact at dist >= margin - gap is listed but gets no rows
      for (int k = 0; k < 6; k++) sc[CS_A + k] = Am[k];
      for (int k = 0; k < 3; k++) { sc[CS_B + k] = bb[k]; sc[CS_F0 + k] = rows ? f[k] : 0.0; S.cf[3 * ci + k] = rows ? f[k] : 0.0; }
      sc[CS_R] = R;
      sc[CS_ROWS] = rows ? 1.0 : 0.0;
      {
        const double mu2[2] = {H.con_mu[0], H.con_mu[1]};
        double pe[7];
        contact_block_constants(Am, mu2, pe);
        for (int k = 0; k < 7; k++) sc[CS_PE + k] = pe[k];   // (every temporary of this record has been read above)
      }
      // the contact's stream in the sweep: its one chain; -1 = no rows; -2 = not exactly one chain block (both fingers, or a slider
      // against a static geom): such a list is swept serially
      S.con_chain[ci] = !rows ? -1 : ((nblk == 1 && !onfree) ? ch[0] : -2);   // (a free object couples every contact on it: serial list)
    }
    SGT_ONE { S.icnt[IC_SERIAL] = 0; }
    SGT_SYNC();
    {
      const int ncc = ncon < S.ncache ? ncon : S.ncache;
      SGT_PAR(i, ncc * SGT_CSC) S.csc[i] = cscal(i / SGT_CSC)[i % SGT_CSC];
    }
    // THE CONTACTS' LEVEL SCHEDULE (r05).  Two contacts commute exactly unless they share a dof: the same chain, or the same slider
    // (two fingers on one capsule).  Level(i) = 1 + the highest level of an EARLIER contact that shares a dof with i: contacts of one level are
    // mutually independent, and every pair that does not commute keeps mj_solPGS's order -- the levels in sequence ARE the
    // sequential sweep.  A level holds at most one contact per chain: the sweep runs it on the chains' lane groups side by side.
    // (Until r04 a slider under two fingers sent the WHOLE list to the one-after-the-other fallback: the four-finger scene at
    // the squeeze -- 13 contacts, always a shared capsule somewhere -- ran 13 serial updates per sweep with two barriers each, 55 % of
    // a substep; its levels: 4 - 5.)  Table S.hit_pair[level][chain] = contact id or -1 (the pair walk's hit list is done with); a
    // contact with two chain blocks, without one, or on a free object still makes the list serial, as does a table overflow.
    SGT_PAR(ci, ncon) {
      if (S.con_chain[ci] == -2) S.icnt[IC_SERIAL] = 1;
      S.hit_sorted[ci] = (int)cscal(ci)[CS_SL];      // (staged for the one lane that builds the schedule: a word of LDS instead of a trip to the work space per contact)
    }
    SGT_PAR(i, SGT_MAXHIT) S.hit_pair[i] = -1;
    SGT_PAR(e, N) S.hit_off[e] = 0;                  // last level + 1 of slider e (N <= SGT_MAXHIT: four elements a lane)
    SGT_PAR(c, K) S.hit_sorted[SGT_MAXCON + c] = 0;  // ... of chain c
    SGT_SYNC();
    SGT_ONE {
      int nlev = 0;
      if (S.icnt[IC_SERIAL] == 0) {
        const int cap = SGT_MAXHIT / K;
        for (int ci = 0; ci < ncon; ci++) {
          const int c = S.con_chain[ci];
          if (c < 0) continue;
          const int sl = S.hit_sorted[ci];
          int L = S.hit_sorted[SGT_MAXCON + c];
          if (sl >= 0 && S.hit_off[sl] > L) L = S.hit_off[sl];
          if (L >= cap) { S.icnt[IC_SERIAL] = 1; break; }
          S.hit_pair[L * K + c] = ci;
          S.hit_sorted[SGT_MAXCON + c] = L + 1;
          if (sl >= 0) S.hit_off[sl] = L + 1;
          nlev = nlev > L + 1 ? nlev : L + 1;
        }
      }
      S.icnt[IC_NLEV] = nlev;
    }
    SGT_SYNC();
    const bool serial_contacts = S.icnt[IC_SERIAL] != 0;
#if SGT_DEVICE
    // the serial list's wave-synchronous pass reads a contact's chains from LDS (S.hit_cnt: the narrowphase counts are done with)
    if (serial_contacts) {
      SGT_PAR(ci, ncon) S.hit_cnt[ci] = (((int)cscal(ci)[CS_C1] + 1) & 0xff) | ((((int)cscal(ci)[CS_C2] + 1) & 0xff) << 8);
      SGT_SYNC();
    }
#endif
    const double con_mu[2] = {H.con_mu[0], H.con_mu[1]};
    SGT_STAMP(10);
    // row count (nefc) and the touch bits of this contact list
    {
      int nl = 0;
      for (int c = 0; c < K; c++) nl += S.icnt[IC_NLIM0 + c];
      double cnt = 0;
      SGT_PAR(e, N) cnt += (S.Rlim[2 * e] != 0.0 ? 1.0 : 0.0) + (S.Rlim[2 * e + 1] != 0.0 ? 1.0 : 0.0);
      SGT_PAR(ci, ncon) cnt += cscal(ci)[CS_ROWS] != 0.0 ? 3.0 : 0.0;
      nefc = N + 1 + nl + H.nnb + (int)wsum(cnt);
      touch_lo = touch_hi = 0;
      for (int ci = 0; ci < ncon; ci++) {   // uniform loop: every lane ends up with the same words
        const int tbit = (int)cscal(ci)[CS_TOUCH];
        if (tbit >= 0 && tbit < 32) touch_lo |= 1u << tbit;
        else if (tbit >= 32 && tbit < 64) touch_hi |= 1u << (tbit - 32);
      }
    }

    // ---------------------------------------------------------------- stage 10: warmstart (kept only if it beats f = 0), PGS
    // a = M^-1 J' f of the current forces: chains in aF, sliders in ae
    auto apply_all = [&]() {
      SGT_PAR(e, N) {
        double g = S.ffix[e] + S.ecoef[e] * ten_f + S.flim[2 * e] - S.flim[2 * e + 1];
        if (NB)
          for (int d = 0; d < 3; d++) {   // its own rows push it with +f, the rows that have it as partner (nbtab's in_slot) with -f
            if (nbtab[d * N + e] >= 0) g += S.nbf[d * N + e];
            const int in = nbtab[6 * N + d * N + e];
            if (in >= 0) g -= S.nbf[in];
          }
        S.ae[e] = S.einvm[e] * g;
      }
      SGT_PAR(idx, K * CS) {
        const int c = idx / CS, dl = idx % CS;
        const double* rows = S.lrow + SGT_LROW * 2 * T.c_dof0[c];
        double s = 0;
        // (r05s: eight rows' words of M^-1 -- work space, behind the row's dof index from LDS -- requested together, then added in the rows'
        //  order: row by row the loop paid a trip to the work space per row, 14 in a row for a finger of the four-finger gripper)
        const int nl = S.icnt[IC_NLIM0 + c];
        for (int i0 = 0; i0 < nl; i0 += 8) {
          double mw[8];
          for (int k = 0; k < 8; k++) {
            const int ii = i0 + k < nl ? i0 + k : 0;   // (past the list: row 0's word, read and not used -- the dof index must be a valid one)
            mw[k] = S.Minv[c * CS * CS + (int)rows[SGT_LROW * ii] * CS + dl];
          }
          for (int k = 0; k < 8; k++) {
            const int i = i0 + k;
            if (i < nl) s += mw[k] * rows[SGT_LROW * i + 1] * rows[SGT_LROW * i + 4];
          }
        }
        S.aF[idx] = s;
      }
      SGT_SYNC();
      // The contacts add their pushes in the list's order (a slider or a chain may carry several).  Every chain word has its lane, which walks
      // the list and adds what is its own -- the same sums in the same order as contact after contact between barriers (two per contact
      // until r05: 4 % of a free-ball substep), without a barrier; the sliders' and the object's words go through one lane meanwhile.
#if SGT_DEVICE && !defined(SGT_X_WSSERIAL)
      // (r05s) a list of at most 64 contacts: a contact's chains come from the lane that holds
      // its record (scalar reads: no trip to the work space per contact and lane), and the next contact's W words are requested while the
      // current one's are added -- the same sums in the same order
      if (ncon <= 64) {
        const int lane = (int)threadIdx.x, cl = lane < ncon ? lane : 0;
        const double* scl = cscr(cl);
        const bool rows_l = lane < ncon && scl[CS_ROWS] != 0.0;
        const int c1_l = rows_l ? (int)scl[CS_C1] : -1, c2_l = rows_l ? (int)scl[CS_C2] : -1;
        for (int w0 = 0; w0 < K * CS; w0 += 64) {   // (the four-finger gripper's 80 chain words: two passes)
        const int wi = w0 + lane;
        const bool word = wi < K * CS;
        const int c = word ? wi / CS : -2, dl = word ? wi % CS : 0;   // (-2: a lane without a word matches no chain)
        double a = word ? S.aF[wi] : 0.0;
        struct WR { double w0, w1, w2, w3, w4, w5; bool m1, m2; };
        auto ldw = [&](WR& q, const int ci) {   // ci uniform
          q.m1 = __builtin_amdgcn_readlane(c1_l, ci) == c; q.m2 = __builtin_amdgcn_readlane(c2_l, ci) == c;
          const double* W1 = crow(ci) + 3 * CS + dl;
          const double* W2 = crow(ci) + 9 * CS + dl;
          q.w0 = q.w1 = q.w2 = q.w3 = q.w4 = q.w5 = 0.0;
          if (q.m1) { q.w0 = W1[0]; q.w1 = W1[CS]; q.w2 = W1[2 * CS]; }
          if (q.m2) { q.w3 = W2[0]; q.w4 = W2[CS]; q.w5 = W2[2 * CS]; }
        };
        auto acc = [&](const WR& q, const int ci) {
          const double* f = S.cf + 3 * ci;
          if (q.m1) a += q.w0 * f[0] + q.w1 * f[1] + q.w2 * f[2];
          if (q.m2) a += q.w3 * f[0] + q.w4 * f[1] + q.w5 * f[2];
        };
        if (ncon > 0) {
          WR qa, qb;
          ldw(qa, 0);
          for (int ci = 0; ci < ncon; ci += 2) {
            ldw(qb, ci + 1 < ncon ? ci + 1 : 0);
            acc(qa, ci);
            ldw(qa, ci + 2 < ncon ? ci + 2 : 0);
            if (ci + 1 < ncon) acc(qb, ci + 1);
          }
        }
        if (word) S.aF[wi] = a;
        }
      } else
#endif
      SGT_PAR(idx, K * CS) {
        const int c = idx / CS, dl = idx % CS;
        double a = S.aF[idx];
        for (int ci = 0; ci < ncon; ci++) {
          const double* sc = cscr(ci);
          if (sc[CS_ROWS] == 0.0) continue;
          const int c1 = (int)sc[CS_C1], c2 = (int)sc[CS_C2];
          if (c1 != c && c2 != c) continue;
          const double* f = S.cf + 3 * ci;
          const double* W1 = crow(ci) + 3 * CS;
          const double* W2 = crow(ci) + 9 * CS;
          if (c1 == c) a += W1[dl] * f[0] + W1[CS + dl] * f[1] + W1[2 * CS + dl] * f[2];
          if (c2 == c) a += W2[dl] * f[0] + W2[CS + dl] * f[1] + W2[2 * CS + dl] * f[2];
        }
        S.aF[idx] = a;
      }
#if SGT_DEVICE && !defined(SGT_X_WSSERIAL)
      // (r05s) the sliders' and the free body's words: every contact's TERMS on a lane of their own (its record's loads side by side with the
      // other contacts'), then the sums in the list's order by scalar reads of the lanes -- the same terms added in the same order as the
      // one-lane walk below, which paid a record's round trip to the work space per contact: ~1.5 us each, 3 % of a free-ball substep
      if (ncon <= 64) {
        const int lane = (int)threadIdx.x, cl = lane < ncon ? lane : 0;
        const double* sc = cscr(cl);
        const bool rows = lane < ncon && sc[CS_ROWS] != 0.0;
        const double* f = S.cf + 3 * cl;
        const double f0 = f[0], f1 = f[1], f2 = f[2];
        const int sl = rows ? (int)sc[CS_SL] : -1;
        const double ts = sl >= 0 ? sc[CS_INVM] * (sc[CS_JS] * f0 + sc[CS_JS + 1] * f1 + sc[CS_JS + 2] * f2) : 0.0;
        const int ob = (FR && rows && sc[CS_OBJ] != 0.0) ? 1 : 0;
        double tq[6] = {0, 0, 0, 0, 0, 0};
        if (FR && ob)
          for (int q = 0; q < 6; q++) tq[q] = sc[CS_JO + q] * f0 + sc[CS_JO + 6 + q] * f1 + sc[CS_JO + 12 + q] * f2;
        double gf[6] = {0, 0, 0, 0, 0, 0};
        if (FR)
          for (int q = 0; q < 6; q++) gf[q] = S.of[OF_GF + q];
        for (int c = 0; c < ncon; c++) {   // (uniform)
          const int slc = __builtin_amdgcn_readlane(sl, c);
          if (slc >= 0) {
            const double t = readlane64(ts, c);
            SGT_ONE { S.ae[slc] += t; }
          }
          if (FR && __builtin_amdgcn_readlane(ob, c))
            for (int q = 0; q < 6; q++) gf[q] += readlane64(tq[q], c);
        }
        if (FR) SGT_ONE { for (int q = 0; q < 6; q++) S.of[OF_GF + q] = gf[q]; }
      } else
#endif
      SGT_ONE {
        for (int ci = 0; ci < ncon; ci++) {
          const double* sc = cscr(ci);
          if (sc[CS_ROWS] == 0.0) continue;
          const double* f = S.cf + 3 * ci;
          const int sl = (int)sc[CS_SL];
          if (sl >= 0) S.ae[sl] += sc[CS_INVM] * (sc[CS_JS] * f[0] + sc[CS_JS + 1] * f[1] + sc[CS_JS + 2] * f[2]);
          if (FR && sc[CS_OBJ] != 0.0)
            for (int q = 0; q < 6; q++) S.of[OF_GF + q] += sc[CS_JO + q] * f[0] + sc[CS_JO + 6 + q] * f[1] + sc[CS_JO + 12 + q] * f[2];
        }
      }
      SGT_SYNC();
      if (FR) {   // S.ae holds the sliders' LOCAL part g_e / D_e; the body: a_f = S^-1 (g_f - sum_e B_e g_e / D_e)
        double red[6] = {0, 0, 0, 0, 0, 0}, rhs[6], af6[6];
        SGT_PAR(e, N)
          for (int q = 0; q < 6; q++) red[q] += S.Be[6 * e + q] * S.ae[e];
        for (int q = 0; q < 6; q++) rhs[q] = S.of[OF_GF + q] - wsum(red[q]);
        mat6vec(af6, S.of + OF_SINV, rhs);
        SGT_SYNC();
        SGT_ONE { for (int q = 0; q < 6; q++) S.of[OF_AF + q] = af6[q]; }
        SGT_SYNC();
      }
    };
    if (FR) { SGT_ONE { for (int q = 0; q < 6; q++) S.of[OF_GF + q] = 0; } }
    apply_all();
    {
      double cost = 0, S_ae = 0;
      SGT_PAR(e, N) {
        const double ae_ = slider_acc(e);
        S_ae += S.ecoef[e] * ae_;
        cost += S.ffix[e] * (0.5 * (ae_ + S.Rfix[e] * S.ffix[e]) + S.bfix[e]);
        cost += S.flim[2 * e] * (0.5 * (ae_ + S.Rlim[2 * e] * S.flim[2 * e]) + S.blim[2 * e]);
        cost += S.flim[2 * e + 1] * (0.5 * (-ae_ + S.Rlim[2 * e + 1] * S.flim[2 * e + 1]) + S.blim[2 * e + 1]);
      }
      if (NB) SGT_PAR(k, 3 * N) {
        const int pe = nbtab[k];
        if (pe >= 0) cost += S.nbf[k] * (0.5 * ((slider_acc(k % N) - slider_acc(pe)) + S.nbR[k] * S.nbf[k]) + S.nbb[k]);
      }
      SGT_PAR(c, K) {
        const double* rows = S.lrow + SGT_LROW * 2 * T.c_dof0[c];
        for (int i = 0; i < S.icnt[IC_NLIM0 + c]; i++) {
          const double* r = rows + SGT_LROW * i;
          cost += r[4] * (0.5 * (r[1] * S.aF[c * CS + (int)r[0]] + r[2] * r[4]) + r[3]);
        }
      }
      SGT_PAR(ci, ncon) {
        const double* sc = cscr(ci);
        if (sc[CS_ROWS] == 0.0) continue;
        const double* f = S.cf + 3 * ci;
        const int sl = (int)sc[CS_SL];
        for (int rr = 0; rr < 3; rr++) {
          double ja = sl >= 0 ? sc[CS_JS + rr] * slider_acc(sl) : 0.0;
          if (FR && sc[CS_OBJ] != 0.0) ja += dot6(sc + CS_JO + 6 * rr, S.of + OF_AF);
          for (int b = 0; b < 2; b++) {
            const int c = (int)sc[b ? CS_C2 : CS_C1], n = (int)sc[b ? CS_N2 : CS_N1];
            if (c < 0) continue;
            const double* J = crow(ci) + (b ? 6 * CS : 0) + rr * CS;
            for (int dl = 0; dl < n; dl++) ja += J[dl] * S.aF[c * CS + dl];
          }
          cost += f[rr] * (0.5 * (ja + sc[CS_R] * f[rr]) + sc[CS_B + rr]);
        }
      }
      S_ae = wsum(S_ae);
      cost = wsum(cost) + ten_f * (0.5 * (S_ae + ten_R * ten_f) + ten_b);
      if (cost > 0) {   // uniform
        ten_f = 0;
        SGT_PAR(e, N) { S.ffix[e] = 0; S.flim[2 * e] = 0; S.flim[2 * e + 1] = 0; S.ae[e] = 0; }
        if (NB) SGT_PAR(k, 3 * N) S.nbf[k] = 0;
        SGT_PAR(i, K * CS) S.aF[i] = 0;
        if (FR) SGT_ONE { for (int q = 0; q < 6; q++) S.of[OF_AF + q] = S.of[OF_GF + q] = 0; }
        SGT_PAR(c, K)
          for (int i = 0; i < S.icnt[IC_NLIM0 + c]; i++) S.lrow[SGT_LROW * (2 * T.c_dof0[c] + i) + 4] = 0;
        SGT_PAR(i, 3 * ncon) S.cf[i] = 0;
      }
      SGT_SYNC();
    }
    iters = 0;
    SGT_STAMP(11);
    // the PGS sweeps run in a function of their own (tree_sweep, above tree_env): its register allocation is not the monolith's --
    // the step's ~40 stages in one function left the sweep's loops 249 spilled registers and 2.3 KB of scratch memory per lane
    SGT_ONE {
      double* w = S.swc;
      w[SWC_TEN_R] = ten_R; w[SWC_TEN_B] = ten_b; w[SWC_TEN_F] = ten_f; w[SWC_TJ_A] = tj_A; w[SWC_TEN_I] = ten_I;
      for (int q = 0; q < 6; q++) w[SWC_CTEN + q] = cten[q];
      w[SWC_NCON] = ncon; w[SWC_SERIAL] = serial_contacts ? 1.0 : 0.0;
    }
    SGT_SYNC();
    {   // (uniform branches: one instantiation of the sweep per scene class)
      const SGT_CONST SgPlanHeader* const hp = (const SGT_CONST SgPlanHeader*)A.H;
      const SGT_CONST SgTreeDev* const tp = (const SGT_CONST SgTreeDev*)A.T;
      const SGT_CONST int* const nbc = (const SGT_CONST int*)A.nbtab;
      const SGT_CONST SgEqSlot* const sc = (const SGT_CONST SgEqSlot*)A.sched;
      SGT_GLOBP double* const cwp = (SGT_GLOBP double*)(A.cws + (size_t)env * A.cws_stride);
      SGT_LDSP double* const lp = (SGT_LDSP double*)lds_base;
      if (FR && NB) tree_sweep<CHD, true, true>(hp, tp, nbc, sc, A.nbtab, cwp, lp, A.secprof);
      else if (FR) tree_sweep<CHD, true, false>(hp, tp, nbc, sc, A.nbtab, cwp, lp, A.secprof);
      else if (NB) tree_sweep<CHD, false, true>(hp, tp, nbc, sc, A.nbtab, cwp, lp, A.secprof);
      else tree_sweep<CHD, false, false>(hp, tp, nbc, sc, A.nbtab, cwp, lp, A.secprof);
    }
    SGT_SYNC();
    SGT_STAMP_RESET();
    iters = (int)S.swc[SWC_ITERS];

    SGT_STAMP(14);
    }
    if constexpr (PH == 4) {
    // ---------------------------------------------------------------- qacc, qfrc_constraint, warmstart, sensors
    SGT_PAR(d, ND) {
      const int c = T.d_chain[d], dl = d - T.c_dof0[c];
      double s = 0;
      const double* rows = S.lrow + SGT_LROW * 2 * T.c_dof0[c];
      for (int i = 0; i < S.icnt[IC_NLIM0 + c]; i++)
        if ((int)rows[SGT_LROW * i] == dl) s += rows[SGT_LROW * i + 1] * rows[SGT_LROW * i + 4];
      for (int ci = 0; ci < ncon; ci++) {
        const double* sc = cscr(ci);
        if (sc[CS_ROWS] == 0.0) continue;
        for (int b = 0; b < 2; b++)
          if ((int)sc[b ? CS_C2 : CS_C1] == c && dl < (int)sc[b ? CS_N2 : CS_N1]) {
            const double* J = crow(ci) + (b ? 6 * CS : 0);
            s += J[dl] * S.cf[3 * ci] + J[CS + dl] * S.cf[3 * ci + 1] + J[2 * CS + dl] * S.cf[3 * ci + 2];
          }
      }
      S.fc[d] = s;
      S.qacc[d] = S.asm_[d] + S.aF[pidx(d)];
      S.warm[d] = S.qacc[d];
    }
    double badacc = 0;
    SGT_PAR(d, ND) badacc += isbad(S.qacc[d]) ? 1.0 : 0.0;
    SGT_PAR(e, N) {
      const double qa = S.asme[e] + slider_acc(e);
      S.we[e] = qa;
      badacc += isbad(qa) ? 1.0 : 0.0;
    }
    if (FR) {   // the body: qacc in dof coordinates (translations along the world axes) is what the next solve warmstarts from
      double qf[6], tw[3];
      for (int q = 0; q < 6; q++) { qf[q] = S.of[OF_ASM + q] + S.of[OF_AF + q]; badacc += isbad(qf[q]) ? 1.0 / 64 : 0.0; }
      mulmat3(tw, S.of + OF_R, qf);
      SGT_SYNC();
      SGT_ONE { for (int q = 0; q < 3; q++) { S.of[OF_WARM + q] = tw[q]; S.of[OF_WARM + 3 + q] = qf[3 + q]; } }
    }
    badacc = wsum(badacc);
    SGT_SYNC();
    if (last && A.sens) {   // sensordata of the call = that of the last forward pass
      tree_motion(S.qacc);
      SGT_SYNC();
      SGT_PAR(i, T.NSENS) {
        const int s = T.sn_site[i], tb = T.s_body[s];
        double sm[9], out[3];
        mulmat33(sm, S.xmat + 9 * tb, T.s_mat[s]);
        if (T.sn_type[i] == SG_SENS_GYRO) {
          mulmatT3(out, sm, S.bw + 3 * tb);
        } else {
          const double *w = S.bw + 3 * tb, *al = S.bal + 3 * tb;
          double r[3], a[3], t[3], t2[3];
          for (int k = 0; k < 3; k++) { r[k] = S.spos[3 * s + k] - S.xpos[3 * tb + k]; a[k] = S.ba[3 * tb + k]; }
          cross3(t, al, r); addscl3(a, t, 1);
          cross3(t, w, r); cross3(t2, w, t); addscl3(a, t2, 1);
          mulmatT3(out, sm, a);
        }
        double* so = A.sens + (size_t)env * A.sens_stride + T.sn_adr[i];
        so[0] = out[0]; so[1] = out[1]; so[2] = out[2];
      }
    }
    if (badacc > 0) { flags |= SG_FLAG_BADQACC; stop = 1; goto stage_done; }
    if (!integrate) goto stage_done;
    SGT_STAMP(15);
    // ---------------------------------------------------------------- stage 12: Euler with implicit joint damping
    SGT_PAR(i, T.NMAT) {
      const int c = i / (CS * CS), a = (i % (CS * CS)) / CS, b = i % CS;
      S.L[i] = Mg[i] + ((a == b && a < T.c_ndof[c]) ? h * T.d_damping[T.c_dof0[c] + a] : 0.0);
    }
    SGT_PAR(i, K * CS) S.tmpP[i] = 0;
    SGT_SYNC();
    SGT_PAR(d, ND) S.tmpP[pidx(d)] = S.fs[d] + S.fc[d];   // right-hand side, padded
    factor_all();
    SGT_PAR(c, K) {
      chain_solve_reg<CHD>(S.L + c * CS * CS, CS, S.tmpP + c * CS);
      double* cs = S.chs + c * CHS_N;
      cs[CHS_ACT] += h * cs[CHS_ACTDOT];
    }
    double Jx = 0, Jy = 0;
    if (FR) {
      // (M + h B) x = f for the arrow matrix: D' = D + h d, S' = M_ff - sum B B' / D' (the sum is a plan constant); the same solve for
      // y = (M + h B)^-1 J' of the volume tendon (D5: Sherman-Morrison on top of it)
      double Sh[36], Shi[36], rf[6] = {0, 0, 0, 0, 0, 0}, xf[6], yf[6], t6[6];
      {
        int qq = 0;
        for (int r = 0; r < 6; r++)
          for (int q = r; q < 6; q++) { Sh[6 * r + q] = Sh[6 * q + r] = S.of[OF_MFF + qq] - H.obj_BBDh[qq]; qq++; }
      }
      spd_inverse6(Sh, Shi);
      SGT_PAR(e, N) {
        const double D = E(SGE_MASS, e) + E(SGE_ARMATURE, e), den = D + h * E(SGE_DAMPING, e), r = (S.fse[e] + D * S.ae[e]) / den;   // (g_e = D x the local part)
        for (int q = 0; q < 6; q++) rf[q] += S.Be[6 * e + q] * r;
      }
      for (int q = 0; q < 6; q++) t6[q] = -S.of[OF_BIAS + q] + S.of[OF_GF + q] - wsum(rf[q]);
      mat6vec(xf, Shi, t6);
      for (int q = 0; q < 6; q++) t6[q] = -H.obj_tenBh[q];
      mat6vec(yf, Shi, t6);
      SGT_PAR(e, N) {
        const double D = E(SGE_MASS, e) + E(SGE_ARMATURE, e), den = D + h * E(SGE_DAMPING, e), co = S.ecoef[e];
        const double x = (S.fse[e] + D * S.ae[e] - dot6(S.Be + 6 * e, xf)) / den, y = (co - dot6(S.Be + 6 * e, yf)) / den;
        S.asme[e] = x; S.Ifix[e] = y;   // (both arrays are rebuilt by the next forward pass)
        Jx += co * x; Jy += co * y;
      }
      Jx = wsum(Jx); Jy = wsum(Jy);
      const double kf = H.t0_implicit ? h * H.t0_damping * Jx / (1 + h * H.t0_damping * Jy) : 0.0;
      SGT_SYNC();
      SGT_PAR(e, N) {
        S.ve[e] += h * (S.asme[e] - kf * S.Ifix[e]);
        S.qe[e] += h * S.ve[e];
      }
      SGT_ONE {   // the body: velocities (world translations, body-frame rotations), then mj_integratePos with the new velocity
        double* o = S.of;
        double xw[3], xb[3] = {xf[0] - kf * yf[0], xf[1] - kf * yf[1], xf[2] - kf * yf[2]};
        mulmat3(xw, o + OF_R, xb);
        for (int q = 0; q < 3; q++) {
          o[OF_VW + q] += h * xw[q];
          o[OF_WL + q] += h * (xf[3 + q] - kf * yf[3 + q]);
          o[OF_P + q] += h * o[OF_VW + q];
        }
        const double* wl = o + OF_WL;
        const double nw = sqrt(dot3(wl, wl)), ang = h * nw;
        if (nw > SG_MINVAL) {   // mju_quatIntegrate: q <- q * (cos, axis sin), the axis in the body frame
          const double sn = sin(0.5 * ang), qr[4] = {cos(0.5 * ang), wl[0] / nw * sn, wl[1] / nw * sn, wl[2] / nw * sn};
          const double nq0 = sqrt(o[OF_Q] * o[OF_Q] + o[OF_Q + 1] * o[OF_Q + 1] + o[OF_Q + 2] * o[OF_Q + 2] + o[OF_Q + 3] * o[OF_Q + 3]);
          (void)nq0;
          quatmul(o + OF_Q, o + OF_Q, qr);
          const double nq = sqrt(o[OF_Q] * o[OF_Q] + o[OF_Q + 1] * o[OF_Q + 1] + o[OF_Q + 2] * o[OF_Q + 2] + o[OF_Q + 3] * o[OF_Q + 3]);
          for (int q = 0; q < 4; q++) o[OF_Q + q] /= nq;
        }
      }
    } else {
      SGT_PAR(e, N) {
        const double m = E(SGE_MASS, e) + E(SGE_ARMATURE, e), fce = m * S.ae[e], den = m + h * E(SGE_DAMPING, e), co = S.ecoef[e];
        const double x = (S.fse[e] + fce) / den;
        S.asme[e] = x;   // (asme is rebuilt by the next forward pass)
        Jx += co * x; Jy += co * co / den;
      }
      Jx = wsum(Jx); Jy = wsum(Jy);
      const double kk = H.t0_implicit ? h * H.t0_damping * Jx / (1 + h * H.t0_damping * Jy) : 0.0;   // D5: Sherman-Morrison
      SGT_SYNC();
      SGT_PAR(e, N) {
        const double den = E(SGE_MASS, e) + E(SGE_ARMATURE, e) + h * E(SGE_DAMPING, e);
        const double x = S.asme[e] - kk * E(SGE_COEF, e) / den;
        S.ve[e] += h * x;
        S.qe[e] += h * S.ve[e];
      }
    }
    SGT_PAR(d, ND) {
      S.v[d] += h * S.tmpP[pidx(d)];
      S.q[d] += h * S.v[d];
    }
    SGT_SYNC();
    SGT_STAMP(16);
    }
  }
stage_done: __attribute__((unused));
  SGT_SYNC();
  SGT_ONE {
    S.ctx[CTX_FLAGS] = flags; S.ctx[CTX_NCON] = ncon; S.ctx[CTX_NEFC] = nefc; S.ctx[CTX_ITERS] = iters;
    S.ctx[CTX_TLO] = touch_lo; S.ctx[CTX_THI] = touch_hi; S.ctx[CTX_STOP] = stop;
  }
  SGT_SYNC();
}

// the whole call for one env.  lane: threadIdx.x on the device, 0 on the host
// CHD: the unroll capacity of the per-chain loops (>= the plan's padded stride CS): the kernel is instantiated for 8, 20 and 24
template <int CHD = SGT_CHD>
SG_HD void tree_env(const TreeArgs& A, const int env, double* lds_base) {
  // The plan tables are read-only for the kernel's lifetime: read through the constant address space, a uniform index is a scalar load
  // (K$) that the compiler may hoist and keep, not a vector load behind a full vmcnt wait after every store
  const SGT_CONST SgPlanHeader& H = *(const SGT_CONST SgPlanHeader*)A.H;
  const SGT_CONST SgTreeDev& T = *(const SGT_CONST SgTreeDev*)A.T;
  const int N = H.nelem, ND = T.ND, NB = T.NB, K = T.K, nv = H.nv, nu = H.nu;
  constexpr int CS = CHD;   // (= T.CS: the plan pads the chains' stride to the instantiation's capacity, sg_plan.cpp)
  const double h = H.timestep;
  Lds S;
  lds_carve(S, lds_base, T, N, H.has_free, A.cws + (size_t)env * A.cws_stride + (size_t)SGT_MAXHIT * SGT_HITREC * SGT_RECW + (size_t)SGT_MAXCON * cws_row_doubles(T.CS) + T.NMAT, nullptr, H.nnb);
  const SGT_CONST double* const elemc = (const SGT_CONST double*)A.elem;
  const SGT_CONST SgGenPair* const gpairs = (const SGT_CONST SgGenPair*)A.gpairs;
  const SGT_CONST int* const nbtab = (const SGT_CONST int*)A.nbtab;
  const SGT_CONST SgEqSlot* const sched = (const SGT_CONST SgEqSlot*)A.sched;
  auto E = [&](int f, int e) { return elemc[(size_t)f * N + e]; };
  double* const cw = A.cws + (size_t)env * A.cws_stride;
  const long long CW = cws_row_doubles(CS);
#if !SGT_DEVICE && defined(SGT_EMU_SEPARATE)
  double* const stage = sep_pool() ? sep_part(0, (size_t)SGT_MAXHIT * SGT_HITREC * SGT_RECW) : cw;
  double* const crow0 = sep_pool() ? sep_part(1, (size_t)SGT_MAXCON * CW) : cw + (size_t)SGT_MAXHIT * SGT_HITREC * SGT_RECW;
#else
  double* const stage = cw;
  double* const crow0 = cw + (size_t)SGT_MAXHIT * SGT_HITREC * SGT_RECW;
#endif
  auto crow = [&](int c) { return crow0 + (size_t)c * CW; };                 // J1[3][CS] | W1[3][CS] | J2[3][CS] | W2[3][CS] | scalars
  auto cscal = [&](int c) { return crow0 + (size_t)c * CW + 12 * CS; };
  auto cscr = [&](int c) -> const double* { return c < S.ncache ? S.csc + (size_t)c * SGT_CSC : crow0 + (size_t)c * CW + 12 * CS; };   // for the sweeps: the LDS copy when there is one
#if !SGT_DEVICE && defined(SGT_EMU_SEPARATE)
  double* const Mg = sep_pool() ? sep_part(2, (size_t)T.NMAT) : crow0 + (size_t)SGT_MAXCON * CW;
#else
  double* const Mg = crow0 + (size_t)SGT_MAXCON * CW;    // the chains' mass-matrix blocks [K][CS][CS], identity-padded
#endif
  auto pidx = [&](int d) { const int c = T.d_chain[d]; return c * CS + d - T.c_dof0[c]; };   // flat chain dof -> index in a padded [K][CS] vector

  if (A.mode == 1 && A.mask && !A.mask[env]) return;   // masked reset: the other envs keep everything
  const bool FR = H.has_free != 0;   // the composite's elements hang off a free body (6 dofs): the "object block" below
  double* const gq = A.qpos + (size_t)env * H.nq;
  double* const gv = A.qvel + (size_t)env * nv;
  double* const gw = A.warm + (size_t)env * nv;
  double* const gact = A.act + (size_t)env * (nu > 0 ? nu : 1);
  double* const gctrl = A.ctrl + (size_t)env * (nu > 0 ? nu : 1);

  // ---------------------------------------------------------------- state in
  const double kenv = A.kenv[env];
  SGT_PAR(d, ND) {
    const int j = T.d_gid[d];
    const bool rs = A.mode == 1;
    S.q[d] = rs ? T.d_qpos0[d] : gq[j];
    S.v[d] = rs ? 0.0 : gv[j];
    S.warm[d] = rs ? 0.0 : gw[j];
    S.kd[d] = A.kmask_jnt[j] ? kenv : T.d_stiffness[d];
    S.qacc[d] = 0;
  }
  SGT_PAR(e, N) {
    const int jd = H.elem_dof0 + e, jq = H.elem_qpos0 + e;
    const bool rs = A.mode == 1;
    S.qe[e] = rs ? E(SGE_QPOS0, e) : gq[jq];
    S.ve[e] = rs ? 0.0 : gv[jd];
    S.we[e] = rs ? 0.0 : gw[jd];
    S.ke[e] = A.kmask_jnt[H.elem_jnt0 + e] ? kenv : E(SGE_K0, e);
    S.einvm[e] = 1.0 / (E(SGE_MASS, e) + E(SGE_ARMATURE, e));
    S.ecoef[e] = E(SGE_COEF, e);
    if (FR) {   // B_e = m_e (a_e ; k_e x a_e): the slider's column of the object's mass matrix, body frame (constant)
      const double m = E(SGE_MASS, e), a[3] = {E(SGE_AX, e), E(SGE_AY, e), E(SGE_AZ, e)}, k0[3] = {E(SGE_KX, e), E(SGE_KY, e), E(SGE_KZ, e)};
      double kxa[3];
      cross3(kxa, k0, a);
      for (int c = 0; c < 3; c++) { S.Be[6 * e + c] = m * a[c]; S.Be[6 * e + 3 + c] = m * kxa[c]; }
    }
  }
  if (FR) {
    SGT_ONE {
      const bool rs = A.mode == 1;
      for (int c = 0; c < 7; c++) S.of[OF_P + c] = rs ? H.free_q0[c] : gq[H.free_qadr + c];
      for (int c = 0; c < 3; c++) { S.of[OF_VW + c] = rs ? 0.0 : gv[H.free_dadr + c]; S.of[OF_WL + c] = rs ? 0.0 : gv[H.free_dadr + 3 + c]; }
      for (int c = 0; c < 6; c++) S.of[OF_WARM + c] = rs ? 0.0 : gw[H.free_dadr + c];   // warmstart in dof coordinates (world translations)
    }
  }
  SGT_PAR(c, K) {
    double* cs = S.chs + c * CHS_N;
    const bool rs = A.mode == 1;
    cs[CHS_ACT] = (T.a_has[c] && !rs) ? gact[T.a_id[c]] : 0.0;
    cs[CHS_CTRL] = (T.a_has[c] && !rs) ? gctrl[T.a_id[c]] : 0.0;
    cs[CHS_KT] = T.t_has[c] ? (A.kmask_ten[T.t_id[c]] ? kenv : T.t_k0[c]) : 0.0;
  }
  SGT_ONE {
    for (int i = 0; i < 32; i++) S.icnt[i] = 0;
    if (A.mode == 1)
      for (int u = 0; u < nu; u++) gctrl[u] = 0.0;   // mj_resetData clears ctrl
  }
  SGT_ONE { for (int i = 0; i < CTX_N; i++) S.ctx[i] = 0; }
  SGT_SYNC();

  const int nfwd = A.nsub + (A.mode == 1 ? 1 : 0);
  SGT_PAR(i, 3 * T.NG) S.gsz[i] = T.g_size[i / 3][i % 3];   // the boxes' half sizes next to their poses (the pair walk's tight test)
  for (int sub = 0; sub < nfwd; sub++) {
    SGT_ONE { S.ctx[CTX_SUB] = sub; S.ctx[CTX_LAST] = sub == nfwd - 1 ? 1.0 : 0.0; S.ctx[CTX_INTEGRATE] = (A.mode == 1 && sub == 0) ? 0.0 : 1.0; }
    SGT_SYNC();
    SGT_STAGE_CALL(1);
    if (S.ctx[CTX_STOP] != 0.0) break;   // (uniform: bad positions / velocities -- the env stops integrating for the rest of the call)
    SGT_STAGE_CALL(2);
    SGT_STAGE_CALL(3);
    SGT_STAGE_CALL(4);
    if (S.ctx[CTX_STOP] != 0.0) break;   // (bad accelerations)
  }
  const int flags = (int)S.ctx[CTX_FLAGS], ncon = (int)S.ctx[CTX_NCON], nefc = (int)S.ctx[CTX_NEFC], iters = (int)S.ctx[CTX_ITERS];
  const unsigned touch_lo = (unsigned)S.ctx[CTX_TLO], touch_hi = (unsigned)S.ctx[CTX_THI];

  // ---------------------------------------------------------------- state and outputs back
  SGT_SYNC();
  SGT_PAR(d, ND) {
    const int j = T.d_gid[d];
    gq[j] = S.q[d]; gv[j] = S.v[d]; gw[j] = S.warm[d];
  }
  SGT_PAR(e, N) {
    const int jd = H.elem_dof0 + e;
    gq[H.elem_qpos0 + e] = S.qe[e]; gv[jd] = S.ve[e]; gw[jd] = S.we[e];
  }
  if (FR) SGT_ONE {
    for (int c = 0; c < 7; c++) gq[H.free_qadr + c] = S.of[OF_P + c];
    for (int c = 0; c < 3; c++) { gv[H.free_dadr + c] = S.of[OF_VW + c]; gv[H.free_dadr + 3 + c] = S.of[OF_WL + c]; }
    for (int c = 0; c < 6; c++) gw[H.free_dadr + c] = S.of[OF_WARM + c];
  }
  SGT_PAR(c, K)
    if (T.a_has[c]) gact[T.a_id[c]] = S.chs[c * CHS_N + CHS_ACT];
#ifdef SG_DEBUG_WORK
  {
    SGT_SYNC();
    const long long nl = (long long)(lds_bytes(T, N, H.has_free, H.nnb) / sizeof(double)), at = cws_doubles(T, N, H.has_free, H.nnb) - nl;
    SGT_PAR(i, nl) cw[at + i] = lds_base[i];
  }
#endif
  SGT_ONE {
    A.flags[env] = flags; A.ncon[env] = ncon; A.nefc[env] = nefc; A.iters[env] = iters;
    A.touch[env] = (int)touch_lo;
    A.touch_words[2 * env] = (int)touch_lo; A.touch_words[2 * env + 1] = (int)touch_hi;
  }
}

}  // namespace sgt

#if defined(__HIPCC__)
// launchers (sg_tree.hip): the kernel is a translation unit of its own
hipError_t sg_tree_prepare();
int sg_tree_occupancy(int CS, size_t lds_bytes);   // workgroups per CU the runtime grants the instantiation (sg_tree.hip)
hipError_t sg_launch_tree(const sgt::TreeArgs& a, int CS, size_t lds_bytes, hipStream_t s);
#endif
