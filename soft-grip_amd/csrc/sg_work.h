// sg_work.h -- what the kernels of the two-finger class share: the batch's device work space (SgWork), the kernel argument blocks,
// the layouts of the contact rows (SG_ROW_INDEX) and of the solver's stream of step factors (SG_CST_INDEX), and the LAUNCHERS.
//
// One translation unit per kernel family -- sg_phase.hip (sg_chain_kernel, sg_phase_kernel), sg_rows.hip (sg_pgs_rows_kernel: the
// solver), sg_tree.hip (the tree pipeline) and, in test builds only (-DSG_LEGACY_PIPELINES), sg_legacy.hip (r01's fused and split
// pipelines, kept as cross-checks) -- compiled side by side; sg_api.hip (the C ABI) sees the launchers below and nothing else.
// A substep of the rows pipeline is the chain   chain -> phase(finish + begin) -> [general pass] -> pgs_rows.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/softgrip.h"
#include "sg_math.h"
#include "sg_general.h"

using namespace sgm;

#define SG_CAP 64        // contact capacity per stream in the split pipeline
#define SG_NF 26         // constant fields of a contact record
#define SG_RF 30         // record fields: 26 constants, 3 force components, slider index (as an int in a double slot)
// Contact records are blocked for the PGS kernel: one block per (slot, PGS wavefront) holds SG_RF fields x SG_SPW streams,
// so a wavefront reads a whole record with ONE vector address plus immediate offsets (field stride 128 B), and its 16
// stream lanes use every byte of the 128-B lines they touch.  Block index nwb (one past the last wave) is a dummy
// target for the unconditional stores of idle lanes.
#define SG_REC_INDEX(slot, wave, field, pos, nwb) (((((size_t)(slot)) * ((nwb) + 1) + (wave)) * SG_RF + (field)) * SG_SPW + (pos))
#define SG_G 8           // lanes per env in the PGS kernel (8 measured best: 16 -> 1.35x slower PGS, 4 -> 1.07x slower)
#define SG_EPW (64 / SG_G)   // envs per PGS wavefront
#define SG_SPW (2 * SG_EPW)  // finger streams per PGS wavefront
#define SG_GEN_GRID 256  // blocks of the general contact pass (they stride over the listed envs: no limit on how many there are)
#define SG_CHW 160       // doubles of chain hand-off per stream (layout: see sg_chain_kernel)

struct SgWork {          // device workspace of one batch (all pointers device memory)
  double* crec;          // [SG_CAP][nwb + 1][SG_RF][16]   nwb = ceil(nenv / 8) PGS wavefronts (+1 dummy block)
  unsigned long long* secprof;  // [32] cycle sums per kernel section (only written when built with -DSG_SECTION_PROF)
  double* crow;          // [SG_CAP + 2][nwb + 2][SG_RK / 2][64][2]   row layout (sg_pgs_rows_kernel), +2 dummy blocks
  double* cdummy;        // [ceil(nenv / 4)][SG_RK / 2][64][2]  one private all-zero block per PGS wavefront for its lanes without an env
  int* ns;               // [S]
  double* envh;          // [4][nenv]: tb, tR, tA, tf
  int *shared, *pending, *status, *iters, *ncon, *nefc, *touch;  // [nenv]
  double *sMinv, *saF;   // [16][S], [4][S]
  int* lim_active;       // [S]
  double* lim;           // [4][SG_MAXLIM][S]: sign, R, b, f
  double *as, *eqf, *eqb, *eqR;  // [nenv][N]
  double *asme, *fsm;    // [nenv][N]  begin -> finish hand-off
  double* chh;           // [nenv][2][SG_CHW]  chain hand-off (enum SGH_*)
  double *nbf, *nbb, *nbR;  // [nenv][3 N]  neighbour equality rows (models with H.nnb > 0) by slot (SgPlan::nbtab): force, right-hand side, regulariser
  double* cst;           // [ceil(nenv / 4)][eq_rounds + 8][64][2]  step factors c = (1/m) / (A + R) of the equality rows in the SOLVER's order: what lane l of
                         // wavefront w reads in round k of its equality block (SG_CST_INDEX); written by the phase kernel, zero where a slot / row is idle
  double* gcon;          // [nenv][SG_GEN_MAXCON][SG_GEN_W]  contacts of envs on the general contact path (sg_general.h), in mj_collision's order
  int* gen;              // [nenv]  number of general contacts of the pending substep (0: the env is on the fast path)
  int* gen_count;        // [1]  envs the main phase pass has put on gen_list in this substep (reset by sg_chain_kernel)
  int* gen_list;         // [nenv]
  // models of four element rounds (193 .. 256 sliders: the ball): the phase kernel's pair list and per-slot slider pushes, which its smaller
  // instantiations keep in LDS (Smem2): without them its block is 19.5 KB instead of 22.6 -- EIGHT workgroups per CU, the 4096 wavefronts in two rounds
  // instead of three (a round is ~60 us whatever its size: profiles/r05s_phase_kernel_occupancy.txt)
  unsigned short* gpairs16;  // [nenv][SG_MAXCH * SG_CG * (4 * 64 + 2)]   (2 entries otherwise)
  double* gcval;             // [nenv][SG_MAXCH][64]                       (2 entries otherwise)
};

struct SgPhaseArgs {
  const SgPlanHeader* H;
  const double* elem;
  double *qpos, *qvel, *warm, *act, *ctrl;
  const double* kenv;
  const int *kmask_jnt, *kmask_ten;
  const unsigned char* mask;
  double* sens;
  long long sens_stride;
  SgWork w;
  int nenv;
  int do_reset, do_finish, finish_integrate, do_begin, first;
  int rowlayout;  // 1: export contact records in the row layout of sg_pgs_rows_kernel
  const int* nbtab;  // SgPlan::nbtab (neighbour rows per element), nullptr when H.nnb == 0
  const int* cpos;   // neighbour-row models: per element, where the four step factors of its equality block sit in a wavefront's round-major
                     // stream W.cst (doubles: round * 128 + 4 * block slot); the host derives it from SgPlan::sched (sg_api.hip)
  int cst_rounds;    // eq_rounds + 8: rounds of W.cst per solver wavefront; 0: the solver keeps the factors in LDS (SG_ROWS_NB_MODE 1), nothing to write
  const SgGenPair* gpairs;  // SgPlan::gpairs (the general contact path's candidate pairs)
  // copies of the plan header's sizes, by value: the kernels' first addresses then do not wait for a load from *H
  int nelem, nv, nu, elem_dof0, nchain, t0_id;
  double timestep;
};

// chain hand-off record (doubles): written by the chain stage (phase kernel or sg_chain_kernel), read by FINISH and by BEGIN
// Row layout of the contact records for sg_pgs_rows_kernel: a finger stream is a QUAD of lanes, lane r < 3 holds row r
// (normal, tangent 1, tangent 2) of every contact, lane 3 what the rows share; lane q also OWNS finger acceleration aF[q].
// Block per (slot, wavefront): 8 field PAIRS x 64 lanes x 2 doubles, so a lane fetches two fields with one 16-byte load.
// Field k of lane (8 * env_in_wave + 4 * chain + r), r < 3:
//   0..3 Jf[r][0..3] | 4 Js[r] | 5 b[r] | 6 f[r] | 7 (A f)[r] | 8..10 A[r][0..2] | 11 invm * Js[r] | 12..14 W_0[r] W_1[r] W_2[r] | 15 R
// of lane r = 3:
//   0..2 inverse friction block P11 P12 P22 | 3 zero | 4 slider index | 5..7 zero | 8..11 eigen-decomposition of the friction-scaled
//   block S = Q diag(e1, e2) Q': e1 e2 cos sin (for the QCQP Newton iteration) | 12..14 W_0[3] W_1[3] W_2[3] | 15 zero
// W_k = M^-1 J_F[k]' (4 values per row k; lane q keeps the q-th of each): the finger update aF[q] += sum_k W_k[q] df_k is three
// multiply-adds on lane q after broadcasting the three force changes, instead of four more quad sums and a 4 x 4 product on
// every lane.  f and A f (fields 6, 7: one pair) are the only fields the solver writes.  Blocks nwb and nwb + 1 of every slot
// are dummies: lanes of envs that do not exist read block nwb (all zero, never written) and write to block nwb + 1.
#define SG_RK 16
// the solver's stream of step factors (SgWork::cst), in doubles: wavefront w (four envs), round k of its equality block, lane l
#define SG_CST_INDEX(w, k, l, rounds) ((((size_t)(w) * (rounds) + (k)) * 64 + (l)) * 2)
#define SG_ROW_INDEX(slot, wave, k, lane, nwb) \
  ((((((size_t)(slot)) * ((nwb) + 2) + (wave)) * (SG_RK / 2) + (k) / 2) * 64 + (lane)) * 2 + ((k) & 1))

enum { SGH_QSM = 0, SGH_QFRC = 4, SGH_ACTDOT = 8, SGH_M = 9, SGH_K = 25, SGH_MINV = 73, SGH_V = 89, SGH_W = 93, SGH_BOX = 97,
       SGH_LIMACT = 121, SGH_LIMSIGN = 122, SGH_LIMR = 130, SGH_LIMB = 138, SGH_LIMF = 146 };

struct StageRec2 {
  double dist, pos[3], n[3];
  int sl, box;
};

struct ChainLds2 {  // what the phase kernel needs of a finger chain (imported from sg_chain_kernel's hand-off record)
  double v[SG_CD], w[SG_CD], qacc_smooth[SG_CD], Minv[16];
  int lim_active, pad;
  double lim_sign[SG_MAXLIM], lim_R[SG_MAXLIM], lim_b[SG_MAXLIM], lim_f[SG_MAXLIM];  // contiguous, in the hand-off record's order
};

#define SG_PHASE_SLIM(R) ((R) >= 4)   // the pair list and the slot pushes live in the work space (SgWork::gpairs16, gcval), not in LDS
#define SG_PAIRS_CAP(R) (SG_MAXCH * SG_CG * ((R) * 64 + 2))
template <int R, int CPL, bool NB>
struct Smem2 {
  ChainKin K[SG_MAXCH];
  ChainLds2 cs[SG_MAXCH];
  double boxp[SG_MAXCH * SG_CG][3], boxm[SG_MAXCH * SG_CG][9];
  double ve[R * 64], asme[R * 64], we[R * 64], as[R * 64];
  StageRec2 stage[SG_MAXCH][32 * CPL];
  unsigned short pairs[SG_PHASE_SLIM(R) ? 4 : SG_PAIRS_CAP(R)];  // broadphase survivors, (box << 12) | element, in contact order
  unsigned char eslot[R * 64][SG_MAXCH * SG_CG];          // per element and box: first contact slot (< 64) | (contact count << 6)
  double cval[SG_PHASE_SLIM(R) ? 1 : SG_MAXCH * 32 * CPL];   // per contact slot [chain][32 CPL]: invm * Js' f (its push on the slider)
};
#define SG_PAIR_CENTER 0xFFF  // element code of the object's centre sphere

__device__ __forceinline__ double wave_sum2(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
  return x;
}
__device__ __forceinline__ int lanes_below2(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
}

#if defined(__HIP_DEVICE_COMPILE__)
#define SG_CONSTAS __attribute__((address_space(4)))   // (the host pass of the same source has no address spaces to convert between)
#else
#define SG_CONSTAS
#endif

// section timing for scripts/section_profile.py (build_native.py --prof): every wavefront sums the cycles between stamps per
// section in registers and adds them to W.secprof[] once, at its end.  Compiled out of the product library.
#ifdef SG_SECTION_PROF
#define SG_T0() unsigned long long t_prev_ = __builtin_readcyclecounter(), t_acc_[26] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define SG_T(k)                                                   \
  do {                                                            \
    unsigned long long t_now_ = __builtin_readcyclecounter();     \
    t_acc_[k] += t_now_ - t_prev_;                                \
    t_prev_ = __builtin_readcyclecounter();                       \
  } while (0)
#define SG_TEND()                                                 \
  do {                                                            \
    if (threadIdx.x == 0)                                         \
      for (int k_ = 0; k_ < 26; k_++)                             \
        if (t_acc_[k_]) atomicAdd(&a.w.secprof[k_], t_acc_[k_]);  \
  } while (0)
#else
#define SG_T0()
#define SG_T(k)
#define SG_TEND()
#endif


struct SgPgsArgs {
  const unsigned* tab;       // the schedule as the solver's LDS table words (lane 2 b + h of a 16-lane group, block slot b; sg_api.hip)
  const SgEqSlot* sched;  // SgPlan::sched (equality-row schedule of neighbour-row models), nullptr when H.nnb == 0
  const int* nbtab;       // SgPlan::nbtab
  const SgPlanHeader* H;
  const double* elem;
  SgWork w;
  int nenv;
};

// LDS of sg_pgs_rows_kernel in doubles (kernel and host use the same expressions): EPW envs per wavefront
#define SG_ROWS_LDS_FIX(EPW, NR) ((size_t)(5 * (EPW) + 2) * (NR) + 72)
// neighbour-row models: slider words [EPW][N + 2] | row states g [EPW][4 (N + 1)] | table: 16 lanes x 4 B per round | 72 + 2 EPW.  (Until r04 the rows'
// step factors c sat beside g -- 16 bytes a row, 70 KB for the ball's 218 elements: two workgroups per CU, i.e. the 1024 wavefronts of
// a 4096-env batch in TWO rounds.  They now stream from memory in round-major order, SgWork::cst: 38 KB, four per CU, one round.)
#define SG_ROWS_NA_NB(N) (((N) + 2) & ~1)
#define SG_ROWS_LDS_NB(EPW, N, ROUNDS, CST) ((size_t)(EPW) * SG_ROWS_NA_NB(N) + (size_t)((CST) ? 4 : 8) * (EPW) * ((N) + 1) + (size_t)((CST) ? 8 : 16) * ((ROUNDS) + 8) + 72 + 2 * (EPW))
// the step factors stay in LDS beside the states (NB = 1) while four workgroups of the solver still share a CU's 160 KB that way
#define SG_ROWS_NB_MODE(N, ROUNDS) (sizeof(double) * SG_ROWS_LDS_NB(4, N, ROUNDS, 0) <= 40 * 1024 ? 1 : 2)
// ---- launchers (defined next to their kernels) ----
hipError_t sg_launch_chain(const SgPhaseArgs& p, int nenv, hipStream_t s);
// the main pass over all envs and, when genpass, the general contact pass behind it (SG_GEN_GRID blocks striding over the listed envs; sg_general.h)
hipError_t sg_launch_phase(const SgPhaseArgs& p, int rounds, bool nb, bool genpass, int nenv, hipStream_t s);
hipError_t sg_rows_prepare();                 // once per device: dynamic-LDS limits of every solver instantiation
int sg_rows_nsl(int nelem);                   // the instantiated joint-fix rows per lane (template parameter NSL) that holds nelem
hipError_t sg_launch_rows(const SgPgsArgs& a, int nsl, int nb /* 0 | 1 | 2: template parameter NB of sg_pgs_rows_kernel */, int epw, int nenv, size_t lds_bytes, hipStream_t s);
#ifdef SG_LEGACY_PIPELINES   // test builds only (build_native.py --legacy): r01's fused kernel and the split pipeline's stream-per-lane solver
struct SgKArgs;
hipError_t sg_legacy_prepare();
hipError_t sg_launch_pgs_split(const SgPgsArgs& a, int nenv, size_t lds_bytes, hipStream_t s);
hipError_t sg_launch_fused(const SgKArgs& a, int rounds, int nenv, hipStream_t s);
#endif
