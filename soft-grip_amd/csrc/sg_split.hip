// sg_split.hip -- "split" pipeline: the same substep as sg_kernels.hip, cut at the constraint solve.
//
//   sg_phase_kernel<R,CPL>  one wavefront per env.  FINISH part: takes the solver's result for the previous
//                           substep (slider and chain constraint accelerations), produces qacc, the sensors,
//                           the warmstart and integrates.  BEGIN part: smooth dynamics, collision, constraint
//                           rows, warmstart test, and EXPORTS the constraint problem to the workspace.
//   sg_pgs_kernel           the PGS sweeps.  8 lanes per env / 8 envs per wavefront: the joint-fix rows of an env
//                           are spread over its 8 lanes (slider arrays in LDS), and each of the two finger streams
//                           gets ITS OWN LANE that walks its contacts sequentially, streaming the contact records
//                           from memory (layout [slot][field][stream]: the stream lanes of consecutive envs read
//                           consecutive addresses).  In the fused kernel 2 of 64 lanes carry the serial contact
//                           sweep; here 16 of 64 do, and the records no longer pin 116 VGPRs per lane.
//
// Row order, update formulas and therefore results are the same Gauss-Seidel sweep as the fused kernel and the
// oracle (checked by the same parity tests).  A launch of n_substeps is the chain
//   phase(begin) -> pgs -> phase(finish+begin) -> pgs -> ... -> phase(finish).
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
#define SG_CHW 160       // doubles of chain hand-off per stream (layout: see sg_chain_kernel)
#define SG_GEN_LIST 256  // envs per substep the general contact pass can take (one block each); further ones are flagged

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
  double* gcon;          // [nenv][SG_GEN_MAXCON][SG_GEN_W]  contacts of envs on the general contact path (sg_general.h), in mj_collision's order
  int* gen;              // [nenv]  number of general contacts of the pending substep (0: the env is on the fast path)
  int* gen_count;        // [1]  envs the main phase pass has put on gen_list in this substep (reset by sg_chain_kernel)
  int* gen_list;         // [SG_GEN_LIST]
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

template <int R, int CPL, bool NB>
struct Smem2 {
  ChainKin K[SG_MAXCH];
  ChainLds2 cs[SG_MAXCH];
  double boxp[SG_MAXCH * SG_CG][3], boxm[SG_MAXCH * SG_CG][9];
  double ve[R * 64], asme[R * 64], we[R * 64], as[R * 64];
  StageRec2 stage[SG_MAXCH][32 * CPL];
  unsigned char owner[SG_MAXCH][R * 64];              // [c][e] != 0: chain c has a contact on this element's slider (plain stores of 1)
  unsigned short pairs[SG_MAXCH * SG_CG * (R * 64 + 2)];  // broadphase survivors, (box << 12) | element, in contact order
  unsigned char eslot[R * 64][SG_MAXCH * SG_CG];          // per element and box: first contact slot (< 64) | (contact count << 6)
  double cval[SG_MAXCH][32 * CPL];                        // per contact slot: invm * Js' f (its push on the slider)
  double nbf[NB ? 3 * R * 64 : 1];                        // neighbour equality rows: warmstart force, by row id
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

// ------------------------------------------------------------------------------------------------
// General contact path in the phase kernel (sg_general.h): an env in which a pair outside the fast path's two kinds is within reach
// builds, for this substep, ONE ordered contact list over the plan's candidate pairs and exports generic rows (W.gcon); the solver
// sweeps them as one serial stream.  Rare by construction (never in the reference's scenes), so it is kept OUT of the kernel's
// straight-line code: __noinline__ functions working on the kernel's LDS block through pointers.  In this mode the LDS regions of the
// fast path's contact staging are re-used (layout at the call site in sg_phase_kernel).
// ------------------------------------------------------------------------------------------------
struct GenLds {
  StageRec2* stage;            // [SG_GEN_MAXCON]
  double* gas;                 // [R * 64] per element: invm * sum of Js' f over the general contacts on its slider (inside the staging area)
  double* gg;                  // [SG_MAXCH][SG_CD]: sum over the general contacts of Jf[c]' f
  double* tmp;                 // 96 doubles (box - box clipping)
  const double (*boxp)[3];
  const double (*boxm)[9];
  const ChainKin* K;
  const ChainLds2* cs;
  const double *qe, *ve, *asme, *we;   // per element (LDS): slider position, velocity, smooth acceleration, warmstart
};

// geometry of a capsule of the pair table
__device__ __forceinline__ void sg_gen_capsule(const SgPhaseArgs& a, const GenLds& L, int e, double* cp, double* cax) {
  const int N = a.nelem;
  auto EL = [&](int f, int k) { return a.elem[(size_t)f * N + k]; };
  const double dq = L.qe[e] - EL(SGE_QPOS0, e);
  cp[0] = EL(SGE_GX, e) + EL(SGE_AX, e) * dq; cp[1] = EL(SGE_GY, e) + EL(SGE_AY, e) * dq; cp[2] = EL(SGE_GZ, e) + EL(SGE_AZ, e) * dq;
  cax[0] = EL(SGE_CX, e); cax[1] = EL(SGE_CY, e); cax[2] = EL(SGE_CZ, e);
}

// collision + rows.  Returns the number of contacts; *flags gets CONTACTFULL / UNSUPPORTED_PAIR bits, *touch the finger-box bits.
__device__ __noinline__ int sg_gen_phase(const SgPhaseArgs& a, const SgPlanHeader& H, const int env, const GenLds L, int* flags, int* touch) {
  const int lane = threadIdx.x, N = a.nelem;
  auto EL = [&](int f, int k) { return a.elem[(size_t)f * N + k]; };
  const SgWork& W = a.w;
  int ng = 0, fl = 0;
  // ---- the pair table, 64 pairs per pass: lane = pair.  Pairs with at most two contacts (capsule / sphere against a box, plane against a
  //      capsule) are evaluated by their lanes and appended in order with ballots; a box against a box or the plane (up to 8 / 4
  //      contacts) within reach is evaluated by lane 0 at its place in the order
#pragma unroll 1
  for (int p0 = 0; p0 < H.ngpair; p0 += 64) {
    const int pi = p0 + lane;
    const bool have = pi < H.ngpair;
    SgGenPair gp;
    gp.kind = SGP_UNSUPPORTED; gp.g1 = gp.g2 = 0; gp.pad = 0;
    if (have) gp = a.gpairs[pi];
    ConRec r0, r1;
    int n = 0;
    bool big = false;
    if (have) {
      double cp[3] = {0, 0, 0}, cax[3] = {0, 0, 1};
      if (gp.kind == SGP_PLANE_CAP) {
        sg_gen_capsule(a, L, sgg_index(gp.g2), cp, cax);
        const double dif[3] = {cp[0] - H.plane_pos[0], cp[1] - H.plane_pos[1], cp[2] - H.plane_pos[2]};
        if (!(dot3(dif, H.plane_normal) > H.con_margin + H.cap_rbound))
          n = gen_plane_capsule(H.plane_pos, H.plane_normal, cp, cax, H.cap_radius, H.cap_hl, H.con_margin, r0, r1);
      } else if (gp.kind == SGP_PLANE_BOX) {
        const double *p2, *R2, *s2;
        double rb2;
        gen_box_of(gp.g2, H, L.boxp, L.boxm, p2, R2, s2, rb2);
        const double dif[3] = {p2[0] - H.plane_pos[0], p2[1] - H.plane_pos[1], p2[2] - H.plane_pos[2]};
        big = !(dot3(dif, H.plane_normal) > H.con_margin + rb2);
      } else {
        const double *p1 = nullptr, *R1 = nullptr, *s1 = nullptr, *p2, *R2, *s2;
        double rb1 = 0, rb2;
        const bool boxes = sgg_kind(gp.g1) == SGG_BOX || sgg_kind(gp.g1) == SGG_STATIC;
        if (gp.kind == SGP_SPH_BOX) { cp[0] = H.center_pos[0]; cp[1] = H.center_pos[1]; cp[2] = H.center_pos[2]; rb1 = H.center_radius; }
        else if (gp.kind == SGP_CAP_BOX) { sg_gen_capsule(a, L, sgg_index(gp.g1), cp, cax); rb1 = H.cap_rbound; }
        else if (boxes) { gen_box_of(gp.g1, H, L.boxp, L.boxm, p1, R1, s1, rb1); cp[0] = p1[0]; cp[1] = p1[1]; cp[2] = p1[2]; }
        if (gp.kind == SGP_UNSUPPORTED && !boxes) fl |= SG_FLAG_UNSUPPORTED_PAIR;   // cannot even be tested: flagged whenever this path runs
        else {
          gen_box_of(gp.g2, H, L.boxp, L.boxm, p2, R2, s2, rb2);
          const double dif[3] = {p2[0] - cp[0], p2[1] - cp[1], p2[2] - cp[2]}, bound = rb1 + rb2 + H.con_margin;
          if (dot3(dif, dif) <= bound * bound) {
            if (gp.kind == SGP_SPH_BOX) n = sphere_box(cp, H.center_radius, p2, R2, s2, H.con_margin, r0);
            else if (gp.kind == SGP_CAP_BOX) {
              const int mk = capsule_box(cp, cax, H.cap_radius, H.cap_hl, p2, R2, s2, H.con_margin, r0, r1);
              if ((mk & 2) && !(mk & 1)) r0 = r1;
              n = (mk & 1) + ((mk >> 1) & 1);
            } else big = true;
          }
        }
      }
      // only contacts inside the margin become constraints (and count, as on the fast path)
      if (n == 2 && !(r1.dist < H.con_margin)) n = 1;
      if (n >= 1 && !(r0.dist < H.con_margin)) { r0 = r1; n--; }
    }
    // ordered append, split at the big pairs
    unsigned long long todo = __ballot(n > 0 || big);
#pragma unroll 1
    while (todo) {
      const unsigned long long bigm = __ballot(big) & todo;
      const int Lb = bigm ? __ffsll((long long)bigm) - 1 : 64;
      const unsigned long long seg = Lb == 64 ? todo : (todo & ((1ull << Lb) - 1ull));
      const bool mine = (seg >> lane) & 1ull;
      const unsigned long long m1 = __ballot(mine && n >= 1), m2 = __ballot(mine && n >= 2);
      const int base = ng + lanes_below2(m1) + lanes_below2(m2);
      auto put = [&](int at, const ConRec& r) {
        StageRec2& o = L.stage[at];
        o.dist = r.dist; o.sl = pi; o.box = 0;
        for (int q = 0; q < 3; q++) { o.pos[q] = r.pos[q]; o.n[q] = r.n[q]; }
      };
      if (mine && n >= 1 && base < SG_GEN_MAXCON) put(base, r0);
      if (mine && n >= 2 && base + 1 < SG_GEN_MAXCON) put(base + 1, r1);
      ng += __popcll(m1) + __popcll(m2);
      todo &= ~seg;
      if (Lb < 64) {
        const int pb = p0 + Lb;            // uniform
        const SgGenPair gb = a.gpairs[pb];
        int cnt = 0;
        if (lane == 0) {
          const int room = SG_GEN_MAXCON - (ng < SG_GEN_MAXCON ? ng : SG_GEN_MAXCON);
          StageRec2* out = L.stage + (ng < SG_GEN_MAXCON ? ng : 0);
          const double *p2, *R2, *s2;
          double rb2;
          gen_box_of(gb.g2, H, L.boxp, L.boxm, p2, R2, s2, rb2);
          int nb = 0;
          if (room >= 8) {   // the routines write up to 8 records: without room for them the list is full
            if (gb.kind == SGP_PLANE_BOX) nb = gen_plane_box(H.plane_pos, H.plane_normal, p2, R2, s2, H.con_margin, out);
            else {
              const double *p1, *R1, *s1;
              double rb1;
              gen_box_of(gb.g1, H, L.boxp, L.boxm, p1, R1, s1, rb1);
              nb = gen_box_box(p1, R1, s1, p2, R2, s2, H.con_margin, out, (double (*)[3])L.tmp, (double (*)[3])(L.tmp + 48));
            }
            for (int q = 0; q < nb; q++)      // keep the contacts inside the margin, in place
              if (out[q].dist < H.con_margin) { if (cnt != q) out[cnt] = out[q]; out[cnt].sl = pb; cnt++; }
            if (gb.kind == SGP_UNSUPPORTED && cnt > 0) { cnt = 0; fl |= SG_FLAG_UNSUPPORTED_PAIR; }
          } else {
            fl |= SG_FLAG_CONTACTFULL;
          }
        }
        ng += __shfl(cnt, 0);
        todo &= ~(1ull << Lb);
        if (lane == Lb) big = false;
      }
    }
  }
  if (ng > SG_GEN_MAXCON) { ng = SG_GEN_MAXCON; fl |= SG_FLAG_CONTACTFULL; }
  __syncthreads();
  // ---- rows: lane = contact, SG_GEN_ROUNDS rounds.  Exported to W.gcon; what the warmstart test and the slider accelerations need
  //      (slider index and push per contact) goes back into the staging area once every lane has read its record
  double Minv2[SG_MAXCH][16], vc2[SG_MAXCH][SG_CD], asm2[SG_MAXCH][SG_CD], warm2[SG_MAXCH][SG_CD];
#pragma unroll
  for (int c = 0; c < SG_MAXCH; c++) {
#pragma unroll
    for (int i = 0; i < 16; i++) Minv2[c][i] = L.cs[c].Minv[i];
#pragma unroll
    for (int d = 0; d < SG_CD; d++) { vc2[c][d] = L.cs[c].v[d]; asm2[c][d] = L.cs[c].qacc_smooth[d]; warm2[c][d] = L.cs[c].w[d]; }
  }
  double gsum[SG_MAXCH][SG_CD] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, push[SG_GEN_ROUNDS];
  int slk[SG_GEN_ROUNDS], tch = 0;
  StageRec2 rec[SG_GEN_ROUNDS];
#pragma unroll
  for (int k = 0; k < SG_GEN_ROUNDS; k++)
    if (lane + 64 * k < ng) rec[k] = L.stage[lane + 64 * k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < SG_GEN_ROUNDS; k++) {
    const int i = lane + 64 * k;
    push[k] = 0; slk[k] = -1;
    if (i < ng) {
      const SgGenPair gp = a.gpairs[rec[k].sl];
      const GenSide S1 = gen_side_of(gp.g1, H, a.elem + (size_t)SGE_BINVW * N), S2 = gen_side_of(gp.g2, H, a.elem + (size_t)SGE_BINVW * N);
      const int sl = S1.sl >= 0 ? S1.sl : S2.sl;
      double ax[3] = {0, 0, 0}, hint[3] = {0, 0, 0}, ve_ = 0, as_ = 0, we_ = 0, im = 0;
      if (sl >= 0) {
        ax[0] = EL(SGE_AX, sl); ax[1] = EL(SGE_AY, sl); ax[2] = EL(SGE_AZ, sl);
        ve_ = L.ve[sl]; as_ = L.asme[sl]; we_ = L.we[sl];
        im = 1.0 / (EL(SGE_MASS, sl) + EL(SGE_ARMATURE, sl));
      }
      if (gp.kind == SGP_PLANE_CAP) { hint[0] = EL(SGE_CX, sl); hint[1] = EL(SGE_CY, sl); hint[2] = EL(SGE_CZ, sl); }   // first tangent along the capsule
      GenContact c;
      gen_contact_build(c, rec[k], gp.kind == SGP_PLANE_CAP ? hint : nullptr, S1, S2, L.K, Minv2, vc2, asm2, warm2, ax, ve_, as_, we_, im, H);
      gen_contact_store(W.gcon + ((size_t)env * SG_GEN_MAXCON + i) * SG_GEN_W, c);
#pragma unroll
      for (int ch = 0; ch < SG_MAXCH; ch++)
#pragma unroll
        for (int d = 0; d < SG_CD; d++) gsum[ch][d] += c.Jf[ch][0][d] * c.f[0] + c.Jf[ch][1][d] * c.f[1] + c.Jf[ch][2][d] * c.f[2];
      slk[k] = sl;
      push[k] = c.invm * (c.Js[0] * c.f[0] + c.Js[1] * c.f[1] + c.Js[2] * c.f[2]);
      // touch bits: a finger box against an object geom (capsule or centre sphere)
      if ((gp.kind == SGP_CAP_BOX || gp.kind == SGP_SPH_BOX) && sgg_kind(gp.g2) == SGG_BOX) tch |= 1 << sgg_index(gp.g2);
    }
  }
  double* const gpush = (double*)L.stage;                    // [SG_GEN_MAXCON]
  int* const gsl = (int*)(gpush + SG_GEN_MAXCON);            // [SG_GEN_MAXCON]
#pragma unroll
  for (int k = 0; k < SG_GEN_ROUNDS; k++)
    if (lane + 64 * k < SG_GEN_MAXCON) { gpush[lane + 64 * k] = push[k]; gsl[lane + 64 * k] = slk[k]; }
#pragma unroll
  for (int ch = 0; ch < SG_MAXCH; ch++)
#pragma unroll
    for (int d = 0; d < SG_CD; d++) {
      const double x = wave_sum2(gsum[ch][d]);
      if (lane == 0) L.gg[ch * SG_CD + d] = x;
    }
  __syncthreads();
  for (int e = lane; e < N; e += 64) {   // per element: the pushes of its general contacts, in list order
    double acc = 0;
    for (int i = 0; i < ng; i++)
      if (gsl[i] == e) acc += gpush[i];
    L.gas[e] = acc;
  }
  int t = 0;
#pragma unroll
  for (int bb = 0; bb < SG_MAXCH * SG_CG; bb++)
    if (__ballot((tch >> bb) & 1)) t |= 1 << bb;
  *touch = t;
  int f2 = 0;
#pragma unroll
  for (int bit = 0; bit < 6; bit++)
    if (__ballot((fl >> bit) & 1)) f2 |= 1 << bit;
  *flags = f2;
  __syncthreads();
  return ng;
}

// warmstart cost of the general contacts (per-lane partial sums) for the current accelerations; zero != 0: set their forces to 0 instead
__device__ __noinline__ double sg_gen_cost(const SgPhaseArgs& a, const int env, const int ng, const double (*aF)[SG_CD], const double* as_lds,
                                           const int zero) {
  const int lane = threadIdx.x;
  double cp = 0;
  for (int i = lane; i < ng; i += 64) {
    double* rec = a.w.gcon + ((size_t)env * SG_GEN_MAXCON + i) * SG_GEN_W;
    if (zero) { rec[SG_GEN_F_OFF] = rec[SG_GEN_F_OFF + 1] = rec[SG_GEN_F_OFF + 2] = 0.0; continue; }
    GenContact c;
    gen_contact_load(c, rec);
    const double as_ = c.sl >= 0 ? as_lds[c.sl] : 0.0;
#pragma unroll
    for (int r = 0; r < 3; r++) {
      double Ja = c.Js[r] * as_;
#pragma unroll
      for (int ch = 0; ch < SG_MAXCH; ch++)
#pragma unroll
        for (int d = 0; d < SG_CD; d++) Ja += c.Jf[ch][r][d] * aF[ch][d];
      cp += c.f[r] * (0.5 * (Ja + c.R * c.f[r]) + c.b[r]);
    }
  }
  return cp;
}

// ------------------------------------------------------------------------------------------------
// phase kernel: [finish previous substep] [begin next substep]
// ------------------------------------------------------------------------------------------------
// GEN = false: the kernel every env runs.  An env in which a collision pair outside the fast path's two kinds is within reach is put on
// W.gen_list; the GEN = true instantiation -- launched after it with a small fixed grid, its blocks walk that list -- then redoes
// the BEGIN part of exactly those envs on the general contact path (sg_gen_phase) and overwrites their exports.  The general path
// needs a stack (scratch memory) and every register; compiled into the main instantiation it doubled that kernel's time.
template <int R, int CPL, bool NB, bool GEN = false>  // NB: the model has neighbour equality rows (H.nnb > 0)
__global__ __launch_bounds__(64, 2) void sg_phase_kernel(SgPhaseArgs a) {
  const int lane = threadIdx.x;
  int env = blockIdx.x;
  if constexpr (GEN) {
    const int cnt = a.w.gen_count[0];            // uniform (entries beyond the list were flagged by the main pass)
    if ((int)blockIdx.x >= cnt || blockIdx.x >= SG_GEN_LIST) return;
    env = a.w.gen_list[blockIdx.x];
    a.do_finish = 0; a.do_reset = 0; a.sens = nullptr;   // BEGIN only: the main pass has finished the previous substep and stored the state
  }
  if (env >= a.nenv) return;
  if (a.mask && !a.mask[env]) return;
  SG_T0();
  // the plan tables are read-only for the kernel's lifetime: through the constant address space a uniform index is a scalar load the
  // compiler may hoist and keep, not a vector load behind a full wait after every store
  const SG_CONSTAS SgPlanHeader& H = *(const SG_CONSTAS SgPlanHeader*)a.H;
  const int N = a.nelem, nv = a.nv, nu = a.nu, e0 = a.elem_dof0, nchain = a.nchain;
  const size_t S = 2 * (size_t)a.nenv;
  const double h = a.timestep;
  __shared__ Smem2<R, CPL, NB> Sm;
  const SG_CONSTAS double* const elemc = (const SG_CONSTAS double*)a.elem;
  [[maybe_unused]] const SG_CONSTAS int* const nbtabc = (const SG_CONSTAS int*)a.nbtab;
  auto EL = [&](int f, int e) { return elemc[(size_t)f * N + e]; };
  const int half = lane >> 5;
  const bool high = half != 0;
  const bool is_chain_lane = (lane & 31) == 0 && half < nchain;
  // the chains' model constants are read straight from the plan header (a few cached loads per wavefront); the chain stage itself,
  // which read them hundreds of times, runs in sg_chain_kernel
  const SG_CONSTAS SgChain& C = H.chain[half < nchain ? half : 0];
  ChainLds2& CS = Sm.cs[half];
  SgWork& W = a.w;

  // status and pending are LOADED here and TESTED below, after the state loads have been issued: an early return on them would put
  // one memory round trip in front of every other load of the kernel (a wavefront lives ~30 us, a round trip costs 1 - 2)
  int status = W.status[env];  // sg_chain_kernel, which runs first, resets it at the start of a call
  const int pend = W.pending[env];

  double* gq = a.qpos + (size_t)env * nv;
  double* gv = a.qvel + (size_t)env * nv;
  double* gw = a.warm + (size_t)env * nv;
  const double kenv = a.kenv[env];
  const int kt0_masked = a.kmask_ten[a.t0_id];

  // ---------------- load state ----------------
  double qe[R], ve[R], we[R], ke[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    int e = r * 64 + lane;
    qe[r] = ve[r] = we[r] = ke[r] = 0;
    if (e < N) {
      if (a.do_reset) { qe[r] = EL(SGE_QPOS0, e); }
      else { qe[r] = gq[e0 + e]; ve[r] = gv[e0 + e]; we[r] = gw[e0 + e]; }
      ke[r] = a.kmask_jnt[e0 + e] ? kenv : EL(SGE_K0, e);
    }
  }
  // the loads of FINISH (solver result, smooth acceleration and force of the previous substep) and the import of the chain
  // hand-off record are issued here, together with the state: one memory latency instead of three in a row (the kernel waits for
  // memory two thirds of its time, profiles/r02)
  double ase[R], asme_p[R], fsm_p[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    const int e = r * 64 + lane;
    ase[r] = asme_p[r] = fsm_p[r] = 0;
    if (a.do_finish && e < N) {  // whether a substep is pending is tested below: the workspace words exist either way
      ase[r] = W.as[(size_t)env * N + e]; asme_p[r] = W.asme[(size_t)env * N + e];
      if (a.finish_integrate) fsm_p[r] = W.fsm[(size_t)env * N + e];
    }
  }
  if (a.do_begin && half < nchain) {  // the chain stage ran in sg_chain_kernel: import its hand-off record, the 32 lanes of a half sharing the loads
    const double* ch = W.chh + ((size_t)env * 2 + half) * SG_CHW;
    double* const kd = (double*)&Sm.K[half];
    double* const box = &Sm.boxp[half * SG_CG][0];
    double* const boxm = &Sm.boxm[half * SG_CG][0];
    double* const lim = CS.lim_sign;  // lim_sign, lim_R, lim_b, lim_f are contiguous, as SGH_LIMSIGN .. SGH_LIMF are
#pragma unroll
    for (int j0 = 0; j0 < SG_CHW; j0 += 32) {
      const int j = j0 + (lane & 31);
      const double v = j < SG_CHW ? ch[j] : 0.0;
      if (j >= SGH_QSM && j < SGH_QSM + SG_CD) CS.qacc_smooth[j - SGH_QSM] = v;
      else if (j >= SGH_K && j < SGH_K + 48) kd[j - SGH_K] = v;
      else if (j >= SGH_MINV && j < SGH_MINV + 16) CS.Minv[j - SGH_MINV] = v;
      else if (j >= SGH_V && j < SGH_V + SG_CD) CS.v[j - SGH_V] = v;
      else if (j >= SGH_W && j < SGH_W + SG_CD) CS.w[j - SGH_W] = v;
      else if (j >= SGH_BOX && j < SGH_BOX + 12 * SG_CG) {
        const int g = (j - SGH_BOX) / 12, k = (j - SGH_BOX) % 12;
        if (k < 3) box[3 * g + k] = v; else boxm[9 * g + k - 3] = v;
      }
      else if (j == SGH_LIMACT) CS.lim_active = (int)v;
      else if (j >= SGH_LIMSIGN && j < SGH_LIMSIGN + 4 * SG_MAXLIM) lim[j - SGH_LIMSIGN] = v;
    }
  }
  const bool dead = (status & (SG_FLAG_BADQPOS | SG_FLAG_BADQVEL | SG_FLAG_BADQACC)) != 0;
  if (dead) return;  // the env stopped integrating earlier in this call
  const bool fin = a.do_finish && pend;
  const double kt0 = kt0_masked ? kenv : H.t0_k0;
  __syncthreads();

  SG_T(0);
  // =============================== FINISH the previous substep ===============================
  if (fin) {
    int badacc = 0;
    double qacc_e[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
      int e = r * 64 + lane;
      qacc_e[r] = 0;
      if (e < N) {
        qacc_e[r] = asme_p[r] + ase[r];
        if (isbad(qacc_e[r])) badacc = 1;
      }
    }
    const bool anybadacc = __ballot(badacc) != 0;
    if (anybadacc) {
      status |= SG_FLAG_BADQACC;
    } else {
      double qa[R], yc[R];  // yc = coef / (m + h d): the tendon's column of (M + h B)^-1 J'
      double Sp = 0;
#pragma unroll
      for (int r = 0; r < R; r++) {
        int e = r * 64 + lane;
        we[r] = qacc_e[r];
        qa[r] = yc[r] = 0;
        if (e < N && a.finish_integrate) {
          double m = EL(SGE_MASS, e) + EL(SGE_ARMATURE, e);
          double den = m + h * EL(SGE_DAMPING, e);
          qa[r] = (fsm_p[r] + m * ase[r]) / den;
          if (H.t0_implicit) { const double cf = EL(SGE_COEF, e); yc[r] = cf / den; Sp += cf * qa[r]; }
        }
      }
      if (H.t0_implicit && a.finish_integrate) {
        // deviation D5: (M + h B + h c J'J) qacc = f by Sherman-Morrison, qacc = x - y (h c J x) / (1 + h c J y), x = (M + h B)^-1 f,
        // y = (M + h B)^-1 J' (the sliders' block of M is diagonal, J is zero on the fingers); oracle sgo_step
        const double kk = h * H.t0_damping * wave_sum2(Sp) / (1.0 + H.t0_hcT);
#pragma unroll
        for (int r = 0; r < R; r++) qa[r] -= yc[r] * kk;
      }
#pragma unroll
      for (int r = 0; r < R; r++) {
        int e = r * 64 + lane;
        if (e < N && a.finish_integrate) {
          ve[r] += h * qa[r];
          qe[r] += h * ve[r];
        }
      }
    }
    if (lane == 0) W.pending[env] = 0;
  }
  __syncthreads();

  SG_T(1);
  // =============================== BEGIN the next substep ===============================
  int flags = 0;
  if (a.do_begin && !(status & SG_FLAG_BADQACC)) {
    int bad = 0;
#pragma unroll
    for (int r = 0; r < R; r++) bad |= (isbad(qe[r]) ? SG_FLAG_BADQPOS : 0) | (isbad(ve[r]) ? SG_FLAG_BADQVEL : 0);
    if (__ballot(bad != 0)) {
      flags |= (__ballot(bad & 1) ? 1 : 0) | (__ballot(bad & 2) ? 2 : 0);
    } else {
      // ---- chains: imported at the top of the kernel ----
      // ---- elements ----
      double invm[R], asme[R], coef[R];
      double L0p = 0, Ldp = 0;
#pragma unroll
      for (int r = 0; r < R; r++) {
        int e = r * 64 + lane;
        coef[r] = e < N ? EL(SGE_COEF, e) : 0.0;
        L0p += coef[r] * qe[r]; Ldp += coef[r] * ve[r];
      }
      const double L0 = wave_sum2(L0p), Ld = wave_sum2(Ldp);
      const double frc_t0 = -kt0 * (L0 - H.t0_lspring) - H.t0_damping * Ld;
      int unsupported = 0, ns0 = 0, ns1 = 0, touch = 0;
      bool special = false;
      {
        double cpos[R][3];
#pragma unroll
        for (int r = 0; r < R; r++) {
          int e = r * 64 + lane;
          invm[r] = asme[r] = 0;
          cpos[r][0] = cpos[r][1] = cpos[r][2] = 1e30;
          if (e < N) {
            double ax[3] = {EL(SGE_AX, e), EL(SGE_AY, e), EL(SGE_AZ, e)}, m = EL(SGE_MASS, e);
            double bias = -m * dot3(H.gravity, ax);
            double f = -ke[r] * (qe[r] - EL(SGE_SPRINGREF, e)) - EL(SGE_DAMPING, e) * ve[r] + coef[r] * frc_t0 - bias;
            invm[r] = 1.0 / (m + EL(SGE_ARMATURE, e));
            asme[r] = f * invm[r];
            W.fsm[(size_t)env * N + e] = f; W.asme[(size_t)env * N + e] = asme[r];
            double dq = qe[r] - EL(SGE_QPOS0, e);
            cpos[r][0] = EL(SGE_GX, e) + ax[0] * dq; cpos[r][1] = EL(SGE_GY, e) + ax[1] * dq; cpos[r][2] = EL(SGE_GZ, e) + ax[2] * dq;
            if (!(qe[r] > EL(SGE_QLO, e) && qe[r] < EL(SGE_QHI, e))) unsupported = 1;
            Sm.ve[e] = ve[r]; Sm.asme[e] = asme[r]; Sm.we[e] = we[r];
            Sm.owner[0][e] = 0; Sm.owner[1][e] = 0;
            *(unsigned int*)&Sm.eslot[e][0] = 0u;
            Sm.as[e] = qe[r];  // scratch until recompute_a: the dense narrowphase below reads other lanes' slider positions
          }
        }
        SG_T(2);
        __syncthreads();
        int overflow = 0;
        // ---- broadphase: (box, element) pairs that pass MuJoCo's bounding-sphere filter and the grown-box test, listed in
        //      contact order (chain, box, element; the object's centre sphere precedes the box's capsules).  Real loops: unrolled,
        //      the inlined narrowphase copies push the kernel far beyond the instruction cache
        int np = 0;
#pragma unroll 1
        for (int b = 0; b < nchain * SG_CG; b++) {
          const int c = b / SG_CG, g = b % SG_CG;
          const SgChain& Cc = H.chain[c];
          if (g >= Cc.ngeom) continue;
          double bp[3], bm[9], sz[3];
#pragma unroll
          for (int k = 0; k < 3; k++) { bp[k] = Sm.boxp[b][k]; sz[k] = Cc.g_size[g][k]; }
#pragma unroll
          for (int k = 0; k < 9; k++) bm[k] = Sm.boxm[b][k];
          const double rb = Cc.g_rbound[g];
          if (H.has_center) {
            double dif[3] = {bp[0] - H.center_pos[0], bp[1] - H.center_pos[1], bp[2] - H.center_pos[2]}, bound = H.center_radius + rb + H.con_margin;
            if (dot3(dif, dif) <= bound * bound) {
              if (lane == 0) Sm.pairs[np] = (unsigned short)((b << 12) | SG_PAIR_CENTER);
              np++;
            }
          }
#pragma unroll
          for (int r = 0; r < R; r++) {
            const int e = r * 64 + lane;
            // bounding spheres, then the capsule centre against the box grown by the capsule's bounding radius in the box frame --
            // still conservative, so the contact set is unchanged
            double dif[3] = {cpos[r][0] - bp[0], cpos[r][1] - bp[1], cpos[r][2] - bp[2]}, bound = H.cap_rbound + rb + H.con_margin;
            bool near = e < N && dot3(dif, dif) <= bound * bound;
            if (near) {
              double lc[3], grow = H.cap_rbound + H.con_margin;
              mulmatT3(lc, bm, dif);
              near = fabs(lc[0]) <= sz[0] + grow && fabs(lc[1]) <= sz[1] + grow && fabs(lc[2]) <= sz[2] + grow;
              if (near) {  // third filter: the capsule's own extent along the box axes (|half segment| + radius) instead of its bounding radius
                const double cax[3] = {EL(SGE_CX, e), EL(SGE_CY, e), EL(SGE_CZ, e)}, rm = H.cap_radius + H.con_margin;
                double hb[3];
                mulmatT3(hb, bm, cax);
                near = fabs(lc[0]) <= sz[0] + rm + H.cap_hl * fabs(hb[0]) && fabs(lc[1]) <= sz[1] + rm + H.cap_hl * fabs(hb[1]) &&
                       fabs(lc[2]) <= sz[2] + rm + H.cap_hl * fabs(hb[2]);
              }
            }
            const unsigned long long m = __ballot(near);
            if (near) Sm.pairs[np + lanes_below2(m)] = (unsigned short)((b << 12) | e);
            np += __popcll(m);
          }
        }
        __syncthreads();
        // ---- narrowphase over the dense pair list, 64 pairs per pass (every lane works; the per-box loop ran 8 passes at 10-30 %
        //      lane occupancy), then ordered compaction into the two finger streams
        int nsc[SG_MAXCH] = {0, 0};
#pragma unroll 1
        for (int p0 = 0; p0 < np; p0 += 64) {
          const bool have = p0 + lane < np;
          const int code = have ? (int)Sm.pairs[p0 + lane] : 0, b = code >> 12, e = code & 0xFFF, c = b / SG_CG, g = b % SG_CG;
          const bool is_center = have && e == SG_PAIR_CENTER;
          ConRec r0, r1;
          bool v0 = false, v1 = false;
          double bp[3], bm[9], sz[3];
#pragma unroll
          for (int k = 0; k < 3; k++) { bp[k] = Sm.boxp[b][k]; sz[k] = H.chain[c].g_size[g][k]; }
#pragma unroll
          for (int k = 0; k < 9; k++) bm[k] = Sm.boxm[b][k];
          if (__ballot(is_center)) {
            if (is_center) v0 = sphere_box(H.center_pos, H.center_radius, bp, bm, sz, H.con_margin, r0) && r0.dist < H.con_margin;
          }
          if (have && !is_center) {
            const double dq = Sm.as[e] - EL(SGE_QPOS0, e);
            double cp[3] = {EL(SGE_GX, e) + EL(SGE_AX, e) * dq, EL(SGE_GY, e) + EL(SGE_AY, e) * dq, EL(SGE_GZ, e) + EL(SGE_AZ, e) * dq};
            double cax[3] = {EL(SGE_CX, e), EL(SGE_CY, e), EL(SGE_CZ, e)};
            int mk = capsule_box(cp, cax, H.cap_radius, H.cap_hl, bp, bm, sz, H.con_margin, r0, r1);
            v0 = (mk & 1) && r0.dist < H.con_margin;
            v1 = (mk & 2) && r1.dist < H.con_margin;
          }
          const int n = (int)v0 + (int)v1;
#pragma unroll
          for (int cc = 0; cc < SG_MAXCH; cc++) {
            const bool mine = have && c == cc;
            const unsigned long long m1 = __ballot(mine && n >= 1), m2 = __ballot(mine && n >= 2);
            const int base = nsc[cc] + lanes_below2(m1) + lanes_below2(m2);
            if (mine && v0 && base < 32 * CPL) {
              StageRec2& s = Sm.stage[cc][base];
              s.dist = r0.dist; s.sl = is_center ? -1 : e; s.box = g;
              for (int q = 0; q < 3; q++) { s.pos[q] = r0.pos[q]; s.n[q] = r0.n[q]; }
            }
            if (mine && v1 && base + (int)v0 < 32 * CPL) {
              StageRec2& s = Sm.stage[cc][base + (int)v0];
              s.dist = r1.dist; s.sl = e; s.box = g;
              for (int q = 0; q < 3; q++) { s.pos[q] = r1.pos[q]; s.n[q] = r1.n[q]; }
            }
            if (mine && n > 0 && !is_center) {
              const int room = 32 * CPL - base, nst = n < room ? n : (room > 0 ? room : 0);
              static_assert(32 * CPL <= 64, "a contact slot index must fit 6 bits");
              Sm.eslot[e][b] = nst ? (unsigned char)(base | (nst << 6)) : (unsigned char)0;
            }
            nsc[cc] += __popcll(m1) + __popcll(m2);
            if (nsc[cc] > 32 * CPL) { nsc[cc] = 32 * CPL; overflow = 1; }
          }
          if (n > 0 && !is_center) Sm.owner[c][e] = 1;
#pragma unroll
          for (int bb = 0; bb < SG_MAXCH * SG_CG; bb++)
            if (__ballot(n > 0 && b == bb)) touch |= 1 << bb;
        }
        ns0 = nsc[0]; ns1 = nsc[1];
#ifdef SG_SECTION_PROF
        if (lane == 0) { atomicAdd(&a.w.secprof[30], (unsigned long long)np); atomicAdd(&a.w.secprof[31], (unsigned long long)((np + 63) / 64)); }
#endif
        if (overflow) flags |= SG_FLAG_CONTACTFULL;
      }
      SG_T(3);
      {  // envelope checks (same pairs as the fused kernel).  Finger box against static box (lanes 0 .. npairs-1) and finger box
         // against finger box (lanes 32 .. 35) go through ONE separating-axis test: each lane sets up its pair, then all of
         // them run the 15 axes together (two copies of the test, one per kind of pair, ran one after the other before)
        int nb = nchain * SG_CG, npairs = nb * H.nstatic;
        const double *p1 = nullptr, *R1 = nullptr, *s1 = nullptr, *p2 = nullptr, *R2 = nullptr, *s2 = nullptr;
        double bd = 0;
        bool pair = false;
        if (lane < npairs) {
          int b = lane / H.nstatic, s = lane % H.nstatic, c = b / SG_CG, g = b % SG_CG;
          if (g < H.chain[c].ngeom) {
            pair = true;
            p1 = Sm.boxp[b]; R1 = Sm.boxm[b]; s1 = H.chain[c].g_size[g];
            p2 = H.st_pos[s]; R2 = H.st_mat[s]; s2 = H.st_size[s];
            bd = H.chain[c].g_rbound[g] + H.st_rbound[s];
          }
        } else if (lane >= 32 && lane < 32 + SG_CG * SG_CG && nchain == 2) {
          int g = (lane - 32) / SG_CG, g2 = (lane - 32) % SG_CG;
          if (g < H.chain[0].ngeom && g2 < H.chain[1].ngeom) {
            int b = g, b2 = SG_CG + g2;
            pair = true;
            p1 = Sm.boxp[b]; R1 = Sm.boxm[b]; s1 = H.chain[0].g_size[g];
            p2 = Sm.boxp[b2]; R2 = Sm.boxm[b2]; s2 = H.chain[1].g_size[g2];
            bd = H.chain[0].g_rbound[g] + H.chain[1].g_rbound[g2];
          }
        }
        if (pair) {
          double dif[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
          if (dot3(dif, dif) <= bd * bd && box_box_overlap(p1, R1, s1, p2, R2, s2, 0)) unsupported = 1;
        }
        if (lane >= 48 && lane < 48 + SG_MAXCH * SG_CG && H.has_plane) {
          int b = lane - 48, c = b / SG_CG, g = b % SG_CG;
          if (c < nchain && g < H.chain[c].ngeom) {
            double dif[3] = {Sm.boxp[b][0] - H.plane_pos[0], Sm.boxp[b][1] - H.plane_pos[1], Sm.boxp[b][2] - H.plane_pos[2]}, ext = 0;
            for (int k = 0; k < 3; k++)
              ext += H.chain[c].g_size[g][k] * fabs(H.plane_normal[0] * Sm.boxm[b][k] + H.plane_normal[1] * Sm.boxm[b][3 + k] + H.plane_normal[2] * Sm.boxm[b][6 + k]);
            if (dot3(dif, H.plane_normal) - ext <= 0) unsupported = 1;
          }
        }
        special = __ballot(unsupported) != 0;
      }
      SG_T(4);
      __syncthreads();
      // a pair outside the fast path's two kinds is within reach: this substep's contacts are rebuilt as ONE ordered list over all
      // candidate pairs (sg_gen_phase; the rows pipeline's solver sweeps it as one stream).  The other pipelines flag the env.
      int ngen = 0;
      [[maybe_unused]] GenLds GL;
      if (special) {
        if (!a.rowlayout) flags |= SG_FLAG_UNSUPPORTED_PAIR;
        else if constexpr (!GEN) {
          int at = 0;   // hand the env to the general pass (sg_phase_kernel<.., true>)
          if (lane == 0) {
            at = atomicAdd(a.w.gen_count, 1);
            if (at < SG_GEN_LIST) a.w.gen_list[at] = env;
          }
          if (__shfl(at, 0) >= SG_GEN_LIST) flags |= SG_FLAG_UNSUPPORTED_PAIR;   // more envs on the general path than the pass takes
        } else {
          // the contact staging area in this mode: [0, 8 M) doubles the contact list (M = SG_GEN_MAXCON records of 8 doubles), then 96
          // doubles of box - box work space and the two chains' J' f sums; once the rows are built the list's place is taken by
          // the per-contact pushes [M], slider indices [M ints] and the per-element sums [R * 64]
          static_assert(sizeof(StageRec2) == 64 && sizeof(Sm.stage) >= 8 * (8 * SG_GEN_MAXCON + 96 + SG_MAXCH * SG_CD) &&
                        8 * SG_GEN_MAXCON >= SG_GEN_MAXCON + SG_GEN_MAXCON / 2 + R * 64, "the general path's lists live in the contact staging area");
          GL.stage = &Sm.stage[0][0];
          GL.gas = (double*)&Sm.stage[0][0] + SG_GEN_MAXCON + SG_GEN_MAXCON / 2;
          GL.tmp = (double*)&Sm.stage[0][0] + 8 * SG_GEN_MAXCON; GL.gg = GL.tmp + 96;
          GL.boxp = Sm.boxp; GL.boxm = Sm.boxm; GL.K = Sm.K; GL.cs = Sm.cs;
          GL.qe = Sm.as; GL.ve = Sm.ve; GL.asme = Sm.asme; GL.we = Sm.we;
          int gfl = 0, gtouch = 0;
          ngen = sg_gen_phase(a, *a.H, env, GL, &gfl, &gtouch);
          flags |= gfl;
          touch = gtouch;
          ns0 = ns1 = 0;                       // no contact stays on the per-finger streams
#pragma unroll
          for (int r = 0; r < R; r++) {
            const int e = r * 64 + lane;
            if (e < N) { *(unsigned int*)&Sm.eslot[e][0] = 0u; Sm.owner[0][e] = 0; Sm.owner[1][e] = 0; }
          }
          __syncthreads();
        }
      }
      int shared_slider = 0;
#pragma unroll
      for (int r = 0; r < R; r++) {
        int e = r * 64 + lane;
        if (e < N && Sm.owner[0][e] && Sm.owner[1][e]) shared_slider = 1;
      }
      shared_slider = __ballot(shared_slider) != 0;

      // ---- contact rows: built one slot at a time and exported at once; only a 7-double summary per slot stays
      //      in registers for the warmstart test (g = Jf' f, Js.f, invm, f.(R f/2 + b), slider index)
      const int myn = high ? ns1 : ns0;
      const size_t st = 2 * (size_t)env + half;
      const int nwb = (a.nenv + SG_EPW - 1) / SG_EPW;
      double cg[CPL][SG_CD], cjsf[CPL], cinvm[CPL], ccost0[CPL];
      int csl_[CPL];
#pragma unroll
      for (int k = 0; k < CPL; k++) {
        int i = (lane & 31) + 32 * k;
        csl_[k] = -1; cjsf[k] = cinvm[k] = ccost0[k] = 0;
#pragma unroll
        for (int d = 0; d < SG_CD; d++) cg[k][d] = 0;
        double2 fp[4][SG_RK / 2];  // my contact's three rows + the quad's fourth lane in the solver's field pairs (row layout only)
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
          for (int pr = 0; pr < SG_RK / 2; pr++) fp[r][pr] = make_double2(0.0, 0.0);
        if (i < myn) {
          Contact c;
          const StageRec2& sr = Sm.stage[half][i];
          ConRec rec;
          rec.dist = sr.dist;
          for (int q = 0; q < 3; q++) { rec.pos[q] = sr.pos[q]; rec.n[q] = sr.n[q]; }
          int sl = sr.sl, g = sr.box, bi = C.g_body[g], nd = chain_ndof_of_body(bi);
          double ax[3] = {0, 0, 0}, ve_ = 0, as_ = 0, we_ = 0, im = 0, bw = 0;
          if (sl >= 0) {
            ax[0] = EL(SGE_AX, sl); ax[1] = EL(SGE_AY, sl); ax[2] = EL(SGE_AZ, sl);
            ve_ = Sm.ve[sl]; as_ = Sm.asme[sl]; we_ = Sm.we[sl];
            im = 1.0 / (EL(SGE_MASS, sl) + EL(SGE_ARMATURE, sl)); bw = EL(SGE_BINVW, sl);
          }
          contact_build(c, rec, Sm.K[half], nd, CS.Minv, CS.v, CS.qacc_smooth, CS.w, C.b_invw_tran[bi], sl, ax, ve_, as_, we_, im, bw, *a.H);
          csl_[k] = sl; cinvm[k] = c.invm;
          cjsf[k] = c.Js[0] * c.f[0] + c.Js[1] * c.f[1] + c.Js[2] * c.f[2];
          Sm.cval[half][i] = c.invm * cjsf[k];
          ccost0[k] = c.f[0] * (0.5 * c.R * c.f[0] + c.b[0]) + c.f[1] * (0.5 * c.R * c.f[1] + c.b[1]) + c.f[2] * (0.5 * c.R * c.f[2] + c.b[2]);
#pragma unroll
          for (int d = 0; d < SG_CD; d++) cg[k][d] = c.Jf[0][d] * c.f[0] + c.Jf[1][d] * c.f[1] + c.Jf[2][d] * c.f[2];
          SG_T(24);
          if (a.rowlayout) {
            const double S11 = c.A[3] * H.con_mu[0] * H.con_mu[0], S22 = c.A[5] * H.con_mu[1] * H.con_mu[1], S12 = c.A[4] * H.con_mu[0] * H.con_mu[1];
            const double det = S11 * S22 - S12 * S12, di = det < 1e-10 ? 0.0 : sg_div(1.0, det);
            const double P11 = S22 * di, P22 = S11 * di, P12 = -S12 * di;
            // eigen-decomposition of the (friction-scaled) block S = Q diag(e1, e2) Q', Q = [[cs, sn], [-sn, cs]] (one Jacobi rotation):
            // constant over the solve, so mju_QCQP2's Newton iteration in the solver runs in these coordinates (sg_pgs_rows_kernel)
            double ecs = 1.0, esn = 0.0, ee1 = S11, ee2 = S22;
            if (fabs(S12) > 1e-300) {
              const double tau = (S22 - S11) / (2.0 * S12), tt = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
              ecs = 1.0 / sqrt(1.0 + tt * tt); esn = tt * ecs;
              ee1 = S11 - tt * S12; ee2 = S22 + tt * S12;
            }
            const double Afull[3][3] = {{c.A[0], c.A[1], c.A[2]}, {c.A[1], c.A[3], c.A[4]}, {c.A[2], c.A[4], c.A[5]}};
            double Wm[3][SG_CD];  // W_r = M^-1 J_F[r]': lane q of the quad keeps (W_0[q], W_1[q], W_2[q]), its column of the finger update
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
              for (int d = 0; d < SG_CD; d++) {
                double sw = 0;
#pragma unroll
                for (int e = 0; e < SG_CD; e++) sw += c.Jf[r][e] * CS.Minv[4 * e + d];
                Wm[r][d] = sw;
              }
#pragma unroll
            for (int r = 0; r < 4; r++) {
              double fld[SG_RK];
#pragma unroll
              for (int q = 0; q < SG_RK; q++) fld[q] = 0.0;
              if (r < 3) {
#pragma unroll
                for (int d = 0; d < SG_CD; d++) fld[d] = c.Jf[r][d];
                fld[4] = c.Js[r]; fld[5] = c.b[r]; fld[6] = c.f[r];
                fld[7] = Afull[r][0] * c.f[0] + Afull[r][1] * c.f[1] + Afull[r][2] * c.f[2];
                fld[8] = Afull[r][0]; fld[9] = Afull[r][1]; fld[10] = Afull[r][2];
                fld[11] = c.invm * c.Js[r];
                fld[15] = c.R;
              } else {  // the fourth lane carries what the three rows share (R is replicated on the row lanes: no broadcast)
                fld[0] = P11; fld[1] = P12; fld[2] = P22; fld[4] = __hiloint2double(0, sl);
                fld[8] = ee1; fld[9] = ee2; fld[10] = ecs; fld[11] = esn;
              }
              fld[12] = Wm[0][r]; fld[13] = Wm[1][r]; fld[14] = Wm[2][r];
#pragma unroll
              for (int pr = 0; pr < SG_RK / 2; pr++) fp[r][pr] = make_double2(fld[2 * pr], fld[2 * pr + 1]);
            }
          }
          if (!a.rowlayout) {
          double* ro = W.crec + SG_REC_INDEX(i, env / SG_EPW, 0, 2 * (env % SG_EPW) + half, nwb);
#pragma unroll
          for (int r = 0; r < 3; r++)
#pragma unroll
            for (int d = 0; d < SG_CD; d++) ro[(4 * r + d) * SG_SPW] = c.Jf[r][d];
#pragma unroll
          for (int r = 0; r < 3; r++) ro[(12 + r) * SG_SPW] = c.Js[r];
#pragma unroll
          for (int q = 0; q < 6; q++) ro[(15 + q) * SG_SPW] = c.A[q];
#pragma unroll
          for (int r = 0; r < 3; r++) ro[(21 + r) * SG_SPW] = c.b[r];
          ro[24 * SG_SPW] = c.R;
          ro[25 * SG_SPW] = c.invm;
#pragma unroll
          for (int r = 0; r < 3; r++) ro[(26 + r) * SG_SPW] = c.f[r];
          ((int*)(ro + 29 * SG_SPW))[0] = sl;
          }
        }
        // ---- export in row layout, transposed through LDS so that every store instruction writes whole 128-byte lines: the line
        //      (slot, pair) of this env is the 8 lanes (finger, row) x 16 B, held by two contact lanes (fingers 0 and 1 of slot i).
        //      The staging area is the part of Sm.stage this pass has consumed (slots 32 k .. 32 k + 31 of both fingers: 2 x 2 KB).
        const int nk = (ns0 > ns1 ? ns0 : ns1) - 32 * k;  // slots of this pass that any finger uses
        if (a.rowlayout && nk > 0) {
          const int nwb8 = (a.nenv + 7) / 8;
          const int li = lane & 31;
          auto X = [&](int sl_, int L) -> double2* {                                   // entry (slot, lane-of-8) for the reads
            return (double2*)&Sm.stage[sl_ >> 4][32 * k] + ((sl_ & 15) * 8 + L);
          };
          __syncthreads();  // every lane has copied its stage record
          // my entries: (slot li, lane-of-8 4 half + r); slot li lives in piece li >> 4
          double2* const e0 = (double2*)&Sm.stage[li >> 4][32 * k] + ((li & 15) * 8 + 4 * half);
#pragma unroll
          for (int pr = 0; pr < SG_RK / 2; pr++) {
            e0[0] = fp[0][pr]; e0[1] = fp[1][pr]; e0[2] = fp[2][pr]; e0[3] = fp[3][pr];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; j++) {
              if (8 * j >= nk) break;  // 8 slots per store instruction
              const int e = j * 64 + lane, s_ = e >> 3, L = e & 7;
              if (s_ < nk) *(double2*)(W.crow + SG_ROW_INDEX(32 * k + s_, env >> 3, 2 * pr, 8 * (env & 7) + L, nwb8)) = *X(s_, L);
            }
            __syncthreads();
          }
        }
      }
      SG_T(5);
      // ---- equality rows ----
      double eqR[R], eqb[R], eqf[R];
      double tbp = 0, tjp = 0, tAp = 0;
#pragma unroll
      for (int r = 0; r < R; r++) {
        int e = r * 64 + lane;
        eqR[r] = 1; eqb[r] = 0; eqf[r] = 0;
        if (e < N) {
          double pos = qe[r] - EL(SGE_QPOS0, e), imp = impedance(H.eqj_solimp, pos, 0);
          eqR[r] = fmax(SG_MINVAL, (1 - imp) / imp * EL(SGE_INVW, e));
          double aref = -H.eqj_B * ve[r] - H.eqj_K * imp * pos;
          eqb[r] = asme[r] - aref;
          eqf[r] = -(we[r] - aref) / eqR[r];
          tbp += coef[r] * asme[r]; tjp += coef[r] * we[r]; tAp += coef[r] * coef[r] * invm[r];
        }
      }
      // ---- neighbour rows (slider e = slider e2, J = +1 / -1): built by the lane of their first element, up to three each.
      //      b and R go straight to the workspace; the warmstart force lives in LDS (Sm.nbf, by row id) for the gathers below
      const int nnb = NB ? H.nnb : 0;
      constexpr int ND = NB ? 3 : 0;  // neighbour rows per element (loops over d vanish without them)
      int nbe2[R][3], nbid[R][3];
      double nbc0[R][3];  // f (R f / 2 + b)
#pragma unroll
      for (int r = 0; r < R; r++)
#pragma unroll
        for (int d = 0; d < 3; d++) { nbe2[r][d] = -1; nbid[r][d] = -1; nbc0[r][d] = 0; }
      if constexpr (NB) {
#pragma unroll
        for (int r = 0; r < R; r++) {
          const int e = r * 64 + lane;
          if (e < N) {
#pragma unroll
            for (int d = 0; d < 3; d++) {
              const int e2 = nbtabc[d * N + e];
              if (e2 >= 0) {
                const int id = nbtabc[(3 + d) * N + e];
                const double pos = (qe[r] - EL(SGE_QPOS0, e)) - (Sm.as[e2] - EL(SGE_QPOS0, e2)), imp = impedance(H.eqj_solimp, pos, 0);
                const double Rr = fmax(SG_MINVAL, (1 - imp) / imp * (EL(SGE_INVW, e) + EL(SGE_INVW, e2)));
                const double aref = -H.eqj_B * (ve[r] - Sm.ve[e2]) - H.eqj_K * imp * pos;
                const double bb = (asme[r] - Sm.asme[e2]) - aref, ff = -((we[r] - Sm.we[e2]) - aref) / Rr;
                nbe2[r][d] = e2; nbid[r][d] = id; nbc0[r][d] = ff * (0.5 * Rr * ff + bb);
                Sm.nbf[id] = ff;
                W.nbb[(size_t)env * 3 * N + id] = bb; W.nbR[(size_t)env * 3 * N + id] = Rr;
              }
            }
          }
        }
      }
      const double tpos = L0 - H.t0_L0, timp = impedance(H.eqt_solimp, tpos, 0), tR = fmax(SG_MINVAL, (1 - timp) / timp * H.eqt_invw);
      const double taref = -H.eqt_B * Ld - H.eqt_K * timp * tpos;
      const double tb = wave_sum2(tbp) - taref, tjar = wave_sum2(tjp) - taref, tA = wave_sum2(tAp) + tR;
      double tf = -tjar / tR;
      const int nmaxs = ns0 > ns1 ? ns0 : ns1;

      double aF[SG_CD];
      auto recompute_a = [&]() {
        __syncthreads();
        // slider accelerations M^-1 J' f: every element lane adds the pushes of its own contacts, finger 0's slots then finger 1's,
        // ascending -- the order of the solver's stream sweep (a serial loop over all contact slots used to do this)
#pragma unroll
        for (int r = 0; r < R; r++) {
          int e = r * 64 + lane;
          if (e < N) {
            double fe = eqf[r] + coef[r] * tf;
            if constexpr (NB) {  // + its own neighbour rows (J = +1), - the rows that have it as second joint (J = -1), in row order per side
#pragma unroll
              for (int d = 0; d < 3; d++) if (nbid[r][d] >= 0) fe += Sm.nbf[nbid[r][d]];
#pragma unroll
              for (int d = 0; d < 3; d++) { const int ii = nbtabc[(6 + d) * N + e]; if (ii >= 0) fe -= Sm.nbf[ii]; }
            }
            double as_ = invm[r] * fe;
#pragma unroll
            for (int cb = 0; cb < SG_MAXCH * SG_CG; cb++) {
              const int u = Sm.eslot[e][cb], i0 = u & 0x3F, nst = u >> 6;
              if (nst >= 1) as_ += Sm.cval[cb / SG_CG][i0];
              if (nst >= 2) as_ += Sm.cval[cb / SG_CG][i0 + 1];
            }
            if constexpr (GEN)
              if (ngen) as_ += GL.gas[e];   // general contact path: the pushes of the env's one contact list on this slider
            Sm.as[e] = as_;
          }
        }
        __syncthreads();
        double g[SG_CD] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < CPL; k++)
#pragma unroll
          for (int d = 0; d < SG_CD; d++) g[d] += cg[k][d];
        if (is_chain_lane) {
          const int la = CS.lim_active;
#pragma unroll
          for (int k = 0; k < SG_MAXLIM; k++)
            if (la >> k & 1) g[k / 2] += CS.lim_sign[k] * CS.lim_f[k];
          if constexpr (GEN)
            if (ngen) {
#pragma unroll
              for (int d = 0; d < SG_CD; d++) g[d] += GL.gg[half * SG_CD + d];
            }
        }
#pragma unroll
        for (int d = 0; d < SG_CD; d++) {
          double x = g[d];
#pragma unroll
          for (int o = 16; o > 0; o >>= 1) x += __shfl_xor(x, o);
          g[d] = x;
        }
#pragma unroll
        for (int a2 = 0; a2 < SG_CD; a2++) {
          double s2 = 0;
#pragma unroll
          for (int b2 = 0; b2 < SG_CD; b2++) s2 += CS.Minv[4 * a2 + b2] * g[b2];
          aF[a2] = s2;
        }
        __syncthreads();
      };
      recompute_a();
      SG_T(6);
      {
        double cp = 0, tJap = 0;
#pragma unroll
        for (int r = 0; r < R; r++) {
          int e = r * 64 + lane;
          if (e < N) {
            double ae = Sm.as[e]; cp += eqf[r] * (0.5 * (ae + eqR[r] * eqf[r]) + eqb[r]); tJap += coef[r] * ae;
#pragma unroll
            for (int d = 0; d < ND; d++)
              if (nbid[r][d] >= 0) cp += 0.5 * Sm.nbf[nbid[r][d]] * (ae - Sm.as[nbe2[r][d]]) + nbc0[r][d];
          }
        }
        double tJa = wave_sum2(tJap);
        if (lane == 0) cp += tf * (0.5 * (tJa + tR * tf) + tb);
        if (is_chain_lane) {
          const int la = CS.lim_active;
#pragma unroll
          for (int k = 0; k < SG_MAXLIM; k++)
            if (la >> k & 1) cp += CS.lim_f[k] * (0.5 * (CS.lim_sign[k] * aF[k / 2] + CS.lim_R[k] * CS.lim_f[k]) + CS.lim_b[k]);
        }
#pragma unroll
        for (int k = 0; k < CPL; k++)
          if ((lane & 31) + 32 * k < myn) {
            // sum_r f_r (J_r a / 2 + R f_r / 2 + b_r) = (g.aF + (Js.f) a_s) / 2 + f.(R f / 2 + b)
            double as_ = csl_[k] >= 0 ? Sm.as[csl_[k]] : 0.0, ga = 0;
#pragma unroll
            for (int d = 0; d < SG_CD; d++) ga += cg[k][d] * aF[d];
            cp += 0.5 * (ga + cjsf[k] * as_) + ccost0[k];
          }
        [[maybe_unused]] double aF2[SG_MAXCH][SG_CD];   // both chains' accelerations on every lane (general contact path only)
        if constexpr (GEN)
          if (ngen) {
#pragma unroll
            for (int d = 0; d < SG_CD; d++) {
              const double other = __shfl_xor(aF[d], 32);
              aF2[0][d] = high ? other : aF[d]; aF2[1][d] = high ? aF[d] : other;
            }
            cp += sg_gen_cost(a, env, ngen, aF2, Sm.as, 0);
          }
        double cost = wave_sum2(cp);
        if (cost > 0) {
          if constexpr (GEN)
            if (ngen) {
              sg_gen_cost(a, env, ngen, aF2, Sm.as, 1);
              for (int e = lane; e < N; e += 64) GL.gas[e] = 0.0;
              if (lane < SG_MAXCH * SG_CD) GL.gg[lane] = 0.0;
            }
#pragma unroll
          for (int r = 0; r < R; r++) {
            eqf[r] = 0;
#pragma unroll
            for (int d = 0; d < ND; d++) if (nbid[r][d] >= 0) Sm.nbf[nbid[r][d]] = 0.0;
          }
          tf = 0;
          if (is_chain_lane) {
#pragma unroll
            for (int k = 0; k < SG_MAXLIM; k++) CS.lim_f[k] = 0;
          }
#pragma unroll
          for (int k = 0; k < CPL; k++) {
            int i = (lane & 31) + 32 * k;
            cjsf[k] = 0;
#pragma unroll
            for (int d = 0; d < SG_CD; d++) cg[k][d] = 0;
            if (i < myn) {
              Sm.cval[half][i] = 0.0;
              if (!a.rowlayout) {
              double* ro = W.crec + SG_REC_INDEX(i, env / SG_EPW, 0, 2 * (env % SG_EPW) + half, nwb);
#pragma unroll
              for (int r = 0; r < 3; r++) ro[(26 + r) * SG_SPW] = 0.0;
              }
              if (a.rowlayout) {
#pragma unroll
                for (int r = 0; r < 3; r++)  // f and A f (one pair)
                  *(double2*)(W.crow + SG_ROW_INDEX(i, env >> 3, 6, 8 * (env & 7) + 4 * half + r, (a.nenv + 7) / 8)) = make_double2(0.0, 0.0);
              }
            }
          }
          __syncthreads();
          recompute_a();
        }
      }
      SG_T(7);
      // ---- export the rest of the constraint problem ----
#pragma unroll
      for (int r = 0; r < R; r++) {
        int e = r * 64 + lane;
        if (e < N) {
          size_t o = (size_t)env * N + e;
          W.as[o] = Sm.as[e]; W.eqf[o] = eqf[r]; W.eqb[o] = eqb[r]; W.eqR[o] = eqR[r];
#pragma unroll
          for (int d = 0; d < ND; d++)
            if (nbid[r][d] >= 0) W.nbf[(size_t)env * 3 * N + nbid[r][d]] = Sm.nbf[nbid[r][d]];
        }
      }
      if (is_chain_lane) {
        W.ns[st] = myn < SG_CAP ? myn : SG_CAP;
        W.lim_active[st] = CS.lim_active;
      }
      if (half < nchain) {  // the 32 lanes of a half write its chain's 4 x SG_MAXLIM limit values (contiguous in CS) and M^-1 J' f
        static_assert(4 * SG_MAXLIM == 32, "one limit value per lane of a half");
        const int i = lane & 31;
        W.lim[(size_t)i * S + st] = (&CS.lim_sign[0])[i];
        if (i < SG_CD) W.saF[(size_t)i * S + st] = i == 0 ? aF[0] : (i == 1 ? aF[1] : (i == 2 ? aF[2] : aF[3]));
      }
      if (lane == 0) {
        W.envh[(size_t)0 * a.nenv + env] = tb; W.envh[(size_t)1 * a.nenv + env] = tR;
        W.envh[(size_t)2 * a.nenv + env] = tA; W.envh[(size_t)3 * a.nenv + env] = tf;
        W.shared[env] = shared_slider;
        W.pending[env] = 1;
        W.ncon[env] = ns0 + ns1 + ngen;
        W.gen[env] = ngen;
        W.nefc[env] = N + nnb + 1 + 3 * (ns0 + ns1 + ngen) + __popc(Sm.cs[0].lim_active) + (nchain > 1 ? __popc(Sm.cs[1].lim_active) : 0);
        W.touch[env] = touch;
      }
    }
  }

  SG_T(8);
  // ---------------- store state ----------------
  __syncthreads();
#pragma unroll
  for (int r = 0; r < R; r++) {
    int e = r * 64 + lane;
    if (e < N) { gq[e0 + e] = qe[r]; gv[e0 + e] = ve[r]; gw[e0 + e] = we[r]; }
  }
  if (lane == 0) {
    if (status | flags) atomicOr(&W.status[env], status | flags);
  }
  SG_T(9);
  SG_TEND();
}

// ------------------------------------------------------------------------------------------------
// chain kernel: ONE LANE PER FINGER CHAIN (64 chains per wavefront, all with the same chain index).  The chain stage is a few thousand strictly serial
// instructions; inside the phase kernel it ran on 2 of 64 lanes of every env's wavefront, here 64 chains share one
// instruction stream.  FINISH: qacc of the chain, its sensors, warmstart, integration.  BEGIN: kinematics, mass matrix,
// bias, tendon/actuator, limit rows, box poses -> hand-off record (enum SGH_*) for the phase and PGS kernels.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void sg_chain_kernel(SgPhaseArgs a) {
  // wavefront 2 b + c holds chain c of envs 64 b .. 64 b + 63: the chain index is uniform over the wavefront, so the chain's model
  // constants (SgChain, ~230 doubles read all over the stage) are scalar loads / SGPR operands instead of per-lane vector loads
  const int lane = threadIdx.x, c = blockIdx.x & 1, env = (int)(blockIdx.x >> 1) * 64 + lane;
  if (blockIdx.x == 0 && lane == 0) a.w.gen_count[0] = 0;   // the phase kernel of this substep refills the general pass's list
  SG_T0();
  const SG_CONSTAS SgPlanHeader& H = *(const SG_CONSTAS SgPlanHeader*)a.H;
  const int nv = a.nv, nu = a.nu;
  const size_t S = 2 * (size_t)a.nenv;
  if (env >= a.nenv) return;
  const size_t st = 2 * (size_t)env + c;
  if (a.mask && !a.mask[env]) return;
  SgWork& W = a.w;
  if (a.first && c == 0) { W.status[env] = 0; if (!a.do_finish) W.pending[env] = 0; }
  if (c >= a.nchain) return;
  // loaded here, tested after the state loads have been issued (an early return would put a memory round trip in front of them)
  const int status = a.first ? 0 : W.status[env];
  const int pend = a.do_finish ? W.pending[env] : 0;
  const SgChain& C = H.chain[c];
  const double h = a.timestep;
  double* gq = a.qpos + (size_t)env * nv;
  double* gv = a.qvel + (size_t)env * nv;
  double* gw = a.warm + (size_t)env * nv;
  const double kenv = a.kenv[env];
  double q[SG_CD], v[SG_CD], w[SG_CD], kk[SG_CD], act = 0, ctrl = 0;
#pragma unroll
  for (int d = 0; d < SG_CD; d++) {
    int j = C.dof0 + d;
    if (a.do_reset) { q[d] = C.qpos0[d]; v[d] = 0; w[d] = 0; }
    else { q[d] = gq[j]; v[d] = gv[j]; w[d] = gw[j]; }
    kk[d] = a.kmask_jnt[j] ? kenv : C.stiffness[d];
  }
  if (C.has_act) {
    if (a.do_reset) a.ctrl[(size_t)env * nu + C.act_id] = 0;
    else { act = a.act[(size_t)env * nu + C.act_id]; ctrl = a.ctrl[(size_t)env * nu + C.act_id]; }
  }
  const double kten = C.has_ten ? (a.kmask_ten[C.ten_id] ? kenv : C.ten_k0) : 0.0;
  double* ch = W.chh + st * SG_CHW;
  bool bad_acc = false;
  if (status & (SG_FLAG_BADQPOS | SG_FLAG_BADQVEL | SG_FLAG_BADQACC)) return;

  SG_T(17);
  if (a.do_finish && pend) {
    double aF[SG_CD], qacc_c[SG_CD];
#pragma unroll
    for (int d = 0; d < SG_CD; d++) {
      aF[d] = W.saF[(size_t)d * S + st];
      qacc_c[d] = ch[SGH_QSM + d] + aF[d];
      if (isbad(qacc_c[d])) bad_acc = true;
    }
    ChainKin K;
    {
      double* kd = (double*)&K;
#pragma unroll
      for (int i = 0; i < 48; i++) kd[i] = ch[SGH_K + i];
    }
    if (a.sens) {
      ChainMotion Mo;
      chain_motion(C, K, v, qacc_c, H.gravity, Mo);
      double* so = a.sens + (size_t)env * a.sens_stride;
      for (int s = 0; s < C.nsite; s++) {
        int bi = C.s_body[s];
        double r3[3], sm[9], t[3], t2[3], acc[3], out[3], sbp[3], sbm[9], bw[3], bal[3];
        chain_body_pose(K, bi, sbp, sbm);
        mulmat3(r3, sbm, C.s_pos[s]);
        mulmat33(sm, sbm, C.s_mat[s]);
        for (int k = 0; k < 3; k++) { bw[k] = bi == 0 ? Mo.w[0][k] : Mo.w[SG_CB - 1][k]; bal[k] = bi == 0 ? Mo.al[0][k] : Mo.al[SG_CB - 1][k]; }
        if (C.s_gyro_adr[s] >= 0) {
          mulmatT3(out, sm, bw);
          for (int k = 0; k < 3; k++) so[C.s_gyro_adr[s] + k] = out[k];
        }
        if (C.s_acc_adr[s] >= 0) {
          for (int k = 0; k < 3; k++) acc[k] = bi == 0 ? Mo.a[0][k] : Mo.a[SG_CB - 1][k];
          cross3(t, bal, r3); addscl3(acc, t, 1);
          cross3(t, bw, r3); cross3(t2, bw, t); addscl3(acc, t2, 1);
          mulmatT3(out, sm, acc);
          for (int k = 0; k < 3; k++) so[C.s_acc_adr[s] + k] = out[k];
        }
      }
    }
    if (bad_acc) {
      atomicOr(&W.status[env], SG_FLAG_BADQACC);
    } else {
#pragma unroll
      for (int d = 0; d < SG_CD; d++) w[d] = qacc_c[d];
      if (a.finish_integrate) {
        bool damp = false;
#pragma unroll
        for (int d = 0; d < SG_CD; d++) damp |= C.damping[d] > 0;
        double qa[SG_CD];
        if (damp) {
          double MhB[16], MhBinv[16], rhs[SG_CD], Mm[16];
#pragma unroll
          for (int i = 0; i < 16; i++) { Mm[i] = ch[SGH_M + i]; MhB[i] = Mm[i]; }
#pragma unroll
          for (int d = 0; d < SG_CD; d++) MhB[5 * d] += h * C.damping[d];
          spd_inverse4(MhB, MhBinv);
#pragma unroll
          for (int a2 = 0; a2 < SG_CD; a2++) {
            double s2 = ch[SGH_QFRC + a2];
#pragma unroll
            for (int b2 = 0; b2 < SG_CD; b2++) s2 += Mm[4 * a2 + b2] * aF[b2];
            rhs[a2] = s2;
          }
#pragma unroll
          for (int a2 = 0; a2 < SG_CD; a2++) {
            double s2 = 0;
#pragma unroll
            for (int b2 = 0; b2 < SG_CD; b2++) s2 += MhBinv[4 * a2 + b2] * rhs[b2];
            qa[a2] = s2;
          }
        } else {
#pragma unroll
          for (int d = 0; d < SG_CD; d++) qa[d] = qacc_c[d];
        }
        act += h * ch[SGH_ACTDOT];
#pragma unroll
        for (int d = 0; d < SG_CD; d++) { v[d] += h * qa[d]; q[d] += h * v[d]; }
      }
    }
  }

  SG_T(18);
  if (a.do_begin && !bad_acc) {
    int bad = 0;
#pragma unroll
    for (int d = 0; d < SG_CD; d++) bad |= (isbad(q[d]) ? SG_FLAG_BADQPOS : 0) | (isbad(v[d]) ? SG_FLAG_BADQVEL : 0);
    if (bad) {
      atomicOr(&W.status[env], bad);
    } else {
      ChainKin K;
      ChainDyn D;
      chain_kinematics(C, q, K);
      SG_T(19);
      chain_dynamics(C, K, q, v, act, ctrl, kk, kten, H.gravity, D);
      SG_T(20);
      // hand-off record: assembled in registers and written as 16-byte stores (a lane's record is 1280 contiguous bytes; every store
      // instruction touches 64 different lines, so their number is what counts)
      double rec[SG_CHW];
#pragma unroll
      for (int i = 0; i < SG_CHW; i++) rec[i] = 0.0;
#pragma unroll
      for (int d = 0; d < SG_CD; d++) { rec[SGH_QSM + d] = D.qacc_smooth[d]; rec[SGH_QFRC + d] = D.qfrc_smooth[d]; rec[SGH_V + d] = v[d]; rec[SGH_W + d] = w[d]; }
      rec[SGH_ACTDOT] = D.act_dot;
#pragma unroll
      for (int i = 0; i < 16; i++) { rec[SGH_M + i] = D.M[i]; rec[SGH_MINV + i] = D.Minv[i]; W.sMinv[(size_t)i * S + st] = D.Minv[i]; }
      {
        const double* kd = (const double*)&K;
#pragma unroll
        for (int i = 0; i < 48; i++) rec[SGH_K + i] = kd[i];
      }
#pragma unroll
      for (int g = 0; g < SG_CG; g++) {
        double t[3] = {0, 0, 0}, bp_[3] = {0, 0, 0}, bm_[9], bm2[9];
#pragma unroll
        for (int k = 0; k < 9; k++) bm2[k] = 0;
        if (g < C.ngeom) {
          chain_body_pose(K, C.g_body[g], bp_, bm_);
          mulmat3(t, bm_, C.g_pos[g]);
          mulmat33(bm2, bm_, C.g_mat[g]);
        }
#pragma unroll
        for (int k = 0; k < 3; k++) rec[SGH_BOX + 12 * g + k] = bp_[k] + t[k];
#pragma unroll
        for (int k = 0; k < 9; k++) rec[SGH_BOX + 12 * g + 3 + k] = bm2[k];
      }
      SG_T(21);
      LimitRows L;
      limits_build(C, q, v, D.qacc_smooth, w, L);
      rec[SGH_LIMACT] = (double)L.active;
#pragma unroll
      for (int k = 0; k < SG_MAXLIM; k++) { rec[SGH_LIMSIGN + k] = L.sign[k]; rec[SGH_LIMR + k] = L.R[k]; rec[SGH_LIMB + k] = L.b[k]; rec[SGH_LIMF + k] = L.f[k]; }
      static_assert(SG_CHW % 2 == 0 && SGH_LIMF + SG_MAXLIM <= SG_CHW, "hand-off record layout");
#pragma unroll
      for (int i = 0; i < (SGH_LIMF + SG_MAXLIM + 1) / 2; i++) ((double2*)ch)[i] = make_double2(rec[2 * i], rec[2 * i + 1]);
    }
  }
  SG_T(22);
  // store the chain's state
#pragma unroll
  for (int d = 0; d < SG_CD; d++) { int j = C.dof0 + d; gq[j] = q[d]; gv[j] = v[d]; gw[j] = w[d]; }
  if (C.has_act) a.act[(size_t)env * nu + C.act_id] = act;
  SG_T(23);
  SG_TEND();
}

// ------------------------------------------------------------------------------------------------
// PGS kernel: 8 lanes per env, 8 envs per wavefront
// ------------------------------------------------------------------------------------------------
struct SgPgsArgs {
  const uint2* tab;       // the schedule as the solver's LDS table words (lane 2 b + h of a 16-lane group, block slot b; sg_api.hip)
  const SgEqSlot* sched;  // SgPlan::sched (equality-row schedule of neighbour-row models), nullptr when H.nnb == 0
  const int* nbtab;       // SgPlan::nbtab
  const SgPlanHeader* H;
  const double* elem;
  SgWork w;
  int nenv;
};

__global__ __launch_bounds__(64) void sg_pgs_kernel(SgPgsArgs a) {
  extern __shared__ double lds[];  // [8 envs][4 arrays][N] + invm[N] + coef[N] + limits
  const int lane = threadIdx.x, le = lane / SG_G, g = lane % SG_G;
  const int env = blockIdx.x * SG_EPW + le;
  const SG_CONSTAS SgPlanHeader& H = *(const SG_CONSTAS SgPlanHeader*)a.H;
  const int N = H.nelem;
  const size_t S = 2 * (size_t)a.nenv;
  const int nwb = (a.nenv + SG_EPW - 1) / SG_EPW;
  const SgWork& W = a.w;
  const double mu[2] = {H.con_mu[0], H.con_mu[1]}, pgs_scale = H.pgs_scale, tolerance = H.tolerance;
  const int max_iter = H.iterations;
  const bool valid = env < a.nenv && W.pending[env] != 0;
  if (!__ballot(valid)) return;
  double* Las = lds + (size_t)le * 4 * N;
  double *Lf = Las + N, *Lb = Lf + N, *LR = Lb + N;
  double* Linvm = lds + (size_t)SG_EPW * 4 * N;
  double* Lcoef = Linvm + N;
  double* Llim = Lcoef + N + (size_t)(le * 2) * 4 * SG_MAXLIM;  // per stream: sign, R, b, f x 8
  for (int j = lane; j < N; j += 64) { Linvm[j] = 1.0 / (a.elem[(size_t)SGE_MASS * N + j] + a.elem[(size_t)SGE_ARMATURE * N + j]); Lcoef[j] = a.elem[(size_t)SGE_COEF * N + j]; }
  if (valid)
    for (int j = g; j < N; j += SG_G) {
      size_t o = (size_t)env * N + j;
      Las[j] = W.as[o]; Lf[j] = W.eqf[o]; Lb[j] = W.eqb[o]; LR[j] = W.eqR[o];
    }
  const bool is_stream = valid && g < 2;
  const size_t st = 2 * (size_t)(env < a.nenv ? env : 0) + (g & 1);
  int ns = 0, lim_active = 0, shared = 0;
  double Minv[16], aF[SG_CD] = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 16; i++) Minv[i] = 0;
  double tb = 0, tR = 1, tA = 1, tf = 0;
  if (valid) {
    tb = W.envh[(size_t)0 * a.nenv + env]; tR = W.envh[(size_t)1 * a.nenv + env];
    tA = W.envh[(size_t)2 * a.nenv + env]; tf = W.envh[(size_t)3 * a.nenv + env];
    shared = W.shared[env];
  }
  double* mylim = Llim + (size_t)(g & 1) * 4 * SG_MAXLIM;
  if (is_stream) {
    ns = W.ns[st];
    lim_active = W.lim_active[st];
#pragma unroll
    for (int i = 0; i < 16; i++) Minv[i] = W.sMinv[(size_t)i * S + st];
#pragma unroll
    for (int d = 0; d < SG_CD; d++) aF[d] = W.saF[(size_t)d * S + st];
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int k = 0; k < SG_MAXLIM; k++) mylim[q * SG_MAXLIM + k] = W.lim[((size_t)q * SG_MAXLIM + k) * S + st];
  }
  __syncthreads();
  int nsmax = ns;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(nsmax, o); nsmax = t > nsmax ? t : nsmax; }
  const unsigned long long any_lim = __ballot(lim_active != 0);

  bool running = valid;
  int iters = 0;
  // my record column inside the wave's block; idle lanes read their env's stream (same address as its stream lane)
  // and store to the dummy block
  double* const rec0 = W.crec + SG_REC_INDEX(0, blockIdx.x, 0, 2 * le + (g & 1), nwb);
  double* const rec0_store = (valid && g < 2) ? rec0 : W.crec + SG_REC_INDEX(0, nwb, 0, lane % SG_SPW, nwb);
  const size_t slot_stride = (size_t)(nwb + 1) * SG_RF * SG_SPW;

  for (int it = 0; it < max_iter; it++) {
    if (!__ballot(running)) break;
    double imp_acc = 0, tJap = 0;
    if (running) {
      for (int j = g; j < N; j += SG_G) {
        double ae = Las[j], f = Lf[j], old = f, im = Linvm[j];
        double Rr = LR[j];
        imp_acc -= scalar_update(f, Lb[j], ae, Rr, im + Rr, false);
        ae += im * (f - old);
        Lf[j] = f; Las[j] = ae;
        tJap += Lcoef[j] * ae;
      }
    }
    {  // tendon row: sum over the env's 8 lanes
      double Ja = tJap;
#pragma unroll
      for (int o = 1; o < SG_G; o <<= 1) Ja += __shfl_xor(Ja, o);
      if (running) {
        double old = tf, tfn = tf;
        double ch = scalar_update(tfn, tb, Ja, tR, tA, false);
        if (g == 0) imp_acc -= ch;
        tf = tfn;
        double dft = tf - old;
        for (int j = g; j < N; j += SG_G) Las[j] += Linvm[j] * Lcoef[j] * dft;
      }
    }
    __syncthreads();
    // limits + contacts.  pass 0: stream 0 everywhere and stream 1 where the streams share no slider; pass 1: the rest
    for (int pass = 0; pass < 2; pass++) {
      const bool mine = is_stream && running && ((g == 0 || !shared) ? pass == 0 : pass == 1);
      if (!__ballot(mine)) continue;
      if (any_lim) {
#pragma unroll
        for (int k = 0; k < SG_MAXLIM; k++) {
          if (mine && (lim_active >> k & 1)) {
            const int d = k / 2;
            double f = mylim[3 * SG_MAXLIM + k], old = f, sg = mylim[k], Rr = mylim[SG_MAXLIM + k];
            imp_acc -= scalar_update(f, mylim[2 * SG_MAXLIM + k], sg * aF[d], Rr, Minv[5 * d] + Rr, true);
            mylim[3 * SG_MAXLIM + k] = f;
            double df = sg * (f - old);
#pragma unroll
            for (int q = 0; q < SG_CD; q++) aF[q] += Minv[4 * q + d] * df;
          }
        }
      }
      // Software pipeline: the record of contact i+1 is requested before contact i is updated, so the L2 / Infinity
      // Cache latency overlaps the update arithmetic.  Every lane issues the same loads and stores unconditionally
      // (idle lanes read their env's stream -- same addresses as the active lane -- and write to the dummy block), so
      // the compiler can count vmcnt and only waits for the previous batch of loads.
      const int nsl = mine ? ns : 0;
      auto load_rec = [&](Contact& c, int i) {
        const double* rec = rec0 + (size_t)i * slot_stride;
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int d = 0; d < SG_CD; d++) c.Jf[r][d] = rec[(4 * r + d) * SG_SPW];
#pragma unroll
        for (int r = 0; r < 3; r++) c.Js[r] = rec[(12 + r) * SG_SPW];
#pragma unroll
        for (int q = 0; q < 6; q++) c.A[q] = rec[(15 + q) * SG_SPW];
#pragma unroll
        for (int r = 0; r < 3; r++) c.b[r] = rec[(21 + r) * SG_SPW];
        c.R = rec[24 * SG_SPW];
        c.invm = rec[25 * SG_SPW];
#pragma unroll
        for (int r = 0; r < 3; r++) c.f[r] = rec[(26 + r) * SG_SPW];
        c.sl = ((const int*)(rec + 29 * SG_SPW))[0];
      };
      auto update_rec = [&](Contact& c, int i) {
        if (i < nsl) {
          double as_ = c.sl >= 0 ? Las[c.sl] : 0.0, df[3];
          imp_acc -= contact_update(c, aF, as_, mu, df);
          if (c.sl >= 0) Las[c.sl] = as_ + c.invm * (c.Js[0] * df[0] + c.Js[1] * df[1] + c.Js[2] * df[2]);
          double gg[SG_CD];
#pragma unroll
          for (int d = 0; d < SG_CD; d++) gg[d] = c.Jf[0][d] * df[0] + c.Jf[1][d] * df[1] + c.Jf[2][d] * df[2];
#pragma unroll
          for (int q = 0; q < SG_CD; q++) aF[q] += (Minv[4 * q] * gg[0] + Minv[4 * q + 1] * gg[1]) + (Minv[4 * q + 2] * gg[2] + Minv[4 * q + 3] * gg[3]);
        }
        double* recs = rec0_store + (size_t)i * slot_stride;
#pragma unroll
        for (int r = 0; r < 3; r++) recs[(26 + r) * SG_SPW] = c.f[r];
      };
      // two contacts per trip with the buffers swapping roles (no register copies); requesting records two updates
      // ahead (three buffers) measured no faster
      Contact ca, cb;
      load_rec(ca, 0);
      for (int i = 0; i < nsmax; i += 2) {
        load_rec(cb, i + 1 < SG_CAP ? i + 1 : i);
        update_rec(ca, i);
        load_rec(ca, i + 2 < SG_CAP ? i + 2 : i);
        update_rec(cb, i + 1);
      }
      __syncthreads();
    }
    double imp = imp_acc;
#pragma unroll
    for (int o = 1; o < SG_G; o <<= 1) imp += __shfl_xor(imp, o);
    if (running) {
      iters = it + 1;
      if (imp * pgs_scale < tolerance) running = false;
    }
  }
  __syncthreads();
  // ---- fresh M^-1 J' f from the final forces (same as the fused kernel's recompute) ----
  if (valid)
    for (int j = g; j < N; j += SG_G) Las[j] = Linvm[j] * (Lf[j] + Lcoef[j] * tf);
  __syncthreads();
  double gF[SG_CD] = {0, 0, 0, 0};
  for (int pass = 0; pass < 2; pass++) {  // stream 0 then stream 1: deterministic when they share a slider
    if (is_stream && g == pass) {
#pragma unroll
      for (int k = 0; k < SG_MAXLIM; k++)
        if (lim_active >> k & 1) gF[k / 2] += mylim[k] * mylim[3 * SG_MAXLIM + k];
      for (int i = 0; i < ns; i++) {
        const double* rec = rec0 + (size_t)i * slot_stride;
        double f0 = rec[26 * SG_SPW], f1 = rec[27 * SG_SPW], f2 = rec[28 * SG_SPW];
        int sl = ((const int*)(rec + 29 * SG_SPW))[0];
        if (sl >= 0) Las[sl] += rec[25 * SG_SPW] * (rec[12 * SG_SPW] * f0 + rec[13 * SG_SPW] * f1 + rec[14 * SG_SPW] * f2);
#pragma unroll
        for (int d = 0; d < SG_CD; d++) gF[d] += rec[d * SG_SPW] * f0 + rec[(4 + d) * SG_SPW] * f1 + rec[(8 + d) * SG_SPW] * f2;
      }
    }
    __syncthreads();
  }
  if (is_stream) {
#pragma unroll
    for (int q = 0; q < SG_CD; q++) {
      double s = 0;
#pragma unroll
      for (int d = 0; d < SG_CD; d++) s += Minv[4 * q + d] * gF[d];
      W.saF[(size_t)q * S + st] = s;
    }
  }
  if (valid) {
    for (int j = g; j < N; j += SG_G) W.as[(size_t)env * N + j] = Las[j];
    if (g == 0) W.iters[env] = iters;
  }
}

// ------------------------------------------------------------------------------------------------
// PGS kernel, row-parallel contact update: as sg_pgs_kernel (8 lanes per env, 8 envs per wavefront, joint-fix rows over
// the env's 8 lanes) but each finger stream is a QUAD of lanes: lane r < 3 owns row r of every contact (normal, tangent
// 1, tangent 2), so the 3 x 5 residual products, A f, A d, J' df ... take one instruction for the three rows instead of
// three, and the few cross-row sums / broadcasts are DPP quad permutes (no LDS).  A lone wavefront issues one instruction
// every ~6.3 cycles whatever it computes (scripts/ubench), so instructions per contact update are what this cuts.
// ------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double sg_dpp(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
template <int R_>
__device__ __forceinline__ double sg_qb(double x) { return sg_dpp<R_ * 0x55>(x); }  // value of quad lane R_ in all four lanes
__device__ __forceinline__ double sg_qsum(double x) {                                   // (x0 + x1) + (x2 + x3) in all four lanes
  x += sg_dpp<0xB1>(x);
  x += sg_dpp<0x4E>(x);
  return x;
}

// sum over the 8 lanes of an env's group, result in all 8: two quad steps and a mirror inside the 8-lane half row (all DPP)
__device__ __forceinline__ double sg_gsum8(double x) {
  x = sg_qsum(x);
  x += sg_dpp<0x141>(x);  // row_half_mirror: lane i <-> 7 - i, i.e. the other quad of the group (whose lanes all hold its sum)
  return x;
}

// sum over the 16 lanes of a DPP row, result in all 16
__device__ __forceinline__ double sg_gsum16(double x) {
  x = sg_gsum8(x);
  x += sg_dpp<0x140>(x);  // row_mirror: lane i <-> 15 - i, i.e. the other half row (whose lanes all hold its sum)
  return x;
}

// NB = true: the model has neighbour equality rows (slider e = slider e2).  The joint-fix rows are then no longer mutually
// independent, and the equality block of a sweep -- MuJoCo's order [fix_0, nb_0.., fix_1, nb_1.., ...] -- runs as the plan's
// list schedule of BLOCKS (SgEqSlot: element e's fix row and its neighbour rows): H.eq_rounds rounds, one block per lane pair
// of the env's 16-lane group and round; the blocks of a round share no slider and every block comes after the blocks it
// depends on, so the rounds in order ARE the sequential sweep (details at the equality block below).  The slider
// accelerations are kept incrementally (every update is applied to them as it happens), so there is no closing
// "fresh M^-1 J' f" pass.
// LDS of sg_pgs_rows_kernel in doubles (kernel and host use the same expressions): EPW envs per wavefront
#define SG_ROWS_LDS_FIX(EPW, NR) ((size_t)(5 * (EPW) + 2) * (NR) + 72)
#define SG_ROWS_LDS_NB(EPW, NA, N, ROUNDS) ((size_t)(EPW) * (NA) + (size_t)8 * (EPW) * ((N) + 1) + (size_t)16 * ((ROUNDS) + 4) + 72 + 2 * (EPW))  // table: 16 lanes x 8 B per round
// EPW: envs per wavefront, 8 lanes each: 8 fills the wavefront (16 finger streams advance per instruction); 4 leaves lanes 32 .. 63 idle
// but spreads a batch of 4096 envs over 1024 wavefronts -- one per SIMD of the whole chip instead of half of it -- and a wavefront
// then runs the QCQP fallback (entered when ANY of its streams slides, for as many Newton evaluations as its slowest stream needs)
// for 8 streams instead of 16.  The wavefront's instruction stream is what a launch waits for, not its lane count.
template <int NSL, bool NB, int EPW>  // NSL >= ceil(nelem / 8): joint-fix rows per lane, unrolled and padded (straight-line code, LDS reads issue back to back)
__global__ __launch_bounds__(64) void sg_pgs_rows_kernel(SgPgsArgs a) {
  extern __shared__ double lds[];
  // an env's lane group: 8 lanes (fix-only models: two finger quads); 16 lanes for neighbour-row models (EPW = 4): the two finger
  // quads (g < 8) plus two more quads that only work in the equality block (one block of rows per quad and round)
  static_assert(!NB || EPW == 4, "neighbour-row models: four envs of 16 lanes per wavefront");
  constexpr int LSH = NB ? 4 : 3, LPE = 1 << LSH;
  const int lane = threadIdx.x, le = lane >> LSH, g = lane & (LPE - 1), c = g >> 2, r = g & 3;
  const bool in_wave = le < EPW;             // lanes beyond the wavefront's envs stay idle (they own no LDS)
  const int lec = in_wave ? le : 0;
  const int env = blockIdx.x * EPW + le;
  const SG_CONSTAS SgPlanHeader& H = *(const SG_CONSTAS SgPlanHeader*)a.H;
  const int N = H.nelem;
  const size_t S = 2 * (size_t)a.nenv;
  const int nwb = (a.nenv + 7) / 8;
  const SgWork& W = a.w;
  const double mu0 = H.con_mu[0], mu1 = H.con_mu[1], pgs_scale = H.pgs_scale, tolerance = H.tolerance;
  const double mur = r == 1 ? mu0 : mu1;
  const double rsel0 = r == 0 ? 1.0 : 0.0, rsel1 = r == 1 ? 1.0 : 0.0, rsel2 = r == 2 ? 1.0 : 0.0;
  const int max_iter = H.iterations;
  // which of the wavefront's envs have a substep pending: read ONCE, all EPW words together (uniform addresses: scalar loads).  The staging
  // loops and the epilogue used to re-read the word per env behind a branch -- EPW dependent memory round trips in a row, twice per launch
  unsigned pendmask = 0;
#pragma unroll
  for (int e2 = 0; e2 < EPW; e2++) {
    const int env2 = blockIdx.x * EPW + e2;
    pendmask |= (env2 < a.nenv && W.pending[env2 < a.nenv ? env2 : 0] != 0) ? 1u << e2 : 0u;
  }
  const bool valid = in_wave && ((pendmask >> lec) & 1u) != 0;
  if (pendmask == 0) return;
  // LDS: joint-fix rows padded to NR = 8 * NSL per env (padding rows are neutral: b = 0, R = 1, 1/(A+R) = 0, so their update
  // is a no-op and the row loop needs no bound test).  Per env: AF[j] = (a_s, f) [the only pair written], BR[j] = (b, R),
  // RI[j] = 1 / (A_jj + R_j); shared by the wavefront's envs: IC[j] = (1/m, tendon coefficient).  A row is three 16-byte reads + one 8-byte.
  constexpr int NR = 8 * NSL;
  // NB layout: Ae[NA] slider accelerations per env (word N is a dummy that stays 0: "no second slider"), IC[NA] shared,
  constexpr int NA = NR + 8;
  const int RECW = 8 * (N + 1);  // doubles of an env's equality records: (N + 1) groups of four (g, 1 / (A + R)) pairs
  double2* const AF = (double2*)lds + (size_t)lec * NR;
  double2* const BR = (double2*)lds + (size_t)EPW * NR + (size_t)lec * NR;
  double* const RI = lds + (size_t)4 * EPW * NR + (size_t)lec * NR;
  double2* const IC = (double2*)(lds + (size_t)5 * EPW * NR);  // fix-only models: (1/m, tendon coefficient) per element
  // NB layout: Ae[NA] per env: slider accelerations MINUS the env's offset aoff (below); word N is a zero word ("no partner").
  // REC per env: group e = the four rows of element e's block -- its fix row, then its up to three neighbour rows -- each
  // (g, c) with g = b + R f, c = (1/m) / (A + R); rows that do not exist and group N (idle slots) hold (0, 0).  TAB (shared by the
  // wavefront's envs): the plan's block schedule as LDS byte offsets per lane of a 16-lane group: lane 2 b + h holds, for the block e
  // in slot b of the round, (x | y << 16, record offset) with (x, y) = (e, p0) for h = 0 and (p1, p2) for h = 1.
  double* const Ae = lds + (size_t)lec * NA;
  double2* const REC = (double2*)(lds + (size_t)EPW * NA + (size_t)lec * RECW);
  unsigned* const TAB = (unsigned*)(lds + (size_t)EPW * NA + (size_t)EPW * RECW);
  double* const Lnb = lds + (size_t)EPW * NA + (size_t)EPW * RECW + (size_t)16 * (H.eq_rounds + 4);  // [72 + 2 EPW]
  double* Lzero = NB ? Lnb : lds + (size_t)(5 * EPW + 2) * NR;  // [0]: a word that stays 0 (reads of "no slider"), [1 + lane]: write sink
  double* const Lenv = Lnb + 72;  // NB: [e2] sum of the env's slider accelerations at the start, [EPW + e2] its final offset aoff
  double* const ASb = NB ? Ae : (double*)AF;  // slider acceleration of element j: ASb[ASS * j]
  constexpr int ASS = NB ? 1 : 2;
  if (lane == 0) Lzero[0] = 0.0;
  if constexpr (!NB)
  for (int j = lane; j < NR; j += 64)
    IC[j] = j < N ? make_double2(1.0 / (a.elem[(size_t)SGE_MASS * N + j] + a.elem[(size_t)SGE_ARMATURE * N + j]), a.elem[(size_t)SGE_COEF * N + j]) : make_double2(0.0, 0.0);
  // Staging of the envs' rows: all 64 lanes take one env after the other, lane = row, so every load instruction reads 512
  // contiguous bytes and the loads of an env are independent of each other (until r02 every lane strode through its own env 8
  // rows apart, 41 trips with two dependent loads each: ~100 us per launch for the neighbour-row models, more than the 30 sweeps
  // of a contact-free substep).
  // (unrolled over the envs: their loads are independent and go out together -- the prologue is a chain of memory round trips, ~1.5 us
  // each, and used to take ~35 us of a 160 us contact-free launch)
#pragma unroll
  for (int e2 = 0; e2 < EPW; e2++) {
    const int env2 = blockIdx.x * EPW + e2;
    const bool v2 = ((pendmask >> e2) & 1u) != 0;  // uniform
    if constexpr (!NB) {
      double2* const AF2 = (double2*)lds + (size_t)e2 * NR;
      double2* const BR2 = (double2*)lds + (size_t)EPW * NR + (size_t)e2 * NR;
      double* const RI2 = lds + (size_t)4 * EPW * NR + (size_t)e2 * NR;
#pragma unroll
      for (int j0 = 0; j0 < NR; j0 += 64) {
        const int j = j0 + lane;
        double2 af = make_double2(0.0, 0.0), br = make_double2(0.0, 1.0);
        double ri = 0.0;
        if (v2 && j < N) {
          const size_t o = (size_t)env2 * N + j;
          const double Rr = W.eqR[o], im = 1.0 / (a.elem[(size_t)SGE_MASS * N + j] + a.elem[(size_t)SGE_ARMATURE * N + j]);
          af = make_double2(W.as[o], W.eqf[o]); br = make_double2(W.eqb[o], Rr);
          ri = sg_div(1.0, im + Rr);  // 1 / (A_jj + R_j): sg_div(res, A_jj + R_j) == res * this
        }
        if (j < NR) { AF2[j] = af; BR2[j] = br; RI2[j] = ri; }
      }
    } else {
      double* const A2 = lds + (size_t)e2 * NA;
      double2* const REC2 = (double2*)(lds + (size_t)EPW * NA + (size_t)e2 * RECW);
      const double im0 = 1.0 / (a.elem[(size_t)SGE_MASS * N] + a.elem[(size_t)SGE_ARMATURE * N]);  // equal for all elements (sg_plan_build)
      double ssum = 0.0;
#pragma unroll
      for (int j0 = 0; j0 < NA; j0 += 64) {
        const int j = j0 + lane;
        const double av = (v2 && j < N) ? W.as[(size_t)env2 * N + j] : 0.0;
        if (j < NA) A2[j] = av;
        ssum += av;
      }
      ssum = wave_sum2(ssum);
      if (lane == 0) { Lenv[e2] = ssum; Lenv[EPW + e2] = 0.0; }
    }
  }
  if constexpr (NB) {
    // equality records of all the wavefront's envs, lane = row: the loads of a row (one table word, three values per env) are
    // independent of each other and of the other rows' -- they are issued together, not one memory round trip after the other
    const double im0s = 1.0 / (a.elem[(size_t)SGE_MASS * N] + a.elem[(size_t)SGE_ARMATURE * N]);
    constexpr int REC_TRIPS = (4 * (NR + 1) + 63) / 64;   // NR >= N
#pragma unroll 4
    for (int t = 0; t < REC_TRIPS; t++) {  // row u = 4 e + d: d = 0 the fix row of e, d = 1 .. 3 its neighbour row in workspace slot (d - 1) N + e
      const int u = lane + 64 * t;
      if (u >= 4 * (N + 1)) continue;
      const int e = u >> 2, d = u & 3, ec = e < N ? e : 0;
      const bool fix = d == 0;
      const int tabw = a.nbtab[(fix ? 0 : d - 1) * N + ec];   // loaded beside the rows, not in front of them
      const bool have = e < N && (fix || tabw >= 0);
      double bb[EPW], Rr[EPW], ff[EPW];
#pragma unroll
      for (int e2 = 0; e2 < EPW; e2++) {
        const int env2 = blockIdx.x * EPW + e2, envc = env2 < a.nenv ? env2 : 0;
        const size_t o = fix ? (size_t)envc * N + ec : (size_t)envc * 3 * N + (size_t)(d - 1) * N + ec;
        bb[e2] = (fix ? W.eqb : W.nbb)[o]; Rr[e2] = (fix ? W.eqR : W.nbR)[o]; ff[e2] = (fix ? W.eqf : W.nbf)[o];
      }
#pragma unroll
      for (int e2 = 0; e2 < EPW; e2++) {
        const int env2 = blockIdx.x * EPW + e2;
        const bool v2 = ((pendmask >> e2) & 1u) != 0;  // uniform
        double2* const REC2 = (double2*)(lds + (size_t)EPW * NA + (size_t)e2 * RECW);
        double gg = 0.0, cc = 0.0;
        if (v2 && have) { gg = bb[e2] + Rr[e2] * ff[e2]; cc = sg_div(im0s, (fix ? im0s : 2.0 * im0s) + Rr[e2]); }
        REC2[u] = make_double2(gg, cc);  // group e = rows 4 e .. 4 e + 3
      }
    }
  }
  if constexpr (NB) {  // the schedule (plus four idle rounds for the look-ahead) as LDS byte offsets: lane 2 b + h of a group, block slot b
    const int ntab = 16 * (H.eq_rounds + 4);   // a.tab: the same words, laid out by the host once per batch (sg_api.hip)
#pragma unroll 4
    for (int i = lane; i < ntab; i += 64) ((uint2*)TAB)[i] = a.tab[i];
  }
  const bool sv = valid && g < 8;  // lanes of the env's two finger quads
  const size_t st = 2 * (size_t)(sv ? env : 0) + (sv ? c : 0);
  int ns = 0, lim_active = 0, shared = 0;
  // lane r of the quad owns finger acceleration aF[r]; of M^-1 it needs row r (its share of a limit row's push) and the diagonal
  double Mrow[SG_CD] = {0, 0, 0, 0}, Mdiag[SG_CD] = {0, 0, 0, 0}, aFo = 0;
  double tb = 0, tR = 1, tA = 1, tf = 0;
  double lsign[SG_MAXLIM], lR[SG_MAXLIM], lb[SG_MAXLIM], lf[SG_MAXLIM];
#pragma unroll
  for (int k = 0; k < SG_MAXLIM; k++) { lsign[k] = 0; lR[k] = 1; lb[k] = 0; lf[k] = 0; }
  if (valid) {
    tb = W.envh[(size_t)0 * a.nenv + env]; tR = W.envh[(size_t)1 * a.nenv + env];
    tA = W.envh[(size_t)2 * a.nenv + env]; tf = W.envh[(size_t)3 * a.nenv + env];
    shared = W.shared[env];
  }
  const int ngen = valid ? W.gen[env] : 0;  // contacts of an env on the general contact path (sg_general.h); 0 on the fast path
  const bool wave_gen = __ballot(ngen > 0) != 0;
  if (sv) {
    ns = W.ns[st];
    lim_active = W.lim_active[st];
#pragma unroll
    for (int d = 0; d < SG_CD; d++) { Mrow[d] = W.sMinv[(size_t)(4 * r + d) * S + st]; Mdiag[d] = W.sMinv[(size_t)(5 * d) * S + st]; }
    aFo = W.saF[(size_t)r * S + st];
#pragma unroll
    for (int k = 0; k < SG_MAXLIM; k++) {  // the stream's limit rows live in registers, the same values on the four lanes of its quad
      lsign[k] = W.lim[((size_t)0 * SG_MAXLIM + k) * S + st]; lR[k] = W.lim[((size_t)1 * SG_MAXLIM + k) * S + st];
      lb[k] = W.lim[((size_t)2 * SG_MAXLIM + k) * S + st]; lf[k] = W.lim[((size_t)3 * SG_MAXLIM + k) * S + st];
    }
  }
  __syncthreads();
  int nsmax = ns;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(nsmax, o); nsmax = t > nsmax ? t : nsmax; }
  const unsigned long long any_lim = __ballot(lim_active != 0);

  bool running = valid;
  int iters = 0;
  // NB: the env's acceleration offset, the tracked sum of its slider accelerations and this lane's not yet reduced share of it
  double aoff = 0.0, Ssum = NB ? Lenv[lec] : 0.0, dS = 0.0;
  const double im0 = NB ? 1.0 / (a.elem[(size_t)SGE_MASS * N] + a.elem[(size_t)SGE_ARMATURE * N]) : 0.0;
  // my column in the wave's block, biased by 4 field pairs (immediate offsets -4096 .. 3072).  Lanes without an env (idle half of
  // an EPW = 4 wavefront, ragged tail, env not pending) read and write a block of their own wavefront that holds zeros and
  // never advance: a dummy block shared by all wavefronts made every idle lane of the chip hammer the same 8 KB
  // (profiles/r02: 1024 wavefronts x 32 idle lanes, contact rows 2.3x slower than with 16 streams per wavefront).
  // (the phase kernel writes blocks of 8 envs: env e sits in block e >> 3 at lanes 8 (e & 7) .. + 7, whatever EPW is)
  const bool has_row = sv;
  const double2* const row0 = has_row ? (const double2*)(W.crow + SG_ROW_INDEX(0, env >> 3, 8, 8 * (env & 7) + g, nwb))
                                      : (const double2*)(W.cdummy + ((size_t)blockIdx.x * (SG_RK / 2) + 4) * 128 + 2 * lane);
  const size_t slot_stride = has_row ? (size_t)(nwb + 2) * (SG_RK / 2) * 64 : 0;  // in double2 units (per lane: idle lanes stay put)
  constexpr ptrdiff_t sink_off = 0;

  SG_T0();
  for (int it = 0; it < max_iter; it++) {
    if (!__ballot(running)) break;
    SG_T(10);
    double imp_acc = 0;
    [[maybe_unused]] double tJap = 0;
    if constexpr (NB) {
      // Equality block as the plan's block schedule: H.eq_rounds rounds, one BLOCK per lane of the env's group -- element e's fix
      // row and its up to three neighbour rows, in MuJoCo's order, with slider e's acceleration carried in a register.  A row's
      // state is g = b + R f; with res = g + a1 - a2 and t = res / (A + R) the update is a1 -= t / m, a2 += t / m and g -= R t,
      // which is g' = a2' - a1' (the row's residual is zero after its update), so R is not needed in the sweep.  All elements
      // have the same mass and tendon coefficient 1 (sg_plan_build), hence:
      //  * the tendon row's push is the same for every slider: it goes into ONE per-env offset, aoff (true a_e = A[e] + aoff),
      //    instead of a pass over all sliders; neighbour rows see differences of sliders (offset-free), fix rows and contacts add it;
      //  * the tendon row's J a = sum of the slider accelerations is TRACKED (Ssum): a neighbour row leaves it unchanged, a fix row
      //    changes it by -t / m, a contact by its push on its slider, the tendon row by sum(1/m) dft -- no pass over the sliders.
      // What counts is rounds x (instructions per round x ~7 cycles + one LDS round trip): 24 block rounds for softbox instead of
      // 53 row rounds, and no pass for the tendon row.
      if (running) {
        // One block per lane PAIR and round: lane h = 0 holds rows 0, 1 of the block (the fix row and the first neighbour row), lane
        // h = 1 rows 2, 3 -- per row its state g_k, its step factor c_k = (1/m) / (A + R) and its partner's acceleration P_k (row 0: the
        // constant -aoff, a "partner" that is never pushed).  With d_k = g_k - P_k the block's sequential sweep over slider e's
        // acceleration is   e_{k+1} = e_k (1 - c_k) - c_k d_k,   k = 0 .. 3,   e_0 = a_e:
        // two dependent multiply-adds on lane 0, one DPP hand-over, two on lane 1; each row's residual s_k = d_k + e_k, push
        // w_k = c_k s_k on its partner and new state g_k' = P_k' - e_{k+1} follow.  A wavefront alone on its SIMD issues one
        // instruction per ~7 cycles whatever it is, so rounds x instructions per round is the cost: 24 x ~47 here; one row per lane
        // (a quad per block) was 32 x 48, a block per lane 24 x 80.
        char* const Ab = (char*)Ae;
        char* const Rb = (char*)REC;
        const uint2* tp = (const uint2*)TAB + g;
        const int hh = g & 1;
        const double hm = hh ? 1.0 : 0.0, h0 = 1.0 - hm;
        const double Pfix = h0 * -aoff;  // row 0's "partner" -aoff on lane 0 (added to hm * X)
        double qc = 0.0, sc = 0.0;
        const int nrounds = H.eq_rounds;
        struct Rec { double2 a, b; };  // (g, c) of my two rows
        struct Adr { double *px, *py; double2* rec; };   // lane 0: slider e itself / partner of row 1; lane 1: partners of rows 2, 3; my rows' records
        struct Off { double sA, wA, sB, wB, ga, gb; };   // what a round leaves for its off-chain part
        auto adr_of = [&](const uint2 tt) {
          Adr q;
          q.px = (double*)(Ab + (tt.x & 0xffffu)); q.py = (double*)(Ab + (tt.x >> 16)); q.rec = (double2*)(Rb + (tt.y & 0xffffu));
          return q;
        };
        auto ld_rec = [&](const Adr q) { Rec rc; rc.a = q.rec[0]; rc.b = q.rec[1]; return rc; };
        // The CHAIN of a round -- slider reads -> two rows on lane 0 -> hand-over -> two rows on lane 1 -> hand-over -> slider writes, which
        // the next round's reads wait for -- is ~250 cycles of latency; the round's other ~24 instructions (cost sums, new row states and
        // their stores, the next records / table words / addresses) are OFF the chain.  r03 measured 459 cycles per round against
        // 44 x 7 = 308 of issue: so the loop is written as chain(r) | reads(r + 1) | off-chain(r), with scheduling barriers between the
        // groups, and the off-chain work of a round runs in the shadow of the next round's LDS reads.
        auto chain = [&](const Adr q, const Rec rc, const double X, const double Y, Off& o) {
          const double cA = rc.a.y, cB = rc.b.y;
          const double PA = fma(X, hm, Pfix), dA = rc.a.x - PA, dB = rc.b.x - Y;
          // a row's step on slider e: s = d + e (its residual), w = c s (its push on the partner), e' = e - w.  Lane 0 runs its two rows
          // from e = a_e, hands the result over, lane 1 runs its two rows from there (both lanes execute both passes: in the second one
          // lane 0 repeats its own numbers)
          const double sA0 = dA + X, O10 = X - cA * sA0, O20 = O10 - cB * (dB + O10);   // lane 0: e after rows 0, 1
          const double T = sg_dpp<0xB1>(O20);                    // quad_perm [1,0,3,2]: the pair's other lane
          const double I = hh ? T : X;                           // my first row's e_k
          const double sA = dA + I, wA = cA * sA, O1 = I - wA, sB = dB + O1, wB = cB * sB, O2 = O1 - wB;
          const double PAn = fma(wA, hm, PA), Yn = Y + wB;
          const double e4 = sg_dpp<0xB1>(O2);                    // lane 0 receives the block's result from lane 1
          *q.px = hh ? PAn : e4;
          *q.py = Yn;
          o.sA = sA; o.wA = wA; o.sB = sB; o.wB = wB; o.ga = PAn - O1; o.gb = Yn - O2;
        };
        auto offchain = [&](const Adr q, const Off& o) {
          qc += o.sA * o.wA + o.sB * o.wB;
          sc += o.wA * h0;
          q.rec[0].x = o.ga; q.rec[1].x = o.gb;                  // the rows' new states g' = P' - e' (their residual is zero after the update)
        };
        // two rounds per trip (register sets A, B); table words are fetched two rounds ahead, records and addresses one round ahead
        uint2 tC = tp[0], tD = tp[16];
        Adr qA = adr_of(tC), qB = adr_of(tD);
        Rec rA = ld_rec(qA), rB = ld_rec(qB);
        double XA = *qA.px, YA = *qA.py, XB, YB;
        Off oA, oB;
        for (int k = 0; k < nrounds; k += 2) {  // an odd count runs one idle round (the table ends with four)
          tp += 32;
          tC = tp[0]; tD = tp[16];
          chain(qA, rA, XA, YA, oA);
          XB = *qB.px; YB = *qB.py;
          __builtin_amdgcn_sched_barrier(0);
          offchain(qA, oA);
          qA = adr_of(tC); rA = ld_rec(qA);
          __builtin_amdgcn_sched_barrier(0);
          chain(qB, rB, XB, YB, oB);
          XA = *qA.px; YA = *qA.py;
          __builtin_amdgcn_sched_barrier(0);
          offchain(qB, oB);
          qB = adr_of(tD); rB = ld_rec(qB);
          __builtin_amdgcn_sched_barrier(0);
        }
        imp_acc += 0.5 * qc * (1.0 / im0);
        dS -= sc;
      }
      SG_T(11);
      // tendon row: J a = the tracked sum (+ what this group's lanes and contact quads have added since the last tendon row)
      {
        const double Ja = Ssum + sg_gsum16(dS);
        double old = tf, tfn = tf;
        double ch = scalar_update(tfn, tb, Ja, tR, tA, false);
        if (running) {
          if (g == 0) imp_acc -= ch;
          tf = tfn;
          const double dft = tf - old;
          aoff += im0 * dft;
          Ssum = Ja + (tA - tR) * dft;  // tA - tR = sum over the sliders of coef^2 / m
          dS = 0.0;
        }
      }
    } else {
    double ael[NSL], fnw[NSL], imc[NSL];
    if (running) {
      // joint-fix rows: unconstrained scalar rows, so the step d = -res / (A + R) always lowers the cost (change =
      // -res^2 / (2 (A + R)) <= 0) and the generic "revert if the cost went up" test of the other row types can never fire
      // the LDS reads of row t + 2 are issued before row t is computed (three register sets in flight)
      double2 afq[NSL], brq[NSL], icq[NSL];
      double riq[NSL];
#pragma unroll
      for (int t = 0; t < 2 && t < NSL; t++) { afq[t] = AF[g + 8 * t]; brq[t] = BR[g + 8 * t]; icq[t] = IC[g + 8 * t]; riq[t] = RI[g + 8 * t]; }
#pragma unroll
      for (int t = 0; t < NSL; t++) {
        if (t + 2 < NSL) { const int jn = g + 8 * (t + 2); afq[t + 2] = AF[jn]; brq[t + 2] = BR[jn]; icq[t + 2] = IC[jn]; riq[t + 2] = RI[jn]; }
        const double2 af = afq[t], br = brq[t], ic = icq[t];
        const double ri = riq[t];
        const double ae = af.x, old = af.y, Rr = br.y, im = ic.x;
        const double res = br.x + ae + Rr * old;
        const double fn = old - res * ri;
        const double d = fn - old, change = 0.5 * d * d * (im + Rr) + d * res;
        imp_acc -= change;
        fnw[t] = fn;
        ael[t] = ae + im * d;
        tJap += ic.y * ael[t];
        imc[t] = im * ic.y;
      }
    }
    SG_T(11);
    {
      double Ja = tJap;
      Ja = sg_gsum8(Ja);
      double old = tf, tfn = tf;
      double ch = scalar_update(tfn, tb, Ja, tR, tA, false);
      if (running) {
        if (g == 0) imp_acc -= ch;
        tf = tfn;
      }
      const double dft = tf - old;
      if (running) {
#pragma unroll
        for (int t = 0; t < NSL; t++) AF[g + 8 * t] = make_double2(ael[t] + imc[t] * dft, fnw[t]);
      }
    }
    }
    __syncthreads();
    SG_T(12);
    for (int pass = 0; pass < 2; pass++) {
      const bool mine = running && g < 8 && ((c == 0 || !shared) ? pass == 0 : pass == 1);
      if (!__ballot(mine)) continue;
      auto qbd = [&](double x, int d) { return d == 0 ? sg_qb<0>(x) : (d == 1 ? sg_qb<1>(x) : (d == 2 ? sg_qb<2>(x) : sg_qb<3>(x))); };  // d is a constant after unrolling
      if (any_lim) {  // limit rows: every lane of the quad computes the same scalars, lane 0 records the force
#pragma unroll
        for (int k = 0; k < SG_MAXLIM; k++) {
          if (mine && (lim_active >> k & 1)) {
            const int d = k / 2;
            double f = lf[k], old = f, sg = lsign[k], Rr = lR[k];
            double ch = scalar_update(f, lb[k], sg * qbd(aFo, d), Rr, Mdiag[d] + Rr, true);
            lf[k] = f;
            if (r == 0) imp_acc -= ch;
            double df = sg * (f - old);
            aFo += Mrow[d] * df;
          }
        }
      }
      SG_T(13);
      const int nsl = mine ? ns : 0;
      if (nsmax == 0) { SG_T(14); continue; }  // no contacts anywhere in the wavefront: no look-ahead loads to wait for, no barrier
      struct Row { double2 j01, j23, jsb, fw, a01, a2s, p12, p3i; };
      auto load_row = [&](Row& w, const double2* p) {
        w.j01 = p[-4 * 64]; w.j23 = p[-3 * 64]; w.jsb = p[-2 * 64]; w.fw = p[-1 * 64];
        w.a01 = p[0]; w.a2s = p[1 * 64]; w.p12 = p[2 * 64]; w.p3i = p[3 * 64];
      };
      auto update_row = [&](Row& w, int i, const double2* pl) {
        if (i < nsl) {
          const double J0 = w.j01.x, J1 = w.j01.y, J2 = w.j23.x, J3 = w.j23.y, Js = w.jsb.x, bb = w.jsb.y, fo = w.fw.x, wv = w.fw.y;
          const double A0 = w.a01.x, A1 = w.a01.y, A2 = w.a2s.x, JsI = w.a2s.y, W0 = w.p12.x, W1 = w.p12.y, W2 = w.p3i.x;
          // what the rows share sits on lane 3 (fields 0 .. 4); its own "row" is inert: f = A f = A = invm Js = 0 and no rsel
          const double Rr = w.p3i.y;  // R: replicated on the row lanes (field 15), 0 on lane 3
          const double P11 = sg_qb<3>(J0), P12 = sg_qb<3>(J1), P22 = sg_qb<3>(J2);  // early: off the update's dependency chain
          const int sl = __double2loint(sg_qb<3>(Js));
          const double araw = *(sl >= 0 ? (const double*)(ASb + ASS * sl) : &Lzero[0]);  // branch-free: "no slider" reads a zero word
          const double as_ = NB ? araw + aoff : araw;  // (a contact without a slider has J_s = 0: the offset it then sees is inert)
          const double f0_ = sg_qb<0>(aFo), f1_ = sg_qb<1>(aFo), f2_ = sg_qb<2>(aFo), f3_ = sg_qb<3>(aFo);  // the finger's four accelerations
          const double res = ((bb + Js * as_) + (J0 * f0_ + J1 * f1_)) + ((Rr * fo + J2 * f2_) + J3 * f3_);  // unused on lane 3
          const double o0 = sg_qb<0>(fo);
          // ---- normal or ray update (wv = row r of A f, kept with f)
          double denom = fo * wv, num = fo * res;  // two quad sums, interleaved so the DPP read-after-write hazards hide each other
          { const double t0 = sg_dpp<0xB1>(denom), t1 = sg_dpp<0xB1>(num); denom += t0; num += t1; }
          { const double t0 = sg_dpp<0x4E>(denom), t1 = sg_dpp<0x4E>(num); denom += t0; num += t1; }
          double x = denom >= SG_MINVAL ? sg_div(-num, denom) : 0.0;
          x = (o0 + x * o0 < 0) ? -1.0 : x;
          double gr = fo + x * fo;
#ifdef SG_SECTION_COUNT  // event counters (build_native.py --count): atomics inside the update, so the cycle stamps of such a build are not timings
          if (r == 0) atomicAdd(&a.w.secprof[26], 1ull);  // contact updates (per stream)
#endif
          if (o0 < SG_MINVAL) {  // uncommon: no normal force yet (lane 0 holds res_0 and A_00)
            double gn = o0 - sg_div(res, A0);
            gr = r == 0 ? (gn < 0 ? 0.0 : gn) : 0.0;
          }
          const double g0 = sg_qb<0>(gr);
          // ---- friction rows with the normal force fixed: every lane solves the 2 x 2 block, so the new force triple is
          //      known on all lanes without broadcasts
          const double bc = (res - wv) + A0 * g0;   // res - (A f)_r + A_r0 f_0 + A_r0 (g0 - f_0)
          const bool nofric = g0 < SG_MINVAL;
          const double bmu = bc * (nofric ? 0.0 : mur);  // no normal force, no friction: zero right-hand sides give u = v = 0
          const double b1 = sg_qb<1>(bmu), b2 = sg_qb<2>(bmu);
          const double u1 = -(P11 * b1 + P12 * b2), u2 = -(P12 * b1 + P22 * b2);  // 0 when the friction block is singular
          const double val = (u1 * u1 + u2 * u2) - g0 * g0;
          double v1 = u1 * mu0, v2 = u2 * mu1;
          if (!(val < 1e-10) && !nofric) {  // uncommon: outside the cone -- the generic Newton iteration
#ifdef SG_SECTION_COUNT
            if (r == 0) atomicAdd(&a.w.secprof[27], 1ull);  // sliding contact updates (per stream)
            if (lane == __ffsll((long long)__ballot(true)) - 1) atomicAdd(&a.w.secprof[32], 1ull);  // fallback entries per wavefront
#endif
            // mju_QCQP2's Newton iteration on the multiplier la of |v(la)|^2 = g0^2, v(la) = -(S + la)^-1 b in friction-scaled
            // coordinates, continued from its first evaluation (la = 0: P, (u1, u2) and val are the fast path's).  Same iterates
            // and stopping rules (val < 1e-10, step < 1e-10, singular block, 20 evaluations).  A wavefront runs as many
            // evaluations as its slowest stream needs and an evaluation is one dependent chain, so it is written (a) without
            // data-dependent branches and (b) with ONE division: with w = -adj(S + la) b and det = |S + la|, v = w / det,
            // val = (|w|^2 - g0^2 det^2) / det^2 and the Newton step -val / (d val / d la) = (|w|^2 - g0^2 det^2) det / (2 w' adj w);
            // v itself is only needed after the last evaluation.  (Two divisions and three nested exec-mask branches per
            // evaluation before: 460 cycles each, profiles/r01_v11_nb_kernel_sections.txt.)
            // In the eigen-coordinates of S (exported by the phase kernel on the quad's fourth lane: S = Q diag(e1, e2) Q', constant over
            // the solve) with c = Q' b, x_k = e_k + la:  |v|^2 = c1^2 / x1^2 + c2^2 / x2^2, so with y_k = x_k^2
            //   val det^2 = N = c1^2 y2 + c2^2 y1 - g0^2 y1 y2,   w' adj w = D = c1^2 y2 x2 + c2^2 y1 x1,   det = x1 x2,
            // and the Newton step is N x1 x2 / (2 D): 18 flops per evaluation instead of 27, and the loop is a plain divergent one (lanes
            // that have stopped are masked off and keep their last x1, x2; no selects, no wave-wide flags).
            const double e1 = sg_qb<3>(A0), e2 = sg_qb<3>(A1), qcs = sg_qb<3>(A2), qsn = sg_qb<3>(JsI);
            const double c1 = qcs * b1 - qsn * b2, c2 = qsn * b1 + qcs * b2;
            const double C1h = 0.5 * c1 * c1, C2h = 0.5 * c2 * c2, R2h = 0.5 * g0 * g0;
            double la = 0.0;
            bool run = true;
            {
              const double deriv = -2.0 * (P11 * u1 * u1 + 2.0 * P12 * u1 * u2 + P22 * u2 * u2), delta = sg_div(-val, deriv);
              run = !(delta < 1e-10);
              la = run ? delta : 0.0;
            }
            const bool ever = run;
            double x1 = e1, x2 = e2;
            if (run) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SG_SECTION_COUNT)
              // the loop by hand (the compiler's lowering of the divergent loop spends 14 of its 41 instructions per evaluation on exec-mask
              // bookkeeping; a lone wavefront pays ~7 cycles for each, scalar or vector): 28 instructions per evaluation, two evaluations
              // per trip.  Lanes leave by having their exec bit cleared; exec is restored at the end.  v_rcp_f64 (a transcendental-unit op) has two independent
              // instructions between it and its first consumer.
              double y1, y2, xx, ah, bh, yy, nh, dh, rc, nx, er, dl, ox1, ox2;  // ox1, ox2: early-clobber outputs (as read-write
              unsigned long long sv, m0, m1;                                    // operands initialised with e1, e2 they were given e1's, e2's registers)
              unsigned cnt;
              asm volatile(
                  "s_mov_b64 %[sv], exec\n\t"
                  "s_mov_b32 %[cnt], 19\n"      // evaluations 1 .. 19 (the fast path did evaluation 0): nine trips of two and one more
                  "1:\n\t"
                  "v_add_f64 %[x1], %[e1], %[la]\n\t"
                  "v_add_f64 %[x2], %[e2], %[la]\n\t"
                  "v_mul_f64 %[y1], %[x1], %[x1]\n\t"
                  "v_mul_f64 %[y2], %[x2], %[x2]\n\t"
                  "v_mul_f64 %[xx], %[x1], %[x2]\n\t"
                  "v_mul_f64 %[ah], %[C1h], %[y2]\n\t"
                  "v_mul_f64 %[bh], %[C2h], %[y1]\n\t"
                  "v_mul_f64 %[yy], %[y1], %[y2]\n\t"
                  "v_add_f64 %[nh], %[ah], %[bh]\n\t"
                  "v_mul_f64 %[dh], %[bh], %[x1]\n\t"
                  "v_fma_f64 %[nh], %[nR2h], %[yy], %[nh]\n\t"
                  "v_fma_f64 %[dh], %[ah], %[x2], %[dh]\n\t"
                  "v_cmp_gt_f64_e64 %[m0], %[tol], %[xx]\n\t"
                  "v_add_f64 %[dh], %[dh], %[dh]\n\t"
                  "v_mul_f64 %[yy], %[yy], %[tolh]\n\t"
                  "v_rcp_f64_e32 %[rc], %[dh]\n\t"
                  "v_mul_f64 %[nx], %[nh], %[xx]\n\t"
                  "v_cmp_lt_f64_e64 %[m1], %[nh], %[yy]\n\t"
                  "v_fma_f64 %[er], -%[dh], %[rc], 1.0\n\t"
                  "s_or_b64 %[m0], %[m0], %[m1]\n\t"
                  "v_fma_f64 %[rc], %[er], %[rc], %[rc]\n\t"
                  "v_mul_f64 %[dl], %[nx], %[rc]\n\t"
                  "v_cmp_gt_f64_e64 %[m1], %[tol], %[dl]\n\t"
                  "s_or_b64 %[m0], %[m0], %[m1]\n\t"
                  "s_andn2_b64 exec, exec, %[m0]\n\t"
                  "s_cbranch_execz 2f\n\t"
                  "v_add_f64 %[la], %[la], %[dl]\n\t"
                  "s_cmp_eq_u32 %[cnt], 1\n\t"
                  "s_cbranch_scc1 2f\n\t"
                  "v_add_f64 %[x1], %[e1], %[la]\n\t"
                  "v_add_f64 %[x2], %[e2], %[la]\n\t"
                  "v_mul_f64 %[y1], %[x1], %[x1]\n\t"
                  "v_mul_f64 %[y2], %[x2], %[x2]\n\t"
                  "v_mul_f64 %[xx], %[x1], %[x2]\n\t"
                  "v_mul_f64 %[ah], %[C1h], %[y2]\n\t"
                  "v_mul_f64 %[bh], %[C2h], %[y1]\n\t"
                  "v_mul_f64 %[yy], %[y1], %[y2]\n\t"
                  "v_add_f64 %[nh], %[ah], %[bh]\n\t"
                  "v_mul_f64 %[dh], %[bh], %[x1]\n\t"
                  "v_fma_f64 %[nh], %[nR2h], %[yy], %[nh]\n\t"
                  "v_fma_f64 %[dh], %[ah], %[x2], %[dh]\n\t"
                  "v_cmp_gt_f64_e64 %[m0], %[tol], %[xx]\n\t"
                  "v_add_f64 %[dh], %[dh], %[dh]\n\t"
                  "v_mul_f64 %[yy], %[yy], %[tolh]\n\t"
                  "v_rcp_f64_e32 %[rc], %[dh]\n\t"
                  "v_mul_f64 %[nx], %[nh], %[xx]\n\t"
                  "v_cmp_lt_f64_e64 %[m1], %[nh], %[yy]\n\t"
                  "v_fma_f64 %[er], -%[dh], %[rc], 1.0\n\t"
                  "s_or_b64 %[m0], %[m0], %[m1]\n\t"
                  "v_fma_f64 %[rc], %[er], %[rc], %[rc]\n\t"
                  "v_mul_f64 %[dl], %[nx], %[rc]\n\t"
                  "v_cmp_gt_f64_e64 %[m1], %[tol], %[dl]\n\t"
                  "s_or_b64 %[m0], %[m0], %[m1]\n\t"
                  "s_andn2_b64 exec, exec, %[m0]\n\t"
                  "v_add_f64 %[la], %[la], %[dl]\n\t"   // (with every lane masked off this adds nothing)
                  "s_sub_u32 %[cnt], %[cnt], 2\n\t"
                  "s_cbranch_execnz 1b\n"
                  "2:\n\t"
                  "s_mov_b64 exec, %[sv]"
                  : [x1] "=&v"(ox1), [x2] "=&v"(ox2), [la] "+v"(la), [y1] "=&v"(y1), [y2] "=&v"(y2), [xx] "=&v"(xx), [ah] "=&v"(ah), [bh] "=&v"(bh),
                    [yy] "=&v"(yy), [nh] "=&v"(nh), [dh] "=&v"(dh), [rc] "=&v"(rc), [nx] "=&v"(nx), [er] "=&v"(er), [dl] "=&v"(dl),
                    [sv] "=&s"(sv), [m0] "=&s"(m0), [m1] "=&s"(m1), [cnt] "=&s"(cnt)
                  : [e1] "v"(e1), [e2] "v"(e2), [C1h] "v"(C1h), [C2h] "v"(C2h), [nR2h] "v"(-R2h), [tol] "s"(1e-10), [tolh] "s"(0.5e-10)
                  : "vcc", "scc");
              x1 = ox1; x2 = ox2;
#else
#pragma unroll 1
              for (int it = 1; it < 20; it++) {
#ifdef SG_SECTION_COUNT
                if (lane == __ffsll((long long)__ballot(true)) - 1) atomicAdd(&a.w.secprof[33], 1ull);  // Newton iterations per wavefront
#endif
                x1 = e1 + la; x2 = e2 + la;
                const double y1 = x1 * x1, y2 = x2 * x2, ah = C1h * y2, bh = C2h * y1, yy = y1 * y2;
                const double Nh = fma(-R2h, yy, ah + bh);            // val det^2 / 2
                const double Dh = fma(ah, x2, bh * x1);              // w' adj w / 2
                const double xx = x1 * x2;                           // det
                const double delta = sg_div_fast(Nh * xx, Dh + Dh);  // a Newton step: its last bits are absorbed by the next evaluation
                if (xx < 1e-10 || Nh < 0.5e-10 * yy || delta < 1e-10) break;
                la += delta;
              }
#endif
            }
            const double det = x1 * x2;
            // last evaluation, back in the contact's coordinates: v = Q (-c1 / x1, -c2 / x2) = (t1, t2) / det
            const double t1e = -c1 * x2, t2e = -c2 * x1;
            const double t1 = qcs * t1e + qsn * t2e, t2 = qcs * t2e - qsn * t1e;
            const bool sing = ever && det < 1e-10;
            const double w1 = ever ? t1 : u1, w2 = ever ? t2 : u2, wdet = ever ? det : 1.0;  // last evaluation: v = (w1, w2) / wdet
            const bool active = la != 0.0 && !sing;
            // v = w / wdet, then rescaled onto the cone when the constraint is active: |(v1 / mu0, v2 / mu1)| = g0, i.e.
            // (w1, w2) -> g0 / |w| (w1, w2) in scaled coordinates (the division by wdet cancels)
            const double q = fmax(SG_MINVAL, active ? w1 * w1 + w2 * w2 : wdet * wdet);
            double y = __builtin_amdgcn_rsq(q);
            y = y * (1.5 - 0.5 * q * y * y);
            y = y * (1.5 - 0.5 * q * y * y);  // active: 1 / |w|; otherwise 1 / |wdet| (wdet > 0)
            const double sc = sing ? 0.0 : (active ? g0 * y : y);
            v1 = w1 * mu0 * sc; v2 = w2 * mu1 * sc;
          }
          double fn = rsel0 * g0 + (rsel1 * v1 + rsel2 * v2);  // my row of (g0, v1, v2): exact, two of the three terms are 0
          double wn = (A0 * g0 + A1 * v1) + A2 * v2;   // row r of A f_new
          double dr = fn - fo;
          const double change = sg_qsum(dr * (0.5 * (wn - wv) + res));
          const bool reject = change > 1e-10;
          fn = reject ? fo : fn;
          wn = reject ? wv : wn;
          dr = reject ? 0.0 : dr;
          imp_acc -= (reject || r != 0) ? 0.0 : change;
          // the slider's share: one quad sum (invm J_s' df); the finger's: lane q adds its column of M^-1 J_F' df
          const double jsdf = sg_qsum(JsI * dr);
          *((sl >= 0 && r == 0) ? (double*)(ASb + ASS * sl) : &Lzero[1 + lane]) = araw + jsdf;  // other lanes write to their sink word
          if constexpr (NB) dS += 0.25 * jsdf;  // the four lanes of the quad hold the same push: the group sum counts it once
          aFo += (W0 * sg_qb<0>(dr) + W1 * sg_qb<1>(dr)) + W2 * sg_qb<2>(dr);
          w.fw = make_double2(fn, wn);
        }
        ((double2*)pl)[sink_off - 1 * 64] = w.fw;
      };
#ifdef SG_SECTION_PROF
      if (lane == 0) { atomicAdd(&a.w.secprof[28], (unsigned long long)nsmax); atomicAdd(&a.w.secprof[29], 1ull); }
#endif
      // slots 0 .. SG_CAP+1 exist in memory (two spare slots), so the look-ahead never needs a bound check
      Row ra, rb;
      const double2* pa = row0;
      load_row(ra, pa);
      for (int i = 0; i < nsmax; i += 2) {
        const double2* pb = pa + slot_stride;
        load_row(rb, pb);
        update_row(ra, i, pa);
        pa = pb + slot_stride;
        load_row(ra, pa);
        update_row(rb, i + 1, pb);
      }
      __syncthreads();
      SG_T(14);
    }
    if (wave_gen) {
      // General contact path: the env's contacts are ONE ordered list (W.gcon) with both chains' Jacobian blocks; lane 0 of the env's
      // group sweeps it serially after both chains' limit rows (MuJoCo's row order).  Rare by construction -- the wavefront's other
      // envs wait meanwhile -- so nothing here is tuned: the two chains' accelerations are gathered from / scattered to their quad
      // lanes with shuffles, the records stream from memory one at a time.
      const int gbase = lane - g;
      double aF2[SG_MAXCH][SG_CD];
#pragma unroll
      for (int cc = 0; cc < SG_MAXCH; cc++)
#pragma unroll
        for (int q = 0; q < SG_CD; q++) aF2[cc][q] = __shfl(aFo, gbase + 4 * cc + q);
      if (ngen > 0 && g == 0 && running) {
        const size_t st0 = 2 * (size_t)env;
        double Mi[SG_MAXCH][16];
#pragma unroll
        for (int cc = 0; cc < SG_MAXCH; cc++)
#pragma unroll
          for (int i = 0; i < 16; i++) Mi[cc][i] = W.sMinv[(size_t)i * S + st0 + cc];
        const double mu[2] = {mu0, mu1};
#pragma unroll 1
        for (int i = 0; i < ngen; i++) {
          double* rec = W.gcon + ((size_t)env * SG_GEN_MAXCON + i) * SG_GEN_W;
          GenContact gc;
          gen_contact_load(gc, rec);
          const double araw = gc.sl >= 0 ? ASb[ASS * gc.sl] : 0.0, as_ = NB ? araw + aoff : araw;
          double df[3];
          imp_acc -= gen_contact_update(gc, aF2, as_, mu, df);
          rec[SG_GEN_F_OFF] = gc.f[0]; rec[SG_GEN_F_OFF + 1] = gc.f[1]; rec[SG_GEN_F_OFF + 2] = gc.f[2];
#pragma unroll
          for (int cc = 0; cc < SG_MAXCH; cc++) {
            double gd[SG_CD];
#pragma unroll
            for (int d = 0; d < SG_CD; d++) gd[d] = gc.Jf[cc][0][d] * df[0] + gc.Jf[cc][1][d] * df[1] + gc.Jf[cc][2][d] * df[2];
#pragma unroll
            for (int q = 0; q < SG_CD; q++)
#pragma unroll
              for (int d = 0; d < SG_CD; d++) aF2[cc][q] += Mi[cc][4 * q + d] * gd[d];
          }
          if (gc.sl >= 0) {
            const double jsdf = gc.invm * (gc.Js[0] * df[0] + gc.Js[1] * df[1] + gc.Js[2] * df[2]);
            ASb[ASS * gc.sl] = araw + jsdf;
            if constexpr (NB) dS += jsdf;   // one lane holds the push: the group sum counts it once
          }
        }
      }
#pragma unroll
      for (int cc = 0; cc < SG_MAXCH; cc++)
#pragma unroll
        for (int q = 0; q < SG_CD; q++) {
          const double v = __shfl(aF2[cc][q], gbase);
          if (ngen > 0 && g == 4 * cc + q) aFo = v;
        }
      __syncthreads();
    }
    double imp = imp_acc;
    imp = NB ? sg_gsum16(imp) : sg_gsum8(imp);
    if (running) {
      iters = it + 1;
      if (imp * pgs_scale < tolerance) running = false;
    }
  }
  SG_T(15);
  __syncthreads();
  // ---- the solver's result is M^-1 J' f.  Every row update has applied its force change to the accelerations it touches
  //      (aF of the stream's finger, the slider words in LDS), so they ARE M^-1 J' f of the final forces up to the round-off of
  //      ~10^3 additions (parity against the oracle, which multiplies out the final forces: 9e-12 over the episode).  Recomputing
  //      them from the forces cost one more pass over all contact rows, 3.3 % of the kernel (until r01 v11)
  __syncthreads();
  if (sv) W.saF[(size_t)r * S + st] = aFo;
  if (valid && g == 0) W.iters[env] = iters;
  if constexpr (NB) {
    if (valid && g == 0) Lenv[EPW + le] = aoff;
    __syncthreads();
  }
#pragma unroll 1
  for (int e2 = 0; e2 < EPW; e2++) {  // slider accelerations back to the workspace, lane = element
    const int env2 = blockIdx.x * EPW + e2;
    if ((pendmask >> e2) & 1u) {
      const double* const AS2 = NB ? lds + (size_t)e2 * NA : lds + (size_t)2 * e2 * NR;
      const double off2 = NB ? Lenv[EPW + e2] : 0.0;
      for (int j = lane; j < N; j += 64) W.as[(size_t)env2 * N + j] = AS2[ASS * j] + off2;
    }
  }
  SG_T(16);
  SG_TEND();
}
