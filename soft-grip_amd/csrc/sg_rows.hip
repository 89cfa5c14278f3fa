// sg_rows.hip -- the solver of the rows pipeline: mj_fwdConstraint's PGS (soft_scene.xml:13: PGS, 30 iterations, tolerance 1e-7,
// elliptic cones) for one physics substep of every env, reference environment/manenv.py:48-49.
#include "sg_work.h"

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
// EPW: envs per wavefront, 8 lanes each: 8 fills the wavefront (16 finger streams advance per instruction); 4 leaves lanes 32 .. 63 idle
// but spreads a batch of 4096 envs over 1024 wavefronts -- one per SIMD of the whole chip instead of half of it -- and a wavefront
// then runs the QCQP fallback (entered when ANY of its streams slides, for as many Newton evaluations as its slowest stream needs)
// for 8 streams instead of 16.  The wavefront's instruction stream is what a launch waits for, not its lane count.
// NB: 0 = fix rows only; 1 = neighbour rows, the rows' step factors c in LDS beside their states (models whose four envs then still fit
// a quarter of a CU's LDS: the box scene); 2 = neighbour rows, the step factors streamed from memory (SgWork::cst: the ball, the cylinder)
template <int NSL, int NB, int EPW>  // NSL >= ceil(nelem / 8): joint-fix rows per lane, unrolled and padded (straight-line code, LDS reads issue back to back)
__global__ __launch_bounds__(64) void sg_pgs_rows_kernel(SgPgsArgs a) {
  extern __shared__ double lds[];
  // an env's lane group: 8 lanes (fix-only models: two finger quads); 16 lanes for neighbour-row models (EPW = 4): the two finger
  // quads (g < 8) plus two more quads that only work in the equality block (one block of rows per quad and round)
  static_assert(!NB || EPW == 4, "neighbour-row models: four envs of 16 lanes per wavefront");
  constexpr bool CST = NB == 2;
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
  constexpr int NA = NR + 8;     // (static bound of the staging loops)
  const int NAr = SG_ROWS_NA_NB(N);  // words of an env's slider array: N sliders, the zero word, padding to a 16-byte multiple
  const int GW = (CST ? 4 : 8) * (N + 1);    // doubles of an env's row records: (N + 1) groups of four states g -- or four pairs (g, c)
  double2* const AF = (double2*)lds + (size_t)lec * NR;
  double2* const BR = (double2*)lds + (size_t)EPW * NR + (size_t)lec * NR;
  double* const RI = lds + (size_t)4 * EPW * NR + (size_t)lec * NR;
  double2* const IC = (double2*)(lds + (size_t)5 * EPW * NR);  // fix-only models: (1/m, tendon coefficient) per element
  // NB layout: Ae[NAr] per env: slider accelerations MINUS the env's offset aoff (below); word N is a zero word ("no partner").
  // GS per env: group e = the states g = b + R f of the four rows of element e's block -- its fix row, then its up to three
  // neighbour rows; rows that do not exist and group N (idle slots) hold 0 (their step factor in W.cst is 0 too); NB = 1: pairs (g, c)
  // with c = (1/m) / (A + R) in the same order.  TAB (shared by the
  // wavefront's envs): the plan's block schedule as LDS offsets per lane of a 16-lane group: lane 2 b + h holds, for the block e
  // in slot b of the round, x | y << 11 | (2 e + h) << 22 with the sliders' byte offsets (x, y) = 8 (e, p0) for h = 0 and 8 (p1, p2)
  // for h = 1, and 2 e + h the lane's pair of row states in 16-byte units (NB = 2; NB = 1: two words, x | y << 16 and the byte offset of
  // the lane's pair of 32-byte records -- TABW below).
  double* const Ae = lds + (size_t)lec * NAr;
  double* const GS = lds + (size_t)EPW * NAr + (size_t)lec * GW;
  unsigned* const TAB = (unsigned*)(lds + (size_t)EPW * NAr + (size_t)EPW * GW);
  // (NB = 1, the box scene: TWO words per lane and round, (x | y << 16, record offset) -- halves and a plain word decode inside the address
  //  additions; the packed word costs four more integer instructions per round: the 8 us per launch that r04 lost against r03 on one box,
  //  profiles/r05_ab_rounds.txt.  NB = 2 keeps the packed word: its LDS block has no 2.5 KB to spare)
  constexpr int TABW = CST ? 1 : 2;
  double* const Lnb = lds + (size_t)EPW * NAr + (size_t)EPW * GW + (size_t)8 * TABW * (H.eq_rounds + 8);  // [72 + 2 EPW]
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
      double* const A2 = lds + (size_t)e2 * NAr;
      double ssum = 0.0;
#pragma unroll
      for (int j0 = 0; j0 < NA; j0 += 64) {
        const int j = j0 + lane;
        const double av = (v2 && j < N) ? W.as[(size_t)env2 * N + j] : 0.0;
        if (j < NAr) A2[j] = av;
        ssum += av;
      }
      ssum = wave_sum2(ssum);
      if (lane == 0) { Lenv[e2] = ssum; Lenv[EPW + e2] = 0.0; }
    }
  }
  if constexpr (NB) {
    // row states of all the wavefront's envs, lane = row: the loads of a row (one table word, three values per env) are
    // independent of each other and of the other rows' -- they are issued together, not one memory round trip after the other
    [[maybe_unused]] const double im0s = 1.0 / (a.elem[(size_t)SGE_MASS * N] + a.elem[(size_t)SGE_ARMATURE * N]);
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
        double* const GS2 = lds + (size_t)EPW * NAr + (size_t)e2 * GW;   // group e = rows 4 e .. 4 e + 3
        const double gg = (v2 && have) ? bb[e2] + Rr[e2] * ff[e2] : 0.0;
        if constexpr (CST) GS2[u] = gg;
        else ((double2*)GS2)[u] = make_double2(gg, (v2 && have) ? sg_div(im0s, (fix ? im0s : 2.0 * im0s) + Rr[e2]) : 0.0);
      }
    }
  }
  if constexpr (NB) {  // the schedule (plus eight idle rounds: padding to a multiple of four and the look-ahead) as LDS offsets: lane 2 b + h of a group, block slot b
    const int ntab = 16 * TABW * (H.eq_rounds + 8);   // a.tab: the same words, laid out by the host once per batch (sg_api.hip)
#pragma unroll 4
    for (int i = lane; i < ntab; i += 64) TAB[i] = a.tab[i];
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

  // NB: the equality rows' step factors, round k of my lane: cp0[64 k] -- 64 lanes x 16 bytes contiguous per round, written by the
  // phase kernel in this order (SgWork::cst); the first four rounds' are on their way from here on
  [[maybe_unused]] const double2* const cp0 = (const double2*)(W.cst + SG_CST_INDEX(blockIdx.x, 0, lane, H.eq_rounds + 8));
  [[maybe_unused]] double2 c0, c1, c2, c3;
  if constexpr (CST) { c0 = cp0[0]; c1 = cp0[64]; c2 = cp0[128]; c3 = cp0[192]; }
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
        char* const Gb = (char*)GS;
        using TabW = std::conditional_t<CST, unsigned, uint2>;
        const TabW* tp = (const TabW*)TAB + g;
        const double2* cp = cp0;   // my step factors (c of my two rows), round k: cp0[64 k]
        const int hh = g & 1;
        const double hm = hh ? 1.0 : 0.0, h0 = 1.0 - hm;
        const double Pfix = h0 * -aoff;  // row 0's "partner" -aoff on lane 0 (added to hm * X)
        double qc = 0.0, sc = 0.0;
        const int nrounds = H.eq_rounds;
        struct Adr { double *px, *py; double2* rec; };   // lane 0: slider e itself / partner of row 1; lane 1: partners of rows 2, 3; my rows' states
        struct Off { double sA, wA, sB, wB, ga, gb; };   // what a round leaves for its off-chain part
        auto adr_of = [&](const TabW tt) {
          Adr q;
          if constexpr (CST) { q.px = (double*)(Ab + (tt & 0x7ffu)); q.py = (double*)(Ab + ((tt >> 11) & 0x7ffu)); q.rec = (double2*)(Gb + ((tt >> 22) << 4)); }
          else { q.px = (double*)(Ab + (tt.x & 0xffffu)); q.py = (double*)(Ab + (tt.x >> 16)); q.rec = (double2*)(Gb + tt.y); }
          return q;
        };
        // The CHAIN of a round -- slider reads -> two rows on lane 0 -> hand-over -> two rows on lane 1 -> hand-over -> slider writes, which
        // the next round's reads wait for -- is ~250 cycles of latency; the round's other ~24 instructions (cost sums, new row states and
        // their store, the next states / table words / addresses) are OFF the chain.  r03 measured 459 cycles per round against
        // 44 x 7 = 308 of issue: so the loop is written as chain(r) | reads(r + 1) | off-chain(r), with scheduling barriers between the
        // groups, and the off-chain work of a round runs in the shadow of the next round's LDS reads.
        auto chain = [&](const Adr q, const double2 gg, const double2 cc, const double X, const double Y, Off& o) {
          const double cA = cc.x, cB = cc.y;
          const double PA = fma(X, hm, Pfix), dA = gg.x - PA, dB = gg.y - Y;
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
          if constexpr (CST) *q.rec = make_double2(o.ga, o.gb);  // the rows' new states g' = P' - e' (their residual is zero after the update)
          else { q.rec[0].x = o.ga; q.rec[1].x = o.gb; }
        };
        // my two rows' states and step factors: one 16-byte LDS read + the stream's pair (CST), or two records (g, c) from LDS
        auto ld_rec = [&](const Adr q, double2& gg, double2& cc) {
          if constexpr (CST) gg = *q.rec;
          else { const double2 ra = q.rec[0], rb = q.rec[1]; gg = make_double2(ra.x, rb.x); cc = make_double2(ra.y, rb.y); }
        };
        // Four rounds per trip: slider words, states and addresses in two register sets (A, B) as before -- table words fetched two
        // rounds ahead, states and addresses one round ahead -- and the step factors in four (c0 .. c3), each requested again right
        // after the chain that consumed it: FOUR rounds (~1 400 cycles) ahead, which covers a miss in the Infinity Cache (two rounds
        // ahead left ~150 cycles of every round waiting: 0.405 -> 0.448 ms per launch on the box scene).  The factors do not change
        // over a substep's sweeps, so the last trip of a sweep requests the first four rounds again -- they arrive during the contact
        // phase -- and the round count is padded to a multiple of four with idle rounds (table and stream end with eight).
        TabW tC = tp[0], tD = tp[16];
        Adr qA = adr_of(tC), qB = adr_of(tD);
        double2 gA, gB;
        ld_rec(qA, gA, c0); ld_rec(qB, gB, c1);   // (NB = 1: the LDS path keeps a set's factors in c0 / c1)
        double XA = *qA.px, YA = *qA.py, XB, YB;
        Off oA, oB;
        constexpr int TRIP = CST ? 4 : 2;   // rounds per trip (LDS path: two, as ever -- an odd count runs one idle round)
        const int P = (nrounds + TRIP - 1) & ~(TRIP - 1);
        for (int k = 0; k < P; k += TRIP) {
          [[maybe_unused]] const double2* const cpn = k + 4 < P ? cp + 256 : cp0;   // (uniform)
#define SG_EQ_ROUND_PAIR(CA, CB, IA, IB, LA, LB)        \
          tp += 32;                                     \
          tC = tp[0]; tD = tp[16];                      \
          chain(qA, gA, CA, XA, YA, oA);                \
          if constexpr (CST) CA = cpn[IA];              \
          XB = *qB.px; YB = *qB.py;                     \
          __builtin_amdgcn_sched_barrier(0);            \
          offchain(qA, oA);                             \
          qA = adr_of(tC); ld_rec(qA, gA, LA);          \
          __builtin_amdgcn_sched_barrier(0);            \
          chain(qB, gB, CB, XB, YB, oB);                \
          if constexpr (CST) CB = cpn[IB];              \
          XA = *qA.px; YA = *qA.py;                     \
          __builtin_amdgcn_sched_barrier(0);            \
          offchain(qB, oB);                             \
          qB = adr_of(tD); ld_rec(qB, gB, LB);          \
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (CST) {   // (LA, LB: unused -- ld_rec reads the states only)
            SG_EQ_ROUND_PAIR(c0, c1, 0, 64, c0, c1)
            SG_EQ_ROUND_PAIR(c2, c3, 128, 192, c2, c3)
          } else {               // LDS path: a set's next record brings its factors along (c0: set A, c1: set B)
            SG_EQ_ROUND_PAIR(c0, c1, 0, 64, c0, c1)
          }
#undef SG_EQ_ROUND_PAIR
          cp = cpn;
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
      const double* const AS2 = NB ? lds + (size_t)e2 * NAr : lds + (size_t)2 * e2 * NR;
      const double off2 = NB ? Lenv[EPW + e2] : 0.0;
      for (int j = lane; j < N; j += 64) W.as[(size_t)env2 * N + j] = AS2[ASS * j] + off2;
    }
  }
  SG_T(16);
  SG_TEND();
}


// ---- launchers ----
static const int sg_nsl_set[] = {8, 14, 20, 26, 29, 32};
int sg_rows_nsl(int nelem) {
  for (int v : sg_nsl_set)
    if (v * 8 >= nelem) return v;
  return 32;
}
hipError_t sg_rows_prepare() {
  hipError_t e = hipSuccess;
#define SG_ATTR1(v, nb, ep)                                                                                                        \
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)sg_pgs_rows_kernel<v, nb, ep>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
#define SG_ATTR(v) SG_ATTR1(v, 0, 8); SG_ATTR1(v, 0, 4); SG_ATTR1(v, 1, 4); SG_ATTR1(v, 2, 4)
  SG_ATTR(8); SG_ATTR(14); SG_ATTR(20); SG_ATTR(26); SG_ATTR(29); SG_ATTR(32);
#undef SG_ATTR
#undef SG_ATTR1
  return e;
}
hipError_t sg_launch_rows(const SgPgsArgs& ga, int nsl, int nb, int epw, int nenv, size_t lds_bytes, hipStream_t s) {
  const dim3 grid((nenv + epw - 1) / epw);
#define SG_ROWS1(v, nbv, e) hipLaunchKernelGGL((sg_pgs_rows_kernel<v, nbv, e>), grid, dim3(64), lds_bytes, s, ga)
#define SG_ROWS(v)                                                                    \
  case v:                                                                             \
    if (nb == 2) SG_ROWS1(v, 2, 4);                                                   \
    else if (nb == 1) SG_ROWS1(v, 1, 4);                                              \
    else { if (epw == 8) SG_ROWS1(v, 0, 8); else SG_ROWS1(v, 0, 4); }                 \
    break
  switch (nsl) {
    SG_ROWS(8); SG_ROWS(14); SG_ROWS(20); SG_ROWS(26); SG_ROWS(29);
    default:
    SG_ROWS(32);
  }
#undef SG_ROWS
#undef SG_ROWS1
  return hipGetLastError();
}
