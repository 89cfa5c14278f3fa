// sg_general.h -- the GENERAL contact path: every collision pair the model class allows, in MuJoCo's order.
//
// The kernels' fast path knows one kind of contact: a moving finger box against an element capsule or the object's centre sphere,
// kept as two independent per-finger streams (sg_phase.hip, sg_rows.hip).  mj_collision for this model class also produces (SURVEY.md 8(a) a11,
// reference data/gripper/soft_grip_two_fingers.xml:55-94, soft_scene.xml:46)
//   plane - capsule, static box - capsule      (an element reaching the ground or the gripper's base: one slider, no finger),
//   plane - box, static box - finger box       (a finger reaching the ground or the base: one finger chain, up to 4 / 8 contacts),
//   finger box - finger box of the other hand  (the fingers closing on each other: BOTH chains in one constraint row).
// These never occur in the reference's three scenes, so they do not get a place in the two-stream layout; an env in which one of
// them comes within reach switches, for that substep, to this path: ONE ordered contact list over the plan's candidate-pair table
// (SgPlan::gpairs: body pairs ascending, geoms of the first body outer -- mj_collision's order, as the oracle's pair list), generic
// rows with both chains' Jacobian blocks and a slider column, and one serial Gauss-Seidel stream in the solver.
//
// Everything here is per-lane math (SG_HD): the phase / solver kernels call it on the device, tests/emu runs it lane-serially on the
// host against the oracle.  Narrowphase routines follow oracle/sg_oracle.c (box_box, plane_box, the plane - capsule branch of
// collision()), which documents what they restate and what they deviate from.
#pragma once
#include "sg_math.h"

namespace sgm {

// geometry reference: kind << 16 | index
enum { SGG_NONE = 0, SGG_PLANE = 1, SGG_STATIC = 2, SGG_CENTER = 3, SGG_BOX = 4, SGG_ELEM = 5 };
SG_HD int sgg_kind(int ref) { return ref >> 16; }
SG_HD int sgg_index(int ref) { return ref & 0xFFFF; }

// narrowphase routine of a candidate pair (geom1 / geom2 in mj_collideGeoms' order: by type, plane < sphere < capsule < box)
enum { SGP_PLANE_CAP = 0, SGP_PLANE_BOX = 1, SGP_SPH_BOX = 2, SGP_CAP_BOX = 3, SGP_BOX_BOX = 4, SGP_UNSUPPORTED = 5,
       SGP_PLANE_SPH = 6 /* tree plans with a free object only */ };
// the narrowphase routine a pair's two geometry references call for (a pair marked SGP_UNSUPPORTED -- other contact parameters than the
// plan's one set -- keeps its geometry: the tree pipeline walks it like any other pair and flags the env where it would touch)
SG_HD int sgp_geometry(int g1, int g2) {
  const int k1 = sgg_kind(g1), k2 = sgg_kind(g2);
  if (k1 == SGG_PLANE) return k2 == SGG_ELEM ? SGP_PLANE_CAP : (k2 == SGG_CENTER ? SGP_PLANE_SPH : SGP_PLANE_BOX);
  if (k1 == SGG_CENTER) return SGP_SPH_BOX;
  if (k1 == SGG_ELEM) return SGP_CAP_BOX;
  return SGP_BOX_BOX;
}
#define SG_GEN_MAXCON 112   // contacts of an env on the general path (the fast path: 64 per finger); with the box - box work space
                            // the list fills the phase kernel's 8 KB contact staging area
#define SG_GEN_ROUNDS ((SG_GEN_MAXCON + 63) / 64)
#define SG_GEN_W 48         // doubles per exported general contact (GenContact, padded)

SG_HD void make_frame_hint(const double* n, const double* hint, double* fr) {  // mju_makeFrame: normal, optional first-tangent hint
  double nn = sqrt(dot3(n, n));
  fr[0] = n[0] / nn; fr[1] = n[1] / nn; fr[2] = n[2] / nn;
  fr[3] = hint ? hint[0] : 0.0; fr[4] = hint ? hint[1] : 0.0; fr[5] = hint ? hint[2] : 0.0;
  if (sqrt(dot3(fr + 3, fr + 3)) < 0.5) {
    fr[3] = fr[4] = fr[5] = 0;
    if (fr[1] < 0.5 && fr[1] > -0.5) fr[4] = 1; else fr[5] = 1;
  }
  double t = dot3(fr, fr + 3);
  addscl3(fr + 3, fr, -t);
  nn = sqrt(dot3(fr + 3, fr + 3));
  fr[3] /= nn; fr[4] /= nn; fr[5] /= nn;
  cross3(fr + 6, fr, fr + 3);
}

// The narrowphase routines write records with members dist, pos[3], n[3] (ConRec on the host harness, the phase kernel's LDS staging
// record on the device).  Run-time indices into small private arrays are written as selects (pick3): a dynamically indexed private
// array would live in scratch memory on the device.
SG_HD double pick1(const double* v, int k) { return k == 0 ? v[0] : (k == 1 ? v[1] : v[2]); }
SG_HD void pick3(const double (*M)[3], int k, double* out) {
  for (int c = 0; c < 3; c++) out[c] = k == 0 ? M[0][c] : (k == 1 ? M[1][c] : M[2][c]);
}

// plane (point pp, unit normal pn) against a capsule: one sphere test per end cap (oracle collision(), plane - capsule branch)
template <class Rec>
SG_HD_HEAVY int gen_plane_capsule(const double* pp, const double* pn, const double* cp, const double* cax, double r, double hl, double margin,
                            Rec& o0, Rec& o1) {
  int n = 0;
  for (int s = -1; s <= 1; s += 2) {
    double c[3] = {cp[0] + s * hl * cax[0], cp[1] + s * hl * cax[1], cp[2] + s * hl * cax[2]};
    double e[3] = {c[0] - pp[0], c[1] - pp[1], c[2] - pp[2]}, dist = dot3(e, pn) - r;
    if (dist > margin) continue;
    Rec& o = n == 0 ? o0 : o1;
    o.dist = dist;
    for (int q = 0; q < 3; q++) { o.pos[q] = c[q] - pn[q] * (r + 0.5 * dist); o.n[q] = pn[q]; }
    n++;
  }
  return n;
}

// plane against a box: the corners within the margin, in corner order (x fastest), at most 4 (oracle plane_box)
template <class Rec>
SG_HD_HEAVY int gen_plane_box(const double* pp, const double* pn, const double* bp, const double* bm, const double* sz, double margin, Rec* out) {
  int n = 0;
  for (int q = 0; q < 8 && n < 4; q++) {
    double loc[3] = {(q & 1 ? 1 : -1) * sz[0], (q & 2 ? 1 : -1) * sz[1], (q & 4 ? 1 : -1) * sz[2]}, w[3];
    mulmat3(w, bm, loc);
    for (int c = 0; c < 3; c++) w[c] += bp[c] - pp[c];
    double dist = dot3(w, pn);
    if (dist > margin) continue;
    out[n].dist = dist;
    for (int c = 0; c < 3; c++) { out[n].pos[c] = w[c] + pp[c] - 0.5 * dist * pn[c]; out[n].n[c] = pn[c]; }
    n++;
  }
  return n;
}

// box - box: separating axes, then face clipping or the closest points of two edges (oracle box_box; up to 8 contacts).
// Normal from box 1 towards box 2, dist < 0 = penetration.  poly / tmp: work space for the clipped polygon, 16 points each.
#define SG_BB_FUDGE 1.05
template <class Rec>
SG_HD_HEAVY int gen_box_box(const double* p1, const double* R1, const double* s1, const double* p2, const double* R2, const double* s2, double margin,
                      Rec* out, double (*poly)[3], double (*tmp)[3]) {
  double T[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]}, A[3][3], B[3][3];
  for (int k = 0; k < 3; k++) { A[k][0] = R1[k]; A[k][1] = R1[3 + k]; A[k][2] = R1[6 + k]; B[k][0] = R2[k]; B[k][1] = R2[3 + k]; B[k][2] = R2[6 + k]; }
  double C[3][3], Q[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) { C[i][j] = dot3(A[i], B[j]); Q[i][j] = fabs(C[i][j]); }
  double best = -1e300, bn[3] = {0, 0, 0};
  int code = -1;  // 0..2 face of box 1, 3..5 face of box 2, 6 + 3 i + j edge i x edge j
  for (int k = 0; k < 3; k++) {
    double t = dot3(T, A[k]), sep = fabs(t) - (s1[k] + s2[0] * Q[k][0] + s2[1] * Q[k][1] + s2[2] * Q[k][2]);
    if (sep > margin) return 0;
    if (sep > best) { best = sep; code = k; double sg = t < 0 ? -1 : 1; for (int c = 0; c < 3; c++) bn[c] = sg * A[k][c]; }
  }
  for (int k = 0; k < 3; k++) {
    double t = dot3(T, B[k]), sep = fabs(t) - (s2[k] + s1[0] * Q[0][k] + s1[1] * Q[1][k] + s1[2] * Q[2][k]);
    if (sep > margin) return 0;
    if (sep > best) { best = sep; code = 3 + k; double sg = t < 0 ? -1 : 1; for (int c = 0; c < 3; c++) bn[c] = sg * B[k][c]; }
  }
  double ebest = -1e300, en[3] = {0, 0, 0}, euv = 0;
  int ecode = -1;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double L[3];
      cross3(L, A[i], B[j]);
      double n = sqrt(dot3(L, L));
      if (n < 1e-6) continue;  // parallel edges: covered by the face axes
      for (int c = 0; c < 3; c++) L[c] /= n;
      double ra = 0, rb = 0;
      for (int k = 0; k < 3; k++) { ra += s1[k] * fabs(dot3(L, A[k])); rb += s2[k] * fabs(dot3(L, B[k])); }
      double t = dot3(T, L), sep = fabs(t) - (ra + rb);
      if (sep > margin) return 0;
      if (sep > ebest) { ebest = sep; ecode = 6 + 3 * i + j; euv = C[i][j]; double sg = t < 0 ? -1 : 1; for (int c = 0; c < 3; c++) en[c] = sg * L[c]; }
    }
  if (ecode >= 0 && ebest > best + (SG_BB_FUDGE - 1.0) * fabs(best) + 1e-9) { best = ebest; code = ecode; bn[0] = en[0]; bn[1] = en[1]; bn[2] = en[2]; }
  if (code >= 6) {  // edge - edge: closest points of the two supporting edges
    const int i = (code - 6) / 3, j = (code - 6) % 3;
    double pa[3], pb[3], Ai[3], Bj[3];
    pick3(A, i, Ai); pick3(B, j, Bj);
    for (int c = 0; c < 3; c++) { pa[c] = p1[c]; pb[c] = p2[c]; }
    for (int k = 0; k < 3; k++) {
      if (k != i) { double sg = dot3(bn, A[k]) > 0 ? 1 : -1; addscl3(pa, A[k], sg * s1[k]); }
      if (k != j) { double sg = dot3(bn, B[k]) > 0 ? -1 : 1; addscl3(pb, B[k], sg * s2[k]); }
    }
    double r[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]}, uv = euv, den = 1 - uv * uv;
    double ta = 0, tb = 0;
    if (den > 1e-12) { double q1 = dot3(Ai, r), q2 = dot3(Bj, r); ta = (q1 - uv * q2) / den; tb = (uv * q1 - q2) / den; }
    const double s1i = pick1(s1, i), s2j = pick1(s2, j);
    ta = ta > s1i ? s1i : ta < -s1i ? -s1i : ta;
    tb = tb > s2j ? s2j : tb < -s2j ? -s2j : tb;
    out[0].dist = best;
    for (int c = 0; c < 3; c++) { out[0].pos[c] = 0.5 * ((pa[c] + ta * Ai[c]) + (pb[c] + tb * Bj[c])); out[0].n[c] = bn[c]; }
    return 1;
  }
  // face contact: the reference box owns the axis
  const bool ref1 = code < 3;
  const int ka = ref1 ? code : code - 3;
  double Ra[3][3], Rb[3][3], pa[3], pb[3], sa[3], sb[3];
  for (int k = 0; k < 3; k++) {
    for (int c = 0; c < 3; c++) { Ra[k][c] = ref1 ? A[k][c] : B[k][c]; Rb[k][c] = ref1 ? B[k][c] : A[k][c]; }
    pa[k] = ref1 ? p1[k] : p2[k]; pb[k] = ref1 ? p2[k] : p1[k]; sa[k] = ref1 ? s1[k] : s2[k]; sb[k] = ref1 ? s2[k] : s1[k];
  }
  double nrm[3];  // outward normal of the reference face (towards the incident box)
  for (int c = 0; c < 3; c++) nrm[c] = ref1 ? bn[c] : -bn[c];
  int kb = 0;
  double mx = -1;
  for (int k = 0; k < 3; k++) { double v = fabs(dot3(nrm, Rb[k])); if (v > mx + 1e-12) { mx = v; kb = k; } }
  double Rbk[3], Rbu[3], Rbv[3], Rau[3], Rav[3];
  const int u = (kb + 1) % 3, v = (kb + 2) % 3, ua = (ka + 1) % 3, va = (ka + 2) % 3;
  pick3(Rb, kb, Rbk); pick3(Rb, u, Rbu); pick3(Rb, v, Rbv); pick3(Ra, ua, Rau); pick3(Ra, va, Rav);
  const double sbk = pick1(sb, kb), sbu = pick1(sb, u), sbv = pick1(sb, v), sau = pick1(sa, ua), sav = pick1(sa, va), sak = pick1(sa, ka);
  const double sgb = dot3(nrm, Rbk) > 0 ? -1 : 1;
  double cen[3];
  for (int c = 0; c < 3; c++) cen[c] = pb[c] + sgb * sbk * Rbk[c];
  int np = 4;
  for (int q = 0; q < 4; q++) {
    const double cu = (q == 0 || q == 3) ? 1.0 : -1.0, cv = q < 2 ? 1.0 : -1.0;   // (1,1), (-1,1), (-1,-1), (1,-1)
    for (int c = 0; c < 3; c++) poly[q][c] = cen[c] + cu * sbu * Rbu[c] + cv * sbv * Rbv[c];
  }
  for (int pl = 0; pl < 4 && np > 0; pl++) {  // clip against the four side planes of the reference face
    const double* Rax = pl < 2 ? Rau : Rav;
    const double sg = (pl & 1) ? -1 : 1, lim = pl < 2 ? sau : sav;
    int nq = 0;
    for (int q = 0; q < np; q++) {
      const double *x0 = poly[q], *x1 = poly[(q + 1) % np];
      double e0[3] = {x0[0] - pa[0], x0[1] - pa[1], x0[2] - pa[2]}, e1[3] = {x1[0] - pa[0], x1[1] - pa[1], x1[2] - pa[2]};
      double d0 = sg * dot3(e0, Rax) - lim, d1 = sg * dot3(e1, Rax) - lim;
      if (d0 <= 0) { for (int c = 0; c < 3; c++) tmp[nq][c] = x0[c]; nq++; }
      if ((d0 <= 0) != (d1 <= 0)) { double w = d0 / (d0 - d1); for (int c = 0; c < 3; c++) tmp[nq][c] = x0[c] + w * (x1[c] - x0[c]); nq++; }
    }
    np = nq;
    for (int q = 0; q < np; q++)
      for (int c = 0; c < 3; c++) poly[q][c] = tmp[q][c];
  }
  int n = 0;
  for (int q = 0; q < np && n < 8; q++) {
    double e[3] = {poly[q][0] - pa[0], poly[q][1] - pa[1], poly[q][2] - pa[2]}, dist = dot3(e, nrm) - sak;
    if (dist > margin) continue;
    out[n].dist = dist;
    for (int c = 0; c < 3; c++) { out[n].pos[c] = poly[q][c] - 0.5 * dist * nrm[c]; out[n].n[c] = bn[c]; }
    n++;
  }
  return n;
}

// ------------------------------------------------------------------------------------------------
// general contact row: J = J(body of geom2) - J(body of geom1), each body a finger-chain body, an element slider or static
// ------------------------------------------------------------------------------------------------
struct GenContact {
  double Jf[SG_MAXCH][3][SG_CD];  // chain blocks (zero where the contact does not touch the chain)
  double Js[3];                   // slider column (0 when sl < 0)
  double A[6], b[3], f[3], R, invm;
  int sl, cmask;                  // slider or -1; bit c set: chain c is touched
};

// one side of a contact: which dofs move the body the geom sits on, and with which sign they enter the row
struct GenSide {
  int chain, nd;   // chain index or -1; dofs of the chain that move the body
  int sl;          // element index or -1
  double binvw;    // body_invweight0 (translational) of the body, 0 for a static one
};

// poses, sizes and sides of a geometry reference.  boxp / boxm: world poses of the SG_MAXCH * SG_CG finger boxes of this substep.
SG_HD void gen_box_of(int ref, const SgPlanHeader& H, const double (*boxp)[3], const double (*boxm)[9], const double*& p, const double*& R,
                      const double*& s, double& rbound) {
  const int i = sgg_index(ref);
  if (sgg_kind(ref) == SGG_BOX) { p = boxp[i]; R = boxm[i]; s = H.chain[i / SG_CG].g_size[i % SG_CG]; rbound = H.chain[i / SG_CG].g_rbound[i % SG_CG]; }
  else { p = H.st_pos[i]; R = H.st_mat[i]; s = H.st_size[i]; rbound = H.st_rbound[i]; }
}
SG_HD GenSide gen_side_of(int ref, const SgPlanHeader& H, const double* elem_binvw) {
  GenSide S;
  S.chain = -1; S.nd = 0; S.sl = -1; S.binvw = 0;
  const int i = sgg_index(ref);
  if (sgg_kind(ref) == SGG_BOX) {
    const SgChain& C = H.chain[i / SG_CG];
    const int bi = C.g_body[i % SG_CG];
    S.chain = i / SG_CG; S.nd = chain_ndof_of_body(bi); S.binvw = C.b_invw_tran[bi];
  } else if (sgg_kind(ref) == SGG_ELEM) {
    S.sl = i; S.binvw = elem_binvw[i];
  }
  return S;
}

// geometry -> rows.  K / Minv / vc / asm_c / warm_c: per chain (index 0, 1); slider quantities of the (at most one) element involved.
template <class Rec>
SG_HD void gen_contact_build(GenContact& c, const Rec& rec, const double* hint, const GenSide& S1, const GenSide& S2, const ChainKin* K,
                             const double (*Minv)[16], const double (*vc)[SG_CD], const double (*asm_c)[SG_CD], const double (*warm_c)[SG_CD],
                             const double* eaxis, double ve, double asm_e, double warm_e, double invm_e, const SgPlanHeader& H) {
  double fr[9];
  make_frame_hint(rec.n, hint, fr);
  c.sl = S1.sl >= 0 ? S1.sl : S2.sl;
  c.invm = c.sl >= 0 ? invm_e : 0.0;
  c.cmask = (S1.chain >= 0 ? 1 << S1.chain : 0) | (S2.chain >= 0 ? 1 << S2.chain : 0);
  for (int ch = 0; ch < SG_MAXCH; ch++)
    for (int r = 0; r < 3; r++)
      for (int d = 0; d < SG_CD; d++) c.Jf[ch][r][d] = 0.0;
  for (int r = 0; r < 3; r++) c.Js[r] = 0.0;
  for (int side = 0; side < 2; side++) {
    const GenSide& S = side ? S2 : S1;
    const double sg = side ? 1.0 : -1.0;   // geom2's body enters with +, geom1's with -
    if (S.chain >= 0)
      for (int d = 0; d < S.nd; d++) {
        double jp[3];
        chain_jacp(K[S.chain], d, rec.pos, jp);
        for (int r = 0; r < 3; r++) c.Jf[S.chain][r][d] += sg * dot3(fr + 3 * r, jp);
      }
    if (S.sl >= 0)
      for (int r = 0; r < 3; r++) c.Js[r] += sg * dot3(fr + 3 * r, eaxis);
  }
  const double dA = S1.binvw + S2.binvw, imp = impedance(H.con_solimp, rec.dist, H.con_margin);
  c.R = fmax(SG_MINVAL, (1 - imp) / imp * dA);
  const double D = 1 / c.R;
  double W[SG_MAXCH][3][SG_CD];
  for (int ch = 0; ch < SG_MAXCH; ch++)
    for (int r = 0; r < 3; r++)
      for (int d = 0; d < SG_CD; d++) {
        double s = 0;
        for (int e = 0; e < SG_CD; e++) s += c.Jf[ch][r][e] * Minv[ch][4 * e + d];
        W[ch][r][d] = s;
      }
  int k = 0;
  for (int r = 0; r < 3; r++)
    for (int s2 = r; s2 < 3; s2++) {
      double s = c.Js[r] * c.Js[s2] * c.invm + (r == s2 ? c.R : 0.0);
      for (int ch = 0; ch < SG_MAXCH; ch++)
        for (int d = 0; d < SG_CD; d++) s += W[ch][r][d] * c.Jf[ch][s2][d];
      c.A[k++] = s;
    }
  double jar[3];
  for (int r = 0; r < 3; r++) {
    double vel = c.Js[r] * ve, js = c.Js[r] * asm_e, jw = c.Js[r] * warm_e;
    for (int ch = 0; ch < SG_MAXCH; ch++)
      for (int d = 0; d < SG_CD; d++) { vel += c.Jf[ch][r][d] * vc[ch][d]; js += c.Jf[ch][r][d] * asm_c[ch][d]; jw += c.Jf[ch][r][d] * warm_c[ch][d]; }
    const double aref = -H.con_B * vel - (r == 0 ? H.con_K * imp * (rec.dist - H.con_margin) : 0.0);
    c.b[r] = js - aref;
    jar[r] = jw - aref;
  }
  // warmstart force: primal -> dual map of the elliptic cone (mj_constraintUpdate), as contact_build
  const double mu = H.con_mu[0], U0 = jar[0] * mu, U1 = jar[1] * H.con_mu[0], U2 = jar[2] * H.con_mu[1];
  const double N = U0, T = sqrt(U1 * U1 + U2 * U2);
  if (N >= mu * T || (T <= 0 && N >= 0)) { c.f[0] = c.f[1] = c.f[2] = 0; }
  else if (mu * N + T <= 0 || (T <= 0 && N < 0)) { for (int r = 0; r < 3; r++) c.f[r] = -D * jar[r]; }
  else {
    const double Dm = D / (mu * mu * (1 + mu * mu)), NmT = N - mu * T;
    c.f[0] = -Dm * NmT * mu;
    c.f[1] = -c.f[0] / T * U1 * H.con_mu[0];
    c.f[2] = -c.f[0] / T * U2 * H.con_mu[1];
  }
}

// Gauss-Seidel block update of a general contact: aF[c][d] = current M^-1 J' f of the chains, as_ = of the slider
SG_HD double gen_contact_update(GenContact& c, const double (*aF)[SG_CD], double as_, const double* mu, double* df) {
  // the residual is the only place the chains enter: hand contact_update a record whose finger block carries chain 0 and add
  // chain 1's share to its right-hand side for the call (b is restored afterwards)
  Contact t;
  for (int r = 0; r < 3; r++) {
    for (int d = 0; d < SG_CD; d++) t.Jf[r][d] = c.Jf[0][r][d];
    double extra = 0;
    for (int d = 0; d < SG_CD; d++) extra += c.Jf[1][r][d] * aF[1][d];
    t.Js[r] = c.Js[r]; t.b[r] = c.b[r] + extra; t.f[r] = c.f[r];
  }
  for (int q = 0; q < 6; q++) t.A[q] = c.A[q];
  t.R = c.R; t.invm = c.invm; t.sl = c.sl;
  const double change = contact_update(t, aF[0], as_, mu, df);
  for (int r = 0; r < 3; r++) c.f[r] = t.f[r];
  return change;
}

// exported record (SG_GEN_W doubles): Jf[2][3][4] | Js[3] | A[6] | b[3] | f[3] | R | invm | sl, cmask (as ints in one double slot each)
SG_HD void gen_contact_store(double* o, const GenContact& c) {
  int k = 0;
  for (int ch = 0; ch < SG_MAXCH; ch++)
    for (int r = 0; r < 3; r++)
      for (int d = 0; d < SG_CD; d++) o[k++] = c.Jf[ch][r][d];
  for (int r = 0; r < 3; r++) o[k++] = c.Js[r];
  for (int q = 0; q < 6; q++) o[k++] = c.A[q];
  for (int r = 0; r < 3; r++) o[k++] = c.b[r];
  for (int r = 0; r < 3; r++) o[k++] = c.f[r];
  o[k++] = c.R; o[k++] = c.invm;
  o[k++] = (double)c.sl; o[k++] = (double)c.cmask;
}
#define SG_GEN_F_OFF (SG_MAXCH * 3 * SG_CD + 3 + 6 + 3)   // offset of f[3] in the exported record
SG_HD void gen_contact_load(GenContact& c, const double* o) {
  int k = 0;
  for (int ch = 0; ch < SG_MAXCH; ch++)
    for (int r = 0; r < 3; r++)
      for (int d = 0; d < SG_CD; d++) c.Jf[ch][r][d] = o[k++];
  for (int r = 0; r < 3; r++) c.Js[r] = o[k++];
  for (int q = 0; q < 6; q++) c.A[q] = o[k++];
  for (int r = 0; r < 3; r++) c.b[r] = o[k++];
  for (int r = 0; r < 3; r++) c.f[r] = o[k++];
  c.R = o[k++]; c.invm = o[k++];
  c.sl = (int)o[k++]; c.cmask = (int)o[k++];
}

}  // namespace sgm
