// sg_math.h -- per-lane scalar building blocks of the soft-gripper step kernels.
//
// Everything here is a pure function of its arguments (no cross-lane traffic, no memory
// side effects besides its out-parameters), compiled both for gfx950 device code
// (sg_phase.hip; the test build's sg_kernels.hip) and for the host, where tests/emu drives the very same functions lane
// by lane to validate the matrix-free restructuring against the CPU oracle.
//
// Stage numbers refer to SURVEY.md App. B (the mj_step pipeline behind
// reference environment/manenv.py:49).
#pragma once
#include <math.h>

#include "sg_plan.h"

#if defined(__HIPCC__)
#define SG_HD __host__ __device__ __forceinline__
#else
#define SG_HD inline
#endif
// the narrowphase routines: forced inline everywhere but in the tree pipeline's translation unit (sg_tree.hip defines SG_HD_HEAVY as a
// called function: one env's whole step is ONE kernel there, and with every geometry routine pasted into it the register allocator
// spilled 800 scalar + 500 vector registers)
#ifndef SG_HD_HEAVY
#define SG_HD_HEAVY SG_HD
#endif

#define SG_MINVAL 1e-15
#define SG_MAXVAL 1e10
#define SG_MINIMP 1e-4
#define SG_MAXIMP 0.9999
#define SG_MAXLIM 8  // limit rows per chain (2 sides x 4 dofs)

namespace sgm {

SG_HD double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
SG_HD void cross3(double* r, const double* a, const double* b) {
  double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z;
}
SG_HD void addscl3(double* r, const double* a, double s) { r[0] += a[0] * s; r[1] += a[1] * s; r[2] += a[2] * s; }
SG_HD void mulmat3(double* r, const double* M, const double* v) {
  double x = M[0] * v[0] + M[1] * v[1] + M[2] * v[2], y = M[3] * v[0] + M[4] * v[1] + M[5] * v[2], z = M[6] * v[0] + M[7] * v[1] + M[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
SG_HD void mulmatT3(double* r, const double* M, const double* v) {
  double x = M[0] * v[0] + M[3] * v[1] + M[6] * v[2], y = M[1] * v[0] + M[4] * v[1] + M[7] * v[2], z = M[2] * v[0] + M[5] * v[1] + M[8] * v[2];
  r[0] = x; r[1] = y; r[2] = z;
}
SG_HD void mulmat33(double* r, const double* A, const double* B) {
  double t[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
#pragma unroll
  for (int i = 0; i < 9; i++) r[i] = t[i];
}
SG_HD void quat2mat(double* M, const double* q) {
  double w = q[0], x = q[1], y = q[2], z = q[3];
  M[0] = w * w + x * x - y * y - z * z; M[1] = 2 * (x * y - w * z); M[2] = 2 * (x * z + w * y);
  M[3] = 2 * (x * y + w * z); M[4] = w * w - x * x + y * y - z * z; M[5] = 2 * (y * z - w * x);
  M[6] = 2 * (x * z - w * y); M[7] = 2 * (y * z + w * x); M[8] = w * w - x * x - y * y + z * z;
}
SG_HD void quatmul(double* r, const double* a, const double* b) {
  double t0 = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], t1 = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
         t2 = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], t3 = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  r[0] = t0; r[1] = t1; r[2] = t2; r[3] = t3;
}
SG_HD bool isbad(double x) { return !(x == x) || x > SG_MAXVAL || x < -SG_MAXVAL; }

// a / b for the solver's inner loop.  Device: v_rcp_f64 + two Newton steps (<= 2 ulp, half the latency of the
// IEEE sequence); host: plain division.
SG_HD double sg_div(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(b);
  r = fma(fma(-b, r, 1.0), r, r);
  r = fma(fma(-b, r, 1.0), r, r);
  return a * r;
#else
  return a / b;
#endif
}

// a / b for Newton corrections whose last bits do not matter (the next evaluation absorbs them): one Newton step on v_rcp_f64
SG_HD double sg_div_fast(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(b);
  r = fma(fma(-b, r, 1.0), r, r);
  return a * r;
#else
  return a / b;
#endif
}

// "some lane of the wavefront has x": a wavefront-uniform condition, so the guarded block is a real (scalar) branch that the
// compiler cannot flatten into predicated code executed by everyone.  Host (lane-serial emulation): just x.
SG_HD bool sg_any(bool x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_ballot_w64(x) != 0;
#else
  return x;
#endif
}

// constraint impedance d(pos) (App. B.5)
SG_HD double impedance(const double* si, double pos, double margin) {
  double s0 = fmin(SG_MAXIMP, fmax(SG_MINIMP, si[0])), s1 = fmin(SG_MAXIMP, fmax(SG_MINIMP, si[1])), s2 = fmax(0.0, si[2]),
         s3 = fmin(SG_MAXIMP, fmax(SG_MINIMP, si[3])), s4 = fmax(1.0, si[4]);
  if (s0 == s1 || s2 <= SG_MINVAL) return 0.5 * (s0 + s1);
  double x = fabs((pos - margin) / s2);
  if (x >= 1 || x <= 0) return x >= 1 ? s1 : s0;
  double y;
  if (s4 == 1) y = x;
  else if (s4 == 2) y = x <= s3 ? x * x / s3 : 1 - (1 - x) * (1 - x) / (1 - s3);
  else if (x <= s3) y = pow(x, s4) / pow(s3, s4 - 1);
  else y = 1 - pow(1 - x, s4) / pow(1 - s3, s4 - 1);
  return s0 + y * (s1 - s0);
}

// ------------------------------------------------------------------------------------
// Finger chain: kinematics, mass matrix, bias, tendon, actuation (stages 1,3,4,7,8,9)
// ------------------------------------------------------------------------------------
struct ChainKin {
  double xaxis[SG_CD][3], xanchor[SG_CD][3];
  double xpos[SG_CB][3], xmat[SG_CB][9];
};

SG_HD void chain_kinematics(const SgChain& C, const double* q, ChainKin& K) {
  double ppos[3] = {C.root_pos[0], C.root_pos[1], C.root_pos[2]}, pmat[9], pquat[4] = {1, 0, 0, 0};
  // orientations are carried as (root matrix) * (quaternion relative to the root)
  double rootmat[9];
#pragma unroll
  for (int i = 0; i < 9; i++) { pmat[i] = C.root_mat[i]; rootmat[i] = C.root_mat[i]; }
#pragma unroll
  for (int bi = 0; bi < SG_CB; bi++) {
    double t[3], pos[3], quat[4], mat[9], lm[9];
    mulmat3(t, pmat, C.b_pos[bi]);
#pragma unroll
    for (int c = 0; c < 3; c++) pos[c] = ppos[c] + t[c];
    quatmul(quat, pquat, C.b_quat[bi]);
#pragma unroll
    for (int k = 0; k < SG_CJ; k++) {
      const int d = SG_CJ * bi + k;
      quat2mat(lm, quat);
      mulmat33(mat, rootmat, lm);
      mulmat3(t, mat, C.j_pos[d]);
#pragma unroll
      for (int c = 0; c < 3; c++) K.xanchor[d][c] = pos[c] + t[c];
      mulmat3(K.xaxis[d], mat, C.j_axis[d]);
      double dq = q[d] - C.qpos0[d], s, cs;
      sincos(0.5 * dq, &s, &cs);
      double ql[4] = {cs, C.j_axis[d][0] * s, C.j_axis[d][1] * s, C.j_axis[d][2] * s};
      quatmul(quat, quat, ql);
      quat2mat(lm, quat);
      mulmat33(mat, rootmat, lm);
      mulmat3(t, mat, C.j_pos[d]);
#pragma unroll
      for (int c = 0; c < 3; c++) pos[c] = K.xanchor[d][c] - t[c];
    }
    double n = sqrt(quat[0] * quat[0] + quat[1] * quat[1] + quat[2] * quat[2] + quat[3] * quat[3]);
#pragma unroll
    for (int c = 0; c < 4; c++) quat[c] /= n;
    quat2mat(lm, quat);
    mulmat33(mat, rootmat, lm);
#pragma unroll
    for (int c = 0; c < 3; c++) { K.xpos[bi][c] = pos[c]; ppos[c] = pos[c]; }
#pragma unroll
    for (int c = 0; c < 9; c++) { K.xmat[bi][c] = mat[c]; pmat[c] = mat[c]; }
#pragma unroll
    for (int c = 0; c < 4; c++) pquat[c] = quat[c];
  }
}

// world pose of chain body bi (bi is run-time data: select, do not index)
SG_HD void chain_body_pose(const ChainKin& K, int bi, double* pos, double* mat) {
#pragma unroll
  for (int c = 0; c < 3; c++) pos[c] = bi == 0 ? K.xpos[0][c] : K.xpos[SG_CB - 1][c];
#pragma unroll
  for (int c = 0; c < 9; c++) mat[c] = bi == 0 ? K.xmat[0][c] : K.xmat[SG_CB - 1][c];
}

// translational / rotational Jacobian columns of a world point on chain body `bi`, for dofs 0..nd-1 (nd = dofs up to and incl. that body)
SG_HD int chain_ndof_of_body(int bi) { return SG_CJ * (bi + 1); }
SG_HD void chain_jacp(const ChainKin& K, int d, const double* point, double* jp) {
  double r[3] = {point[0] - K.xanchor[d][0], point[1] - K.xanchor[d][1], point[2] - K.xanchor[d][2]};
  cross3(jp, K.xaxis[d], r);
}

// 4x4 SPD inverse through LDL' (unused dofs carry a unit diagonal)
SG_HD void spd_inverse4(const double* M, double* Minv) {
  double L[16], D[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    double dj = M[4 * j + j];
    for (int k = 0; k < j; k++) dj -= L[4 * j + k] * L[4 * j + k] * D[k];
    D[j] = dj;
    for (int i = j + 1; i < 4; i++) {
      double s = M[4 * i + j];
      for (int k = 0; k < j; k++) s -= L[4 * i + k] * L[4 * j + k] * D[k];
      L[4 * i + j] = s / dj;
    }
  }
#pragma unroll
  for (int c = 0; c < 4; c++) {
    double x[4];
    for (int i = 0; i < 4; i++) {
      double s = (i == c) ? 1.0 : 0.0;
      for (int k = 0; k < i; k++) s -= L[4 * i + k] * x[k];
      x[i] = s;
    }
    for (int i = 0; i < 4; i++) x[i] /= D[i];
    for (int i = 3; i >= 0; i--) {
      double s = x[i];
      for (int k = i + 1; k < 4; k++) s -= L[4 * k + i] * x[k];
      x[i] = s;
    }
    for (int i = 0; i < 4; i++) Minv[4 * i + c] = x[i];
  }
}

struct ChainMotion {  // per body: angular velocity, angular acceleration, linear acceleration of the body origin
  double w[SG_CB][3], al[SG_CB][3], a[SG_CB][3];
};
// Tree walk of the oracle's tree_motion restricted to one serial hinge chain on a static root.
SG_HD void chain_motion(const SgChain& C, const ChainKin& K, const double* v, const double* qacc, const double* gravity, ChainMotion& Mo) {
  double w[3] = {0, 0, 0}, al[3] = {0, 0, 0}, a[3] = {-gravity[0], -gravity[1], -gravity[2]};
  double P[3] = {C.root_pos[0], C.root_pos[1], C.root_pos[2]};
#pragma unroll
  for (int bi = 0; bi < SG_CB; bi++) {
#pragma unroll
    for (int k = 0; k <= SG_CJ; k++) {
      const bool last = k == SG_CJ;
      const int d = last ? 0 : SG_CJ * bi + k;
      const double* Q = last ? K.xpos[bi] : K.xanchor[d];
      double r[3] = {Q[0] - P[0], Q[1] - P[1], Q[2] - P[2]}, t[3], t2[3];
      cross3(t, w, r);
      cross3(t2, al, r); addscl3(a, t2, 1);
      cross3(t2, w, t); addscl3(a, t2, 1);
      P[0] = Q[0]; P[1] = Q[1]; P[2] = Q[2];
      if (last) break;
      const double* u = K.xaxis[d];
      cross3(t, w, u);
      addscl3(al, u, qacc ? qacc[d] : 0.0);
      addscl3(al, t, v[d]);
      addscl3(w, u, v[d]);
    }
#pragma unroll
    for (int c = 0; c < 3; c++) { Mo.w[bi][c] = w[c]; Mo.al[bi][c] = al[c]; Mo.a[bi][c] = a[c]; }
  }
}

struct ChainDyn {
  double Minv[16];      // (M)^-1, 4x4 (unit diagonal on unused dofs)
  double M[16];
  double qfrc_smooth[SG_CD], qacc_smooth[SG_CD];
  double ten_len, ten_vel, ten_J[SG_CD], act_dot;
};

// M, M^-1, bias, passive, actuator for one chain.  stiff[d] = effective joint stiffness, ten_k = effective tendon stiffness.
SG_HD void chain_dynamics(const SgChain& C, const ChainKin& K, const double* q, const double* v, double act, double ctrl,
                          const double* stiff, double ten_k, const double* gravity, ChainDyn& D) {
#pragma unroll
  for (int i = 0; i < 16; i++) D.M[i] = 0;
#pragma unroll
  for (int d = 0; d < SG_CD; d++) D.M[5 * d] = d < C.ndof ? C.armature[d] : 1.0;
  double bias[SG_CD] = {0, 0, 0, 0};
  ChainMotion Mo;
  chain_motion(C, K, v, nullptr, gravity, Mo);
#pragma unroll
  for (int bi = 0; bi < SG_CB; bi++) {
    double com[3], t[3], Iw[9], RI[9], Rt[9];
    mulmat3(t, K.xmat[bi], C.b_ipos[bi]);
    for (int c = 0; c < 3; c++) com[c] = K.xpos[bi][c] + t[c];
    mulmat33(RI, K.xmat[bi], C.b_imat[bi]);
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) Rt[3 * a + b] = K.xmat[bi][3 * b + a];
    mulmat33(Iw, RI, Rt);
    const int nd = chain_ndof_of_body(bi);
    double jp[SG_CD][3], Ijr[SG_CD][3];
#pragma unroll
    for (int d = 0; d < SG_CD; d++) {
      if (d >= nd) break;
      chain_jacp(K, d, com, jp[d]);
      mulmat3(Ijr[d], Iw, K.xaxis[d]);
    }
    double mass = C.b_mass[bi];
#pragma unroll
    for (int a = 0; a < SG_CD; a++) {
      if (a >= nd) break;
#pragma unroll
      for (int b = 0; b < SG_CD; b++) {
        if (b >= nd) break;
        D.M[4 * a + b] += mass * dot3(jp[a], jp[b]) + dot3(Ijr[a], K.xaxis[b]);
      }
    }
    // RNE bias for this body (qacc = 0, world accelerating at -g)
    const double *w = Mo.w[bi], *al = Mo.al[bi];
    double cc[3] = {com[0] - K.xpos[bi][0], com[1] - K.xpos[bi][1], com[2] - K.xpos[bi][2]}, f[3], n[3], t2[3], Iww[3];
    for (int c = 0; c < 3; c++) f[c] = Mo.a[bi][c];
    cross3(t, al, cc); addscl3(f, t, 1);
    cross3(t, w, cc); cross3(t2, w, t); addscl3(f, t2, 1);
    for (int c = 0; c < 3; c++) f[c] *= mass;
    mulmat3(n, Iw, al);
    mulmat3(Iww, Iw, w);
    cross3(t, w, Iww); addscl3(n, t, 1);
#pragma unroll
    for (int d = 0; d < SG_CD; d++) {
      if (d >= nd) break;
      bias[d] += dot3(jp[d], f) + dot3(K.xaxis[d], n);
    }
  }
  spd_inverse4(D.M, D.Minv);
  // spatial tendon: fixed world site -> site on chain body
  D.ten_len = 0; D.ten_vel = 0; D.act_dot = 0;
#pragma unroll
  for (int d = 0; d < SG_CD; d++) D.ten_J[d] = 0;
  double frc_t = 0;
  if (C.has_ten) {
    double sp[3], t[3], tbp[3], tbm[9];
    chain_body_pose(K, C.ten_body, tbp, tbm);
    mulmat3(t, tbm, C.ten_site);
    for (int c = 0; c < 3; c++) sp[c] = tbp[c] + t[c];
    double dif[3] = {sp[0] - C.ten_fixed[0], sp[1] - C.ten_fixed[1], sp[2] - C.ten_fixed[2]}, len = sqrt(dot3(dif, dif));
    D.ten_len = len;
    if (len >= SG_MINVAL) {
      for (int c = 0; c < 3; c++) dif[c] /= len;
      const int nd = chain_ndof_of_body(C.ten_body);
#pragma unroll
      for (int d = 0; d < SG_CD; d++) {
        double jp[3];
        chain_jacp(K, d, sp, jp);
        D.ten_J[d] = d < nd ? dot3(dif, jp) : 0.0;  // direction sign cancels: J is the derivative of the (positive) length
      }
    }
#pragma unroll
    for (int d = 0; d < SG_CD; d++) D.ten_vel += D.ten_J[d] * v[d];
    frc_t = -ten_k * (len - C.ten_lspring) - C.ten_damping * D.ten_vel;
    if (C.has_act) {
      double g = C.act_gear;
      D.act_dot = (ctrl - act) / fmax(SG_MINVAL, C.act_tc);
      frc_t += g * (C.act_gain * act + C.act_bias[0] + C.act_bias[1] * g * len + C.act_bias[2] * g * D.ten_vel);
    }
  }
#pragma unroll
  for (int d = 0; d < SG_CD; d++) {
    double f = 0;
    if (d < C.ndof) f = -stiff[d] * (q[d] - C.springref[d]) - C.damping[d] * v[d] - bias[d] + D.ten_J[d] * frc_t;
    D.qfrc_smooth[d] = f;
  }
#pragma unroll
  for (int a = 0; a < SG_CD; a++) {
    double s = 0;
#pragma unroll
    for (int b = 0; b < SG_CD; b++) s += D.Minv[4 * a + b] * D.qfrc_smooth[b];
    D.qacc_smooth[a] = s;
  }
}

// ------------------------------------------------------------------------------------
// Collision (stage 5): sphere-box and capsule-box.  Normals point from the object geom
// (geom1: sphere / capsule) into the finger box (geom2), positions are mid-penetration.
// ------------------------------------------------------------------------------------
struct ConRec {
  double dist, pos[3], n[3];
};

SG_HD_HEAVY int sphere_box(const double* c, double r, const double* bp, const double* bm, const double* sz, double margin, ConRec& out) {
  double t[3] = {c[0] - bp[0], c[1] - bp[1], c[2] - bp[2]}, cen[3], dif[3];
  mulmatT3(cen, bm, t);
  for (int k = 0; k < 3; k++) {
    double cl = cen[k] < -sz[k] ? -sz[k] : cen[k] > sz[k] ? sz[k] : cen[k];
    dif[k] = cl - cen[k];
  }
  double dist = sqrt(dot3(dif, dif));
  if (dist - r > margin) return 0;
  double nl[3], pl[3], cd;
  if (dist <= SG_MINVAL) {
    double closest = 1e300, sg = 1;
    int ka = 0;
    for (int k = 0; k < 3; k++) {
      if (sz[k] - cen[k] < closest - 1e-12) { closest = sz[k] - cen[k]; ka = k; sg = 1; }
      if (sz[k] + cen[k] < closest - 1e-12) { closest = sz[k] + cen[k]; ka = k; sg = -1; }
    }
    for (int k = 0; k < 3; k++) nl[k] = (k == ka) ? -sg : 0.0;
    for (int k = 0; k < 3; k++) pl[k] = cen[k] + nl[k] * (r - closest) * 0.5;
    cd = -closest - r;
  } else {
    for (int k = 0; k < 3; k++) { nl[k] = dif[k] / dist; pl[k] = cen[k] + nl[k] * (r + dist) * 0.5; }
    cd = dist - r;
  }
  mulmat3(out.n, bm, nl);
  mulmat3(out.pos, bm, pl);
  for (int k = 0; k < 3; k++) out.pos[k] += bp[k];
  out.dist = cd;
  return 1;
}

SG_HD double box_sdist(const double* q, const double* sz) {
  double o2 = 0, in = -1e300;
  for (int k = 0; k < 3; k++) {
    double e = fabs(q[k]) - sz[k];
    if (e > 0) o2 += e * e;
    if (e > in) in = e;
  }
  return o2 > 0 ? sqrt(o2) : in;
}

// parameter t in [-1,1] of the point of the segment p + t*h (box frame) closest to / deepest in the solid box
SG_HD double seg_box_param(const double* p, const double* h, const double* sz) {
  double tk[8], gk[8];
  tk[0] = -1; tk[1] = 1;
#pragma unroll
  for (int k = 0; k < 3; k++)
#pragma unroll
    for (int s = 0; s < 2; s++) {
      double t = 2.0;  // sentinel: outside (-1,1)
      if (fabs(h[k]) > SG_MINVAL) t = ((s ? sz[k] : -sz[k]) - p[k]) / h[k];
      tk[2 + 2 * k + s] = (t > -1 && t < 1) ? t : 2.0;
    }
#pragma unroll
  for (int i = 0; i < 8; i++) {
    double g = 0, ti = tk[i] > 1.5 ? 0.0 : tk[i];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      double q = p[k] + ti * h[k];
      if (q > sz[k]) g += (q - sz[k]) * h[k];
      else if (q < -sz[k]) g += (q + sz[k]) * h[k];
    }
    gk[i] = g;
  }
  double tbest;
  if (gk[0] >= 0) tbest = -1;
  else if (gk[1] <= 0) tbest = 1;
  else {
    double tlo = -1, glo = gk[0], thi = 1, ghi = gk[1];
#pragma unroll
    for (int i = 2; i < 8; i++) {
      if (tk[i] > 1.5) continue;
      if (gk[i] <= 0 && tk[i] > tlo) { tlo = tk[i]; glo = gk[i]; }
      if (gk[i] >= 0 && tk[i] < thi) { thi = tk[i]; ghi = gk[i]; }
    }
    tbest = (ghi - glo > SG_MINVAL && thi > tlo) ? tlo + (thi - tlo) * (-glo) / (ghi - glo) : tlo;
  }
  double q[3] = {p[0] + tbest * h[0], p[1] + tbest * h[1], p[2] + tbest * h[2]};
  if (box_sdist(q, sz) > 0) return tbest;
  // the segment reaches into the box: minimise max_k(|q_k(t)| - s_k), a max of six lines, over [-1,1]
  double a[6], b[6];
  for (int k = 0; k < 3; k++) { a[2 * k] = h[k]; b[2 * k] = p[k] - sz[k]; a[2 * k + 1] = -h[k]; b[2 * k + 1] = -p[k] - sz[k]; }
  double best = 1e300, tb = -1;
  for (int ci = 0; ci < 17; ci++) {
    double t;
    if (ci == 0) t = -1;
    else if (ci == 1) t = 1;
    else {
      // enumerate pairs (i<j) in the order (0,1),(0,2)...(4,5)
      int idx = ci - 2, i = 0, rem = idx;
      while (rem >= 5 - i) { rem -= 5 - i; i++; }
      int j = i + 1 + rem;
      if (!(fabs(a[i] - a[j]) > SG_MINVAL)) continue;
      t = (b[j] - b[i]) / (a[i] - a[j]);
      if (!(t > -1 && t < 1)) continue;
    }
    double f = -1e300;
    for (int k = 0; k < 6; k++) { double v = a[k] * t + b[k]; if (v > f) f = v; }
    if (f < best - 1e-12) { best = f; tb = t; }  // ties: first candidate wins in every implementation
  }
  return tb;
}

// capsule (centre cp, unit axis cax, radius r, half length hl) against a box; up to two contacts.
// Returns a bit mask: bit 0 -> o0 valid (closest / deepest point), bit 1 -> o1 valid (far end cap).
SG_HD_HEAVY int capsule_box(const double* cp, const double* cax, double r, double hl, const double* bp, const double* bm, const double* sz,
                      double margin, ConRec& o0, ConRec& o1) {
  double t[3] = {cp[0] - bp[0], cp[1] - bp[1], cp[2] - bp[2]}, p[3], h[3];
  mulmatT3(p, bm, t);
  mulmatT3(h, bm, cax);
  for (int k = 0; k < 3; k++) h[k] *= hl;
  double t1 = seg_box_param(p, h, sz);
  double c1[3] = {cp[0] + cax[0] * hl * t1, cp[1] + cax[1] * hl * t1, cp[2] + cax[2] * hl * t1};
  int mask = sphere_box(c1, r, bp, bm, sz, margin, o0);
  double t2 = t1 >= 0 ? -1.0 : 1.0;
  if (fabs(t2 - t1) * hl > 1e-6) {
    double c2[3] = {cp[0] + cax[0] * hl * t2, cp[1] + cax[1] * hl * t2, cp[2] + cax[2] * hl * t2};
    mask |= sphere_box(c2, r, bp, bm, sz, margin, o1) << 1;
  }
  return mask;
}

// separating-axis overlap test of two boxes (detection only: such pairs are outside the supported envelope).  The axes are
// not normalised: |T.L| > ra + rb scales with |L| on both sides (the 9 edge-edge axes cost a square root and three divisions
// each otherwise); near-parallel edge pairs (|L|^2 < 1e-18) are skipped as before.
SG_HD bool box_box_overlap(const double* p1, const double* R1, const double* s1, const double* p2, const double* R2, const double* s2,
                           double margin) {
  double T[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]}, ax[6][3];
  for (int k = 0; k < 3; k++) {
    ax[k][0] = R1[k]; ax[k][1] = R1[3 + k]; ax[k][2] = R1[6 + k];
    ax[3 + k][0] = R2[k]; ax[3 + k][1] = R2[3 + k]; ax[3 + k][2] = R2[6 + k];
  }
  bool separated = false;
  for (int a = 0; a < 15; a++) {
    double L[3], n2 = 1.0;
    if (a < 6) { L[0] = ax[a][0]; L[1] = ax[a][1]; L[2] = ax[a][2]; }
    else {
      cross3(L, ax[(a - 6) / 3], ax[3 + (a - 6) % 3]);
      n2 = dot3(L, L);
    }
    double ra = 0, rb = 0;
    for (int k = 0; k < 3; k++) { ra += s1[k] * fabs(dot3(L, ax[k])); rb += s2[k] * fabs(dot3(L, ax[3 + k])); }
    const double nrm = a < 6 ? 1.0 : sqrt(n2);  // margin is a length: scale it like the projections (0 in every call of the kernels)
    if (n2 >= 1e-18 && fabs(dot3(T, L)) > ra + rb + margin * nrm) separated = true;
    if (!sg_any(!separated)) break;  // device: every lane of the wavefront that runs the test has found a separating axis
  }
  return !separated;
}

SG_HD void make_frame(const double* n, double* fr) {  // mju_makeFrame with an undefined tangent hint
  double nn = sqrt(dot3(n, n));
  fr[0] = n[0] / nn; fr[1] = n[1] / nn; fr[2] = n[2] / nn;
  fr[3] = 0; fr[4] = 0; fr[5] = 0;
  if (fr[1] < 0.5 && fr[1] > -0.5) fr[4] = 1; else fr[5] = 1;
  double t = dot3(fr, fr + 3);
  addscl3(fr + 3, fr, -t);
  nn = sqrt(dot3(fr + 3, fr + 3));
  fr[3] /= nn; fr[4] /= nn; fr[5] /= nn;
  cross3(fr + 6, fr, fr + 3);
}

// ------------------------------------------------------------------------------------
// Contact rows (stages 6, 7, 10): a condim-3 elliptic contact between a finger box
// (chain dofs, +) and an element capsule (one slider, -) or the static centre sphere.
// ------------------------------------------------------------------------------------
struct Contact {
  double Jf[3][SG_CD];  // rows: normal, tangent1, tangent2; columns: chain dofs
  double Js[3];         // slider column (0 when sl < 0)
  double A[6];          // symmetric 3x3 block J M^-1 J' + R: [00,01,02,11,12,22]
  double b[3], f[3];
  double R;             // regulariser (identical on the 3 rows: impratio 1, isotropic friction)
  double invm;          // 1/M of the slider (0 when sl < 0)
  int sl;               // element index or -1
};

// geometry -> rows.  nd = number of chain dofs that move the box's body.
SG_HD void contact_build(Contact& c, const ConRec& rec, const ChainKin& K, int nd, const double* Minv, const double* vc, const double* asm_c,
                         const double* warm_c, double binvw_box, int sl, const double* eaxis, double ve, double asm_e, double warm_e,
                         double invm_e, double binvw_e, const SgPlanHeader& H) {
  double fr[9];
  make_frame(rec.n, fr);
  c.sl = sl;
  c.invm = sl >= 0 ? invm_e : 0.0;
#pragma unroll
  for (int r = 0; r < 3; r++) {
#pragma unroll
    for (int d = 0; d < SG_CD; d++) {
      double jp[3] = {0, 0, 0};
      if (d < nd) chain_jacp(K, d, rec.pos, jp);
      c.Jf[r][d] = d < nd ? dot3(fr + 3 * r, jp) : 0.0;
    }
    c.Js[r] = sl >= 0 ? -dot3(fr + 3 * r, eaxis) : 0.0;
  }
  double dA = binvw_box + (sl >= 0 ? binvw_e : 0.0);
  double imp = impedance(H.con_solimp, rec.dist, H.con_margin);
  c.R = fmax(SG_MINVAL, (1 - imp) / imp * dA);
  double D = 1 / c.R, jar[3];
  double W[3][SG_CD];  // Jf * Minv
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int d = 0; d < SG_CD; d++) {
      double s = 0;
#pragma unroll
      for (int e = 0; e < SG_CD; e++) s += c.Jf[r][e] * Minv[4 * e + d];
      W[r][d] = s;
    }
  int k = 0;
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int s2 = r; s2 < 3; s2++) {
      double s = c.Js[r] * c.Js[s2] * c.invm + (r == s2 ? c.R : 0.0);
#pragma unroll
      for (int d = 0; d < SG_CD; d++) s += W[r][d] * c.Jf[s2][d];
      c.A[k++] = s;
    }
#pragma unroll
  for (int r = 0; r < 3; r++) {
    double vel = c.Js[r] * ve, js = c.Js[r] * asm_e, jw = c.Js[r] * warm_e;
#pragma unroll
    for (int d = 0; d < SG_CD; d++) { vel += c.Jf[r][d] * vc[d]; js += c.Jf[r][d] * asm_c[d]; jw += c.Jf[r][d] * warm_c[d]; }
    double aref = -H.con_B * vel - (r == 0 ? H.con_K * imp * (rec.dist - H.con_margin) : 0.0);
    c.b[r] = js - aref;
    jar[r] = jw - aref;
  }
  // warmstart force: primal -> dual map of the elliptic cone (mj_constraintUpdate)
  double mu = H.con_mu[0], U0 = jar[0] * mu, U1 = jar[1] * H.con_mu[0], U2 = jar[2] * H.con_mu[1];
  double N = U0, T = sqrt(U1 * U1 + U2 * U2);
  if (N >= mu * T || (T <= 0 && N >= 0)) { c.f[0] = c.f[1] = c.f[2] = 0; }
  else if (mu * N + T <= 0 || (T <= 0 && N < 0)) { for (int r = 0; r < 3; r++) c.f[r] = -D * jar[r]; }
  else {
    double Dm = D / (mu * mu * (1 + mu * mu)), NmT = N - mu * T;
    c.f[0] = -Dm * NmT * mu;
    c.f[1] = -c.f[0] / T * U1 * H.con_mu[0];
    c.f[2] = -c.f[0] / T * U2 * H.con_mu[1];
  }
}

SG_HD int qcqp2(double* res, const double* Ain, const double* bin, const double* dd, double r) {
  double b1 = bin[0] * dd[0], b2 = bin[1] * dd[1];
  double A11 = Ain[0] * dd[0] * dd[0], A22 = Ain[3] * dd[1] * dd[1], A12 = Ain[1] * dd[0] * dd[1];
  double la = 0, v1 = 0, v2 = 0;
  for (int it = 0; it < 20; it++) {
    double det = (A11 + la) * (A22 + la) - A12 * A12;
    if (det < 1e-10) { res[0] = res[1] = 0; return 0; }
    double di = sg_div(1.0, det), P11 = (A22 + la) * di, P22 = (A11 + la) * di, P12 = -A12 * di;
    v1 = -P11 * b1 - P12 * b2; v2 = -P12 * b1 - P22 * b2;
    double val = v1 * v1 + v2 * v2 - r * r;
    if (val < 1e-10) break;
    double deriv = -2 * (P11 * v1 * v1 + 2 * P12 * v1 * v2 + P22 * v2 * v2), delta = sg_div(-val, deriv);
    if (delta < 1e-10) break;
    la += delta;
  }
  res[0] = v1 * dd[0]; res[1] = v2 * dd[1];
  return la != 0;
}

// One Gauss-Seidel block update of an elliptic contact (mj_solPGS).  aF = current M^-1 J' f of the chain dofs, as_ = same for
// the slider.  Writes the force change df and returns the cost change (<= 0).
//
// Same operations as the textbook sequence (normal-or-ray update, friction QCQP with the normal fixed, cost check); the
// common case -- active normal force, friction inside the cone after one Newton evaluation -- runs without a data-dependent
// branch (selects), because on the GPU every branch drains the wave's pipeline; the uncommon cases share one fallback.
// the block update proper: A = the contact's 3 x 3 block [00,01,02,11,12,22], res = its residual (A + R) f + b at the current forces
// f (in / out); writes the force change df and returns the cost change (<= 0)
SG_HD double contact_block_update(const double* A, const double* res, double* f, const double* mu, double* df) {
  const double o0 = f[0], o1 = f[1], o2 = f[2];
  const double A00 = A[0], A01 = A[1], A02 = A[2], A11 = A[3], A12 = A[4], A22 = A[5];
  // ---- normal (f0 ~ 0) or ray update ----
  double g0, g1, g2;
  {
    double w0 = A00 * o0 + A01 * o1 + A02 * o2, w1 = A01 * o0 + A11 * o1 + A12 * o2, w2 = A02 * o0 + A12 * o1 + A22 * o2;
    double denom = o0 * w0 + o1 * w1 + o2 * w2;
    double x = denom >= SG_MINVAL ? sg_div(-(o0 * res[0] + o1 * res[1] + o2 * res[2]), denom) : 0.0;
    x = (o0 + x * o0 < 0) ? -1.0 : x;
    g0 = o0 + x * o0; g1 = o1 + x * o1; g2 = o2 + x * o2;
  }
  if (o0 < SG_MINVAL) {  // uncommon: no normal force yet
    g0 = o0 - sg_div(res[0], A00);
    g0 = g0 < 0 ? 0.0 : g0;
    g1 = g2 = 0;
  }
  // ---- friction update with the normal force fixed: min 1/2 v'Ac v + v'bc  s.t. |v/mu| <= g0 (mju_QCQP2) ----
  double v0 = 0, v1 = 0;
  {
    const double bc0 = res[1] - (A11 * o1 + A12 * o2) + A01 * (g0 - o0), bc1 = res[2] - (A12 * o1 + A22 * o2) + A02 * (g0 - o0);
    const double b1 = bc0 * mu[0], b2 = bc1 * mu[1];
    const double S11 = A11 * mu[0] * mu[0], S22 = A22 * mu[1] * mu[1], S12 = A12 * mu[0] * mu[1];
    const double r = g0;
    double la = 0, u1 = 0, u2 = 0;
    bool zero = false, more;
    {  // Newton evaluation at la = 0
      double det = S11 * S22 - S12 * S12;
      zero = det < 1e-10;
      double di = sg_div(1.0, zero ? 1.0 : det), P11 = S22 * di, P22 = S11 * di, P12 = -S12 * di;
      u1 = -P11 * b1 - P12 * b2; u2 = -P12 * b1 - P22 * b2;
      double val = u1 * u1 + u2 * u2 - r * r;
      double deriv = -2 * (P11 * u1 * u1 + 2 * P12 * u1 * u2 + P22 * u2 * u2);
      more = !zero && !(val < 1e-10) && !(g0 < SG_MINVAL);
      double delta = more ? sg_div(-val, deriv) : 0.0;
      more = more && !(delta < 1e-10);
      la = more ? delta : 0.0;
    }
#ifdef SGT_X_NOQCQP   // (timing experiment only: build_native.py --ko noqcqp -DSGT_X_NOQCQP -DSG_SECTION_PROF)
    more = false;
#endif
    if (more) {  // uncommon: the unconstrained minimiser leaves the cone -- continue Newton on the multiplier
      for (int it = 1; it < 20; it++) {
        double det = (S11 + la) * (S22 + la) - S12 * S12;
        if (det < 1e-10) { zero = true; break; }
        double di = sg_div(1.0, det), P11 = (S22 + la) * di, P22 = (S11 + la) * di, P12 = -S12 * di;
        u1 = -P11 * b1 - P12 * b2; u2 = -P12 * b1 - P22 * b2;
        double val = u1 * u1 + u2 * u2 - r * r;
        if (val < 1e-10) break;
        double deriv = -2 * (P11 * u1 * u1 + 2 * P12 * u1 * u2 + P22 * u2 * u2), delta = sg_div(-val, deriv);
        if (delta < 1e-10) break;
        la += delta;
      }
    }
    v0 = zero ? 0.0 : u1 * mu[0];
    v1 = zero ? 0.0 : u2 * mu[1];
    if (!zero && la != 0) {  // on the cone: put v exactly on the ellipse
      double s = v0 * v0 / (mu[0] * mu[0]) + v1 * v1 / (mu[1] * mu[1]);
      s = sqrt(g0 * g0 / fmax(SG_MINVAL, s));
      v0 *= s; v1 *= s;
    }
  }
  const bool nofric = g0 < SG_MINVAL;
  g1 = nofric ? 0.0 : v0;
  g2 = nofric ? 0.0 : v1;
  double d0 = g0 - o0, d1 = g1 - o1, d2 = g2 - o2;
  double change = 0.5 * (d0 * (A00 * d0 + A01 * d1 + A02 * d2) + d1 * (A01 * d0 + A11 * d1 + A12 * d2) + d2 * (A02 * d0 + A12 * d1 + A22 * d2)) +
                  d0 * res[0] + d1 * res[1] + d2 * res[2];
  const bool reject = change > 1e-10;  // the update must not increase the cost
  f[0] = reject ? o0 : g0; f[1] = reject ? o1 : g1; f[2] = reject ? o2 : g2;
  df[0] = reject ? 0.0 : d0; df[1] = reject ? 0.0 : d1; df[2] = reject ? 0.0 : d2;
  return reject ? 0.0 : change;
}
// ---- the same block update with the contact's constants PRECOMPUTED (tree pipeline, r05; the rows pipeline's solver has worked this way
// since r03, sg_rows.hip): Pe = {P11, P12, P22, e1, e2, cs, sn} -- the inverse of the friction-scaled 2 x 2 block S (0 when it is
// singular) and S's eigen-decomposition S = Q diag(e1, e2) Q', Q = [[cs, sn], [-sn, cs]], built once per contact and substep
// (contact_block_constants).  Same iterates, same stopping rules and the same results as contact_block_update to round-off (mju_QCQP2's
// Newton iteration in S's eigen-coordinates: DESIGN.md 4.5) -- but one division on the common path instead of three and no square root,
// which is what a wavefront alone on its SIMD pays for: every instruction of this dependent chain costs its full latency.
SG_HD void contact_block_constants(const double* A, const double* mu, double* Pe) {
  const double S11 = A[3] * mu[0] * mu[0], S22 = A[5] * mu[1] * mu[1], S12 = A[4] * mu[0] * mu[1];
  const double det = S11 * S22 - S12 * S12, di = det < 1e-10 ? 0.0 : 1.0 / det;
  Pe[0] = S22 * di; Pe[1] = -S12 * di; Pe[2] = S11 * di;
  double ecs = 1.0, esn = 0.0, ee1 = S11, ee2 = S22;
  if (fabs(S12) > 1e-300) {   // one Jacobi rotation
    const double tau = (S22 - S11) / (2.0 * S12), tt = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
    ecs = 1.0 / sqrt(1.0 + tt * tt); esn = tt * ecs;
    ee1 = S11 - tt * S12; ee2 = S22 + tt * S12;
  }
  Pe[3] = ee1; Pe[4] = ee2; Pe[5] = ecs; Pe[6] = esn;
}
SG_HD double sg_rsqrt(double q) {   // 1 / sqrt(q), q > 0
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rsq(q);
  y = y * (1.5 - 0.5 * q * y * y);
  return y * (1.5 - 0.5 * q * y * y);
#else
  return 1.0 / sqrt(q);
#endif
}
// mju_QCQP2's Newton iteration on the multiplier, evaluations 1 .. 19, in the eigen-coordinates of the friction block (contact_block_update_pre
// below) -- the loop BY HAND on the device: the rows pipeline's (sg_rows.hip update_row, where it is written out in place: moved behind this
// function the compiler fused the products after it differently and the box scene lost 1.4 %), for the tree pipeline's block update (r05:
// four-finger scene +0.5 %).  la: in / out; x1, x2: e_k + la of the last evaluation.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SG_SECTION_COUNT)
#define SG_NEWTON_BY_HAND 1
__device__ __forceinline__ void qcqp_newton_by_hand(const double e1, const double e2, const double C1h, const double C2h, const double R2h, double& la, double& x1, double& x2) {
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
}
#endif
SG_HD double contact_block_update_pre(const double* A, const double* Pe, const double* res, double* f, const double* mu, double* df) {
  const double o0 = f[0], o1 = f[1], o2 = f[2];
  const double A00 = A[0], A01 = A[1], A02 = A[2], A11 = A[3], A12 = A[4], A22 = A[5];
  const double P11 = Pe[0], P12 = Pe[1], P22 = Pe[2];
  const double w0 = A00 * o0 + A01 * o1 + A02 * o2, w1 = A01 * o0 + A11 * o1 + A12 * o2, w2 = A02 * o0 + A12 * o1 + A22 * o2;   // A f
  // ---- normal (f0 ~ 0) or ray update ----
  double g0;
  {
    const double denom = o0 * w0 + o1 * w1 + o2 * w2, num = o0 * res[0] + o1 * res[1] + o2 * res[2];
    double x = denom >= SG_MINVAL ? sg_div(-num, denom) : 0.0;
    x = (o0 + x * o0 < 0) ? -1.0 : x;
    g0 = o0 + x * o0;
  }
  if (o0 < SG_MINVAL) {  // uncommon: no normal force yet
    g0 = o0 - sg_div(res[0], A00);
    g0 = g0 < 0 ? 0.0 : g0;
  }
  // ---- friction rows with the normal force fixed: min 1/2 v'Ac v + v'bc  s.t. |v/mu| <= g0 (mju_QCQP2) ----
  const bool nofric = g0 < SG_MINVAL;
  const double b1 = nofric ? 0.0 : ((res[1] - w1) + A01 * g0) * mu[0], b2 = nofric ? 0.0 : ((res[2] - w2) + A02 * g0) * mu[1];
  const double u1 = -(P11 * b1 + P12 * b2), u2 = -(P12 * b1 + P22 * b2);   // 0 when the friction block is singular
  const double val = (u1 * u1 + u2 * u2) - g0 * g0;
  double v1 = u1 * mu[0], v2 = u2 * mu[1];
  if (!(val < 1e-10) && !nofric) {  // uncommon: outside the cone -- Newton on the multiplier, in S's eigen-coordinates
    const double e1 = Pe[3], e2 = Pe[4], qcs = Pe[5], qsn = Pe[6];
    const double c1 = qcs * b1 - qsn * b2, c2 = qsn * b1 + qcs * b2;
    const double C1h = 0.5 * c1 * c1, C2h = 0.5 * c2 * c2, R2h = 0.5 * g0 * g0;
    double la = 0.0;
    bool run;
    {
      const double deriv = -2.0 * (P11 * u1 * u1 + 2.0 * P12 * u1 * u2 + P22 * u2 * u2), delta = sg_div(-val, deriv);
      run = !(delta < 1e-10);
      la = run ? delta : 0.0;
    }
    const bool ever = run;
    double x1 = e1, x2 = e2;
    if (run) {
#ifdef SG_NEWTON_BY_HAND
      qcqp_newton_by_hand(e1, e2, C1h, C2h, R2h, la, x1, x2);
#else
      for (int it = 1; it < 20; it++) {
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
    const double t1e = -c1 * x2, t2e = -c2 * x1;   // last evaluation, back in the contact's coordinates: v = Q (-c1 / x1, -c2 / x2) = (t1, t2) / det
    const double t1 = qcs * t1e + qsn * t2e, t2 = qcs * t2e - qsn * t1e;
    const bool sing = ever && det < 1e-10;
    const double q1 = ever ? t1 : u1, q2 = ever ? t2 : u2, qdet = ever ? det : 1.0;
    const bool active = la != 0.0 && !sing;
    // v = q / qdet, rescaled onto the cone when the constraint is active: |(v1 / mu0, v2 / mu1)| = g0 (the division by qdet cancels)
    const double qq = fmax(SG_MINVAL, active ? q1 * q1 + q2 * q2 : qdet * qdet);
    const double y = sg_rsqrt(qq);
    const double sc = sing ? 0.0 : (active ? g0 * y : y);
    v1 = q1 * mu[0] * sc; v2 = q2 * mu[1] * sc;
  }
  const double g1 = v1, g2 = v2;
  const double n0 = A00 * g0 + A01 * g1 + A02 * g2, n1 = A01 * g0 + A11 * g1 + A12 * g2, n2 = A02 * g0 + A12 * g1 + A22 * g2;   // A f_new
  const double d0 = g0 - o0, d1 = g1 - o1, d2 = g2 - o2;
  const double change = d0 * (0.5 * (n0 - w0) + res[0]) + d1 * (0.5 * (n1 - w1) + res[1]) + d2 * (0.5 * (n2 - w2) + res[2]);
  const bool reject = change > 1e-10;  // the update must not increase the cost
  f[0] = reject ? o0 : g0; f[1] = reject ? o1 : g1; f[2] = reject ? o2 : g2;
  df[0] = reject ? 0.0 : d0; df[1] = reject ? 0.0 : d1; df[2] = reject ? 0.0 : d2;
  return reject ? 0.0 : change;
}
SG_HD double contact_update(Contact& c, const double* aF, double as_, const double* mu, double* df) {
  double res[3];
#pragma unroll
  for (int r = 0; r < 3; r++)  // two independent partial sums: half the dependency depth of a serial accumulation
    res[r] = ((c.b[r] + c.Js[r] * as_) + (c.Jf[r][0] * aF[0] + c.Jf[r][1] * aF[1])) + ((c.R * c.f[r] + c.Jf[r][2] * aF[2]) + c.Jf[r][3] * aF[3]);
  return contact_block_update(c.A, res, c.f, mu, df);
}

// scalar row update (equality: free, limit: f >= 0); returns cost change, writes new force
SG_HD double scalar_update(double& f, double b, double Ja, double R, double Adiag, bool inequality) {
  double res = b + Ja + R * f, old = f, fn = f - sg_div(res, Adiag);
  if (inequality && fn < 0) fn = 0;
  double d = fn - old, change = 0.5 * d * d * Adiag + d * res;
  if (change > 1e-10) { fn = old; change = 0; }
  f = fn;
  return change;
}

// joint limit rows of one chain (stage 6): slot 2*d + s (s = 0 lower side, 1 upper side), i.e. MuJoCo's row order
struct LimitRows {
  int active;  // bit mask over the SG_MAXLIM slots
  double sign[SG_MAXLIM], R[SG_MAXLIM], b[SG_MAXLIM], f[SG_MAXLIM];
};
SG_HD void limits_build(const SgChain& C, const double* q, const double* v, const double* asm_c, const double* warm_c, LimitRows& L) {
  L.active = 0;
#pragma unroll
  for (int d = 0; d < SG_CD; d++) {
#pragma unroll
    for (int sd = 0; sd < 2; sd++) {
      const int k = 2 * d + sd, side = 2 * sd - 1;
      L.sign[k] = -side; L.R[k] = 1; L.b[k] = 0; L.f[k] = 0;
      if (!C.limited[d]) continue;
      double dist = side * (C.range[d][sd] - q[d]);
      const bool on = dist < C.jmargin[d];
      if (sg_any(on) && on) {  // a joint sits at one of its limits only now and then: the ~150-instruction row is skipped as a whole
        double sg = -side, imp = impedance(C.lim_solimp[d], dist, C.jmargin[d]);
        double R = fmax(SG_MINVAL, (1 - imp) / imp * C.invw[d]);
        double aref = -C.lim_B[d] * sg * v[d] - C.lim_K[d] * imp * (dist - C.jmargin[d]);
        double jar = sg * warm_c[d] - aref;
        L.active |= 1 << k;
        L.R[k] = R; L.b[k] = sg * asm_c[d] - aref;
        L.f[k] = jar < 0 ? -jar / R : 0.0;
      }
    }
  }
}

}  // namespace sgm
