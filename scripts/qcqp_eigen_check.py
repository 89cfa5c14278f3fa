"""mju_QCQP2's Newton iteration as the oracle writes it (inverse of the 2 x 2 block S + la per evaluation) against the form the solver
kernel runs since r03 (eigen-coordinates of S, csrc/sg_rows.hip): same evaluation counts, same results?  NumPy only.
usage: python scripts/qcqp_eigen_check.py > profiles/r03_qcqp_eigen_check.txt"""
import numpy as np


def reference(S11, S12, S22, b1, b2, r):
    """oracle/sg_oracle.c qcqp2 (friction-scaled quantities)"""
    la, its, v1, v2 = 0.0, 0, 0.0, 0.0
    for _ in range(20):
        det = (S11 + la) * (S22 + la) - S12 * S12
        if det < 1e-10:
            return 0.0, 0.0, 0.0, its
        di = 1 / det
        P11, P22, P12 = (S22 + la) * di, (S11 + la) * di, -S12 * di
        v1, v2 = -P11 * b1 - P12 * b2, -P12 * b1 - P22 * b2
        val = v1 * v1 + v2 * v2 - r * r
        its += 1
        if val < 1e-10:
            break
        delta = -val / (-2 * (P11 * v1 * v1 + 2 * P12 * v1 * v2 + P22 * v2 * v2))
        if delta < 1e-10:
            break
        la += delta
    return v1, v2, la, its


def eigen(S11, S12, S22, b1, b2, r):
    """the kernel's form: evaluation 0 with the inverse block (the fast path), the rest in the eigen-coordinates the phase kernel exports"""
    if abs(S12) > 1e-300:
        tau = (S22 - S11) / (2 * S12)
        t = (1.0 if tau >= 0 else -1.0) / (abs(tau) + np.sqrt(1 + tau * tau))
        cs = 1 / np.sqrt(1 + t * t); sn = t * cs; e1 = S11 - t * S12; e2 = S22 + t * S12
    else:
        cs, sn, e1, e2 = 1.0, 0.0, S11, S22
    det = S11 * S22 - S12 * S12
    di = 1 / det
    P11, P22, P12 = S22 * di, S11 * di, -S12 * di
    u1, u2 = -(P11 * b1 + P12 * b2), -(P12 * b1 + P22 * b2)
    val = u1 * u1 + u2 * u2 - r * r
    if val < 1e-10:
        return u1, u2, 0.0, 1
    delta = -val / (-2 * (P11 * u1 * u1 + 2 * P12 * u1 * u2 + P22 * u2 * u2))
    if delta < 1e-10:
        return u1, u2, 0.0, 1
    la, its = delta, 1
    c1, c2 = cs * b1 - sn * b2, sn * b1 + cs * b2
    C1h, C2h, R2h = .5 * c1 * c1, .5 * c2 * c2, .5 * r * r
    x1, x2 = e1, e2
    for _ in range(1, 20):
        x1, x2 = e1 + la, e2 + la
        y1, y2 = x1 * x1, x2 * x2
        ah, bh, yy = C1h * y2, C2h * y1, y1 * y2
        Nh, Dh, xx = -R2h * yy + (ah + bh), ah * x2 + bh * x1, x1 * x2
        delta = Nh * xx / (2 * Dh)
        its += 1
        if xx < 1e-10 or Nh < 0.5e-10 * yy or delta < 1e-10:
            break
        la += delta
    det = x1 * x2
    if det < 1e-10:
        return 0.0, 0.0, 0.0, its
    t1e, t2e = -c1 * x2, -c2 * x1
    return (cs * t1e + sn * t2e) / det, (cs * t2e - sn * t1e) / det, la, its


def main():
    rng = np.random.RandomState(1)
    worst, n, hist = 0.0, 0, {}
    for _ in range(100000):
        a = rng.randn(2, 2)
        S = a @ a.T * 10 ** rng.uniform(-1, 1) + np.eye(2) * 10 ** rng.uniform(-3, 0)
        b = rng.randn(2) * 10 ** rng.uniform(-2, 2)
        r = 10 ** rng.uniform(-2, 1)
        v = reference(S[0, 0], S[0, 1], S[1, 1], b[0], b[1], r)
        if v[3] == 0:
            continue          # singular at the first evaluation: the kernel's fast path returns zero friction there too
        w = eigen(S[0, 0], S[0, 1], S[1, 1], b[0], b[1], r)
        assert v[3] == w[3], (v, w)
        worst = max(worst, max(abs(v[0] - w[0]), abs(v[1] - w[1])) / max(abs(v[0]), abs(v[1]), 1e-300))
        hist[v[3]] = hist.get(v[3], 0) + 1
        n += 1
    print("%d random friction blocks (S positive definite over 5 decades, b over 4, cone radius over 3)" % n)
    print("evaluation counts identical in every case; histogram:", dict(sorted(hist.items())))
    print("largest relative difference of the friction force: %.2e" % worst)


if __name__ == "__main__":
    main()
