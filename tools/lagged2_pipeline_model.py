"""numpy model of the PIPELINED two-iterations-per-sweep Gram-Schmidt form (successor of
tools/lagged2_gs_model.py, whose single-pass short cuts failed): the one-sweep form of DESIGN.md 3.2 run for TWO iterations
between sweeps.

Lanczos vectors come in pairs (a_m, b_m).  Between two sweeps the operator is applied twice, each time to a RAW vector, and the
three-term recurrences are formed with raw vectors and estimated coefficients only:
    y1 = A (a s_a)         b = y1 - e1 (a s_a) - f1 (bp s_bp)          (bp = the previous pair's b, still alive as a raw vector)
    y2 = A (b s_b)         c = y2 - e2 (b s_b) - f2 (a s_a)
so every O(1) coefficient multiplies a raw vector, and the mutual inner products of the raw vectors are REAL dots (a small
kernel over four vectors).  ONE sweep over the stored basis S then
  * writes the two Lanczos vectors whose stored-basis coefficients were measured by the PREVIOUS sweep (u_bp, u_a: late updates
    with measured coefficients, like the one-sweep form),
  * measures S^T b and S^T c (and, strip by strip, the dots of b and c with the two vectors it is writing),
  * subtracts from c its PREDICTED stored-basis components (they follow from the measured coefficients of bp and a through the
    recorded tridiagonal before the sweep starts) - the next operator input carries only fresh rounding along span(S).
Stored-basis coefficients are eps-sized everywhere, so nothing is ever multiplied by the basis' own non-orthogonality.

    python tools/lagged2_pipeline_model.py          -> profiles/r04_lagged2_pipeline_model.txt
"""
import os

import numpy as np
import scipy.sparse as sp

n, K = 4000, 260
rng = np.random.default_rng(1)
A = sp.random(n, n, density=8 / n, random_state=3, format="csr")
A = ((A + A.T) * 0.5 + sp.diags(np.linspace(2, 12, n))).tocsr()
v0 = rng.uniform(-1, 1, n)
v0 /= np.linalg.norm(v0)


def reference():
    U = np.zeros((K + 1, n)); U[0] = v0; al = []; be = []
    for k in range(1, K + 1):
        w = A @ U[k - 1]; a = U[k - 1] @ w; w -= a * U[k - 1]
        if k > 1: w -= be[-1] * U[k - 2]
        w -= (U[:k] @ w) @ U[:k]
        b = np.linalg.norm(w); U[k] = w / b; al.append(a); be.append(b)
    return np.array(al), np.array(be), U


def pipelined(compensate=True, inject=0.0):
    S = np.zeros((K + 4, n))
    al, be = [], []                      # al[j] = <u_j, A u_j>, be[j] couples u_j and u_{j+1}
    # ---- start-up with two clean iterations: S = [u_0], pending raw bp (-> u_1) and a (-> u_2), coefficients measured
    S[0] = v0
    y = A @ S[0]; al.append(S[0] @ y); w = y - al[0] * S[0]
    w -= (S[:1] @ w) @ S[:1]
    be.append(np.linalg.norm(w)); u1 = w / be[0]
    y = A @ u1; al.append(u1 @ y); w2 = y - al[1] * u1 - be[0] * S[0]
    w2 -= (S[:1] @ w2) @ S[:1]; w2 -= (u1 @ w2) * u1
    P = 1
    bp, a = u1 * 1.0, w2                 # raw vectors
    g_bp = S[:P] @ bp                    # measured coefficients against the stored basis (real dots)
    g_a = S[:P] @ a
    rho_bp = np.sqrt(bp @ bp - g_bp @ g_bp)
    ga_bp = (bp @ a - g_bp @ g_a) / rho_bp          # <u_bp, a>
    rho_a = np.sqrt(a @ a - g_a @ g_a - ga_bp ** 2)
    be.append(rho_a)                     # be[1]: couples u_1 (= u_bp) and u_2 (= u_a)
    # al has alpha_0, alpha_1 (alpha of u_bp); be has be[0] (u_0 - u_bp), be[1] (u_bp - u_a)
    maxg = 0.0
    while len(al) + 2 <= K:
        nb = P + 1                       # index of u_bp in the Lanczos numbering; u_a = nb + 1
        s_bp, s_a = 1.0 / rho_bp, 1.0 / rho_a
        # ---- operator twice on raw vectors, three-term with raw vectors and estimated coefficients
        xa = a * s_a
        y1 = A @ xa
        e1 = xa @ y1                     # (fused dot of the operator kernel)
        f1 = rho_a                       # estimate of the coupling to the previous vector
        b = y1 - e1 * xa - f1 * (bp * s_bp)
        s_b = 1.0 / np.linalg.norm(b)    # estimate (raw norm; a dot kernel)
        xb = b * s_b
        y2 = A @ xb
        e2 = xb @ y2
        f2 = 1.0 / s_b
        c = y2 - e2 * xb - f2 * xa
        # ---- predictions of S^T b and S^T c through the recorded tridiagonal (before the sweep; eps-sized numbers)
        alh, beh = np.array(al[:P]), np.array(be[:P])     # beh[P-1] couples S[P-1] and u_bp

        def stencil(v, v_bp):
            out = alh * v
            if P > 1:
                out[1:] += beh[:P - 1] * v[:-1]
                out[:-1] += beh[:P - 1] * v[1:]
            out[P - 1] += beh[P - 1] * v_bp
            return out
        Sa, Sbp = g_a * s_a, g_bp * s_bp                  # S^T xa, S^T (bp s_bp)
        ubp_xa = ga_bp * s_a                              # <u_bp, xa>
        Sy1 = stencil(Sa, ubp_xa)
        Sb_pred = Sy1 - e1 * Sa - f1 * Sbp
        # <u_bp, b> through raw dots: u_bp = (bp - S g_bp) / rho_bp
        ubp_b_pred = ((bp @ b) - g_bp @ Sb_pred) / rho_bp
        Sy2 = stencil(Sb_pred * s_b, ubp_b_pred * s_b)
        Sc_pred = Sy2 - e2 * s_b * Sb_pred - f2 * Sa
        # ---- ONE sweep over the stored basis
        u_bp = (bp - g_bp @ S[:P]) / rho_bp               # late updates with MEASURED coefficients
        u_a = (a - g_a @ S[:P] - ga_bp * u_bp) / rho_a
        m_b = S[:P] @ b                                   # measurements
        m_c = S[:P] @ c
        if inject and P == 21:                            # plant a known perturbation along stored vectors in the next operator input
            pert = inject * np.linalg.norm(c) * (S[3] - S[7] + 0.5 * S[P - 1])
            c = c + pert; m_c = m_c + S[:P] @ pert
        if compensate:
            c = c - Sc_pred @ S[:P]
            m_c = m_c - Sc_pred                           # by linearity (S orthonormal to rounding; eps-sized numbers)
        S[P] = u_bp; S[P + 1] = u_a
        gb_new = np.concatenate([m_b, [u_bp @ b, u_a @ b]])   # in-strip dots with the two vectors just written
        gc_new = np.concatenate([m_c, [u_bp @ c, u_a @ c]])
        maxg = max(maxg, np.abs(gc_new).max())
        # ---- fold: recurrence coefficients from the operator kernels' fused dots and eps-sized corrections (no extra products):
        #   xa = u_a + eps_a, eps_a = (S g_a + ga_bp u_bp) / rho_a:   e1 = alpha_a + 2 <eps_a, A u_a> + <eps_a, A eps_a>,
        #   <eps_a, A u_a> = be(bp,a) <eps_a, u_bp> = ga_bp   (only u_bp couples to u_a inside the span of eps_a)
        def quad(cS, c_bp, c_a=None):
            """<E, A E> for E = S cS + c_bp u_bp (+ c_a u_a) through the recorded tridiagonal (second order in eps)"""
            Pold = len(cS)
            v = np.concatenate([cS, [c_bp]] + ([[c_a]] if c_a is not None else []))
            m = len(v)
            tv = np.array(al[:m]) * v
            tv[1:] += np.array(be[:m - 1]) * v[:-1]
            tv[:-1] += np.array(be[:m - 1]) * v[1:]
            return v @ tv
        alpha_a = e1 - 2.0 * ga_bp - quad(g_a / rho_a, ga_bp / rho_a)
        al.append(alpha_a)
        P += 2
        rho_b = np.sqrt(b @ b - gb_new @ gb_new)
        be.append(rho_b)                                  # couples u_a and u_b
        #   xb = (rho_b u_b + E_b) s_b, E_b = S_new g_b:  e2 / s_b^2 = rho_b^2 alpha_b + 2 rho_b^2 <u_a, b> + <E_b, A E_b>
        alpha_b = (e2 / s_b ** 2 - 2.0 * rho_b * rho_b * gb_new[-1] - quad(gb_new[:-2], gb_new[-2], gb_new[-1])) / rho_b ** 2
        al.append(alpha_b)
        # next pair
        bp, a = b, c
        g_bp, g_a = gb_new, gc_new
        rho_bp = rho_b
        ga_bp = (bp @ a - g_bp @ g_a) / rho_bp
        rho_a = np.sqrt(a @ a - g_a @ g_a - ga_bp ** 2)
        be.append(rho_a)
    m = min(len(al), K)
    return np.array(al[:m]), np.array(be[:m]), S[:P], maxg


def main():
    ra, rb, RU = reference()
    lines = ["model: n = %d random symmetric + diagonal 2..12, %d iterations; reference = full re-orthogonalisation" % (n, K)]
    for label, comp, inj in (("pipelined pairs, predicted compensation of the next operator input inside the sweep", True, 0.0),
                             ("the same, stored-basis components of relative size 1e-6 planted in one operator input", True, 1e-6),
                             ("the same, relative size 1e-3 planted", True, 1e-3),
                             ("pipelined pairs, NO compensation", False, 0.0)):
        try:
            a, b, S, maxg = pipelined(comp, inj)
        except FloatingPointError as e:
            lines.append("%s: %s" % (label, e)); continue
        m = min(len(a), len(ra)); mb = m - 1
        with np.errstate(invalid="ignore"):
            lines += ["%s: %d iterations compared" % (label, m),
                      "  max|dalpha| %.2e   max|dbeta| %.2e   (tolerance 1e-10 * ||A|| = %.1e)" % (
                          np.nanmax(np.abs(a[:m] - ra[:m])), np.nanmax(np.abs(b[:mb] - rb[:mb])), 1e-10 * 12),
                      "  orthogonality of the stored basis max|S^T S - I| = %.2e ; largest stored-basis coefficient of an operator input %.1e" % (
                          np.abs(S @ S.T - np.eye(len(S))).max(), maxg)]
        for k in (10, 100, m - 1):
            lines.append("  k = %3d: dalpha %.1e dbeta %.1e" % (k, abs(a[k] - ra[k]), abs(b[min(k, mb - 1)] - rb[min(k, mb - 1)])))
    text = "\n".join(lines) + "\n"
    print(text)
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r04_lagged2_pipeline_model.txt"), "w") as f:
        f.write(text)


if __name__ == "__main__":
    np.seterr(invalid="raise")
    main()
