"""Executable specification (numpy) of the TWO-ITERATIONS-PER-SWEEP Gram-Schmidt form of the Lanczos loop, written kernel by
kernel the way csrc/kernels.hip and csrc/engine.cpp implement it (pair_* kernels, LoopState::enqueue_pair).  Successor of
tools/lagged2_pipeline_model.py (round 4): same algebra, restructured into the launches of the device code, real AND complex
Hermitian operators, entry from the one-sweep ("lagged") state, exit (flush) to a complete basis, and a self-check of every
folded quantity against the directly computed truth.

State between sweeps (P stored, complete, orthonormal vectors S = u_0 .. u_{P-1}; T recorded up to alpha_{P-1}, beta_{P-1}):
    r1 -> u_P      raw, with measured g1 = S^H r1, rho1 = sqrt(|r1|^2 - |g1|^2)
    r2 -> u_{P+1}  raw, with measured g2 = S^H r2, gam = <u_P, r2> = (<r1, r2> - g1^H g2) / rho1,
                   rho2 = sqrt(|r2|^2 - |g2|^2 - |gam|^2)
One pair = two operator applications on RAW vectors, one three-term kernel over raw vectors (the second three-term update is
formed inside the sweep, from strips it reads anyway), one small predict kernel, ONE sweep over S and one fold:
    y1 = A x2,  x2 = r2 / rho2                 e1 = <x2, y1>            (operator kernel, fused dot)
    r3 = y1 - e1 x2 - rho2 x1,  x1 = r1 / rho1                          (pair_three_term; also |r3|^2)
    y2 = A x3,  x3 = r3 / n3,  n3 = |r3|       e2 = <x3, y2>
    r4 = y2 - e2 x3 - n3 x2                                             (inside the sweep)
    predict S^H r3, S^H r4 through the recorded tridiagonal                (pair_predict: eps-sized numbers)
    sweep: u_P = (r1 - S g1) / rho1, u_{P+1} = (r2 - S g2 - gam u_P) / rho2   -> stored
           m3 = S^H r3, m4 = S^H r4 (raw), r4 -= S pred4 (compensation of the NEXT operator input)
           in-strip: <u_P, r3>, <u_{P+1}, r3>, <u_P, r4>, <u_{P+1}, r4>, |r3|^2, <r3, r4>, |r4|^2
    fold:  alpha_{P+1}, beta_{P+1} = rho3, alpha_{P+2}, beta_{P+2} = rho4, and the next pair's (g1, rho1, g2, gam, rho2)

    python tools/pair_gs_model.py        -> profiles/r05_pair_gs_model.txt
"""
import os
import sys

import numpy as np
import scipy.sparse as sp


def make_problem(n, complex_, seed=3):
    rng = np.random.default_rng(seed)
    A = sp.random(n, n, density=8 / n, random_state=seed, format="csr")
    if complex_:
        B = sp.random(n, n, density=8 / n, random_state=seed + 1, format="csr")
        A = A + 1j * B
    A = ((A + A.conj().T) * 0.5 + sp.diags(np.linspace(2, 12, n))).tocsr()
    v0 = rng.uniform(-1, 1, n) + (1j * rng.uniform(-1, 1, n) if complex_ else 0)
    return A, v0 / np.linalg.norm(v0)


def reference(A, v0, K, Z=None):
    """Full re-orthogonalisation, sequential (the reference's loop, LL:216-322; Z: locked eigenvectors, LL:259)."""
    n = v0.shape[0]
    U = np.zeros((K + 1, n), dtype=v0.dtype)
    U[0] = v0
    al, be = [], []
    for k in range(1, K + 1):
        w = A @ U[k - 1]
        a = np.vdot(U[k - 1], w).real
        w = w - a * U[k - 1]
        if k > 1:
            w = w - be[-1] * U[k - 2]
        for z in (Z if Z is not None else ()):
            w = w - np.vdot(z, w) * z
        for j in range(k):
            w = w - np.vdot(U[j], w) * U[j]
        b = np.linalg.norm(w)
        U[k] = w / b
        al.append(a)
        be.append(b)
    return np.array(al), np.array(be), U


def tri_apply(al, be, v, extra=None, lam=()):
    """The image of E = sum v_col (stored column col) under the operator, in the same columns: lam_i v_i for the len(lam) locked
    EIGENvectors in front (A z_i = lam_i z_i up to their residual), (T v) through the recorded tridiagonal for the Lanczos vectors
    behind them; extra = (coefficient on the vector BEHIND the last one, its coupling beta) adds that neighbour's contribution to
    the last row."""
    L = len(lam)
    out = np.empty_like(v)
    out[:L] = np.asarray(lam) * v[:L]
    w = v[L:]
    m = len(w)
    t = np.asarray(al[:m]) * w
    if m > 1:
        t[1:] += np.asarray(be[:m - 1]) * w[:-1]
        t[:-1] += np.asarray(be[:m - 1]) * w[1:]
    if extra is not None:
        t[m - 1] += extra[1] * extra[0]
    out[L:] = t
    return out


def quad(al, be, v, lam=()):
    """Re <E, A E> for E = sum v_col (stored column col): locked eigenvectors, then the first Lanczos vectors (through T: second
    order in eps)."""
    return np.vdot(v, tri_apply(al, be, v, lam=lam)).real


class PairLoop:
    """The device loop.  Vectors are numpy arrays here; every method below is one kernel launch (or one fold) of the device code."""

    def __init__(self, A, v0, K, plant=0.0, Z=None, lam=()):
        """Z (L x n), lam: locked eigenpairs of a restart pass (LL:233,259): the start vector is orthogonal to them, every Lanczos
        vector is kept orthogonal to them; they are the first L stored columns of every sweep."""
        self.A, self.K, self.plant = A, K, plant
        n = v0.shape[0]
        self.L = 0 if Z is None else len(Z)
        self.lam = tuple(lam)
        self.S = np.zeros((self.L + K + 6, n), dtype=v0.dtype)
        if self.L:
            self.S[:self.L] = Z
        self.S[self.L] = v0
        self.P = 1
        self.al, self.be = [], []
        self.maxcoef = 0.0
        self.checks = []

    # ---- entry: two clean iterations (device: the clean / one-sweep iterations that precede the first pair leave the same state)
    def start(self):
        A, S, L = self.A, self.S, self.L
        K1 = L + 1                                   # stored columns: the locked vectors and u_0
        y = A @ S[L]
        a0 = np.vdot(S[L], y).real
        w = y - a0 * S[L]
        w = w - (S[:K1].conj() @ w) @ S[:K1]
        b0 = np.linalg.norm(w)
        u1 = w / b0
        self.al.append(a0)
        self.be.append(b0)
        # pending raw pair: r1 = u1 (already complete: g1 = 0 up to rounding, rho1 = 1), r2 = raw w of iteration 2
        y = A @ u1
        e = np.vdot(u1, y).real
        r2 = y - e * u1 - b0 * S[L]
        self.r1, self.r2 = u1.copy(), r2
        self.g1 = S[:K1].conj() @ self.r1
        self.g2 = S[:K1].conj() @ self.r2
        self.rho1 = np.sqrt(np.vdot(self.r1, self.r1).real - np.vdot(self.g1, self.g1).real)
        self.gam = (np.vdot(self.r1, self.r2) - np.vdot(self.g1, self.g2)) / self.rho1
        self.rho2 = np.sqrt(np.vdot(self.r2, self.r2).real - np.vdot(self.g2, self.g2).real - abs(self.gam) ** 2)
        # alpha of u_1: e = <u1, A u1> exactly (u1 complete)
        self.al.append(e)
        self.be.append(self.rho2)
        # now: P = 1 stored Lanczos vector; al = [alpha_0, alpha_1], be = [beta_0 (u0-u1), beta_1 = rho2 (u1-u2)]

    def pair(self):
        A, S, P, L, lam = self.A, self.S, self.P, self.L, self.lam
        Kc = L + P                                   # stored columns of this sweep
        al, be = self.al, self.be
        r1, r2, g1, g2, rho1, rho2, gam = self.r1, self.r2, self.g1, self.g2, self.rho1, self.rho2, self.gam
        # ---- operator 1 (input r2 / rho2, fused dot) + three-term 1
        x2 = r2 / rho2
        y1 = A @ x2
        e1 = np.vdot(x2, y1).real
        x1 = r1 / rho1
        r3 = y1 - e1 * x2 - rho2 * x1
        n3sq = np.vdot(r3, r3).real                 # (partial sums of the three-term kernel)
        d13 = np.vdot(r1, r3)                       # <r1, r3>: the same kernel reads r1 anyway
        n3 = np.sqrt(n3sq)
        # ---- operator 2 (input r3 / n3); its three-term update is formed inside the sweep on the device
        x3 = r3 / n3
        y2 = A @ x3
        e2 = np.vdot(x3, y2).real
        r4 = y2 - e2 * x3 - n3 * x2
        # ---- predict (small kernel): stored-column components of r3 and r4 from those of r1, r2 through T (and lam)
        c1, c2 = g1 / rho1, g2 / rho2               # S^H x1, S^H x2
        uP_x2 = gam / rho2                          # <u_P, x2>
        Sy1 = tri_apply(al, be, c2, extra=(uP_x2, be[P - 1]), lam=lam)   # S^H A x2 (only S and u_P reach back into S)
        p3 = Sy1 - e1 * c2 - rho2 * c1              # predicted S^H r3
        uP_r3 = (d13 - np.vdot(g1, p3)) / rho1      # predicted <u_P, r3>
        Sy2 = tri_apply(al, be, p3 / n3, extra=(uP_r3 / n3, be[P - 1]), lam=lam)
        p4 = Sy2 - e2 * p3 / n3 - n3 * c2           # predicted S^H r4
        # ---- ONE sweep over S
        uP = (r1 - g1 @ S[:Kc]) / rho1
        uQ = (r2 - g2 @ S[:Kc] - gam * uP) / rho2
        m3 = S[:Kc].conj() @ r3
        m4 = S[:Kc].conj() @ r4
        if self.plant and P == 21:                  # test: a known perturbation along stored vectors in the next operator input
            pert = self.plant * np.linalg.norm(r4) * (S[L + 3] - S[L + 7] + 0.5 * S[Kc - 1])
            r4 = r4 + pert
            m4 = m4 + S[:Kc].conj() @ pert
        r4 = r4 - p4 @ S[:Kc]                       # compensation: the next operator input carries fresh rounding only
        S[Kc], S[Kc + 1] = uP, uQ
        tail3 = np.array([np.vdot(uP, r3), np.vdot(uQ, r3)])
        tail4 = np.array([np.vdot(uP, r4), np.vdot(uQ, r4)])
        n4sq = np.vdot(r4, r4).real
        d34 = np.vdot(r3, r4)
        # ---- fold
        g3 = np.concatenate([m3, tail3])
        g4 = np.concatenate([m4 - p4, tail4])       # by linearity
        self.maxcoef = max(self.maxcoef, np.abs(g4).max() / np.sqrt(n4sq), np.abs(g3).max() / n3)
        #   alpha of u_{P+1}: e1 = alpha + 2 Re <eps, A u_{P+1}> + <eps, A eps>, eps = (S g2 + gam u_P) / rho2
        v = np.concatenate([c2, [uP_x2]])
        alpha_q = e1 - 2.0 * gam.real - quad(al, be, v, lam)
        al.append(alpha_q)
        rho3 = np.sqrt(n3sq - np.vdot(g3, g3).real)
        be.append(rho3)                             # couples u_{P+1} and u_{P+2}
        #   alpha of u_{P+2}: <r3, A r3> = rho3^2 alpha + 2 rho3^2 Re <u_{P+1}, r3> + <E, A E>, E = S_new g3
        alpha_n = (e2 * n3sq - 2.0 * rho3 * rho3 * g3[-1].real - quad(al, be, g3, lam)) / (rho3 * rho3)
        al.append(alpha_n)
        gam_n = (d34 - np.vdot(g3, g4)) / rho3
        rho4 = np.sqrt(n4sq - np.vdot(g4, g4).real - abs(gam_n) ** 2)
        be.append(rho4)
        # ---- self-check of the folded quantities against the directly computed truth (model only)
        Kn = Kc + 2
        proj = lambda w: w - (S[:Kn].conj() @ w) @ S[:Kn]
        t3 = proj(r3)
        self.checks.append((abs(np.linalg.norm(t3) - rho3), abs(np.vdot(t3 / np.linalg.norm(t3), r4) - gam_n)))
        self.P = P + 2
        self.r1, self.r2, self.g1, self.g2, self.rho1, self.rho2, self.gam = r3, r4, g3, g4, rho3, rho4, gam_n

    def flush(self):
        """Leave the pair form: complete the two pending vectors with their measured coefficients (device: two multi-axpys)."""
        S, P, Kc = self.S, self.P, self.L + self.P
        uP = (self.r1 - self.g1 @ S[:Kc]) / self.rho1
        uQ = (self.r2 - self.g2 @ S[:Kc] - self.gam * uP) / self.rho2
        S[Kc], S[Kc + 1] = uP, uQ
        self.P = P + 2


def measure(complex_, plant, n=4000, K=260, locked=0):
    """Run the pair loop and the reference on the same problem; the numbers the statements of DESIGN.md 3.2 rest on.
    locked: that many converged eigenpairs (the largest) are locked like in a restart pass."""
    A, v0 = make_problem(n, complex_)
    Z, lam = None, ()
    if locked:
        w, V = np.linalg.eigh(A.toarray())
        Z = np.ascontiguousarray(V[:, -locked:].T).astype(v0.dtype)
        lam = tuple(w[-locked:])
        v0 = v0 - (Z.conj() @ v0) @ Z
        v0 = v0 / np.linalg.norm(v0)
    ra, rb, RU = reference(A, v0, K, Z)
    L = PairLoop(A, v0, K, plant, Z, lam)
    L.start()
    while len(L.al) + 2 <= K:
        L.pair()
    L.flush()
    a, b = np.array(L.al), np.array(L.be)
    m = min(len(a), K)
    mb = m - 1
    S = L.S[:L.L + L.P]
    chk = np.array(L.checks)
    return dict(iterations=m, dalpha=np.abs(a[:m] - ra[:m]).max(), dbeta=np.abs(b[:mb] - rb[:mb]).max(),
                orth=np.abs(S.conj() @ S.T - np.eye(len(S))).max(),
                dvec=max(np.linalg.norm(S[L.L + j] - RU[j]) for j in range(min(L.P, K))),   # same sign convention: positive beta
                maxcoef=L.maxcoef, fold_rho=chk[:, 0].max(), fold_gam=chk[:, 1].max())


def run(label, complex_, plant, n=4000, K=260, locked=0):
    r = measure(complex_, plant, n, K, locked)
    return ["%s: %d iterations" % (label, r["iterations"]),
            "  max|dalpha| %.2e  max|dbeta| %.2e  (tolerance 1e-10 ||A|| = 1.2e-09);  max|S^H S - I| %.2e;  max|u_j - u_j(ref)| %.2e" % (
                r["dalpha"], r["dbeta"], r["orth"], r["dvec"]),
            "  largest stored-basis coefficient of an operator input (relative) %.1e;  fold vs truth: rho %.1e, gam %.1e" % (
                r["maxcoef"], r["fold_rho"], r["fold_gam"])]


def main():
    out = ["two-iterations-per-sweep Gram-Schmidt, kernel-structured model (tools/pair_gs_model.py); n = 4000, 260 iterations"]
    out += run("real symmetric, restart pass with 3 locked eigenvectors (n = 1500, 160 iterations)", False, 0.0, 1500, 160, 3)
    out += run("complex Hermitian, 2 locked eigenvectors (n = 1500, 160 iterations)", True, 0.0, 1500, 160, 2)
    for label, cplx, plant in (("real symmetric", False, 0.0), ("complex Hermitian", True, 0.0),
                               ("real, components of relative size 1e-8 planted in one operator input", False, 1e-8),
                               ("real, 1e-6 planted", False, 1e-6), ("real, 1e-3 planted", False, 1e-3),
                               ("complex, 1e-6 planted", True, 1e-6)):
        out += run(label, cplx, plant)
    out += ["",
            "Reading: one pass over the stored basis per TWO Lanczos iterations reproduces the recurrence of full re-orthogonalisation to",
            "1e-14 (real and complex), the stored vectors themselves to 5e-14, with every stored-basis coefficient of an operator input at",
            "a few 1e-14.  Every quantity the fold derives (rho, gam) equals its directly computed value to rounding even when components",
            "of relative size 1e-3 are planted: the basis stays exact.  What is NOT tracked is a term of second order in the planted",
            "components in one beta (8.7 d^2: 9e-12 at d = 1e-6, 5e-15 at 1e-8): the device code therefore uses the form only while the",
            "largest relative coefficient stays below kPairGate = 1e-8 and falls back to the one-sweep form (exact for any size) beyond.",
            "Traffic per pair (k, k+1): three-term 1 (3R 1W) + sweep (k - 2 stored vectors + 4 raw read, 3 written; the second three-term",
            "update is formed inside the sweep) = s n (k + 9) against 2 s n (k + 4.5) of the one-sweep form."]
    text = "\n".join(out) + "\n"
    print(text)
    if "--no-write" not in sys.argv:
        with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05_pair_gs_model.txt"), "w") as f:
            f.write(text)


if __name__ == "__main__":
    main()
