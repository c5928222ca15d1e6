"""numpy model of a TWO-iterations-per-sweep block Gram-Schmidt form: the operator is applied
twice to raw vectors before ONE sweep over the stored basis takes the coefficients of both and applies the late updates of
the previous pair — the basis streams once per two Lanczos iterations.  Everything else is small dense algebra on
inner products ("frame algebra"): the two new Lanczos vectors and the next raw vector are known as COEFFICIENTS over the
frame F = [U (stored, orthonormal), r, y1, y2] with r the raw vector carried over, y1 = A r, y2 = A (three-term of y1), whose
Gram matrix the sweep delivers and on which the action of A is known where it is needed (A U through the recorded
tridiagonal, A r = y1, A y1 through y2).

    python tools/lagged2_gs_model.py          -> profiles/r04_lagged2_gs_model.txt

Purpose: decide whether the recurrence stays at the reference's numbers (alpha / beta to 1e-10 ||A||) over hundreds of
iterations before any kernel is written.  The model keeps vectors explicitly where the kernels would (r, y1, y2 and the
stored basis) and everything else as coefficients, and forms inner products only the way one sweep could."""
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


def two_per_sweep(explicit_tail):
    """State between sweeps: the stored basis U[:p] (complete, orthonormal to rounding) and ONE real vector r with
    u_p = (r - U[:p] g) / rho  (g = U[:p]^T r and rho known from the last sweep's inner products).
    A double step:  y1 = A r ;  z = y1 - a1 r - b_prev * rho * U[p-1]  with the ESTIMATE a1 = <r, y1> / <r, r>  (raw three-term,
    unnormalised: a real vector the operator can take) ;  y2 = A z.
    ONE sweep: writes u_p (late update from r, g), takes U[:p]^T y1, U[:p]^T y2 (and thereby U^T z), and the 3 x 3 Gram matrix of
    (r, y1, y2).  Frame algebra then yields alpha_p, beta_p, alpha_{p+1}, beta_{p+1} exactly, u_{p+1} and the next raw vector as
    coefficients; u_{p+1} is materialised by the NEXT sweep (here: at once, explicit pass flagged as `late`), the next r is
    formed from the real vectors at hand plus its (tiny) stored-basis part, which a kernel would fold into the next sweep."""
    U = np.zeros((K + 2, n)); U[0] = v0
    al, be = [], []
    # first iteration clean
    y = A @ U[0]; a = U[0] @ y; w = y - a * U[0]
    al.append(a)
    p = 1                      # stored basis size
    r = w; g = U[:p] @ r; rho = np.sqrt(r @ r - g @ g); be.append(rho)
    worst = 0.0
    while p + 1 <= K - 1 and len(al) + 2 <= K:
        # ---- two operator applications on raw vectors
        y1 = A @ r
        rr = r @ r
        a1 = (r @ y1) / rr
        bprev = be[-2] if len(be) >= 2 else 0.0
        z = y1 - a1 * r - (bprev * rho * U[p - 1] if p >= 1 and len(be) >= 2 else 0.0) if False else y1 - a1 * r
        y2 = A @ z
        # ---- ONE sweep over the stored basis: late update of u_p, coefficients of y1 and y2
        up = (r - g @ U[:p]) / rho
        h1 = U[:p] @ y1
        h2 = U[:p] @ y2
        G3 = np.array([[rr, r @ y1, r @ y2], [r @ y1, y1 @ y1, y1 @ y2], [r @ y2, y1 @ y2, y2 @ y2]])
        # ---- frame algebra.  Frame F = [U[:p] | r, y1, y2]; coefficients c = (cU (p), c3 (3)).
        P = p
        def gram(c, d):
            cu, c3 = c; du, d3 = d
            Uf = np.stack([g, h1, h2], axis=1)       # U^T [r y1 y2]  (p x 3)
            return cu @ du + cu @ (Uf @ d3) + du @ (Uf @ c3) + c3 @ (G3 @ d3)
        zero = np.zeros(P)
        e = lambda i: np.eye(3)[i]
        # images under A of the frame vectors we need: A U[:p] (recorded tridiagonal + boundary), A r = y1, A y1 via y2
        alh, beh = np.array(al), np.array(be)
        def A_U(cu):
            """A (U cu) as frame coefficients: A u_j = be[j-1] u_{j-1} + al[j] u_j + be[j] u_{j+1}, u_p = (r - U g)/rho pending."""
            out = np.zeros(P); c3 = np.zeros(3)
            for j in range(P):
                if cu[j] == 0.0: continue
                if j >= 1: out[j - 1] += beh[j - 1] * cu[j]
                out[j] += alh[j] * cu[j]
                if j + 1 < P: out[j + 1] += beh[j] * cu[j]
                else:  # u_p = (r - U g) / rho
                    out -= beh[j] * cu[j] * g / rho; c3[0] += beh[j] * cu[j] / rho
            return out, c3
        def A_frame(c):
            cu, c3 = c
            ou, o3 = A_U(cu)
            # A r = y1 ; A y1: y2 = A z = A y1 - a1 A r  =>  A y1 = y2 + a1 y1 ; A y2 unknown (must not be needed)
            assert abs(c3[2]) < 1e-300
            o3 = o3 + np.array([0.0, c3[0] + a1 * c3[1], c3[1]])
            return ou, o3
        def axpy(c, s, d):
            return (c[0] + s * d[0], c[1] + s * d[1])
        def orth(c, against):
            for q in against:
                c = axpy(c, -gram(q, c), q)
            return c
        Ubasis = [((np.eye(P)[j]), np.zeros(3)) for j in range(P)]
        # u_p as coefficients
        cup = (-g / rho, e(0) / rho)
        worst = max(worst, abs(gram(cup, cup) - 1.0))
        # iteration p+1 (1-based numbering of alpha): w = A u_p - alpha u_p - beta u_{p-1}, full reorth
        Aup = A_frame(cup)
        a_p = gram(cup, Aup)
        w = axpy(Aup, -a_p, cup)
        w = axpy(w, -rho, Ubasis[P - 1])
        w = orth(w, Ubasis + [cup])
        b_p = np.sqrt(max(gram(w, w), 0.0))
        if not b_p > 0.0:
            return np.array(al), np.array(be), U, -float(p)   # derived norm not positive: the scheme broke down at basis size p
        cup1 = (w[0] / b_p, w[1] / b_p)
        # iteration p+2
        Aup1 = A_frame(cup1)
        a_p1 = gram(cup1, Aup1)
        w2 = axpy(Aup1, -a_p1, cup1)
        w2 = axpy(w2, -b_p, cup)
        w2 = orth(w2, Ubasis + [cup, cup1])
        b_p1 = np.sqrt(max(gram(w2, w2), 0.0))
        al += [a_p, a_p1]; be += [b_p, b_p1]
        # ---- materialise: u_p (done by this sweep), u_{p+1} (the NEXT sweep's late update; here explicit), next raw vector
        U[p] = up
        U[p + 1] = cup1[0] @ U[:P] + cup1[1][0] * r + cup1[1][1] * y1 + cup1[1][2] * y2
        if explicit_tail is None:      # fully materialised (a pass over the whole stored basis)
            rn = w2[0] @ U[:P] + w2[1][0] * r + w2[1][1] * y1 + w2[1][2] * y2   # u_{p+2} * b_p1 as a real vector
        else:                          # what a kernel would form: the three raw vectors and the last `explicit_tail` stored vectors;
            lo = max(0, P - explicit_tail)  # the O(eps) rest of the stored-basis part is dropped — the next sweep projects it out anyway
            rn = w2[0][lo:] @ U[lo:P] + w2[1][0] * r + w2[1][1] * y1 + w2[1][2] * y2
        p += 2
        r = rn
        g = U[:p] @ r                      # (tiny: the frame algebra orthogonalised it; a kernel takes these in the next sweep)
        rho = np.sqrt(r @ r - g @ g)
        be[-1] = rho                       # the measured norm of the materialised vector replaces the Gram-derived one
    m = min(len(al), K)
    return np.array(al[:m]), np.array(be[:m]), U, worst


def two_per_sweep_predicted():
    """The kernel-shaped variant: everything the sweep WRITES (u_p, u_{p+1} and the next raw vector) is formed with coefficients
    known BEFORE the sweep — g and rho from the previous fold, the 3 x 3 Gram matrix of (r, y1, y2) from a small kernel, and the
    stored-basis coefficients of y1 and y2 PREDICTED through the recorded tridiagonal:
        <u_j, y1> = <A u_j, r>   = be[j-1] g[j-1] + al[j] g[j] + be[j] g[j+1]          (g[p] := <u_p, r> = rho)
        <u_j, y2> = <A u_j, z>   = the same stencil on  zU = U^T z = h1 - a1 g,  zU[p] = <u_p, z>
    The one pass over the basis forms the three outputs with those coefficients and MEASURES h1 = U^T y1, h2 = U^T y2; the fold
    then computes, from the measured numbers, the recurrence coefficients and the exact inner products of the next raw vector
    with every stored vector (its new g and rho).  What the predicted coefficients leave along span(U) is fresh rounding."""
    U = np.zeros((K + 2, n)); U[0] = v0
    al, be = [], []
    y = A @ U[0]; a = U[0] @ y; w = y - a * U[0]
    al.append(a)
    p = 1
    r = w; g = U[:p] @ r; rho = np.sqrt(r @ r - g @ g); be.append(rho)
    maxdev = 0.0
    while len(al) + 2 <= K:
        P = p
        y1 = A @ r
        rr = r @ r; ry1 = r @ y1
        a1 = ry1 / rr
        z = y1 - a1 * r
        y2 = A @ z
        G3 = np.array([[rr, ry1, r @ y2], [ry1, y1 @ y1, y1 @ y2], [r @ y2, y1 @ y2, y2 @ y2]])   # small kernel over r, y1, y2
        alh, beh = np.array(al), np.array(be)

        def stencil(v, vP):
            """(T applied to the coefficient vector v of length P, with the pending entry vP at index P)"""
            out = alh * v
            out[1:] += beh[:P - 1] * v[:-1]
            out[:-1] += beh[:P - 1] * v[1:]
            out[P - 1] += beh[P - 1] * vP
            return out
        # ---- predictions (before the sweep)
        h1p = stencil(g, rho)
        up_y1 = (ry1 - g @ h1p) / rho                    # <u_p, y1> with the predicted h1
        zU = h1p - a1 * g
        zP = up_y1 - a1 * rho
        h2p = stencil(zU, zP)

        def algebra(h1, h2):
            Uf = np.stack([g, h1, h2], axis=1)
            def gram(c, d):
                return c[0] @ d[0] + c[0] @ (Uf @ d[1]) + d[0] @ (Uf @ c[1]) + c[1] @ (G3 @ d[1])
            def A_U(cu):
                out = alh * cu
                out[1:] += beh[:P - 1] * cu[:-1]
                out[:-1] += beh[:P - 1] * cu[1:]
                c3 = np.zeros(3)
                out = out - beh[P - 1] * cu[P - 1] * g / rho; c3[0] += beh[P - 1] * cu[P - 1] / rho
                return out, c3
            def A_frame(c):
                ou, o3 = A_U(c[0])
                return ou, o3 + np.array([0.0, c[1][0] + a1 * c[1][1], c[1][1]])
            def axpy(c, s_, d):
                return (c[0] + s_ * d[0], c[1] + s_ * d[1])
            def orthU(c):      # against every stored vector at once: <u_j, c> = c_U[j] + Uf[j] . c_3
                return (c[0] - (c[0] + Uf @ c[1]), c[1])
            cup = (-g / rho, np.array([1.0, 0, 0]) / rho)
            Aup = A_frame(cup)
            a_p = gram(cup, Aup)
            w = axpy(Aup, -a_p, cup)
            w = (w[0].copy(), w[1]); w[0][P - 1] -= rho
            w = orthU(w); w = axpy(w, -gram(cup, w), cup)
            b_p = np.sqrt(gram(w, w))
            cup1 = (w[0] / b_p, w[1] / b_p)
            Aup1 = A_frame(cup1)
            a_p1 = gram(cup1, Aup1)
            w2 = axpy(Aup1, -a_p1, cup1)
            w2 = axpy(w2, -b_p, cup)
            w2 = orthU(w2); w2 = axpy(w2, -gram(cup, w2), cup); w2 = axpy(w2, -gram(cup1, w2), cup1)
            return cup, cup1, w2, a_p, b_p, a_p1, gram
        cup_p, cup1_p, w2_p, *_ = algebra(h1p, h2p)
        # ---- ONE sweep: outputs with the predicted coefficients, measurements of h1, h2
        U[p] = cup_p[0] @ U[:P] + cup_p[1][0] * r
        U[p + 1] = cup1_p[0] @ U[:P] + cup1_p[1] @ np.stack([r, y1, y2])
        rn = w2_p[0] @ U[:P] + w2_p[1] @ np.stack([r, y1, y2])
        h1 = U[:P] @ y1
        h2 = U[:P] @ y2
        maxdev = max(maxdev, np.abs(h1 - h1p).max(), np.abs(h2 - h2p).max())
        # ---- fold: the numbers of the recurrence from the MEASURED inner products; inner products of rn with the stored basis
        cup, cup1, w2, a_p, b_p, a_p1, gram = algebra(h1, h2)
        al += [a_p, a_p1]; be += [b_p]
        Uf = np.stack([g, h1, h2], axis=1)
        gn_old = w2_p[0] + Uf @ w2_p[1]                       # <u_j, rn>, j < P (measured Gram, predicted coefficients)
        gn_p = gram(cup_p, w2_p); gn_p1 = gram(cup1_p, w2_p)  # with the two vectors written by this sweep
        rn2 = gram(w2_p, w2_p)
        p += 2
        r = rn
        g = np.concatenate([gn_old, [gn_p, gn_p1]])
        rho = np.sqrt(rn2 - g @ g)
        be.append(rho)
    m = min(len(al), K)
    return np.array(al[:m]), np.array(be[:m]), U, maxdev


def main():
    ra, rb, RU = reference()
    lines = ["model: n = %d random symmetric + diagonal 2..12, %d iterations; reference = full re-orthogonalisation" % (n, K)]
    for label, tail in (("next raw vector fully materialised", None), ("next raw vector from (r, y1, y2) + the last 2 stored vectors", 2),
                        ("next raw vector from (r, y1, y2) only", 0)):
        a, b, U, worst = two_per_sweep(tail)
        if worst < 0:
            lines.append("two iterations per sweep, %s: derived norm not positive at basis size %d" % (label, int(-worst)))
            continue
        m = min(len(a), len(ra))
        mb = m - 1
        lines += ["two iterations per sweep (frame algebra on [U, r, A r, A(A r - a r)]), %s: %d iterations compared" % (label, m),
                  "  max|dalpha| %.2e   max|dbeta| %.2e   (tolerance 1e-10 * ||A|| = %.1e)" % (
                      np.max(np.abs(a[:m] - ra[:m])), np.max(np.abs(b[:mb] - rb[:mb])), 1e-10 * 12),
                  "  orthogonality of the materialised basis max|U^T U - I| = %.2e ; | |u_p|_G - 1 | <= %.1e" % (
                      np.abs(U[:m] @ U[:m].T - np.eye(m)).max(), worst)]
        for k in (10, 100, m - 1):
            lines.append("  k = %3d: dalpha %.1e dbeta %.1e" % (k, abs(a[k] - ra[k]), abs(b[min(k, mb - 1)] - rb[min(k, mb - 1)])))
    a, b, U, dev = two_per_sweep_predicted()
    m = min(len(a), len(ra)); mb = m - 1
    lines += ["two iterations per sweep, ONE pass over the basis per pair (outputs from predicted coefficients, fold from measured ones): %d iterations compared" % m,
              "  max|dalpha| %.2e   max|dbeta| %.2e   orthogonality max|U^T U - I| = %.2e   max |measured - predicted coefficient| = %.1e" % (
                  np.max(np.abs(a[:m] - ra[:m])), np.max(np.abs(b[:mb] - rb[:mb])), np.abs(U[:m] @ U[:m].T - np.eye(m)).max(), dev)]
    for k in (10, 100, m - 1):
        lines.append("  k = %3d: dalpha %.1e dbeta %.1e" % (k, abs(a[k] - ra[k]), abs(b[min(k, mb - 1)] - rb[min(k, mb - 1)])))
    text = "\n".join(lines) + "\n"
    print(text)
    import os
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r04_lagged2_gs_model.txt"), "w") as f:
        f.write(text)


if __name__ == "__main__":
    main()
