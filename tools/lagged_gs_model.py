"""numpy model of the one-sweep (lagged) block Gram-Schmidt form of DESIGN.md section 3.2 (kernels.hip: lagged_kernel,
lagged_fold_kernel), against Lanczos with full re-orthogonalisation on the same matrix and start vector.

    python tools/lagged_gs_model.py            -> profiles/r03_lagged_gs_model.txt

Variants: no compensation (the late coefficients double every iteration), compensation of w and the coefficients with
alpha corrected to first order only, and the full scheme (alpha also loses <e, A e>) — the last one also with a large
perturbation injected at iteration 20 (|c| = 1e-3 and 0.5) to show that the algebra is exact for any size of c."""
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


def lagged(compensate, second_order, inject=0.0):
    U = np.zeros((K + 1, n)); U[0] = v0; al = []; be = []; maxc = []
    r = g = t = s = None; q = 0.0
    for k in range(1, K + 1):
        if k == 20 and inject:                         # leave the lagged form once, to plant a known perturbation
            U[k - 1] = (r - g @ U[:k - 1]) * s; r = None
        if r is None:                                  # clean iteration: operator on a complete u_{k-1}
            x = U[k - 1]; y = A @ x; a = x @ y
            w = y - a * x - (be[-1] * U[k - 2] if k > 1 else 0)
            if k == 20 and inject: w = w + inject * np.linalg.norm(w) * (U[3] - U[7])
            gn = U[:k] @ w
        else:                                          # lagged sweep
            x = r * s; y = A @ x; a = x @ y
            if compensate: a -= 2 * g[-1] + (q if second_order else 0.0)
            wr = y - a * x - (be[-1] * U[k - 2] if k > 1 else 0)
            uc = (r - g @ U[:k - 1]) * s; U[k - 1] = uc  # the late update
            m = U[:k - 1] @ wr
            if compensate:
                d = t.copy(); d[:k - 1] -= a * s * g
                w = wr - d[:k - 1] @ U[:k - 1] - d[k - 1] * uc
                gn = np.concatenate([m - d[:k - 1], [uc @ w]])
            else:
                w = wr; gn = np.concatenate([m, [uc @ w]])
        c1 = w @ w - gn @ gn
        if not c1 > 0: return k, np.array(maxc)
        b = np.sqrt(c1); al.append(a); be.append(b)
        c = gn / b; maxc.append(np.abs(c).max())
        alh, beh = np.array(al), np.array(be)
        tt = np.zeros(k + 1); tt[:k] += alh * c; tt[1:k + 1] += beh * c; tt[:k - 1] += beh[:k - 1] * c[1:]
        q = c @ tt[:k]
        r, g, t, s = w, gn, tt, 1.0 / b
    U[K] = (r - g @ U[:K]) * s
    return np.array(al), np.array(be), U, np.array(maxc)


def main():
    ra, rb, RU = reference()
    lines = ["model: n = %d random symmetric + diagonal 2..12, %d iterations; reference = full re-orthogonalisation" % (n, K),
             "reference orthogonality max|U^T U - I| = %.2e" % np.abs(RU[:K] @ RU[:K].T - np.eye(K)).max(), ""]
    for name, comp, second, inject in (("no compensation", False, False, 0.0),
                                       ("compensated, alpha to first order", True, False, 0.0),
                                       ("compensated, alpha to first order, |c| = 1e-3 injected at k = 20", True, False, 1e-3),
                                       ("full scheme", True, True, 0.0),
                                       ("full scheme, |c| = 1e-3 injected at k = 20", True, True, 1e-3),
                                       ("full scheme, |c| = 0.5 injected at k = 20", True, True, 0.5)):
        out = lagged(comp, second, inject)
        if len(out) == 2:
            mc = out[1]
            lines.append("%-70s derived norm negative at k = %d; |c| at k=10/20/30/40/50 %.0e %.0e %.0e %.0e %.0e, max %.1e"
                         % (name, out[0], mc[9], mc[19], mc[29], mc[39], mc[49], mc.max()))
            continue
        a, b, U, mc = out
        with np.errstate(invalid="ignore"):
            bad = np.flatnonzero(~(np.abs(a - ra) <= 1e-9))
        lines.append("%-70s max|dalpha| %.1e  max|dbeta| %.1e  orth %.1e  max|c| %.1e  |c| at k=10/30/60 %.0e %.0e %.0e  first k with |dalpha|>1e-9: %s"
                     % (name, np.nanmax(np.abs(a - ra)), np.nanmax(np.abs(b - rb)), np.abs(U[:K] @ U[:K].T - np.eye(K)).max(),
                        np.nanmax(mc), mc[9], mc[29], mc[59], bad[0] + 1 if len(bad) else "-"))
    text = "\n".join(lines) + "\n"
    print(text)
    import os
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r03_lagged_gs_model.txt"), "w") as f:
        f.write(text)


if __name__ == "__main__":
    main()
