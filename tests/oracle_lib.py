"""ctypes access to the CHECKERS: oracle/_build/liboracle.so (our CPU restatement) and, when present,
oracle/_ref/libref.so (the real reference headers compiled in place).  Test infrastructure only — imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product package."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref.so")

i64, i32, f64, vp = C.c_int64, C.c_int32, C.c_double, C.c_void_p
EPS = float(np.finfo(np.float64).eps)


class Params(C.Structure):
    _fields_ = [
        ("matrix_size", i64),
        ("max_iteration", i64),
        ("eps", f64),
        ("find_maximum", i32),
        ("full_orthogonalize", i32),
        ("num_eigs", i64),
        ("eigenvalue_offset", f64),
        ("num_eigs_per_iteration", i64),
    ]


class Trace(C.Structure):
    _fields_ = [("alpha", vp), ("beta", vp), ("len", vp), ("t_mv", vp), ("t_total", vp)]


def build_oracle():
    """Compile the restatement (and the reference shim when /root/reference exists)."""
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"], check=True)


def _p(a):
    return None if a is None else a.ctypes.data_as(vp)


class _Checker:
    """Common driver for both libraries; `prefix` is 'oracle_' or 'ref_'."""

    def __init__(self, path, prefix):
        self.lib = C.CDLL(path)
        self.prefix = prefix
        L = self.lib
        for sfx in "dz":
            f = getattr(L, prefix + "lanczos_run_" + sfx)
            f.restype = i64
            f.argtypes = [vp, vp, vp, C.POINTER(Params), vp, vp, vp, vp, vp, C.POINTER(Trace)]
        for sfx in "dz":
            f = getattr(L, prefix + "run_iteration_" + sfx)
            f.restype = i64
            f.argtypes = [vp, vp, vp, C.POINTER(Params), vp, i64, i64, vp, vp, vp, vp]
        getattr(L, prefix + "expo_run_d").restype = i64
        getattr(L, prefix + "expo_run_d").argtypes = [vp, vp, vp, C.POINTER(Params), f64, vp, vp, C.POINTER(Trace)]
        getattr(L, prefix + "expo_run_z").restype = i64
        getattr(L, prefix + "expo_run_z").argtypes = [vp, vp, vp, C.POINTER(Params), f64, f64, vp, vp, C.POINTER(Trace)]
        getattr(L, prefix + "taylor_run_d").restype = i64
        getattr(L, prefix + "taylor_run_d").argtypes = [vp, vp, vp, C.POINTER(Params), f64, vp, vp]
        getattr(L, prefix + "taylor_run_z").restype = i64
        getattr(L, prefix + "taylor_run_z").argtypes = [vp, vp, vp, C.POINTER(Params), f64, f64, vp, vp]
        getattr(L, prefix + "tridiag_eig").restype = i64
        getattr(L, prefix + "tridiag_eig").argtypes = [i64, vp, vp, i64, vp, vp]
        getattr(L, prefix + "mth_eigenvalue").restype = f64
        getattr(L, prefix + "mth_eigenvalue").argtypes = [i64, vp, vp, i64]
        getattr(L, prefix + "inner_prod_z").argtypes = [i64, vp, vp, vp]
        getattr(L, prefix + "m_norm_z").restype = f64
        getattr(L, prefix + "m_norm_z").argtypes = [i64, vp]
        getattr(L, prefix + "schmidt_orth_z").argtypes = [i64, i64, vp, vp]

    # ------------------------------------------------------------ engines
    def lanczos(self, csr, init, find_maximum, num_eigs=1, max_iteration=None, eps=None, offset=0.0,
                num_eigs_per_iteration=5, trace=True):
        rp, ci, va = csr
        rp = np.ascontiguousarray(rp, np.int64)
        ci = np.ascontiguousarray(ci, np.int32)
        va = np.ascontiguousarray(va)
        n = rp.shape[0] - 1
        z = va.dtype == np.complex128
        init = np.ascontiguousarray(init, dtype=va.dtype)
        p = Params(n, n if max_iteration is None else max_iteration, EPS * 1e3 if eps is None else eps,
                   int(find_maximum), 0, num_eigs, offset, num_eigs_per_iteration)
        vals = np.zeros(num_eigs)
        vecs = np.zeros((num_eigs, n), dtype=va.dtype)
        counts = np.zeros(4 * num_eigs + 64, dtype=np.int64)
        npass = np.zeros(1, dtype=np.int64)
        cap = int(p.max_iteration) + 2
        alpha, beta = np.zeros(cap), np.zeros(cap)
        ln = np.zeros(1, dtype=np.int64)
        tmv, ttot = np.zeros(1), np.zeros(1)
        tr = Trace(_p(alpha) if trace else None, _p(beta) if trace else None, _p(ln), _p(tmv), _p(ttot))
        fn = getattr(self.lib, self.prefix + "lanczos_run_" + ("z" if z else "d"))
        found = fn(_p(rp), _p(ci), _p(va), C.byref(p), _p(init), _p(vals), _p(vecs), _p(counts), _p(npass), C.byref(tr))
        k = int(ln[0])
        return {
            "eigenvalues": vals[:found].copy(),
            "eigenvectors": vecs[:found].copy(),
            "iter_counts": [int(c) for c in counts[: int(npass[0])]],
            "alpha": alpha[:k].copy(),
            "beta": beta[:k].copy(),
            "t_mv": float(tmv[0]),
            "t_total": float(ttot[0]),
        }

    def run_iteration(self, csr, init, find_maximum, nroot, orth=None, max_iteration=None, eps=None, offset=0.0):
        """LambdaLanczos<T>::run_iteration (LL:216-322): one pass, nroot pairs, orthogonalised against the rows of orth."""
        rp, ci, va = csr
        rp = np.ascontiguousarray(rp, np.int64)
        ci = np.ascontiguousarray(ci, np.int32)
        va = np.ascontiguousarray(va)
        n = rp.shape[0] - 1
        init = np.ascontiguousarray(init, dtype=va.dtype)
        orth = np.zeros((0, n), dtype=va.dtype) if orth is None else np.ascontiguousarray(orth, dtype=va.dtype).reshape(-1, n)
        p = Params(n, n if max_iteration is None else max_iteration, EPS * 1e3 if eps is None else eps,
                   int(find_maximum), 0, 1, offset, 5)
        vals = np.zeros(nroot)
        vecs = np.zeros((nroot, n), dtype=va.dtype)
        found = np.zeros(1, dtype=np.int64)
        fn = getattr(self.lib, self.prefix + "run_iteration_" + ("z" if va.dtype == np.complex128 else "d"))
        it = fn(_p(rp), _p(ci), _p(va), C.byref(p), _p(init), int(nroot), orth.shape[0], _p(orth), _p(vals), _p(vecs),
                _p(found))
        k = int(found[0])
        return {"eigenvalues": vals[:k].copy(), "eigenvectors": vecs[:k].copy(), "itern": int(it)}

    def expo(self, csr, a, input, max_iteration=None, eps=None, full_orthogonalize=False, taylor=False):
        rp, ci, va = csr
        rp = np.ascontiguousarray(rp, np.int64)
        ci = np.ascontiguousarray(ci, np.int32)
        va = np.ascontiguousarray(va)
        n = rp.shape[0] - 1
        z = va.dtype == np.complex128
        inp = np.ascontiguousarray(input, dtype=va.dtype)
        out = np.zeros_like(inp)
        p = Params(n, n if max_iteration is None else max_iteration, EPS * 1e2 if eps is None else eps, 0,
                   int(full_orthogonalize), 1, 0.0, 5)
        tmv, ttot = np.zeros(1), np.zeros(1)
        tr = Trace(None, None, None, _p(tmv), _p(ttot))
        name = self.prefix + ("taylor_run_" if taylor else "expo_run_") + ("z" if z else "d")
        fn = getattr(self.lib, name)
        args = [_p(rp), _p(ci), _p(va), C.byref(p)]
        args += [float(np.real(a)), float(np.imag(a))] if z else [float(a)]
        args += [_p(inp), _p(out)]
        if not taylor:
            args.append(C.byref(tr))
        it = fn(*args)
        return out, int(it), {"t_mv": float(tmv[0]), "t_total": float(ttot[0])}

    # ------------------------------------------------------------ small pieces
    def tridiag_eig(self, alpha, beta, want_vectors=True):
        alpha = np.ascontiguousarray(alpha, np.float64)
        beta = np.ascontiguousarray(beta, np.float64)
        m = alpha.shape[0]
        ev = np.zeros(m)
        q = np.zeros((m, m)) if want_vectors else None
        unc = getattr(self.lib, self.prefix + "tridiag_eig")(m, _p(alpha), _p(beta), beta.shape[0], _p(ev), _p(q))
        return ev, q, int(unc)

    def mth_eigenvalue(self, alpha, beta, m):
        alpha = np.ascontiguousarray(alpha, np.float64)
        beta = np.ascontiguousarray(np.concatenate([np.asarray(beta, np.float64), np.zeros(1)]))[: alpha.shape[0]]
        beta = np.ascontiguousarray(np.concatenate([beta, np.zeros(alpha.shape[0] - beta.shape[0])]))
        return getattr(self.lib, self.prefix + "mth_eigenvalue")(alpha.shape[0], _p(alpha), _p(beta), int(m))

    def inner_prod(self, a, b):
        a = np.ascontiguousarray(a, np.complex128)
        b = np.ascontiguousarray(b, np.complex128)
        out = np.zeros(1, np.complex128)
        getattr(self.lib, self.prefix + "inner_prod_z")(a.shape[0], _p(a), _p(b), _p(out))
        return complex(out[0])

    def m_norm(self, a):
        a = np.ascontiguousarray(a, np.complex128)
        return getattr(self.lib, self.prefix + "m_norm_z")(a.shape[0], _p(a))

    def schmidt_orth(self, basis, w):
        basis = np.ascontiguousarray(basis, np.complex128)
        w = np.ascontiguousarray(w, np.complex128).copy()
        getattr(self.lib, self.prefix + "schmidt_orth_z")(w.shape[0], basis.shape[0], _p(basis), _p(w))
        return w


class Oracle(_Checker):
    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        super().__init__(ORACLE_SO, "oracle_")
        L = self.lib
        L.oracle_set_threads.restype = C.c_int
        L.oracle_set_threads.argtypes = [C.c_int]
        L.oracle_spmv_d.argtypes = [i64, vp, vp, vp, vp, vp]
        L.oracle_spmv_z.argtypes = [i64, vp, vp, vp, vp, vp]

    def set_threads(self, threads):
        """1 = the reference's single-threaded behaviour (default); <= 0 = all host cores (courtesy baseline)."""
        return self.lib.oracle_set_threads(int(threads))

    def spmv(self, csr, x):
        rp, ci, va = csr
        rp = np.ascontiguousarray(rp, np.int64)
        ci = np.ascontiguousarray(ci, np.int32)
        va = np.ascontiguousarray(va)
        x = np.ascontiguousarray(x, dtype=va.dtype)
        y = np.zeros(rp.shape[0] - 1, dtype=va.dtype)
        fn = self.lib.oracle_spmv_z if va.dtype == np.complex128 else self.lib.oracle_spmv_d
        fn(rp.shape[0] - 1, _p(rp), _p(ci), _p(va), _p(x), _p(y))
        return y


class Reference(_Checker):
    """The real reference (only where oracle/_ref/libref.so exists)."""

    def __init__(self):
        super().__init__(REF_SO, "ref_")
        self.lib.ref_init_mt19937_d.argtypes = [C.c_uint32, i64, vp]
        self.lib.ref_init_mt19937_z.argtypes = [C.c_uint32, i64, vp]

    def init_mt19937(self, seed, n, dtype=np.float64):
        v = np.zeros(n, dtype=dtype)
        fn = self.lib.ref_init_mt19937_z if np.dtype(dtype) == np.complex128 else self.lib.ref_init_mt19937_d
        fn(seed, n, _p(v))
        return v


def have_reference():
    return os.path.exists(REF_SO)


_oracle = None
_reference = None


def oracle():
    global _oracle
    if _oracle is None:
        _oracle = Oracle()
    return _oracle


def reference():
    global _reference
    if _reference is None:
        _reference = Reference()
    return _reference
