"""Shared helpers for the parity tests."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def c2list(a):
    a = np.asarray(a)
    if np.iscomplexobj(a):
        return {"re": a.real.tolist(), "im": a.imag.tolist()}
    return a.tolist()


def list2c(v):
    if isinstance(v, dict):
        return np.asarray(v["re"]) + 1j * np.asarray(v["im"])
    return np.asarray(v)


def overlap(a, b):
    """|<a,b>| / (|a||b|): 1 for parallel vectors, sign/phase independent (T2:66-72)."""
    return abs(np.vdot(a, b)) / (np.linalg.norm(a) * np.linalg.norm(b))


def csr_matvec(csr, x):
    rp, ci, va = csr
    import scipy.sparse as sp

    n = rp.shape[0] - 1
    return sp.csr_matrix((va, ci, rp), shape=(n, x.shape[0])) @ x


def residual(csr, lam, v):
    return np.linalg.norm(csr_matvec(csr, v) - lam * v)


def inf_norm(csr):
    rp, ci, va = csr
    return np.max(np.add.reduceat(np.abs(va), rp[:-1]))
