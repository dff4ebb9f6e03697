"""
ORACLE - TEST INFRASTRUCTURE ONLY.

ctypes front-end of the plain-C restatement (oracle/qampy_oracle.c) exposing the reference's compiled entry points with
their original signatures (``#pythran export`` lines of qampy/core/equalisation/pythran_equalisation.py:33-36, 78-79,
128-129, 304-305 and qampy/core/pythran_dsp.py:45-46, 133-136).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import this module; it is the
checker, never the product.  Parity status: PINNED - tests/test_oracle_golden.py checks every function against vectors
captured from the imported reference (tests/golden/gen_golden.py).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

METHODS = ("cma", "cma2", "sgncma", "mcma", "rde", "mrde", "sbd", "mddma", "dd", "sbd_data")
METHODS_REAL = ("cma", "sgncma", "dd", "dd_data")

_libs = {}


def build(fast_native=False):
    """Compile the restatement with gcc (strict checker build + reference-flag build)."""
    subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B", "libqampy_oracle_fast.so"] if fast_native else []))


def _load(fast=False):
    name = "libqampy_oracle_fast.so" if fast else "libqampy_oracle.so"
    if name not in _libs:
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        _libs[name] = C.CDLL(path)
    return _libs[name]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _suffix(dtype):
    dtype = np.dtype(dtype)
    if dtype in (np.complex64, np.float32):
        return "_f32", np.float32, np.complex64
    if dtype in (np.complex128, np.float64):
        return "_f64", np.float64, np.complex128
    raise TypeError("unsupported dtype %s" % dtype)


def _modes_arr(modes, nmax):
    if modes is None:
        return np.arange(nmax, dtype=np.int64)
    return np.ascontiguousarray(np.atleast_1d(modes), dtype=np.int64)


def train_equaliser(E, TrSyms, Niter, os_, mu, wx, modes, adaptive, symbols, method, fast=False):
    """pythran_equalisation.train_equaliser (:128-173): returns (err, wx, mu); wx is updated in place."""
    if method not in METHODS:
        raise ValueError("Unknown method %s" % method)
    suf, rt, ct = _suffix(E.dtype)
    assert E.dtype == ct and wx.dtype == ct and symbols.dtype == ct and E.flags.c_contiguous and wx.flags.c_contiguous
    symbols = np.ascontiguousarray(symbols)
    nmodes, L = E.shape
    ntaps = wx.shape[-1]
    modes = _modes_arr(modes, nmodes)
    err = np.zeros((nmodes, TrSyms * Niter), dtype=ct)
    mu_c = (C.c_float if rt is np.float32 else C.c_double)(mu)
    fn = getattr(_load(fast), "qo_train_equaliser" + suf)
    rc = fn(_p(E), C.c_int(nmodes), C.c_long(L), C.c_long(TrSyms), C.c_int(Niter), C.c_int(os_), C.byref(mu_c),
            _p(wx), C.c_int(ntaps), _p(modes), C.c_int(modes.size), C.c_int(bool(adaptive)),
            _p(symbols), C.c_long(symbols.shape[1]), C.c_int(METHODS.index(method)), _p(err))
    if rc:
        raise ValueError("Unknown method %s" % method)
    return err, wx, rt(mu_c.value)


def train_equaliser_realvalued(E, TrSyms, Niter, os_, mu, wx, modes, adaptive, symbols, method, fast=False):
    """pythran_equalisation.train_equaliser_realvalued (:78-108)."""
    if method not in METHODS_REAL:
        raise ValueError("Unknown method %s" % method)
    suf, rt, ct = _suffix(E.dtype)
    assert E.dtype == rt and wx.dtype == rt and symbols.dtype == rt and E.flags.c_contiguous and wx.flags.c_contiguous
    symbols = np.ascontiguousarray(symbols)
    nmodes, L = E.shape
    ntaps = wx.shape[-1]
    modes = _modes_arr(modes, nmodes)
    err = np.zeros((nmodes, TrSyms * Niter), dtype=rt)
    mu_c = (C.c_float if rt is np.float32 else C.c_double)(mu)
    fn = getattr(_load(fast), "qo_train_equaliser_real" + suf)
    rc = fn(_p(E), C.c_int(nmodes), C.c_long(L), C.c_long(TrSyms), C.c_int(Niter), C.c_int(os_), C.byref(mu_c),
            _p(wx), C.c_int(ntaps), _p(modes), C.c_int(modes.size), C.c_int(bool(adaptive)),
            _p(symbols), C.c_long(symbols.shape[1]), C.c_int(METHODS_REAL.index(method)), _p(err))
    if rc:
        raise ValueError("Unknown method %s" % method)
    return err, wx, rt(mu_c.value)


def apply_filter_to_signal(E, os_, wx, modes=None, fast=False):
    """pythran_equalisation.apply_filter_to_signal (:33-76), complex and real overloads."""
    suf, rt, ct = _suffix(E.dtype)
    assert E.dtype == wx.dtype and E.flags.c_contiguous and wx.flags.c_contiguous
    nmodes, L = E.shape
    ntaps = wx.shape[-1]
    modes = _modes_arr(modes, wx.shape[0])
    N = (L - ntaps + 1) // os_
    out = np.zeros((modes.size, N), dtype=E.dtype)
    name = "qo_apply_filter" + ("" if np.iscomplexobj(E) else "_real") + suf
    getattr(_load(fast), name)(_p(E), C.c_int(nmodes), C.c_long(L), C.c_int(os_), _p(wx), C.c_int(ntaps),
                               _p(modes), C.c_int(modes.size), _p(out))
    return out


def bps(E, testangles, symbols, N, fast=False):
    """pythran_dsp.bps (:45-85): int32 index of the best test angle per symbol."""
    suf, rt, ct = _suffix(E.dtype)
    E = np.ascontiguousarray(E)
    testangles = np.ascontiguousarray(testangles, dtype=rt)
    symbols = np.ascontiguousarray(symbols, dtype=ct)
    comp = np.ascontiguousarray(np.exp(1j * testangles)).astype(ct, copy=False)   # :72, formed by numpy like the reference
    p, A = testangles.shape
    L = E.shape[0]
    idx = np.zeros(L, dtype=np.int32)
    rc = getattr(_load(fast), "qo_bps" + suf)(_p(E), C.c_long(L), _p(comp), C.c_long(p), C.c_int(A), _p(symbols),
                                              C.c_int(symbols.size), C.c_int(N), _p(idx))
    if rc:
        raise MemoryError("oracle bps could not allocate (L, A) work arrays")
    return idx


def select_angles(angles, idx):
    """pythran_dsp.select_angles (:133-153)."""
    suf, rt, ct = _suffix(angles.dtype)
    angles = np.ascontiguousarray(angles)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    p, A = angles.shape
    L = idx.shape[0] if p <= 1 else p
    out = np.zeros(L, dtype=rt)
    getattr(_load(), "qo_select_angles" + suf)(_p(angles), C.c_long(p), C.c_int(A), _p(idx), C.c_long(L), _p(out))
    return out


def make_decision(E, symbols, fast=False):
    """pythran_equalisation.make_decision (:304-334): (decided symbols, |distance|, index)."""
    suf, rt, ct = _suffix(E.dtype)
    E = np.ascontiguousarray(E)
    symbols = np.ascontiguousarray(symbols, dtype=ct)
    L = E.shape[0]
    det = np.zeros(L, dtype=ct)
    dist = np.zeros(L, dtype=rt)
    idx = np.zeros(L, dtype=np.int32)
    getattr(_load(fast), "qo_make_decision" + suf)(_p(E), C.c_long(L), _p(symbols), C.c_int(symbols.size), _p(det),
                                                   _p(dist), _p(idx))
    return det, dist, idx
