"""Op-level test hooks over the engine's kernels (nc_op_* in include/nc_mi355x.h): host numpy in/out."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def _p(a):
    return None if a is None else a.ctypes.data


def conv1d(x, weight, bias=None, stride=1, pad=0, dil=1, alpha_in=None, alpha_out=None, residual=None, transposed=False,
           out_pad=0, tanh_out=False, device_index=0):
    x = np.ascontiguousarray(x, np.float32); w = np.ascontiguousarray(weight, np.float32)
    B, Cin, Tin = x.shape
    if transposed:
        Cout, K = w.shape[1], w.shape[2]
        Tout = (Tin - 1) * stride - 2 * pad + K + out_pad
    else:
        Cout, K = w.shape[0], w.shape[2]
        Tout = (Tin + 2 * pad - dil * (K - 1) - 1) // stride + 1
    d = _lib.NcConvDesc(B, Cin, Cout, K, stride, pad, dil, out_pad, Tin, 1 if transposed else 0, 1 if tanh_out else 0)
    f = lambda a: None if a is None else np.ascontiguousarray(a, np.float32)
    b, ai, ao, r = f(bias), f(alpha_in), f(alpha_out), f(residual)
    y = np.empty((B, Cout, Tout), np.float32)
    to = C.c_int64()
    _lib.check(_lib.lib().nc_op_conv1d(device_index, C.byref(d), x.ctypes.data, w.ctypes.data, _p(b), _p(ai), _p(ao), _p(r),
                                       y.ctypes.data, C.byref(to)))
    assert to.value == Tout
    return y


def res_unit(x, w7, b7, a1, a2, w1, b1, dil=1, fused=True, iters=0, device_index=0):
    """One DAC ResidualUnit; returns y (and the average ms per repetition when iters > 0)."""
    f = lambda a: np.ascontiguousarray(a, np.float32)
    x, w7, b7, a1, a2, w1, b1 = map(f, (x, w7, b7, a1, a2, w1, b1))
    B, Cc, T = x.shape
    y = np.empty_like(x)
    ms = C.c_double()
    _lib.check(_lib.lib().nc_op_res_unit(device_index, B, Cc, T, dil, x.ctypes.data, w7.ctypes.data, b7.ctypes.data, a1.ctypes.data,
                                         a2.ctypes.data, w1.ctypes.data, b1.ctypes.data, 1 if fused else 0, y.ctypes.data, iters,
                                         C.byref(ms)))
    return (y, ms.value) if iters > 0 else y


def vq_argmin(z_e, codebook, device_index=0):
    z = np.ascontiguousarray(z_e, np.float32); cb = np.ascontiguousarray(codebook, np.float32)
    B, D, T = z.shape
    idx = np.empty((B, T), np.int64); st = np.empty_like(z)
    _lib.check(_lib.lib().nc_op_vq_argmin(device_index, z.ctypes.data, B, D, T, cb.ctypes.data, cb.shape[0], idx.ctypes.data,
                                          st.ctypes.data))
    return idx, st


def euclid_rvq(residual, codebooks, form=1, device_index=0):
    """Encodec RVQ encode on residual [B,D,T] with codebooks [n_q,N,D] -> (codes [B,n_q,T], residual after the last stage)."""
    r = np.ascontiguousarray(residual, np.float32); cb = np.ascontiguousarray(codebooks, np.float32)
    B, D, T = r.shape
    nq, N, _ = cb.shape
    codes = np.empty((B, nq, T), np.int64); out = np.empty_like(r)
    _lib.check(_lib.lib().nc_op_euclid_rvq(device_index, r.ctypes.data, B, D, T, cb.ctypes.data, nq, N, int(form), codes.ctypes.data, out.ctypes.data))
    return codes, out


def fold_weight_norm(v, g):
    v = np.ascontiguousarray(v, np.float32); g = np.ascontiguousarray(g, np.float32).reshape(-1)
    w = np.empty_like(v)
    _lib.check(_lib.lib().nc_op_fold_weight_norm(v.ctypes.data, g.ctypes.data, v.shape[0], int(np.prod(v.shape[1:])), w.ctypes.data))
    return w
