"""TEST INFRASTRUCTURE ONLY -- ctypes bindings for the parity checkers.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  It binds

* ``oracle/_ref/libsplpak_ref.so`` -- the unmodified reference compiled from
  ``/root/reference/src/splpak.F90`` (``kind = "reference"``), and
* ``oracle/libsplpak_oracle.so``   -- the C restatement of the reference
  algorithm (``kind = "port"``), see ``splpak_oracle.c``.

Both expose the same Python surface: ``fit(...) -> (coef, ierror, work)`` and
``evaluate(...) -> (values, ierror)``.
"""
from __future__ import annotations

import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)


def _ptr(a, ty):
    return a.ctypes.data_as(ty)


def ref_available(real32: bool = False) -> bool:
    name = "libsplpak_ref32.so" if real32 else "libsplpak_ref.so"
    return os.path.exists(os.path.join(_HERE, "_ref", name))


def port_available() -> bool:
    return os.path.exists(os.path.join(_HERE, "libsplpak_oracle.so"))


class _Base:
    dtype = np.float64

    def _prep(self, ndim, xdata, ydata, wdata, xmin, xmax, nodes):
        dt = self.dtype
        xdata = np.ascontiguousarray(xdata, dtype=dt)
        if xdata.ndim == 1:
            xdata = xdata.reshape(-1, 1)
        ydata = np.ascontiguousarray(ydata, dtype=dt)
        if wdata is not None:
            wdata = np.ascontiguousarray(wdata, dtype=dt)
        xmin = np.ascontiguousarray(np.atleast_1d(xmin), dtype=dt)
        xmax = np.ascontiguousarray(np.atleast_1d(xmax), dtype=dt)
        nodes = np.ascontiguousarray(np.atleast_1d(nodes), dtype=np.int32)
        return xdata, ydata, wdata, xmin, xmax, nodes


class Reference(_Base):
    """The real reference (Fortran) through oracle/ref_shim.f90."""

    kind = "reference"

    def __init__(self, real32: bool = False):
        name = "libsplpak_ref32.so" if real32 else "libsplpak_ref.so"
        self.lib = C.CDLL(os.path.join(_HERE, "_ref", name))
        self.dtype = np.float32 if real32 else np.float64
        self.rp = _fp if real32 else _dp
        self.rt = C.c_float if real32 else C.c_double
        assert self.lib.ref_wp_bytes() == (4 if real32 else 8)

    def fit(self, ndim, xdata, ydata, wdata, xmin, xmax, nodes, xtrap, nwrk=None, ncf=None,
            ndata=None, l1xdat=None):
        """``wdata=None`` -> splcc.  Returns (coef, ierror, work)."""
        xdata, ydata, wdata, xmin, xmax, nodes = self._prep(ndim, xdata, ydata, wdata, xmin, xmax, nodes)
        rp = self.rp
        if ndata is None:
            ndata = xdata.shape[0]
        if l1xdat is None:
            l1xdat = xdata.shape[1]
        ncol = int(np.prod(nodes[:max(ndim, 1)].astype(np.int64)))
        if ncf is None:
            ncf = ncol
        if nwrk is None:
            nwrk = ncol * (ncol + 1)
        coef = np.zeros(max(ncf, 1), dtype=self.dtype)
        work = np.zeros(max(nwrk, 1), dtype=self.dtype)
        ierr = C.c_int(0)
        if wdata is None:
            self.lib.ref_splcc(C.c_int(ndim), _ptr(xdata, rp), C.c_int(l1xdat), _ptr(ydata, rp),
                               C.c_int(ndata), _ptr(xmin, rp), _ptr(xmax, rp), _ptr(nodes, _ip),
                               self.rt(xtrap), _ptr(coef, rp), C.c_int(ncf), _ptr(work, rp),
                               C.c_int(nwrk), C.byref(ierr))
        else:
            self.lib.ref_splcw(C.c_int(ndim), _ptr(xdata, rp), C.c_int(l1xdat), _ptr(ydata, rp),
                               _ptr(wdata, rp), C.c_int(wdata.size), C.c_int(ndata),
                               _ptr(xmin, rp), _ptr(xmax, rp), _ptr(nodes, _ip), self.rt(xtrap),
                               _ptr(coef, rp), C.c_int(ncf), _ptr(work, rp), C.c_int(nwrk),
                               C.byref(ierr))
        return coef, ierr.value, work

    def evaluate(self, ndim, xq, nderiv, coef, xmin, xmax, nodes):
        """``nderiv=None`` -> splfe.  Returns (values, ierror of the last failing call)."""
        dt = self.dtype
        rp = self.rp
        xq = np.ascontiguousarray(xq, dtype=dt)
        if xq.ndim == 1:
            xq = xq.reshape(-1, 1)
        nq, ldx = xq.shape
        coef = np.ascontiguousarray(coef, dtype=dt).copy()
        xmin = np.ascontiguousarray(np.atleast_1d(xmin), dtype=dt)
        xmax = np.ascontiguousarray(np.atleast_1d(xmax), dtype=dt)
        nodes = np.ascontiguousarray(np.atleast_1d(nodes), dtype=np.int32)
        out = np.zeros(nq, dtype=dt)
        ierr = C.c_int(0)
        if nderiv is None:
            self.lib.ref_splfe_many(C.c_int(ndim), C.c_int(nq), _ptr(xq, rp), C.c_int(ldx),
                                    _ptr(coef, rp), C.c_int(coef.size), _ptr(xmin, rp),
                                    _ptr(xmax, rp), _ptr(nodes, _ip), _ptr(out, rp), C.byref(ierr))
        else:
            nd = np.ascontiguousarray(nderiv, dtype=np.int32)
            self.lib.ref_splde_many(C.c_int(ndim), C.c_int(nq), _ptr(xq, rp), C.c_int(ldx),
                                    _ptr(nd, _ip), _ptr(coef, rp), C.c_int(coef.size),
                                    _ptr(xmin, rp), _ptr(xmax, rp), _ptr(nodes, _ip),
                                    _ptr(out, rp), C.byref(ierr))
        return out, ierr.value


class Port(_Base):
    """The C restatement of the reference algorithm (oracle/splpak_oracle.c)."""

    kind = "port"

    def __init__(self):
        self.lib = C.CDLL(os.path.join(_HERE, "libsplpak_oracle.so"))
        L = self.lib
        L.oracle_splcw.restype = C.c_int
        L.oracle_splcw.argtypes = [C.c_int, _dp, C.c_int, _dp, _dp, C.c_int, _dp, _dp, _ip,
                                   C.c_double, _dp, C.c_int, _dp, C.c_long, C.c_int]
        L.oracle_splde.restype = C.c_double
        L.oracle_splde.argtypes = [C.c_int, _dp, _ip, _dp, _dp, _dp, _ip, _ip]
        L.oracle_splde_many.restype = C.c_int
        L.oracle_splde_many.argtypes = [C.c_int, C.c_long, _dp, C.c_int, _ip, _dp, _dp, _dp, _ip, _dp]
        L.oracle_last_reserr.restype = C.c_double
        L.oracle_last_reserr.argtypes = []
        L.oracle_splcw_banded.restype = C.c_int
        L.oracle_splcw_banded.argtypes = [C.c_int, _dp, C.c_int, _dp, _dp, C.c_int, _dp, _dp, _ip, C.c_double, _dp,
                                          C.c_int, C.c_int, _dp]
        L.oracle_rows_gradient.restype = C.c_int
        L.oracle_rows_gradient.argtypes = [C.c_int, _dp, C.c_int, _dp, _dp, C.c_int, _dp, _dp, _ip, C.c_double, _dp, C.c_int,
                                           C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_long)]
        L.oracle_bascmp.restype = C.c_double
        L.oracle_bascmp.argtypes = [C.c_int, _dp, _ip, _ip, _dp, _dp, _ip, _ip]

    def fit(self, ndim, xdata, ydata, wdata, xmin, xmax, nodes, xtrap, nwrk=None, ncf=None,
            ndata=None, l1xdat=None, quiet=True):
        """``wdata=None`` -> splcc semantics (wdata=[-1], src/splpak.F90:440)."""
        xdata, ydata, wdata, xmin, xmax, nodes = self._prep(ndim, xdata, ydata, wdata, xmin, xmax, nodes)
        if ndata is None:
            ndata = xdata.shape[0]
        if l1xdat is None:
            l1xdat = xdata.shape[1]
        ncol = int(np.prod(nodes[:max(ndim, 1)].astype(np.int64)))
        if ncf is None:
            ncf = ncol
        if nwrk is None:
            nwrk = ncol * (ncol + 1)
        coef = np.zeros(max(ncf, 1), dtype=np.float64)
        work = np.zeros(max(nwrk, 1), dtype=np.float64)
        if wdata is None:
            wdata = np.array([-1.0])
        ierr = self.lib.oracle_splcw(ndim, _ptr(xdata, _dp), l1xdat, _ptr(ydata, _dp),
                                     _ptr(wdata, _dp), ndata, _ptr(xmin, _dp), _ptr(xmax, _dp),
                                     _ptr(nodes, _ip), float(xtrap), _ptr(coef, _dp), ncf,
                                     _ptr(work, _dp), nwrk, 1 if quiet else 0)
        self.last_reserr = float(self.lib.oracle_last_reserr())
        return coef, ierr, work

    def fit_banded(self, ndim, xdata, ydata, wdata, xmin, xmax, nodes, xtrap, nthreads=0):
        """The reference's rows solved as banded normal equations + Cholesky + refinement on all host
        cores (oracle/splpak_banded.c): the "best CPU" comparator.  Returns (coef, ierror, info[10])."""
        xdata, ydata, wdata, xmin, xmax, nodes = self._prep(ndim, xdata, ydata, wdata, xmin, xmax, nodes)
        ncol = int(np.prod(nodes[:max(ndim, 1)].astype(np.int64)))
        coef = np.zeros(max(ncol, 1), dtype=np.float64)
        info = np.zeros(10)
        if wdata is None:
            wdata = np.array([-1.0])
        ierr = self.lib.oracle_splcw_banded(ndim, _ptr(xdata, _dp), xdata.shape[1], _ptr(ydata, _dp), _ptr(wdata, _dp),
                                            xdata.shape[0], _ptr(xmin, _dp), _ptr(xmax, _dp), _ptr(nodes, _ip),
                                            float(xtrap), _ptr(coef, _dp), ncol, int(nthreads), _ptr(info, _dp))
        return coef, ierr, info

    def rows_gradient(self, ndim, xdata, ydata, wdata, xmin, xmax, nodes, xtrap, coef, nthreads=0):
        """Optimality of GIVEN coefficients w.r.t. the reference's rows (data + constraint rows, generated as in
        fit_banded), computed on the host without storing the rows -> (omega, sqrt(ssq), data rows, constraint rows);
        omega = max_i |A^T (b - A x)|_i / (|A|^T (|A||x| + |b|))_i."""
        xdata = np.ascontiguousarray(xdata, dtype=np.float64)
        ydata = np.ascontiguousarray(ydata, dtype=np.float64)
        w = np.ascontiguousarray(wdata, dtype=np.float64) if wdata is not None else np.array([-1.0])
        xmin = np.ascontiguousarray(xmin, dtype=np.float64)
        xmax = np.ascontiguousarray(xmax, dtype=np.float64)
        nodes = np.ascontiguousarray(nodes, dtype=np.int32)
        coef = np.ascontiguousarray(coef, dtype=np.float64)
        om, s2 = C.c_double(0.0), C.c_double(0.0)
        nr = (C.c_long * 2)()
        rc = self.lib.oracle_rows_gradient(ndim, _ptr(xdata, _dp), xdata.shape[1], _ptr(ydata, _dp), _ptr(w, _dp), xdata.shape[0],
                                          _ptr(xmin, _dp), _ptr(xmax, _dp), _ptr(nodes, _ip), float(xtrap), _ptr(coef, _dp),
                                          int(nthreads), C.byref(om), C.byref(s2), nr)
        if rc != 0:
            raise MemoryError("oracle_rows_gradient")
        return om.value, float(np.sqrt(s2.value)), int(nr[0]), int(nr[1])

    def evaluate(self, ndim, xq, nderiv, coef, xmin, xmax, nodes):
        xq = np.ascontiguousarray(xq, dtype=np.float64)
        if xq.ndim == 1:
            xq = xq.reshape(-1, 1)
        nq, ldx = xq.shape
        coef = np.ascontiguousarray(coef, dtype=np.float64)
        xmin = np.ascontiguousarray(np.atleast_1d(xmin), dtype=np.float64)
        xmax = np.ascontiguousarray(np.atleast_1d(xmax), dtype=np.float64)
        nodes = np.ascontiguousarray(np.atleast_1d(nodes), dtype=np.int32)
        out = np.zeros(nq, dtype=np.float64)
        ndp = None
        if nderiv is not None:
            nd = np.ascontiguousarray(nderiv, dtype=np.int32)
            ndp = _ptr(nd, _ip)
        ierr = self.lib.oracle_splde_many(ndim, nq, _ptr(xq, _dp), ldx, ndp, _ptr(coef, _dp),
                                          _ptr(xmin, _dp), _ptr(xmax, _dp), _ptr(nodes, _ip),
                                          _ptr(out, _dp))
        return out, ierr
