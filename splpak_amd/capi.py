"""ctypes plumbing over the C ABI in ``include/splpak_hip.h``.

This is NOT a second implementation: every function forwards to
``libsplpak_hip.so`` (hand-written HIP for gfx950).  It exists so that the
pytest parity suite and ``bench.py`` can drive the same entry points the Fortran
``splpak_module`` binds.  Loading fails loudly when the library is missing, and
every compute entry point fails with ``SPLPAK_E_NODEVICE`` when there is no GPU;
there is no CPU fallback on the product path.
"""
from __future__ import annotations

import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPLPAK_LIB") or os.path.join(_HERE, "libsplpak_hip.so")      # SPLPAK_LIB: another build of the same library (A/B measurements)

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p)

# every symbol include/splpak_hip.h declares
SYMBOLS = [
    "splpak_fit_f64", "splpak_fit_f32", "splpak_eval_f64", "splpak_eval_f32",
    "splpak_plan_comm_len", "splpak_plan_create", "splpak_plan_destroy",
    "splpak_plan_set_allreduce", "splpak_plan_set_allreduce_ex", "splpak_plan_set_rccl", "splpak_rccl_unique_id", "splpak_rccl_comm_create",
    "splpak_rccl_comm_create_from_file", "splpak_rccl_comm_create_from_file_ex", "splpak_rccl_comm_destroy", "splpak_plan_set_refine", "splpak_plan_fit_dev",
    "splpak_plan_hist_dev", "splpak_plan_factorisation", "splpak_plan_enable_kernel_timing", "splpak_plan_kernel_timing", "splpak_plan_stage_timing",
    "splpak_eval_dev_f64", "splpak_eval_dev_f32", "splpak_eval_derivs_f64", "splpak_eval_derivs_f32", "splpak_eval_derivs_dev_f64",
    "splpak_synth_points_f64", "splpak_synth_queries_f64",
    "splpak_mplan_create", "splpak_mplan_destroy", "splpak_mplan_device", "splpak_mplan_rank_bytes", "splpak_mplan_factorisation", "splpak_mplan_fit_dev", "splpak_fit_multi_f64",
    "splpak_plan_device_bytes", "splpak_plan_pcg_stats", "splpak_set_default_option", "splpak_plan_set_option", "splpak_plan_get_option",
    "splpak_debug_spd_band_solve_f64", "splpak_debug_nd_tree", "splpak_debug_nd_partition", "splpak_debug_nd_schedule", "splpak_debug_window_values", "splpak_shutdown", "splpak_set_eval_mode",
    "splpak_last_error_message", "splpak_device_name",
]

E_NODEVICE, E_NOMEM, E_BADARG, E_UNSUPPORTED, E_COMM = -1, -2, -3, -4, -5
AR_ANY_POINTER = 1
AR_STREAM_ORDERED = 4

_lib = None


class SplpakError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load the HIP library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SplpakError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950); the splpak HIP path has no CPU fallback")
    # PyTorch-ROCm ships its own copy of the HIP runtime.  Two HIP runtimes in one process do not share
    # the device (whichever comes second finds "no GPUs"), so if torch is installed it is imported FIRST:
    # libsplpak_hip.so then binds to the runtime that is already loaded.  Without torch (Fortran callers,
    # plain ctypes users) the system runtime under /opt/rocm is used.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    L.splpak_fit_f64.restype = i32
    L.splpak_fit_f64.argtypes = [i32, _dp, i32, _dp, _dp, i64, _dp, _dp, _ip, dbl, _dp, i64, i64, _dp, _dp]
    L.splpak_fit_f32.restype = i32
    L.splpak_fit_f32.argtypes = [i32, _fp, i32, _fp, _fp, i64, _fp, _fp, _ip, C.c_float, _fp, i64, i64, _fp, _dp]
    L.splpak_eval_f64.restype = i32
    L.splpak_eval_f64.argtypes = [i32, i64, _dp, i32, _ip, _dp, _dp, _dp, _ip, _dp]
    L.splpak_eval_f32.restype = i32
    L.splpak_eval_f32.argtypes = [i32, i64, _fp, i32, _ip, _fp, _fp, _fp, _ip, _fp]
    L.splpak_plan_comm_len.restype = i64
    L.splpak_plan_comm_len.argtypes = [i32, _ip]
    L.splpak_plan_create.restype = i32
    L.splpak_plan_create.argtypes = [i32, _ip, _dp, _dp, dbl, i64, vp, i64, C.POINTER(vp)]
    L.splpak_plan_destroy.restype = None
    L.splpak_plan_destroy.argtypes = [vp]
    L.splpak_plan_set_allreduce.restype = None
    L.splpak_plan_set_allreduce.argtypes = [vp, ALLREDUCE_FN, vp, i32, i32]
    L.splpak_plan_set_allreduce_ex.restype = i32
    L.splpak_plan_set_allreduce_ex.argtypes = [vp, ALLREDUCE_FN, vp, i32, i32, i32]
    L.splpak_plan_set_rccl.restype = i32
    L.splpak_plan_set_rccl.argtypes = [vp, vp, i32, i32]
    L.splpak_rccl_unique_id.restype = i32
    L.splpak_rccl_unique_id.argtypes = [C.c_char_p]
    L.splpak_rccl_comm_create.restype = i32
    L.splpak_rccl_comm_create.argtypes = [C.c_char_p, i32, i32, C.POINTER(vp)]
    L.splpak_rccl_comm_create_from_file.restype = i32
    L.splpak_rccl_comm_create_from_file.argtypes = [C.c_char_p, i32, i32, dbl, C.POINTER(vp)]
    L.splpak_rccl_comm_create_from_file_ex.restype = i32
    L.splpak_rccl_comm_create_from_file_ex.argtypes = [C.c_char_p, C.c_char_p, i32, i32, dbl, C.POINTER(vp)]
    L.splpak_rccl_comm_destroy.restype = None
    L.splpak_rccl_comm_destroy.argtypes = [vp]
    L.splpak_plan_set_refine.restype = None
    L.splpak_plan_set_refine.argtypes = [vp, i32, dbl]
    L.splpak_plan_fit_dev.restype = i32
    L.splpak_plan_fit_dev.argtypes = [vp, vp, i32, vp, vp, i64, vp, vp, _dp]
    L.splpak_plan_hist_dev.restype = vp
    L.splpak_plan_hist_dev.argtypes = [vp]
    L.splpak_plan_factorisation.restype = i32
    L.splpak_plan_factorisation.argtypes = [vp, C.c_char_p, i32]
    L.splpak_plan_enable_kernel_timing.restype = None
    L.splpak_plan_enable_kernel_timing.argtypes = [vp, i32]
    L.splpak_plan_kernel_timing.restype = None
    L.splpak_plan_kernel_timing.argtypes = [vp, _dp]
    L.splpak_plan_stage_timing.restype = None
    L.splpak_plan_stage_timing.argtypes = [vp, _dp]
    L.splpak_eval_dev_f64.restype = i32
    L.splpak_eval_dev_f64.argtypes = [i32, i64, vp, i32, _ip, vp, _dp, _dp, _ip, vp, vp]
    L.splpak_eval_dev_f32.restype = i32
    L.splpak_eval_dev_f32.argtypes = [i32, i64, vp, i32, _ip, vp, _fp, _fp, _ip, vp, vp]
    L.splpak_eval_derivs_f64.restype = i32
    L.splpak_eval_derivs_f64.argtypes = [i32, i64, _dp, i32, i32, _dp, _dp, _dp, _ip, _dp, i32]
    L.splpak_eval_derivs_f32.restype = i32
    L.splpak_eval_derivs_f32.argtypes = [i32, i64, _fp, i32, i32, _fp, _fp, _fp, _ip, _fp, i32]
    L.splpak_eval_derivs_dev_f64.restype = i32
    L.splpak_eval_derivs_dev_f64.argtypes = [i32, i64, vp, i32, i32, vp, _dp, _dp, _ip, vp, i32, vp]
    L.splpak_synth_points_f64.restype = i32
    L.splpak_synth_points_f64.argtypes = [i32, i64, i64, vp, vp, vp, vp]
    L.splpak_synth_queries_f64.restype = i32
    L.splpak_synth_queries_f64.argtypes = [i32, i64, i64, i64, vp, vp]
    L.splpak_debug_spd_band_solve_f64.restype = i32
    L.splpak_debug_spd_band_solve_f64.argtypes = [i32, i32, _dp, _dp, _dp]
    L.splpak_debug_nd_tree.restype = i32
    L.splpak_debug_nd_tree.argtypes = [i32, _ip, i32, i32, _dp]
    L.splpak_debug_nd_schedule.restype = i32
    L.splpak_debug_nd_schedule.argtypes = [i32, _ip, i32, i32, i32, _dp]
    L.splpak_debug_nd_partition.restype = i32
    L.splpak_debug_nd_partition.argtypes = [i32, _ip, i32, i32, i32, _dp, _dp]
    L.splpak_debug_window_values.restype = i32
    L.splpak_debug_window_values.argtypes = [i32, dbl, dbl, i64, _dp, _ip, _dp, _dp, _ip]
    L.splpak_mplan_create.restype = i32
    L.splpak_mplan_create.argtypes = [i32, _ip, i32, i32, _ip, _dp, _dp, dbl, i64, C.POINTER(vp)]
    L.splpak_mplan_destroy.restype = None
    L.splpak_mplan_destroy.argtypes = [vp]
    L.splpak_mplan_device.restype = i32
    L.splpak_mplan_device.argtypes = [vp, i32]
    L.splpak_mplan_factorisation.restype = i32
    L.splpak_mplan_factorisation.argtypes = [vp, C.c_char_p, i32]
    L.splpak_mplan_rank_bytes.restype = i64
    L.splpak_mplan_rank_bytes.argtypes = [vp, i32]
    L.splpak_set_default_option.restype = i32
    L.splpak_set_default_option.argtypes = [C.c_char_p, C.c_char_p]
    L.splpak_plan_set_option.restype = i32
    L.splpak_plan_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.splpak_plan_get_option.restype = i32
    L.splpak_plan_get_option.argtypes = [vp, C.c_char_p, C.c_char_p, i32]
    L.splpak_plan_pcg_stats.restype = None
    L.splpak_plan_pcg_stats.argtypes = [vp, _dp]
    L.splpak_plan_device_bytes.restype = i64
    L.splpak_plan_device_bytes.argtypes = [vp]
    L.splpak_mplan_fit_dev.restype = i32
    L.splpak_mplan_fit_dev.argtypes = [vp, C.POINTER(vp), i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(i64), vp, _dp]
    L.splpak_fit_multi_f64.restype = i32
    L.splpak_fit_multi_f64.argtypes = [i32, i32, _dp, i32, _dp, _dp, i64, _dp, _dp, _ip, dbl, _dp, i64, i64, _dp, _dp]
    L.splpak_shutdown.restype = None
    L.splpak_shutdown.argtypes = []
    L.splpak_set_eval_mode.restype = i32
    L.splpak_set_eval_mode.argtypes = [i32, i64]
    L.splpak_last_error_message.restype = i32
    L.splpak_last_error_message.argtypes = [C.c_char_p, i32]
    L.splpak_device_name.restype = i32
    L.splpak_device_name.argtypes = [C.c_char_p, i32]
    _lib = L
    import atexit
    atexit.register(L.splpak_shutdown)
    return L


def last_error() -> str:
    buf = C.create_string_buffer(512)
    lib().splpak_last_error_message(buf, 512)
    return buf.value.decode(errors="replace")


def _check(rc: int) -> int:
    """Negative = infrastructure failure -> raise; >= 0 is the reference's ierror."""
    if rc < 0:
        raise SplpakError(f"splpak HIP library error {rc}: {last_error()}")
    return rc


def set_default_option(name, value) -> int:
    """Process-wide default of an option (what SPLPAK_<NAME> in the environment would set) for plans created afterwards;
    value None removes it.  Raises on an unknown name."""
    return _check(lib().splpak_set_default_option(name.encode(), None if value is None else str(value).encode()))


def shutdown() -> None:
    """Release the library's cached one-shot plan (28 GB at 64^3), staging buffers and evaluation scratch."""
    lib().splpak_shutdown()


def device_name() -> str:
    buf = C.create_string_buffer(256)
    _check(lib().splpak_device_name(buf, 256))
    return buf.value.decode()


def _p(a, ty):
    return None if a is None else a.ctypes.data_as(ty)


def _grid(ndim, xmin, xmax, nodes, dt=np.float64):
    xmin = np.ascontiguousarray(np.atleast_1d(xmin), dtype=dt)
    xmax = np.ascontiguousarray(np.atleast_1d(xmax), dtype=dt)
    nodes = np.ascontiguousarray(np.atleast_1d(nodes), dtype=np.int32)
    return xmin, xmax, nodes


# ---------------------------------------------------------------------------
# host-pointer entry points (what splpak_module binds)
# ---------------------------------------------------------------------------

def fit(ndim, xdata, ydata, wdata, xmin, xmax, nodes, xtrap, ncf=None, nwrk=-1, ndata=None,
        l1xdat=None, want_hist=False, real32=False):
    """splcw (wdata given) / splcc (wdata None).  -> (coef, ierror, hist|None, info)."""
    dt = np.float32 if real32 else np.float64
    rp = _fp if real32 else _dp
    xdata = np.ascontiguousarray(xdata, dtype=dt)
    if xdata.ndim == 1:
        xdata = xdata.reshape(-1, 1)
    ydata = np.ascontiguousarray(ydata, dtype=dt)
    if wdata is not None:
        wdata = np.ascontiguousarray(wdata, dtype=dt)
    xmin, xmax, nodes = _grid(ndim, xmin, xmax, nodes, dt)
    if ndata is None:
        ndata = xdata.shape[0]
    if l1xdat is None:
        l1xdat = xdata.shape[1]
    ncol = int(np.prod(np.maximum(nodes[:max(ndim, 1)].astype(np.int64), 1)))
    if ncf is None:
        ncf = ncol
    coef = np.zeros(max(ncf, 1), dtype=dt)
    hist = np.zeros(max(ncol, 1), dtype=dt) if want_hist else None
    info = np.zeros(10)
    fn = lib().splpak_fit_f32 if real32 else lib().splpak_fit_f64
    xt = C.c_float(xtrap) if real32 else C.c_double(xtrap)
    rc = _check(fn(ndim, _p(xdata, rp), l1xdat, _p(ydata, rp), _p(wdata, rp), ndata, _p(xmin, rp),
                   _p(xmax, rp), _p(nodes, _ip), xt, _p(coef, rp), ncf, nwrk, _p(hist, rp),
                   _p(info, _dp)))
    return coef, rc, hist, info


def fit_multi(ngpus, ndim, xdata, ydata, wdata, xmin, xmax, nodes, xtrap, want_hist=False):
    """splcw / splcc on `ngpus` GPUs of this node (distributed band, one process).  -> (coef, ierror, hist|None, info)."""
    xdata = np.ascontiguousarray(xdata, dtype=np.float64)
    if xdata.ndim == 1:
        xdata = xdata.reshape(-1, 1)
    ydata = np.ascontiguousarray(ydata, dtype=np.float64)
    if wdata is not None:
        wdata = np.ascontiguousarray(wdata, dtype=np.float64)
    xmin, xmax, nodes = _grid(ndim, xmin, xmax, nodes)
    ncol = int(np.prod(np.maximum(nodes[:max(ndim, 1)].astype(np.int64), 1)))
    coef = np.zeros(max(ncol, 1))
    hist = np.zeros(max(ncol, 1)) if want_hist else None
    info = np.zeros(10)
    rc = _check(lib().splpak_fit_multi_f64(int(ngpus), ndim, _p(xdata, _dp), xdata.shape[1], _p(ydata, _dp), _p(wdata, _dp),
                                           xdata.shape[0], _p(xmin, _dp), _p(xmax, _dp), _p(nodes, _ip), float(xtrap),
                                           _p(coef, _dp), ncol, -1, _p(hist, _dp), _p(info, _dp)))
    return coef, rc, hist, info


class MultiPlan:
    """A fit plan over several GPUs of this node (one process; torch tensors own the shards)."""

    def __init__(self, ngpus, ndim, nodes, xmin, xmax, xtrap, max_ndata_per_gpu, devices=None, chunk=0):
        self._L = lib()
        self.ngpus = int(ngpus)
        self.xmin, self.xmax, self.nodes = _grid(ndim, xmin, xmax, nodes)
        dv = None if devices is None else np.ascontiguousarray(devices, dtype=np.int32)
        h = C.c_void_p()
        rc = self._L.splpak_mplan_create(self.ngpus, _p(dv, _ip), int(chunk), ndim, _p(self.nodes, _ip), _p(self.xmin, _dp),
                                         _p(self.xmax, _dp), float(xtrap), int(max_ndata_per_gpu), C.byref(h))
        if rc != 0:
            if rc < 0:
                _check(rc)
            raise SplpakError(f"multi-GPU plan rejected with ierror {rc}")
        self._h = h

    def device(self, rank):
        return int(self._L.splpak_mplan_device(self._h, int(rank)))

    def factorisation(self):
        """-> (code, description): 3 distributed band, 5 distributed nested dissection (one rank: the single-GPU codes)."""
        buf = C.create_string_buffer(640)
        code = self._L.splpak_mplan_factorisation(self._h, buf, 640)
        return int(code), buf.value.decode()

    def rank_bytes(self, rank):
        """Device memory rank `rank` of the plan holds (bytes)."""
        return int(self._L.splpak_mplan_rank_bytes(self._h, int(rank)))

    def fit(self, xs, ys, ws, coef):
        """xs/ys/ws: lists (one entry per rank) of float64 device tensors on that rank's GPU (ws may be
        None); coef: float64 tensor on rank 0's GPU.  Blocks.  -> (ierror, info)."""
        R = self.ngpus
        vp = C.c_void_p
        ax = (vp * R)(*[vp(t.data_ptr()) for t in xs])
        ay = (vp * R)(*[vp(t.data_ptr()) for t in ys])
        aw = None if ws is None else (vp * R)(*[vp(t.data_ptr()) for t in ws])
        nn = (C.c_int64 * R)(*[int(t.shape[0]) for t in xs])
        info = np.zeros(10)
        rc = self._L.splpak_mplan_fit_dev(self._h, ax, int(xs[0].shape[1]), ay, aw, nn, vp(coef.data_ptr()), _p(info, _dp))
        return _check(rc), info

    def close(self):
        if getattr(self, "_h", None):
            self._L.splpak_mplan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def evaluate(ndim, xq, nderiv, coef, xmin, xmax, nodes, real32=False):
    """Batched splde (nderiv given) / splfe (nderiv None).  -> (values, ierror)."""
    dt = np.float32 if real32 else np.float64
    rp = _fp if real32 else _dp
    xq = np.ascontiguousarray(xq, dtype=dt)
    if xq.ndim == 1:
        xq = xq.reshape(-1, 1)
    nq, ldx = xq.shape
    coef = np.ascontiguousarray(coef, dtype=dt)
    xmin, xmax, nodes = _grid(ndim, xmin, xmax, nodes, dt)
    nd = None if nderiv is None else np.ascontiguousarray(nderiv, dtype=np.int32)
    out = np.zeros(nq, dtype=dt)
    fn = lib().splpak_eval_f32 if real32 else lib().splpak_eval_f64
    rc = _check(fn(ndim, nq, _p(xq, rp), ldx, _p(nd, _ip), _p(coef, rp), _p(xmin, rp), _p(xmax, rp),
                   _p(nodes, _ip), _p(out, rp)))
    return out, rc


def derivs_nout(ndim, order):
    return 1 + ndim + (ndim * (ndim + 1) // 2 if order == 2 else 0)


def evaluate_derivs(ndim, xq, order, coef, xmin, xmax, nodes, real32=False):
    """Value, gradient (order 1) and Hessian upper triangle (order 2) per query -> (array (nq, nout), ierror)."""
    dt = np.float32 if real32 else np.float64
    rp = _fp if real32 else _dp
    xq = np.ascontiguousarray(xq, dtype=dt)
    if xq.ndim == 1:
        xq = xq.reshape(-1, 1)
    nq, ldx = xq.shape
    coef = np.ascontiguousarray(coef, dtype=dt)
    xmin, xmax, nodes = _grid(ndim, xmin, xmax, nodes, dt)
    nout = derivs_nout(max(ndim, 1), order) if order in (1, 2) else 1
    out = np.zeros((nq, nout), dtype=dt)
    fn = lib().splpak_eval_derivs_f32 if real32 else lib().splpak_eval_derivs_f64
    rc = _check(fn(ndim, nq, _p(xq, rp), ldx, int(order), _p(coef, rp), _p(xmin, rp), _p(xmax, rp),
                   _p(nodes, _ip), _p(out, rp), nout))
    return out, rc


def evaluate_derivs_dev(ndim, xq, order, coef, xmin, xmax, nodes, out, stream=0):
    """Same on torch device tensors; `out` is (nq, >= nout) float64."""
    xmin, xmax, nodes = _grid(ndim, xmin, xmax, nodes)
    nq, ldx = xq.shape
    return _check(lib().splpak_eval_derivs_dev_f64(ndim, int(nq), xq.data_ptr(), int(ldx), int(order),
                                                   coef.data_ptr(), _p(xmin, _dp), _p(xmax, _dp), _p(nodes, _ip),
                                                   out.data_ptr(), int(out.shape[1]), C.c_void_p(stream)))


# ---------------------------------------------------------------------------
# resident-data API (torch tensors own the device memory)
# ---------------------------------------------------------------------------

class Plan:
    """A fit plan for one node grid on the current torch CUDA(HIP) device."""

    def __init__(self, ndim, nodes, xmin, xmax, xtrap, max_ndata, comm=None):
        self._L = lib()
        self.ndim = ndim
        self.xmin, self.xmax, self.nodes = _grid(ndim, xmin, xmax, nodes)
        self.ncol = int(np.prod(self.nodes.astype(np.int64)))
        self.comm_len = int(self._L.splpak_plan_comm_len(ndim, _p(self.nodes, _ip)))
        self.comm = comm          # optional torch tensor (float64, device) of >= comm_len elements
        self._cb = None
        h = C.c_void_p()
        comm_ptr = None if comm is None else C.c_void_p(comm.data_ptr())
        comm_n = 0 if comm is None else comm.numel()
        rc = self._L.splpak_plan_create(ndim, _p(self.nodes, _ip), _p(self.xmin, _dp),
                                        _p(self.xmax, _dp), float(xtrap), int(max_ndata), comm_ptr,
                                        comm_n, C.byref(h))
        if rc != 0:
            if rc < 0:
                _check(rc)
            raise SplpakError(f"plan rejected with ierror {rc}")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.splpak_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_refine(self, max_steps, tol):
        self._L.splpak_plan_set_refine(self._h, int(max_steps), float(tol))

    def set_allreduce(self, fn, rank, world, any_pointer=None, stream_ordered=False):
        """fn(offset_elems, count) must sum-all-reduce self.comm[offset:offset+count] in place.  A hook that also takes a
        third argument -- fn(-1, count, view) with `view` a device tensor over the library's own memory -- declares
        SPLPAK_AR_ANY_POINTER (include/splpak_hip.h): the nested-dissection factorisation of the sharded fit is then
        distributed by subtrees and the fronts they report into arrive that way.  A two-argument hook (rounds 1-2) keeps
        the replicated factorisation.  `any_pointer` overrides the detection.  `stream_ordered`: the hook only ENQUEUES its
        work on the current stream (= the library's, made current for the call) -- SPLPAK_AR_STREAM_ORDERED: the library then does
        not synchronise the stream before and after the call."""
        if any_pointer is None:
            import inspect
            try:
                params = list(inspect.signature(fn).parameters.values())
                any_pointer = len(params) >= 3 or any(q.kind == q.VAR_POSITIONAL for q in params)
            except (TypeError, ValueError):
                any_pointer = False
        base = self.comm.data_ptr()
        ncomm = self.comm.numel()
        device = self.comm.device

        class _Raw:            # a device buffer of the library as a CUDA-array-interface object
            def __init__(self, ptr, n):
                self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}

        def _call(ptr, count):
            off = (int(ptr) - base) // 8
            if 0 <= off and off + int(count) <= ncomm and (int(ptr) - base) % 8 == 0:
                fn(off, int(count))
            else:
                import torch
                fn(-1, int(count), torch.as_tensor(_Raw(ptr, count), device=device))

        def _cb(ptr, count, stream, user):
            # The C ABI's contract: the reduction is ordered ON `stream` (the stream the fit was
            # enqueued on).  torch.distributed orders a collective against torch's CURRENT stream, so
            # make the library's stream current for the duration of the call.
            try:
                import torch
                if torch.cuda.is_available():
                    with torch.cuda.stream(torch.cuda.ExternalStream(int(stream or 0))):
                        _call(ptr, count)
                else:
                    _call(ptr, count)
                return 0
            except Exception as exc:  # pragma: no cover - surfaced as SPLPAK_E_COMM
                print("all-reduce callback failed:", exc, flush=True)
                return 1

        self._cb = ALLREDUCE_FN(_cb)
        _check(self._L.splpak_plan_set_allreduce_ex(self._h, self._cb, None, int(rank), int(world),
                                                    (AR_ANY_POINTER if any_pointer else 0) | (AR_STREAM_ORDERED if stream_ordered else 0)))

    def device_bytes(self):
        return int(self._L.splpak_plan_device_bytes(self._h))

    def set_option(self, name, value):
        """A per-fit option of this plan (value as text, None removes it); options that shape the plan raise: set_default_option."""
        return _check(self._L.splpak_plan_set_option(self._h, name.encode(), None if value is None else str(value).encode()))

    def get_option(self, name):
        buf = C.create_string_buffer(256)
        return buf.value.decode() if _check(self._L.splpak_plan_get_option(self._h, name.encode(), buf, 256)) == 1 else None

    def pcg_stats(self):
        """Iterative solve of the last fit: dict(iterations, solves, last_iterations, last_residual, rho, lam); zeros without it."""
        out = np.zeros(6)
        self._L.splpak_plan_pcg_stats(self._h, _p(out, _dp))
        return dict(iterations=int(out[0]), solves=int(out[1]), last_iterations=int(out[2]), last_residual=float(out[3]),
                    rho=float(out[4]), lam=float(out[5]))

    def set_rccl(self, comm, rank, world):
        """The library's own RCCL hook (no Python in the reductions): `comm` is an ncclComm_t as an integer / c_void_p."""
        _check(self._L.splpak_plan_set_rccl(self._h, C.c_void_p(comm if isinstance(comm, int) else comm.value), int(rank), int(world)))

    def factorisation(self):
        """-> (code, description): 0/1 band Cholesky, 2 two-ended band, 3 distributed band, 4 nested dissection."""
        buf = C.create_string_buffer(640)
        code = self._L.splpak_plan_factorisation(self._h, buf, 640)
        return int(code), buf.value.decode()

    def enable_kernel_timing(self, on=True):
        self._L.splpak_plan_enable_kernel_timing(self._h, 1 if on else 0)

    def kernel_timing(self):
        out = np.zeros(7)
        self._L.splpak_plan_kernel_timing(self._h, _p(out, _dp))
        return dict(syrk_launches=out[0], syrk_ms=out[1], syrk_flop=out[2], factor_ms=out[3], total_flop=out[4],
                    bulk_launches=out[5], bulk_flop=out[6])

    def stage_timing(self):
        out = np.zeros(6)
        self._L.splpak_plan_stage_timing(self._h, _p(out, _dp))
        return dict(bin_ms=out[0], gram_ms=out[1], constraints_ms=out[2], expand_ms=out[3], residual_pass_ms=out[4], solve_ms=out[5])

    def fit(self, xdata, ydata, wdata, coef, stream=0):
        """All arguments are torch float64 device tensors; xdata is (ndata, l1xdat) row-major
        (= the reference's column-major xdata(l1xdat, ndata)).  Returns (ierror, info)."""
        info = np.zeros(10)
        ndata, l1 = xdata.shape
        rc = self._L.splpak_plan_fit_dev(self._h, xdata.data_ptr(), int(l1), ydata.data_ptr(),
                                         None if wdata is None else wdata.data_ptr(), int(ndata),
                                         coef.data_ptr(), C.c_void_p(stream), _p(info, _dp))
        return _check(rc), info

    def hist_ptr(self):
        return self._L.splpak_plan_hist_dev(self._h)


def evaluate_dev(ndim, xq, nderiv, coef, xmin, xmax, nodes, out, stream=0):
    """Batched evaluation on torch device tensors (asynchronous on `stream`); float32 tensors take the REAL32 entry."""
    nd = None if nderiv is None else np.ascontiguousarray(nderiv, dtype=np.int32)
    nq, ldx = xq.shape
    if str(xq.dtype).endswith("float32"):
        xmin, xmax, nodes = _grid(ndim, xmin, xmax, nodes, np.float32)
        return _check(lib().splpak_eval_dev_f32(ndim, int(nq), xq.data_ptr(), int(ldx), _p(nd, _ip),
                                                coef.data_ptr(), _p(xmin, _fp), _p(xmax, _fp),
                                                _p(nodes, _ip), out.data_ptr(), C.c_void_p(stream)))
    xmin, xmax, nodes = _grid(ndim, xmin, xmax, nodes)
    return _check(lib().splpak_eval_dev_f64(ndim, int(nq), xq.data_ptr(), int(ldx), _p(nd, _ip),
                                            coef.data_ptr(), _p(xmin, _dp), _p(xmax, _dp),
                                            _p(nodes, _ip), out.data_ptr(), C.c_void_p(stream)))


EVAL_AUTO, EVAL_DIRECT, EVAL_BINNED = 0, 1, 2


def set_eval_mode(mode=EVAL_AUTO, chunk=0):
    """Evaluation strategy of this thread (bit-identical results): auto / direct gathers / LDS-binned."""
    return _check(lib().splpak_set_eval_mode(int(mode), int(chunk)))


def synth_points_dev(ndim, first_point, ndata, xdata, ydata, wdata, stream=0):
    return _check(lib().splpak_synth_points_f64(ndim, int(first_point), int(ndata),
                                                None if xdata is None else xdata.data_ptr(),
                                                None if ydata is None else ydata.data_ptr(),
                                                None if wdata is None else wdata.data_ptr(),
                                                C.c_void_p(stream)))


def synth_queries_dev(ndim, ndata_before, first_query, nq, xq, stream=0):
    return _check(lib().splpak_synth_queries_f64(ndim, int(ndata_before), int(first_query), int(nq),
                                                 xq.data_ptr(), C.c_void_p(stream)))


def rccl_comm_create_from_file(path, rank, world, timeout_s=60.0, job=None):
    """ncclComm_t (as an int) made by the library: rank 0 draws the id and publishes it through `path` together with the
    job's tag (`job`: any string the ranks of ONE run share; None = SPLPAK_RCCL_JOB or what the launcher exports), the others
    wait for a file of THEIR job -- a file left by another run is ignored (include/splpak_hip.h)."""
    comm = C.c_void_p()
    rc = lib().splpak_rccl_comm_create_from_file_ex(str(path).encode(), None if job is None else str(job).encode(), int(rank), int(world),
                                                    float(timeout_s), C.byref(comm))
    if rc != 0:
        raise SplpakError(f"splpak_rccl_comm_create_from_file: {rc}: {last_error()}")
    return comm.value


def rccl_unique_id():
    """128-byte ncclUniqueId drawn by the library's RCCL (rank 0; hand it to the other ranks by any means)."""
    buf = C.create_string_buffer(128)
    rc = lib().splpak_rccl_unique_id(buf)
    if rc != 0:
        raise SplpakError(f"splpak_rccl_unique_id: {rc}: {last_error()}")
    return buf.raw


def rccl_comm_create(id128, rank, world):
    """ncclComm_t (as an int) on the CURRENT device from an id all ranks share."""
    comm = C.c_void_p()
    rc = lib().splpak_rccl_comm_create(bytes(id128), int(rank), int(world), C.byref(comm))
    if rc != 0:
        raise SplpakError(f"splpak_rccl_comm_create: {rc}: {last_error()}")
    return comm.value


def rccl_comm_destroy(comm):
    lib().splpak_rccl_comm_destroy(C.c_void_p(comm))


ND_TREE_FIELDS = ("fronts", "depth", "factor_bytes", "arena_bytes", "schur_bytes_per_fit", "flop", "flop_exact",
                  "max_separator", "max_border", "diag_blocks", "vec_len", "own_rows", "border_rows", "inverse_bytes")


def debug_nd_tree(nodes, split_min=0, check=True):
    """Host-only: size of the nested-dissection elimination tree the fit uses for large 3-D / 4-D grids
    (csrc/ndtree.hpp), optionally with its invariants verified.  -> dict (ND_TREE_FIELDS)."""
    nodes = np.ascontiguousarray(np.atleast_1d(nodes), dtype=np.int32)
    out = np.zeros(16)
    rc = _check(lib().splpak_debug_nd_tree(len(nodes), _p(nodes, _ip), int(split_min), 1 if check else 0, _p(out, _dp)))
    if rc != 0:
        raise SplpakError(f"grid rejected: {rc}")
    return dict(zip(ND_TREE_FIELDS, out.tolist()))


def debug_nd_schedule(nodes, cut=0, packed=True, split_min=0):
    """Host-only: the elimination schedule of the nested-dissection factorisation for a given cut depth (csrc/ndtree.hpp
    NdSchedule), its invariants verified.  -> dict(stages, arena_bytes, schur_bytes_per_fit, factor_bytes, cut, depth)."""
    nodes = np.ascontiguousarray(np.atleast_1d(nodes), dtype=np.int32)
    out = np.zeros(8)
    rc = _check(lib().splpak_debug_nd_schedule(len(nodes), _p(nodes, _ip), int(split_min), int(cut), 1 if packed else 0, _p(out, _dp)))
    if rc != 0:
        raise SplpakError(f"schedule rejected: {rc}: {last_error()}")
    return dict(stages=int(out[0]), arena_bytes=out[1], schur_bytes_per_fit=out[2], factor_bytes=out[3], cut=int(out[4]), depth=int(out[5]))


ND_RANK_FIELDS = ("bytes", "panel_bytes", "schur_bytes", "top_bytes", "inverse_bytes", "other_bytes", "flop_subtrees", "flop_top")


def debug_nd_partition(nodes, ngpus, chunk=0, split_min=0):
    """Host-only: how the one-process multi-GPU fit distributes the nested-dissection factorisation of a grid over
    `ngpus` GPUs (csrc/ndtree.hpp NdPartition).  -> (list of per-rank dicts (ND_RANK_FIELDS), summary dict)."""
    nodes = np.ascontiguousarray(np.atleast_1d(nodes), dtype=np.int32)
    per = np.zeros(8 * int(ngpus))
    out = np.zeros(8)
    rc = _check(lib().splpak_debug_nd_partition(len(nodes), _p(nodes, _ip), int(split_min), int(ngpus), int(chunk), _p(per, _dp), _p(out, _dp)))
    if rc != 0:
        raise SplpakError(f"grid rejected: {rc}")
    ranks = [dict(zip(ND_RANK_FIELDS, per[8 * r:8 * r + 8].tolist())) for r in range(int(ngpus))]
    summ = dict(dcut=int(out[0]), top_fronts=int(out[1]), top_steps=int(out[2]), max_panel_bytes=out[3], normal_eq_bytes=out[4],
                fronts=int(out[5]), depth=int(out[6]), flop=out[7])
    return ranks, summ


def debug_window_values(nodes, xmin, xmax, x):
    """Host-only: the 4 basis values of the window of every x on a 1-D grid, as the evaluation kernels select the form
    (0 interior closed form / 1 next to an end / 2 general) and in the general form.  -> (ws, used[n,4], general[n,4], form)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    n = x.size
    ws = np.zeros(n, dtype=np.int32)
    form = np.zeros(n, dtype=np.int32)
    used = np.zeros((n, 4))
    gen = np.zeros((n, 4))
    rc = _check(lib().splpak_debug_window_values(int(nodes), float(xmin), float(xmax), int(n), _p(x, _dp),
                                                 _p(ws, _ip), _p(used, _dp), _p(gen, _dp), _p(form, _ip)))
    if rc != 0:
        raise SplpakError(f"grid rejected: {rc}")
    return ws, used, gen, form


def debug_spd_band_solve(a_lower, halfbw, b):
    """Solve with the library's band Cholesky (diagnostics).  a_lower: (n, n) array, lower part used."""
    a = np.asfortranarray(a_lower, dtype=np.float64)
    n = a.shape[0]
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros(n)
    rc = _check(lib().splpak_debug_spd_band_solve_f64(n, int(halfbw), _p(a, _dp), _p(b, _dp), _p(x, _dp)))
    return x, rc
