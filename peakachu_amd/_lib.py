"""ctypes binding of libpeakachu_hip.so (include/peakachu_hip.h).

The library is built in-tree by `make -C peakachu_amd/csrc` (or
__graft_entry__.build()).  There is no CPU fallback: if the shared object is
missing, or no gfx950 device is visible when a compute call is made, the
call raises.
"""
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PEAKACHU_HIP_LIB: another build of the same library (tools/ab.sh compares builds on one box
# without copying anything over the product's file)
LIB_PATH = os.environ.get("PEAKACHU_HIP_LIB") or os.path.join(_HERE, "libpeakachu_hip.so")
CSRC = os.path.join(_HERE, "csrc")

_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
_u16p = np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_vp = C.c_void_p

# name -> (restype, argtypes); every symbol include/peakachu_hip.h declares
SIGNATURES = {
    "pk_abi_version": (C.c_int, []),
    "pk_last_error": (C.c_char_p, []),
    "pk_device_count": (C.c_int, []),
    "pk_device_name": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "pk_device_synchronize": (C.c_int, [C.c_int]),
    "pk_runtime_versions": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pk_forest_create": (_vp, [C.c_int, C.c_int, C.c_int, _i32p, _i32p, _i32p, _i32p, _f64p,
                               _u8p, _f64p]),
    "pk_forest_destroy": (None, [_vp]),
    "pk_forest_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                 C.POINTER(C.c_int64), C.POINTER(C.c_int)]),
    "pk_matrix_create": (_vp, [C.c_int, C.c_int32, _i32p, _i32p, _f64p, _f64p, C.c_int32,
                               C.c_int32, C.c_int32]),
    "pk_matrix_destroy": (None, [_vp]),
    "pk_csr_upload": (_vp, [C.c_int, C.c_int32, _i32p, _i32p, _f64p]),
    "pk_csr_upload_upper": (_vp, [C.c_int, C.c_int32, _i32p, _i32p, _vp, C.c_int, _vp]),
    "pk_csr_view": (_vp, [_vp, _vp]),
    "pk_csr_destroy": (None, [_vp]),
    "pk_csr_info": (C.c_int, [_vp, _i64p, C.POINTER(C.c_double)]),
    "pk_matrix_from_csr": (_vp, [_vp, C.c_int32, C.c_int32, C.c_int]),
    "pk_matrix_set_expected": (C.c_int, [_vp, _f64p, C.c_int32]),
    "pk_csr_expected_means": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _f64p]),
    "pk_extract": (C.c_int, [_vp, C.c_int, C.c_int64, _i32p, _i32p, _vp, _vp, _i64p,
                             C.POINTER(C.c_int64)]),
    "pk_predict": (C.c_int, [_vp, C.c_int64, _f32p, _f64p]),
    "pk_cands_create": (_vp, [C.c_int, C.c_int64, _i32p, _i32p]),
    "pk_cands_destroy": (None, [_vp]),
    "pk_cands_set_prune": (C.c_int, [_vp, C.c_int]),
    "pk_score_run": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_double, C.c_int64,
                               C.POINTER(C.c_int64)]),
    "pk_score_fetch": (C.c_int, [_vp, _i32p, _i32p, _f64p, _f64p]),
    "pk_score_fetch_all": (C.c_int, [_vp, _u8p, _f64p]),
    "pk_score": (C.c_int, [_vp, _vp, C.c_int, C.c_double, C.c_int64, C.c_int64, _i32p, _i32p,
                           _i32p, _i32p, _f64p, _f64p, C.POINTER(C.c_int64)]),
    "pk_expected_means": (C.c_int, [_vp, C.c_int, _u8p, _f64p]),
    "pk_candidates_create": (_vp, [_vp, C.c_int, C.c_int, _vp, _f64p, _vp, _vp, C.c_int64,
                                   C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "pk_cands_fetch": (C.c_int, [_vp, _i32p, _i32p]),
    "pk_set_gauss_taps": (C.c_int, [_f64p]),
    "pk_get_gauss_taps": (C.c_int, [_f64p]),
    "pk_set_option": (C.c_int, [C.c_char_p, C.c_int64]),
    "pk_get_option": (C.c_int64, [C.c_char_p]),
    "pk_forest_set_option": (C.c_int, [_vp, C.c_char_p, C.c_int64]),
    "pk_forest_get_option": (C.c_int64, [_vp, C.c_char_p]),
    "pk_matrix_set_option": (C.c_int, [_vp, C.c_char_p, C.c_int64]),
    "pk_matrix_get_option": (C.c_int64, [_vp, C.c_char_p]),
    "pk_cands_set_option": (C.c_int, [_vp, C.c_char_p, C.c_int64]),
    "pk_cands_get_option": (C.c_int64, [_vp, C.c_char_p]),
    "pk_prof_enable": (C.c_int, [C.c_int]),
    "pk_prof_reset": (C.c_int, []),
    "pk_prof_get": (C.c_int, [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "pk_debug_read": (C.c_int, [C.c_int, _i64p, C.c_int64]),
    "pk_debug_lock_probe": (C.c_int, [C.c_int, C.c_int]),
    "pk_debug_forest_image": (C.c_int, [C.c_int, C.c_int, _i32p, _i32p, _i32p, _i32p, _f64p, _u8p,
                                        _f64p, C.c_int, _i32p, C.c_int64, _u64p,
                                        C.POINTER(C.c_int64), C.c_int64, _i32p,
                                        C.POINTER(C.c_int32), _u64p, _i32p]),
    "pk_debug_forest_qimage": (C.c_int, [C.c_int, C.c_int, _i32p, _i32p, _i32p, _i32p, _f64p, _u8p,
                                         _f64p, C.c_int, C.c_int, _i32p, _i32p, C.c_int64, _f32p,
                                         _u32p, _f32p, C.c_int64, _u64p, C.POINTER(C.c_int64),
                                         C.c_int64, _i32p, C.POINTER(C.c_int32), _i32p, C.c_int32, _i32p]),
    "pk_debug_prune_bound": (C.c_double, [C.c_double, C.c_int, C.c_int64]),
    "pk_debug_classify_coords": (C.c_int, [C.c_int64, _i32p, _i32p]),
    "pk_debug_cut_policy": (C.c_int, [_i32p, C.c_int, C.c_double, C.c_int, _f64p, C.c_int, _i32p]),
    "pk_host_unfilter_chunks": (C.c_int, [C.c_int, _vp, _i64p, C.c_int, C.c_int, C.c_int64, _i64p, _i64p, _vp,
                                          C.c_int]),
    "pk_comm_unique_id": (C.c_int, [_u8p]),
    "pk_comm_create": (_vp, [C.c_int, C.c_int, C.c_int, _u8p]),
    "pk_comm_destroy": (None, [_vp]),
    "pk_comm_ranks": (C.c_int, [_vp]),
    "pk_comm_version": (C.c_int, [C.POINTER(C.c_int)]),
    "pk_comm_gather_scored": (C.c_int, [_vp, _vp, _i64p, C.c_int64, _vp, _vp, _vp, _vp]),
    "pk_comm_gatherv_bytes": (C.c_int, [_vp, _vp, C.c_int64, _i64p, _vp, C.c_int64]),
}

_LIB = None


class PeakachuHipError(RuntimeError):
    pass


def load():
    """Load the C-ABI library (once).  Raises if it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise PeakachuHipError(
            "%s not found: build it with `make -C %s` (hipcc, gfx950). "
            "peakachu_amd has no CPU fallback." % (LIB_PATH, CSRC))
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError if the header and the .so disagree
        fn.restype = res
        fn.argtypes = args
    if L.pk_abi_version() != 1:
        raise PeakachuHipError("ABI version mismatch: %d" % L.pk_abi_version())
    # the blur's taps as THIS interpreter's numpy evaluates them (the reference gets them from
    # scipy.ndimage's _gaussian_kernel1d, same expression): their last bits vary with numpy
    rc = L.pk_set_gauss_taps(gauss_taps())
    if rc != 0:
        raise PeakachuHipError("pk_set_gauss_taps failed: %s" % L.pk_last_error().decode("utf-8", "replace"))
    _LIB = L
    return L


def gauss_taps(sigma=1.0, radius=4):
    """Centre tap and the taps of one side of scipy.ndimage's Gaussian kernel (sigma 1, truncate
    4): `phi = exp(-0.5 / sigma**2 * x**2); phi /= phi.sum()` -- _gaussian_kernel1d, unchanged
    between scipy 1.7 and 1.15 -- evaluated by the numpy at hand."""
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    phi = phi / phi.sum()
    if not np.array_equal(phi, phi[::-1]):
        raise PeakachuHipError("numpy's exp gave an asymmetric Gaussian kernel")  # the kernels pair the taps
    return np.ascontiguousarray(phi[radius:], np.float64)


def last_error():
    return load().pk_last_error().decode("utf-8", "replace")


# error codes of include/peakachu_hip.h
PK_OK, PK_E_INVALID, PK_E_NODEVICE, PK_E_HIP, PK_E_NOMEM, PK_E_UNSUPPORTED, PK_E_COMM = 0, -1, -2, -3, -4, -5, -6


def check(rc, what):
    if rc != 0:
        raise PeakachuHipError("%s failed (%d): %s" % (what, rc, last_error()))


def require_device():
    L = load()
    if L.pk_device_count() < 1:
        raise PeakachuHipError("no HIP device visible; peakachu_amd needs an MI355X (gfx950) "
                               "and has no CPU fallback")
    return L


ROCM_SONAMES = ("librccl.so", "libamdhip64.so", "libhsa-runtime64.so")


def mapped_rocm_libs():
    """Paths of the RCCL / HIP / HSA runtimes mapped into THIS process (/proc/self/maps), per
    soname stem.  A soname is bound once per process: whoever maps a ROCm first -- e.g. the copy
    a deep-learning framework bundles under the same sonames -- decides which one this library runs on."""
    found = {k: [] for k in ROCM_SONAMES}
    try:
        with open("/proc/self/maps") as fh:
            for line in fh:
                path = line.rstrip("\n").split(None, 5)[-1] if line.count("/") else ""
                base = os.path.basename(path)
                for k in ROCM_SONAMES:
                    if base.startswith(k) and path not in found[k]:
                        found[k].append(path)
    except OSError:
        pass
    return found


def expected_rocm_dir():
    """The directory the loader resolves this library's ROCm dependencies from in a fresh process
    (what `ldd libpeakachu_hip.so` names): read from the dynamic loader itself, in a child."""
    import subprocess
    try:
        out = subprocess.run(["ldd", LIB_PATH], capture_output=True, text=True, timeout=30).stdout
    except Exception:
        return None
    for line in out.splitlines():
        if "libamdhip64.so" in line and "=>" in line:
            path = line.split("=>", 1)[1].split("(")[0].strip()
            if path and os.path.exists(path):
                return os.path.dirname(os.path.realpath(path))
    # (no usable ldd: where the Makefile links against)
    cand = os.path.realpath(os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib"))
    return cand if os.path.isdir(cand) else None


def runtime_info():
    """What bench.py stamps into its line: the versions the bound runtimes report, the paths they
    were mapped from, and whether all of them lie where `ldd` resolves them (`product_runtime`)."""
    L = load()
    hr, hd, rv = C.c_int(-1), C.c_int(-1), C.c_int(-1)
    L.pk_runtime_versions(C.byref(hr), C.byref(hd))
    L.pk_comm_version(C.byref(rv))
    libs = mapped_rocm_libs()
    want = expected_rocm_dir()
    paths = [p for v in libs.values() for p in v]
    # (nothing to compare with -- no ldd, no /opt/rocm --: None = unknown, which refuses nothing)
    ok = None if not (want and paths) else (all(len(v) <= 1 for v in libs.values()) and
                                            all(os.path.dirname(os.path.realpath(p)) == want for p in paths))
    return {"hip_runtime_version": hr.value, "hip_driver_version": hd.value, "rccl_version": rv.value,
            "rocm_libs": {k: (v[0] if len(v) == 1 else v) for k, v in libs.items()},
            "rocm_dir_ldd": want, "product_runtime": ok}


# ------------------------------------------------------------------ handles
class _Options:
    """Per-handle options (include/peakachu_hip.h: every handle carries its own set; pk_set_option
    only changes the defaults of handles created afterwards).  `options` may hold names of any
    handle kind: each handle takes the ones that concern it (forest_* -> forests, extract_* ->
    matrices, chunk / overlap / sub_chunk / early_exit -> candidate lists and, for the calls
    without a list, matrices)."""
    _KIND = None  # "forest" | "matrix" | "cands"

    @staticmethod
    def concerns(kind, name):
        if name.startswith("forest_"):
            return kind == "forest" or (name == "forest_lds" and kind == "matrix")
        if name.startswith("extract_"):
            return kind == "matrix"
        if name == "chunk":
            return True
        return kind in ("cands", "matrix")  # overlap, sub_chunk, early_exit

    def set_option(self, name, value):
        check(getattr(self._L, "pk_%s_set_option" % self._KIND)(self.h, name.encode(), int(value)),
              "pk_%s_set_option(%s)" % (self._KIND, name))
        return self

    def get_option(self, name):
        return getattr(self._L, "pk_%s_get_option" % self._KIND)(self.h, name.encode())

    def set_options(self, options):
        for k, v in (options or {}).items():
            if self.concerns(self._KIND, k):
                self.set_option(k, v)
        return self


class HipForest(_Options):
    """Device-resident forest (pk_forest)."""
    _KIND = "forest"

    def __init__(self, flat, device=0, options=None):
        L = require_device()
        self._L = L
        self.T, self.F = int(flat.T), int(flat.F)
        self.h = L.pk_forest_create(
            device, self.T, self.F,
            np.ascontiguousarray(flat.tree_off, np.int32),
            np.ascontiguousarray(flat.left, np.int32),
            np.ascontiguousarray(flat.right, np.int32),
            np.ascontiguousarray(flat.feat, np.int32),
            np.ascontiguousarray(flat.thr, np.float64),
            np.ascontiguousarray(flat.miss_left, np.uint8),
            np.ascontiguousarray(flat.p1, np.float64))
        if not self.h:
            raise PeakachuHipError("pk_forest_create: " + last_error())
        self.device = device
        self.set_options(options)

    def info(self):
        T, F, d = C.c_int(), C.c_int(), C.c_int()
        nn = C.c_int64()
        check(self._L.pk_forest_info(self.h, C.byref(T), C.byref(F), C.byref(nn), C.byref(d)),
              "pk_forest_info")
        return dict(T=T.value, F=F.value, n_nodes=nn.value, max_depth=d.value)

    def predict(self, fea32):
        fea32 = np.ascontiguousarray(fea32, np.float32)
        if fea32.ndim != 2 or fea32.shape[1] != self.F:
            raise ValueError("features must be [N, %d]" % self.F)
        N = fea32.shape[0]
        out = np.empty(max(N, 1), np.float64)
        check(self._L.pk_predict(self.h, N, fea32.reshape(-1) if N else np.zeros(1, np.float32),
                                 out), "pk_predict")
        return out[:N]

    def close(self):
        if getattr(self, "h", None):
            self._L.pk_forest_destroy(self.h)
            self.h = None

    __del__ = close


class HipCsr:
    """A contact matrix uploaded once (pk_csr): canonical CSR in, device-side facts out."""

    def __init__(self, M, device=0):
        L = require_device()
        self._L = L
        self.n = int(M.shape[0])
        nnz = int(M.indptr[-1])
        self.h = L.pk_csr_upload(device, self.n, np.ascontiguousarray(M.indptr, np.int32),
                                 np.ascontiguousarray(M.indices, np.int32) if nnz else np.zeros(1, np.int32),
                                 np.ascontiguousarray(M.data, np.float64) if nnz else np.zeros(1))
        if not self.h:
            raise PeakachuHipError("pk_csr_upload: " + last_error())
        self.device = device
        self._facts()

    def _facts(self):
        info = np.zeros(4, np.int64)
        vmax = C.c_double(0.0)
        check(self._L.pk_csr_info(self.h, info, C.byref(vmax)), "pk_csr_info")
        self.n_finite, self.n_nonfinite, self.n_noninteger, self.n_negative = [int(v) for v in info]
        self.vmax = float(vmax.value)

    @classmethod
    def from_upper(cls, n, indptr, cols, counts, bias=None, device=0):
        """A chromosome's upper triangle as a contact-map file stores it (pk_csr_upload_upper):
        the handle stands for the mirrored matrix, balanced by `bias` when given."""
        L = require_device()
        self = cls.__new__(cls)
        self._L, self.n, self.device = L, int(n), device
        indptr = np.ascontiguousarray(indptr, np.int32)
        nnz = int(indptr[-1])
        cols = np.ascontiguousarray(cols, np.int32) if nnz else np.zeros(1, np.int32)
        counts = np.asarray(counts)
        f64 = counts.dtype != np.int32
        counts = np.ascontiguousarray(counts, np.float64 if f64 else np.int32) if nnz else np.zeros(1, np.int32)
        b = None if bias is None else np.ascontiguousarray(bias, np.float64)
        self.h = L.pk_csr_upload_upper(device, self.n, indptr, cols, counts.ctypes.data, 1 if (f64 and nnz) else 0,
                                       None if b is None else b.ctypes.data)
        if not self.h:
            raise PeakachuHipError("pk_csr_upload_upper: " + last_error())
        self._facts()
        return self

    def view(self, bias=None):
        """The same stored entries with other biases (None: the plain values), nothing uploaded again."""
        b = None if bias is None else np.ascontiguousarray(bias, np.float64)
        h = self._L.pk_csr_view(self.h, None if b is None else b.ctypes.data)
        if not h:
            raise PeakachuHipError("pk_csr_view: " + last_error())
        other = type(self).__new__(type(self))
        other._L, other.n, other.device, other.h = self._L, self.n, self.device, h
        other._facts()
        return other

    def band(self, dlo, dhi, keep_nan=False):
        h = self._L.pk_matrix_from_csr(self.h, int(dlo), int(dhi), 1 if keep_nan else 0)
        if not h:
            raise PeakachuHipError("pk_matrix_from_csr: " + last_error())
        return HipMatrix._wrap(self._L, h, self.n, self.device)

    def expected_means(self, band, first, top, balanced):
        out = np.empty(top - first + 1, np.float64)
        check(self._L.pk_csr_expected_means(self.h, band.h, int(first), int(top), 1 if balanced else 0, out),
              "pk_csr_expected_means")
        return out

    def close(self):
        if getattr(self, "h", None):
            self._L.pk_csr_destroy(self.h)
            self.h = None

    __del__ = close


class HipMatrix(_Options):
    """Device-resident band matrix + expected vector (pk_matrix)."""
    _KIND = "matrix"

    def __init__(self, indptr, indices, data, n, exp_arr, dlo, dhi, device=0, options=None):
        L = require_device()
        self._L = L
        exp_arr = np.ascontiguousarray(exp_arr, np.float64)
        indices = np.ascontiguousarray(indices, np.int32)
        data = np.ascontiguousarray(data, np.float64)
        if indices.size == 0:
            indices = np.zeros(1, np.int32)
            data = np.zeros(1, np.float64)
        self.h = L.pk_matrix_create(device, int(n), np.ascontiguousarray(indptr, np.int32),
                                    indices, data, exp_arr, int(exp_arr.size), int(dlo), int(dhi))
        if not self.h:
            raise PeakachuHipError("pk_matrix_create: " + last_error())
        self.n, self.dlo, self.dhi, self.device = int(n), int(dlo), int(dhi), device
        self.set_options(options)

    @classmethod
    def _wrap(cls, L, h, n, device):
        self = cls.__new__(cls)
        self._L, self.h, self.n, self.device = L, h, n, device
        return self

    def set_expected(self, exp_arr):
        e = np.ascontiguousarray(exp_arr, np.float64)
        check(self._L.pk_matrix_set_expected(self.h, e, int(e.size)), "pk_matrix_set_expected")

    def expected_means(self, top, valid):
        """Diagonal means of calculate_expected (this band must start at diagonal 0)."""
        means = np.empty(int(top) + 1, np.float64)
        check(self._L.pk_expected_means(self.h, int(top), np.ascontiguousarray(valid, np.uint8),
                                        means), "pk_expected_means")
        return means

    def extract(self, w, x, y, want64=True, want32=False):
        x = np.ascontiguousarray(x, np.int32)
        y = np.ascontiguousarray(y, np.int32)
        N = x.size
        F = (2 * w + 1) ** 2
        keep = np.empty(max(N, 1), np.int64)
        f64 = np.empty((max(N, 1), F), np.float64) if want64 else None
        f32 = np.empty((max(N, 1), F), np.float32) if want32 else None
        nk = C.c_int64(0)
        if N:
            check(self._L.pk_extract(self.h, int(w), N, x, y,
                                     f64.ctypes.data if want64 else None,
                                     f32.ctypes.data if want32 else None, keep, C.byref(nk)),
                  "pk_extract")
        k = nk.value
        return (f64[:k] if want64 else None), (f32[:k] if want32 else None), keep[:k]

    def score(self, forest, w, thre, x, y, batch=100000):
        x = np.ascontiguousarray(x, np.int32)
        y = np.ascontiguousarray(y, np.int32)
        N = x.size
        # the C call wants room for N pixels; the buffers are kept with the matrix (grow-only) and
        # the scored pixels handed out as copies -- four fresh N-sized arrays per call cost more
        # than copying the few thousand pixels that pass
        with self.__dict__.setdefault("_score_lock", threading.Lock()):  # (the buffers are this object's)
            out = getattr(self, "_score_out", None)
            if out is None or out[0].size < max(N, 1):
                out = self._score_out = (np.empty(max(N, 1), np.int32), np.empty(max(N, 1), np.int32),
                                         np.empty(max(N, 1), np.float64), np.empty(max(N, 1), np.float64))
            ox, oy, op, osig = out
            nout = C.c_int64(0)
            if N:
                check(self._L.pk_score(self.h, forest.h, int(w), float(thre), int(batch), N, x, y,
                                       ox, oy, op, osig, C.byref(nout)), "pk_score")
            k = nout.value
            return ox[:k].copy(), oy[:k].copy(), op[:k].copy(), osig[:k].copy()

    def close(self):
        if getattr(self, "h", None):
            self._L.pk_matrix_destroy(self.h)
            self.h = None

    __del__ = close


class HipCands(_Options):
    """Device-resident candidate list and per-candidate outputs (pk_cands)."""
    _KIND = "cands"

    @classmethod
    def from_band(cls, raw_matrix, lower, upper, bg, kstar=None, weights=None, mustar=None):
        """get_candidate on the device (pk_candidates_create).  Returns
        (HipCands, n_ambiguous); with n_ambiguous > 0 the list is not complete
        and the caller must use the host path."""
        L = require_device()
        bg = np.ascontiguousarray(bg, np.float64)
        n = C.c_int64(0)
        amb = C.c_int64(0)
        ks = np.ascontiguousarray(kstar, np.int64) if kstar is not None else None
        w = np.ascontiguousarray(weights, np.float64) if weights is not None else None
        ms = np.ascontiguousarray(mustar, np.float64) if mustar is not None else None
        h = L.pk_candidates_create(raw_matrix.h, int(lower), int(upper),
                                   ks.ctypes.data if ks is not None else None, bg,
                                   w.ctypes.data if w is not None else None,
                                   ms.ctypes.data if ms is not None else None,
                                   int(ms.size) if ms is not None else 0, C.byref(n), C.byref(amb))
        if not h:
            raise PeakachuHipError("pk_candidates_create: " + last_error())
        self = cls.__new__(cls)
        self._L, self.h, self.N, self.n_out = L, h, n.value, 0
        return self, amb.value

    def coords(self):
        x = np.empty(max(self.N, 1), np.int32)
        y = np.empty(max(self.N, 1), np.int32)
        check(self._L.pk_cands_fetch(self.h, x, y), "pk_cands_fetch")
        return x[:self.N], y[:self.N]

    def __init__(self, x, y, device=0, options=None):
        L = require_device()
        self._L = L
        x = np.ascontiguousarray(x, np.int32)
        y = np.ascontiguousarray(y, np.int32)
        self.N = int(x.size)
        self.h = L.pk_cands_create(device, self.N, x if self.N else np.zeros(1, np.int32),
                                   y if self.N else np.zeros(1, np.int32))
        if not self.h:
            raise PeakachuHipError("pk_cands_create: " + last_error())
        self.n_out = 0
        self.set_options(options)

    def set_prune(self, on=True):
        """Exact early termination for this list's runs (same scored pixels)."""
        check(self._L.pk_cands_set_prune(self.h, 1 if on else 0), "pk_cands_set_prune")

    def run(self, matrix, forest, w, thre, batch=100000):
        nout = C.c_int64(0)
        check(self._L.pk_score_run(matrix.h, forest.h, self.h, int(w), float(thre), int(batch),
                                   C.byref(nout)), "pk_score_run")
        self.n_out = nout.value
        return self.n_out

    def fetch(self):
        k = max(self.n_out, 1)
        ox = np.empty(k, np.int32)
        oy = np.empty(k, np.int32)
        op = np.empty(k, np.float64)
        osig = np.empty(k, np.float64)
        check(self._L.pk_score_fetch(self.h, ox, oy, op, osig), "pk_score_fetch")
        k = self.n_out
        return ox[:k], oy[:k], op[:k], osig[:k]

    def fetch_all(self):
        st = np.empty(max(self.N, 1), np.uint8)
        pr = np.empty(max(self.N, 1), np.float64)
        check(self._L.pk_score_fetch_all(self.h, st, pr), "pk_score_fetch_all")
        return st[:self.N], pr[:self.N]

    def close(self):
        if getattr(self, "h", None):
            self._L.pk_cands_destroy(self.h)
            self.h = None

    __del__ = close


def set_option(name, value):
    check(load().pk_set_option(name.encode(), int(value)), "pk_set_option(%s)" % name)


def prof_get(name):
    ms = C.c_double(0)
    n = C.c_int64(0)
    check(load().pk_prof_get(name.encode(), C.byref(ms), C.byref(n)), "pk_prof_get")
    return ms.value, n.value
