"""ctypes binding of libhvs.so (C ABI: include/hvs.h)."""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(_HERE)
_CSRC = os.path.join(_HERE, "csrc")
_LIB = os.path.join(_CSRC, "libhvs.so")
_HDR = os.path.join(_REPO, "include", "hvs.h")

ENGINE_AUTO, ENGINE_EXACT_SCAN, ENGINE_MFMA_FILTER, ENGINE_MFMA_I8, ENGINE_MFMA_F16 = 0, 1, 2, 3, 4
K, DCOLS, QCOLS = 100, 102, 104

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17"]


class HvsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"hvs error {code}: {msg}")
        self.code = code


class Timing(C.Structure):
    _fields_ = [("query_ms", C.c_double), ("main_kernel_ms", C.c_double), ("main_kernel_launches", C.c_uint32),
                ("nq", C.c_uint32), ("pairs", C.c_uint64), ("scanned_pairs", C.c_uint64), ("load_ms", C.c_double),
                ("engine", C.c_uint32), ("fallback_queries", C.c_uint32), ("rescored_pairs", C.c_uint64),
                ("n_gpus", C.c_uint32), ("untimed_launches", C.c_uint32), ("host_ms", C.c_double),
                ("retry_queries", C.c_uint32), ("flags", C.c_uint32)]

    def as_dict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


def library_path():
    return _LIB


def sources():
    return [os.path.join(_CSRC, f) for f in sorted(os.listdir(_CSRC)) if f.endswith((".hip", ".h", ".cpp"))] + [
        _HDR, os.path.join(_REPO, "include", "hvs_gen.h"), os.path.join(_REPO, "include", "hvs_vec_query.hpp"),
        os.path.join(_REPO, "tests", "seam_main.cpp")]


def build_library(force=False, verbose=False):
    """Compile csrc/hvs.hip for gfx950 into csrc/libhvs.so (hipcc cross-compiles without a GPU)."""
    if (not force and os.path.exists(_LIB) and os.path.exists(cli_path()) and os.path.exists(seam_path())
            and all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in sources())):
        return _LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + HIPCC_FLAGS + [os.path.join(_CSRC, "hvs.hip"), "-o", _LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=_CSRC)
    build_cli(verbose)
    build_seam(verbose)
    return _LIB


def cli_path():
    return os.path.join(_CSRC, "hvs_search.out")


def build_cli(verbose=False):
    """The reference-compatible command-line driver (csrc/hvs_main.cpp, argv contract of src/test.cpp)."""
    cmd = ["g++", "-std=c++17", "-O2", "-ffp-contract=off", os.path.join(_CSRC, "hvs_main.cpp"), "-L" + _CSRC, "-lhvs",
           "-Wl,-rpath,$ORIGIN", "-o", cli_path()]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=_CSRC)
    cmd = ["g++", "-std=c++17", "-O2", "-ffp-contract=off", os.path.join(_CSRC, "hvs_compare.cpp"), "-o", compare_path()]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=_CSRC)
    return cli_path()


def seam_path():
    """tests/seam_main.cpp: the reference's src/test.cpp with include/hvs_vec_query.hpp as its engine header."""
    return os.path.join(_REPO, "tests", "seam_main.out")


def build_seam(verbose=False):
    cmd = ["g++", "-std=c++17", "-O2", "-I", os.path.join(_REPO, "include"), os.path.join(_REPO, "tests", "seam_main.cpp"),
           "-L" + _CSRC, "-lhvs", "-Wl,-rpath," + _CSRC, "-o", seam_path()]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return seam_path()


def compare_path():
    """The reference-compatible result checker (csrc/hvs_compare.cpp, CLI of src/compare_data.cpp)."""
    return os.path.join(_CSRC, "hvs_compare.out")


def exported_symbols():
    """Names declared in include/hvs.h (every one must be exported by libhvs.so)."""
    txt = open(_HDR).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hvs_[a-z_0-9]+)\s*\(", txt)))


_lib = None
_f32p, _u32p = C.POINTER(C.c_float), C.POINTER(C.c_uint32)


def library():
    """Load libhvs.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("HVS_LIB", _LIB)   # A/B experiments load an alternative build of the same ABI
    if not os.path.exists(path):
        raise HvsError(-100, f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = C.CDLL(path)
    vp = C.c_void_p
    sig = {
        "hvs_create": (C.c_int, [C.POINTER(vp), C.c_int]),
        "hvs_create_multi": (C.c_int, [C.POINTER(vp), C.c_int]),
        "hvs_create_on_devices": (C.c_int, [C.POINTER(vp), C.POINTER(C.c_int), C.c_int]),
        "hvs_num_gpus": (C.c_int, [vp]),
        "hvs_device_count": (C.c_int, []),
        "hvs_set_gather": (C.c_int, [vp, C.c_int]),
        "hvs_reserve": (C.c_int, [vp, C.c_uint32]),
        "hvs_destroy": (None, [vp]),
        "hvs_last_error": (C.c_char_p, [vp]),
        "hvs_last_global_error": (C.c_char_p, []),
        "hvs_set_engine": (C.c_int, [vp, C.c_int]),
        "hvs_set_distance_order": (C.c_int, [vp, C.c_int]),
        "hvs_set_padding": (C.c_int, [vp, C.c_int]),
        "hvs_set_k": (C.c_int, [vp, C.c_uint32]),
        "hvs_get_k": (C.c_uint32, [vp]),
        "hvs_load_data": (C.c_int, [vp, _f32p, C.c_uint32]),
        "hvs_gen_data": (C.c_int, [vp, C.c_uint32, C.c_uint64, C.c_int, C.c_uint32]),
        "hvs_download_data": (C.c_int, [vp, C.c_uint32, C.c_uint32, _f32p]),
        "hvs_num_rows": (C.c_uint32, [vp]),
        "hvs_query": (C.c_int, [vp, _f32p, C.c_uint32, C.c_float, _u32p, _f32p]),
        "hvs_upload_queries": (C.c_int, [vp, _f32p, C.c_uint32]),
        "hvs_gen_queries": (C.c_int, [vp, C.c_uint32, C.c_uint64, C.c_int, C.c_uint32, C.c_int, C.c_uint64]),
        "hvs_download_queries": (C.c_int, [vp, C.c_uint32, C.c_uint32, _f32p]),
        "hvs_query_resident": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_float]),
        "hvs_sync": (C.c_int, [vp]),
        "hvs_download_results": (C.c_int, [vp, C.c_uint32, C.c_uint32, _u32p, _f32p]),
        "hvs_export_results_device": (C.c_int, [vp, C.c_uint32, C.c_uint32, vp, vp]),
        "hvs_stream_wait": (C.c_int, [vp, vp]),
        "hvs_merge_shards_device": (C.c_int, [vp, C.c_uint32, C.c_uint32, vp, vp, C.POINTER(C.c_uint64), C.c_uint32, vp, vp, vp]),
        "hvs_last_timing": (C.c_int, [vp, C.POINTER(Timing)]),
        "hvs_last_reruns": (C.c_int, [vp, C.c_int, _u32p, C.c_uint32]),
        "hvs_version": (C.c_char_p, []),
        "hvs_plan_guess_m": (C.c_uint32, [C.c_uint32, C.c_double, C.c_uint32]),
        "hvs_plan_batches": (C.c_uint32, [C.c_uint32, C.c_int, _u32p, C.c_uint32]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def _fp(a):
    return a.ctypes.data_as(_f32p)


def _up(a):
    return a.ctypes.data_as(_u32p)


class Engine:
    """One hvs_ctx.  Engine(device): one GPU.  Engine(n_gpus=N) (0 = all visible) or Engine(devices=[...]): the
    multi-GPU context of hvs_create_multi / hvs_create_on_devices -- D replicated, the queries of a call partitioned,
    every GPU writing its slice of the result (a device index may repeat: virtual ranks on one GPU)."""

    def __init__(self, device=-1, n_gpus=None, devices=None):
        self._lib = library()
        h = C.c_void_p()
        if devices is not None:
            arr = (C.c_int * len(devices))(*[int(d) for d in devices])
            rc = self._lib.hvs_create_on_devices(C.byref(h), arr, len(devices))
        elif n_gpus is not None:
            rc = self._lib.hvs_create_multi(C.byref(h), int(n_gpus))
        else:
            rc = self._lib.hvs_create(C.byref(h), device)
        if rc != 0:
            raise HvsError(rc, self._lib.hvs_last_global_error().decode())
        self._h = h

    @property
    def num_gpus(self):
        return int(self._lib.hvs_num_gpus(self._h))

    def set_gather(self, mode):
        """0 = every GPU copies its block into its slice of the caller's array, 1 = peer gather to GPU 0 first."""
        self._ck(self._lib.hvs_set_gather(self._h, int(mode)))

    def reserve(self, nq):
        """Allocate query/result buffers and the batch workspace for calls of up to nq queries now."""
        self._ck(self._lib.hvs_reserve(self._h, int(nq)))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.hvs_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _ck(self, rc):
        if rc != 0:
            raise HvsError(rc, self._lib.hvs_last_error(self._h).decode())

    def set_engine(self, engine):
        self._ck(self._lib.hvs_set_engine(self._h, engine))

    def set_distance_order(self, order):
        """0 = the hot path's SIMD order (default), 1 = the baseline engine's sequential order."""
        self._ck(self._lib.hvs_set_distance_order(self._h, order))

    def set_k(self, k):
        """Neighbours per query (the reference's compile-time KNN_LIMIT): 8..256, default 100."""
        self._ck(self._lib.hvs_set_k(self._h, int(k)))

    @property
    def k(self):
        return int(self._lib.hvs_get_k(self._h))

    def set_padding(self, enabled):
        """Off: answers of a data shard keep id 0xFFFFFFFF / distance +inf in unmatched slots."""
        self._ck(self._lib.hvs_set_padding(self._h, int(bool(enabled))))

    # --- data
    def load_data(self, rows):
        rows = np.ascontiguousarray(rows, np.float32)
        if rows.ndim != 2 or rows.shape[1] != DCOLS:
            raise HvsError(-1, "data rows must be n x 102 float32")
        self._ck(self._lib.hvs_load_data(self._h, _fp(rows), rows.shape[0]))

    def gen_data(self, n, seed, profile=1, ncat=100):
        self._ck(self._lib.hvs_gen_data(self._h, n, seed, profile, ncat))

    def download_data(self, row0, nrows):
        out = np.empty((nrows, DCOLS), np.float32)
        self._ck(self._lib.hvs_download_data(self._h, row0, nrows, _fp(out)))
        return out

    @property
    def n(self):
        return int(self._lib.hvs_num_rows(self._h))

    # --- the vec_query seam
    def query(self, q_rows, sample_proportion=1.0, want_dists=True, out_ids=None, out_dists=None):
        """hvs_query: host rows in, host ids (and distances) out.  `out_ids` / `out_dists`: caller-provided arrays
        (e.g. views of pinned memory) to be filled instead of fresh ones."""
        q = np.ascontiguousarray(q_rows, np.float32)
        if q.ndim != 2 or q.shape[1] != QCOLS:
            raise HvsError(-1, "query rows must be nq x 104 float32")
        nq, K = q.shape[0], self.k
        ids = out_ids if out_ids is not None else np.empty((nq, K), np.uint32)
        d = out_dists if out_dists is not None else (np.empty((nq, K), np.float32) if want_dists else None)
        if ids.shape != (nq, K) or ids.dtype != np.uint32 or not ids.flags.c_contiguous:
            raise HvsError(-1, "out_ids must be a C-contiguous nq x k uint32 array")
        if d is not None and (d.shape != (nq, K) or d.dtype != np.float32 or not d.flags.c_contiguous):
            raise HvsError(-1, "out_dists must be a C-contiguous nq x k float32 array")
        self._ck(self._lib.hvs_query(self._h, _fp(q), nq, sample_proportion, _up(ids), _fp(d) if d is not None else None))
        return (ids, d) if d is not None else ids

    # --- resident variant
    def upload_queries(self, q_rows):
        q = np.ascontiguousarray(q_rows, np.float32)
        self._ck(self._lib.hvs_upload_queries(self._h, _fp(q), q.shape[0]))

    def gen_queries(self, nq, seed, profile=1, ncat=100, force_type=-1, first_row=0):
        self._ck(self._lib.hvs_gen_queries(self._h, nq, seed, profile, ncat, force_type, first_row))

    def download_queries(self, q0, nq):
        out = np.empty((nq, QCOLS), np.float32)
        self._ck(self._lib.hvs_download_queries(self._h, q0, nq, _fp(out)))
        return out

    def query_resident(self, q0, nq, sample_proportion=1.0):
        self._ck(self._lib.hvs_query_resident(self._h, q0, nq, sample_proportion))

    def sync(self):
        self._ck(self._lib.hvs_sync(self._h))

    def download_results(self, q0, nq, want_dists=True):
        K = self.k
        ids = np.empty((nq, K), np.uint32)
        d = np.empty((nq, K), np.float32) if want_dists else None
        self._ck(self._lib.hvs_download_results(self._h, q0, nq, _up(ids), _fp(d) if want_dists else None))
        return (ids, d) if want_dists else ids

    def export_results_device(self, q0, nq, ids_ptr, dists_ptr=None):
        """Copy results into device buffers given as raw pointers (e.g. torch tensor.data_ptr())."""
        self._ck(self._lib.hvs_export_results_device(self._h, q0, nq, C.c_void_p(ids_ptr),
                                                     C.c_void_p(dists_ptr) if dists_ptr else None))

    def stream_wait(self, stream_ptr):
        """Work enqueued on the given hipStream_t (raw pointer, e.g. torch.cuda.current_stream().cuda_stream) from now on
        waits for everything this context has enqueued so far -- a stream-ordered hand-off, no host wait for the kernels."""
        self._ck(self._lib.hvs_stream_wait(self._h, C.c_void_p(stream_ptr)))

    def merge_shards_device(self, ids_all_ptr, dists_all_ptr, shard_row0, nq, n_total, pad_dists_ptr, out_ids_ptr,
                            out_dists_ptr=None):
        """D-sharded mode: merge [nshards][nq][100] partial answers (device pointers, e.g. the output of an
        all_gather) into the whole-set answer on the device; see hvs_merge_shards_device in include/hvs.h."""
        rows = (C.c_uint64 * len(shard_row0))(*[int(r) for r in shard_row0])
        self._ck(self._lib.hvs_merge_shards_device(self._h, len(shard_row0), nq, C.c_void_p(ids_all_ptr),
                                                   C.c_void_p(dists_all_ptr), rows, n_total, C.c_void_p(pad_dists_ptr),
                                                   C.c_void_p(out_ids_ptr),
                                                   C.c_void_p(out_dists_ptr) if out_dists_ptr else None))

    def last_reruns(self, which):
        """Query indices of the last call answered a second time: which = 0 the exact engine's list, 1 the retry list."""
        n = self._lib.hvs_last_reruns(self._h, int(which), None, 0)
        if n < 0:
            self._ck(n)
        out = np.empty(n, np.uint32)
        if n:
            self._ck(min(0, self._lib.hvs_last_reruns(self._h, int(which), _up(out), n)))
        return out

    def last_timing(self):
        t = Timing()
        self._ck(self._lib.hvs_last_timing(self._h, C.byref(t)))
        return t
