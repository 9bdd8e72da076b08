"""MI355X-native filtered brute-force k-NN: Python host mirror of the reference seam.

The product is csrc/ (HIP kernels + the C ABI of include/hvs.h, built into csrc/libhvs.so).
This package is the thin host side used by tests and bench.py: ctypes bindings plus the
reference's own entry-point names (`vec_query`, `ReadBin`, `SaveKNN`, `SaveKNNFull`; reference
include/optimized_parallel.hpp:61-62 and include/io.h:23-136).  There is no CPU fallback:
everything raises if libhvs.so or a GPU is missing.

The directory name contains dashes, so import it with
    importlib.import_module("project---hybrid-vector-search-queries_amd")
"""
from .engine import (Engine, HvsError, Timing, library, library_path, build_library, build_cli, build_seam, seam_path, cli_path, compare_path, exported_symbols,  # noqa: F401
                     ENGINE_AUTO, ENGINE_EXACT_SCAN, ENGINE_MFMA_FILTER, ENGINE_MFMA_I8, ENGINE_MFMA_F16)
from .vec_query import vec_query, ReadBin, SaveKNN, SaveKNNFull, calc_dist  # noqa: F401

__all__ = ["Engine", "HvsError", "Timing", "library", "library_path", "build_library", "build_cli", "build_seam", "seam_path", "cli_path", "compare_path", "exported_symbols",
           "vec_query", "ReadBin", "SaveKNN", "SaveKNNFull", "calc_dist",
           "ENGINE_AUTO", "ENGINE_EXACT_SCAN", "ENGINE_MFMA_FILTER", "ENGINE_MFMA_I8", "ENGINE_MFMA_F16"]
