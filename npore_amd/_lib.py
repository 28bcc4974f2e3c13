"""ctypes binding of libnpore_amd.so (include/npore_amd.h).

The library holds the gfx950 kernels; there is no CPU implementation of the DP
behind this module.  Loading fails loudly if the shared object has not been
built (python __graft_entry__.py build) and context creation fails loudly if
no MI355X is visible.
"""
import ctypes as C
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
# NPORE_AMD_LIB selects another build of the same ABI (measurement builds: tests/tools/ab_sync.py, scripts/ab_fill.py)
LIB_PATH = os.environ.get("NPORE_AMD_LIB") or os.path.join(_HERE, "libnpore_amd.so")
RELAXED_LIB_PATH = os.path.join(_HERE, "libnpore_amd_relaxed.so")
CSRC = os.path.join(_HERE, "csrc")
_LIB = None

# name -> (restype, argtypes): must list every symbol include/npore_amd.h declares
SIGNATURES = {
    "npore_abi_version": (C.c_int, []),
    "npore_last_error": (C.c_char_p, []),
    "npore_device_count": (C.c_int, []),
    "npore_ctx_create": (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "npore_ctx_destroy": (None, [C.c_void_p]),
    "npore_align_batch": (C.c_int, [C.c_void_p, C.c_int64] + [C.c_void_p] * 6 +
                          [C.c_float, C.c_float, C.c_int, C.c_int] + [C.c_void_p] * 4),
    "npore_align_batch_async": (C.c_int, [C.c_void_p, C.c_int64] + [C.c_void_p] * 6 +
                                [C.c_float, C.c_float, C.c_int, C.c_int] + [C.c_void_p] * 4),
    "npore_align_batch_cigars": (C.c_int, [C.c_void_p, C.c_int64] + [C.c_void_p] * 6 +
                                 [C.c_float, C.c_float, C.c_int, C.c_int] + [C.c_void_p] * 4),
    "npore_align_batch_device": (C.c_int, [C.c_void_p, C.c_int64] + [C.c_void_p] * 6 +
                                 [C.c_float, C.c_float, C.c_int, C.c_int] + [C.c_void_p] * 4 +
                                 [C.c_void_p, C.c_int]),
    "npore_ctx_wait": (C.c_int, [C.c_void_p]),
    "npore_total_timing": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "npore_get_np_info": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "npore_np_regions": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_void_p),
                                   C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "npore_standardize_batch": (C.c_int, [C.c_int64] + [C.c_void_p] * 9 + [C.c_int]),
    "npore_standardize_ops_batch": (C.c_int, [C.c_int64] + [C.c_void_p] * 9 + [C.c_int]),
    "npore_confusion_counts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                         C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_int]),
    "npore_last_timing": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "npore_ctx_set": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "npore_round_chunks": (C.c_int64, [C.c_void_p, C.c_int]),
    "npore_fill_shape": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "npore_bam_open": (C.c_void_p, [C.c_char_p, C.c_int]),
    "npore_bam_open_mode": (C.c_void_p, [C.c_char_p, C.c_int, C.c_int, C.c_char_p]),
    "npore_bam_is_streamed": (C.c_int, [C.c_void_p]),
    "npore_bam_save_index": (C.c_int, [C.c_void_p, C.c_char_p]),
    "npore_bam_close": (None, [C.c_void_p]),
    "npore_bam_dump_inflated": (C.c_int, [C.c_void_p, C.c_char_p]),
    "npore_bam_inflated_size": (C.c_int64, [C.c_void_p]),
    "npore_bam_n_records": (C.c_int64, [C.c_void_p]),
    "npore_bam_n_refs": (C.c_int, [C.c_void_p]),
    "npore_bam_ref_name": (C.c_char_p, [C.c_void_p, C.c_int]),
    "npore_bam_ref_len": (C.c_int64, [C.c_void_p, C.c_int]),
    "npore_bam_ref_has_reads": (C.c_int, [C.c_void_p, C.c_int]),
    "npore_bam_select": (C.c_int64, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]),
    "npore_fasta_open": (C.c_void_p, [C.c_char_p]),
    "npore_fasta_close": (None, [C.c_void_p]),
    "npore_fasta_n": (C.c_int, [C.c_void_p]),
    "npore_fasta_name": (C.c_char_p, [C.c_void_p, C.c_int]),
    "npore_fasta_len": (C.c_int64, [C.c_void_p, C.c_int]),
    "npore_fasta_seq": (C.c_void_p, [C.c_void_p, C.c_int]),
    "npore_bam_pack_sizes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "npore_bam_pack": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64] + [C.c_void_p] * 6 + [C.c_int]),
    "npore_bam_format_sam": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_int, C.c_void_p, C.c_void_p]),
    "npore_bam_realign_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                          C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "npore_bam_realign_file": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64,
                                         C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_void_p]),
    "npore_bam_realign_sequential": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_int64, C.c_int64, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_char_p,
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "npore_bam_set_share": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char_p]),
    "npore_bam_share_info": (C.c_int, [C.c_void_p, C.c_void_p]),
    "npore_bam_last_timing": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "npore_bam_file_timing": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "npore_debug_inflate": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int]),
    "npore_debug_inflate_pair": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int]),
    "npore_debug_crc32": (C.c_int64, [C.c_void_p, C.c_int64, C.c_uint32]),
    "npore_debug_dpp": (C.c_int, [C.c_void_p]),
    "npore_debug_divcheck": (C.c_int, [C.c_void_p]),
    "npore_debug_fetch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]),
    "npore_debug_fetch_tb": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]),
}


def sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if os.path.isfile(os.path.join(CSRC, f))] + \
        [os.path.join(_HERE, "..", "include", "npore_amd.h")]


def build(force=False, verbose=False, relaxed=False, defines=(), out=None):
    """hipcc --offload-arch=gfx950 -> npore_amd/libnpore_amd.so (in-tree): the product library.
    Measurement builds (never made by __graft_entry__.build()): `defines` = experiment macros of csrc/experiments.hpp
    (compiled with -DNPORE_EXPERIMENTS, `out` = where to put the library); relaxed=True is shorthand for
    libnpore_amd_relaxed.so with -DNPORE_RELAXED_SYNC (tests/tools/ab_sync.py)."""
    if relaxed:
        defines, out = tuple(defines) + ("NPORE_RELAXED_SYNC",), out or RELAXED_LIB_PATH
    if defines and not out:
        raise ValueError("a measurement build needs its own output path")
    out = out or os.path.join(_HERE, "libnpore_amd.so")
    if not force and os.path.exists(out) and all(os.path.getmtime(s) <= os.path.getmtime(out) for s in sources()):
        return out
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-Wall", "-Wno-unused-function"] + \
          (["-DNPORE_EXPERIMENTS"] + ["-D" + d for d in defines] if defines else []) + \
          ["-o", out, os.path.join(CSRC, "npore_api.cpp"), "-lz"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out


def load():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python __graft_entry__.py build` "
                "(hipcc --offload-arch=gfx950).  npore_amd has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)   # AttributeError if the header and the library diverge
            fn.restype = res
            fn.argtypes = args
        if lib.npore_abi_version() != 2:
            raise ImportError("libnpore_amd.so ABI version mismatch")
        _LIB = lib
    return _LIB


def last_error():
    return load().npore_last_error().decode()
