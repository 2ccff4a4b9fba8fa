"""ctypes binding of libflmr_hip.so (the C ABI declared in include/flmr_hip.h).

The HIP library is the product path.  There is NO CPU fallback here: if the shared object is missing or no
MI355X is visible, loading fails loudly (FlmrNativeError).  PyTorch is used by callers only as plumbing for
device memory and streams; every entry point takes raw device pointers.
"""
import ctypes as C
import os
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)
LIB_PATH = os.environ.get("FLMR_HIP_LIB") or os.path.join(PKG_DIR, "lib", "libflmr_hip.so")  # override: A/B-ing kernel builds
CSRC = os.path.join(PKG_DIR, "csrc")
SOURCES = ["flmr_index.hip", "flmr_stage0.hip", "flmr_candidates.hip", "flmr_filter.hip", "flmr_stage1_dense.hip", "flmr_stage2_walk.hip", "flmr_stage2_xcd.hip", "flmr_maxsim.hip", "flmr_search.hip", "flmr_ops.hip", "flmr_build.hip", "flmr_ivf.hip", "flmr_collective.hip"]
HEADERS = ["flmr_common.h", "flmr_device.h"]

FLMR_MEM_HOST, FLMR_MEM_DEVICE = 0, 1
(TAP_CENTROID_SCORES, TAP_IDX_BITS, TAP_CELLS, TAP_CANDIDATES, TAP_STAGE1, TAP_STAGE2, TAP_DOC_SCORES, TAP_Q_ERR,
 TAP_Q_ERR_SUM, TAP_STAGE1_FORM) = range(10)
NUM_STAGES = 9
ABI_VERSION = 5


class FlmrNativeError(RuntimeError):
    pass


class IndexDesc(C.Structure):
    _fields_ = [("dim", C.c_int32), ("nbits", C.c_int32), ("num_centroids", C.c_int32), ("memory", C.c_int32),
                ("num_embeddings", C.c_int64), ("num_passages", C.c_int64), ("pid_base", C.c_int64),
                ("codes", C.c_void_p), ("residuals", C.c_void_p), ("doc_offsets", C.c_void_p),
                ("ivf_pids", C.c_void_p), ("ivf_offsets", C.c_void_p), ("centroids", C.c_void_p),
                ("bucket_weights", C.c_void_p)]


class IndexInfo(C.Structure):
    _fields_ = [("derived_bytes", C.c_int64), ("max_doclen", C.c_int64), ("centroids_f16_exact", C.c_int32),
                ("stage2_slices", C.c_int32), ("xcd_round_robin", C.c_int32), ("stage2_sliced", C.c_int32),
                ("passage_chunks", C.c_int32), ("duplicate_permille", C.c_int32)]


class SearchParams(C.Structure):
    _fields_ = [("k", C.c_int32), ("ncells", C.c_int32), ("centroid_score_threshold", C.c_float),
                ("ndocs", C.c_int32), ("nq_cand", C.c_int32)]


def hipcc_path():
    for p in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if p and (os.path.isabs(p) and os.path.exists(p) or not os.path.isabs(p)):
            return p
    return "hipcc"


def _record_toolchain(lib_path, flags):
    """The compiler a binary came from, written beside it: the hand-scheduled kernels (inline-asm loads with counted waits)
    are verified bit-exact with THIS toolchain; bench.py reports the string."""
    try:
        ver = subprocess.run([hipcc_path(), "--version"], capture_output=True, text=True).stdout.strip().splitlines()
        with open(lib_path + ".toolchain", "w") as f:
            f.write("\n".join(ver[:2]) + "\nflags: " + " ".join(flags) + "\n")
    except OSError:
        pass


def build_native(force=False, verbose=False, extra_flags=(), lib_path=None, obj_dir=None):
    """Compile csrc/*.hip for gfx950 into lib/libflmr_hip.so (cross-compiles without a GPU).  One object per source,
    compiled in parallel and only when the source or a header is newer; `extra_flags` (e.g. -DFLMR_EXPERIMENTAL_VARIANTS,
    -DX2_PROFILE) build a variant library into `lib_path` with its own object directory."""
    from concurrent.futures import ThreadPoolExecutor
    lib_path = lib_path or LIB_PATH
    obj_dir = obj_dir or os.path.join(os.path.dirname(lib_path), "obj" if not extra_flags else "obj_%08x" % __import__("zlib").crc32(" ".join(extra_flags).encode()))
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(REPO_ROOT, "include", "flmr_hip.h")]
    hdr_time = max(os.path.getmtime(h) for h in hdrs)
    os.makedirs(obj_dir, exist_ok=True)
    base = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-inline-asm",
            "-I" + os.path.join(REPO_ROOT, "include"), "-I" + CSRC] + list(extra_flags)
    jobs, objs = [], []
    for name in SOURCES:
        src, obj = os.path.join(CSRC, name), os.path.join(obj_dir, name[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            jobs.append(base + ["-c", src, "-o", obj])
    if not jobs and os.path.exists(lib_path) and all(os.path.getmtime(lib_path) >= os.path.getmtime(o) for o in objs):
        if not os.path.exists(lib_path + ".toolchain"):
            _record_toolchain(lib_path, base[1:6] + list(extra_flags))
        return lib_path

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    os.makedirs(os.path.dirname(lib_path), exist_ok=True)
    run([hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib_path])
    _record_toolchain(lib_path, base[1:6] + list(extra_flags))
    return lib_path


def toolchain():
    """The hipcc / clang version lines recorded when lib/libflmr_hip.so was built ("" if the record is missing)."""
    try:
        with open(LIB_PATH + ".toolchain") as f:
            return f.read().strip()
    except OSError:
        return ""


_lib = None

_SIGS = {
    "flmr_abi_version": (C.c_int, []),
    "flmr_last_error": (C.c_char_p, []),
    "flmr_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "flmr_set_option": (C.c_int, [C.c_char_p, C.c_char_p]),
    "flmr_searcher_check": (C.c_int, [C.c_void_p]),
    "flmr_searcher_status_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "flmr_searcher_probe_supported": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(SearchParams), C.POINTER(C.c_int32)]),
    "flmr_index_open": (C.c_int, [C.POINTER(IndexDesc), C.POINTER(C.c_void_p)]),
    "flmr_index_close": (C.c_int, [C.c_void_p]),
    "flmr_index_info": (C.c_int, [C.c_void_p, C.POINTER(IndexInfo)]),
    "flmr_searcher_create": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(SearchParams), C.POINTER(C.c_void_p)]),
    "flmr_searcher_destroy": (C.c_int, [C.c_void_p]),
    "flmr_searcher_workspace_bytes": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "flmr_search_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(SearchParams),
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "flmr_search_phase1": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(SearchParams), C.c_void_p, C.c_void_p]),
    "flmr_searcher_probe_dims": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "flmr_search_probe": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(SearchParams), C.c_int32,
                                    C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "flmr_search_phase1_probed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(SearchParams),
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "flmr_search_phase2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(SearchParams), C.c_void_p,
                                     C.c_int32, C.c_void_p, C.c_void_p]),
    "flmr_search_phase3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(SearchParams), C.c_void_p,
                                     C.c_int32, C.c_void_p, C.c_void_p]),
    "flmr_topn_keys": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "flmr_select_keys": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "flmr_unpack_keys": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "flmr_nearest_centroids": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "flmr_build_ivf": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p]),
    "flmr_compress_residuals": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32,
                                          C.c_void_p, C.c_void_p]),
    "flmr_searcher_tap": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    "flmr_searcher_set_profiling": (C.c_int, [C.c_void_p, C.c_int32]),
    "flmr_searcher_set_full_table": (C.c_int, [C.c_void_p, C.c_int32]),
    "flmr_searcher_set_numerics": (C.c_int, [C.c_void_p, C.c_int32]),
    "flmr_searcher_stage_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "flmr_stage_name": (C.c_char_p, [C.c_int32]),
    "flmr_filter_pids": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "flmr_decompress_residuals": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                            C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "flmr_segmented_lookup": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                        C.c_int64, C.c_void_p, C.c_void_p]),
    "flmr_segmented_maxsim": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "flmr_score_pids": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "flmr_colbert_score_padded": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                            C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "flmr_colbert_score_cross": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                           C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "flmr_colbert_colmax_padded": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                             C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "flmr_topk_allgather": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "flmr_keys_allgather": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "flmr_keys_allreduce_sum": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "flmr_merge_topk": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
}
EXPORTED_SYMBOLS = sorted(_SIGS)


def load(require_device=True):
    """dlopen the library (never builds implicitly, never falls back).  With require_device the call also
    fails when no HIP device is visible -- the product path is GPU-only."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FlmrNativeError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
                "There is no CPU fallback for the search path.")
        lib = C.CDLL(LIB_PATH)
        lib.flmr_abi_version.restype = C.c_int
        if lib.flmr_abi_version() != ABI_VERSION:   # (checked before the symbols are bound: a stale library says so, not AttributeError)
            raise FlmrNativeError(f"{LIB_PATH}: ABI version {lib.flmr_abi_version()}, this package expects {ABI_VERSION}: rebuild it "
                                  "(python -c 'import __graft_entry__ as g; g.build()')")
        for name, (res, args) in _SIGS.items():
            fn = getattr(lib, name)  # AttributeError => ABI mismatch, surfaced loudly
            fn.restype, fn.argtypes = res, args
        _lib = lib
    if require_device:
        n = C.c_int(0)
        rc = _lib.flmr_device_count(C.byref(n))
        if rc != 0 or n.value < 1:
            raise FlmrNativeError("no MI355X / HIP device visible: " + _lib.flmr_last_error().decode())
    return _lib


def device_visible():
    """True when the library loads AND reports at least one HIP device (a missing / mismatching library still raises)."""
    lib = load(require_device=False)
    n = C.c_int(0)
    return lib.flmr_device_count(C.byref(n)) == 0 and n.value >= 1


options_epoch = 0  # bumped by set_option(): IndexScorer re-creates its native searcher so the new switches apply


def set_option(name, value=None):
    """flmr_set_option: kernel-variant switch for A/B runs / cross-check tests (value None clears it)."""
    global options_epoch
    check(load(False).flmr_set_option(name.encode(), None if value is None else str(value).encode()))
    options_epoch += 1


class options:
    """`with _native.options(FLMR_S1_IMPL="scan"): ...` -- switches set for the block, cleared afterwards."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        for k, v in self.kw.items():
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k in self.kw:
            set_option(k, None)
        return False


def check(rc):
    if rc != 0:
        raise FlmrNativeError(f"libflmr_hip status {rc}: {load(False).flmr_last_error().decode()}")


def stream_ptr(torch_stream=None):
    """Raw hipStream_t of the current (or given) torch stream."""
    import torch
    s = torch_stream if torch_stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)
