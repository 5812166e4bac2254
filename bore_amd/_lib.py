"""ctypes binding of libbore_hip.so (C-ABI: include/bore_hip.h).

The shared library is built IN-TREE (bore_amd/csrc/libbore_hip.so) by
``build_native()`` / ``__graft_entry__.build()`` with ``hipcc --offload-arch=gfx950``.
There is no CPU fallback: if the library is missing, every compute entry point
raises ``RuntimeError`` -- the product path never routes around the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# (BORE_LIB_PATH: another build of the same ABI, for A/B comparisons -- tools/ab_fit.py.  It is
# only ever LOADED: build_native() writes the default path and nothing else.)
DEFAULT_LIB_PATH = os.path.join(CSRC, "libbore_hip.so")
LIB_PATH = os.environ.get("BORE_LIB_PATH") or DEFAULT_LIB_PATH
SOURCES = ["bore_all.hip"]   # a unity build of bore_{hip,argmax,svgd,iter,engine}.hip
HEADER = os.path.join(os.path.dirname(_HERE), "include", "bore_hip.h")

# The replica engine runs up to a dozen independent launches on as many HIP streams; ROCm gives a
# process GPU_MAX_HW_QUEUES hardware queues (default 4) and streams sharing one serialise.  Read at
# HIP initialisation, so it has to be in the environment before the first GPU call.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

MAX_LAYERS = 8
BATCH_MAX = 64

ACT = dict(linear=0, relu=1, elu=2, sigmoid=3, tanh=4)
TRANSFORM = dict(identity=0, sigmoid=1, exp=2)

# every symbol include/bore_hip.h declares (tests check the .so exports them all)
EXPORTS = [
    "bore_abi_version", "bore_source_digest", "bore_last_error", "bore_param_count", "bore_mlp_forward",
    "bore_mlp_value_and_input_grad", "bore_mlp_fit", "bore_mlp_evaluate",
    "bore_shuffle_perm", "bore_labels", "bore_uniform_candidates", "bore_screen_topk", "bore_sample_screen_topk",
    "bore_lbfgsb_minimize", "bore_append_observations", "bore_select_best",
    "bore_svgd_optimize", "bore_set_batch", "bore_engine_create", "bore_engine_run", "bore_engine_size", "bore_engine_observations",
    "bore_engine_state", "bore_engine_get_stats", "bore_engine_destroy", "bore_objective_branin01",
]


class MlpDesc(C.Structure):
    _fields_ = [("input_dim", C.c_int32), ("n_layers", C.c_int32),
                ("units", C.c_int32 * MAX_LAYERS), ("act", C.c_int32 * MAX_LAYERS),
                ("l2_kernel", C.c_float * MAX_LAYERS), ("l2_bias", C.c_float * MAX_LAYERS),
                ("compute", C.c_int32)]


class LbfgsbOpts(C.Structure):
    _fields_ = [("maxcor", C.c_int32), ("maxiter", C.c_int32), ("maxfun", C.c_int32),
                ("maxls", C.c_int32), ("ftol", C.c_double), ("gtol", C.c_double)]


class AdamCfg(C.Structure):
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("eps", C.c_float)]


class SvgdOpts(C.Structure):
    _fields_ = [("n_iter", C.c_int32), ("distortion", C.c_int32), ("step_size", C.c_double),
                ("alpha", C.c_double), ("eps", C.c_double), ("tau", C.c_double),
                ("length_scale", C.c_double), ("distortion_param", C.c_double)]


class EngineCfg(C.Structure):
    _fields_ = [("n_loops", C.c_int32), ("groups", C.c_int32), ("loop_id0", C.c_int64),
                ("n_init", C.c_int32), ("epochs", C.c_int32), ("batch_size", C.c_int32),
                ("num_starts", C.c_int32), ("num_samples", C.c_int32), ("transform", C.c_int32),
                ("deduplicate", C.c_int32), ("async_loops", C.c_int32), ("seed", C.c_uint64),
                ("gamma", C.c_double), ("adam", AdamCfg), ("lbfgsb", LbfgsbOpts),
                ("low", C.POINTER(C.c_double)), ("high", C.POINTER(C.c_double)),
                ("resident_wait_us", C.c_int32), ("worker_streams", C.c_int32), ("work_queue", C.c_int32),
                ("reserved", C.c_int32)]


class EngineStats(C.Structure):
    _fields_ = [("fit_ms", C.c_double), ("fit_bytes", C.c_double), ("argmax_ms", C.c_double),
                ("argmax_bytes", C.c_double), ("host_enqueue_s", C.c_double),
                ("host_finalize_s", C.c_double), ("fit_launches", C.c_int64),
                ("argmax_launches", C.c_int64), ("n_fg_rows", C.c_int64), ("n_rounds", C.c_int64),
                ("none_results", C.c_int64),
                ("phase_ns_labels", C.c_double), ("phase_ns_fit", C.c_double),
                ("phase_ns_screen", C.c_double), ("phase_ns_lbfgsb", C.c_double),
                ("ready_to_launch_s", C.c_double), ("launch_to_result_s", C.c_double),
                ("result_to_ready_s", C.c_double), ("phase_iterations", C.c_int64),
                ("batches", C.c_int64), ("worker_streams", C.c_int64),
                ("stream_concurrency", C.c_int64), ("n_fg_requests", C.c_int64),
                ("loops_per_cu", C.c_int64), ("side_by_side_workgroups", C.c_int64),
                ("host_threads", C.c_int64)]


OBJECTIVE_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int64, C.c_int32,
                           C.POINTER(C.c_double), C.c_void_p)


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


def source_digest():
    """sha256 over the kernel sources (bore_amd/csrc/*.hip, *.h and include/bore_hip.h, names and contents in sorted
    order): compiled into the library (bore_source_digest), stored with every committed PMC pass."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    files.append(HEADER)
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def built_digest(path=None):
    """The source digest compiled into the library at `path` (None: no library, or one from before ABI 11).
    Read from the FILE (the 64-hex-digit string `bore_source_digest()` returns sits in its read-only data), not by
    loading it: this runs before anything else of the package -- `build()` -- and a library opened ahead of torch
    binds to the system's HIP runtime instead of the one torch brings, after which no call of it sees a device
    (`lib()` imports torch first for that reason)."""
    path = path or DEFAULT_LIB_PATH
    if not os.path.exists(path):
        return None
    import re
    with open(path, "rb") as f:
        found = set(re.findall(rb"(?<![0-9a-f])([0-9a-f]{64})\x00", f.read()))
    want = source_digest().encode()
    if want in found:
        return want.decode()
    return sorted(found)[0].decode() if found else None


def build_native(force=False, verbose=False):
    """Compile libbore_hip.so for gfx950.  Cross-compiles without a GPU.  Always the default
    in-tree path: an experimental build named by BORE_LIB_PATH is never overwritten.
    Up to date = the digest compiled into the library is the digest of the sources in the tree (file times say
    nothing once a tree has travelled); force=True compiles regardless."""
    out = DEFAULT_LIB_PATH
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    digest = source_digest()
    if not force and built_digest(out) == digest:
        if verbose:
            print(f"{out}: built from these sources ({digest[:16]}...), nothing to do")
        return out
    hipcc = hipcc_path()
    if hipcc is None:
        raise RuntimeError("hipcc not found: cannot build libbore_hip.so")
    # (-Wall -Wextra is clean; -Wno-pass-failed: the 161 "loop not unrolled" notes of round 4's build log are the
    # optimiser declining `#pragma unroll` hints on loops with run-time trip counts, not defects)
    # -ffp-contract=off: no implicit FMA formation.  The fp32 network code spells its FMAs
    # out (fmaf); the fp64 L-BFGS-B then rounds exactly like its host build (tests compare
    # the two bit for bit) and like the unfused numpy/scipy arithmetic of the oracle.
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC",
           "-ffp-contract=off", "-Wall", "-Wextra", "-Wno-pass-failed", f'-DBORE_SRC_DIGEST="{digest}"', *srcs, "-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    return out


_lib = None


def abi_version_of_header():
    with open(HEADER) as f:
        return int(f.read().split("#define BORE_ABI_VERSION")[1].split()[0])


def lib():
    """The loaded library (loads on first use; raises if it was never built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950).  bore_amd has no CPU fallback.")
    # torch bundles its own libamdhip64 (SONAME libamdhip64.so.7).  It must be in the
    # process BEFORE this library is opened so that both resolve to ONE HIP runtime --
    # device pointers from torch tensors are only valid in the runtime that made them.
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u64 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64
    dp = C.POINTER(MlpDesc)
    L.bore_abi_version.restype = i32
    want = abi_version_of_header()
    if L.bore_abi_version() != want:      # (struct layouts below are this header's)
        raise RuntimeError(f"{LIB_PATH} has ABI {L.bore_abi_version()}, include/bore_hip.h declares "
                           f"{want}: rebuild it (`python -c 'import __graft_entry__ as g; g.build()'`)")
    L.bore_last_error.restype = C.c_char_p
    L.bore_param_count.restype = i64
    L.bore_param_count.argtypes = [dp]
    L.bore_mlp_forward.argtypes = [dp, i32, vp, vp, i64, i32, vp, vp]
    L.bore_mlp_value_and_input_grad.argtypes = [dp, i32, vp, vp, i64, i32, i32, vp, vp, vp]
    L.bore_mlp_fit.argtypes = [dp, i32, vp, vp, vp, vp, vp, vp, i64, i32, i32, vp, u64, i64,
                               i64, C.POINTER(AdamCfg), vp, vp]
    L.bore_mlp_evaluate.argtypes = [dp, i32, vp, vp, vp, i64, vp, vp, vp]
    L.bore_shuffle_perm.argtypes = [u64, i64, i32, i64, i32, i64, vp, vp]
    L.bore_labels.argtypes = [i32, vp, i64, C.c_double, vp, vp, vp]
    dpp = C.POINTER(C.c_double)
    L.bore_uniform_candidates.argtypes = [u64, i64, i32, i64, i64, i32, dpp, dpp, vp, vp]
    L.bore_screen_topk.argtypes = [dp, i32, vp, vp, i64, i32, i32, vp, vp, vp, vp]
    L.bore_sample_screen_topk.argtypes = [dp, i32, vp, u64, i64, i64, i64, dpp, dpp, i32, vp, vp, vp, vp]
    L.bore_lbfgsb_minimize.argtypes = [dp, i32, vp, i32, i32, vp, i32, dpp, dpp,
                                       C.POINTER(LbfgsbOpts), vp, vp, vp, vp, vp]
    L.bore_append_observations.argtypes = [i32, i32, vp, vp, i64, i64, vp, vp, vp, vp, vp]
    L.bore_select_best.argtypes = [i32, i32, i32, vp, vp, vp, vp, i64, i64, C.c_double,
                                   C.c_double, vp, vp, vp]
    L.bore_svgd_optimize.argtypes = [dp, i32, vp, i32, vp, i32, dpp, dpp, C.POINTER(SvgdOpts), vp, vp]
    L.bore_engine_create.argtypes = [dp, C.POINTER(EngineCfg), vp, vp, vp, vp, OBJECTIVE_FN, vp,
                                     C.POINTER(vp)]
    L.bore_engine_run.argtypes = [vp, i32]
    L.bore_engine_size.argtypes = [vp]
    L.bore_engine_observations.argtypes = [vp, vp, vp]
    L.bore_engine_state.argtypes = [vp, vp, vp, vp, vp]
    L.bore_engine_get_stats.argtypes = [vp, C.POINTER(EngineStats), i32]
    L.bore_engine_destroy.argtypes = [vp]
    for name in EXPORTS:
        if name not in ("bore_last_error", "bore_param_count", "bore_engine_size",
                        "bore_engine_destroy", "bore_set_batch"):
            getattr(L, name).restype = i32
    L.bore_engine_size.restype = i64
    L.bore_engine_destroy.restype = None
    L.bore_set_batch.restype = None
    L.bore_set_batch.argtypes = [vp]
    _lib = L
    return L


class UnsupportedError(RuntimeError):
    """BORE_E_UNSUPPORTED: the request does not fit this build's kernels (shape, LDS budget, ...)."""


class NeedsPermError(UnsupportedError):
    """BORE_E_NEEDS_PERM: bore_mlp_fit cannot draw the shuffles of this many rows in LDS; the same call
    with explicit permutations works (bore_amd.models passes them)."""


def check(rc):
    if rc != 0:
        cls = {-2: UnsupportedError, -5: NeedsPermError}.get(rc, RuntimeError)
        raise cls(f"libbore_hip: {lib().bore_last_error().decode()} (code {rc})")


COMPUTE = dict(float32=0, bfloat16=1)


def make_desc(input_dim, units, acts, l2_kernel=None, l2_bias=None, compute="float32"):
    n = len(units)
    if compute not in COMPUTE:
        raise ValueError(f"compute must be one of {sorted(COMPUTE)}, got {compute!r}")
    if not 1 <= n <= MAX_LAYERS:
        raise ValueError(f"1..{MAX_LAYERS} Dense layers supported, got {n}")
    d = MlpDesc()
    d.input_dim = int(input_dim)
    d.n_layers = n
    for i in range(n):
        d.units[i] = int(units[i])
        a = acts[i] if acts[i] is not None else "linear"
        if a not in ACT:
            raise ValueError(f"unsupported activation {a!r}; supported: {sorted(ACT)}")
        d.act[i] = ACT[a]
        d.l2_kernel[i] = float(l2_kernel[i]) if l2_kernel and l2_kernel[i] else 0.0
        d.l2_bias[i] = float(l2_bias[i]) if l2_bias and l2_bias[i] else 0.0
    d.compute = COMPUTE[compute]
    return d


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("bore_amd needs a ROCm GPU (MI355X / gfx950): torch.cuda.is_available() "
                           "is False and there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a contiguous torch tensor (None -> NULL)."""
    if t is None:
        return C.c_void_p(0)
    assert t.is_cuda and t.is_contiguous()
    return C.c_void_p(t.data_ptr())
