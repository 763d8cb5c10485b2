"""ctypes binding of libpnec_hip.so (the C ABI in include/pnec_hip.h).

This is the only way Python reaches the solver: there is no CPU or PyTorch fallback.  If the
shared library is missing or cannot be loaded, importing this module's ``lib()`` raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PNEC_HIP_LIB: load an alternative build of the same ABI (kernel A/B experiments only)
LIB_PATH = os.environ.get("PNEC_HIP_LIB") or os.path.join(_HERE, "libpnec_hip.so")

ABI_VERSION = 7  # PNEC_HIP_ABI_VERSION of include/pnec_hip.h this binding was written against
MODE_NEC, MODE_TARGET, MODE_HOST, MODE_SYM = 0, 1, 2, 3
MEM_HOST, MEM_DEVICE = 0, 1
# pnec_hip_eigensolver_scheme: which iteration stands in for opengv's eigenvalue minimisation (include/pnec_hip.h)
ES_NEWTON, ES_DESCENT, ES_LM = 0, 1, 2
# PNEC_HIP_RANSAC_* bits (pnec_hip_pipeline_options.ransac_flags, pnec_hip_problem_set_ransac_flags)
RANSAC_CHAINED_STARTS = 1
# pnec_hip_status
OK, ERR_INVALID_ARGUMENT, ERR_HIP_RUNTIME, ERR_UNSUPPORTED, ERR_BUSY = 0, -1, -2, -3, -4
TERM_NAMES = {
    0: "function_tolerance",
    1: "parameter_tolerance",
    2: "gradient_tolerance",
    3: "max_iterations",
    4: "min_trust_region_radius",
    5: "invalid_steps",
    6: "bad_initial_point",
}
TERM_MAX_ITERATIONS = 3  # PNEC_HIP_TERM_MAX_ITERATIONS
NUM_COMPONENTS = {MODE_NEC: 6, MODE_TARGET: 12, MODE_HOST: 12, MODE_SYM: 18}

# every symbol include/pnec_hip.h declares (tests check the library exports all of them)
SYMBOLS = [
    "pnec_hip_abi_version",
    "pnec_hip_last_error",
    "pnec_hip_device_count",
    "pnec_hip_default_options",
    "pnec_hip_problem_create",
    "pnec_hip_problem_destroy",
    "pnec_hip_problem_fill",
    "pnec_hip_problem_fill_keypoints",
    "pnec_hip_problem_payload_doubles",
    "pnec_hip_problem_export_payload",
    "pnec_hip_problem_num_pairs",
    "pnec_hip_problem_num_correspondences",
    "pnec_hip_problem_max_correspondences",
    "pnec_hip_problem_payload_bytes",
    "pnec_hip_problem_offsets",
    "pnec_hip_problem_mode",
    "pnec_hip_problem_device",
    "pnec_hip_problem_set_eigensolver_scheme",
    "pnec_hip_problem_eigensolver_scheme",
    "pnec_hip_problem_set_ransac_flags",
    "pnec_hip_problem_ransac_flags",
    "pnec_hip_solve",
    "pnec_hip_select_best",
    "pnec_hip_cost_function",
    "pnec_hip_nec_eigensolver",
    "pnec_hip_ransac_eigensolver",
    "pnec_hip_problem_select",
    "pnec_hip_problem_select_view",
    "pnec_hip_problem_launch_order_hint",
    "pnec_hip_weighted_eigensolver",
    "pnec_hip_default_pipeline_options",
    "pnec_hip_solve_pipeline",
    "pnec_hip_partition",
    "pnec_hip_solve_pipeline_multi",
    "pnec_hip_multi_create",
    "pnec_hip_multi_destroy",
    "pnec_hip_multi_num_devices",
    "pnec_hip_multi_bounds",
    "pnec_hip_multi_fill",
    "pnec_hip_multi_solve",
    "pnec_hip_multi_solve_pipeline",
    "pnec_hip_stream_create",
    "pnec_hip_stream_destroy",
    "pnec_hip_problem_create_capacity",
    "pnec_hip_problem_reshape",
    "pnec_hip_frame_create",
    "pnec_hip_frame_destroy",
    "pnec_hip_frame_capacity",
    "pnec_hip_frame_stream",
    "pnec_hip_frame_load",
    "pnec_hip_frame_solve",
    "pnec_hip_stream_submit",
    "pnec_hip_stream_poll",
    "pnec_hip_stream_wait",
    "pnec_hip_unscented_transform",
    "pnec_hip_describe_launch",
    "pnec_hip_selftest",
    "pnec_hip_work_counters",
    "pnec_hip_release_cache",
    "pnec_hip_alloc_counters",
]


OPT_COUNT_PASSES = 1               # pnec_hip_options.flags: PNEC_HIP_OPT_COUNT_PASSES
OPT_JACOBIAN_NUMERIC_CENTRAL = 2   # ... PNEC_HIP_OPT_JACOBIAN_NUMERIC_CENTRAL (verification mode)


class Options(C.Structure):
    """``pnec_hip_options``: the ceres::Solver::Options subset + launch tuning."""

    _fields_ = [
        ("max_num_iterations", C.c_int32),
        ("max_num_consecutive_invalid_steps", C.c_int32),
        ("jacobi_scaling", C.c_int32),
        ("check_convergence", C.c_int32),
        ("corr_per_lane", C.c_int32),
        ("waves_per_pair", C.c_int32),
        ("lds_corr_per_lane", C.c_int32),
        ("flags", C.c_int32),
        ("function_tolerance", C.c_double),
        ("gradient_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double),
        ("initial_trust_region_radius", C.c_double),
        ("max_trust_region_radius", C.c_double),
        ("min_trust_region_radius", C.c_double),
        ("min_relative_decrease", C.c_double),
        ("min_lm_diagonal", C.c_double),
        ("max_lm_diagonal", C.c_double),
    ]


class PipelineOptions(C.Structure):
    """``pnec_hip_pipeline_options``: the Options fields PNEC::Solve reads."""

    _fields_ = [
        ("use_ransac", C.c_int32),
        ("use_nec", C.c_int32),
        ("use_ceres", C.c_int32),
        ("weighted_iterations", C.c_int32),
        ("max_ransac_iterations", C.c_int32),
        ("ransac_sample_size", C.c_int32),
        ("first_pair_id", C.c_int64),
        ("regularization", C.c_double),
        ("ransac_threshold", C.c_double),
        ("ransac_seed", C.c_uint64),
        ("solver", Options),
        ("eigensolver_scheme", C.c_int32),
        ("ransac_flags", C.c_int32),
    ]


class PnecHipError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libpnec_hip error {code}: {message}")
        self.code = code


_lib = None
_vp = C.c_void_p


def lib() -> C.CDLL:
    """Load libpnec_hip.so; raises (loudly) if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C pnec_amd/csrc`. "
            "There is no CPU fallback.")
    # PyTorch ships its own HIP runtime; a process that loads the system's first (through this library) and
    # PyTorch's afterwards ends up with two, and the second one finds no device ("hipSetDevice failed").
    # The package hands torch tensors to this library, so PyTorch's runtime must be the one both use:
    # load PyTorch first whenever it is installed.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    L.pnec_hip_abi_version.restype = C.c_int
    if L.pnec_hip_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} has ABI version {L.pnec_hip_abi_version()}, this binding needs "
                          f"{ABI_VERSION}: rebuild with `make -C pnec_amd/csrc`")
    L.pnec_hip_last_error.restype = C.c_char_p
    L.pnec_hip_device_count.argtypes = [C.POINTER(C.c_int)]
    L.pnec_hip_default_options.argtypes = [C.POINTER(Options)]
    L.pnec_hip_default_options.restype = None
    L.pnec_hip_problem_create.argtypes = [C.c_int, C.c_int, C.c_int64, _vp, C.POINTER(_vp)]
    L.pnec_hip_problem_destroy.argtypes = [_vp]
    L.pnec_hip_problem_fill.argtypes = [_vp, C.c_int64, C.c_int64, _vp, _vp, _vp, _vp, C.c_int, _vp]
    L.pnec_hip_problem_fill_keypoints.argtypes = [_vp, C.c_int64, C.c_int64, _vp, _vp, _vp, _vp, _vp, C.c_double,
                                                  C.c_int, C.c_int, _vp]
    L.pnec_hip_problem_export_payload.argtypes = [_vp, _vp, C.c_int, _vp]
    for name in ("num_pairs", "num_correspondences", "max_correspondences", "payload_bytes", "payload_doubles"):
        f = getattr(L, "pnec_hip_problem_" + name)
        f.argtypes = [_vp]
        f.restype = C.c_int64
    L.pnec_hip_problem_offsets.argtypes = [_vp, _vp]
    L.pnec_hip_problem_mode.argtypes = [_vp]
    L.pnec_hip_problem_device.argtypes = [_vp]
    L.pnec_hip_problem_set_eigensolver_scheme.argtypes = [_vp, C.c_int32]
    L.pnec_hip_problem_eigensolver_scheme.argtypes = [_vp]
    L.pnec_hip_problem_set_ransac_flags.argtypes = [_vp, C.c_int32]
    L.pnec_hip_problem_ransac_flags.argtypes = [_vp]
    L.pnec_hip_solve.argtypes = [_vp, _vp, _vp, C.c_int32, _vp, C.c_double, C.POINTER(Options),
                                 _vp, _vp, _vp, _vp, _vp, C.c_int, _vp]
    L.pnec_hip_select_best.argtypes = [C.c_int64, C.c_int32, _vp, _vp, C.c_int, C.c_int, _vp]
    L.pnec_hip_cost_function.argtypes = [_vp, _vp, _vp, _vp, C.c_int, _vp]
    L.pnec_hip_describe_launch.argtypes = [_vp, C.POINTER(Options)] + [C.POINTER(C.c_int32)] * 5
    L.pnec_hip_unscented_transform.argtypes = [C.c_int64, _vp, _vp, _vp, C.c_double, C.c_int, _vp, _vp,
                                               C.c_int, C.c_int, _vp]
    L.pnec_hip_nec_eigensolver.argtypes = [_vp, _vp, _vp, _vp, C.c_int, _vp]
    L.pnec_hip_ransac_eigensolver.argtypes = [_vp, _vp, C.c_uint64, C.c_int32, C.c_int32, C.c_double, _vp, _vp,
                                              _vp, _vp, _vp, C.c_int, _vp]
    L.pnec_hip_problem_select.argtypes = [_vp, _vp, C.c_int, _vp, C.POINTER(_vp)]
    L.pnec_hip_problem_select_view.argtypes = [_vp, _vp, C.c_int, _vp, C.POINTER(_vp)]
    L.pnec_hip_problem_launch_order_hint.argtypes = [_vp, C.c_int32]
    L.pnec_hip_partition.argtypes = [C.c_int64, _vp, C.c_int32, _vp]
    L.pnec_hip_work_counters.argtypes = [C.c_int, C.c_int, _vp, C.POINTER(C.c_int32)]
    L.pnec_hip_solve_pipeline_multi.argtypes = [C.c_int32, _vp, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp,
                                                C.POINTER(PipelineOptions), _vp, _vp, _vp, _vp]
    L.pnec_hip_multi_create.argtypes = [C.c_int32, _vp, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.POINTER(_vp)]
    L.pnec_hip_multi_destroy.argtypes = [_vp]
    L.pnec_hip_multi_num_devices.argtypes = [_vp]
    L.pnec_hip_multi_bounds.argtypes = [_vp, _vp]
    L.pnec_hip_multi_fill.argtypes = [_vp, C.c_int64, _vp, _vp, _vp, _vp, _vp]
    L.pnec_hip_multi_solve.argtypes = [_vp, _vp, _vp, C.c_int32, _vp, C.c_double, C.POINTER(Options), _vp, _vp, _vp, _vp, _vp]
    L.pnec_hip_multi_solve_pipeline.argtypes = [_vp, _vp, _vp, C.POINTER(PipelineOptions), _vp, _vp, _vp, _vp]
    L.pnec_hip_alloc_counters.argtypes = [_vp]
    L.pnec_hip_weighted_eigensolver.argtypes = [_vp, _vp, _vp, C.c_double, C.c_int32, _vp, _vp, C.c_int, _vp]
    L.pnec_hip_default_pipeline_options.argtypes = [C.POINTER(PipelineOptions)]
    L.pnec_hip_default_pipeline_options.restype = None
    L.pnec_hip_solve_pipeline.argtypes = [_vp, _vp, _vp, C.POINTER(PipelineOptions), _vp, _vp, _vp, _vp, C.c_int, _vp]
    L.pnec_hip_problem_create_capacity.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int64, C.POINTER(_vp)]
    L.pnec_hip_problem_reshape.argtypes = [_vp, C.c_int64, _vp, _vp]
    L.pnec_hip_frame_create.argtypes = [C.c_int, C.c_int64, _vp, C.POINTER(_vp)]
    L.pnec_hip_frame_destroy.argtypes = [_vp]
    L.pnec_hip_frame_capacity.argtypes = [_vp]
    L.pnec_hip_frame_capacity.restype = C.c_int64
    L.pnec_hip_frame_stream.argtypes = [_vp]
    L.pnec_hip_frame_stream.restype = _vp
    L.pnec_hip_frame_load.argtypes = [_vp, C.c_int64, _vp, _vp, _vp, C.POINTER(_vp)]
    L.pnec_hip_frame_solve.argtypes = [_vp, C.c_int64, _vp, _vp, _vp, _vp, _vp, C.POINTER(PipelineOptions), _vp, _vp,
                                       _vp, _vp]
    L.pnec_hip_stream_create.argtypes = [C.c_int, C.c_int32, C.c_int32, C.c_int32, _vp, C.POINTER(_vp)]
    L.pnec_hip_stream_destroy.argtypes = [_vp]
    L.pnec_hip_stream_submit.argtypes = [_vp, C.c_int, C.c_int64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_double,
                                         C.POINTER(Options), C.POINTER(C.c_int64)]
    L.pnec_hip_stream_poll.argtypes = [_vp, C.c_int64, C.POINTER(C.c_int32)]
    L.pnec_hip_stream_wait.argtypes = [_vp, C.c_int64, _vp, _vp, _vp, _vp, _vp]
    L.pnec_hip_selftest.argtypes = [C.c_int]
    L.pnec_hip_release_cache.argtypes = [C.c_int]
    L.pnec_hip_release_cache.restype = C.c_int64
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != 0:
        raise PnecHipError(rc, (lib().pnec_hip_last_error() or b"").decode())


def default_options(**overrides) -> Options:
    o = Options()
    lib().pnec_hip_default_options(C.byref(o))
    for k, v in overrides.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def default_pipeline_options(**overrides) -> PipelineOptions:
    o = PipelineOptions()
    lib().pnec_hip_default_pipeline_options(C.byref(o))
    for k, v in overrides.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


def device_count() -> int:
    n = C.c_int(0)
    rc = lib().pnec_hip_device_count(C.byref(n))
    return n.value if rc == 0 else 0
