"""CPU suite for the boundary: libpnec_hip.so loads without a GPU, exports every symbol that
include/pnec_hip.h declares, agrees with the oracle on enum values and option defaults, and the
product path fails loudly (no CPU fallback) when no device is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from pnec_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    L = capi.lib()
    assert L.pnec_hip_abi_version() == capi.ABI_VERSION
    header = open(os.path.join(ROOT, "include", "pnec_hip.h")).read()
    declared = set(re.findall(r"\b(pnec_hip_[a-z_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(capi.SYMBOLS)
    for sym in declared:
        assert getattr(L, sym) is not None


def test_no_oracle_or_fallback_in_product_package():
    """the product package must never import / call the oracle or a CPU fallback"""
    pkg = os.path.join(ROOT, "pnec_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", ".inl")):
                text = open(os.path.join(dirpath, f)).read()
                assert "pnec_oracle" not in text, f
                assert "import oracle" not in text and "from oracle" not in text, f


def test_option_defaults_match_ceres_defaults_and_oracle(oracle):
    o = capi.default_options()
    r = oracle.default_options()
    for name in ("max_num_iterations", "max_num_consecutive_invalid_steps", "jacobi_scaling",
                 "check_convergence", "function_tolerance", "gradient_tolerance",
                 "parameter_tolerance", "initial_trust_region_radius", "max_trust_region_radius",
                 "min_trust_region_radius", "min_relative_decrease", "min_lm_diagonal",
                 "max_lm_diagonal"):
        assert getattr(o, name) == getattr(r, name), name
    assert (o.max_num_iterations, o.function_tolerance, o.parameter_tolerance) == (50, 1e-6, 1e-8)
    assert (o.gradient_tolerance, o.initial_trust_region_radius) == (1e-10, 1e4)
    assert C.sizeof(capi.Options) == 8 * 4 + 9 * 8


def test_enum_values_shared_with_oracle(oracle):
    assert (capi.MODE_NEC, capi.MODE_TARGET, capi.MODE_HOST, capi.MODE_SYM) == \
           (oracle.MODE_NEC, oracle.MODE_TARGET, oracle.MODE_HOST, oracle.MODE_SYM)
    assert capi.TERM_NAMES == oracle.TERM_NAMES
    hdr = open(os.path.join(ROOT, "include", "pnec_hip.h")).read()
    ohdr = open(os.path.join(ROOT, "oracle", "pnec_oracle.h")).read()
    for k, v in (("FUNCTION_TOL", 0), ("PARAMETER_TOL", 1), ("GRADIENT_TOL", 2),
                 ("MAX_ITERATIONS", 3), ("MIN_RADIUS", 4), ("INVALID_STEPS", 5), ("BAD_INITIAL", 6)):
        assert re.search(rf"PNEC_HIP_TERM_{k} = {v}\b", hdr)
        assert re.search(rf"PNEC_ORACLE_TERM_{k} = {v}\b", ohdr)


def test_argument_validation_without_device():
    L = capi.lib()
    h = C.c_void_p()
    offs = np.array([0, 4, 2], dtype=np.int64)  # decreasing
    rc = L.pnec_hip_problem_create(0, capi.MODE_TARGET, 2, offs.ctypes.data, C.byref(h))
    assert rc == -1 and b"non-decreasing" in L.pnec_hip_last_error()
    offs = np.array([1, 4], dtype=np.int64)
    assert L.pnec_hip_problem_create(0, capi.MODE_TARGET, 1, offs.ctypes.data, C.byref(h)) == -1
    offs = np.array([0, 4], dtype=np.int64)
    assert L.pnec_hip_problem_create(0, 9, 1, offs.ctypes.data, C.byref(h)) == -1
    assert L.pnec_hip_solve(None, None, None, 1, None, 1e-13, None, None, None, None, None, None, 0, None) == -1
    # capacity-shaped batches and the per-frame handle: arguments are checked before any device is touched
    assert L.pnec_hip_problem_create_capacity(0, capi.MODE_TARGET, 0, 100, C.byref(h)) == -1
    assert L.pnec_hip_problem_create_capacity(0, 9, 4, 100, C.byref(h)) == -1
    assert L.pnec_hip_problem_reshape(None, 1, offs.ctypes.data, None) == -1
    assert L.pnec_hip_problem_select_view(None, None, 0, None, C.byref(h)) == -1
    assert L.pnec_hip_frame_create(0, 0, None, C.byref(h)) == -1 and b"max_corr" in L.pnec_hip_last_error()
    assert L.pnec_hip_frame_create(0, 100, None, None) == -1
    assert L.pnec_hip_frame_solve(None, 1, None, None, None, None, None, None, None, None, None, None) == -1
    assert L.pnec_hip_frame_load(None, 1, None, None, None, None) == -1
    assert L.pnec_hip_frame_capacity(None) == 0 and L.pnec_hip_frame_destroy(None) == 0
    po = capi.default_pipeline_options()
    assert (po.use_ransac, po.weighted_iterations, po.first_pair_id, po.ransac_seed) == (1, 10, 0, 1)
    assert C.sizeof(capi.PipelineOptions) == 6 * 4 + 8 + 8 + 8 + 8 + C.sizeof(capi.Options) + 2 * 4   # first_pair_id where ABI 2 had reserved[2]; ABI 5: + eigensolver_scheme, reserved (ABI 7: ransac_flags)
    assert po.eigensolver_scheme == capi.ES_NEWTON
    assert L.pnec_hip_problem_set_eigensolver_scheme(None, 1) == -1 and L.pnec_hip_problem_eigensolver_scheme(None) == 0
    assert L.pnec_hip_problem_set_ransac_flags(None, 1) == -1 and L.pnec_hip_problem_ransac_flags(None) == 0


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_fails_loudly_without_gpu():
    """No silent CPU path: creating a batch without a device is an error, not a fallback."""
    from pnec_amd import Batch
    with pytest.raises(capi.PnecHipError) as ei:
        Batch.uniform(capi.MODE_TARGET, 2, 10)
    assert ei.value.code == -2


def test_hand_written_loads_are_not_touched_between_issue_and_wait():
    """The scalar-base global loads of pnec_device.hpp are issued by one asm statement and waited for by a later one;
    the compiler may legally move a copy or a spill of the destination in between (it believes the register written
    when the issue statement returns).  tools/check_asm_loads.py reads the gfx950 code of the BUILT library and fails
    when any instruction names such a destination before the s_waitcnt that retires the load; the Makefile runs it
    after every link.  Here: the checker itself on hand-made listings (it must see a violation when there is one),
    then on the library the other tests load."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_asm_loads", os.path.join(ROOT, "tools", "check_asm_loads.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    good = """0000 <k>:
	s_mov_b64 s[4:5], s[2:3]
	global_load_dwordx2 v[28:29], v186, s[4:5]
	v_mov_b64_e32 v[30:31], 0
	s_mov_b64 s[4:5], s[2:3]
	global_load_dwordx2 v[30:31], v186, s[4:5] offset:512
	s_waitcnt vmcnt(0)
	v_add_f64 v[2:3], v[28:29], v[30:31]
	s_endpgm""".split("\n")
    assert chk.check_disassembly(good) == (2, [])
    moved = list(good)
    moved.insert(6, "\tv_mov_b32_e32 v40, v29")                     # a copy of a register whose load is in flight
    assert len(chk.check_disassembly(moved)[1]) == 1
    short = [l.replace("vmcnt(0)", "vmcnt(1)") for l in good]       # the wait leaves the second load outstanding
    assert len(chk.check_disassembly(short)[1]) == 1
    if not os.path.exists(chk.OBJDUMP):
        pytest.skip("no llvm-objdump in this image")
    seen, bad = chk.check_disassembly(chk.disassemble(capi.LIB_PATH))
    assert seen > 500 and bad == [], bad[:5]
