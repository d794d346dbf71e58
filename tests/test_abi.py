"""CPU-only: the C-ABI library loads and exports exactly what include/vlmc.h declares."""
import ctypes

import pytest

from vlmc import _lib


def test_library_exports_every_header_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    declared = _lib.header_functions()
    assert declared, "no functions parsed from include/vlmc.h"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/vlmc.h but not exported"
    assert sorted(_lib.SIGNATURES) == declared, "vlmc/_lib.py SIGNATURES out of sync with include/vlmc.h"


def test_abi_version_and_error_string():
    lib = _lib.load()
    assert lib.vlmc_abi_version() == _lib.header_abi_version()
    assert isinstance(lib.vlmc_last_error(), bytes)


def test_workspace_query_needs_no_gpu():
    lib = _lib.load()
    assert lib.vlmc_wanda_select_workspace(_lib.SEL_ROW, 5120, 2048) == 0
    assert lib.vlmc_wanda_select_partials(_lib.SEL_ROW, 5120, 2048) == 5120
    assert lib.vlmc_wanda_select_workspace(_lib.SEL_MATRIX, 6144, 1408) >= (4096 + 2048) * 4
    assert lib.vlmc_wanda_select_workspace(7, 1, 1) == 0


def test_argument_validation_fails_loudly_without_touching_the_gpu():
    lib = _lib.load()
    rc = lib.vlmc_act_sqnorm(None, _lib.BF16, 1, 1, 8, 8, 8, None, None)
    assert rc == _lib.VLMC_EINVAL and b"null pointer" in lib.vlmc_last_error()
    with pytest.raises(_lib.VlmcError):
        _lib.check(rc)


def test_ops_refuse_cpu_tensors():
    """No CPU fallback: the product path raises instead of computing on the host."""
    import torch
    from vlmc import ops
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.act_sqnorm(torch.zeros(1, 4, 8))
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.wanda_select(torch.zeros(4, 8), torch.zeros(8), "row", k=2)


def test_crosscheck_list_turns_into_the_individual_switches():
    """`VLMC_CROSSCHECK=a,b` is one spelling for the cross-check routes' individual variables; explicit ones win; typos raise."""
    from vlmc import crosscheck
    env = {"VLMC_CROSSCHECK": "gemm_staged, select_multi", "VLMC_SELECT_MIXED": "1"}
    assert crosscheck.apply(env) == ["gemm_staged", "select_multi"]
    assert env["VLMC_GEMM_RING"] == "0" and env["VLMC_MATRIX_FUSED"] == "0" and env["VLMC_SELECT_MIXED"] == "1"
    env = {"VLMC_CROSSCHECK": "all"}
    assert len(crosscheck.apply(env)) == len(crosscheck.ROUTES) and env["VLMC_BATCH_REPLAY"] == "1"
    assert crosscheck.apply({}) == []
    import pytest
    with pytest.raises(ValueError):
        crosscheck.apply({"VLMC_CROSSCHECK": "gemm_stagd"})


def test_every_environment_switch_is_registered():
    """vlmc/config.py is the ONE list of `VLMC_*` environment variables: at most ten public ones, every other one a named cross-check
    route or an internal aid.  A variable read anywhere in the tree (os.environ / getenv) that is not listed there fails here."""
    import glob
    import os
    import re

    from vlmc import config, crosscheck
    assert len(config.PUBLIC) <= 10
    routed = {k for vars_, _doc in crosscheck.ROUTES.values() for k in vars_}
    assert routed <= config.CROSSCHECK | set(config.PUBLIC), sorted(routed - config.CROSSCHECK)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    read = set()
    pats = [re.compile(r"""environ(?:\.get|\.setdefault|\.pop)?\(\s*["'](VLMC_[A-Z0-9_]+)"""), re.compile(r"""environ\[\s*["'](VLMC_[A-Z0-9_]+)"""),
            re.compile(r"""getenv\(\s*["'](VLMC_[A-Z0-9_]+)"""), re.compile(r"""setenv\(\s*["'](VLMC_[A-Z0-9_]+)""")]
    files = glob.glob(os.path.join(root, "vlm-compression_amd", "**", "*.py"), recursive=True) + \
        glob.glob(os.path.join(root, "vlm-compression_amd", "csrc", "**", "*.h*"), recursive=True) + \
        glob.glob(os.path.join(root, "vlm-compression_amd", "csrc", "**", "*.cpp"), recursive=True) + [os.path.join(root, "bench.py")]
    for f in files:
        src = open(f, errors="ignore").read()
        for p in pats:
            read |= set(p.findall(src))
    unknown = sorted(read - config.all_names())
    assert not unknown, f"VLMC_* variables read but not registered in vlmc/config.py: {unknown}"
