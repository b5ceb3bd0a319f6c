"""The C-ABI boundary: header <-> binding <-> shared library, and the no-CPU-fallback rule.
No GPU needed (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rsx.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rsx_[a-z_0-9]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    if not os.path.exists(g.HIP_SO):
        g.build()
    import torch  # noqa: F401  (brings libamdhip64 into the process first, as rsoccer_amd._lib does)
    return ctypes.CDLL(g.HIP_SO)


def test_header_and_binding_list_the_same_symbols():
    from rsoccer_amd import _lib
    assert _declared() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol(lib):
    for name in _declared():
        assert hasattr(lib, name), f"librsx_hip.so does not export {name}"
    assert lib.rsx_abi_version() == 6


def test_header_cites_the_reference_interface():
    src = open(HEADER).read()
    for cite in ("rsim.py:116-124", "rsim.py:102", "rsim.py:105,158", "rsim.py:38", "rsim.py:50", "rsim.py:41",
                 "Entities/Field.py:5-21", "vss_gym.py", "static_defenders.py"):
        assert cite in src, cite


def test_structs_match_the_header():
    from rsoccer_amd import _lib
    src = open(HEADER).read()
    for struct, cls in (("rsx_dev_view", _lib.DevView), ("rsx_task_view", _lib.TaskView)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), src, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        names = re.findall(r"(\w+)\s*;", body)
        assert names == [f[0] for f in cls._fields_], struct


def test_no_cpu_fallback_without_a_gpu(lib):
    """Creation must fail loudly when no HIP device is visible (this container has none)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    assert lib.rsx_device_count() == 0
    h = ctypes.c_void_p()
    rc = lib.rsx_create(ctypes.byref(h), 0, 0, 3, 3, 25, 8, 0)
    assert rc == -2 and not h.value
    lib.rsx_last_error.restype = ctypes.c_char_p
    assert b"no HIP device" in lib.rsx_last_error()
    from rsoccer_amd import _lib
    with pytest.raises(_lib.RsxError):
        _lib.Sim(0, 0, 3, 3, 25, 8)
    from rsoccer_amd.vss.env_vss import VSSEnv
    with pytest.raises(_lib.RsxError):
        VSSEnv()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "rsoccer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.replace("no oracle", ""), os.path.join(dirpath, f)


def test_bench_cpu_baseline_leg_runs(oracle_mod):
    """bench.py's cpu_baseline (the only place outside tests/ and smoke() that may touch oracle/)"""
    import bench
    r = bench.cpu_baseline(64, budget_s=0.3, single_s=0.1)
    assert r["kind"] == "port" and r["cores"] >= 1 and r["value"] > 1e4 and r["single_thread_value"] > 1e4
    assert "64 envs" in r["sample"]


def test_header_is_plain_c_and_the_c_example_builds(tmp_path):
    """include/rsx.h must be usable from C (the reference-side binding could be cgo / JNI / a C
    extension): examples/rsx_c_host.c includes it and compiles with gcc -Wall -Werror."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "rsx_c_host"
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "examples", "rsx_c_host.c"), "-o", str(out), "-ldl"])
    assert out.exists()
