"""Padded rows (ABI 5): handles of 786 432 envs and more keep their [rows][B] arrays (state, commands, per-env scalars) with rows
B + 16 448 floats apart (+ 65 600 from 1.5 M envs; rsx_api.hip: row_pad_for — DRAM banks); RSX_ROW_PAD=<floats> forces a pad at any batch size, which is how
every kernel family is exercised with padded rows here.  Results must not depend on the pad."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    sys.path.insert(0, ROOT)
    from rsoccer_amd import _lib as L
    return L


@pytest.mark.parametrize("cfg", [(0, 0, 3, 3, 1, "lanes"), (0, 0, 3, 3, 1, "epl"), (1, 2, 1, 6, 2, "epl"), (1, 1, 11, 11, 7, "quad")])
def test_padded_rows_change_nothing(monkeypatch, cfg):
    """the same fused run with dense and with padded rows: views report the stride, every array agrees bit for bit at every step,
    and a checkpoint taken from the padded handle continues in the dense one"""
    import torch
    L = _lib()
    kind, ft, nb, ny, task, layout = cfg
    B = 131
    monkeypatch.setenv("RSX_LAYOUT", layout)
    monkeypatch.delenv("RSX_ROW_PAD", raising=False)
    dense = L.Sim(kind, ft, nb, ny, 25, B); dense.task_attach(task, 5, 0, 0); dense.task_reset()
    monkeypatch.setenv("RSX_ROW_PAD", "200")   # rounded up to 256 floats
    padded = L.Sim(kind, ft, nb, ny, 25, B); padded.task_attach(task, 5, 0, 0); padded.task_reset()
    monkeypatch.delenv("RSX_ROW_PAD", raising=False)
    assert dense._view.row_stride == B and padded._view.row_stride == B + 256 and padded._tview.row_stride == B + 256
    assert padded.task_layout() == dense.task_layout()
    st_d, st_p = dense.state_tensor(), padded.state_tensor()
    assert st_p.shape == st_d.shape and st_p.stride() == (B + 256, 1) and st_d.stride() == (B, 1)
    td, tp = dense.task_tensors(), padded.task_tensors()
    assert tp["info"].stride() == (B + 256, 1)
    rng = np.random.default_rng(3)
    for t in range(40):
        a = torch.from_numpy(rng.uniform(-1, 1, tuple(td["actions"].shape)).astype(np.float32)).cuda()
        td["actions"].copy_(a); tp["actions"].copy_(a)
        dense.task_step(td["actions"].data_ptr()); padded.task_step(tp["actions"].data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(st_d.view(torch.int32), st_p.contiguous().view(torch.int32)), t
        for k in ("obs", "reward", "terminated", "truncated", "info", "steps"):
            x, y = td[k], tp[k].contiguous()
            assert torch.equal(x.view(torch.uint8) if x.dtype == torch.uint8 else x.view(torch.int32), y.view(torch.uint8) if y.dtype == torch.uint8 else y.view(torch.int32)), (k, t)
    blob = padded.task_checkpoint()
    assert len(blob) == len(dense.task_checkpoint())   # the blob's rows are dense whatever the handle's pad
    dense.task_restore(blob)
    for t in range(10):
        dense.task_step_n(1); padded.task_step_n(1)
    torch.cuda.synchronize()
    assert torch.equal(st_d.view(torch.int32), st_p.contiguous().view(torch.int32))
    assert np.array_equal(dense.get_state_full(), padded.get_state_full())
    assert np.array_equal(dense.read_metrics(), padded.read_metrics())
    dense.close(); padded.close()


def test_default_pad_applies_to_large_batches_only():
    L = _lib()
    small = L.Sim(0, 0, 3, 3, 25, 4096)
    assert small._view.row_stride == 4096
    small.close()
    big = L.Sim(0, 0, 3, 3, 25, 1 << 20)
    assert big._view.row_stride == (1 << 20) + 16448
    big.task_attach(1, 0, 0, 0); big.task_reset(); big.task_step_n(3)
    assert big.check_finite() == 0
    big.close()
    huge = L.Sim(0, 0, 3, 3, 25, 1 << 21)
    assert huge._view.row_stride == (1 << 21) + 65600
    huge.close()


def test_parity_suites_with_padded_rows():
    """the oracle parity tests, the graph-replay tests and the env-level tests once more, every handle with rows 320 floats longer than its batch"""
    env = dict(os.environ, RSX_ROW_PAD="320")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_gpu_parity.py", "tests/test_gpu_graph.py", "tests/test_gpu_envs.py",
                        "-k", "not full_size and not soak"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
