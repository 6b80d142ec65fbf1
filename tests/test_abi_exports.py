"""The C-ABI library loads on a machine without a GPU and exports every symbol that
include/rtucker_hip.h declares; argument validation returns error codes without
touching a device.  (No compute calls here.)"""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rtucker_hip.h")


@pytest.fixture(scope="module")
def lib():
    from r_tucker_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):   # build in-tree (hipcc cross-compiles gfx950 without a GPU)
        subprocess.run(["bash", os.path.join(ROOT, "r-tucker_amd", "csrc", "build.sh")], check=True)
    return _lib.load()


def declared_symbols():
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(rtk_[A-Za-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from r_tucker_amd import _lib
    names = declared_symbols()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in rtucker_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in _lib.py"
    assert sorted(_lib.SIGNATURES) == names


def test_version_and_size_queries(lib):
    assert lib.rtk_version() >= 200
    # C2 shapes: tables 22*200*200*4 + v 512*200*4 + packed planes + headers
    need = lib.rtk_workspace_bytes(0, 512, 22, 10, 200, 200)
    assert 22 * 200 * 200 * 4 + 512 * 200 * 4 <= need <= 8 << 20
    assert lib.rtk_workspace_bytes(0, 0, 22, 10, 200, 200) == 0
    assert lib.rtk_packed_query_bytes(0, 512, 200) == 16 * (128 + 2 * 13 * 1024)
    assert lib.rtk_packed_query_bytes(0, 33, 20) == 2 * (128 + 2 * 2 * 1024)


def test_argument_validation_returns_codes(lib):
    null = None
    # b != c  ->  RTK_ERR_BAD_ARG with the reference's reason in the message
    rc = lib.rtk_score_1vN_f32(1, 3, 5, 7, 1, 4, 1, 10, 1, 10, 1, 1, 2, 1, 10, 1, 1, 1 << 20, null)
    assert rc == -1 and b"must equal" in lib.rtk_last_error_string()
    rc = lib.rtk_score_1vN_f32(null, 3, 5, 5, 1, 4, 1, 10, 1, 10, 1, 1, 2, 1, 10, 1, 1, 1 << 20, null)
    assert rc == -1 and b"null" in lib.rtk_last_error_string()
    # workspace too small -> RTK_ERR_WORKSPACE
    rc = lib.rtk_score_1vN_f32(256, 3, 5, 5, 256, 4, 256, 10, 256, 10, 256, 256, 2, 256, 10, 1, 256, 16, null)
    assert rc == -2 and b"workspace" in lib.rtk_last_error_string()
    rc = lib.rtk_gemm_f32(1, 1, 4, 1, 1, 4, 1, 2, 4, 4, 4, 0, null)
    assert rc == -1 and b"ldc" in lib.rtk_last_error_string()
    # kernel timer: null handles are argument errors (creating one needs a device)
    assert lib.rtk_timer_arm(null) == -1 and lib.rtk_timer_destroy(null) == -1
    assert lib.rtk_timer_elapsed_ms(null, null) == -1 and lib.rtk_timer_create(null) == -1


def test_cpu_tensors_are_rejected_loudly():
    import torch
    import r_tucker_amd as rt
    with pytest.raises(RuntimeError, match="no CPU path"):
        rt.score_1vN(torch.zeros(2, 3, 3), torch.zeros(4, 2), torch.zeros(5, 3), torch.zeros(5, 3),
                     torch.tensor([0]), torch.tensor([0]))


def test_model_surface_on_cpu():
    """Constructor / init / state_dict keys / closure protocol exist without a GPU."""
    import torch
    import r_tucker_amd as rt
    m = rt.AsymmetricR_TuckER((50, 4), (3, 5, 5), device="cpu")
    m.init()
    assert list(m.state_dict().keys()) == ["core", "S.weight", "R.weight", "O.weight"]
    for w in (m.S.weight, m.O.weight, m.R.weight):          # orthonormal columns after init (R_TuckER.py:36-39)
        assert torch.allclose(w.T @ w, torch.eye(w.shape[1]), atol=1e-5)
    s = rt.SymmetricR_TuckER((50, 4), (3, 5, 5))
    s.init()
    assert list(s.state_dict().keys()) == ["core", "E.weight", "R.weight"]
    assert callable(m(torch.tensor([1]), torch.tensor([2])))


def test_collective_entry_points_validate_without_a_gpu(lib):
    """rtk_comm_* / rtk_allgather_scores: argument errors are status codes, nothing touches a device."""
    comm = C.c_void_p()
    assert lib.rtk_comm_init(0, 0, b"\0" * 128, C.byref(comm)) == -1 and b"world" in lib.rtk_last_error_string()
    assert lib.rtk_comm_init(2, 2, b"\0" * 128, C.byref(comm)) == -1 and comm.value is None
    assert lib.rtk_comm_init(0, 1, None, C.byref(comm)) == -1 and b"unique_id" in lib.rtk_last_error_string()
    assert lib.rtk_comm_init(0, 1, b"\0" * 128, None) == -1
    assert lib.rtk_comm_unique_id(None) == -1
    assert lib.rtk_allgather_scores(None, 1, 16, None) == -1 and b"communicator" in lib.rtk_last_error_string()
    assert lib.rtk_comm_destroy(None) == 0
    assert lib.rtk_gram_factor_f64(None, 1, 4, 1, 0.0, 0.0, 1, 2, None) == -1
    assert lib.rtk_gram_factor_f64(1, 1, 300, 1, 0.0, 0.0, 2, 3, None) == -3 and b"k=300" in lib.rtk_last_error_string()
