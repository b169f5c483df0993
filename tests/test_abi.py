"""CPU: the C-ABI library loads and exports every symbol include/set_hip.h declares, the ctypes
prototypes cover exactly that set, and argument validation fails loudly (no compute calls)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "set_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(set_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from show_edit_tell_amd import build
    build.build()
    from show_edit_tell_amd import _lib
    return _lib.load()


def test_exports_every_declared_symbol(lib):
    from show_edit_tell_amd import _lib
    assert _lib.MISSING == []
    raw = C.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 35
    for n in names:
        getattr(raw, n)                      # AttributeError = missing export
    assert sorted(_lib.PROTOTYPES) == names, set(_lib.PROTOTYPES) ^ set(names)


def test_version_and_arch(lib):
    assert lib.set_abi_version() == 1
    assert lib.set_target_arch() == b"gfx950"
    assert b"workspace" in lib.set_error_string(4)


def test_workspace_queries_and_validation(lib):
    from show_edit_tell_amd._lib import DcnetDims, EditNetDims
    d = EditNetDims(B=128, T=20, R=36, F=2048, D=1024, A=512, V=10000, maxT=19, adaptive=0)
    n = lib.set_editnet_workspace_bytes(C.byref(d))
    assert 100e6 < n < 2e9
    bad = EditNetDims(B=128, T=20, R=36, F=2048, D=1000, A=512, V=10000, maxT=19, adaptive=0)   # D % 32 != 0
    assert lib.set_editnet_workspace_bytes(C.byref(bad)) == 0
    dd = DcnetDims(B=4, T=18, D=1024, A=512, C=512, E=1024, V=10000, maxT=19)
    assert lib.set_dcnet_workspace_bytes(C.byref(dd)) > 0
    # null pointers are rejected before any HIP call
    assert lib.set_editnet_begin(None, C.byref(d), None, None, None, None, None, 0, None) == 1
    assert lib.set_linear_f32(None, 0, None, 0, None, None, 0, 1, 1, 32, 0, None, 0, None) == 1
    # round 6: the soft selection entry points validate before any HIP call too (SET_ERR_ARG = 1, SET_ERR_UNSUPPORTED = 2)
    assert lib.set_select_soft_f32(None, None, None, 4, 9, 64, None) == 1
    assert lib.set_select_soft_bwd_f32(None, None, None, None, None, 4, 9, 64, None) == 1
    one = C.c_void_p(16)                      # (never dereferenced: D % 4 != 0 is refused first)
    assert lib.set_select_soft_f32(one, one, one, 4, 9, 66, None) == 2


def test_modules_fail_loudly_on_cpu():
    import torch
    from show_edit_tell_amd import editnet_rl, synth
    wm = synth.word_map(203)
    dec = editnet_rl.DecoderC(wm, 64, 64, 64, 32, 128).eval()
    from show_edit_tell_amd._lib import SetError
    with pytest.raises(SetError):
        dec(wm, torch.zeros(2, 9, dtype=torch.long), torch.ones(2, 1, dtype=torch.long), torch.zeros(2, 7, 128))
    dec.train()
    with pytest.raises(SetError):           # train mode takes the autograd path; CPU tensors still refused
        dec(wm, torch.zeros(2, 9, dtype=torch.long), torch.ones(2, 1, dtype=torch.long), torch.zeros(2, 7, 128))


def test_state_dict_keys_match_reference_layout():
    """state_dict keys == the synthetic layout, which is checked against the reference's own
    state_dict in oracle/make_golden.py (load_state asserts equality)."""
    from show_edit_tell_amd import dcnet, editnet, synth
    wm = synth.word_map(203)
    dec = editnet.DecoderC(wm, 64, 64, 64, 32, 128)
    keys = set(dec.state_dict())
    want = set(synth.editnet_state(1, 203, 64, 32, 128)) | {"caption_encoder.embed.embedding.weight"}
    assert keys == want, keys ^ want
    for k, v in synth.editnet_state(1, 203, 64, 32, 128).items():
        assert tuple(dec.state_dict()[k].shape) == v.shape, k
    dae = dcnet.DAE(wm, None, 64, 32, 32, 64)
    want = set(synth.dcnet_state(1, 203, 64, 32, 32, 64)) | {"caption_encoder.embed.embedding.weight"}
    assert set(dae.state_dict()) == want, set(dae.state_dict()) ^ want
