"""GPU parity of the operator-level entry points (the sub-module forwards the reference's beam
search calls directly, editnet.py:645-653) against the golden per-operator vectors."""
import numpy as np
import pytest
import torch

import parity
from hip_adapter import adaptive_module, editnet_modules, to_dev

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("name", ["editnet_small", "editnet_full_b4", "editnet_full_v9490"])
def test_editnet_operators(name):
    d, xe, rl = editnet_modules(name)
    c, g = d["case"], parity.load(name)
    p = {k: to_dev(v) for k, v in d["probe"].items()}
    X, prev, plen = to_dev(d["X"]), to_dev(d["prev"]), to_dev(d["plen"])
    with torch.no_grad():
        H, M, fh, mask = xe.caption_encoder(prev, plen)
        parity.assert_close(_np(H), g["enc_H"], parity.STATE_TOL, "encoder H")
        parity.assert_close(_np(M), g["enc_M"], parity.STATE_TOL, "encoder M")
        parity.assert_close(_np(fh), g["enc_final"], parity.STATE_TOL, "encoder final_hidden")
        assert np.array_equal(_np(mask), g["enc_mask"])
        emb = xe.embed(p["ids"])
        assert np.array_equal(_np(emb), g["op_embed"])
        mean = X.mean(1)
        h1, c1 = xe.attention_lstm(torch.cat([emb, fh, p["h2"], mean], 1), (p["h1"], p["c1"]))
        parity.assert_close(_np(h1), g["op_h1"], parity.STATE_TOL, "attention_lstm h")
        parity.assert_close(_np(c1), g["op_c1"], parity.STATE_TOL, "attention_lstm c")
        gated, alpha_c = xe.caption_attention(H, p["h1"], p["word"], mask)
        parity.assert_close(_np(gated), g["op_gated"], parity.STATE_TOL, "caption_attention")
        parity.assert_close(_np(alpha_c), g["op_alpha_c"], parity.STATE_TOL, "alpha_c")
        vis = xe.visual_attention(X, p["h1"])
        parity.assert_close(_np(vis), g["op_vis"], parity.STATE_TOL, "visual_attention")
        sel = xe.select(M, alpha_c)
        parity.assert_close(_np(sel), g["op_sel"], parity.STATE_TOL, "select")
        h2, c2 = xe.copy_lstm(torch.cat([p["h1"], gated, vis], 1), (p["h2"], p["c2"]), sel)
        parity.assert_close(_np(h2), g["op_h2"], parity.STATE_TOL, "copy_lstm h")
        parity.assert_close(_np(c2), g["op_c2"], parity.STATE_TOL, "copy_lstm c")
        logits = xe.fc(h2)
        hz, cz = xe.init_hidden_state(3)
        assert hz.shape == (3, c["D"]) and hz.is_cuda and not hz.any()
    if "op_logits" in g:
        parity.assert_close(_np(logits), g["op_logits"], parity.LOGIT_TOL, "fc")
    else:
        parity.check_logit_summary(_np(logits), g, "op_logits_", c["V"], what="fc")


@pytest.mark.parametrize("name", ["editnet_adaptive_small", "editnet_adaptive_full_b4"])
def test_adaptive(name):
    d, dec = adaptive_module(name)
    c, g = d["case"], parity.load(name)
    X = to_dev(d["X"])
    with torch.no_grad():
        vis = dec.visual_attention(X, to_dev(d["probe"]["h1"]))
        parity.assert_close(_np(vis), g["op_vis"], parity.STATE_TOL, "adaptive visual_attention")
        pred, caps_s, dl, sort_ind, gd_fh, last_h = dec(X, to_dev(d["image_mean"]), to_dev(d["caps"]),
                                                        to_dev(d["clen"]), to_dev(d["prev"]), to_dev(d["plen"]),
                                                        False, 0.0)
    parity.check_xe(_np(pred), dl, _np(sort_ind), g, c["V"], small=c["D"] < 1024)
    g_inv, inv = parity.unsort(g["xe_sort_ind"]), parity.unsort(_np(sort_ind))
    parity.assert_close(_np(gd_fh)[inv], g["xe_gd_final"][g_inv], parity.STATE_TOL, "gd_final_hidden")
    parity.assert_close(_np(last_h)[inv], g["xe_last_hidden"][g_inv], 5e-5, "decoder_last_hidden")


def test_linear_shapes():
    """set_linear_f32 against fp64 numpy over awkward shapes (M not a tile multiple, N ragged)."""
    from show_edit_tell_amd.editnet import _HipLinear
    rng = np.random.RandomState(0)
    for M, N, K in [(1, 7, 32), (5, 203, 64), (33, 64, 96), (65, 130, 1024), (128, 4096, 3072), (200, 1000, 512),
                    (4608, 512, 1024)]:
        lin = _HipLinear(K, N).to("cuda:0")
        x = rng.uniform(-1, 1, (M, K)).astype(np.float32)
        with torch.no_grad():
            y = lin(to_dev(x))
        ref = x.astype(np.float64) @ _np(lin.weight).astype(np.float64).T + _np(lin.bias)
        err = np.abs(_np(y) - ref).max()
        assert err < 2e-5 * max(1.0, np.sqrt(K) / 8), (M, N, K, err)


@pytest.mark.parametrize("a_kminor,b_kminor", [(False, False), (False, True), (True, True), (True, False)])
def test_gemm_general_layouts(a_kminor, b_kminor):
    """set_gemm_f32 (csrc/gemm_gen.hip): all four operand layouts, contraction tails, tile overhang,
    strided views, split-K and in-place accumulation, against float64 numpy."""
    from show_edit_tell_amd.autograd_ops import gemm
    rng = np.random.default_rng(5)
    #        M     N     K
    shapes = [(128, 1024, 4096), (132, 196, 100), (8, 64, 36), (4096, 3072, 128), (60, 1000, 2432), (4, 4, 4),
              (1024, 2048, 2436)]
    for (M, N, K) in shapes:
        a = rng.standard_normal((K, M) if a_kminor else (M, K)).astype(np.float32)
        b = rng.standard_normal((K, N) if b_kminor else (N, K)).astype(np.float32)
        ref = (a.T if a_kminor else a).astype(np.float64) @ (b if b_kminor else b.T).astype(np.float64)
        ta, tb = to_dev(a), to_dev(b)
        got = gemm(ta, a_kminor, tb, b_kminor, M, N, K)
        tol = 2e-6 * np.sqrt(K) * 4
        assert np.abs(_np(got) - ref).max() <= tol * max(1.0, np.abs(ref).max()), (M, N, K)
        c0 = rng.standard_normal((M, N)).astype(np.float32)
        out = to_dev(c0)
        gemm(ta, a_kminor, tb, b_kminor, M, N, K, out=out, accumulate=True)
        assert np.abs(_np(out) - (ref + c0)).max() <= tol * max(1.0, np.abs(ref).max()), ("acc", M, N, K)
        again = gemm(ta, a_kminor, tb, b_kminor, M, N, K)
        assert torch.equal(got, again)                       # slab order is fixed: bit-reproducible
    # operands read in place through column-slice views (how the backward reads gate_w[:, D:2D])
    M, N, K = 128, 1024, 1024
    w = to_dev(rng.standard_normal((N, 3 * K)).astype(np.float32))
    dy = to_dev(rng.standard_normal((M, N)).astype(np.float32))
    got = gemm(dy, False, w[:, K:2 * K], True, M, K, N)
    ref = _np(dy).astype(np.float64) @ _np(w)[:, K:2 * K].astype(np.float64)
    assert np.abs(_np(got) - ref).max() <= 1e-3


def test_gemm_group():
    """set_gemm_group_f32: several dX = dY.W products (k-major x k-minor) in one launch, with split-K, column-slice
    weights and in-place accumulation, against float64 numpy."""
    from show_edit_tell_amd.autograd_ops import gemm_group
    rng = np.random.default_rng(9)
    M = 128
    dy = to_dev(rng.standard_normal((M, 4096)).astype(np.float32))
    w1 = to_dev(rng.standard_normal((4096, 5120)).astype(np.float32) * 0.05)
    w2 = to_dev(rng.standard_normal((4096, 1024)).astype(np.float32) * 0.05)
    dz = to_dev(rng.standard_normal((M, 1024)).astype(np.float32))
    wg = to_dev(rng.standard_normal((1024, 3072)).astype(np.float32) * 0.05)
    o1, o2 = gemm_group([(dy, w1, M, 5120, 4096, None, False), (dy, w2, M, 1024, 4096, None, False)], False, True)
    assert np.abs(_np(o1) - _np(dy).astype(np.float64) @ _np(w1).astype(np.float64)).max() < 2e-4
    assert np.abs(_np(o2) - _np(dy).astype(np.float64) @ _np(w2).astype(np.float64)).max() < 2e-4
    base = rng.standard_normal((3, M, 1024)).astype(np.float32)
    outs = [to_dev(base[i]) for i in range(3)]
    gemm_group([(dz, wg[:, 1024 * i:1024 * (i + 1)], M, 1024, 1024, outs[i], True) for i in range(3)], False, True)
    for i in range(3):
        ref = base[i] + _np(dz).astype(np.float64) @ _np(wg)[:, 1024 * i:1024 * (i + 1)].astype(np.float64)
        assert np.abs(_np(outs[i]) - ref).max() < 1e-4, i
