"""GPU: boundary behaviour around the decode path:
input pipeline overlap, DCNet SCST step + DAEWithAR, module pickling / deepcopy without runtime caches,
token-table invalidation, differentiable / train-mode direct sub-module calls, id clamping."""
import copy
import io
import os

import numpy as np
import pytest
import torch

import parity
from hip_adapter import adaptive_module, dcnet_modules, editnet_modules, to_dev

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _np(t):
    return t.detach().cpu().numpy()


def test_input_pipeline_overlaps_h2d_with_decode(tmp_path):
    """Reader workers -> pinned ring -> side-stream H2D -> adaptive decode: values equal the reference collate,
    and the H2D copy of batch i+1 runs while batch i is being decoded (event timestamps)."""
    from show_edit_tell_amd import pipeline
    from test_pipeline_cpu import _reference_collate, _write_dataset
    name = "editnet_adaptive_full_b4"
    d, xe = adaptive_module(name)
    c = d["case"]
    R, F, B = c["R"], c["F"], 32
    att, fc, feats = _write_dataset(str(tmp_path), 48, F, R, seed=3)
    ids = sorted(feats)
    rng = np.random.default_rng(0)
    batches = [list(rng.choice(ids, B, replace=False)) for _ in range(10)]
    for _ in pipeline.AdaptiveFeatureReader(att, fc, batches[:2], max_regions=R, feat_dim=F, workers=8, depth=2, pin=False):
        pass                                             # warm the page cache: the test is about overlap, not disk speed
    from show_edit_tell_amd import synth
    caps, clen = synth.captions(5, B, c["V"], L=20, min_len=20)
    prev, plen = synth.prev_captions(5, B, c["T"], c["V"], 5)
    caps, clen, prev, plen = to_dev(caps), to_dev(clen), to_dev(prev), to_dev(plen)
    reader = pipeline.AdaptiveFeatureReader(att, fc, batches, max_regions=R, feat_dim=F, workers=8, depth=4)
    pf = pipeline.DevicePrefetcher(reader, DEV, depth=2, record_timing=True)
    base = torch.cuda.Event(enable_timing=True)
    base.record()
    spans = []
    with torch.no_grad():
        for k, (images, means) in enumerate(pf):
            assert images.is_cuda and images.dtype == torch.float32
            if k == 0:
                want_i, want_m = _reference_collate(feats, batches[0], R, F)
                assert np.array_equal(_np(images), want_i) and np.array_equal(_np(means), want_m)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            pred, *_ = xe(images, means, caps, clen, prev, plen, False, 0.0)
            e.record()
            spans.append((s, e))
    torch.cuda.synchronize()
    assert len(spans) == len(batches) == len(pf.timeline)
    decode = [(base.elapsed_time(s), base.elapsed_time(e)) for s, e in spans]
    copies = [(base.elapsed_time(s), base.elapsed_time(e)) for s, e in pf.timeline]
    overlapped = 0
    for i in range(len(batches) - 1):
        for j in range(i + 1, len(batches)):             # a later batch's copy inside decode i
            lo, hi = max(decode[i][0], copies[j][0]), min(decode[i][1], copies[j][1])
            if hi > lo:
                overlapped += 1
                break
    assert overlapped >= 1, (decode, copies)
    assert torch.isfinite(pred).all()


def test_dcnet_scst_step_and_dae_with_ar():
    """dcnet_rl.py:348-361,451-493: DAEWithAR keeps the reference's state_dict layout (dae.*, affine_hidden.*), its
    forward delegates to the DAE, and the text-only SCST step updates the DAE with finite numbers."""
    from show_edit_tell_amd import ciderd
    from show_edit_tell_amd.dcnet_rl import DAEWithAR
    from show_edit_tell_amd.train import dcnet_scst_train_step
    d, xe, rl = dcnet_modules("dcnet_small")
    wm = d["wm"]
    model = DAEWithAR(dae=rl).to(DEV)
    keys = set(model.state_dict())
    assert {"affine_hidden.weight", "affine_hidden.bias", "dae.fc.weight", "dae.attention_lstm.weight_ih",
            "dae.language_lstm.weight_hh", "dae.caption_encoder.lstm_encoder.weight_ih_l0_reverse",
            "dae.caption_encoder.concat.weight", "dae.caption_attention.cap_full_att.weight",
            "dae.embed.embedding.weight"} <= keys
    assert all(k.startswith(("dae.", "affine_hidden.")) for k in keys)
    prev, plen = to_dev(d["prev"]), to_dev(d["plen"])
    g = parity.load("dcnet_small")
    model.eval()
    with torch.no_grad():
        seq, logp = model(wm, prev, plen, sample_max=True, sample_rl=False)
    parity.check_greedy(_np(seq), _np(logp), g)
    B, V = prev.shape[0], len(wm)
    rng = np.random.default_rng(3)
    allcaps = np.zeros((B, 5, 12), dtype=np.int64)
    for b in range(B):
        for j in range(5):
            n = int(rng.integers(3, 9))
            allcaps[b, j, 0] = wm["<start>"]
            allcaps[b, j, 1:1 + n] = rng.integers(1, V - 4, n)
            allcaps[b, j, 1 + n] = wm["<end>"]
    gt = ciderd.ground_truth_lists(allcaps, wm)
    df, docs = ciderd.document_frequency([[ciderd.tokens_to_str(c) for c in caps] for caps in gt])
    scorer = ciderd.CiderD(df, max(docs, 2))
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    torch.manual_seed(4)
    for n_samples in (1, 5):
        reward, loss = dcnet_scst_train_step(model, opt, wm, prev, plen, gt, scorer, n_samples=n_samples)
        assert np.isfinite(reward) and np.isfinite(loss)
    after = model.state_dict()
    assert any(not torch.equal(before[k], after[k]) for k in before if k.startswith("dae."))
    assert torch.equal(before["affine_hidden.weight"], after["affine_hidden.weight"])      # no gradient from this loss
    assert all(torch.isfinite(v).all() for v in after.values())


def test_modules_pickle_and_deepcopy_without_runtime_caches():
    """The reference checkpoints pickle whole modules (editnet.py:168-175, dcnet.py:131-138): workspaces, the token
    table and the last autograd graph must not travel; copies must decode identically."""
    d, xe, rl = editnet_modules("editnet_small")
    wm = d["wm"]
    prev, plen, X = to_dev(d["prev"]), to_dev(d["plen"]), to_dev(d["X"])
    with torch.no_grad():
        for _ in range(3):                                # third call runs with the token table built
            seq, logp = rl(wm, prev, plen, X, True, False)
    assert rl.__dict__.get("_tok_state", {}).get("table") is not None and rl._ws is not None
    pred, *_ = xe(X, to_dev(d["caps"]), to_dev(d["clen"]), prev, plen, False, 0.0)      # grad-enabled forward
    assert pred.requires_grad
    buf = io.BytesIO()
    torch.save({"decoder": rl, "decoder_xe": xe}, buf)
    assert buf.tell() < 3 * sum(p.numel() * 4 for p in rl.parameters()) + (1 << 20)
    buf.seek(0)
    back = torch.load(buf, weights_only=False)
    rl2, rl3 = back["decoder"], copy.deepcopy(rl)
    copy.deepcopy(xe)
    for m in (rl2, rl3):
        assert m._ws is None and "_tok_state" not in m.__dict__ and "_ws_cache" not in m.__dict__
        with torch.no_grad():
            s2, l2 = m(wm, prev, plen, X, True, False)
        assert torch.equal(s2, seq) and torch.allclose(l2, logp, atol=1e-5)
    dd, dxe, drl = dcnet_modules("dcnet_small")
    with torch.no_grad():
        sd, ld = drl(dd["wm"], to_dev(dd["prev"]), to_dev(dd["plen"]), True, False)
    buf = io.BytesIO()
    torch.save({"dae": drl}, buf)
    buf.seek(0)
    for m in (torch.load(buf, weights_only=False)["dae"], copy.deepcopy(drl)):
        assert m._ws is None and m.caption_encoder._owner() is m
        with torch.no_grad():
            s2, _ = m(dd["wm"], to_dev(dd["prev"]), to_dev(dd["plen"]), True, False)
        assert torch.equal(s2, sd)


def test_token_table_invalidation():
    """The folded token table must never serve stale weights: autograd-visible updates, train()/eval() switches,
    load_state_dict and the documented invalidate_token_table() after `.data` writes."""
    d, xe, rl = editnet_modules("editnet_small")
    wm = d["wm"]
    prev, plen, X = to_dev(d["prev"]), to_dev(d["plen"]), to_dev(d["X"])

    def decode():
        with torch.no_grad():
            return rl(wm, prev, plen, X, True, False)

    def table():
        return rl.__dict__.get("_tok_state", {}).get("table")

    for _ in range(3):
        base_seq, base_logp = decode()
    assert table() is not None
    rl.train()
    rl.eval()
    assert table() is None                              # mode switch drops it
    for _ in range(3):
        decode()
    assert table() is not None
    # a .data write (invisible to tensor._version) + the documented call
    with torch.no_grad():
        rl.embed.embedding.weight.data.mul_(-1.0)
    rl.invalidate_token_table()
    s1, l1 = decode()
    for _ in range(2):
        s2, l2 = decode()                               # rebuilt table: must describe the NEW weights
    assert torch.equal(s1, s2) and torch.allclose(l1, l2, atol=1e-5)
    assert not torch.equal(s1, base_seq)
    # SET_TOKEN_TABLE_VERIFY catches the undocumented case
    with torch.no_grad():
        rl.embed.embedding.weight.data.mul_(-1.0)
    os.environ["SET_TOKEN_TABLE_VERIFY"] = "1"
    try:
        with pytest.raises(Exception, match="stale"):
            decode()
    finally:
        os.environ.pop("SET_TOKEN_TABLE_VERIFY")
    rl.load_state_dict(rl.state_dict())
    assert table() is None
    s3, _ = decode()
    assert torch.equal(s3, base_seq)


def test_direct_submodule_calls_are_differentiable_and_train_mode_works():
    """evaluate()-style direct use of the sub-modules under autograd (the reference's modules are ordinary
    differentiable nn.Modules) and in train mode (dropout sites) — formerly NotImplementedError."""
    d, xe, rl = editnet_modules("editnet_small")
    prev, plen, X = to_dev(d["prev"]), to_dev(d["plen"]), to_dev(d["X"])
    B, D = X.shape[0], xe.decoder_dim
    xe.eval()
    with torch.no_grad():
        H0, M0, fh0, mask0 = xe.caption_encoder(prev, plen)
        emb0 = xe.embed(prev[:, 0])
    H, M, fh, mask = xe.caption_encoder(prev, plen)            # grad enabled, eval mode
    assert H.requires_grad and torch.allclose(H, H0, atol=2e-5) and torch.equal(mask, mask0)
    emb = xe.embed(prev[:, 0])
    assert emb.requires_grad and torch.equal(emb.detach(), emb0)
    h1, c1 = xe.init_hidden_state(B)
    h2, c2 = xe.init_hidden_state(B)
    h1, c1 = xe.attention_lstm(torch.cat([emb, fh, h2, X.mean(1)], 1), (h1, c1))
    cap, alpha = xe.caption_attention(H, h1, emb, mask)
    img = xe.visual_attention(X, h1)
    sel = xe.select(M, alpha)
    h2, c2 = xe.copy_lstm(torch.cat([h1, cap, img], 1), (h2, c2), sel)
    logits = xe.fc(h2)
    assert logits.requires_grad
    logits.logsumexp(1).sum().backward()
    for k, p in xe.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), k
    xe.train()                                                  # dropout active in embed / att_embed
    e1, e2 = xe.embed(prev[:, 0]), xe.embed(prev[:, 0])
    assert not torch.equal(e1, e2) and float((e1 == 0).float().mean()) > 0.4
    v1 = xe.visual_attention(X, h1.detach())
    assert torch.isfinite(v1).all()
    Ht, *_ = xe.caption_encoder(prev, plen)
    assert torch.isfinite(Ht).all()


def test_select_soft_matches_reference_branch():
    """SelectC.forward(soft=True) (/root/reference/editnet.py:419-420) — the branch the reference's loops never take but
    a drop-in must still compute: output (no-grad kernel and autograd route) and both input gradients against the
    reference's own class (tests/golden/boundary_ops.npz, oracle/make_boundary_golden.py)."""
    from oracle.make_boundary_golden import boundary_inputs
    g = parity.load("boundary_ops")
    d, xe, rl = editnet_modules("editnet_small")
    prev, plen = to_dev(d["prev"]), to_dev(d["plen"])
    with torch.no_grad():
        _, M, _, _ = xe.caption_encoder(prev, plen)
    p = boundary_inputs(d, M.shape[1])
    alpha = to_dev(p["alpha"])
    with torch.no_grad():
        sel0 = xe.select(M, alpha, soft=True)
    parity.assert_close(_np(sel0), g["soft_sel"], 1e-5, "soft selection (no-grad)")
    Mg, ag = M.clone().requires_grad_(True), alpha.clone().requires_grad_(True)
    sel = xe.select(Mg, ag, soft=True)
    assert sel.requires_grad and torch.equal(sel.detach(), sel0)
    sel.backward(to_dev(p["dsel"]))
    parity.assert_close(_np(Mg.grad), g["soft_dM"], 1e-5, "soft selection dM")
    parity.assert_close(_np(ag.grad), g["soft_dalpha"], 2e-5, "soft selection dalpha")
    # the hard branch is untouched: same call without the flag still takes the arg-max row
    with torch.no_grad():
        hard = xe.select(M, alpha)
    j = alpha.argmax(1)
    assert torch.allclose(hard, M[torch.arange(M.shape[0]), j], atol=1e-6)


def test_adaptive_visual_attention_direct_call_is_differentiable():
    """A differentiable DIRECT call of the adaptive VisualAttentionC (/root/reference/adaptive_features/
    editnet_adaptive.py:438-457) — formerly NotImplementedError: context, d/d(decoder_hidden) and every parameter
    gradient of the sub-module against the reference's autograd (eval mode, ragged region counts)."""
    from oracle.make_boundary_golden import boundary_inputs
    g = parity.load("boundary_ops")
    d, xe = adaptive_module("editnet_adaptive_small")
    va = xe.visual_attention
    X = to_dev(d["X"])
    p = boundary_inputs(d, 1)
    with torch.no_grad():
        ctx0 = va(X, to_dev(d["probe"]["h1"]))
    parity.assert_close(_np(ctx0), g["ada_ctx"], 2e-5, "adaptive visual context (no-grad)")
    h1 = to_dev(d["probe"]["h1"]).requires_grad_(True)
    va.zero_grad()
    ctx = va(X, h1)
    assert ctx.requires_grad
    parity.assert_close(_np(ctx), g["ada_ctx"], 2e-5, "adaptive visual context (autograd route)")
    ctx.backward(to_dev(p["dctx"]))
    parity.assert_close(_np(h1.grad), g["ada_dh1"], 5e-5, "d context / d decoder_hidden")
    for k, q in va.named_parameters():
        want = g["ada_grad." + k]
        tol = 1e-4 * max(1.0, float(np.abs(want).max()))
        parity.assert_close(_np(q.grad), want, tol, "grad " + k)
    va.train()                                                 # dropout site active: finite, and padded regions stay masked
    v1 = va(X, h1.detach())
    assert torch.isfinite(v1).all()
    va.eval()


def test_out_of_range_token_ids_are_clamped_with_and_without_token_table():
    """ADVICE: the table gathers must clamp ids exactly like the embedding gather does, so behaviour does not depend
    on whether the token table is active (no out-of-bounds device read)."""
    d, xe, rl = editnet_modules("editnet_small")
    wm, V = d["wm"], d["case"]["V"]
    prev, plen, X = to_dev(d["prev"]).clone(), to_dev(d["plen"]), to_dev(d["X"])
    prev[0, 0] = V + 1000                                     # clamps to V-1
    prev[1, 0] = -5 if int(plen[1]) > 0 else prev[1, 0]      # clamps to 0
    outs = []
    with torch.no_grad():
        for _ in range(3):                                    # calls 1-2 gather embeddings, call 3 uses the table
            outs.append(rl(wm, prev, plen, X, True, False))
    assert rl.__dict__.get("_tok_state", {}).get("table") is not None
    assert torch.equal(outs[0][0], outs[2][0]) and torch.allclose(outs[0][1], outs[2][1], atol=1e-5)
    ok = prev.clone()
    ok[0, 0] = V - 1
    ok[1, 0] = 0 if int(plen[1]) > 0 else ok[1, 0]
    with torch.no_grad():
        ref = rl(wm, ok, plen, X, True, False)
    assert torch.equal(ref[0], outs[2][0])


def test_persistent_encoder_under_concurrent_streams():
    """The weights-stationary persistent encoder (csrc/encoder_persistent.hip: one launch, grid barrier per step) while
    other decodes run on other streams: instances are serialised by the library's event chain, the kernels of the other
    streams finish on their own, so nothing may hang, no barrier may time out (status word stays 0) and every stream's
    tokens equal the single-stream decode."""
    d, xe, rl = editnet_modules("editnet_full_b4")          # (the persistent encoder is the default up to 32 rows)
    wm = d["wm"]
    prev, plen, X = to_dev(d["prev"]), to_dev(d["plen"]), to_dev(d["X"])
    with torch.no_grad():
        rl(wm, prev, plen, X, True, False)
        ref, ref_lp = rl(wm, prev, plen, X, True, False)
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream() for _ in range(8)]
        outs = []
        for rep in range(4):
            for s in streams:
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    outs.append(rl(wm, prev, plen, X, True, False))
        torch.cuda.synchronize()
    for seq, lp in outs:
        assert torch.equal(seq, ref)
        assert float((lp - ref_lp).abs().max()) < 1e-5
    # the status word behind the barrier words of every workspace that ran the encoder
    dims = rl._dims(X.shape[0], prev.shape[1], X.shape[1], rl.max_len + 1)
    for ws in rl._ws_cache.values():
        rl._ws = ws
        bar = rl.ws_tensor(dims, "enc_bar", (18 * 32,), dtype=torch.int32)
        assert int(bar[17 * 32]) == 0, "a grid barrier of the persistent encoder timed out"


_FAULT_SCRIPT = r"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.getcwd())
import parity
from hip_adapter import editnet_modules, to_dev
from show_edit_tell_amd import _lib
d, xe, rl = editnet_modules("editnet_full_b4")
g = parity.load("editnet_full_b4")
args = (d["wm"], to_dev(d["prev"]), to_dev(d["plen"]), to_dev(d["X"]), True, False)
mode = sys.argv[1]
with torch.no_grad():
    if mode == "stall":
        seq, logp = rl(*args)                      # workgroup 0 never arrives: every barrier wait times out
        torch.cuda.synchronize()
        # every row's first log-prob is NaN and its word 0 (= finished): the reference's break then ends the decode
        assert torch.isnan(logp[:, 0]).all() and not seq.any(), "a timed-out persistent encoder must poison the decode"
        try:
            rl(*args)
            raise SystemExit("the call after a barrier timeout must raise SetError")
        except _lib.SetError as e:
            assert "code 5" in str(e), str(e)
    for _ in range(3):                             # from now on the per-step kernels: parity with the golden
        seq, logp = rl(*args)
        torch.cuda.synchronize()
        parity.check_greedy(seq.cpu().numpy(), logp.cpu().numpy(), g)
    lib = _lib.load()
    lib.set_profile_enable(1)
    rl(*args); torch.cuda.synchronize()
    names = [r["tag"] for r in _lib.profile_report()]
    lib.set_profile_enable(0)
    assert "persistent_encoder" not in names, names
print("OK", mode)
"""


@pytest.mark.parametrize("mode,env", [("stall", {"SET_PENC_TEST_STALL": "1", "SET_PENC_TIMEOUT_US": "20000"}),
                                      ("capacity", {"SET_PENC_TEST_CAPACITY": "100"})])
def test_persistent_encoder_failure_is_loud(mode, env, tmp_path):
    """The persistent caption encoder can fail in two ways and neither may be silent (csrc/encoder_persistent.hip):
    * `capacity`: the device does not admit the whole grid at once (occupancy query x CU count < workgroups; forced here
      with a test hook) -> the launch is refused up front and the per-step kernels run: results equal the golden;
    * `stall`: a workgroup never reaches the grid barrier (test hook; stands for a workgroup that is not resident) -> the
      bounded waits time out, the kernel overwrites its outputs with NaN (the decode returns NaN log-probs, never plausible
      numbers), the NEXT call raises SetError(SET_ERR_FAULT) once, and the library keeps to the per-step kernels after that."""
    import subprocess, sys, os
    e = dict(os.environ)
    e.update(env)
    e["SET_PERSISTENT_LOCK_DIR"] = str(tmp_path)       # this pytest process owns the device's persistent launches; the child gets its own lock
    r = subprocess.run([sys.executable, "-c", _FAULT_SCRIPT, mode], env=e, capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and ("OK " + mode) in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


_TWO_PROC_SCRIPT = r"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.getcwd())
import parity
from hip_adapter import editnet_modules, to_dev
from show_edit_tell_amd import _lib
rank, rdv = int(sys.argv[1]), sys.argv[2]
d, xe, rl = editnet_modules("editnet_full_b4")
g = parity.load("editnet_full_b4")
args = (d["wm"], to_dev(d["prev"]), to_dev(d["plen"]), to_dev(d["X"]), True, False)
lib = _lib.load()
with torch.no_grad():
    if rank == 1:                                  # rank 0 warms up first: it is the process that takes the device's lock
        while not os.path.exists(os.path.join(rdv, "ready0")): time.sleep(0.01)
    for _ in range(3):                             # (the third call runs with the token table: persistent launch if allowed)
        rl(*args)
    torch.cuda.synchronize()
    open(os.path.join(rdv, "ready%d" % rank), "w").close()
    while not (os.path.exists(os.path.join(rdv, "ready0")) and os.path.exists(os.path.join(rdv, "ready1"))): time.sleep(0.005)
    lib.set_profile_enable(1)
    worst = 0.0
    for i in range(60):                            # both processes decode at the same time on the one device
        t0 = time.perf_counter()
        seq, logp = rl(*args)
        torch.cuda.synchronize()
        worst = max(worst, time.perf_counter() - t0)
        if i % 10 == 0:
            parity.check_greedy(seq.cpu().numpy(), logp.cpu().numpy(), g)
    tags = sorted(set(r["tag"] for r in _lib.profile_report()))
    lib.set_profile_enable(0)
print("RESULT " + json.dumps(dict(rank=rank, worst_s=worst, persistent=any(t.startswith("persistent") for t in tags), tags=tags)))
"""


def test_two_processes_on_one_device_never_interleave_persistent_grids(tmp_path):
    """the persistent launches (every workgroup must be resident) were guarded per PROCESS only — two
    processes sharing a GPU could each get half a grid resident and sit in the 1-s time-out.  Now the first process owns the
    device's persistent launches (advisory lock, csrc/encoder_persistent.hip penc_process_owns); the other one is answered
    SET_ERR_UNSUPPORTED and runs the per-step kernels.  Two processes decode concurrently on cuda:0: both match the
    reference's golden, exactly one of them ran persistent kernels, nobody waited anywhere near the time-out, no fault."""
    import json, subprocess, sys, os
    e = dict(os.environ)
    e["SET_PERSISTENT_LOCK_DIR"] = str(tmp_path)        # (a lock an earlier test process of this run still holds is not ours)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = [subprocess.Popen([sys.executable, "-c", _TWO_PROC_SCRIPT, str(r), str(tmp_path)], env=e, cwd=root,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in (0, 1)]
    outs = [p.communicate(timeout=900) for p in procs]
    res = []
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-2000:] + se[-3000:]
        res.append(json.loads([l for l in so.splitlines() if l.startswith("RESULT ")][-1][7:]))
    res.sort(key=lambda r: r["rank"])
    assert res[0]["persistent"] and not res[1]["persistent"], res
    assert max(r["worst_s"] for r in res) < 0.5, res       # a time-out is a 1-s stall followed by SetError


def test_begin_ahead_pipelined_decode_is_bit_identical():
    """One caller, one decode after the other (the reference's evaluate() / train() pattern): DevicePrefetcher(begin_ahead=)
    runs the prologue of batch i+1 on its copy stream while batch i's timestep loop runs; the decoder call then runs
    set_editnet_greedy_begun on the prepared workspace.  Token ids and log-probs are bit-identical to the un-pipelined
    decode (and equal the reference's golden); a batch whose tensors were modified in between, or whose weights changed,
    is NOT taken from the stale prologue."""
    from show_edit_tell_amd.pipeline import DevicePrefetcher
    d, xe, rl = editnet_modules("editnet_full_b128")
    g = parity.load("editnet_full_b128")
    wm = d["wm"]
    X, prev, plen = to_dev(d["X"]), to_dev(d["prev"]), to_dev(d["plen"])
    perm = torch.from_numpy(np.random.RandomState(1).permutation(X.shape[0])).to(X.device)
    X2, prev2, plen2 = X[perm].contiguous(), prev[perm].contiguous(), plen[perm].contiguous()
    with torch.no_grad():
        for _ in range(3):
            ref = rl(wm, prev, plen, X, True, False)           # (token table active from the second call on)
        ref2 = rl(wm, prev2, plen2, X2, True, False)
        torch.cuda.synchronize()
        batches = [(X, prev, plen), (X2, prev2, plen2)] * 4
        pf = DevicePrefetcher(iter(batches), X.device, depth=2, begin_ahead=lambda b: rl.begin_ahead(b[1], b[2], b[0]))
        hits0 = rl.__dict__.get("_ahead_hits", 0)
        outs = [rl(wm, b[1], b[2], b[0], True, False) for b in pf]
        torch.cuda.synchronize()
        assert rl.__dict__.get("_ahead_hits", 0) - hits0 == len(batches), "every decode must have found its prologue done"
        assert not rl.__dict__.get("_ahead")
        for i, (seq, lp) in enumerate(outs):
            want = ref if i % 2 == 0 else ref2
            assert torch.equal(seq, want[0]) and torch.equal(lp, want[1]), "pipelined decode %d differs" % i
        parity.check_greedy(outs[0][0].cpu().numpy(), outs[0][1].cpu().numpy(), g)
        # stale prologues are not used: inputs modified in place after begin_ahead ...
        Xc = X.clone()
        rl.begin_ahead(prev, plen, Xc)
        Xc.mul_(1.0)                                           # bumps the version
        hits = rl.__dict__["_ahead_hits"]
        a = rl(wm, prev, plen, Xc, True, False)
        assert rl.__dict__["_ahead_hits"] == hits and torch.equal(a[0], ref[0])
        # ... and a weight update in between (the table is dropped, the decode runs its own prologue)
        rl.begin_ahead(prev, plen, X)
        rl.fc.bias.add_(0.0)
        hits = rl.__dict__["_ahead_hits"]
        b = rl(wm, prev, plen, X, True, False)
        assert rl.__dict__["_ahead_hits"] == hits
        assert torch.equal(b[0], ref[0])


def test_decode_ahead_keeps_decodes_in_flight_and_is_bit_identical():
    """DevicePrefetcher(begin_ahead=decoder.decode_ahead, streams=3): the caller still walks the batches one by one, three
    whole decodes are in flight on the prefetcher's side streams; every result is bit-identical to the plain call, and a
    weight update between staging and use makes the forward decode again instead of returning the stale result."""
    from show_edit_tell_amd.pipeline import DevicePrefetcher
    d, xe, rl = editnet_modules("editnet_full_b128")
    wm = d["wm"]
    X, prev, plen = to_dev(d["X"]), to_dev(d["prev"]), to_dev(d["plen"])
    perm = torch.from_numpy(np.random.RandomState(2).permutation(X.shape[0])).to(X.device)
    X2, prev2, plen2 = X[perm].contiguous(), prev[perm].contiguous(), plen[perm].contiguous()
    with torch.no_grad():
        for _ in range(3):
            ref = rl(wm, prev, plen, X, True, False)
        ref2 = rl(wm, prev2, plen2, X2, True, False)
        torch.cuda.synchronize()
        batches = [(X, prev, plen), (X2, prev2, plen2)] * 5
        pf = DevicePrefetcher(iter(batches), X.device, depth=3, streams=3,
                              begin_ahead=lambda b: rl.decode_ahead(wm, b[1], b[2], b[0]))
        hits0 = rl.__dict__.get("_ahead_hits", 0)
        outs = [rl(wm, b[1], b[2], b[0], True, False) for b in pf]
        torch.cuda.synchronize()
        assert rl.__dict__.get("_ahead_hits", 0) - hits0 == len(batches)
        for i, (seq, lp) in enumerate(outs):
            want = ref if i % 2 == 0 else ref2
            assert torch.equal(seq, want[0]) and torch.equal(lp, want[1]), "decode %d differs" % i
        rl.decode_ahead(wm, prev, plen, X)
        rl.fc.bias.add_(1.0)                                   # changes the scores: the stale result would be wrong
        hits = rl.__dict__["_ahead_hits"]
        fresh = rl(wm, prev, plen, X, True, False)
        plain = rl(wm, prev, plen, X, True, False)
        assert rl.__dict__["_ahead_hits"] == hits
        assert torch.equal(fresh[0], plain[0]) and torch.equal(fresh[1], plain[1])
        assert not torch.equal(fresh[1], ref[1])
