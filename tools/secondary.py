"""Secondary measurements of the path's other callers, as functions so that bench.py
can put them on its JSON line (`secondary` object, rank 0, N = 1) instead of leaving them builder-only numbers:
  scst      BASELINE.json configs[4]: self-critical step, B=64, 5 sampled rollouts per image, CIDEr-D reward
  adaptive  BASELINE.json configs[3]: adaptive features B=64, R=100 (10..100 valid): XE forward (eval) and training step
  dcnet     BASELINE.json configs[0] shape: DCNet / EditNet greedy decode at B=4 (and DCNet at B=128)
  beam      row f2: beam search k=3 over 128 images at once (EditNet, EditNet+DCNet ensemble)
Synthetic inputs / random-init weights of the reference's architecture (show_edit_tell_amd.synth), GPU box only."""
import time

import numpy as np
import torch

R, F, T, V, D, A = 36, 2048, 20, 10000, 1024, 512


def _timed(fn, n, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n, out


def _editnet(cls, dev, wm, end_boost=0.0):
    from show_edit_tell_amd import synth
    sd = synth.editnet_state(14, V, D, A, F, emb_scale=3.0, fc_scale=8.0, gain=3.0)
    if end_boost:
        sd["fc.bias"] = sd["fc.bias"].copy()
        sd["fc.bias"][wm["<end>"]] += end_boost
    sd["caption_encoder.embed.embedding.weight"] = sd["embed.embedding.weight"]
    m = cls(wm, D, D, D, A, F)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.to(dev)


def _dcnet(cls, dev, wm, end_boost=0.0):
    from show_edit_tell_amd import synth
    sd = synth.dcnet_state(17, V, D, A, D // 2, D, 3.0, 8.0, 3.0)
    if end_boost:
        sd["fc.bias"] = sd["fc.bias"].copy()
        sd["fc.bias"][wm["<end>"]] += end_boost
    m = cls(wm, None, D, A, D // 2, D)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    return m.to(dev)


def scst(dev, batch=64, samples=5, steps=3):
    from show_edit_tell_amd import ciderd, editnet_rl, synth
    from show_edit_tell_amd.train import scst_train_step
    wm = synth.word_map(V)
    dec = _editnet(editnet_rl.DecoderC, dev, wm)
    opt = torch.optim.Adam(dec.parameters(), lr=5e-5)
    X = torch.from_numpy(synth.features(41, batch, R, F)).to(dev)
    prev, plen = (torch.from_numpy(x).to(dev) for x in synth.prev_captions(41, batch, T, V, 5))
    rng = np.random.default_rng(41)
    allcaps = np.zeros((batch, 5, 20), dtype=np.int64)
    for b in range(batch):
        for j in range(5):
            n = int(rng.integers(6, 17))
            allcaps[b, j, 0] = wm["<start>"]
            allcaps[b, j, 1:1 + n] = rng.integers(1, V - 4, n)
            allcaps[b, j, 1 + n] = wm["<end>"]
    gt = ciderd.ground_truth_lists(allcaps, wm)
    df, docs = ciderd.document_frequency([[ciderd.tokens_to_str(c) for c in caps] for caps in gt])
    scorer = ciderd.CiderD(df, docs)
    # the step contains host work (rewards) and a host sync: time every step on its own, report the median
    step = lambda: scst_train_step(dec, opt, wm, X, prev, plen, gt, scorer, n_samples=samples)
    step()
    times = []
    for _ in range(max(steps, 5)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    t = float(np.median(times))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):                     # the training loop's protocol: steps issued back to back, one synchronisation at the end
        step()
    torch.cuda.synchronize()
    t_b2b = (time.perf_counter() - t0) / 5
    return {"workload": "SCST step (editnet_rl.py:649-686): B=%d, %d sampled rollouts per image + greedy baseline + CIDEr-D "
                        "reward + backward + clip + Adam" % (batch, samples),
            "ms_per_step": round(1e3 * t, 2), "steps_ms": [round(1e3 * x, 1) for x in times],
            "ms_per_step_back_to_back": round(1e3 * t_b2b, 2),
            "decode_steps_per_sec": round(19 * (samples + 1) / t, 1), "native_ciderd": bool(scorer._native)}


def adaptive(dev, batch=64, regions=100):
    from show_edit_tell_amd import editnet_adaptive, synth
    from show_edit_tell_amd.autograd_ops import deferred_param_grads
    from show_edit_tell_amd.optim import clip_grad_norm_and_step
    from show_edit_tell_amd.train import xe_loss_sum
    wm = synth.word_map(V)
    dec = _editnet(editnet_adaptive.DecoderC, dev, wm)
    Xn, mean_n, nvalid = synth.adaptive_features(33, batch, regions, F, 10)
    X, mean = torch.from_numpy(Xn).to(dev), torch.from_numpy(mean_n).to(dev)
    prev, plen = (torch.from_numpy(x).to(dev) for x in synth.prev_captions(33, batch, T, V, 5))
    caps, clen = (torch.from_numpy(x).to(dev) for x in synth.captions(33, batch, V, 20, 20))
    dec.eval()
    with torch.no_grad():
        t_fwd, _ = _timed(lambda: dec(X, mean, caps, clen, prev, plen, False, 0.0), 8, 3)
    opt = torch.optim.Adam(dec.parameters(), lr=5e-4)

    def train_step():
        dec.train()
        opt.zero_grad()
        pred, caps_s, dl, _, gd_fh, last_h = dec(X, mean, caps, clen, prev, plen, False, 0.0)
        ls, n, _, _ = xe_loss_sum(pred, caps_s, dl)
        loss = ls / n + torch.nn.functional.mse_loss(last_h, gd_fh)          # editnet_adaptive.py:594-596
        with deferred_param_grads():
            loss.backward()
        clip_grad_norm_and_step(list(dec.parameters()), opt, 0.25)

    t_train, _ = _timed(train_step, 4, 2)
    return {"workload": "adaptive features (editnet_adaptive.py), B=%d, R=%d (%d..%d valid regions)" % (
                batch, regions, int(nvalid.min()), int(nvalid.max())),
            "xe_forward_ms": round(1e3 * t_fwd, 3), "xe_forward_decode_steps_per_sec": round(19 / t_fwd, 1),
            "train_step_ms": round(1e3 * t_train, 2)}


def dcnet(dev):
    from show_edit_tell_amd import dcnet_rl, editnet_rl, synth
    wm = synth.word_map(V)
    out = {"workload": "greedy decode, 19 timesteps (dcnet_rl.py:286-346 / editnet_rl.py:485-549)"}
    with torch.no_grad():
        dae = _dcnet(dcnet_rl.DAE, dev, wm).eval()
        er = _editnet(editnet_rl.DecoderC, dev, wm).eval()
        # BASELINE.json configs[0] verbatim: DCNet XE (teacher-forced) forward, batch 4, captions of 20 words
        from show_edit_tell_amd import dcnet as dcnet_xe
        dxe = _dcnet(dcnet_xe.DAE, dev, wm).eval()
        prev, plen = (torch.from_numpy(x).to(dev) for x in synth.prev_captions(3, 4, T, V, 5))
        caps, clen = (torch.from_numpy(x).to(dev) for x in synth.captions(3, 4, V, 20, 20))
        t, _ = _timed(lambda: dxe(caps, clen, prev, plen), 10, 3)
        out["dcnet_xe_forward_b4_ms"] = round(1e3 * t, 3)
        for B in (4, 128):
            prev, plen = (torch.from_numpy(x).to(dev) for x in synth.prev_captions(3, B, T, V, 5))
            X = torch.from_numpy(synth.features(3, B, R, F)).to(dev)
            t, _ = _timed(lambda: dae(wm, prev, plen, True, False), 10, 3)
            out["dcnet_greedy_b%d_ms" % B] = round(1e3 * t, 3)
            if B == 4:
                t, _ = _timed(lambda: er(wm, prev, plen, X, True, False), 10, 3)
                out["editnet_greedy_b4_ms"] = round(1e3 * t, 3)
                # the same two decodes on the per-step loop (round 4: up to 8 rows the greedy loop is ONE persistent launch,
                # csrc/decode_persistent*.hip; SET_DEC_PERSISTENT=0 is read per call)
                import os
                old = os.environ.get("SET_DEC_PERSISTENT")
                os.environ["SET_DEC_PERSISTENT"] = "0"
                try:
                    t, _ = _timed(lambda: dae(wm, prev, plen, True, False), 10, 3)
                    out["dcnet_greedy_b4_per_step_loop_ms"] = round(1e3 * t, 3)
                    t, _ = _timed(lambda: dxe(caps, clen, prev, plen), 10, 3)
                    out["dcnet_xe_forward_b4_per_step_loop_ms"] = round(1e3 * t, 3)
                    t, _ = _timed(lambda: er(wm, prev, plen, X, True, False), 10, 3)
                    out["editnet_greedy_b4_per_step_loop_ms"] = round(1e3 * t, 3)
                finally:
                    if old is None:
                        del os.environ["SET_DEC_PERSISTENT"]
                    else:
                        os.environ["SET_DEC_PERSISTENT"] = old
    # BASELINE.json configs[0] (batch 4) is weight-streaming bound (265.7 MB per EditNet timestep incl.
    # 2.3 MB of activations, 160.6 MB of DCNet weights): bytes / time against the 8 TB/s HBM peak, prologue included in
    # the time (19 timesteps per decode)
    out["roofline_b4"] = {
        "bound": "hbm", "peak": 8000.0, "unit": "GB/s",
        "editnet": _stream_roof(265.7e6, out["editnet_greedy_b4_ms"]),
        "dcnet": _stream_roof(160.6e6 + 1.2e6, out["dcnet_greedy_b4_ms"]),
        "note": "algorithmic bytes per timestep (SURVEY.md 8d) x 19 / decode time incl. the prologue; the token tables fold "
                "~89 MB (EditNet) of those weights into row gathers and the persistent launch hoists 33.5 / 16 MB more, so "
                "fewer bytes actually move: `executed` = the bytes the kernels stream per timestep (PMC: "
                "profiles/r04_persistent_decode_pmc.txt) on the same clock",
        "executed": {"editnet": _stream_roof(139e6, out["editnet_greedy_b4_ms"]), "dcnet": _stream_roof(107e6, out["dcnet_greedy_b4_ms"])}}
    return out


def _stream_roof(bytes_per_timestep, ms_per_decode):
    us = 1e3 * ms_per_decode / 19.0
    gbs = bytes_per_timestep / (us * 1e-6) / 1e9
    return {"us_per_timestep": round(us, 2), "bound_us_per_timestep": round(bytes_per_timestep / 8e12 * 1e6, 2),
            "achieved": round(gbs, 1), "frac": round(gbs / 8000.0, 4)}


def dcnet_train(dev, batch=128):
    """DCNet XE training step (dcnet.py:352-402): train mode, B=128, fwd + bwd + clip + Adam"""
    from show_edit_tell_amd import dcnet as dc, synth
    from show_edit_tell_amd.train import dcnet_xe_train_step
    wm = synth.word_map(V)
    dae = _dcnet(dc.DAE, dev, wm)
    opt = torch.optim.Adam(dae.parameters(), lr=5e-4)
    prev, plen = (torch.from_numpy(x).to(dev) for x in synth.prev_captions(7, batch, T, V, 5))
    caps, clen = (torch.from_numpy(x).to(dev) for x in synth.captions(7, batch, V, 20, 20))
    t, _ = _timed(lambda: dcnet_xe_train_step(dae, opt, caps, clen, prev, plen), 5, 3)
    return {"workload": "DCNet XE training step (dcnet.py:352-402): train mode, B=%d, 19 timesteps, fwd + bwd + clip + Adam" % batch,
            "ms_per_step": round(1e3 * t, 2)}


def beam(dev, images=128, k=3):
    from show_edit_tell_amd import dcnet as dc, editnet, evaluate, synth
    wm = synth.word_map(V)
    dec = _editnet(editnet.DecoderC, dev, wm, end_boost=4.0).eval()      # captions end after ~10-20 words
    dae = _dcnet(dc.DAE, dev, wm, end_boost=4.0).eval()
    X = torch.from_numpy(synth.features(31, images, R, F)).to(dev)
    prev, plen = (torch.from_numpy(x).to(dev) for x in synth.prev_captions(31, images, T, V, 5))
    t_e, seqs = _timed(lambda: evaluate.beam_search_editnet_batched(dec, X, prev, plen, wm, k), 3, 2)   # 2 warm-ups: the token tables are built on the second call
    t_x, _ = _timed(lambda: evaluate.beam_search_ensemble_batched(dec, dae, X, prev, plen, wm, k), 3, 2)
    # the reference's own evaluate() shape: ONE image per call (editnet.py:601-613).  Persistent launch in beam mode vs the
    # per-step search (SET_DEC_PERSISTENT=0), 16 images each after two warm-up calls (token table, workspaces)
    import os

    def per_image(model=None):
        model = dec if model is None else model
        for i in range(2):
            evaluate.beam_search_editnet(model, X[i:i + 1], prev[i:i + 1], plen[i:i + 1], wm, k)
        t, out = _timed(lambda: [evaluate.beam_search_editnet(model, X[i:i + 1], prev[i:i + 1], plen[i:i + 1], wm, k) for i in range(16)], 2, 1)
        return t / 16, out
    t_one, out_one = per_image()
    # the same with captions that END (a larger <end> bias: searches finish after ~10 picks instead of running into the
    # reference's 50-step limit, which every search of the model above does)
    dec_short = _editnet(editnet.DecoderC, dev, wm, end_boost=5.5).eval()
    t_short, out_short = per_image(dec_short)
    old = os.environ.get("SET_DEC_PERSISTENT")
    os.environ["SET_DEC_PERSISTENT"] = "0"
    try:
        t_one_steps, out_steps = per_image()
        t_short_steps, _ = per_image(dec_short)
    finally:
        if old is None:
            del os.environ["SET_DEC_PERSISTENT"]
        else:
            os.environ["SET_DEC_PERSISTENT"] = old
    same = sum(int(a[0] == b[0]) for a, b in zip(out_one, out_steps))
    # the reference's PUBLISHED protocol (eval_full.py:88-237, README.md:105-108): the EditNet + DCNet ensemble, one image per
    # call — and DCNet alone (dcnet.py:405-541).  Both run the NI = 1 case of the per-step batched search (a .cpu() per pick);
    # only EditNet's own search has a persistent launch
    dae_short = _dcnet(dc.DAE, dev, wm, end_boost=5.5).eval()

    def per_image_fn(fn):
        for i in range(2):
            fn(i)
        t, out = _timed(lambda: [fn(i) for i in range(16)], 2, 1)
        return t / 16, out
    t_ens, out_ens = per_image_fn(lambda i: evaluate.beam_search_ensemble(dec, dae, X[i:i + 1], prev[i:i + 1], plen[i:i + 1], wm, k))
    t_ens_short, out_ens_short = per_image_fn(lambda i: evaluate.beam_search_ensemble(dec_short, dae_short, X[i:i + 1], prev[i:i + 1], plen[i:i + 1], wm, k))
    t_dc, _ = per_image_fn(lambda i: evaluate.beam_search_dcnet(dae, prev[i:i + 1], plen[i:i + 1], wm, k))
    t_dc_short, _ = per_image_fn(lambda i: evaluate.beam_search_dcnet(dae_short, prev[i:i + 1], plen[i:i + 1], wm, k))
    return {"workload": "beam search k=%d over %d images at once (editnet.py:595-718, eval_full.py:88-218)" % (k, images),
            "editnet_ms": round(1e3 * t_e, 2), "ensemble_ms": round(1e3 * t_x, 2),
            "editnet_one_image_per_call_ms": round(1e3 * t_one, 3), "editnet_one_image_per_call_per_step_kernels_ms": round(1e3 * t_one_steps, 3),
            "one_image_per_call_same_tokens": "%d of 16" % same,
            "one_image_per_call_mean_len": round(float(np.mean([len(o[0]) for o in out_one])), 2),
            "one_image_per_call_searches_at_step_limit": "%d of 16" % sum(int(np.isnan(o[1])) for o in out_one),
            "editnet_one_image_per_call_ending_captions_ms": round(1e3 * t_short, 3),
            "editnet_one_image_per_call_ending_captions_per_step_kernels_ms": round(1e3 * t_short_steps, 3),
            "ending_captions_mean_len": round(float(np.mean([len(o[0]) for o in out_short])), 2),
            "ending_captions_searches_at_step_limit": "%d of 16" % sum(int(np.isnan(o[1])) for o in out_short),
            "ensemble_one_image_per_call_ms": round(1e3 * t_ens, 3), "ensemble_one_image_per_call_ending_captions_ms": round(1e3 * t_ens_short, 3),
            "ensemble_one_image_per_call_mean_len": round(float(np.mean([len(o[0]) for o in out_ens])), 2),
            "ensemble_ending_captions_mean_len": round(float(np.mean([len(o[0]) for o in out_ens_short])), 2),
            "dcnet_one_image_per_call_ms": round(1e3 * t_dc, 3), "dcnet_one_image_per_call_ending_captions_ms": round(1e3 * t_dc_short, 3),
            "one_image_per_call_paths": "editnet: persistent launch (k <= 4); ensemble, dcnet: per-step kernels, NI = 1 of the batched search",
            "images_per_sec_editnet": round(images / t_e, 1), "mean_caption_len": round(float(np.mean([len(s) for s in seqs])), 2)}


def realistic_lengths(dev, batch=128, mean_len=10.5, std_len=2.4):
    """Captions that END.  Real captions take 9-10 of 18 words on average; a random-weight model never emits
    <end> at realistic times, so the finish times are IMPOSED through the per-row length cap (decoder.row_limits,
    set_decode_row_limits): row b ends at a step drawn from N(mean_len, std_len) clipped to [5, 18] (COCO caption lengths
    incl. <end>).  Greedy decode at the metric batch (a) as the reference computes it — every row for all timesteps until the
    WHOLE batch has finished, after which the loop's kernels return at once (loop gate) — and (b) with no row ever finishing
    (all 19 timesteps execute)."""
    from show_edit_tell_amd import editnet_rl, synth
    wm = synth.word_map(V)
    dec = _editnet(editnet_rl.DecoderC, dev, wm).eval()
    X = torch.from_numpy(synth.features(25, batch, R, F)).to(dev)
    prev, plen = (torch.from_numpy(x).to(dev) for x in synth.prev_captions(25, batch, T, V, 5))
    rng = np.random.default_rng(7)
    lens = np.clip(np.rint(rng.normal(mean_len, std_len, batch)), 5, 18).astype(np.int32)
    try:
        dec.row_limits = torch.from_numpy(lens).to(dev)
        with torch.no_grad():
            t0, (seq, lp) = _timed(lambda: dec(wm, prev, plen, X, True, False), 30, 5)
    finally:
        dec.row_limits = None
    with torch.no_grad():
        t_all, _ = _timed(lambda: dec(wm, prev, plen, X, True, False), 30, 5)      # no row ever finishes: all 19 timesteps
    z = seq == 0
    steps = torch.where(z.any(1), z.float().argmax(1) + 1, torch.full((batch,), seq.shape[1], device=dev)).float()
    return {"workload": "EditNet greedy decode B=%d, finish times imposed from N(%.1f, %.1f) clipped to [5, 18]: %.1f decode "
                        "steps per row on average (min %d, max %d of %d)" % (batch, mean_len, std_len, float(steps.mean()),
                                                                              int(steps.min()), int(steps.max()), seq.shape[1]),
            "ms_per_decode_all_timesteps": round(1e3 * t_all, 3),
            "ms_per_decode_reference_semantics": round(1e3 * t0, 3),
            "speedup_vs_all_timesteps": round(t_all / t0, 3),
            "note": "reference semantics = all rows decoded until the whole batch has finished (editnet_rl.py:546), after which the "
                    "loop's kernels return at once (loop gate); all_timesteps = no row ever finishes.  Round 4's opt-in "
                    "finished-row skipping (+0.9 % beyond the gate) was removed in round 5"}


def batch_sweep(dev, batches=(1, 4, 8, 16, 32, 64, 128)):
    """EditNet greedy decode (editnet_rl.py:485-549: prologue + 19 timesteps, token table active) per batch size, one decode at
    a time: ms per decode and the fraction of the binding roofline of SURVEY.md §8(d) — per timestep max(t_HBM, t_fp32) with
    263.4 MB of weights + 0.573 MB of activations per row at 8 TB/s, 0.1319 GFLOP per row at 157.3 TFLOP/s — that 19
    timesteps would take.  `path` = which loop ran (one persistent launch / the per-step kernels)."""
    from show_edit_tell_amd import _lib, editnet_rl, synth
    wm = synth.word_map(V)
    dec = _editnet(editnet_rl.DecoderC, dev, wm).eval()
    lib = _lib.load()
    rows = []
    for Bn in batches:
        X = torch.from_numpy(synth.features(25, Bn, R, F)).to(dev)
        prev, plen = (torch.from_numpy(x).to(dev) for x in synth.prev_captions(25, Bn, T, V, 5))
        with torch.no_grad():
            fn = lambda: dec(wm, prev, plen, X, True, False)
            t, _ = _timed(fn, 20, 6)          # (token table on the 2nd call, first persistent launch after it: all inside the warm-ups)
            t2, _ = _timed(fn, 20, 0)         # ... and the better of two windows: a one-off 20-ms hiccup doubled the B = 1 row once
            t = min(t, t2)
            lib.set_profile_enable(1)
            fn()
            torch.cuda.synchronize()
            tags = [r["tag"] for r in _lib.profile_report()]
            lib.set_profile_enable(0)
        t_hbm = (263.4e6 + 0.573e6 * Bn) / 8e12
        t_mma = 0.1319e9 * Bn / 157.3e12
        bound = 19 * max(t_hbm, t_mma)
        rows.append({"batch": Bn, "ms_per_decode": round(1e3 * t, 3), "decode_steps_per_sec": round(19 / t, 1),
                     "us_per_timestep_incl_prologue": round(1e6 * t / 19, 1), "bound": "hbm" if t_hbm >= t_mma else "mfma",
                     "roofline_ms": round(1e3 * bound, 3), "frac_of_roofline": round(bound / t, 3),
                     "path": "persistent" if "persistent_decode" in tags else "per-step"})
    return {"workload": "EditNet greedy decode, one batch at a time, B in %s (36x2048 feats, prev len 20, V=10000)" % (list(batches),),
            "rows": rows}


def concurrent_small_requests(dev, callers=4, rows=4, rounds=30):
    """A server with several small requests at once: `callers` independent greedy decodes of `rows` rows,
    each on its own stream.  Persistent launches are serialised process-wide (one 256-workgroup grid owns the chip), the
    per-step kernels of different streams overlap: requests per second (a) persistent, one launch per request, (b) per-step
    loop on `callers` streams, (c) the requests coalesced into ONE persistent call of callers x rows rows (<= 16)."""
    import os
    from show_edit_tell_amd import editnet_rl, synth
    wm = synth.word_map(V)
    dec = _editnet(editnet_rl.DecoderC, dev, wm).eval()
    reqs = []
    for i in range(callers):
        X = torch.from_numpy(synth.features(60 + i, rows, R, F)).to(dev)
        prev, plen = (torch.from_numpy(x).to(dev) for x in synth.prev_captions(60 + i, rows, T, V, 5))
        reqs.append((prev, plen, X))
    streams = [torch.cuda.Stream(dev) for _ in range(callers)]
    allp, alll, allx = (torch.cat([r[k] for r in reqs], 0) for k in range(3))

    def separate():
        for st, (prev, plen, X) in zip(streams, reqs):
            with torch.cuda.stream(st):
                dec(wm, prev, plen, X, True, False)

    def run(fn):
        with torch.no_grad():
            for _ in range(4):
                fn()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(rounds):
                fn()
            torch.cuda.synchronize()
        return callers * rounds / (time.perf_counter() - t)

    out = {"workload": "%d concurrent EditNet greedy requests of %d rows, each caller on its own stream" % (callers, rows)}
    with torch.no_grad():
        dec(wm, *reqs[0], True, False); dec(wm, *reqs[0], True, False)      # token table
    out["persistent_serialised_requests_per_sec"] = round(run(separate), 1)
    old = os.environ.get("SET_DEC_PERSISTENT")
    os.environ["SET_DEC_PERSISTENT"] = "0"                                  # (read per call)
    try:
        out["per_step_loop_on_streams_requests_per_sec"] = round(run(separate), 1)
    finally:
        if old is None:
            os.environ.pop("SET_DEC_PERSISTENT", None)
        else:
            os.environ["SET_DEC_PERSISTENT"] = old
    if callers * rows <= 16:
        out["coalesced_one_persistent_call_requests_per_sec"] = round(run(lambda: dec(wm, allp, alll, allx, True, False)), 1)
        # the same through the package's helper (show_edit_tell_amd/serving.py): `callers` threads submit their request and
        # wait for their rows; the coalescer's worker batches what is waiting
        import threading
        from show_edit_tell_amd import serving
        with serving.RequestCoalescer(lambda p, l, x: dec(wm, p, l, x, True, False), max_rows=16, window_s=0.0005) as co:
            def caller(i, n):
                for _ in range(n):
                    co.submit(*reqs[i]).result()
                torch.cuda.synchronize()

            def burst(n):
                ths = [threading.Thread(target=caller, args=(i, n)) for i in range(callers)]
                for t_ in ths:
                    t_.start()
                for t_ in ths:
                    t_.join()
            burst(4)
            t = time.perf_counter()
            burst(rounds)
            out["request_coalescer_requests_per_sec"] = round(callers * rounds / (time.perf_counter() - t), 1)
            out["request_coalescer_requests_per_decode"] = round(co.requests / max(co.batches, 1), 2)
    out["note"] = ("persistent launches of one process run one after the other (grid_barrier.h PersistentGuard); a server that has "
                   "several small requests at hand coalesces them into one call of up to 16 rows")
    return out


def all_secondary(dev):
    out = {}
    for name, fn in (("scst", scst), ("adaptive", adaptive), ("dcnet", dcnet), ("dcnet_train", dcnet_train), ("beam", beam),
                     ("realistic_lengths", realistic_lengths), ("batch_sweep", batch_sweep),
                     ("concurrent_small_requests", concurrent_small_requests)):
        try:
            out[name] = fn(dev)
        except Exception as e:              # a secondary figure must never break the bench line
            out[name] = {"error": repr(e)[:200]}
        torch.cuda.empty_cache()
    return out
