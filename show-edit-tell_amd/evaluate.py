"""Beam-search callers of the decode step, as the reference writes them (batch = 1 image, beam k):
`evaluate()` of `editnet.py:595-718` / `dcnet.py:405-541` and the EditNet+DCNet ensemble
`evaluate_full()` of `eval/eval xe/eval_full.py:88-218`.

They re-inline the timestep through the decoder's sub-module attributes exactly like the reference
(`decoder.embed`, `.attention_lstm`, `.caption_attention`, `.visual_attention`, `.select`,
`.copy_lstm`, `.fc`), so every call lands in a HIP kernel of libset_hip.so; the beam bookkeeping
(log-softmax, flat top-k over k*V, parent / word split, state re-indexing) stays in torch on the
device.  One fix relative to the reference text: `top_k_words // vocab_size` (the reference's `/`
yields a float index on torch >= 1.5).  COCO scoring is out of scope.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


class _EditNetBeam:
    def __init__(self, decoder, image_features, prev, prevlen, k):
        d = decoder
        self.d = d
        H, M, fh, mask = d.caption_encoder(prev, prevlen)                       # editnet.py:613
        self.X = image_features.expand(k, -1, -1).contiguous()
        self.mean = image_features.mean(1).expand(k, -1).contiguous()
        self.H, self.M = H.expand(k, -1, -1).contiguous(), M.expand(k, -1, -1).contiguous()
        self.fh, self.mask = fh.expand(k, -1).contiguous(), mask.expand(k, -1).contiguous()
        self.h1, self.c1 = d.init_hidden_state(k)
        self.h2, self.c2 = d.init_hidden_state(k)

    def step(self, words):                                                       # editnet.py:645-653
        d = self.d
        emb = d.embed(words).squeeze(1)
        self.h1, self.c1 = d.attention_lstm(torch.cat([emb, self.fh, self.h2, self.mean], 1), (self.h1, self.c1))
        attend_cap, alpha_c = d.caption_attention(self.H, self.h1, emb, self.mask)
        attend_img = d.visual_attention(self.X, self.h1)
        sel = d.select(self.M, alpha_c)
        self.h2, self.c2 = d.copy_lstm(torch.cat([self.h1, attend_cap, attend_img], 1), (self.h2, self.c2), sel)
        return d.fc(self.h2)

    def reindex(self, idx):                                                      # editnet.py:687-696
        for n in ("h1", "c1", "h2", "c2", "X", "mean", "fh", "H", "M", "mask"):
            setattr(self, n, getattr(self, n)[idx].contiguous())


class _DcnetBeam:
    def __init__(self, dae, prev, prevlen, k):
        self.d = dae
        enc, fh, mask = dae.caption_encoder(prev, prevlen)                       # dcnet.py / eval_full.py:109
        self.enc, self.fh, self.mask = (enc.expand(k, -1, -1).contiguous(), fh.expand(k, -1).contiguous(),
                                        mask.expand(k, -1).contiguous())
        self.h1, self.c1 = dae.init_hidden_state(k)
        self.h2, self.c2 = dae.init_hidden_state(k)

    def step(self, words):                                                       # eval_full.py:143-149
        d = self.d
        emb = d.embed(words).squeeze(1)
        self.h1, self.c1 = d.attention_lstm(torch.cat([emb, self.fh, self.h2], 1), (self.h1, self.c1))
        attend_cap = d.caption_attention(self.enc, self.h1, self.mask)
        self.h2, self.c2 = d.language_lstm(torch.cat([self.h1, attend_cap], 1), (self.h2, self.c2))
        return d.fc(self.h2)

    def reindex(self, idx):
        for n in ("h1", "c1", "h2", "c2", "enc", "fh", "mask"):
            setattr(self, n, getattr(self, n)[idx].contiguous())


def _beam_loop(states, combine, word_map, k, dev, max_steps=50):
    vocab_size = len(word_map)
    k_prev_words = torch.full((k, 1), int(word_map['<start>']), dtype=torch.long, device=dev)
    seqs = k_prev_words
    top_k_scores = torch.zeros(k, 1, device=dev)
    complete_seqs, complete_scores = [], []
    step = 1
    infinite_pred = False
    while True:
        scores = combine([s.step(k_prev_words) for s in states])
        scores = top_k_scores.expand_as(scores) + scores
        if step == 1:
            top_k_scores, top_k_words = scores[0].topk(k, 0, True, True)
        else:
            top_k_scores, top_k_words = scores.view(-1).topk(k, 0, True, True)
        prev_word_inds = top_k_words // vocab_size
        next_word_inds = top_k_words % vocab_size
        seqs = torch.cat([seqs[prev_word_inds], next_word_inds.unsqueeze(1)], 1)
        nxt = next_word_inds.tolist()
        incomplete = [i for i, w in enumerate(nxt) if w != word_map['<end>']]
        complete = [i for i in range(len(nxt)) if i not in incomplete]
        if complete:
            complete_seqs.extend(seqs[complete].tolist())
            complete_scores.extend(top_k_scores[complete].tolist())
        k -= len(complete)
        if k == 0:
            break
        seqs = seqs[incomplete]
        idx = prev_word_inds[incomplete]
        for s in states:
            s.reindex(idx)
        top_k_scores = top_k_scores[incomplete].unsqueeze(1)
        k_prev_words = next_word_inds[incomplete].unsqueeze(1)
        if step > max_steps:
            infinite_pred = True
            break
        step += 1
    if not infinite_pred:
        i = complete_scores.index(max(complete_scores))
        return complete_seqs[i], complete_scores[i]
    return seqs[0][:18].tolist(), float("nan")


@torch.no_grad()
def beam_search_editnet(decoder, image_features, previous_caption, prev_caplen, word_map, beam_size=3):
    """One image (1,R,F) / previous caption (1,T) / length (1,1) -> (token list incl. <start>/<end>, score)."""
    decoder.eval()
    st = _EditNetBeam(decoder, image_features, previous_caption, prev_caplen, beam_size)
    return _beam_loop([st], lambda ls: F.log_softmax(ls[0], dim=1), word_map, beam_size, image_features.device)


@torch.no_grad()
def beam_search_dcnet(dae, previous_caption, prev_caplen, word_map, beam_size=3):
    dae.eval()
    st = _DcnetBeam(dae, previous_caption, prev_caplen, beam_size)
    return _beam_loop([st], lambda ls: F.log_softmax(ls[0], dim=1), word_map, beam_size, previous_caption.device)


@torch.no_grad()
def beam_search_ensemble(decoder, dae, image_features, previous_caption, prev_caplen, word_map, beam_size=3):
    """eval_full.py:88-218: average the two models' softmax probabilities, then log (:151-153)."""
    decoder.eval()
    dae.eval()
    e = _EditNetBeam(decoder, image_features, previous_caption, prev_caplen, beam_size)
    d = _DcnetBeam(dae, previous_caption, prev_caplen, beam_size)
    comb = lambda ls: ((F.softmax(ls[0], dim=1) + F.softmax(ls[1], dim=1)) / 2).log()
    return _beam_loop([e, d], comb, word_map, beam_size, image_features.device)


def sentence(seq, word_map):
    """editnet.py:715-716"""
    rev = {v: k for k, v in word_map.items()}
    skip = {word_map['<start>'], word_map['<end>'], word_map['<pad>']}
    return ' '.join(rev[w] for w in seq if w not in skip)


# ------------------------------------------------------------------------------------------------
# the same beam search for MANY images at once (the reference is batch 1), entirely
# on the device.  Per timestep: ONE fused decode step over all NI*k hypothesis rows (`set_editnet_step`,
# plus `set_dcnet_step` for the ensemble), ONE beam epilogue kernel (`set_beam_pick_f32`: log-softmax or
# ensemble averaging, flat top-k over k*V per image, the completed/live bookkeeping of editnet.py:666-699)
# and ONE in-place state re-index (`set_beam_gather_f32`).  The host only polls "all images finished"
# every few steps.  Semantics per image are those of editnet.py:643-713: k shrinks as hypotheses emit
# <end>; the answer is the best COMPLETED hypothesis (first maximum), or seqs[0][:18] at the step limit.
# ------------------------------------------------------------------------------------------------
class _FusedModel:
    """Prologue once per image, invariants replicated k times into a (NI*k)-row workspace."""

    def _views(self, lib_ws_tensor, dims, ws, name, shape):
        import ctypes as C
        p = lib_ws_tensor(C.byref(dims), ws.data_ptr(), name.encode())
        if not p:
            raise KeyError(name)
        off = p - ws.data_ptr()
        n = 4
        for s_ in shape:
            n *= s_
        return ws[off:off + n].view(torch.float32).view(*shape)


class _FusedEditNet(_FusedModel):
    def __init__(self, decoder, X, prev, plen, k, max_steps):
        import ctypes as C
        from . import _lib
        from ._lib import check, ptr, stream_of
        lib = self.lib = _lib.load()
        dev = X.device
        NI, R, _ = X.shape
        T, D, A = prev.shape[1], decoder.decoder_dim, decoder._attention_dim
        self.st = stream_of(dev)
        d_img = decoder._dims(NI, T, R, max_steps + 1)
        self.w = decoder._weights(d_img)
        ws_img = torch.empty(lib.set_editnet_workspace_bytes(C.byref(d_img)), dtype=torch.uint8, device=dev)
        check(lib.set_editnet_begin(C.byref(self.w), C.byref(d_img), ptr(X), None, ptr(prev), ptr(plen), ptr(ws_img),
                                    ws_img.numel(), self.st), "set_editnet_begin")
        self.B = B = NI * k
        self.dims = d_b = decoder._dims(B, T, R, max_steps + 1)
        self.ws = ws_b = torch.empty(lib.set_editnet_workspace_bytes(C.byref(d_b)), dtype=torch.uint8, device=dev)
        for name, shp in (("H", (T, D)), ("M", (T, D)), ("mask", (T,)), ("att1", (R, A)), ("att1_c", (T, A)),
                          ("pre1", (4 * D,)), ("rmask", (R,))):
            self._views(lib.set_editnet_ws_tensor, d_b, ws_b, name, (B,) + shp).copy_(
                self._views(lib.set_editnet_ws_tensor, d_img, ws_img, name, (NI,) + shp).repeat_interleave(k, 0))
        self.Xk = X.repeat_interleave(k, 0).contiguous()
        self.states = [self._views(lib.set_editnet_ws_tensor, d_b, ws_b, n, (B, D)) for n in ("h1", "c1", "h2", "c2")]
        for s_ in self.states:
            s_.zero_()
        self.D = D

    def step(self, words, logits):
        import ctypes as C
        from ._lib import check, ptr
        check(self.lib.set_editnet_step(C.byref(self.w), C.byref(self.dims), ptr(self.Xk), ptr(words), 1, self.B,
                                        ptr(logits), logits.shape[1], ptr(self.ws), self.ws.numel(), self.st),
              "set_editnet_step")


class _FusedDcnet(_FusedModel):
    def __init__(self, dae, prev, plen, k, max_steps):
        import ctypes as C
        from . import _lib
        from ._lib import check, ptr, stream_of
        lib = self.lib = _lib.load()
        dev = prev.device
        NI, T = prev.shape
        D, A, Cc, _ = dae._dims_cfg
        self.st = stream_of(dev)
        d_img = dae._dims(NI, T, max_steps + 1)
        self.w = dae._weights()
        ws_img = torch.empty(lib.set_dcnet_workspace_bytes(C.byref(d_img)), dtype=torch.uint8, device=dev)
        check(lib.set_dcnet_begin(C.byref(self.w), C.byref(d_img), ptr(prev), ptr(plen), ptr(ws_img), ws_img.numel(),
                                  self.st), "set_dcnet_begin")
        self.B = B = NI * k
        self.dims = d_b = dae._dims(B, T, max_steps + 1)
        self.ws = ws_b = torch.empty(lib.set_dcnet_workspace_bytes(C.byref(d_b)), dtype=torch.uint8, device=dev)
        for name, shp in (("enc", (T, 2 * Cc)), ("final_hidden", (2 * Cc,)), ("mask", (T,)), ("att1_c", (T, A)),
                          ("pre1", (4 * D,))):
            self._views(lib.set_dcnet_ws_tensor, d_b, ws_b, name, (B,) + shp).copy_(
                self._views(lib.set_dcnet_ws_tensor, d_img, ws_img, name, (NI,) + shp).repeat_interleave(k, 0))
        self.states = [self._views(lib.set_dcnet_ws_tensor, d_b, ws_b, n, (B, D)) for n in ("h1", "c1", "h2", "c2")]
        for s_ in self.states:
            s_.zero_()
        self.D = D

    def step(self, words, logits):
        import ctypes as C
        from ._lib import check, ptr
        check(self.lib.set_dcnet_step(C.byref(self.w), C.byref(self.dims), ptr(words), 1, self.B, ptr(logits),
                                      logits.shape[1], ptr(self.ws), self.ws.numel(), self.st), "set_dcnet_step")


def _fused_beam(models, NI, k, V, word_map, dev, max_steps, poll=4):
    from . import _lib
    from ._lib import check, ptr, stream_of
    lib = _lib.load()
    st = stream_of(dev)
    start, end = int(word_map['<start>']), int(word_map['<end>'])
    B, Lmax = NI * k, max_steps + 2
    neg = float("-inf")
    scores = torch.full((NI, k), neg, device=dev)
    scores[:, 0] = 0.0                                   # step 1: all k rows are identical, only row 0 counts
    k_left = torch.full((NI,), k, dtype=torch.int32, device=dev)
    words = torch.full((B,), start, dtype=torch.long, device=dev)
    rows = torch.empty(B, dtype=torch.int32, device=dev)
    seqs = [torch.full((NI, k, Lmax), start, dtype=torch.long, device=dev) for _ in range(2)]
    best_score = torch.full((NI,), neg, device=dev)
    best_seq = torch.zeros(NI, Lmax, dtype=torch.long, device=dev)
    best_len = torch.zeros(NI, dtype=torch.int32, device=dev)
    logits = [torch.empty(B, V, dtype=torch.float32, device=dev) for _ in models]
    step = 1
    while True:
        for m, lg in zip(models, logits):
            m.step(words, lg)
        check(lib.set_beam_pick_f32(ptr(logits[0]), ptr(logits[1]) if len(models) > 1 else None, V, NI, k, V, end, step,
                                    Lmax, ptr(scores), ptr(k_left), ptr(seqs[0]), ptr(seqs[1]), ptr(best_score),
                                    ptr(best_seq), ptr(best_len), ptr(words), ptr(rows), st), "set_beam_pick_f32")
        seqs.reverse()
        for m in models:
            s0, s1, s2, s3 = m.states
            check(lib.set_beam_gather_f32(ptr(s0), ptr(s1), ptr(s2), ptr(s3), ptr(rows), NI, k, m.D, st),
                  "set_beam_gather_f32")
        if step > max_steps:
            break
        if step % poll == 0 and int(k_left.max()) == 0:      # the only host synchronisation of the search
            break
        step += 1
    seqs_c, best_c, len_c, left_c = seqs[0].cpu(), best_seq.cpu(), best_len.cpu(), k_left.cpu()
    out = []
    for i in range(NI):
        if int(left_c[i]) > 0:                               # ran into the step limit (editnet.py:702-704,711)
            out.append(seqs_c[i, 0, :18].tolist())
        else:
            out.append(best_c[i, :int(len_c[i])].tolist())
    return out


@torch.no_grad()
def beam_search_editnet_batched(decoder, image_features, previous_caption, prev_caplen, word_map, beam_size=3,
                                max_steps=50):
    """image_features (NI,R,F), previous_caption (NI,T), prev_caplen (NI,1) -> list of NI token lists."""
    decoder.eval()
    X = image_features.float().contiguous()
    prev = previous_caption.long().contiguous()
    plen = prev_caplen.reshape(-1).long().contiguous()
    m = _FusedEditNet(decoder, X, prev, plen, beam_size, max_steps)
    return _fused_beam([m], X.shape[0], beam_size, decoder.vocab_size, word_map, X.device, max_steps)


@torch.no_grad()
def beam_search_dcnet_batched(dae, previous_caption, prev_caplen, word_map, beam_size=3, max_steps=50):
    dae.eval()
    prev = previous_caption.long().contiguous()
    plen = prev_caplen.reshape(-1).long().contiguous()
    m = _FusedDcnet(dae, prev, plen, beam_size, max_steps)
    return _fused_beam([m], prev.shape[0], beam_size, dae.vocab_size, word_map, prev.device, max_steps)


@torch.no_grad()
def beam_search_ensemble_batched(decoder, dae, image_features, previous_caption, prev_caplen, word_map, beam_size=3,
                                 max_steps=50):
    """eval_full.py:88-218 for NI images at once: both models step on the same words, the epilogue averages
    their softmax probabilities."""
    decoder.eval()
    dae.eval()
    X = image_features.float().contiguous()
    prev = previous_caption.long().contiguous()
    plen = prev_caplen.reshape(-1).long().contiguous()
    e = _FusedEditNet(decoder, X, prev, plen, beam_size, max_steps)
    d = _FusedDcnet(dae, prev, plen, beam_size, max_steps)
    return _fused_beam([e, d], X.shape[0], beam_size, decoder.vocab_size, word_map, X.device, max_steps)


# ------------------------------------------------------------------------------------------------
# First version of row f2, kept as the cross-check of the fused one above (tests/test_hip_beam.py): same
# fused HIP decode step, but the beam bookkeeping is vectorised torch on the device (top-k over k*V per
# image, parent/word split, completed-hypothesis tracking, state re-indexing; ~40 small kernels and a
# host synchronisation per step).  Semantics per image are those of
# editnet.py:643-713: k shrinks as hypotheses emit <end>; the answer is the best COMPLETED
# hypothesis (first maximum), or seqs[0][:18] if the step limit is hit.
# ------------------------------------------------------------------------------------------------
@torch.no_grad()
def beam_search_editnet_batched_torch(decoder, image_features, previous_caption, prev_caplen, word_map, beam_size=3,
                                max_steps=50):
    """image_features (NI,R,F), previous_caption (NI,T), prev_caplen (NI,1) -> list of NI token lists."""
    import ctypes as C
    from . import _lib
    from ._lib import check, ptr, stream_of
    decoder.eval()
    lib = _lib.load()
    dev = image_features.device
    X = image_features.float().contiguous()
    prev = previous_caption.long().contiguous()
    plen = prev_caplen.reshape(-1).long().contiguous()
    NI, R, Fd = X.shape
    T, k, V, D = prev.shape[1], beam_size, decoder.vocab_size, decoder.decoder_dim
    A = decoder._attention_dim
    start, end = int(word_map['<start>']), int(word_map['<end>'])
    st = stream_of(dev)
    # 1. prologue once per image
    d_img = decoder._dims(NI, T, R, max_steps + 1)
    w = decoder._weights(d_img)
    ws_img = torch.empty(lib.set_editnet_workspace_bytes(C.byref(d_img)), dtype=torch.uint8, device=dev)
    check(lib.set_editnet_begin(C.byref(w), C.byref(d_img), ptr(X), None, ptr(prev), ptr(plen), ptr(ws_img),
                                ws_img.numel(), st), "set_editnet_begin")

    def view(ws, dims, name, shape):
        p = lib.set_editnet_ws_tensor(C.byref(dims), ptr(ws), name.encode())
        off = p - ws.data_ptr()
        n = 4
        for s_ in shape:
            n *= s_
        return ws[off:off + n].view(torch.float32).view(*shape)

    # 2. replicate the per-image invariants k times into the hypothesis workspace (rows i*k + j)
    B = NI * k
    d_b = decoder._dims(B, T, R, max_steps + 1)
    ws_b = torch.empty(lib.set_editnet_workspace_bytes(C.byref(d_b)), dtype=torch.uint8, device=dev)
    for name, shp in (("H", (T, D)), ("M", (T, D)), ("mask", (T,)), ("att1", (R, A)), ("att1_c", (T, A)),
                      ("pre1", (4 * D,)), ("rmask", (R,))):
        view(ws_b, d_b, name, (B,) + shp).copy_(view(ws_img, d_img, name, (NI,) + shp).repeat_interleave(k, 0))
    Xk = X.repeat_interleave(k, 0).contiguous()
    states = [view(ws_b, d_b, n, (B, D)) for n in ("h1", "c1", "h2", "c2")]
    for s_ in states:
        s_.zero_()
    # 3. search
    neg = float("-inf")
    scores = torch.full((NI, k), neg, device=dev)
    scores[:, 0] = 0.0                                   # step 1: all k rows are identical, only row 0 counts
    k_left = torch.full((NI,), k, dtype=torch.long, device=dev)
    words = torch.full((B,), start, dtype=torch.long, device=dev)
    seqs = torch.full((NI, k, 1), start, dtype=torch.long, device=dev)
    best_score = torch.full((NI,), neg, device=dev)
    best_seq = torch.zeros(NI, max_steps + 2, dtype=torch.long, device=dev)
    best_len = torch.zeros(NI, dtype=torch.long, device=dev)
    logits = torch.empty(B, V, dtype=torch.float32, device=dev)
    ar = torch.arange(k, device=dev)
    base = (torch.arange(NI, device=dev) * k).unsqueeze(1)
    infinite = torch.zeros(NI, dtype=torch.bool, device=dev)
    step = 1
    while True:
        check(lib.set_editnet_step(C.byref(w), C.byref(d_b), ptr(Xk), ptr(words), 1, B, ptr(logits), V, ptr(ws_b),
                                   ws_b.numel(), st), "set_editnet_step")
        cand = scores.unsqueeze(2) + F.log_softmax(logits, dim=1).view(NI, k, V)
        top_s, top_i = cand.view(NI, k * V).topk(k, 1, True, True)
        parent, word = top_i // V, top_i % V
        valid = ar.unsqueeze(0) < k_left.unsqueeze(1)                       # only the first k_left picks count
        is_end = valid & (word == end)
        new_seqs = torch.cat([seqs.gather(1, parent.unsqueeze(2).expand(-1, -1, seqs.shape[2])), word.unsqueeze(2)], 2)
        # completed hypotheses: first maximum over time (complete_seqs_scores.index(max(...)))
        comp = torch.where(is_end, top_s, torch.full_like(top_s, neg))
        c_best, c_arg = comp.max(1)
        upd = c_best > best_score
        if bool(upd.any()):
            L = new_seqs.shape[2]
            sel = new_seqs[torch.arange(NI, device=dev), c_arg]
            best_seq[upd, :L] = sel[upd]
            best_len[upd] = L
            best_score[upd] = c_best[upd]
        k_left = k_left - is_end.sum(1)
        live = valid & ~is_end
        order = torch.sort((~live).to(torch.int8), dim=1, stable=True)[1]   # live picks first, selection order kept
        parent, word, top_s, live = parent.gather(1, order), word.gather(1, order), top_s.gather(1, order), live.gather(1, order)
        seqs = new_seqs.gather(1, order.unsqueeze(2).expand(-1, -1, new_seqs.shape[2]))
        scores = torch.where(live, top_s, torch.full_like(top_s, neg))
        rows = (base + parent).reshape(-1)
        for s_ in states:
            s_.copy_(s_[rows])
        words = torch.where(live, word, torch.zeros_like(word)).reshape(-1).contiguous()
        if int(k_left.max()) == 0:
            break
        if step > max_steps:
            infinite = k_left > 0
            break
        step += 1
    out = []
    seqs_c, best_c, len_c, inf_c = seqs.cpu(), best_seq.cpu(), best_len.cpu(), infinite.cpu()
    for i in range(NI):
        if bool(inf_c[i]):
            out.append(seqs_c[i, 0, :18].tolist())
        else:
            out.append(best_c[i, :int(len_c[i])].tolist())
    return out
