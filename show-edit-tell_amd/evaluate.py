"""Beam-search callers of the decode step: `evaluate()` of `editnet.py:595-718` / `dcnet.py:405-541` and the
EditNet+DCNet ensemble `evaluate_full()` of `eval/eval xe/eval_full.py:88-218`.

The reference searches ONE image at a time (batch = 1 image, beam k) and re-indexes ~11 tensors per step on
the host.  Here the search runs for MANY images at once, entirely on the device: per
timestep ONE fused decode step over all NI*k hypothesis rows, ONE beam epilogue kernel and ONE in-place state
re-index; the per-image entry points (`beam_search_editnet`, `beam_search_dcnet`, `beam_search_ensemble`,
the signatures a caller of the reference's loop needs) are the NI = 1 case of the same search.  One fix
relative to the reference text: parent = flat_index // vocab_size (the reference's `/` yields a float index on
torch >= 1.5).  Parity: tests/golden/beam_*.npz hold the outputs of the reference's own loops.
COCO scoring is out of scope.
"""
from __future__ import annotations

import torch


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
                          ("pre1", (4 * D,)), ("rmask", (R,)), ("cap_proj", (T, 2 * D)), ("mem_proj", (T, D))):
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
        self.w = dae._weights(d_img)
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


def _fused_beam(models, NI, k, V, word_map, dev, max_steps, poll=4, return_scores=False):
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
    seqs_c, best_c, len_c, left_c, score_c = seqs[0].cpu(), best_seq.cpu(), best_len.cpu(), k_left.cpu(), best_score.cpu()
    out, out_scores = [], []
    for i in range(NI):
        if int(left_c[i]) > 0:                               # ran into the step limit (editnet.py:702-704,711)
            out.append(seqs_c[i, 0, :18].tolist())
            out_scores.append(float("nan"))
        else:
            out.append(best_c[i, :int(len_c[i])].tolist())
            out_scores.append(float(score_c[i]))
    return (out, out_scores) if return_scores else out


@torch.no_grad()
def beam_search_editnet_batched(decoder, image_features, previous_caption, prev_caplen, word_map, beam_size=3,
                                max_steps=50, return_scores=False):
    """image_features (NI,R,F), previous_caption (NI,T), prev_caplen (NI,1) -> list of NI token lists
    (with return_scores: also the list of their scores; NaN where the step limit was hit)."""
    decoder.eval()
    X = image_features.float().contiguous()
    prev = previous_caption.long().contiguous()
    plen = prev_caplen.reshape(-1).long().contiguous()
    m = _FusedEditNet(decoder, X, prev, plen, beam_size, max_steps)
    return _fused_beam([m], X.shape[0], beam_size, decoder.vocab_size, word_map, X.device, max_steps,
                       return_scores=return_scores)


@torch.no_grad()
def beam_search_dcnet_batched(dae, previous_caption, prev_caplen, word_map, beam_size=3, max_steps=50,
                              return_scores=False):
    dae.eval()
    prev = previous_caption.long().contiguous()
    plen = prev_caplen.reshape(-1).long().contiguous()
    m = _FusedDcnet(dae, prev, plen, beam_size, max_steps)
    return _fused_beam([m], prev.shape[0], beam_size, dae.vocab_size, word_map, prev.device, max_steps,
                       return_scores=return_scores)


@torch.no_grad()
def beam_search_ensemble_batched(decoder, dae, image_features, previous_caption, prev_caplen, word_map, beam_size=3,
                                 max_steps=50, return_scores=False):
    """eval_full.py:88-218 for NI images at once: both models step on the same words, the epilogue averages
    their softmax probabilities."""
    decoder.eval()
    dae.eval()
    X = image_features.float().contiguous()
    prev = previous_caption.long().contiguous()
    plen = prev_caplen.reshape(-1).long().contiguous()
    e = _FusedEditNet(decoder, X, prev, plen, beam_size, max_steps)
    d = _FusedDcnet(dae, prev, plen, beam_size, max_steps)
    return _fused_beam([e, d], X.shape[0], beam_size, decoder.vocab_size, word_map, X.device, max_steps,
                       return_scores=return_scores)


# ------------------------------------------------------------------------------------------------
# The reference's calling convention: one image per call (editnet.py:601-613, dcnet.py:413-423,
# eval_full.py:96-109) -> (token list incl. <start>/<end>, score of the chosen hypothesis; NaN if the
# 50-step limit was hit).  NI = 1 case of the batched on-device search above.
# ------------------------------------------------------------------------------------------------
@torch.no_grad()
def _beam_search_editnet_persistent(decoder, image_features, previous_caption, prev_caplen, word_map, beam_size, max_steps=50):
    """ONE image, k <= 4: prologue + one persistent launch for the whole search (include/set_hip.h
    set_editnet_beam_persistent; the rows of the launch are the k hypotheses).  Returns None when the library answers
    SET_ERR_UNSUPPORTED (no token table yet, k > 4, dimensions outside the persistent launch): the caller takes the
    per-step search."""
    import ctypes as C
    from . import _lib
    from ._lib import check, ptr, stream_of
    k = int(beam_size)
    if k < 1 or k > 4 or image_features.shape[0] != 1 or getattr(decoder, "_adaptive", 0):
        return None
    decoder.eval()
    lib = _lib.load()
    dev = image_features.device
    X = image_features.float().expand(k, -1, -1).contiguous()
    prev = previous_caption.long().expand(k, -1).contiguous()
    plen = prev_caplen.reshape(-1).long().expand(k).contiguous()
    picks = max_steps + 1
    dims = decoder._dims(k, prev.shape[1], X.shape[1], picks)
    w = decoder._weights(dims)
    if not w.tok_table:
        return None
    ws = decoder._workspace(dims)
    # every output of the launch in ONE device buffer: [hist_word (picks, 4) i64 | best_word i64 | hist_parent (picks, 4) i32 |
    # result (4) i32 | best_score f32], read back with a single copy (the search's only host synchronisation)
    n_hw, n_hp = picks * 4 * 8, picks * 4 * 4
    buf = torch.empty(n_hw + 8 + n_hp + 16 + 8, dtype=torch.uint8, device=dev)
    o_bw, o_hp, o_res, o_bs = n_hw, n_hw + 8, n_hw + 8 + n_hp, n_hw + 8 + n_hp + 16
    base = buf.data_ptr()
    rc = lib.set_editnet_beam_persistent(C.byref(w), C.byref(dims), ptr(X), None, ptr(prev), ptr(plen), int(word_map['<start>']),
                                         int(word_map['<end>']), picks, base + o_hp, base, base + o_bs, base + o_bw, base + o_res,
                                         ptr(ws), ws.numel(), stream_of(dev))
    if rc == 2:                                                    # SET_ERR_UNSUPPORTED: no output was touched (set_hip.h: answered
        return None                                                # before the prologue except on a device too small for the grid)
    check(rc, "set_editnet_beam_persistent")
    host = buf.cpu().numpy()
    hw = host[:n_hw].view("int64").reshape(picks, 4)
    hp = host[o_hp:o_hp + n_hp].view("int32").reshape(picks, 4)
    best_t, best_parent, k_left, made = (int(v) for v in host[o_res:o_res + 16].view("int32"))
    best_word_h = int(host[o_bw:o_bw + 8].view("int64")[0])
    best_score_h = float(host[o_bs:o_bs + 4].view("float32")[0])
    if made < 0:
        raise _lib.SetError("set_editnet_beam_persistent: the persistent launch timed out (result poisoned)")

    def trace(t_last, slot):
        out = []
        for t in range(t_last, -1, -1):
            out.append(int(hw[t, slot]))
            slot = int(hp[t, slot])
        return out[::-1]

    start = int(word_map['<start>'])
    if k_left > 0:                                                 # ran into the step limit (editnet.py:702-704,711)
        return ([start] + trace(made - 1, 0))[:18], float("nan")
    return [start] + trace(best_t - 1, best_parent) + [best_word_h], best_score_h


def beam_search_editnet(decoder, image_features, previous_caption, prev_caplen, word_map, beam_size=3):
    """The reference's own calling convention, ONE image per call (editnet.py:601-613).  k <= 4 with the token table
    active: one persistent launch (csrc/decode_persistent_wide.hip, beam mode); otherwise the NI = 1 case of the batched
    search."""
    one = _beam_search_editnet_persistent(decoder, image_features, previous_caption, prev_caplen, word_map, beam_size)
    if one is not None:
        return one
    seqs, scores = beam_search_editnet_batched(decoder, image_features, previous_caption, prev_caplen, word_map,
                                               beam_size, return_scores=True)
    return seqs[0], scores[0]


def beam_search_dcnet(dae, previous_caption, prev_caplen, word_map, beam_size=3):
    seqs, scores = beam_search_dcnet_batched(dae, previous_caption, prev_caplen, word_map, beam_size, return_scores=True)
    return seqs[0], scores[0]


def beam_search_ensemble(decoder, dae, image_features, previous_caption, prev_caplen, word_map, beam_size=3):
    seqs, scores = beam_search_ensemble_batched(decoder, dae, image_features, previous_caption, prev_caplen, word_map,
                                                beam_size, return_scores=True)
    return seqs[0], scores[0]
