"""Test comparator: a one-image beam search that drives the decoder through its SUB-MODULE attributes
(`decoder.embed`, `.caption_encoder`, `.attention_lstm`, `.caption_attention`, `.visual_attention`, `.select`,
`.copy_lstm`, `.fc`, `.init_hidden_state`; DCNet: `.language_lstm`), i.e. the way the reference's evaluate()
loops call them (SURVEY.md §8b "attributes called directly by callers").  Every module call lands in a HIP
kernel; the hypothesis bookkeeping is plain numpy on the host.  Not part of the product: the product's search
is the batched on-device one in show_edit_tell_amd/evaluate.py, which the tests compare against this and
against the reference's own results (tests/golden/beam_*.npz)."""
import numpy as np
import torch


class EditNetRows:
    """k identical hypothesis rows of one image; `advance` = one timestep through the module attributes"""
    fields = ("h1", "c1", "h2", "c2", "X", "mean", "fh", "H", "M", "mask")

    def __init__(self, dec, X1, prev1, plen1, k):
        self.dec = dec
        H, M, fh, mask = dec.caption_encoder(prev1, plen1)
        rep = lambda t: t.expand(k, *t.shape[1:]).contiguous()
        self.X, self.mean = rep(X1), rep(X1.mean(1))
        self.H, self.M, self.fh, self.mask = rep(H), rep(M), rep(fh), rep(mask)
        self.h1, self.c1 = dec.init_hidden_state(k)
        self.h2, self.c2 = dec.init_hidden_state(k)

    def advance(self, words):
        d = self.dec
        emb = d.embed(words.view(-1, 1)).squeeze(1)
        self.h1, self.c1 = d.attention_lstm(torch.cat([emb, self.fh, self.h2, self.mean], 1), (self.h1, self.c1))
        cap, alpha = d.caption_attention(self.H, self.h1, emb, self.mask)
        img = d.visual_attention(self.X, self.h1)
        self.h2, self.c2 = d.copy_lstm(torch.cat([self.h1, cap, img], 1), (self.h2, self.c2), d.select(self.M, alpha))
        return d.fc(self.h2)

    def keep(self, rows):
        for n in self.fields:
            setattr(self, n, getattr(self, n)[rows].contiguous())


class DcnetRows:
    fields = ("h1", "c1", "h2", "c2", "enc", "fh", "mask")

    def __init__(self, dae, prev1, plen1, k):
        self.dae = dae
        enc, fh, mask = dae.caption_encoder(prev1, plen1)
        rep = lambda t: t.expand(k, *t.shape[1:]).contiguous()
        self.enc, self.fh, self.mask = rep(enc), rep(fh), rep(mask)
        self.h1, self.c1 = dae.init_hidden_state(k)
        self.h2, self.c2 = dae.init_hidden_state(k)

    def advance(self, words):
        d = self.dae
        emb = d.embed(words.view(-1, 1)).squeeze(1)
        self.h1, self.c1 = d.attention_lstm(torch.cat([emb, self.fh, self.h2], 1), (self.h1, self.c1))
        self.h2, self.c2 = d.language_lstm(torch.cat([self.h1, d.caption_attention(self.enc, self.h1, self.mask)], 1),
                                           (self.h2, self.c2))
        return d.fc(self.h2)

    def keep(self, rows):
        for n in self.fields:
            setattr(self, n, getattr(self, n)[rows].contiguous())


def _log_probs(logit_list):
    if len(logit_list) == 1:
        return torch.log_softmax(logit_list[0], 1)
    return torch.log(sum(torch.softmax(l, 1) for l in logit_list) / len(logit_list))      # eval_full.py:151-153


@torch.no_grad()
def search(models, word_map, k, dev, max_steps=50):
    """-> (token list incl. <start>/<end>, score or NaN at the step limit)"""
    V, start, end = len(word_map), int(word_map["<start>"]), int(word_map["<end>"])
    words = torch.full((k,), start, dtype=torch.long, device=dev)
    hyps = [[start] for _ in range(k)]
    acc = np.zeros(k, np.float32)
    finished = []                                   # (score, tokens) in completion order
    for step in range(1, max_steps + 2):
        lp = _log_probs([m.advance(words) for m in models]).cpu().numpy()
        total = acc[:, None] + lp
        flat = total[0] if step == 1 else total.reshape(-1)
        pick = np.argsort(-flat, kind="stable")[:k]
        parents, nxt = pick // V, pick % V
        new_hyps = [hyps[p] + [int(w)] for p, w in zip(parents, nxt)]
        live = [i for i, w in enumerate(nxt) if w != end]
        finished += [(float(flat[pick[i]]), new_hyps[i]) for i in range(len(nxt)) if nxt[i] == end]
        k = len(live)
        if k == 0:
            best = max(range(len(finished)), key=lambda i: (finished[i][0], -i))     # first maximum
            return finished[best][1], finished[best][0]
        hyps = [new_hyps[i] for i in live]
        rows = torch.as_tensor(parents[live], device=dev)
        for m in models:
            m.keep(rows)
        acc = flat[pick[live]].astype(np.float32)
        words = torch.as_tensor(nxt[live], device=dev)
    return hyps[0][:18], float("nan")


def beam_editnet(dec, X1, prev1, plen1, wm, k=3):
    dec.eval()
    return search([EditNetRows(dec, X1, prev1, plen1, k)], wm, k, X1.device)


def beam_dcnet(dae, prev1, plen1, wm, k=3):
    dae.eval()
    return search([DcnetRows(dae, prev1, plen1, k)], wm, k, prev1.device)


def beam_ensemble(dec, dae, X1, prev1, plen1, wm, k=3):
    dec.eval()
    dae.eval()
    return search([EditNetRows(dec, X1, prev1, plen1, k), DcnetRows(dae, prev1, plen1, k)], wm, k, X1.device)
