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
