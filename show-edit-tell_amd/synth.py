"""Deterministic, portable synthetic inputs and weights for the EditNet / DCNet decode path.

Everything here is a pure function of (seed, name, shape): a counter-based splitmix64 hash,
evaluated with numpy uint64 arithmetic, so the same tensors are produced in the authoring
container (where the golden fixtures are captured from the reference) and on the GPU box
(where the HIP path is checked against those fixtures).  torch's RNG is never used.

Shapes / token conventions follow the reference data format:
  * features  (B, 36, 2048) fp32, non-negative with ~30 % exact zeros (post-ReLU pooled
    bottom-up activations; reference `editnet.py:48-74`)
  * captions  (B, 20)  = <start>, words, <end>, <pad>*   (`preprocess_caps.py:116-117`)
  * prev caps (B, T)   = words, <pad>*                   (`preprocess_existing_caps.py:23`)
  * word map: <pad>=0, words 1..N, <unk>=N+1, <start>=N+2, <end>=N+3 (`preprocess_caps.py:87-91`)
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        x += np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _stream(seed: int, name: str, n: int) -> np.ndarray:
    """n 64-bit hashes for (seed, name)."""
    key = np.uint64((int(seed) * 0x100000001B3 + zlib.crc32(name.encode())) & 0xFFFFFFFFFFFFFFFF)
    base = _splitmix64(np.array([key], dtype=np.uint64))[0]
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) + base
    return _splitmix64(ctr)


def unit(seed: int, name: str, shape) -> np.ndarray:
    """U[0,1) fp32 with 24 random mantissa bits (exactly representable)."""
    n = int(np.prod(shape)) if len(shape) else 1
    bits = _stream(seed, name, n) >> np.uint64(40)
    return (bits.astype(np.float32) * np.float32(1.0 / 16777216.0)).reshape(shape)


def uniform(seed: int, name: str, shape, lo: float, hi: float) -> np.ndarray:
    u = unit(seed, name, shape)
    return (np.float32(lo) + u * np.float32(hi - lo)).astype(np.float32)


def integers(seed: int, name: str, shape, lo: int, hi: int) -> np.ndarray:
    """int64 in [lo, hi)."""
    n = int(np.prod(shape)) if len(shape) else 1
    r = _stream(seed, name, n) >> np.uint64(11)
    return (lo + (r % np.uint64(hi - lo)).astype(np.int64)).reshape(shape)


# ----------------------------------------------------------------------------------------
# vocabulary
# ----------------------------------------------------------------------------------------
def word_map(vocab_size: int) -> "OrderedDict[str, int]":
    """Synthetic word map with the reference's special-token layout (`preprocess_caps.py:87-91`)."""
    n_words = vocab_size - 4
    wm = OrderedDict()
    wm["<pad>"] = 0
    for i in range(1, n_words + 1):
        wm["w%d" % i] = i
    wm["<unk>"] = n_words + 1
    wm["<start>"] = n_words + 2
    wm["<end>"] = n_words + 3
    assert len(wm) == vocab_size
    return wm


# ----------------------------------------------------------------------------------------
# inputs
# ----------------------------------------------------------------------------------------
def features(seed: int, B: int, R: int = 36, F: int = 2048) -> np.ndarray:
    """(B,R,F) non-negative, ~30 % exact zeros, values in [0,2)."""
    u = unit(seed, "features", (B, R, F))
    keep = unit(seed, "features.keep", (B, R, F)) >= np.float32(0.3)
    return (u * np.float32(2.0) * keep).astype(np.float32)


def adaptive_features(seed: int, B: int, Rmax: int = 100, F: int = 2048, lo: int = 10):
    """Zero-padded (B,Rmax,F) with n_b valid regions, plus image_mean over valid rows
    (`adaptive_features/editnet_adaptive.py:58-80`)."""
    x = features(seed, B, Rmax, F)
    n = integers(seed, "features.nvalid", (B,), lo, Rmax + 1)
    valid = np.arange(Rmax)[None, :] < n[:, None]
    x = x * valid[:, :, None].astype(np.float32)
    mean = (x.sum(1) / n[:, None].astype(np.float32)).astype(np.float32)
    return x, mean, n


def prev_captions(seed: int, B: int, T: int, vocab_size: int, min_len: int = 5):
    """(B,T) int64 words in 1..V-4, zero padded; lengths (B,1) in [min_len,T]."""
    lens = integers(seed, "prev.len", (B,), min_len, T + 1)
    toks = integers(seed, "prev.tok", (B, T), 1, vocab_size - 3)
    toks = toks * (np.arange(T)[None, :] < lens[:, None])
    return toks.astype(np.int64), lens.reshape(B, 1).astype(np.int64)


def captions(seed: int, B: int, vocab_size: int, L: int = 20, min_len: int = 20):
    """(B,L) = <start> w.. <end> <pad>..; caplens (B,1) counts <start>..<end> inclusive."""
    wm_start, wm_end = vocab_size - 2, vocab_size - 1
    lens = integers(seed, "cap.len", (B,), min_len, L + 1)
    toks = integers(seed, "cap.tok", (B, L), 1, vocab_size - 3)
    pos = np.arange(L)[None, :]
    toks = np.where(pos == 0, wm_start, toks)
    toks = np.where(pos == lens[:, None] - 1, wm_end, toks)
    toks = np.where(pos >= lens[:, None], 0, toks)
    return toks.astype(np.int64), lens.reshape(B, 1).astype(np.int64)


# ----------------------------------------------------------------------------------------
# weights (state_dict layouts enumerated in SURVEY.md §8b)
# ----------------------------------------------------------------------------------------
def _lin(sd, seed, name, out_f, in_f, scale=1.0, bias=True):
    k = scale / np.sqrt(in_f)
    sd[name + ".weight"] = uniform(seed, name + ".weight", (out_f, in_f), -k, k)
    if bias:
        sd[name + ".bias"] = uniform(seed, name + ".bias", (out_f,), -k, k)


def editnet_state(seed: int, vocab_size: int, decoder_dim: int = 1024, attention_dim: int = 512,
                  image_features_dim: int = 2048, emb_scale: float = 1.0, fc_scale: float = 1.0,
                  gain: float = 1.0) -> "OrderedDict[str, np.ndarray]":
    """EditNet `DecoderC.state_dict()` (reference `editnet.py:449-471`), fp32 numpy.

    emb_dim == caption_features_dim == decoder_dim (forced by the reference's cat shapes).
    `gain` widens the recurrent/attention weights so greedy outputs are non-degenerate."""
    D, A, F, V = decoder_dim, attention_dim, image_features_dim, vocab_size
    sd = OrderedDict()
    sd["embed.embedding.weight"] = uniform(seed, "embed.embedding.weight", (V, D), -emb_scale, emb_scale)
    _lin(sd, seed, "caption_encoder.lstm_encoder_cell.x2h", 4 * D, D, gain)
    _lin(sd, seed, "caption_encoder.lstm_encoder_cell.h2h", 4 * D, D, gain)
    _lin(sd, seed, "caption_encoder.affine_hn", D, D, gain)
    _lin(sd, seed, "caption_attention.cap_features_att", A, D, gain)
    _lin(sd, seed, "caption_attention.cap_decoder_att", A, D, gain)
    _lin(sd, seed, "caption_attention.cap_full_att", 1, A, gain * 4)
    _lin(sd, seed, "caption_attention.context_gate", D, 3 * D, gain)
    _lin(sd, seed, "caption_attention.sc_affine", D, D, gain)
    _lin(sd, seed, "caption_attention.tc_affine", D, 2 * D, gain)
    _lin(sd, seed, "visual_attention.att_embed.0", D, F, gain)
    _lin(sd, seed, "visual_attention.features_att", A, D, gain)
    _lin(sd, seed, "visual_attention.decoder_att", A, D, gain)
    _lin(sd, seed, "visual_attention.full_att", 1, A, gain * 4)
    k = gain / np.sqrt(D)
    sd["attention_lstm.weight_ih"] = uniform(seed, "attention_lstm.weight_ih", (4 * D, 3 * D + F), -k, k)
    sd["attention_lstm.weight_hh"] = uniform(seed, "attention_lstm.weight_hh", (4 * D, D), -k, k)
    sd["attention_lstm.bias_ih"] = uniform(seed, "attention_lstm.bias_ih", (4 * D,), -k, k)
    sd["attention_lstm.bias_hh"] = uniform(seed, "attention_lstm.bias_hh", (4 * D,), -k, k)
    for n, (o, i) in (("x2h", (4 * D, 2 * D + F)), ("h2h", (4 * D, D)),
                      ("gate_cnew", (D, D)), ("gate_cmem", (D, D))):
        sd["copy_lstm.%s.weight" % n] = uniform(seed, "copy_lstm.%s.weight" % n, (o, i), -k, k)
        sd["copy_lstm.%s.bias" % n] = uniform(seed, "copy_lstm.%s.bias" % n, (o,), -k, k)
    _lin(sd, seed, "fc", V, D, fc_scale)
    return sd


def dcnet_state(seed: int, vocab_size: int, decoder_dim: int = 1024, attention_dim: int = 512,
                caption_features_dim: int = 512, emb_dim: int = 1024, emb_scale: float = 1.0,
                fc_scale: float = 1.0, gain: float = 1.0) -> "OrderedDict[str, np.ndarray]":
    """DCNet `DAE.state_dict()` (reference `dcnet.py:273-295`), fp32 numpy."""
    D, A, C, E, V = decoder_dim, attention_dim, caption_features_dim, emb_dim, vocab_size
    sd = OrderedDict()
    k = gain / np.sqrt(D)
    sd["attention_lstm.weight_ih"] = uniform(seed, "attention_lstm.weight_ih", (4 * D, 3 * E), -k, k)
    sd["attention_lstm.weight_hh"] = uniform(seed, "attention_lstm.weight_hh", (4 * D, D), -k, k)
    sd["attention_lstm.bias_ih"] = uniform(seed, "attention_lstm.bias_ih", (4 * D,), -k, k)
    sd["attention_lstm.bias_hh"] = uniform(seed, "attention_lstm.bias_hh", (4 * D,), -k, k)
    sd["language_lstm.weight_ih"] = uniform(seed, "language_lstm.weight_ih", (4 * D, 2 * E), -k, k)
    sd["language_lstm.weight_hh"] = uniform(seed, "language_lstm.weight_hh", (4 * D, D), -k, k)
    sd["language_lstm.bias_ih"] = uniform(seed, "language_lstm.bias_ih", (4 * D,), -k, k)
    sd["language_lstm.bias_hh"] = uniform(seed, "language_lstm.bias_hh", (4 * D,), -k, k)
    sd["embed.embedding.weight"] = uniform(seed, "embed.embedding.weight", (V, E), -emb_scale, emb_scale)
    kc = gain / np.sqrt(C)
    for sfx in ("", "_reverse"):
        sd["caption_encoder.lstm_encoder.weight_ih_l0" + sfx] = uniform(
            seed, "enc.weight_ih_l0" + sfx, (4 * C, E), -kc, kc)
        sd["caption_encoder.lstm_encoder.weight_hh_l0" + sfx] = uniform(
            seed, "enc.weight_hh_l0" + sfx, (4 * C, C), -kc, kc)
        sd["caption_encoder.lstm_encoder.bias_ih_l0" + sfx] = uniform(
            seed, "enc.bias_ih_l0" + sfx, (4 * C,), -kc, kc)
        sd["caption_encoder.lstm_encoder.bias_hh_l0" + sfx] = uniform(
            seed, "enc.bias_hh_l0" + sfx, (4 * C,), -kc, kc)
    _lin(sd, seed, "caption_encoder.concat", 2 * C, 2 * C, gain)
    _lin(sd, seed, "caption_attention.cap_features_att", A, 2 * C, gain)
    _lin(sd, seed, "caption_attention.cap_decoder_att", A, D, gain)
    _lin(sd, seed, "caption_attention.cap_full_att", 1, A, gain * 4)
    _lin(sd, seed, "fc", V, D, fc_scale)
    return sd
