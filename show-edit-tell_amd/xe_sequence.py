"""The teacher-forced EditNet decode loop (reference `editnet.py:479-548`) as ONE autograd node.

`autograd_ops.py` wraps every module call of a timestep in its own autograd node; PyTorch then stitches the timesteps
together with `cat`s of the LSTM inputs, tensor-sized `add`s wherever a state feeds several consumers, a `sum` per bias
and timestep, its own dropout kernels ... (23 % of the training step's GPU time and ~1800 launches per step at B = 128).
Here the whole loop is one `torch.autograd.Function`:

* forward: the same HIP operators (`set_*_train_f32`), but every saved activation is written straight into a
  per-sequence LOG of shape (T, B, .) — the concatenated LSTM input rows `[emb | final_hidden | h2 | mean]` and
  `[h1 | attend_cap | attend_img]` are packed in place by `set_pack_f32`, the three `nn.Dropout` sites use the library's
  Philox kernels (`set_dropout_f32`);
* backward: back-propagation through time on raw buffers.  Every "+=" of the chain rule is the accumulate flag of the
  fp32-MFMA GEMM (`set_gemm_group_f32`) or of the attention / select backward kernels (`set_*_bwd_acc_f32`): the running
  dh1 / dh2 and the gradients of the loop-invariant operands (H, Mem, cap_features_att(H), final_hidden,
  relu(att_embed(X))) are summed where they are produced; no tensor is added to another by a separate kernel;
* parameter gradients: ONE contraction per parameter over all T x B log rows (`autograd_ops._wgrad` / `_bgrad`, which also
  feed the data-parallel step's bucketed all-reduce through `deferred_param_grads(on_ready=...)`).

Values and gradients equal the per-operator path's (same kernels, same contraction order inside a GEMM; the sums over
timesteps are taken in reverse time order instead of autograd's engine order) — `tests/test_hip_train.py` checks both
against the reference's own autograd gradients.  Modes (SeqConfig): teacher-forced, teacher-forced with scheduled sampling,
adaptive features (region masks + `decoder_last_hidden`), free-running sampled rollout (SCST).
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib
from . import autograd_ops as A
from . import rng
from ._lib import EditNetWeights, check, ptr, stream_of

PARAM_NAMES = (
    "E", "al_wih", "al_whh", "al_bih", "al_bhh",
    "ca_dec_w", "ca_dec_b", "ca_full_w", "ca_full_b", "ca_gate_w", "ca_gate_b", "ca_sc_w", "ca_sc_b", "ca_tc_w", "ca_tc_b",
    "va_fa_w", "va_fa_b", "va_dec_w", "va_dec_b", "va_full_w", "va_full_b",
    "cl_x2h_w", "cl_x2h_b", "cl_h2h_w", "cl_h2h_b", "cl_cnew_w", "cl_cnew_b", "cl_cmem_w", "cl_cmem_b",
    "fc_w", "fc_b")


def decoder_params(dec):
    """the decoder's parameters in PARAM_NAMES order"""
    ca, va, cl, al = dec.caption_attention, dec.visual_attention, dec.copy_lstm, dec.attention_lstm
    return (dec.embed.embedding.weight, al.weight_ih, al.weight_hh, al.bias_ih, al.bias_hh,
            ca.cap_decoder_att.weight, ca.cap_decoder_att.bias, ca.cap_full_att.weight, ca.cap_full_att.bias,
            ca.context_gate.weight, ca.context_gate.bias, ca.sc_affine.weight, ca.sc_affine.bias,
            ca.tc_affine.weight, ca.tc_affine.bias,
            va.features_att.weight, va.features_att.bias, va.decoder_att.weight, va.decoder_att.bias,
            va.full_att.weight, va.full_att.bias,
            cl.x2h.weight, cl.x2h.bias, cl.h2h.weight, cl.h2h.bias, cl.gate_cnew.weight, cl.gate_cnew.bias,
            cl.gate_cmem.weight, cl.gate_cmem.bias, dec.fc.weight, dec.fc.bias)


class SeqConfig:
    """non-tensor arguments of one sequence"""

    def __init__(self, decode_lengths, train, p_embed, p_region, p_out, seed, rollout=None):
        self.decode_lengths = [int(x) for x in decode_lengths]
        self.train = bool(train)
        self.p_embed, self.p_region, self.p_out = float(p_embed), float(p_region), float(p_out)
        self.seed = int(seed)
        # free-running sampled rollout (editnet_rl.py:485-549 with sample_rl): dict(max_len, start_idx, end_idx) — the token
        # of step t is what the device sampling epilogue drew at step t-1, the node returns (seq, seq_logp)
        self.rollout = rollout
        # adaptive features (adaptive_features/editnet_adaptive.py:438-457, 560-562): Yin arrives with the padded regions
        # zeroed; train mode re-derives the region mask from every step's dropped-out embedding, eval mode uses `rmask`
        # (B, R); the node also returns every row's last h2 (`decoder_last_hidden`)
        self.adaptive = False
        self.rmask = None
        # scheduled sampling (editnet.py:508-520): from step 1 on every row's input word is, with probability ss_prob, a
        # draw from softmax(previous step's scores) instead of the ground-truth word (device Philox draw, no host sync)
        self.ss_prob = 0.0


def _z(*shape, dev):
    return torch.zeros(*shape, dtype=torch.float32, device=dev)


def _e(*shape, dev):
    return torch.empty(*shape, dtype=torch.float32, device=dev)


class _Ops:
    """thin, allocation-free callers of the C ABI for one (device, stream)"""

    def __init__(self, dev):
        self.lib = _lib.load()
        self.dev = dev
        self.st = stream_of(dev)
        self._ws = {}

    def ws(self, key, nbytes):
        w = self._ws.get(key)
        if w is None or w.numel() < nbytes:
            w = self._ws[key] = torch.empty(max(16, nbytes), dtype=torch.uint8, device=self.dev)
        return w

    def pack(self, dst, rows, srcs, accumulate=False):
        """dst[:rows, :sum(cols)] (+)= [src0 | src1 | ...]; srcs: 2-D views with unit inner stride"""
        n = len(srcs)
        ps = (C.c_void_p * n)(*[s.data_ptr() for s in srcs])
        ls = (C.c_int64 * n)(*[s.stride(0) for s in srcs])
        cs = (C.c_int * n)(*[s.shape[1] for s in srcs])
        check(self.lib.set_pack_f32(dst.data_ptr(), dst.stride(0), rows, n, ps, ls, cs, int(accumulate), self.st), "set_pack_f32")

    def dropout(self, x, y, rows, cols, p, seed, offset):
        check(self.lib.set_dropout_f32(x.data_ptr(), x.stride(-2), y.data_ptr(), y.stride(-2), rows, cols, p, seed, offset,
                                       self.st), "set_dropout_f32")

    def dropout_bwd(self, dy, y, dx, rows, cols, scale, accumulate):
        check(self.lib.set_dropout_bwd_f32(dy.data_ptr(), dy.stride(-2), y.data_ptr(), y.stride(-2), dx.data_ptr(),
                                           dx.stride(-2), rows, cols, scale, int(accumulate), self.st), "set_dropout_bwd_f32")

    def linear(self, x, w, b, y, M):
        """y[:M] = x[:M] w^T + b   (x, y: 2-D views, unit inner stride)"""
        K, N = w.shape[1], w.shape[0]
        ws = self.ws("lin", self.lib.set_linear_workspace_bytes(M, N, K))
        check(self.lib.set_linear_f32(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), ptr(b), y.data_ptr(), y.stride(0),
                                      M, N, K, _lib.ACT_NONE, ws.data_ptr(), ws.numel(), self.st), "set_linear_f32")


_DEMB_HOIST = os.environ.get("SET_DEMB_HOIST", "1") != "0"     # time-batched d emb products of the backward (round 4)
# round 5: the per-timestep dX products of the backward hand their split-K partials to the kernels that consume them
# (include/set_hip.h SetSlabSrc / *_src entry points) instead of reducing them with a launch each: 6 launches less per
# timestep (5 reductions + the output dropout's backward).  SET_SLAB_DIRECT=0: the round-4 chain.
_SLAB_DIRECT = os.environ.get("SET_SLAB_DIRECT", "1") != "0"
# round 5: the two timestep loops as ONE C call each (csrc/train_loop.hip: the same entry points in the same order) — issued
# from Python their ~45 launches per timestep cost the host more than the kernels take.  SET_XE_C_LOOPS=0: the Python loops.
_C_LOOPS = os.environ.get("SET_XE_C_LOOPS", "1") != "0"
# per-timestep output logs instead of a packing launch inside the loop (SetXELoopArgs.step_logs): measured 15.51 vs 15.52 ms at
# B = 128 — the packing launch it removes from the dependent chain comes back as a third problem of the copy cell's grouped
# product and a (T B)-row pack after the loop.  Correct (tests/test_hip_sequence.py) but not a win: opt-in.
_STEP_LOGS = os.environ.get("SET_XE_STEP_LOGS", "0") == "1"
_slab_ws = {}


def _slab_scratch(dev, slot, nbytes=32 << 20):
    """partials of one product position of the backward chain: its own region, alive until the consumers have run"""
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream, slot)
    ws = _slab_ws.get(key)
    if ws is None:
        if len(_slab_ws) >= 40:
            _slab_ws.clear()
        ws = _slab_ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    return ws
_ATT1_HOIST_LIVE = 0.75     # all-timestep region projection when at least this fraction of the (t, b) rows is live
# Overlapped weight gradients (round 3): the time-batched dW contractions (0.4 TFLOP, the chip's full width) only become
# possible when the back-propagation through time has produced the gradient logs — but the BPTT loop itself is a chain of
# M = 128 launches and small kernels that leaves most of the matrix pipe idle.  Half way through the loop the rows of the
# later timesteps are final: their share of every dW is contracted on a side stream underneath the rest of the loop, the
# earlier timesteps' share is accumulated on top after it.  Measured (B = 128, same box, ABAB): 18.10 / 17.81 ms without,
# 17.84 / 18.82 ms with — as with every other second-queue experiment on this path the loop's kernels slow down by what the
# side stream gains.  Correct (the gradient-parity tests pass with it on) but not a win: opt-in, SET_WGRAD_OVERLAP=1.
_WGRAD_OVERLAP = __import__("os").environ.get("SET_WGRAD_OVERLAP", "0") == "1"
_side = {}


def _side_stream(dev):
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    st = _side.get(key)
    if st is None:
        if len(_side) > 16:
            _side.clear()
        st = _side[key] = torch.cuda.Stream(dev)
    return st


def _dvalues(alpha, dctx, ops):
    """dH (B, Tc, D) = sum_t alpha[t, b, :] (outer) dctx[t, b, :] over the per-sequence logs (rows of finished sequences are
    zero in both), one launch"""
    T, B, Tc = alpha.shape
    D = dctx.shape[2]
    dH = torch.empty(B, Tc, D, dtype=torch.float32, device=alpha.device)
    check(ops.lib.set_attention_dvalues_f32(alpha.data_ptr(), dctx.data_ptr(), dH.data_ptr(), T, B, Tc, D, 0, ops.st),
          "set_attention_dvalues_f32")
    return dH


def _rows(t2d, n):
    return t2d if t2d.shape[0] == n else t2d[:n]


class _XESequence(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg, X, mean, H, Mem, final_hidden, mask, att1_c, Yin, caps, *params):
        P = dict(zip(PARAM_NAMES, params))
        dev = X.device
        ops = _Ops(dev)
        lib, st = ops.lib, ops.st
        ro = cfg.rollout
        lens = cfg.decode_lengths if ro is None else [int(ro["max_len"])] * X.shape[0]
        T, B = max(lens), X.shape[0]
        bts = [sum(1 for l in lens if l > t) for t in range(T)]
        R, F = X.shape[1], X.shape[2]
        Tc, D = H.shape[1], H.shape[2]
        Adim, V = P["ca_dec_w"].shape[0], P["fc_w"].shape[0]
        K1, K2 = 3 * D + F, 2 * D + F
        train = cfg.train
        X, mean, H, Mem, final_hidden, mask, att1_c, Yin = (t.contiguous() for t in (X, mean, H, Mem, final_hidden, mask, att1_c, Yin))
        caps = caps.contiguous()

        # the per-sequence logs.  Rows of sequences that have left the batch are never written: with ragged lengths the
        # logs start as zeros (such rows meet zero gradient rows in the time-batched contractions and must be finite);
        # with uniform lengths every row is written and only the initial states need zeros
        uniform = min(lens) == T
        _zl = _e if uniform else _z
        L = {}
        # editnet.py:523 feeds [emb | final_hidden | h2 | image_mean] to the attention LSTM: the final_hidden and image_mean
        # columns do not change over the sequence, so their products (+ both biases) are contracted once (`pre1`) and
        # every step contracts only the emb / h2 column blocks of weight_ih (half the contraction length)
        wih = P["al_wih"]
        pre1 = _e(B, 4 * D, dev=dev)
        ops.linear(final_hidden, wih[:, D:2 * D], P["al_bih"] + P["al_bhh"], pre1, B)
        A.gemm(mean, False, wih[:, 3 * D:], False, B, 4 * D, F, out=pre1, accumulate=True)
        L["FH"], L["MEAN"] = final_hidden, mean
        L["EMB"] = _zl(T, B, D, dev=dev)
        for k in ("H1", "C1", "H2", "C2"):                    # slot t = state BEFORE step t; slot 0 = the zero initial state
            L[k] = _zl(T + 1, B, D, dev=dev)
            if uniform:
                L[k][0].zero_()
        L["G1"], L["G2"] = _zl(T, B, 4 * D, dev=dev), _zl(T, B, 4 * D, dev=dev)
        L["WHC"] = _zl(T, B, 3 * D, dev=dev)                  # [word | h1 | caption context]
        L["ZT"], L["S"], L["TT"] = (_zl(T, B, D, dev=dev) for _ in range(3))
        L["ALPHAC"], L["ALPHAV"] = _zl(T, B, Tc, dev=dev), _zl(T, B, R, dev=dev)
        L["ATT2C"], L["ATT2V"] = _e(T, B, Adim, dev=dev), _e(T, B, Adim, dev=dev)     # decoder-side projections as scored
        L["SEL"], L["CNEW"], L["CG"] = (_zl(T, B, D, dev=dev) for _ in range(3))
        L["X2"] = _zl(T, B, K2, dev=dev)
        if train:
            L["FE"] = _zl(T, B, R, D, dev=dev)
            L["ATT1"] = _zl(T, B, R, Adim, dev=dev)
            L["H2D"] = _zl(T, B, D, dev=dev)
        gated, cx, aimg = _e(B, D, dev=dev), _e(B, D, dev=dev), _e(B, F, dev=dev)
        adaptive = cfg.adaptive
        if adaptive:
            L["RMASK"] = _zl(T, B, R, dev=dev) if train else None
            rmask_eval = None if train else cfg.rmask.contiguous()

        w = EditNetWeights()
        w.ca_dec_w, w.ca_dec_b = P["ca_dec_w"].data_ptr(), P["ca_dec_b"].data_ptr()
        w.ca_full_w, w.ca_full_b = P["ca_full_w"].data_ptr(), P["ca_full_b"].data_ptr()
        w.ca_gate_w, w.ca_gate_b = P["ca_gate_w"].data_ptr(), P["ca_gate_b"].data_ptr()
        w.ca_sc_w, w.ca_sc_b, w.ca_tc_w, w.ca_tc_b = (P[k].data_ptr() for k in ("ca_sc_w", "ca_sc_b", "ca_tc_w", "ca_tc_b"))
        w.va_dec_w, w.va_dec_b, w.va_full_w, w.va_full_b = (P[k].data_ptr() for k in ("va_dec_w", "va_dec_b", "va_full_w", "va_full_b"))
        w.cl_x2h_w, w.cl_x2h_b, w.cl_h2h_w, w.cl_h2h_b = (P[k].data_ptr() for k in ("cl_x2h_w", "cl_x2h_b", "cl_h2h_w", "cl_h2h_b"))
        w.cl_cnew_w, w.cl_cnew_b, w.cl_cmem_w, w.cl_cmem_b = (P[k].data_ptr() for k in ("cl_cnew_w", "cl_cnew_b", "cl_cmem_w", "cl_cmem_b"))
        wref = C.byref(w)
        ws_l = ops.ws("lstm", lib.set_lstm_cell_workspace_bytes(B, D, K1))
        ws_c = ops.ws("att", lib.set_editnet_attentions_workspace_bytes(B, D, Adim))
        ws_k = ops.ws("copy", lib.set_copy_lstm_workspace_bytes(B, D, K2))
        E = P["E"]
        ss = ro is None and cfg.ss_prob > 0.0
        cap_stride = caps.stride(0) if (ro is None and not ss) else 1
        if ss:
            L["TOK"] = caps[:, :T].t().contiguous()            # (T, B) words actually fed; rows of step t >= 1 may be replaced
            ss_u = (rng.uniforms(T * B, cfg.seed, rng.offset(rng.SITE_SS_COIN), dev) < cfg.ss_prob).view(T, B)
            ss_raw = torch.empty(B, dtype=torch.long, device=dev)
            ss_i64 = torch.zeros(2, B, dtype=torch.long, device=dev)
            ss_i32 = torch.zeros(B + T + 4, dtype=torch.int32, device=dev)
            ss_f32 = _e(2, B, dev=dev)
            pred_tb = _z(T, B, V, dev=dev)
        state = None
        if ro is not None:
            state = A.SampleState(B, T, ro["start_idx"], ro["end_idx"], dev, seed=ro.get("seed"), offset=ro.get("offset", 0))
            L["LOGITS"] = _e(T, B, V, dev=dev)
            L["RAW"] = torch.empty(T, B, dtype=torch.long, device=dev)
            L["LSE"], L["LOGP"] = _e(T, B, dev=dev), _e(T, B, dev=dev)
        scale_off = rng.offset              # (site, t) -> Philox offset; sites 1 = embedding, 2 = regions, 3 = h2 before fc, 4 = SS draw

        # editnet.py:441-443 in train mode: att1(t) = features_att(dropout_t(relu(att_embed(X)))) does not depend on the
        # recurrent state, so all T region projections run as ONE (T*B*R, D) x (D, A) product before the loop (19 products of
        # B*R = 4608 rows each leave the chip 1.1 rounds of tiles; 87552 rows fill it).  Ragged batches with many finished
        # rows keep the per-step products over the live rows only.
        att1_hoisted = train and sum(bts) >= _ATT1_HOIST_LIVE * T * B
        if train:
            if att1_hoisted:               # every row of every timestep: ONE launch (set_dropout_steps_f32 == T launches, bit for bit)
                check(lib.set_dropout_steps_f32(Yin.data_ptr(), D, L["FE"].data_ptr(), D, B * R * D, B * R, D, T, cfg.p_region,
                                                cfg.seed, scale_off(2, 0), st), "set_dropout_steps_f32")
            else:
                for t in range(T):
                    ops.dropout(Yin.view(B * R, D), L["FE"][t].view(B * R, D), bts[t] * R, D, cfg.p_region, cfg.seed, scale_off(2, t))
            if att1_hoisted:
                ops.linear(L["FE"].view(T * B * R, D), P["va_fa_w"], P["va_fa_b"], L["ATT1"].view(T * B * R, Adim), T * B * R)
                if adaptive:               # the data-derived region mask of every timestep (editnet_adaptive.py:449-453): one launch
                    check(lib.set_rowsum_mask_f32(L["FE"].data_ptr(), D, T * B * R, D, L["RMASK"].data_ptr(), st), "set_rowsum_mask_f32")

        c_loop = (_C_LOOPS and ro is None and not ss and (att1_hoisted or not train) and dev.type == "cuda")
        if c_loop:
            a = _lib.XELoopArgs()
            a.T, a.B, a.R, a.F, a.Tc, a.D, a.A, a.V, a.train = T, B, R, F, Tc, D, Adim, E.shape[0], int(bool(train))
            a.p_embed, a.p_out = float(cfg.p_embed if train else 0.0), float(cfg.p_out if train else 0.0)
            a.seed, a.off_embed, a.off_out = int(cfg.seed), scale_off(1, 0), scale_off(3, 0)
            bts_c = (C.c_int * T)(*bts)
            a.bts = C.cast(bts_c, C.c_void_p)
            a.w = C.addressof(w)
            a.E, a.al_wih, a.al_whh = E.data_ptr(), wih.data_ptr(), P["al_whh"].data_ptr()
            a.tok, a.tok_step, a.tok_stride = caps.data_ptr(), 1, cap_stride
            a.X, a.H, a.Mem, a.mask, a.att1_c, a.pre1 = (x.data_ptr() for x in (X, H, Mem, mask, att1_c, pre1))
            a.att1, a.att1_step = (L["ATT1"].data_ptr(), B * R * Adim) if train else (Yin.data_ptr(), 0)
            for k in ("EMB", "H1", "C1", "H2", "C2", "G1", "G2", "WHC", "ZT", "S", "TT", "ALPHAC", "ALPHAV", "ATT2C", "ATT2V", "SEL",
                      "CNEW", "CG", "X2"):
                setattr(a, k, L[k].data_ptr())
            a.H2D = L["H2D"].data_ptr() if (train and cfg.p_out > 0) else None
            if _STEP_LOGS:                 # per-timestep outputs kept: no packing launch inside the loop (see SetXELoopArgs.step_logs)
                gl, cl, al = _zl(T, B, D, dev=dev), _zl(T, B, D, dev=dev), _zl(T, B, F, dev=dev)
                a.gated, a.cx, a.aimg, a.step_logs = gl.data_ptr(), cl.data_ptr(), al.data_ptr(), 1
            else:
                a.gated, a.cx, a.aimg, a.step_logs = gated.data_ptr(), cx.data_ptr(), aimg.data_ptr(), 0
            a.ws_l, a.ws_l_bytes, a.ws_c, a.ws_c_bytes = ws_l.data_ptr(), ws_l.numel(), ws_c.data_ptr(), ws_c.numel()
            a.ws_k, a.ws_k_bytes = ws_k.data_ptr(), ws_k.numel()
            if adaptive:
                a.rmask, a.rmask_step = (L["RMASK"].data_ptr(), B * R) if train else (rmask_eval.data_ptr(), 0)
            check(lib.set_editnet_xe_train_loop_f32(C.byref(a), st), "set_editnet_xe_train_loop_f32")
        for t in (range(T) if not c_loop else ()):
            bt = bts[t]
            emb = L["EMB"][t]
            if ss and t >= 1:              # editnet.py:508-520 on the device: draw from softmax(scores of step t-1)
                check(lib.set_sample_pick_f32(pred_tb[t - 1].data_ptr(), V, bt, V, 0, 1, -1, cfg.seed, scale_off(4, t),
                                              ss_i64[0].data_ptr(), ss_i64[1].data_ptr(), ss_i32.data_ptr(),
                                              ss_i32[B:].data_ptr(), ss_raw.data_ptr(), ss_f32[0].data_ptr(),
                                              ss_f32[1].data_ptr(), st), "set_sample_pick_f32")
                L["TOK"][t, :bt] = torch.where(ss_u[t, :bt], ss_raw[:bt], L["TOK"][t, :bt])
            tok = (L["TOK"][t] if ss else caps[:, t]) if ro is None else state.tokens[t]
            if train and cfg.p_embed > 0:
                check(lib.set_embed_relu_dropout_f32(E.data_ptr(), tok.data_ptr(), cap_stride, emb.data_ptr(), D, bt, D, E.shape[0],
                                                     cfg.p_embed, cfg.seed, scale_off(1, t), st), "set_embed_relu_dropout_f32")
            else:
                check(lib.set_embed_relu_f32(E.data_ptr(), tok.data_ptr(), cap_stride, emb.data_ptr(), D, bt, D, E.shape[0], st),
                      "set_embed_relu_f32")
            h1 = L["H1"][t + 1]
            check(lib.set_lstm_cell_pre_train_f32(emb.data_ptr(), D, wih.data_ptr(), K1, D, L["H2"][t].data_ptr(), D,
                                                  wih[:, 2 * D:].data_ptr(), K1, D, L["H1"][t].data_ptr(),
                                                  P["al_whh"].data_ptr(), pre1.data_ptr(), 4 * D, L["C1"][t].data_ptr(),
                                                  h1.data_ptr(), L["C1"][t + 1].data_ptr(), L["G1"][t].data_ptr(), bt, D,
                                                  ws_l.data_ptr(), ws_l.numel(), st), "set_lstm_cell_pre_train_f32")
            if train:
                fe = L["FE"][t]
                att1 = L["ATT1"][t]
                if not att1_hoisted:
                    ops.linear(fe.view(B * R, D), P["va_fa_w"], P["va_fa_b"], att1.view(B * R, Adim), bt * R)
                if adaptive and not att1_hoisted:
                    check(lib.set_rowsum_mask_f32(fe.data_ptr(), D, bt * R, D, L["RMASK"][t].data_ptr(), st), "set_rowsum_mask_f32")
            else:
                att1 = Yin
            rm = None if not adaptive else (L["RMASK"][t].data_ptr() if train else rmask_eval.data_ptr())
            sel = L["SEL"][t]
            # both attentions + SelectC of this step: 4 launches (editnet.py:534-540)
            check(lib.set_editnet_attentions_train_f32(wref, H.data_ptr(), att1_c.data_ptr(), mask.data_ptr(), Mem.data_ptr(),
                                                       X.data_ptr(), att1.data_ptr(), rm, h1.data_ptr(), emb.data_ptr(),
                                                       gated.data_ptr(), L["ALPHAC"][t].data_ptr(), cx.data_ptr(),
                                                       L["ZT"][t].data_ptr(), L["S"][t].data_ptr(), L["TT"][t].data_ptr(),
                                                       sel.data_ptr(), aimg.data_ptr(), L["ALPHAV"][t].data_ptr(),
                                                       L["ATT2C"][t].data_ptr(), L["ATT2V"][t].data_ptr(), bt, Tc, R, F,
                                                       D, Adim, ws_c.data_ptr(), ws_c.numel(), st),
                  "set_editnet_attentions_train_f32")
            ops.pack(L["WHC"][t], bt, [emb, h1, cx])
            x2 = L["X2"][t]
            ops.pack(x2, bt, [h1, gated, aimg])
            check(lib.set_copy_lstm_train_f32(wref, x2.data_ptr(), K2, K2, L["H2"][t].data_ptr(), L["C2"][t].data_ptr(),
                                              sel.data_ptr(), L["H2"][t + 1].data_ptr(), L["C2"][t + 1].data_ptr(),
                                              L["G2"][t].data_ptr(), L["CNEW"][t].data_ptr(), L["CG"][t].data_ptr(), bt, D,
                                              ws_k.data_ptr(), ws_k.numel(), st), "set_copy_lstm_train_f32")
            if train and cfg.p_out > 0:
                ops.dropout(L["H2"][t + 1], L["H2D"][t], bt, D, cfg.p_out, cfg.seed, scale_off(3, t))
            if ss:                         # the next step may sample from this step's scores: fc per step
                hz = L["H2D"][t] if (train and cfg.p_out > 0) else L["H2"][t + 1]
                ops.linear(hz, P["fc_w"], P["fc_b"], pred_tb[t], bt)
            if ro is not None:             # editnet_rl.py:514-547: scores of this step, then the device sampling epilogue
                hz = L["H2D"][t] if (train and cfg.p_out > 0) else L["H2"][t + 1]
                ops.linear(hz, P["fc_w"], P["fc_b"], L["LOGITS"][t], B)
                check(lib.set_sample_pick_f32(L["LOGITS"][t].data_ptr(), V, B, V, t, T, state.end_idx, state.seed, state.offset,
                                              state.seq.data_ptr(), state.tokens[t + 1].data_ptr(), state.unfinished.data_ptr(),
                                              state.alive.data_ptr(), L["RAW"][t].data_ptr(), L["LSE"][t].data_ptr(),
                                              L["LOGP"][t].data_ptr(), st), "set_sample_pick_f32")
        hout = L["H2D"] if (train and cfg.p_out > 0) else L["H2"][1:]
        if ro is not None:
            ctx.cfg, ctx.L, ctx.bts, ctx.uniform, ctx.hout = cfg, L, bts, True, hout
            ctx.dims = (T, B, R, F, Tc, D, Adim, V)
            ctx.tokens = state.tokens
            ctx.save_for_backward(X, H, Mem, mask, att1_c, Yin, caps, *params)
            ctx.mark_non_differentiable(state.seq)
            return state.seq, L["LOGP"].t()
        if ss:
            cfg.fed_tokens = L["TOK"]          # (T, B): the words the steps consumed (tests / diagnostics)
            out = pred_tb.transpose(0, 1)
        elif uniform:                      # fc over all timesteps at once: (T, B, V), returned as its (B, T, V) view
            pred_tb = _e(T, B, V, dev=dev)
            ops.linear(hout.reshape(T * B, D), P["fc_w"], P["fc_b"], pred_tb.view(T * B, V), T * B)
            out = pred_tb.transpose(0, 1)
        else:                              # rows that left the batch keep zero scores (editnet.py:547)
            out = _z(B, T, V, dev=dev)
            for t in range(T):
                ops.linear(hout[t], P["fc_w"], P["fc_b"], out[:, t], bts[t])
        ctx.cfg, ctx.L, ctx.bts, ctx.uniform, ctx.hout = cfg, L, bts, uniform, hout
        ctx.dims = (T, B, R, F, Tc, D, Adim, V)
        ctx.save_for_backward(X, H, Mem, mask, att1_c, Yin, caps, *params)
        if adaptive:                       # h2 of every row at its last step (slot len_b of the state log), row order
            idx = torch.tensor(lens, dtype=torch.long, device=dev)
            last_h = L["H2"][idx, torch.arange(B, device=dev)]
            return out, last_h
        return out

    @staticmethod
    def backward(ctx, dpred, dlogp=None):
        X, H, Mem, mask, att1_c, Yin, caps = ctx.saved_tensors[:7]
        params = ctx.saved_tensors[7:]
        P = dict(zip(PARAM_NAMES, params))
        cfg, L, bts = ctx.cfg, ctx.L, ctx.bts
        T, B, R, F, Tc, D, Adim, V = ctx.dims
        dev = X.device
        ops = _Ops(dev)
        lib, st = ops.lib, ops.st
        train = cfg.train
        K1, K2 = 3 * D + F, 2 * D + F

        need_p = ctx.needs_input_grad[10:]
        # ---- d scores as (T, B, V), rows of finished sequences zero
        if cfg.rollout is not None:        # d seq_logp -> d scores of every step (sampling epilogue backward)
            dl = dlogp.t().contiguous()                            # (T, B)
            dp = A.zero_padded_rows(T, B, V, dev)          # rows padded to 16 bytes: the fc contractions read them in place
            for t in range(T):
                check(lib.set_sample_logp_bwd_f32(L["LOGITS"][t].data_ptr(), V, L["LSE"][t].data_ptr(), L["RAW"][t].data_ptr(),
                                                  dl[t].data_ptr(), dp[t].data_ptr(), dp.stride(1), B, V, st),
                      "set_sample_logp_bwd_f32")
            L["LOGITS"] = None
            dp2 = dp.as_strided((T * B, V), (dp.stride(1), 1), dp.storage_offset())
        else:                              # the (B, T, V) gradient of the scores, as (t, b) rows
            dp2 = A.score_grad_rows(dpred, bts, ctx.uniform)
        # ---- fc: dH2D for all timesteps in one contraction; fc.weight's gradient is final here (eager: its all-reduce
        # runs underneath the loop below in the data-parallel step)
        dH2D = A._dgrad(dp2, P["fc_w"]).view(T, B, D)
        i_w, i_b = PARAM_NAMES.index("fc_w"), PARAM_NAMES.index("fc_b")
        g_fc_b = A._bgrad(params[i_b], dp2) if need_p[i_b] else None
        g_fc_w = A._wgrad(params[i_w], dp2, ctx.hout.reshape(T * B, D), eager=True) if need_p[i_w] else None

        # ---- gradient logs (zero rows where a sequence has left the batch) and running accumulators
        _zl = _e if ctx.uniform else _z
        DG1, DGW = _zl(T, B, 4 * D, dev=dev), _zl(T, B, 4 * D, dev=dev)
        DU = _zl(T, B, D, dev=dev)
        # context-gate factor gradients side by side, [ds | dz | dt]: [ds | dz] and [dz | dt] are then contiguous column
        # ranges, and each of d(context), d(word), dh1 is ONE contraction against two stacked weight blocks
        DSZT = _zl(T, B, 3 * D, dev=dev)
        DWFC, DWFV = _zl(T, B, Adim, dev=dev), _zl(T, B, Adim, dev=dev)
        DATT2 = _zl(T, B, 2 * Adim, dev=dev)                   # [visual | caption] decoder-projection gradients side by side
        dec_cat = torch.cat([P["va_dec_w"], P["ca_dec_w"]], 0) # (2A, D): both land in dh1 through ONE contraction per step
        DEC, DEV = _zl(T, B, Tc, dev=dev), _zl(T, B, R, dev=dev)
        DEMBRAW = _zl(T, B, D, dev=dev)
        DCTX = _zl(T, B, D, dev=dev)                           # d(caption context) per step: dH = sum_t alpha_t (x) dctx_t, once
        DATT1 = _zl(T, B, R, Adim, dev=dev) if train else None
        dMem = torch.zeros_like(Mem)
        datt1c = torch.zeros_like(att1_c)
        dYin = torch.zeros_like(Yin)                           # train: d relu(att_embed(X)); eval: d features_att(.)
        zero6 = _z(6, B, D, dev=dev)                            # (one fill instead of six)
        DH1, DH2 = zero6[0], zero6[1]
        DC1 = [zero6[2], zero6[3]]
        DC2 = [zero6[4], zero6[5]]
        dcm, dcn, dop = (_e(B, D, dev=dev) for _ in range(3))
        dgated, daimg = _e(B, D, dev=dev), _e(B, F, dev=dev)
        demb, dalc = _e(B, D, dev=dev), _e(B, Tc, dev=dev)
        # d(region embedding) = datt1 . W_fa does not feed the recurrence: with (nearly) full batches it is contracted for all
        # timesteps at once after the loop (87552 rows fill the chip; 4608-row products leave it 2.25 rounds of tiles)
        dfe_after = train and sum(bts) >= _ATT1_HOIST_LIVE * T * B
        dfe = _e(B * R, D, dev=dev) if (train and not dfe_after) else None
        gate_w, tc_w, x2h_w, wih = P["ca_gate_w"], P["ca_tc_w"], P["cl_x2h_w"], P["al_wih"]
        w_ctx = torch.cat([P["ca_sc_w"], gate_w[:, 2 * D:]], 0)        # (2D, D) against [ds | dz]
        w_word = torch.cat([gate_w[:, :D], tc_w[:, :D]], 0)            # (2D, D) against [dz | dt]
        w_h1 = torch.cat([gate_w[:, D:2 * D], tc_w[:, D:]], 0)         # (2D, D) against [dz | dt]
        sc_out = 1.0 / (1.0 - cfg.p_out) if (train and cfg.p_out > 0) else 1.0
        sc_emb = 1.0 / (1.0 - cfg.p_embed) if (train and cfg.p_embed > 0) else 1.0
        sc_reg = 1.0 / (1.0 - cfg.p_region) if train else 1.0
        va_full = P["va_full_w"].reshape(-1)
        ca_full = P["ca_full_w"].reshape(-1)

        def gg(items):
            """[(dy, w_view, out, accumulate)] -> out (+)= dy . w, one grouped launch"""
            A.gemm_group([(dy, wv, dy.shape[0], wv.shape[1], dy.shape[1], out, acc) for dy, wv, out, acc in items], False, True)

        dlast = dlogp.contiguous() if (cfg.rollout is None and cfg.adaptive and dlogp is not None) else None   # 2nd output (adaptive)
        pidx = {n: i for i, n in enumerate(PARAM_NAMES)}
        need = ctx.needs_input_grad[10:]                       # frozen parameters (requires_grad False) get no gradient
        mid = T // 2 if (_WGRAD_OVERLAP and T >= 6 and dev.type == "cuda") else 0
        early = None                                           # (side stream, first row of the early share)

        def wgrad_specs(r0, r1):
            """(parameter name, dy rows, x rows, first column of the block in the weight or None) of the time-batched
            weight gradients over the log rows [r0, r1) — (t, b) rows, t-major"""
            TBs = slice(r0, r1)
            dg1 = DG1.view(T * B, 4 * D)[TBs]
            dgw = DGW.view(T * B, 4 * D)[TBs]
            du = DU.view(T * B, D)[TBs]
            whc = L["WHC"].view(T * B, 3 * D)[TBs]
            dszt = DSZT.view(T * B, 3 * D)[TBs]
            h1_all = L["H1"][1:].reshape(T * B, D)[TBs]
            datt2 = DATT2.view(T * B, 2 * Adim)[TBs]
            specs = [("al_wih", dg1, L["EMB"].view(T * B, D)[TBs], 0), ("al_wih", dg1, L["H2"][:T].reshape(T * B, D)[TBs], 2 * D),
                     ("al_whh", dg1, L["H1"][:T].reshape(T * B, D)[TBs], None),
                     ("cl_x2h_w", dgw, L["X2"].view(T * B, K2)[TBs], None), ("cl_h2h_w", dgw, L["H2"][:T].reshape(T * B, D)[TBs], None),
                     ("cl_cnew_w", du, L["CNEW"].view(T * B, D)[TBs], None), ("cl_cmem_w", du, L["SEL"].view(T * B, D)[TBs], None),
                     ("ca_gate_w", dszt[:, D:2 * D], whc, None), ("ca_tc_w", dszt[:, 2 * D:], whc[:, :2 * D], None),
                     ("ca_sc_w", dszt[:, :D], whc[:, 2 * D:], None),
                     ("ca_dec_w", datt2[:, Adim:], h1_all, None), ("va_dec_w", datt2[:, :Adim], h1_all, None)]
            if train:
                specs.append(("va_fa_w", DATT1.view(T * B * R, Adim)[r0 * R:r1 * R], L["FE"].view(T * B * R, D)[r0 * R:r1 * R], None))
            return [sp for sp in specs if need[pidx[sp[0]]]]

        def launch_early():
            """the later timesteps' share of every weight gradient, on the side stream, straight into `.grad`"""
            cur = torch.cuda.current_stream(dev)
            side = _side_stream(dev)
            targets, seen = [], {}
            for name, dy, x, c0 in wgrad_specs(mid * B, T * B):
                prm = params[pidx[name]]
                if id(prm) not in seen:
                    seen[id(prm)] = prm.grad is None
                    if prm.grad is None:                           # allocated on the caller's stream, first written on the side one
                        # (column-block weights keep blocks that only the final pass fills: those start from zero)
                        prm.grad = torch.zeros_like(prm) if c0 is not None else torch.empty_like(prm)
                targets.append((prm, dy, x, c0, seen[id(prm)] and c0 is None))
            ev = torch.cuda.Event()
            ev.record(cur)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                for prm, dy, x, c0, overwrite in targets:
                    Kb = x.shape[1]
                    out = prm.grad if c0 is None else prm.grad[:, c0:c0 + Kb]
                    A.gemm(dy, True, x, True, prm.shape[0], Kb, dy.shape[0], out=out, accumulate=not overwrite)
            return side

        slab_direct = _SLAB_DIRECT and _DEMB_HOIST and dev.type == "cuda"
        DLAST = None
        if dlast is not None and slab_direct:
            # d(decoder_last_hidden) enters h2 of every row's LAST timestep: a (T, B, D) log that is zero elsewhere, handed to the
            # copy-gate backward of every timestep as one more addend of dh2
            DLAST = _z(T, B, D, dev=dev)
            DLAST[torch.tensor([l - 1 for l in cfg.decode_lengths], dtype=torch.long, device=dev), torch.arange(B, device=dev)] = dlast
        c_loop = slab_direct and _C_LOOPS and (dfe_after or not train) and not mid
        if c_loop:
            a = _lib.XEBwdLoopArgs()
            a.T, a.B, a.R, a.F, a.Tc, a.D, a.A, a.acc_datt1 = T, B, R, F, Tc, D, Adim, 0 if train else 1
            a.p_out = float(cfg.p_out if (train and cfg.p_out > 0) else 0.0)
            a.seed, a.off_out = int(cfg.seed), rng.offset(rng.SITE_OUT, 0)
            bts_c = (C.c_int * T)(*bts)
            a.bts = C.cast(bts_c, C.c_void_p)
            a.cl_cnew_w, a.cl_cmem_w, a.cl_x2h_w, a.cl_h2h_w = (P[k].data_ptr() for k in ("cl_cnew_w", "cl_cmem_w", "cl_x2h_w", "cl_h2h_w"))
            a.w_ctx, a.w_h1, a.dec_cat = w_ctx.data_ptr(), w_h1.data_ptr(), dec_cat.data_ptr()
            a.al_wih, a.al_whh, a.va_full, a.ca_full = wih.data_ptr(), P["al_whh"].data_ptr(), va_full.data_ptr(), ca_full.data_ptr()
            for k in ("G1", "G2", "C1", "C2", "CG", "SEL", "CNEW", "ZT", "S", "TT", "ALPHAC", "ALPHAV", "ATT2C", "ATT2V"):
                setattr(a, k, L[k].data_ptr())
            a.X, a.H, a.Mem, a.att1_c = X.data_ptr(), H.data_ptr(), Mem.data_ptr(), att1_c.data_ptr()
            a.att1, a.att1_step = (L["ATT1"].data_ptr(), B * R * Adim) if train else (Yin.data_ptr(), 0)
            a.dH2D = dH2D.data_ptr()
            a.DLAST = DLAST.data_ptr() if DLAST is not None else None
            a.DU, a.DGW, a.DSZT, a.DATT2, a.DWFC, a.DWFV = (x.data_ptr() for x in (DU, DGW, DSZT, DATT2, DWFC, DWFV))
            a.DEC, a.DEV, a.DCTX, a.DG1 = DEC.data_ptr(), DEV.data_ptr(), DCTX.data_ptr(), DG1.data_ptr()
            a.datt1, a.datt1_step = (DATT1.data_ptr(), B * R * Adim) if train else (dYin.data_ptr(), 0)
            a.datt1c, a.dMem = datt1c.data_ptr(), dMem.data_ptr()
            a.DC1[0], a.DC1[1], a.DC2[0], a.DC2[1] = DC1[0].data_ptr(), DC1[1].data_ptr(), DC2[0].data_ptr(), DC2[1].data_ptr()
            a.dcm, a.dcn, a.dop, a.dalc = dcm.data_ptr(), dcn.data_ptr(), dop.data_ptr(), dalc.data_ptr()
            keep = [_slab_scratch(dev, i) for i in range(5)]
            for i, w_ in enumerate(keep):
                a.slab_ws[i] = w_.data_ptr()
            a.slab_ws_bytes = keep[0].numel()
            tmp_n = (D, D, D, D, F, D, D, D, D, D, D)
            tmps = [_e(B, n_, dev=dev) for n_ in tmp_n]
            for i, t_ in enumerate(tmps):
                a.tmp[i] = t_.data_ptr()
            check(lib.set_editnet_xe_train_bwd_loop_f32(C.byref(a), st), "set_editnet_xe_train_bwd_loop_f32")
        if slab_direct and not c_loop:
            SS = _lib.SlabSrc
            tmp = {}                       # (slot, task) -> (B, N) buffer a product lands in when the plan does not split it

            def gslabs(slot, items):
                """[(dy, w_view, N)] -> one SlabSrc per problem: its split-K partials in this slot's scratch, or (unsplit) the
                product itself; ONE launch, no reduction launch"""
                n = len(items)
                descs = (_lib.GemmDesc * n)()
                outs = (SS * n)()
                for i, (dy, wv) in enumerate(items):
                    M, N, K = dy.shape[0], wv.shape[1], dy.shape[1]
                    o = tmp.get((slot, i))
                    if o is None:
                        o = tmp[(slot, i)] = _e(B, N, dev=dev)
                    a, lda = A._mat(dy)
                    b_, ldb = A._mat(wv)
                    descs[i] = _lib.GemmDesc(a.data_ptr(), lda, b_.data_ptr(), ldb, o.data_ptr(), N, M, N, K, 0)
                ws = _slab_scratch(dev, slot)
                check(lib.set_gemm_group_slabs_f32(descs, n, 0, 1, ws.data_ptr(), ws.numel(), outs, st), "set_gemm_group_slabs_f32")
                res = []
                for i, (dy, wv) in enumerate(items):
                    e = outs[i]
                    if e.nslab == 0:       # written whole: one "partial"
                        e = SS(tmp[(slot, i)].data_ptr(), 0, wv.shape[1], 1, dy.shape[0])
                    else:
                        e = SS(e.p, e.slab_stride, e.ld, e.nslab, e.rows)
                    res.append(e)
                return res

            def srcs(lst):
                lst = [e for e in lst if e is not None]
                return ((SS * len(lst))(*lst) if lst else None), len(lst)

            p_out = cfg.p_out if (train and cfg.p_out > 0) else 0.0
            nxt_dh2, nxt_dh1 = [], None    # addends of dh2 / the W_hh term of dh1 produced by the timestep after this one
            for t in range(T - 1, -1, -1):
                bt = bts[t]
                r = lambda x: _rows(x, bt)
                du, dgw = DU[t], DGW[t]
                dc2_in, dc2_out = DC2[t & 1], DC2[(t & 1) ^ 1]
                # ---- CopyLSTMCellC backward: dh2 = recurrent addends + the output dropout's backward of d fc-input (fused)
                a_, n_ = srcs(nxt_dh2 + ([SS(DLAST[t].data_ptr(), 0, D, 1, bt)] if DLAST is not None else []))
                check(lib.set_copy_gate_bwd_src_f32(a_, n_, dH2D[t].data_ptr(), D, p_out, cfg.seed, rng.offset(rng.SITE_OUT, t),
                                                    dc2_in.data_ptr(), L["G2"][t][:, 3 * D:].data_ptr(), 4 * D,
                                                    L["C2"][t + 1].data_ptr(), L["CG"][t].data_ptr(), L["SEL"][t].data_ptr(),
                                                    L["CNEW"][t].data_ptr(), du.data_ptr(), dcm.data_ptr(), dcn.data_ptr(),
                                                    dop.data_ptr(), bt, D, st), "set_copy_gate_bwd_src_f32")
                g3 = gslabs(0, [(r(du), P["cl_cnew_w"]), (r(du), P["cl_cmem_w"])])
                a_, n_ = srcs([g3[0]])
                check(lib.set_lstm_gates_bwd_src_f32(dcn.data_ptr(), a_, n_, dop.data_ptr(), L["G2"][t].data_ptr(),
                                                     L["C2"][t].data_ptr(), dgw.data_ptr(), dc2_out.data_ptr(), bt, D, st),
                      "set_lstm_gates_bwd_src_f32")
                g5 = gslabs(1, [(r(dgw), x2h_w[:, :D]), (r(dgw), x2h_w[:, D:2 * D]), (r(dgw), x2h_w[:, 2 * D:]), (r(dgw), P["cl_h2h_w"])])
                # ---- SelectC backward: dMem += ..., dalpha_c
                a_, n_ = srcs([g3[1]])
                check(lib.set_select_bwd_src_f32(dcm.data_ptr(), a_, n_, Mem.data_ptr(), L["ALPHAC"][t].data_ptr(), dMem.data_ptr(),
                                                 dalc.data_ptr(), bt, Tc, D, 1, st), "set_select_bwd_src_f32")
                # ---- VisualAttentionC backward
                att1 = L["ATT1"][t] if train else Yin
                datt1 = DATT1[t] if train else dYin
                a_, n_ = srcs([g5[2]])
                check(lib.set_attention_bwd_src_f32(None, a_, n_, None, None, L["ALPHAV"][t].data_ptr(), X.data_ptr(), att1.data_ptr(),
                                                    L["ATT2V"][t].data_ptr(), va_full.data_ptr(), datt1.data_ptr(),
                                                    DATT2[t].data_ptr(), DWFV[t].data_ptr(), DEV[t].data_ptr(), bt, R, F, Adim, 0,
                                                    0 if train else 1, 2 * Adim, st), "set_attention_bwd_src_f32")
                if train and not dfe_after:
                    A.gemm(datt1.view(B * R, Adim)[:bt * R], False, P["va_fa_w"], True, bt * R, D, Adim, out=dfe[:bt * R])
                    ops.dropout_bwd(dfe, L["FE"][t].view(B * R, D), dYin.view(B * R, D), bt * R, D, sc_reg, True)
                # ---- CaptionAttentionC backward
                dszt = DSZT[t]
                a_, n_ = srcs([g5[1]])
                check(lib.set_context_gate_bwd_src_f32(None, a_, n_, L["ZT"][t].data_ptr(), L["S"][t].data_ptr(), L["TT"][t].data_ptr(),
                                                       dszt[:, D:].data_ptr(), dszt.data_ptr(), dszt[:, 2 * D:].data_ptr(), 3 * D, bt,
                                                       D, st), "set_context_gate_bwd_src_f32")
                g9 = gslabs(2, [(r(dszt)[:, :2 * D], w_ctx), (r(dszt)[:, D:], w_h1)])
                a_, n_ = srcs([g9[0]])
                check(lib.set_attention_bwd_src_f32(None, a_, n_, DCTX[t].data_ptr(), dalc.data_ptr(), L["ALPHAC"][t].data_ptr(),
                                                    H.data_ptr(), att1_c.data_ptr(), L["ATT2C"][t].data_ptr(), ca_full.data_ptr(),
                                                    datt1c.data_ptr(), DATT2[t][:, Adim:].data_ptr(), DWFC[t].data_ptr(),
                                                    DEC[t].data_ptr(), bt, Tc, D, Adim, 1, 1, 2 * Adim, st),
                      "set_attention_bwd_src_f32")
                g11 = gslabs(3, [(r(DATT2[t]), dec_cat)])
                # ---- attention LSTM backward: dh1 = W_hh term of the next timestep + the three addends of this one
                dc1_in, dc1_out = DC1[t & 1], DC1[(t & 1) ^ 1]
                a_, n_ = srcs([nxt_dh1, g5[0], g9[1], g11[0]])
                check(lib.set_lstm_cell_bwd_src_f32(a_, n_, dc1_in.data_ptr(), L["G1"][t].data_ptr(), L["C1"][t].data_ptr(),
                                                    L["C1"][t + 1].data_ptr(), DG1[t].data_ptr(), dc1_out.data_ptr(), bt, D, st),
                      "set_lstm_cell_bwd_src_f32")
                dg1 = r(DG1[t])
                g13 = gslabs(4, [(dg1, wih[:, 2 * D:3 * D]), (dg1, P["al_whh"])])
                nxt_dh2, nxt_dh1 = [g5[3], g13[0]], g13[1]
                if mid and t == mid and all(A._is_leaf_param(params[i]) for i in range(len(params)) if need[i]):
                    early = launch_early()
        for t in (range(T - 1, -1, -1) if not slab_direct else ()):
            bt = bts[t]
            r = lambda x: _rows(x, bt)
            h1 = L["H1"][t + 1]
            if dlast is not None:          # rows whose last step this is: d(decoder_last_hidden) joins dh2 here
                b0 = bts[t + 1] if t + 1 < T else 0
                if bt > b0:
                    ops.pack(DH2[b0:], bt - b0, [dlast[b0:bt]], accumulate=True)
            # h2(t) -> fc (through the output dropout); the recurrent / next-step terms are already in DH2
            if train and cfg.p_out > 0:    # h2 does not follow a ReLU: the mask is regenerated, not read off the zero pattern
                check(lib.set_dropout_bwd_philox_f32(dH2D[t].data_ptr(), D, DH2.data_ptr(), D, bt, D, cfg.p_out, cfg.seed,
                                                     rng.offset(rng.SITE_OUT, t), 1, st), "set_dropout_bwd_philox_f32")
            else:
                ops.pack(DH2, bt, [dH2D[t]], accumulate=True)
            # ---- CopyLSTMCellC backward (editnet.py:265-285)
            du, dgw = DU[t], DGW[t]
            dc2_in, dc2_out = DC2[t & 1], DC2[(t & 1) ^ 1]
            check(lib.set_copy_gate_bwd_ld_f32(DH2.data_ptr(), dc2_in.data_ptr(), L["G2"][t][:, 3 * D:].data_ptr(), 4 * D,
                                               L["C2"][t + 1].data_ptr(),
                                               L["CG"][t].data_ptr(), L["SEL"][t].data_ptr(), L["CNEW"][t].data_ptr(),
                                               du.data_ptr(), dcm.data_ptr(), dcn.data_ptr(), dop.data_ptr(), bt, D, st),
                  "set_copy_gate_bwd_ld_f32")
            gg([(r(du), P["cl_cnew_w"], r(dcn), True), (r(du), P["cl_cmem_w"], r(dcm), True)])
            check(lib.set_lstm_gates_bwd_f32(dcn.data_ptr(), dop.data_ptr(), L["G2"][t].data_ptr(), L["C2"][t].data_ptr(),
                                             dgw.data_ptr(), dc2_out.data_ptr(), bt, D, st), "set_lstm_gates_bwd_f32")
            gg([(r(dgw), x2h_w[:, :D], r(DH1), True), (r(dgw), x2h_w[:, D:2 * D], r(dgated), False),
                (r(dgw), x2h_w[:, 2 * D:], r(daimg), False), (r(dgw), P["cl_h2h_w"], r(DH2), False)])
            # ---- SelectC backward: dMem += ..., dalpha_c
            check(lib.set_select_bwd_acc_f32(dcm.data_ptr(), Mem.data_ptr(), L["ALPHAC"][t].data_ptr(), dMem.data_ptr(),
                                             dalc.data_ptr(), bt, Tc, D, 1, st), "set_select_bwd_acc_f32")
            # ---- VisualAttentionC backward
            att1 = L["ATT1"][t] if train else Yin
            datt1 = DATT1[t] if train else dYin
            check(lib.set_attention_bwd_acc_f32(daimg.data_ptr(), None, L["ALPHAV"][t].data_ptr(), X.data_ptr(), att1.data_ptr(),
                                                L["ATT2V"][t].data_ptr(), va_full.data_ptr(), datt1.data_ptr(), DATT2[t].data_ptr(),
                                                DWFV[t].data_ptr(), None, DEV[t].data_ptr(), bt, R, F, Adim, 0,
                                                0 if train else 1, 0, 2 * Adim, st), "set_attention_bwd_acc_f32")
            if train and not dfe_after:
                A.gemm(datt1.view(B * R, Adim)[:bt * R], False, P["va_fa_w"], True, bt * R, D, Adim, out=dfe[:bt * R])
                ops.dropout_bwd(dfe, L["FE"][t].view(B * R, D), dYin.view(B * R, D), bt * R, D, sc_reg, True)
            # ---- CaptionAttentionC backward
            dszt = DSZT[t]
            check(lib.set_context_gate_bwd_ld_f32(dgated.data_ptr(), L["ZT"][t].data_ptr(), L["S"][t].data_ptr(),
                                                  L["TT"][t].data_ptr(), dszt[:, D:].data_ptr(), dszt.data_ptr(),
                                                  dszt[:, 2 * D:].data_ptr(), 3 * D, bt, D, st), "set_context_gate_bwd_ld_f32")
            dctx = DCTX[t]
            if _DEMB_HOIST:                # (d emb feeds no recurrence: its two products run once over all timesteps below)
                gg([(r(dszt)[:, :2 * D], w_ctx, r(dctx), False), (r(dszt)[:, D:], w_h1, r(DH1), True)])
            else:
                gg([(r(dszt)[:, :2 * D], w_ctx, r(dctx), False), (r(dszt)[:, D:], w_word, r(demb), False),
                    (r(dszt)[:, D:], w_h1, r(DH1), True)])
            check(lib.set_attention_bwd_acc_f32(dctx.data_ptr(), dalc.data_ptr(), L["ALPHAC"][t].data_ptr(), H.data_ptr(),
                                                att1_c.data_ptr(), L["ATT2C"][t].data_ptr(), ca_full.data_ptr(), datt1c.data_ptr(),
                                                DATT2[t][:, Adim:].data_ptr(), DWFC[t].data_ptr(), None, DEC[t].data_ptr(), bt, Tc,
                                                D, Adim, 1, 1, 0, 2 * Adim, st), "set_attention_bwd_acc_f32")
            gg([(r(DATT2[t]), dec_cat, r(DH1), True)])
            # ---- attention LSTM backward
            dc1_in, dc1_out = DC1[t & 1], DC1[(t & 1) ^ 1]
            check(lib.set_lstm_cell_bwd_f32(DH1.data_ptr(), dc1_in.data_ptr(), L["G1"][t].data_ptr(), L["C1"][t].data_ptr(),
                                            L["C1"][t + 1].data_ptr(), DG1[t].data_ptr(), dc1_out.data_ptr(), bt, D, st),
                  "set_lstm_cell_bwd_f32")
            dg1 = r(DG1[t])
            # (the final_hidden / image_mean columns are loop-invariant: their gradients come from sum_t dgates below)
            if _DEMB_HOIST:
                gg([(dg1, wih[:, 2 * D:3 * D], r(DH2), True), (dg1, P["al_whh"], r(DH1), False)])
            else:
                gg([(dg1, wih[:, :D], r(demb), True), (dg1, wih[:, 2 * D:3 * D], r(DH2), True), (dg1, P["al_whh"], r(DH1), False)])
                # ---- embedding: dropout + ReLU backward; the table rows are scattered once after the loop
                ops.dropout_bwd(demb, L["EMB"][t], DEMBRAW[t], bt, D, sc_emb, False)
            if mid and t == mid and all(A._is_leaf_param(params[i]) for i in range(len(params)) if need[i]):
                early = launch_early()

        # dH[b, l, :] = sum_t alpha_c[t, b, l] dctx[t, b, :]: one batched (Tc x T)(T x D) product per sample over the logs
        # instead of a read-modify-write of all of dH in every timestep
        if dfe_after:                      # rows of finished sequences are zero in DATT1, hence in DFE
            DFE = A.gemm(DATT1.view(T * B * R, Adim), False, P["va_fa_w"], True, T * B * R, D, Adim).view(T, B * R, D)
            if min(bts) == B:              # all rows live at every timestep: the T accumulating launches as one (same sums, same order)
                check(lib.set_dropout_bwd_steps_f32(DFE.data_ptr(), D, B * R * D, L["FE"].data_ptr(), D, B * R * D, dYin.data_ptr(), D,
                                                    B * R, D, T, sc_reg, 1, st), "set_dropout_bwd_steps_f32")
            else:
                for t in range(T):
                    ops.dropout_bwd(DFE[t], L["FE"][t].view(B * R, D), dYin.view(B * R, D), bts[t] * R, D, sc_reg, True)
            del DFE
        if _DEMB_HOIST:
            # d emb[t] = [dz | dt](t) . [gate_w[:, :D]; tc_w[:, :D]] + dgates1(t) . W_ih[:, :D] is consumed by nothing inside the
            # loop (the word fed at t + 1 is data, not a function of emb[t]): two products over all T * B rows (rows of finished
            # sequences are zero in DSZT / DG1) instead of 2 x T products of 128 rows on the critical chain, and ONE dropout +
            # ReLU backward over the logs
            DEMB = A.gemm(DSZT.view(T * B, 3 * D)[:, D:], False, w_word, True, T * B, D, 2 * D)
            A.gemm(DG1.view(T * B, 4 * D), False, wih[:, :D], True, T * B, D, 4 * D, out=DEMB, accumulate=True)
            ops.dropout_bwd(DEMB, L["EMB"].view(T * B, D), DEMBRAW.view(T * B, D), T * B, D, sc_emb, False)
            del DEMB
        dH = _dvalues(L["ALPHAC"], DCTX, ops)
        # loop-invariant inputs of the attention LSTM: d final_hidden = (sum_t dgates) . W_ih[:, D:2D]
        sdg1 = DG1.sum(0)
        dFH = A.gemm(sdg1, False, wih[:, D:2 * D], True, B, D, 4 * D)
        # ---- parameter gradients: one contraction per parameter over all (t, b) rows (over the rows of the earlier timesteps
        # only, accumulated onto the side stream's share, when that was launched half way through the loop)
        g = [None] * len(PARAM_NAMES)
        TB = T * B
        hi = TB
        if early is not None:
            torch.cuda.current_stream(dev).wait_stream(early)
            hi = mid * B

        def W(name, dy, x):
            if need[pidx[name]]:
                g[pidx[name]] = A._wgrad(params[pidx[name]], dy[:hi], x[:hi])

        def Bg(name, dy):
            if need[pidx[name]]:
                g[pidx[name]] = A._bgrad(params[pidx[name]], dy)

        g[pidx["fc_w"]], g[pidx["fc_b"]] = g_fc_w, g_fc_b
        ids = caps[:, :T].t().reshape(-1) if cfg.rollout is None else ctx.tokens[:T].reshape(-1)
        if "TOK" in L:                     # scheduled sampling: the words actually fed
            ids = L["TOK"].reshape(-1)
        if need[pidx["E"]]:
            dE = torch.zeros_like(P["E"])
            dE.index_add_(0, ids, DEMBRAW.view(TB, D))
            g[pidx["E"]] = dE
        dg1 = DG1.view(TB, 4 * D)
        if need[pidx["al_wih"]]:           # column blocks [emb | final_hidden | h2 | image_mean]; the invariant ones from sum_t
            g[pidx["al_wih"]] = A._wgrad_blocks(params[pidx["al_wih"]], [
                (dg1[:hi], L["EMB"].view(TB, D)[:hi], 0), (sdg1, L["FH"], D), (dg1[:hi], L["H2"][:T].reshape(TB, D)[:hi], 2 * D),
                (sdg1, L["MEAN"], 3 * D)])
        W("al_whh", dg1, L["H1"][:T].reshape(TB, D))
        Bg("al_bih", dg1); Bg("al_bhh", dg1)
        dgw = DGW.view(TB, 4 * D)
        W("cl_x2h_w", dgw, L["X2"].view(TB, K2)); W("cl_h2h_w", dgw, L["H2"][:T].reshape(TB, D))
        Bg("cl_x2h_b", dgw); Bg("cl_h2h_b", dgw)
        du = DU.view(TB, D)
        W("cl_cnew_w", du, L["CNEW"].view(TB, D)); W("cl_cmem_w", du, L["SEL"].view(TB, D))
        Bg("cl_cnew_b", du); Bg("cl_cmem_b", du)
        whc = L["WHC"].view(TB, 3 * D)
        dszt = DSZT.view(TB, 3 * D)
        d_s, d_z, d_t = dszt[:, :D], dszt[:, D:2 * D], dszt[:, 2 * D:]
        W("ca_gate_w", d_z, whc); Bg("ca_gate_b", d_z)
        W("ca_tc_w", d_t, whc[:, :2 * D]); Bg("ca_tc_b", d_t)
        W("ca_sc_w", d_s, whc[:, 2 * D:]); Bg("ca_sc_b", d_s)
        h1_all = L["H1"][1:].reshape(TB, D)
        datt2 = DATT2.view(TB, 2 * Adim)
        W("ca_dec_w", datt2[:, Adim:], h1_all); Bg("ca_dec_b", datt2[:, Adim:])
        W("va_dec_w", datt2[:, :Adim], h1_all); Bg("va_dec_b", datt2[:, :Adim])
        Bg("ca_full_w", DWFC.view(TB, Adim))
        if need[pidx["ca_full_b"]]:
            g[pidx["ca_full_b"]] = DEC.sum().reshape(1)
        Bg("va_full_w", DWFV.view(TB, Adim))
        if need[pidx["va_full_b"]]:
            g[pidx["va_full_b"]] = DEV.sum().reshape(1)
        if train:
            if need[pidx["va_fa_w"]]:
                g[pidx["va_fa_w"]] = A._wgrad(params[pidx["va_fa_w"]], DATT1.view(TB * R, Adim)[:hi * R], L["FE"].view(TB * R, D)[:hi * R])
            Bg("va_fa_b", DATT1.view(TB * R, Adim))
        ctx.L = None
        # inputs: cfg, X, mean, H, Mem, final_hidden, mask, att1_c, Yin, caps
        return (None, None, None, dH, dMem, dFH, None, datt1c, dYin, None) + tuple(g)


def xe_sequence(cfg, X, mean, H, Mem, final_hidden, mask, att1_c, Yin, caps, params):
    return _XESequence.apply(cfg, X, mean, H, Mem, final_hidden, mask, att1_c, Yin, caps, *params)
