/*
 * set_hip.h — C ABI of libset_hip.so: the EditNet / DCNet per-timestep decode path of
 * show-edit-tell as hand-written HIP kernels for MI355X (gfx950).
 *
 * The reference (fawazsammani/show-edit-tell) has no FFI: its boundary for this path is the
 * Python nn.Module surface.  Each entry point below replaces the PyTorch ops
 * behind one of those module calls; the reference file:line it replaces is cited on each.  The
 * Python host side (show-edit-tell_amd/*.py) binds these with ctypes and keeps the reference's
 * class names, constructor signatures, attribute names and state_dict keys.
 *
 * Conventions
 *   - all tensors fp32 row-major contiguous unless a leading stride is given; ids/lengths int64
 *   - weights are PyTorch (out,in) row-major and are used IN PLACE (no repacking)
 *   - every pointer is a DEVICE pointer unless named host_*; nothing is allocated or freed here:
 *     the caller supplies a workspace sized by the matching *_workspace_bytes() query
 *   - all work is enqueued on `stream` (a hipStream_t); functions never synchronise
 *   - return value: SET_OK or an error code; nothing throws across the ABI
 *   - re-entrant per (workspace, stream) pair.  State outside the caller's workspace, all of it listed here:
 *       thread-local (per calling host thread): the last hipError_t (set_last_hip_error), the row-limit pointer
 *         (set_decode_row_limits), the loop gate the decode loops set around each timestep for their own launchers, the
 *         opt-in profiler's records (set_profile_enable);
 *       process-wide, per device: one mutex + completion-event chain that admits ONE persistent launch at a time (caption
 *         encoder, small-batch decode loop: their workgroups must all be resident), the sticky host-mapped fault word and
 *         "persistent kernels disabled" flag behind SET_ERR_FAULT, cached answers of occupancy / LDS-limit queries and of
 *         "dynamic-LDS cap raised for kernel X on device d";
 *       machine-wide, per device: an advisory lock on /tmp/set_hip_persistent_<pci bus id>.lock (SET_PERSISTENT_LOCK_DIR
 *         moves it, SET_PERSISTENT_IPC_LOCK=0 switches it off) held until exit by the FIRST process that makes a persistent
 *         launch on the device; every other process sharing that GPU is answered SET_ERR_UNSUPPORTED by the persistent
 *         entry points (the per-step kernels; asked again at each call) — two half-resident grids of two processes would
 *         otherwise wait for each other until the time-out;
 *       environment: SET_* switches are read once per process, except SET_DEC_PERSISTENT / SET_DEC_PERSISTENT_MAXB, which
 *         the small-batch decode reads per call (tests flip them inside one process).
 *     None of it carries results between calls: outputs depend on the arguments only.
 *   - alignment: base pointers 16-byte aligned, leading strides and K multiples of 4 floats;
 *     every contraction length (D, A, F, 2D, ...) must be a multiple of 32 (SET_ERR_UNSUPPORTED)
 */
#ifndef SET_HIP_H
#define SET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SET_OK 0
#define SET_ERR_ARG 1          /* null pointer / non-positive size / misaligned */
#define SET_ERR_UNSUPPORTED 2  /* dimension not supported by the kernels (see alignment rules) */
#define SET_ERR_HIP 3          /* a HIP runtime call failed: see set_last_hip_error() */
#define SET_ERR_WORKSPACE 4    /* workspace too small */
#define SET_ERR_FAULT 5        /* an EARLIER call's persistent launch (caption encoder, small-batch decode loop) timed out
                                  waiting for its workgroups: that call's outputs were overwritten with NaN on the device;
                                  reported once, at the next call that would have used such a kernel; the library uses the
                                  per-step kernels from then on (retry succeeds) */

#define SET_ACT_NONE 0
#define SET_ACT_RELU 1
#define SET_ACT_TANH 2
#define SET_ACT_SIGMOID 3

int set_abi_version(void);
const char* set_error_string(int code);
/* last hipError_t seen by this thread inside the library (0 = none) and its text */
int set_last_hip_error(void);
const char* set_last_hip_error_string(void);
/* device the library was compiled for ("gfx950") */
const char* set_target_arch(void);

/* ------------------------------------------------------------------------------------------
 * Opt-in per-kernel timing (HIP events on the launch stream); used by bench.py.  Not part of the
 * reference's surface.  set_profile_enable(1) clears previous records; set_profile_report
 * synchronises the events and aggregates by kernel tag, returning the number of entries.
 * flops / bytes are the ALGORITHMIC work of the recorded launches, not counters.
 * ------------------------------------------------------------------------------------------ */
typedef struct SetProfileEntry {
    char tag[32];
    int launches;
    double ms;
    double flops;
    double bytes;
} SetProfileEntry;
int set_profile_enable(int on);
int set_profile_report(SetProfileEntry* out, int max_entries);

/* ------------------------------------------------------------------------------------------
 * EditNet (reference editnet.py:449-548 `DecoderC`, editnet_rl.py:455-549)
 * ------------------------------------------------------------------------------------------ */
typedef struct SetEditNetDims {
    int B;      /* batch rows held by the workspace                                    */
    int T;      /* padded previous-caption length (18 in the reference data, 20 in BASELINE) */
    int R;      /* regions per image (36; up to 100 for adaptive features)              */
    int F;      /* image_features_dim (2048)                                            */
    int D;      /* decoder_dim == emb_dim == caption_features_dim (1024)                */
    int A;      /* attention_dim (512)                                                  */
    int V;      /* vocabulary size                                                      */
    int maxT;   /* XE: max decode length (predictions.shape[1]); greedy: max_len + 1    */
    int adaptive; /* 1: adaptive-features visual attention (editnet_adaptive.py:438-457) */
} SetEditNetDims;

/* state_dict tensors of DecoderC, device pointers, used in place */
typedef struct SetEditNetWeights {
    const float* embed;                     /* embed.embedding.weight (V,D)                         */
    const float *enc_x2h_w, *enc_x2h_b;     /* caption_encoder.lstm_encoder_cell.x2h (4D,D)         */
    const float *enc_h2h_w, *enc_h2h_b;     /* caption_encoder.lstm_encoder_cell.h2h (4D,D)         */
    const float *enc_aff_w, *enc_aff_b;     /* caption_encoder.affine_hn (D,D)                      */
    const float *ca_feat_w, *ca_feat_b;     /* caption_attention.cap_features_att (A,D)             */
    const float *ca_dec_w, *ca_dec_b;       /* caption_attention.cap_decoder_att (A,D)              */
    const float *ca_full_w, *ca_full_b;     /* caption_attention.cap_full_att (1,A)                 */
    const float *ca_gate_w, *ca_gate_b;     /* caption_attention.context_gate (D,3D)                */
    const float *ca_sc_w, *ca_sc_b;         /* caption_attention.sc_affine (D,D)                    */
    const float *ca_tc_w, *ca_tc_b;         /* caption_attention.tc_affine (D,2D)                   */
    const float *va_emb_w, *va_emb_b;       /* visual_attention.att_embed.0 (D,F)                   */
    const float *va_feat_w, *va_feat_b;     /* visual_attention.features_att (A,D)                  */
    const float *va_dec_w, *va_dec_b;       /* visual_attention.decoder_att (A,D)                   */
    const float *va_full_w, *va_full_b;     /* visual_attention.full_att (1,A)                      */
    const float *al_wih, *al_whh;           /* attention_lstm.weight_ih (4D,3D+F), weight_hh (4D,D) */
    const float *al_bih, *al_bhh;           /* attention_lstm.bias_ih / bias_hh (4D)                */
    const float *cl_x2h_w, *cl_x2h_b;       /* copy_lstm.x2h (4D,2D+F)                              */
    const float *cl_h2h_w, *cl_h2h_b;       /* copy_lstm.h2h (4D,D)                                 */
    const float *cl_cnew_w, *cl_cnew_b;     /* copy_lstm.gate_cnew (D,D)                            */
    const float *cl_cmem_w, *cl_cmem_b;     /* copy_lstm.gate_cmem (D,D)                            */
    const float *fc_w, *fc_b;               /* fc (V,D)                                             */
    /* optional DERIVED tensor (NULL = unused): token table (V,10D) built by set_editnet_build_token_table,
     * row v = relu(E[v]) x [W_ih[:, :D] ; tc_affine.W[:, :D] ; context_gate.W[:, :D] ; enc.x2h.W]^T
     * (+ enc.x2h.bias on the last 4D columns) — the contractions whose input is only a token: three of
     * the decode step (editnet.py:527,378-379) and the encoder's input projection (editnet.py:335).
     * Valid while embed / attention_lstm.weight_ih / tc_affine / context_gate / lstm_encoder_cell.x2h
     * are unchanged (inference). */
    const float* tok_table;
} SetEditNetWeights;

size_t set_editnet_workspace_bytes(const SetEditNetDims* d);

/* Inference-time folding of the token-only contractions into a (V,10D) lookup table (see tok_table). */
size_t set_editnet_token_table_bytes(const SetEditNetDims* d);          /* bytes of the table itself   */
size_t set_editnet_token_table_workspace_bytes(const SetEditNetDims* d);
int set_editnet_build_token_table(const SetEditNetWeights* w, const SetEditNetDims* d, float* table,
                                  void* ws, size_t ws_bytes, void* stream);

/* Per-sequence prologue.  Replaces, for one batch: caption_encoder(prev, prevlen)
 * (editnet.py:319-348, called at :501), image_mean = X.mean(1) (:503), init_hidden_state (:494-495)
 * and — eval mode only — the loop-invariant projections that the reference recomputes every
 * timestep: features_att(att_embed(X)) (:441-442), cap_features_att(H) (:370) and the
 * final_hidden / image_mean columns of the attention_lstm input product (:527-532).
 * `image_mean` may be NULL (computed here) or a (B,F) tensor (adaptive variant, editnet_adaptive.py:501).
 * `prev` (B,T) int64, `prevlen` (B) int64, `X` (B,R,F).  Leaves the recurrent state zeroed. */
int set_editnet_begin(const SetEditNetWeights* w, const SetEditNetDims* d, const float* X,
                      const float* image_mean, const int64_t* prev, const int64_t* prevlen,
                      void* ws, size_t ws_bytes, void* stream);

/* One decode timestep for rows [0,bt): embed -> attention_lstm -> caption_attention ->
 * visual_attention -> select -> copy_lstm -> fc  (editnet.py:513-546, editnet_rl.py:505-513),
 * eval mode.  `tokens` (bt) int64 device ids, or NULL to feed the token the previous
 * set_editnet_greedy_pick produced (its fused embedding gather).  `X` is the same (B,R,F) feature
 * tensor given to set_editnet_begin.  `logits` (bt,V) with leading stride ld_logits. */
int set_editnet_step(const SetEditNetWeights* w, const SetEditNetDims* d, const float* X,
                     const int64_t* tokens, int64_t tokens_stride, int bt, float* logits,
                     int64_t ld_logits, void* ws, size_t ws_bytes, void* stream);

/* Greedy epilogue of one free-running step (editnet_rl.py:514-543): log_softmax, argmax (first
 * index on ties), <end> -> 0, `unfinished` latch, seq[:,t] / seqLogprobs[:,t] stores; suppressed
 * once every row has finished (the reference's `break`, :546), and the next step's embedding
 * relu(E[it]) (editnet.py:300-304) gathered into the workspace.  seq (B,max_len) int64,
 * seq_logp (B,max_len) fp32, both pre-zeroed by the caller; call with t = 0,1,2,... in order. */
int set_editnet_greedy_pick(const SetEditNetWeights* w, const SetEditNetDims* d, const float* logits,
                            int64_t ld_logits, int t, int64_t end_idx, int64_t* seq, float* seq_logp,
                            int max_len, void* ws, size_t ws_bytes, void* stream);

/* Whole free-running greedy decode (editnet_rl.py:485-549 with sample_max=True): prologue +
 * (max_len+1) timesteps, no host synchronisation (the reference syncs every step at :546).
 * Outputs seq (B,max_len) int64 and seq_logp (B,max_len) fp32 (overwritten). */
int set_editnet_greedy(const SetEditNetWeights* w, const SetEditNetDims* d, const float* X,
                       const float* image_mean, const int64_t* prev, const int64_t* prevlen,
                       int64_t start_idx, int64_t end_idx, int max_len, int64_t* seq,
                       float* seq_logp, void* ws, size_t ws_bytes, void* stream);

/* Per-row cap on the caption length of the free-running GREEDY loops (set_*_greedy) for the CALLING host thread: while
 * `row_limit_dev` (B int32 on the device; the pointer is read by the launches of later calls, keep it alive) is set, the
 * pick of timestep t takes <end> for every row b with t + 1 >= row_limit_dev[b] whatever the scores say — row b's caption has
 * at most row_limit_dev[b] words (the recorded log-prob of that step stays the arg-max's; every other row and every earlier
 * position is untouched, tests/test_hip_finished_rows.py).  NULL (default): off = the reference's loop (editnet_rl.py:503-547,
 * one global max_len).  While set, small batches take the per-step kernels (the persistent launch has no cap).  Also how
 * bench.py gives a random-weight model the finish times of real captions (secondary.realistic_lengths).
 * In every mode the GEMM / attention kernels of a timestep return at once after the reference's `break` (all rows finished). */
int set_decode_row_limits(const int* row_limit_dev);

/* The timestep loop of set_editnet_greedy alone (editnet_rl.py:503-547) on a workspace that already holds a completed
 * set_editnet_begin for the same (X, prev, prevlen): the per-sequence prologue (editnet_rl.py:499-501 and the hoisted
 * projections) of batch i+1 depends on nothing in the decode of batch i, so a caller that issues one decode after the
 * other can run set_editnet_begin for the NEXT batch on a second stream / workspace while this loop runs
 * (pipeline.DevicePrefetcher(begin_ahead=...) does).  The caller orders `stream` after the stream that ran the prologue.
 * Results are bit-identical to set_editnet_greedy. */
int set_editnet_greedy_begun(const SetEditNetWeights* w, const SetEditNetDims* d, const float* X, int64_t start_idx,
                             int64_t end_idx, int max_len, int64_t* seq, float* seq_logp, void* ws,
                             size_t ws_bytes, void* stream);

/* Beam search of ONE image, the reference's own evaluate() shape (editnet.py:601-713: batch = 1 image, beam k = 3), as
 * prologue + ONE persistent launch (csrc/decode_persistent_wide.hip, beam mode): the d->B <= 4 rows of the workspace are the
 * k hypotheses — X (k, R, F), prev (k, T), prevlen (k) hold the image's inputs k times — and every timestep ends with the
 * reference's pick (editnet.py:654-699: log-softmax, + running scores, flat top-k over k V, parent / word split, completed
 * hypotheses leave, k shrinks) instead of the arg-max; recurrent state follows the parent map inside the launch.  The
 * search stops when every hypothesis has ended or after max_picks picks (the reference's 50-step limit: max_picks = 51).
 * Outputs (device): hist_parent / hist_word (max_picks, 4) = parent slot and appended word of every slot after every pick —
 * the host follows them back to read a sequence; best_score / best_word [1] and result [4] = {pick index of the best
 * completed hypothesis (-1: none), its parent slot, hypotheses still alive, picks made}.  A time-out poisons best_score with
 * NaN and result[2..3] = -1 (SET_ERR_FAULT at the next call).  SET_ERR_UNSUPPORTED (take set_editnet_step +
 * set_beam_pick_f32): no token table, k > 4, adaptive features, dimensions the persistent launch does not cover, another
 * process owns the device's persistent launches — all answered BEFORE anything is touched; only a device whose LDS limit or
 * resident-workgroup capacity turns out too small is answered after the prologue has been written into `ws` (outputs
 * untouched; the answer is the same for every later call with these dims, so a caller remembers it).
 * Parity: tests/test_hip_beam.py against the reference's beam goldens. */
int set_editnet_beam_persistent(const SetEditNetWeights* w, const SetEditNetDims* d, const float* X,
                                const float* image_mean, const int64_t* prev, const int64_t* prevlen,
                                int64_t start_idx, int64_t end_idx, int max_picks, int32_t* hist_parent,
                                int64_t* hist_word, float* best_score, int64_t* best_word, int32_t* result,
                                void* ws, size_t ws_bytes, void* stream);

/* The same loop with multinomial sampling (editnet_rl.py:521-528, sample_rl=True, eval mode, no gradients):
 * it ~ Categorical(softmax(logits)) drawn on the device with Philox4x32-10 (counter = (row, timestep, offset),
 * key = seed): reproducible for a given (seed, offset), independent streams for different offsets.
 * seq_logp holds log_softmax(logits)[it].  No host synchronisation. */
int set_editnet_sample(const SetEditNetWeights* w, const SetEditNetDims* d, const float* X,
                       const float* image_mean, const int64_t* prev, const int64_t* prevlen,
                       int64_t start_idx, int64_t end_idx, int max_len, uint64_t seed, uint64_t offset,
                       int64_t* seq, float* seq_logp, void* ws, size_t ws_bytes, void* stream);

/* Teacher-forced XE forward (editnet.py:479-548, eval mode, use_ss=False) on a batch already
 * sorted by decreasing caption length.  caps (B,Lc) int64 sorted; host_decode_lengths[B] on the
 * HOST, non-increasing; predictions (B,maxT,V) is fully overwritten (zeros where not decoded). */
int set_editnet_xe_forward(const SetEditNetWeights* w, const SetEditNetDims* d, const float* X,
                           const float* image_mean, const int64_t* caps, int64_t caps_stride,
                           const int* host_decode_lengths, const int64_t* prev,
                           const int64_t* prevlen, float* predictions, void* ws, size_t ws_bytes,
                           void* stream);

/* Debug / module-API accessor: device pointer of a named workspace tensor (NULL if unknown).
 * names: "H" (B,T,D) "M" (B,T,D) "final_hidden" (B,D) "mask" (B,T) "att1" (B,R,A) "att1_c" (B,T,A)
 * "image_mean" (B,F) "h1" "c1" "h2" "c2" "emb" "ctx_cap" "attend_cap" "sel" (B,D) "attend_img" (B,F)
 * "alpha_c" (B,T) "alpha" (B,R) "logits" (B,V) "it" (B) int64 "unfinished" (B) int32
 * "cap_proj" (B,T,2D) = [context_gate.W[:,2D:3D] H | sc_affine.W H] and "mem_proj" (B,T,D) = gate_cmem.W M: the
 * per-sequence hoisted projections the step reads instead of contracting the attention context / selected row */
void* set_editnet_ws_tensor(const SetEditNetDims* d, void* ws, const char* name);

/* ------------------------------------------------------------------------------------------
 * DCNet (reference dcnet.py:273-350 `DAE`, dcnet_rl.py:256-346) — text only, no image features
 * ------------------------------------------------------------------------------------------ */
typedef struct SetDcnetDims {
    int B, T;
    int D;      /* decoder_dim (1024)                       */
    int A;      /* attention_dim (512)                      */
    int C;      /* caption_features_dim (512); encoder output is 2C */
    int E;      /* emb_dim (1024)                           */
    int V;
    int maxT;
} SetDcnetDims;

typedef struct SetDcnetWeights {
    const float* embed;                          /* embed.embedding.weight (V,E)                  */
    const float *enc_wih_f, *enc_whh_f, *enc_bih_f, *enc_bhh_f;   /* lstm_encoder.*_l0          */
    const float *enc_wih_b, *enc_whh_b, *enc_bih_b, *enc_bhh_b;   /* lstm_encoder.*_l0_reverse  */
    const float *enc_cat_w, *enc_cat_b;          /* caption_encoder.concat (2C,2C)                */
    const float *ca_feat_w, *ca_feat_b;          /* caption_attention.cap_features_att (A,2C)     */
    const float *ca_dec_w, *ca_dec_b;            /* caption_attention.cap_decoder_att (A,D)       */
    const float *ca_full_w, *ca_full_b;          /* caption_attention.cap_full_att (1,A)          */
    const float *al_wih, *al_whh, *al_bih, *al_bhh;   /* attention_lstm (4D,3E),(4D,D)            */
    const float *ll_wih, *ll_whh, *ll_bih, *ll_bhh;   /* language_lstm (4D,2E),(4D,D)             */
    const float *fc_w, *fc_b;                    /* fc (V,D)                                      */
    /* Optional inference-time token table (V, 4D + 8C) or NULL, built by set_dcnet_build_token_table:
     *   [0, 4D)        attention_lstm.weight_ih[:, :E] relu(E[v])            (dcnet.py:336-337)
     *   [4D, 4D+4C)    lstm_encoder.weight_ih_l0 relu(E[v]) + bias_ih_l0      (dcnet.py:233, forward direction)
     *   [4D+4C, 4D+8C) lstm_encoder.weight_ih_l0_reverse relu(E[v]) + bias_ih_l0_reverse
     * Valid while embed / attention_lstm.weight_ih / lstm_encoder.weight_ih* / bias_ih* are unchanged. */
    const float* tok_table;
} SetDcnetWeights;

size_t set_dcnet_workspace_bytes(const SetDcnetDims* d);
size_t set_dcnet_token_table_bytes(const SetDcnetDims* d);
size_t set_dcnet_token_table_workspace_bytes(const SetDcnetDims* d);
int set_dcnet_build_token_table(const SetDcnetWeights* w, const SetDcnetDims* d, float* table, void* ws,
                                size_t ws_bytes, void* stream);
/* caption_encoder (dcnet.py:220-243) + hoisted cap_features_att(enc) (dcnet.py:261) + zero state */
int set_dcnet_begin(const SetDcnetWeights* w, const SetDcnetDims* d, const int64_t* prev,
                    const int64_t* prevlen, void* ws, size_t ws_bytes, void* stream);
/* one timestep (dcnet.py:336-347 / dcnet_rl.py:306-312), eval mode */
int set_dcnet_step(const SetDcnetWeights* w, const SetDcnetDims* d, const int64_t* tokens,
                   int64_t tokens_stride, int bt, float* logits, int64_t ld_logits, void* ws,
                   size_t ws_bytes, void* stream);
/* greedy epilogue of one free-running step (dcnet_rl.py:313-340); see set_editnet_greedy_pick */
int set_dcnet_greedy_pick(const SetDcnetWeights* w, const SetDcnetDims* d, const float* logits,
                          int64_t ld_logits, int t, int64_t end_idx, int64_t* seq, float* seq_logp,
                          int max_len, void* ws, size_t ws_bytes, void* stream);
int set_dcnet_greedy(const SetDcnetWeights* w, const SetDcnetDims* d, const int64_t* prev,
                     const int64_t* prevlen, int64_t start_idx, int64_t end_idx, int max_len,
                     int64_t* seq, float* seq_logp, void* ws, size_t ws_bytes, void* stream);
/* multinomial twin of set_dcnet_greedy (dcnet_rl.py:320-327); see set_editnet_sample */
int set_dcnet_sample(const SetDcnetWeights* w, const SetDcnetDims* d, const int64_t* prev,
                     const int64_t* prevlen, int64_t start_idx, int64_t end_idx, int max_len, uint64_t seed,
                     uint64_t offset, int64_t* seq, float* seq_logp, void* ws, size_t ws_bytes, void* stream);
int set_dcnet_xe_forward(const SetDcnetWeights* w, const SetDcnetDims* d, const int64_t* caps,
                         int64_t caps_stride, const int* host_decode_lengths, const int64_t* prev,
                         const int64_t* prevlen, float* predictions, void* ws, size_t ws_bytes,
                         void* stream);
/* names: "enc" (B,T,2C) "final_hidden" (B,2C) "mask" (B,T) "att1_c" (B,T,A) "h1" "c1" "h2" "c2"
 * "emb" "attend_cap" (B,2C) "alpha_c" (B,T) "logits" (B,V) "it" "unfinished" */
void* set_dcnet_ws_tensor(const SetDcnetDims* d, void* ws, const char* name);

/* ------------------------------------------------------------------------------------------
 * Operator-level entry points (the sub-module `forward`s the reference's beam search calls
 * directly, editnet.py:645-653 / eval_full.py:133-149)
 * ------------------------------------------------------------------------------------------ */
/* y = act(x W^T + bias): nn.Linear (+ ReLU/tanh/sigmoid).  x (M,K) ldx, w (N,K) ldw, y (M,N) ldy */
size_t set_linear_workspace_bytes(int M, int N, int K);
int set_linear_f32(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias,
                   float* y, int64_t ldy, int M, int N, int K, int act, void* ws, size_t ws_bytes,
                   void* stream);
/* EmbeddingC.forward (editnet.py:300-304), eval: out[i] = relu(table[ids[i]]) */
int set_embed_relu_f32(const float* table, const int64_t* ids, int64_t ids_stride, float* out,
                       int64_t ldo, int n, int D, int V, void* stream);
/* nn.LSTMCell / LSTMCellC (editnet.py:226-244): gates = x Wih^T + bih + h Whh^T + bhh, order i,f,g,o */
size_t set_lstm_cell_workspace_bytes(int M, int D, int Kx);
int set_lstm_cell_f32(const float* x, int64_t ldx, int Kx, const float* h, const float* c,
                      const float* w_ih, int64_t ld_wih, const float* w_hh, const float* b_ih,
                      const float* b_hh, float* h_out, float* c_out, int M, int D, void* ws,
                      size_t ws_bytes, void* stream);
/* CaptionAttentionC.forward (editnet.py:364-381); att1_c may be NULL (computed into ws).
 * H (M,T,Dh) caption features, h1 (M,D) decoder state.  Outputs gated (M,Dh), alpha_c (M,T).
 * Also serves DCNet's CaptionAttention (dcnet.py:254-270) when w->ca_gate_w is NULL: `gated`
 * then receives the plain context (M,Dh) (Dh = 2C there; EditNet requires Dh == D). */
size_t set_caption_attention_workspace_bytes(int M, int T, int Dh, int A);
int set_caption_attention_f32(const SetEditNetWeights* w, const float* H, const float* att1_c,
                              const float* h1, const float* word, const float* mask, float* gated,
                              float* alpha_c, int M, int T, int Dh, int D, int A, void* ws,
                              size_t ws_bytes, void* stream);
/* the un-gated form (DCNet, w->ca_gate_w == NULL) that also emits the decoder-side projection incl. bias it scored with,
 * att2_out (M,A) — the `att2` operand of set_attention_bwd_f32 — for the training node */
int set_caption_attention_att2_f32(const SetEditNetWeights* w, const float* H, const float* att1_c, const float* h1,
                                   const float* mask, float* ctx, float* alpha_c, float* att2_out, int M, int T,
                                   int Dh, int D, int A, void* ws, size_t ws_bytes, void* stream);
/* VisualAttentionC.forward (editnet.py:439-447); att1 may be NULL (att_embed + features_att are
 * then recomputed exactly as the reference does every call).  adaptive != 0 selects the masked
 * variant (editnet_adaptive.py:438-457).  ctx (M,F). */
size_t set_visual_attention_workspace_bytes(int M, int R, int F, int D, int A);
int set_visual_attention_f32(const SetEditNetWeights* w, const float* X, const float* att1,
                             const float* h1, float* ctx, float* alpha, int M, int R, int F, int D,
                             int A, int adaptive, void* ws, size_t ws_bytes, void* stream);
/* Same with att1 (M,R,A) required and the region mask rmask (M,R; 1 = valid, NULL = no masking) supplied
 * by the caller instead of re-derived (grad-enabled adaptive path); ws >= the size above. */
int set_visual_attention_masked_f32(const SetEditNetWeights* w, const float* X, const float* att1,
                                    const float* rmask, const float* h1, float* ctx, float* alpha, int M,
                                    int R, int F, int D, int A, void* ws, size_t ws_bytes, void* stream);
/* SelectC.forward hard mode (editnet.py:403-421): sel = M[b,j*] * (a + (1-a)), j* = argmax alpha_c */
int set_select_f32(const float* Mem, const float* alpha_c, float* sel, int M, int T, int D,
                   void* stream);
/* SelectC.forward with soft = True (editnet.py:419-420): sel (M,D) = sum_t alpha_c[b,t] * Mem[b,t,:] (terms added in
 * index order), and its backward: dM[b,t,:] = alpha[b,t] * dsel[b,:], dalpha[b,t] = <dsel[b,:], Mem[b,t,:]>. */
int set_select_soft_f32(const float* Mem, const float* alpha_c, float* sel, int M, int T, int D,
                        void* stream);
int set_select_soft_bwd_f32(const float* dsel, const float* Mem, const float* alpha, float* dM, float* dalpha,
                            int M, int T, int D, void* stream);
/* CopyLSTMCellC.forward (editnet.py:265-285); x (M,2D+F) */
size_t set_copy_lstm_workspace_bytes(int M, int D, int Kx);
int set_copy_lstm_f32(const SetEditNetWeights* w, const float* x, int64_t ldx, int Kx,
                      const float* h2, const float* c2, const float* c_memory, float* h_out,
                      float* c_out, int M, int D, void* ws, size_t ws_bytes, void* stream);
/* CaptionEncoderC.forward (editnet.py:319-348): outputs padded to T (caller slices to max len).
 * H, Mem (B,T,D) ; final_hidden (B,D) ; mask (B,T) */
size_t set_caption_encoder_workspace_bytes(int B, int T, int D);
int set_caption_encoder_f32(const SetEditNetWeights* w, const int64_t* seq, const int64_t* seq_len,
                            float* H, float* Mem, float* final_hidden, float* mask, int B, int T,
                            int D, int V, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training support (the reference's backward is PyTorch autograd over a1-a12,
 * editnet.py:579).  `*_train_f32` = the operator forward that additionally stores what its
 * backward needs (POST-activation gates i,f,g,o; copy gate; context-gate factors); `*_bwd_f32` =
 * the hand-written pointwise / attention part of the operator's backward.  The plain dX = dY W and
 * dW = dY^T X contractions between them are library GEMMs on the PyTorch side (autograd_ops.py).
 * ------------------------------------------------------------------------------------------ */
int set_lstm_cell_train_f32(const float* x, int64_t ldx, int Kx, const float* h, const float* c,
                            const float* w_ih, int64_t ld_wih, const float* w_hh, const float* b_ih,
                            const float* b_hh, float* h_out, float* c_out, float* gates_out, int M, int D,
                            void* ws, size_t ws_bytes, void* stream);
/* dgates (M,4D) = pre-activation gate gradients, dc_prev (M,D); dh / dc may be NULL (zero) */
/* Token cross-entropy of the training loops (editnet.py:571-577, dcnet.py:391-397: pack_padded_sequence of scores and
 * targets, CrossEntropyLoss) on the (B, T, V) scores in place: element (b, t, v) at scores[b*stride_b + t*stride_t + v],
 * target word of (b, t) at targets[b*tstride_b + t*tstride_t].  The batch is sorted by decreasing decode length and
 * live[t] (HOST array, T <= 64 entries, non-increasing) = number of sequences with decode length > t: row (b, t) is one
 * of the packed rows iff b < live[t].  Forward: rowloss / lse (T*B floats each, index t*B + b; zeros for dead rows) and
 * loss_sum (1 float) = the SUM of the token losses (divide by the token count for CrossEntropyLoss's mean).
 * Backward: grad (T, B, ld_grad) with ld_grad = V rounded up to 4 — (softmax - onehot) * dloss[0] for live rows, zeros for
 * dead rows and for the padding columns (so the buffer can feed set_gemm_f32 as a zero-padded k-major operand). */
int set_xe_loss_f32(const float* scores, int64_t stride_b, int64_t stride_t, const int64_t* targets, int64_t tstride_b,
                    int64_t tstride_t, const int* live, int B, int T, int V, float* rowloss, float* lse,
                    float* loss_sum, void* stream);
int set_xe_loss_bwd_f32(const float* scores, int64_t stride_b, int64_t stride_t, const int64_t* targets,
                        int64_t tstride_b, int64_t tstride_t, const int* live, int B, int T, int V, const float* lse,
                        const float* dloss, float* grad, int64_t ld_grad, void* stream);

/* The tail of the training step — torch.nn.utils.clip_grad_norm_(params, max_norm) followed by torch.optim.Adam.step()
 * (editnet.py:580-581, dcnet.py:399-400, editnet_rl.py:684-686) — over n fp32 tensors in two launches per 40 tensors:
 * per-chunk sums of squares, then the Adam update with the clipping coefficient min(1, max_norm / (norm + 1e-6)) folded
 * into the gradient read (deterministic norm, no host round trip).  All arrays are HOST arrays of length n; the pointers
 * in params / grads / exp_avg / exp_avg_sq are 16-byte aligned DEVICE pointers to numel[i] contiguous floats.  step[i] is
 * the 1-based step count of tensor i (bias corrections and 1 - beta are formed on the host in double, as torch forms
 * them, then rounded to fp32), lr / beta1 / beta2 / eps / weight_decay the hyper-parameters of its param group (weight_decay in torch.optim.Adam's L2 form).  max_norm <= 0: no
 * clipping.  scale_grads != 0: the clipped gradient is also written back (clip_grad_norm_'s in-place effect; costs one
 * more gradient-sized write).  total_norm_out (device, 1 float, may be NULL) receives the gradient norm.
 * ws: set_clip_adam_workspace_bytes(n, numel) bytes. */
size_t set_clip_adam_workspace_bytes(int n, const int64_t* numel);
int set_clip_adam_f32(int n, float* const* params, const float* const* grads, float* const* exp_avg,
                      float* const* exp_avg_sq, const int64_t* numel, const int64_t* step, const double* lr,
                      const double* beta1, const double* beta2, const double* eps, const double* weight_decay,
                      float max_norm, int scale_grads, float* total_norm_out, void* ws, size_t ws_bytes, void* stream);

/* nn.LSTMCell forward (training form, gates saved) for an input that is a concatenation with loop-invariant column
 * blocks: gates = x0 w0^T (+ x1 w1^T) + h w_hh^T + pre, where w0 / w1 are column blocks of weight_ih (row strides
 * ld_w0 / ld_w1) and pre (M,4D) holds the invariant blocks' products plus both biases, contracted once per sequence by
 * the caller (editnet.py:523: final_hidden and image_mean; dcnet.py:336: final_hidden).  x1 may be NULL. */
int set_lstm_cell_pre_train_f32(const float* x0, int64_t ld_x0, const float* w0, int64_t ld_w0, int K0,
                                const float* x1, int64_t ld_x1, const float* w1, int64_t ld_w1, int K1,
                                const float* h, const float* w_hh, const float* pre, int64_t ld_pre, const float* c,
                                float* h_out, float* c_out, float* gates_out, int M, int D, void* ws,
                                size_t ws_bytes, void* stream);
int set_lstm_cell_bwd_f32(const float* dh, const float* dc, const float* gates, const float* c_prev,
                          const float* c_new, float* dgates, float* dc_prev, int M, int D, void* stream);
int set_copy_lstm_train_f32(const SetEditNetWeights* w, const float* x, int64_t ldx, int Kx,
                            const float* h2, const float* c2, const float* c_memory, float* h_out,
                            float* c_out, float* gates, float* c_new, float* cg, int M, int D, void* ws,
                            size_t ws_bytes, void* stream);
/* stage 1 of CopyLSTMCellC backward: du (copy-gate pre-activation grad), direct grads of c_memory and
 * c_new, o-gate pre-activation grad; stage 2 (after dcn += du W_n) is set_lstm_gates_bwd_f32 */
int set_copy_gate_bwd_f32(const float* dh, const float* dadp, const float* ogate, const float* adp,
                          const float* cg, const float* cmem, const float* c_new, float* du,
                          float* dcm_direct, float* dcn_direct, float* do_pre, int M, int D, void* stream);
/* the same with the o-gate read in place from the saved (M,4D) gate block: ogate = gates + 3D, ld_ogate = 4D */
int set_copy_gate_bwd_ld_f32(const float* dh, const float* dadp, const float* ogate, int64_t ld_ogate,
                             const float* adp, const float* cg, const float* cmem, const float* c_new, float* du,
                             float* dcm_direct, float* dcn_direct, float* do_pre, int M, int D, void* stream);
int set_lstm_gates_bwd_f32(const float* dcn, const float* do_pre, const float* gates, const float* c_prev,
                           float* dgates, float* dc_prev, int M, int D, void* stream);
int set_caption_attention_train_f32(const SetEditNetWeights* w, const float* H, const float* att1_c,
                                    const float* h1, const float* word, const float* mask, float* gated,
                                    float* alpha_c, float* ctx, float* zt, float* s, float* t, int M, int T,
                                    int Dh, int D, int A, void* ws, size_t ws_bytes, void* stream);
int set_context_gate_bwd_f32(const float* dout, const float* zt, const float* s, const float* t, float* dz,
                             float* ds, float* dt, int M, int D, void* stream);
/* the same with a common row stride ld_out >= D for dz / ds / dt: the three can then be column blocks of one (M, 3D) buffer
 * whose products with stacked weight blocks are single contractions */
int set_context_gate_bwd_ld_f32(const float* dout, const float* zt, const float* s, const float* t, float* dz,
                                float* ds, float* dt, int64_t ld_out, int M, int D, void* stream);
/* additive-attention backward (use_tanh=1: caption attention, 0: visual attention with ReLU).
 * values (M,L,Dv) are the attended rows (H or X); att2 (M,A) includes the decoder-projection bias.
 * Outputs datt1 (M,L,A), datt2 (M,A), dwfull_part (M,A; sum over M = d full_att.weight),
 * dvalues (M,L,Dv) or NULL, de (M,L) or NULL (sum = d full_att.bias). */
int set_attention_bwd_f32(const float* dctx, const float* dalpha_ext, const float* alpha,
                          const float* values, const float* att1, const float* att2, const float* w_full,
                          float* datt1, float* datt2, float* dwfull_part, float* dvalues, float* de, int M,
                          int L, int Dv, int A, int use_tanh, void* stream);
/* Caption-encoder recurrence of the grad-enabled path (CaptionEncoderC editnet.py:333-338; the packed BiLSTM of
 * dcnet.py:233), one step for all rows with the length handling inside the kernels: rows with t < lens[b] advance, the
 * others carry (h, c) and emit zeros.  xg = hoisted input projection x W_x^T + b_x, element (b, t, :) at
 * xg + b ld_xg_row + t ld_xg_t.  train: h_out / c_out (B,D) new state, H / Mem (Mem may be NULL) row (b, t) at
 * b ld_out_b + t ld_out_t + out_col0, Hprev (same layout, may be NULL) receives h (the state the step started from, the
 * operand of dW_hh), gates (B,4D) post-activations.  bwd: dH / dM (same layout as H / Mem, may be NULL) + dh / dc
 * (B,D, may be NULL) -> dgates rows (b, :) at dgates + b ld_dg (pre-activation gradients = gradient of xg[b,t,:]),
 * dc_prev (B,D) and dh_pass (B,D): the caller adds dgates W_hh to dh_pass to obtain dh of the previous step. */
size_t set_encoder_cell_workspace_bytes(int B, int D);
int set_encoder_cell_train_f32(const float* xg, int64_t ld_xg_row, int64_t ld_xg_t, const float* h, const float* c,
                               const float* w_hh, const float* b_hh, const int64_t* lens, int t, float* h_out, float* c_out,
                               float* H, float* Mem, float* Hprev, int64_t ld_out_b, int64_t ld_out_t, int out_col0,
                               float* gates, int B, int D, void* ws, size_t ws_bytes, void* stream);
int set_encoder_cell_bwd_f32(const float* dh, const float* dc, const float* dH, const float* dM, int64_t ld_d_b,
                             int64_t ld_d_t, int d_col0, const int64_t* lens, int t, const float* gates,
                             const float* c_prev, const float* c_new, float* dgates, int64_t ld_dg, float* dc_prev,
                             float* dh_pass, int B, int D, void* stream);
int set_select_bwd_f32(const float* dsel, const float* Mem, const float* alpha, float* dM, float* dalpha,
                       int M, int T, int D, void* stream);
/* The attention block of one training timestep (editnet.py:534-540) in four launches: the decoder-side projections of
 * both attentions and the [word,h1] parts of tc_affine / context_gate as one grouped GEMM, both attention roles +
 * SelectC as one kernel, the context side of the gate, its pointwise.  att1_c (M,T,A) = cap_features_att(H),
 * att1 (M,R,A) = features_att(att_embed(X)) (this step's, with its dropout), rmask (M,R) or NULL.  Outputs: gated (M,D)
 * = attend_cap, alpha_c (M,T), ctx (M,D) = sum_t alpha_t H_t, zt / s / t (M,D) the gate's factors (for the backward),
 * sel (M,D), attend_img (M,F), alpha_v (M,R); att2_c_out / att2_v_out (M,A) or NULL: the decoder-side projections incl.
 * bias exactly as scored (what set_attention_bwd_f32 takes as `att2`: the backward need not recompute them). */
size_t set_editnet_attentions_workspace_bytes(int M, int D, int A);
int set_editnet_attentions_train_f32(const SetEditNetWeights* w, const float* H, const float* att1_c,
                                     const float* mask, const float* Mem, const float* X, const float* att1,
                                     const float* rmask, const float* h1, const float* word, float* gated,
                                     float* alpha_c, float* ctx, float* zt, float* s, float* t, float* sel,
                                     float* attend_img, float* alpha_v, float* att2_c_out, float* att2_v_out, int M,
                                     int T, int R, int F, int D, int A, void* ws, size_t ws_bytes, void* stream);
/* Accumulating forms used by the whole-sequence training node (xe_sequence.py): gradients of loop-invariant operands
 * (H, Mem, cap_features_att(H), and features_att(att_embed(X)) in eval mode) are summed over the timesteps in place
 * instead of by one tensor-sized add per timestep.  acc_* = 1: `out += contribution`, rows beyond M are not touched.
 * ld_datt2 (>= A, 0 = A): row stride of datt2, so that the two attentions of a step can write the halves of one
 * (M, 2A) buffer whose product with the stacked decoder projections is ONE contraction.
 * select: with acc_dM only the selected row of dM changes (dalpha is always overwritten). */
int set_attention_bwd_acc_f32(const float* dctx, const float* dalpha_ext, const float* alpha,
                              const float* values, const float* att1, const float* att2, const float* w_full,
                              float* datt1, float* datt2, float* dwfull_part, float* dvalues, float* de, int M,
                              int L, int Dv, int A, int use_tanh, int acc_datt1, int acc_dvalues, int64_t ld_datt2,
                              void* stream);
/* dvalues[b, l, :] (+)= sum_t alpha[t, b, l] dctx[t, b, :] over per-sequence logs alpha (T,B,L), dctx (T,B,Dv), T <= 64:
 * the attended rows' gradient of all timesteps in one pass. */
int set_attention_dvalues_f32(const float* alpha, const float* dctx, float* dvalues, int T, int B, int L, int Dv,
                              int accumulate, void* stream);
int set_select_bwd_acc_f32(const float* dsel, const float* Mem, const float* alpha, float* dM, float* dalpha,
                           int M, int T, int D, int acc_dM, void* stream);
/* nn.Dropout(p) in training mode (editnet.py:299-302, :441, :546) with the library's Philox4x32-10 generator:
 * y[r, c] = x[r, c] / (1 - p) with probability 1 - p, else 0; one uniform per element from counter
 * (r, c / 4, offset) and key seed, so a (seed, offset) pair reproduces the mask.  x == y (in place) is allowed.
 * backward: dx (+)= dy * (y != 0 ? scale : 0) with y the forward OUTPUT (scale = 1 / (1 - p)); on the two sites that
 * follow a ReLU this also applies the ReLU's derivative (scale = 1 in eval mode = plain ReLU backward).
 * cols, all leading dimensions: multiples of 4; pointers 16-byte aligned. */
int set_dropout_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int rows, int cols, float p, uint64_t seed,
                    uint64_t offset, void* stream);
/* set_dropout_f32 of ONE operand for `steps` timesteps in one launch: y + t * y_step = dropout(x) drawn at offset + t
 * (t = 0 .. steps - 1), bit for bit what `steps` calls write.  The per-timestep region dropout of the train-mode forward
 * (editnet.py:441: `att_embed`'s nn.Dropout sees the same relu(W X) with a fresh mask every timestep) is the caller. */
int set_dropout_steps_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t y_step, int rows, int cols, int steps,
                          float p, uint64_t seed, uint64_t offset, void* stream);
/* its backward over all timesteps: dx (+)= sum_t dy[t] * (y[t] != 0 ? scale : 0), t ascending — what `steps` accumulating
 * calls of set_dropout_bwd_f32 leave in dx, bit for bit */
int set_dropout_bwd_steps_f32(const float* dy, int64_t lddy, int64_t dy_step, const float* y, int64_t ldy, int64_t y_step,
                              float* dx, int64_t ldx, int rows, int cols, int steps, float scale, int accumulate,
                              void* stream);
/* EmbeddingC.forward in train mode (editnet.py:299-302) in one launch: set_embed_relu_f32 followed by set_dropout_f32
 * in place (the same counters, hence the same mask for a (seed, offset) pair) */
int set_embed_relu_dropout_f32(const float* table, const int64_t* ids, int64_t ids_stride, float* out, int64_t ldo,
                               int n, int D, int V, float p, uint64_t seed, uint64_t offset, void* stream);
int set_dropout_bwd_f32(const float* dy, int64_t lddy, const float* y, int64_t ldy, float* dx, int64_t ldx, int rows,
                        int cols, float scale, int accumulate, void* stream);
/* backward of set_dropout_f32 with the keep mask regenerated from (seed, offset): dx (+)= dy * keep / (1 - p).  For the
 * site that does not follow a ReLU (h2 before fc, editnet.py:545), where an exactly-zero kept input must still pass its
 * gradient — nn.Dropout's backward multiplies by the mask, not by the output's zero pattern. */
int set_dropout_bwd_philox_f32(const float* dy, int64_t lddy, float* dx, int64_t ldx, int rows, int cols, float p,
                               uint64_t seed, uint64_t offset, int accumulate, void* stream);
/* out[c] (+)= sum_r x[r, c] (bias gradients over all (t, b) rows): two deterministic passes through a workspace of
 * set_colsum_workspace_bytes(cols); cols, ld multiples of 4. */
size_t set_colsum_workspace_bytes(int cols);
int set_colsum_f32(const float* x, int64_t ld, int rows, int cols, float* out, int accumulate, void* ws,
                   size_t ws_bytes, void* stream);
/* Several column sums as ONE pair of launches (the ~16 bias gradients of a training step; reference: autograd's per-bias
 * sum over (t, b) of every nn.Linear / nn.LSTMCell bias, editnet.py:223-560): out (+)= colsum(x), and out2 (+)= the same sums
 * when non-NULL (two biases fed by the same gradient: LSTMCell's bias_ih / bias_hh).  Deterministic (no atomics);
 * n <= SET_COLSUM_MAX; cols, ld multiples of 4; ws of set_colsum_group_workspace_bytes(d, n). */
#define SET_COLSUM_MAX 24
typedef struct SetColsumDesc {
    const float* x; int64_t ld; int rows; int cols;
    float* out; int accumulate;
    float* out2; int accumulate2;
} SetColsumDesc;
size_t set_colsum_group_workspace_bytes(const SetColsumDesc* d, int n);
int set_colsum_group_f32(const SetColsumDesc* d, int n, void* ws, size_t ws_bytes, void* stream);
/* mask[r] = (sum_c x[r, c] != 0) as 0/1 floats: the data-derived region mask of the adaptive model
 * (adaptive_features/editnet_adaptive.py:449-453) on the per-step, dropped-out region embedding. */
int set_rowsum_mask_f32(const float* x, int64_t ld, int rows, int cols, float* mask, void* stream);
/* dst[:, :] (+)= [src0 | src1 | ...] column blocks (1..4 segments, cols[i] floats wide, row stride ld[i]): builds the
 * concatenated LSTM input rows of editnet.py:523 / :541 inside the per-sequence operand logs (no torch.cat). */
int set_pack_f32(float* dst, int64_t ldd, int rows, int nseg, const float* const* src, const int64_t* ld,
                 const int* cols, int accumulate, void* stream);

/* Multinomial sampling epilogue of one free-running step as an operator (the grad-enabled SCST rollout,
 * editnet_rl.py:521-543 / dcnet_rl.py:320-340): replaces exp -> torch.multinomial -> gather -> <end> rewrite ->
 * `unfinished` latch -> seq store -> `unfinished.sum() == 0` host check.  logits (B,V) include the bias.
 * State owned by the caller across the steps of one rollout: it (B) int64, unfinished (B) int32,
 * alive (max_len+2) int32 (all initialised here at t == 0); call with t = 0,1,2,... in order.
 * Outputs of the step: seq[:,t] (seq (B,max_len) pre-zeroed), it = next input token, raw_ids (B) = the sampled
 * word before the <end> rewrite or -1 once every row had finished at an earlier step (the reference's `break`),
 * lse (B) = logsumexp(logits), step_logp (B) = log_softmax(logits)[raw_id] (0 after the break).
 * RNG: Philox4x32-10, counter (row, t, offset), key seed. */
int set_sample_pick_f32(const float* logits, int64_t ld_logits, int B, int V, int t, int max_len, int64_t end_idx,
                        uint64_t seed, uint64_t offset, int64_t* seq, int64_t* it, int32_t* unfinished,
                        int32_t* alive, int64_t* raw_ids, float* lse, float* step_logp, void* stream);
/* backward of step_logp w.r.t. logits: dlogits[b,v] = g[b] (1[v == raw_id_b] - exp(logits[b,v] - lse[b])); rows
 * with raw_id < 0 get zeros */
int set_sample_logp_bwd_f32(const float* logits, int64_t ld_logits, const float* lse, const int64_t* raw_ids,
                            const float* g, float* dlogits, int64_t ld_dlogits, int B, int V, void* stream);
/* the device RNG itself (tests: known-answer vectors): out (n,4) uint32 = Philox4x32-10(counter (i,0,offset), key seed) */
int set_philox4x32(uint32_t* out, int n, uint64_t seed, uint64_t offset, void* stream);

/* Beam-search step epilogue for NI images x k hypotheses (rows i*k+j), replacing the host bookkeeping of
 * editnet.py:654-699 / dcnet.py:450-500 / eval_full.py:150-200: log_softmax (or, with logits2, the ensemble
 * log((softmax(logits)+softmax(logits2))/2)), + running scores, flat top-k over k*V per image (ties: lowest
 * flat index), parent/word split, completed-hypothesis tracking (best completed score, first maximum),
 * k_left -= #<end>, live hypotheses compacted to the front, sequences re-indexed and extended
 * (seqs_in -> seqs_out, cur_len valid tokens -> cur_len+1), next input `words`, and `rows` = the parent
 * row of every hypothesis row for set_beam_gather_f32.  k <= 8.  Step 1 is expressed by the caller through
 * scores = {0, -inf, ...}.  Images with k_left == 0 are left untouched. */
int set_beam_pick_f32(const float* logits, const float* logits2, int64_t ld, int NI, int k, int V, int64_t end_idx,
                      int cur_len, int Lmax, float* scores, int32_t* k_left, const int64_t* seqs_in,
                      int64_t* seqs_out, float* best_score, int64_t* best_seq, int32_t* best_len, int64_t* words,
                      int32_t* rows, void* stream);
/* In-place re-index of up to four (NI*k, D) recurrent-state tensors by `rows` (editnet.py:687-696). */
int set_beam_gather_f32(float* s0, float* s1, float* s2, float* s3, const int32_t* rows, int NI, int k, int D,
                        void* stream);

/* General-layout fp32 GEMM on the MFMA pipe, used for the Linear backward (replaces the cuBLAS calls
 * autograd makes for nn.Linear / nn.LSTMCell: dX = dY.W and dW += dY^T.X):
 *     C[m,n] (+)= sum_k a(m,k) * b(n,k)
 *     a(m,k) = a_kminor ? A[k*lda + m] : A[m*lda + k];   b(n,k) = b_kminor ? B[k*ldb + n] : B[n*ldb + k]
 * accumulate != 0 adds into C (in-place .grad accumulation).  `ws` is scratch for split-K slabs (may be
 * NULL: the contraction is then never split).  Leading dimensions are multiples of 4 floats; a k-minor
 * operand needs its own dimension (M or N) to be a multiple of 4 — for A, alternatively rows that are readable up
 * to the next multiple of 4 (lda >= round_up(M,4); output rows >= M are never stored) —; a k-major one needs
 * K % 4 == 0, or rows that are zero-padded by the caller up to the next multiple of 4 (leading dimension >=
 * round_up(K,4)). */
int set_gemm_f32(const float* A, long long lda, int a_kminor, const float* B, long long ldb, int b_kminor,
                 float* C, long long ldc, int M, int N, int K, int accumulate, void* ws, size_t ws_bytes,
                 void* stream);
/* Up to 6 independent problems of the same operand layout in one launch (the dX products of one module's
 * backward share dY and are individually too small to fill the chip).  Outputs must not alias each other. */
typedef struct SetGemmDesc {
    const float* A; long long lda;
    const float* B; long long ldb;
    float* C; long long ldc;
    int M, N, K, accumulate;
} SetGemmDesc;
int set_gemm_group_f32(const SetGemmDesc* descs, int n, int a_kminor, int b_kminor, void* ws, size_t ws_bytes,
                       void* stream);

/* ------------------------------------------------------------------------------------------
 * Split-K partials handed to their consumer instead of being reduced by a launch of their own (round 5; the XE / SCST
 * backward of xe_sequence.py: 130 reduction launches of ~5 us per training step disappear).  A SetSlabSrc describes one
 * addend of a (rows, N) gradient: value(m, j) = sum_{s < nslab} p[s * slab_stride + m * ld + j] for m < rows, 0 for the rows
 * beyond (a sequence that had left the batch when the product ran).  Sums run in slab order, then in list order:
 * deterministic.
 * set_gemm_group_slabs_f32 = set_gemm_group_f32 without its reduction: problem i's partials stay in `ws` and out[i]
 * describes them (nslab >= 2); a problem the plan does not split is written to descs[i].C as usual (accumulate honoured)
 * and out[i].nslab = 0.  The caller keeps `ws` untouched until every consumer has run.
 * The *_src entry points are the backward kernels of the decode step with such lists as (part of) their gradient input:
 * gradient = base (may be NULL) + the listed addends.
 * ------------------------------------------------------------------------------------------ */
typedef struct SetSlabSrc {
    const float* p;
    int64_t slab_stride;     /* floats between two partials */
    int64_t ld;              /* floats between two rows */
    int32_t nslab;           /* 0: this entry contributes nothing */
    int32_t rows;            /* rows the partials hold */
} SetSlabSrc;
#define SET_MAX_SRC 4
int set_gemm_group_slabs_f32(const SetGemmDesc* descs, int n, int a_kminor, int b_kminor, void* ws, size_t ws_bytes,
                             SetSlabSrc* out, void* stream);
/* set_lstm_cell_bwd_f32 with dh = the listed addends (n_dh <= SET_MAX_SRC; 0: dh = 0) */
int set_lstm_cell_bwd_src_f32(const SetSlabSrc* dh_src, int n_dh, const float* dc, const float* gates, const float* c_prev,
                              const float* c_new, float* dgates, float* dc_prev, int M, int D, void* stream);
/* set_copy_gate_bwd_ld_f32 with dh = the listed addends + the backward of the output dropout (editnet.py:545) of dh_drop
 * (M, D; may be NULL): dh_drop * keep / (1 - p) with the keep mask regenerated from (seed, offset) as
 * set_dropout_bwd_philox_f32 does; p = 0: dh_drop is added as it is (eval mode). */
int set_copy_gate_bwd_src_f32(const SetSlabSrc* dh_src, int n_dh, const float* dh_drop, int64_t ld_drop, float p, uint64_t seed,
                              uint64_t offset, const float* dadp, const float* ogate, int64_t ld_ogate, const float* adp,
                              const float* cg, const float* cmem, const float* c_new, float* du, float* dcm_direct,
                              float* dcn_direct, float* do_pre, int M, int D, void* stream);
/* set_lstm_gates_bwd_f32 / set_select_bwd_acc_f32 / set_context_gate_bwd_ld_f32 / set_attention_bwd_acc_f32 with their
 * gradient input = base (NULL allowed where noted) + the listed addends.  set_attention_bwd_src_f32 also stores the gradient
 * it summed to dctx_out (M, Dv) when that is not NULL (the caption context's gradient is needed again after the loop). */
int set_lstm_gates_bwd_src_f32(const float* dcn_base, const SetSlabSrc* src, int n_src, const float* do_pre, const float* gates,
                               const float* c_prev, float* dgates, float* dc_prev, int M, int D, void* stream);
int set_select_bwd_src_f32(const float* dsel_base, const SetSlabSrc* src, int n_src, const float* Mem, const float* alpha,
                           float* dM, float* dalpha, int M, int T, int D, int acc_dM, void* stream);
int set_context_gate_bwd_src_f32(const float* dout_base /* may be NULL */, const SetSlabSrc* src, int n_src, const float* zt,
                                 const float* s, const float* t, float* dz, float* ds, float* dt, int64_t ld_out, int M, int D,
                                 void* stream);
int set_attention_bwd_src_f32(const float* dctx_base /* may be NULL */, const SetSlabSrc* src, int n_src, float* dctx_out,
                              const float* dalpha_ext, const float* alpha, const float* values, const float* att1,
                              const float* att2, const float* w_full, float* datt1, float* datt2, float* dwfull_part,
                              float* de, int M, int L, int Dv, int A, int use_tanh, int acc_datt1, int64_t ld_datt2,
                              void* stream);

/* ------------------------------------------------------------------------------------------
 * The timestep loops of the XE / SCST training node as ONE call each (csrc/train_loop.hip): the loop bodies of
 * editnet.py:505-546 (train mode, teacher-forced, fixed-36 features) and of autograd's BPTT over them, i.e. exactly the
 * sequence of entry points above that show-edit-tell_amd/xe_sequence.py issues per timestep, with the per-sequence logs
 * addressed as base + t * (rows of one timestep).  Issued from Python the ~45 launches per timestep cost more host time than
 * their kernels take (16.3 ms of enqueue for a 16.0-ms step at B = 128).  All pointers are device pointers except `bts`
 * (host: rows still in the batch at every timestep, editnet.py:506) and `w`.  Logs are (T, B, .) row-major, the state logs
 * H1 / C1 / H2 / C2 (T + 1, B, D) with slot t = the state BEFORE timestep t.
 * ------------------------------------------------------------------------------------------ */
typedef struct SetXELoopArgs {
    int T, B, R, F, Tc, D, A, V, train;
    float p_embed, p_out;
    uint64_t seed, off_embed, off_out;           /* Philox: timestep t of a site draws at off_* + t */
    const int* bts;
    const SetEditNetWeights* w;
    const float *E, *al_wih, *al_whh;
    const int64_t* tok; int64_t tok_step, tok_stride;      /* word of (t, b) = tok[t * tok_step + b * tok_stride] */
    const float *X, *H, *Mem, *mask, *att1_c, *pre1, *att1; int64_t att1_step;
    float *EMB, *H1, *C1, *H2, *C2, *G1, *G2, *WHC, *ZT, *S, *TT, *ALPHAC, *ALPHAV, *ATT2C, *ATT2V, *SEL, *CNEW, *CG, *X2, *H2D;
    float *gated, *cx, *aimg;                    /* (B, D), (B, D), (B, F) scratch — or (T, B, .) logs, see step_logs */
    void* ws_l; size_t ws_l_bytes;               /* set_lstm_cell_workspace_bytes */
    void* ws_c; size_t ws_c_bytes;               /* set_editnet_attentions_workspace_bytes */
    void* ws_k; size_t ws_k_bytes;               /* set_copy_lstm_workspace_bytes */
    int step_logs;                               /* 1: gated / cx / aimg hold every timestep; the copy cell contracts its input
                                                    as the segments [h1 | gated | attend_img] where they lie and X2 / WHC are
                                                    packed ONCE after the loop (0: one packing launch per timestep) */
    const float* rmask; int64_t rmask_step;      /* adaptive features (editnet_adaptive.py:449-453): 0/1 region mask (B, R) of
                                                    timestep t at rmask + t * rmask_step, or NULL (fixed-36 features) */
} SetXELoopArgs;
int set_editnet_xe_train_loop_f32(const SetXELoopArgs* a, void* stream);

typedef struct SetXEBwdLoopArgs {
    int T, B, R, F, Tc, D, A, acc_datt1;
    float p_out;                                 /* 0: eval mode (no output dropout) */
    uint64_t seed, off_out;
    const int* bts;
    const float *cl_cnew_w, *cl_cmem_w, *cl_x2h_w, *cl_h2h_w, *w_ctx, *w_h1, *dec_cat, *al_wih, *al_whh, *va_full, *ca_full;
    const float *G1, *G2, *C1, *C2, *CG, *SEL, *CNEW, *ZT, *S, *TT, *ALPHAC, *ALPHAV, *ATT2C, *ATT2V, *X, *H, *Mem, *att1_c, *att1;
    int64_t att1_step;
    const float* dH2D;                           /* (T, B, D) gradient of fc's input */
    float *DU, *DGW, *DSZT, *DATT2, *DWFC, *DWFV, *DEC, *DEV, *DCTX, *DG1, *datt1; int64_t datt1_step;
    float *datt1c, *dMem;
    float *DC1[2], *DC2[2], *dcm, *dcn, *dop, *dalc;
    void* slab_ws[5]; size_t slab_ws_bytes;      /* one scratch region per product position of a timestep */
    float* tmp[11];                              /* (B, N) landing buffers of products the plan does not split */
    const float* DLAST;                          /* (T, B, D) or NULL: gradient entering h2 of timestep t directly (adaptive
                                                    features: d decoder_last_hidden at every row's last timestep, zero elsewhere) */
} SetXEBwdLoopArgs;
int set_editnet_xe_train_bwd_loop_f32(const SetXEBwdLoopArgs* a, void* stream);

/* ------------------------------------------------------------------------------------------
 * Host-side CIDEr-D for the self-critical reward (HOST pointers, no stream: per-sample n-gram work that shards with
 * the batch; the reference calls an external Python scorer at editnet_rl.py:636).  Sentences are int64 token ids.
 * create: the document-frequency table of preprocess_rl.py:7-55 as n_entries n-grams (entry i = lens[i] <= 4 ids at
 * tokens[4 i ..]) with their document counts df[i]; ref_len = number of documents; n = 4, sigma = 6 for CIDEr-D.
 * score: hypothesis i = hyp_tokens[hyp_off[i] .. hyp_off[i+1]) against reference set set_of_hyp[i]; set s = references
 * [ref_set_off[s], ref_set_off[s+1]); reference r = ref_tokens[ref_off[r] .. ref_off[r+1]).  scores: n_hyp doubles.
 * ------------------------------------------------------------------------------------------ */
void* set_ciderd_create(const int64_t* tokens, const int32_t* lens, const double* df, int64_t n_entries, double ref_len,
                        int n, double sigma);
void set_ciderd_destroy(void* scorer);
int set_ciderd_score(void* scorer, const int64_t* hyp_tokens, const int64_t* hyp_off, int n_hyp,
                     const int32_t* set_of_hyp, const int64_t* ref_tokens, const int64_t* ref_off,
                     const int64_t* ref_set_off, int n_sets, double* scores);

#ifdef __cplusplus
}
#endif
#endif /* SET_HIP_H */
