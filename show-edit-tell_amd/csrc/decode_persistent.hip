// Persistent free-running decode for small batches (B <= 8): the whole greedy loop of DCNet (dcnet_rl.py:286-346) as ONE
// launch of D / 4 workgroups that exchange activations through flag-in-data words instead of six launches per timestep.
// (EditNet's twin: decode_persistent_wide.hip.)
//
// At B = 4 a timestep of the per-step path is six dependent launches of 10-25 us that stream 123 MB of weights between
// them: every launch pays a boundary, a start-up and a tail, and the pointwise kernels between the GEMV launches run at the
// launch floor.  Here
//   * a workgroup OWNS 4 hidden units of both LSTM cells: the 16 gate rows of attention_lstm [W_ih[:, h2] | W_hh] and of
//     language_lstm [W_hh | W_ih[:, h1]] are contracted over the full K by its four waves (one K quarter each, weights
//     streamed as ready-made B operands of v_mfma_f32_16x16x4_f32: one 16-byte load per lane feeds four MFMAs, K permuted
//     identically in the activation operand read from LDS), the four partial tiles are added through LDS in wave order and
//     the cell update runs in the same workgroup: no split-K slabs, no pointwise launch, c1 / c2 never leave registers;
//   * the context half of language_lstm's input product is hoisted: W_ih[:, D:] ctx = sum_t alpha_t (W_ih[:, D:] enc_t), so
//     the prologue computes Pc = enc W_ih[:, D:]^T (B, T, 4D) once and a timestep only needs the attention weights —
//     16 MB of weights per timestep are not streamed at all, and every workgroup computes the (tiny) attention itself
//     (its cap_features_att rows stay in registers at B <= 4) instead of waiting for a context vector;
//   * cap_decoder_att(h1) (A x D) is the one product whose output every workgroup needs: A / 256 = 2 rows per workgroup;
//   * fc: a workgroup scores V / 256 (+) vocabulary rows and publishes (max, first arg-max, sum exp) per batch row; every
//     workgroup combines the 256 triples itself (same word everywhere, no broadcast round), workgroup 0 writes seq /
//     seq_logp and the loop's bookkeeping words.
// Four exchanges per timestep (h1, the attention projection, h2, the fc triples), each as 8-byte {value, tag} words
// (grid_barrier.h): a consumer polls the words it needs until they carry the exchange's tag — no counters, no fences.
// Weight tiles rotate through two register buffers of 64 KB per workgroup.  Where their requests sit matters: a wave's loads
// return in order (a poll waits for every tile requested before it), a saturated request queue blocks the wave at the next
// load, and the compiler drains outstanding requests at a loop header or branch join — so a tile is requested right before
// arithmetic that does not need it (fc's tiles around the attention scores), the h-dependent gate products of timestep t + 1
// are contracted while the fc triples of timestep t travel (S1'), and the B <= 4 variant has no loop around its per-row work.
// Teacher-forced mode (set_dcnet_xe_forward): words from the captions, scores written by the owners of the vocabulary rows,
// three exchanges.
// Same residency rule, fault word and event chain as the persistent encoder (grid_barrier.h PersistentGuard).  A poll that
// times out poisons seq_logp with NaN; the host raises SET_ERR_FAULT at its next call.
#include "decode_persistent.h"

namespace set {

struct PDecDcnetArgs {
    // weights
    const float* al_wih_h2; long long ld_al;     // attention_lstm.weight_ih[:, E + 2C:] (4D, D)
    const float* al_whh;                         // (4D, D)
    const float* ll_whh;                         // (4D, D)
    const float* ll_wih; long long ld_ll;        // language_lstm.weight_ih (4D, 2E): columns [0, D) are used here
    const float *ll_bih, *ll_bhh;
    const float *ca_dec_w, *ca_dec_b, *ca_full_w, *ca_full_b;
    const float *fc_w, *fc_b;
    const float* tok_table; long long ld_tab;    // columns [0, 4D): attention_lstm.W_ih[:, :E] relu(E[v])
    // per sequence (prologue outputs)
    const float* pre1;                           // (B, 4D) final_hidden columns + both biases of attention_lstm
    const float* att1_c;                         // (B, T, A)
    const float* mask;                           // (B, T)
    const float* pc;                             // (B, T, 4D) hoisted language_lstm.W_ih[:, D:] enc
    // exchange buffers: flag-in-data words of 8 bytes per float (grid_barrier.h), zero-filled before the launch
    void* x_h1; void* x_h2;                      // (B, D)
    void* x_att2;                                // (B, A) cap_decoder_att(h1), bias not added
    void* x_fc;                                  // (B, G) x (max, arg-max, sum exp, -)
    // outputs / loop words
    long long* it; int* unfinished; int* alive;
    long long* seq; float* seq_logp;
    unsigned* status; unsigned* fault; unsigned spin_limit; int test_stall;
    int B, D, T, A, V, max_len, rpw;
    long long start_idx, end_idx;
    // teacher-forced mode (set_dcnet_xe_forward, dcnet.py:333-348): words from caps, scores of the first bt rows written out,
    // no pick and no fourth exchange
    const long long* caps; long long caps_stride;
    float* predictions; long long ld_pred_b;     // (B, maxT, V)
    int dlen[PDEC_MAXB];                         // decode lengths, descending (0 = free-running)
    int stamp_wg;
    unsigned long long* stamps;                  // diagnostic (SET_PDEC_STAMPS=1): 100-MHz time stamps of workgroup 0, 16 per timestep
};

// RES: B <= 4 and T <= PDEC_TREG — a wave scores ONE fixed row, whose hoisted cap_features_att rows (loop-invariant, T x A
// floats = 160 registers per lane) then stay in registers for the whole decode
template <bool RES>
__global__ void __launch_bounds__(PDEC_THREADS, 1) dcnet_persistent_k(const PDecDcnetArgs P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ long long sTok[PDEC_MAXB];
    __shared__ int sUnf[PDEC_MAXB];
    const int tid = threadIdx.x, lane = tid & 63, kq = tid >> 6, r = lane & 15, g = lane >> 4;
    const int B = P.B, D = P.D, T = P.T, A = P.A, V = P.V;
    const int KQ = D >> 2, LDH = D + 4;
    const int wg = (int)blockIdx.x, u0 = wg * 4, G = (int)gridDim.x;
    float* sH1 = smem;                                   // (B, LDH) h1, rows padded by 16 bytes: conflict-free ds_read_b128
    float* sH2 = sH1 + B * LDH;
    float* sRed = sH2 + B * LDH;                         // [4 waves][3 tiles][16 batch rows][16 weight rows]
    float* sAl = sRed + 4 * 3 * 256;                     // (B, T) attention weights
    float* sG = sAl + PDEC_MAXB * PDEC_TMAX;             // (B, 16) gate pre-activations of language_lstm
    float* sA2 = sG + PDEC_MAXB * 16;                    // (B, A) cap_decoder_att(h1) of every row
    float* sF = sA2 + B * A;                             // (B, G, 4) fc triples of every workgroup
    float* sCon = sF + B * G * 4;                        // [cap_decoder_att.bias | cap_full_att.weight] (2, A): loop-invariant
    float* sPc = sCon + 2 * A;                           // (B, 16, TMAX) hoisted context products of the owned gate rows
    const LLWatch watch{P.status, P.fault, P.spin_limit};
    // exchange buffers (grid_barrier.h, flag-in-data words)
    const __amdgpu_buffer_rsrc_t h1rs = __builtin_amdgcn_make_buffer_rsrc((void*)P.x_h1, 0, B * D * 8, 0x00027000);
    const __amdgpu_buffer_rsrc_t h2rs = __builtin_amdgcn_make_buffer_rsrc((void*)P.x_h2, 0, B * D * 8, 0x00027000);
    const __amdgpu_buffer_rsrc_t a2rs = __builtin_amdgcn_make_buffer_rsrc((void*)P.x_att2, 0, B * A * 8, 0x00027000);
    const __amdgpu_buffer_rsrc_t fcrs = __builtin_amdgcn_make_buffer_rsrc((void*)P.x_fc, 0, B * G * 32, 0x00027000);

    // ---- initial state: h1 = h2 = 0, every row is fed <start>
    for (int i = tid; i < 2 * B * LDH; i += PDEC_THREADS) smem[i] = 0.f;
    for (int i = tid; i < A; i += PDEC_THREADS) { sCon[i] = P.ca_dec_b[i]; sCon[A + i] = P.ca_full_w[i]; }
    const float bf = P.ca_full_b[0];
    if (tid < B) { sTok[tid] = P.start_idx; sUnf[tid] = 1; }
    __syncthreads();

    // ---- weight tiles of this lane: gate row of output column r = gate (r >> 2) of unit u0 + (r & 3)
    const long long grow = (long long)(r >> 2) * D + u0 + (r & 3);
    const int kcol = kq * KQ + 4 * g;
    const float* pT0 = P.al_wih_h2 + grow * P.ld_al + kcol;
    const float* pT1 = P.al_whh + grow * D + kcol;
    const float* pT2 = P.ll_whh + grow * D + kcol;
    const float* pT3 = P.ll_wih + grow * P.ld_ll + kcol;
    // cap_decoder_att: A / G rows of the (A, D) projection per workgroup, full K (lanes r >= apw hold zeros)
    const int apw = A / G;                               // host: A % G == 0, apw <= 16
    const bool vD = r < apw;
    const float* pT4 = P.ca_dec_w + (long long)(vD ? wg * apw + r : 0) * D + kcol;
    const int row0 = wg * P.rpw;
    const float* pF[PDEC_FC_TILES];
    bool vF[PDEC_FC_TILES];
#pragma unroll
    for (int j = 0; j < PDEC_FC_TILES; ++j) {
        const int row = row0 + 16 * j + r;
        vF[j] = (16 * j + r < P.rpw) && row < V;
        pF[j] = P.fc_w + (long long)(vF[j] ? row : 0) * D + kcol;
    }
    const float fcb_lane = (lane < 16 * PDEC_FC_TILES && lane < P.rpw && row0 + lane < V) ? P.fc_b[row0 + lane] : 0.f;   // fc.bias of the row lane l scores
    const int arow = (r < B ? r : B - 1) * LDH;          // rows >= B repeat the last one: their outputs are never read
    const float* aH1 = sH1 + arow + kcol;
    const float* aH2 = sH2 + arow + kcol;

    // ---- thread roles
    const bool pair = tid < B * 4;                       // (batch row, unit) of the two cell updates
    const int pb = tid >> 2, pu = tid & 3;
    const bool gcol = tid < B * 16;                      // (batch row, gate row) of the hoisted-context gather
    const int cb = tid >> 4, crr = tid & 15;
    const long long ccol = (long long)(crr >> 2) * D + u0 + (crr & 3);
    float c1 = 0.f, c2 = 0.f, pre[4] = {0.f, 0.f, 0.f, 0.f};
    float b2 = 0.f;
    if (pair) {
#pragma unroll
        for (int q = 0; q < 4; ++q) pre[q] = P.pre1[(long long)pb * 4 * D + (long long)q * D + u0 + pu];
    }
    if (gcol) b2 = P.ll_bih[ccol] + P.ll_bhh[ccol];

    // weight tiles rotate through two register buffers:  X1: wb<-T3 wa<-T4 | S2 (wb<-F1, wa<-F0) | S5 (wa<-F2, wb<-T0', wa<-T1')
    // | S1' (wb<-T2')
    f32x4 wa[PDEC_KB], wb[PDEC_KB];
    // loop-invariant operands of the attention phase: the hoisted context products of this thread's gate row ...
    if (gcol)
        for (int tt = 0; tt < PDEC_TMAX; ++tt) sPc[tid * (PDEC_TMAX + 1) + tt] = tt < T ? P.pc[((long long)cb * T + tt) * 4 * D + ccol] : 0.f;
    __syncthreads();
    // ... and (RES) this wave's row of cap_features_att
    f32x4 a1r[RES ? PDEC_TREG : 1][2];
    float mk_res = 1.f;                                  // (RES) ... and its mask word
    if constexpr (RES) {
        const int brow = kq < B ? kq : B - 1;
        if (lane < T) mk_res = P.mask[(long long)brow * T + lane];
#pragma unroll
        for (int tt = 0; tt < PDEC_TREG; ++tt) {
            const int t2 = tt < T ? tt : T - 1;
            a1r[tt][0] = *reinterpret_cast<const f32x4*>(P.att1_c + ((long long)brow * T + t2) * A + lane * 4);
            a1r[tt][1] = *reinterpret_cast<const f32x4*>(P.att1_c + ((long long)brow * T + t2) * A + lane * 4 + 256);
        }
    }
    unsigned tag = 0;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // the h-dependent gate products of timestep t do not depend on the word chosen at t - 1: they are contracted at the END
    // of timestep t - 1, while its fc triples travel (S1' below; the F/A merge of the per-step loop).  t = 0: h1 = h2 = 0.
    f32x4 acc1 = zero4, acc2 = zero4;
    for (int t = 0; t < P.max_len; ++t) {
        // ================= S1: attention_lstm cell (h1) from the products of S1'
        PD_STAMP(0);
        float tg[4] = {0.f, 0.f, 0.f, 0.f};
        int bt = B;                                              // teacher-forced: rows whose caption is still running (sorted batch)
        if (P.caps) {
            bt = 0;
            for (int b = 0; b < B; ++b) bt += P.dlen[b] > t ? 1 : 0;
            if (bt == 0) break;
        }
        if (pair) {
            long long tok = P.caps ? P.caps[(long long)pb * P.caps_stride + t] : sTok[pb];
            tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);          // same clamp as embed_relu_k
            const float* trow = P.tok_table + tok * P.ld_tab + u0 + pu;
#pragma unroll
            for (int q = 0; q < 4; ++q) tg[q] = trow[(long long)q * D];
        }
        PD_STAMP(1);
#pragma unroll
        for (int e = 0; e < 4; ++e) sRed[(kq * 3 + 0) * 256 + (4 * g + e) * 16 + r] = acc1[e];
        __syncthreads();
        ++tag;                                                   // X1: h1
        if (pair && !(P.test_stall && wg == 0)) {
            float gq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int o = pb * 16 + q * 4 + pu;
                gq[q] = ((((sRed[o] + sRed[3 * 256 + o]) + sRed[6 * 256 + o]) + sRed[9 * 256 + o]) + pre[q]) + tg[q];
            }
            const float ai = pd_sigm(gq[0]), af = pd_sigm(gq[1]), ag = tanhf(gq[2]), ao = pd_sigm(gq[3]);
            c1 = af * c1 + ai * ag;
            ll_put(h1rs, pb * D + u0 + pu, ao * tanhf(c1), tag);
        }
        PD_STAMP(2);
        pd_load(wb, pT3);
        pd_load_if(wa, pT4, vD);
        PD_STAMP(3);
        ll_stage<256, 8>(h1rs, sH1, B, D, LDH, tag, watch, tid);
        __syncthreads();
        // ================= S2: language_lstm W_ih[:, :D] h1, this workgroup's rows of cap_decoder_att(h1)
        PD_STAMP(4);
        pd_mma(acc2, wb, aH1);
        f32x4 accd = zero4;
        pd_mma(accd, wa, aH1);
        PD_STAMP(5);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sRed[(kq * 3 + 1) * 256 + (4 * g + e) * 16 + r] = accd[e];
            sRed[(kq * 3 + 2) * 256 + (4 * g + e) * 16 + r] = acc2[e];
        }
        __syncthreads();
        ++tag;                                                   // X2: cap_decoder_att(h1) (without its bias)
        float g2 = 0.f;
        if (gcol) {
            const int o = cb * 16 + crr;
            if (crr < apw) {
                const float v = ((sRed[1 * 256 + o] + sRed[4 * 256 + o]) + sRed[7 * 256 + o]) + sRed[10 * 256 + o];
                ll_put(a2rs, cb * A + wg * apw + crr, v, tag);
            }
            g2 = ((sRed[2 * 256 + o] + sRed[5 * 256 + o]) + sRed[8 * 256 + o]) + sRed[11 * 256 + o];
        }
        PD_STAMP(6);
        ll_stage<256, 8>(a2rs, sA2, B, A, A, tag, watch, tid);
        __syncthreads();
        pd_load_if(wa, pF[0], vF[0]);                            // fc's first tile streams under the attention's arithmetic ...
        // ================= S3: caption attention of every row, in every workgroup (dcnet.py:261-268)
        // (RES: one row per wave and no loop around it — the compiler drains every outstanding request at a loop header, and
        // fc's tiles are meant to stream under this arithmetic)
        auto attend = [&](const int b) {
            const int a_lo = lane * 4, a_hi = lane * 4 + 256;
            f32x4 a2[2], wf[2];
            a2[0] = *reinterpret_cast<const f32x4*>(sA2 + b * A + a_lo);
            a2[1] = *reinterpret_cast<const f32x4*>(sA2 + b * A + a_hi);
            const float mk = RES ? mk_res : (lane < T ? P.mask[(long long)b * T + lane] : 1.f);
            a2[0] += *reinterpret_cast<const f32x4*>(sCon + a_lo); a2[1] += *reinterpret_cast<const f32x4*>(sCon + a_hi);
            wf[0] = *reinterpret_cast<const f32x4*>(sCon + A + a_lo); wf[1] = *reinterpret_cast<const f32x4*>(sCon + A + a_hi);
            constexpr int RB = 10;
            float mine = 0.f;                                   // lane tt keeps the score of position tt
            const unsigned long long live = __ballot(lane < T && mk != 0.f);
            const float* a1 = P.att1_c + (long long)b * T * A;
            for (int t0 = 0; t0 < (RES ? PDEC_TREG : T); t0 += RB) {
                f32x4 v[RB][2];
#pragma unroll
                for (int u = 0; u < RB; ++u) {
                    if constexpr (RES) {
                        v[u][0] = a1r[t0 + u < PDEC_TREG ? t0 + u : 0][0];     // (static indices: t0 is a multiple of RB, PDEC_TREG too)
                        v[u][1] = a1r[t0 + u < PDEC_TREG ? t0 + u : 0][1];
                    } else {
                        const int tt = t0 + u < T ? t0 + u : T - 1;
                        v[u][0] = *reinterpret_cast<const f32x4*>(a1 + (long long)tt * A + a_lo);
                        v[u][1] = *reinterpret_cast<const f32x4*>(a1 + (long long)tt * A + a_hi);
                    }
                }
#pragma unroll
                for (int u = 0; u < RB; ++u) {
                    const int tt = t0 + u;
                    // masked position (or past T): its score is -1e10 whatever it is.  Not in the resident variant: a branch here makes
                    // the compiler wait for fc's tile (requested above to stream under this arithmetic) before the first score
                    if (!RES && !((live >> tt) & 1ull)) continue;
                    const float sc = pw_wsum(pd_score8(v[u][0] + a2[0], v[u][1] + a2[1], wf[0], wf[1]));
                    if (lane == tt) mine = sc;
                }
            }
            // masked softmax over the T <= 32 scores inside the wave (one score per lane)
            const float sc = lane < T ? ((mk == 0.f) ? -1e10f : (mine + bf)) : -INFINITY;
            const float m = pw_wmax(sc);
            const float ex = lane < T ? expf(sc - m) : 0.f;
            const float sum = pw_wsum(ex);
            if (lane < T) sAl[b * PDEC_TMAX + lane] = ex / sum;
        };
        if constexpr (RES) { if (kq < B) attend(kq); }
        else { for (int b = kq; b < B; b += 4) attend(b); }
        pd_load_if(wb, pF[1], vF[1]);                            // ... the second under the cell update and the h2 exchange
        __syncthreads();
        PD_STAMP(7);
        if (gcol) {
            float s = 0.f;
            for (int tt = 0; tt < T; ++tt) s += sAl[cb * PDEC_TMAX + tt] * sPc[tid * (PDEC_TMAX + 1) + tt];
            sG[cb * 16 + crr] = (g2 + s) + b2;
        }
        __syncthreads();
        ++tag;                                                   // X3: h2
        if (pair) {
            const float* gp = sG + pb * 16 + pu;
            const float ai = pd_sigm(gp[0]), af = pd_sigm(gp[4]), ag = tanhf(gp[8]), ao = pd_sigm(gp[12]);
            c2 = af * c2 + ai * ag;
            ll_put(h2rs, pb * D + u0 + pu, ao * tanhf(c2), tag);
        }
        PD_STAMP(8);
        ll_stage<256, 8>(h2rs, sH2, B, D, LDH, tag, watch, tid);
        __syncthreads();
        PD_STAMP(9);
        // ================= S5: fc over this workgroup's vocabulary rows, local (max, first arg-max, sum exp) per batch row
        const bool more = t + 1 < P.max_len;
        f32x4 accf0 = zero4, accf1 = zero4, accf2 = zero4;
        pd_mma(accf0, wa, aH2);
        pd_load_if(wa, pF[2], vF[2]);
        pd_mma(accf1, wb, aH2);
        if (more) pd_load(wb, pT0);
        pd_mma(accf2, wa, aH2);
        if (more) pd_load(wa, pT1);
        PD_STAMP(10);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sRed[(kq * 3 + 0) * 256 + (4 * g + e) * 16 + r] = accf0[e];
            sRed[(kq * 3 + 1) * 256 + (4 * g + e) * 16 + r] = accf1[e];
            sRed[(kq * 3 + 2) * 256 + (4 * g + e) * 16 + r] = accf2[e];
        }
        __syncthreads();
        if (P.caps) {
            // teacher-forced: the scores themselves, rows 0 .. bt - 1 (dcnet.py:347: predictions[:batch_size_t, t, :] = preds)
            for (int b = kq; b < bt; b += 4) {
                const int j = lane >> 4, rr = lane & 15, row = row0 + lane;
                if (lane < 16 * PDEC_FC_TILES && lane < P.rpw && row < V) {
                    const int o = j * 256 + b * 16 + rr;
                    P.predictions[(long long)b * P.ld_pred_b + (long long)t * V + row] =
                        (((sRed[o] + sRed[3 * 256 + o]) + sRed[6 * 256 + o]) + sRed[9 * 256 + o]) + fcb_lane;
                }
            }
            if (more) {                                          // S1' (see below)
                acc1 = zero4; acc2 = zero4;
                pd_mma(acc1, wb, aH2);
                pd_load(wb, pT2);
                pd_mma(acc1, wa, aH1);
                pd_mma(acc2, wb, aH2);
            }
            __syncthreads();                                     // sRed is rewritten by the next timestep's S1
            continue;
        }
        ++tag;                                                   // X4: (max, arg-max, sum exp) of every workgroup's rows
        {   // (resident variant: one row per wave, no loop — see the attention phase)
            auto row_work = [&](const int b) {
                const int j = lane >> 4, rr = lane & 15, row = row0 + lane;
                const bool ok = lane < 16 * PDEC_FC_TILES && lane < P.rpw && row < V;
                float x = -INFINITY;
                if (ok) {
                    const int o = j * 256 + b * 16 + rr;
                    x = (((sRed[o] + sRed[3 * 256 + o]) + sRed[6 * 256 + o]) + sRed[9 * 256 + o]) + fcb_lane;
                }
                float best = -INFINITY;
                int bi = 0x7fffffff;
                if (x > best) { best = x; bi = row; }
                    pw_wargmax(best, bi);
                // (a NaN score never wins a comparison: it reaches the sum instead and the row's log-prob is NaN)
                float se = ok ? expf(x - best) : 0.f;
                if (best == -INFINITY) se = ok ? x : 0.f;            // no finite score here: 0 for an empty range, NaN for NaN scores
                se = pw_wsum(se);
                if (lane < 4) ll_put(fcrs, (b * G + wg) * 4 + lane, lane == 0 ? best : (lane == 1 ? __int_as_float(bi) : (lane == 2 ? se : 0.f)), tag);
            };
            if constexpr (RES) { if (kq < B) row_work(kq); }
            else { for (int b = kq; b < B; b += 4) row_work(b); }
        }
        PD_STAMP(11);
        // ================= S1': [W_ih[:, h2] | W_hh] of attention_lstm and language_lstm.W_hh for timestep t + 1, while the
        // triples travel: nothing here waits for the word
        if (more) {
            acc1 = zero4; acc2 = zero4;
            pd_mma(acc1, wb, aH2);
            pd_load(wb, pT2);
            pd_mma(acc1, wa, aH1);
            pd_mma(acc2, wb, aH2);
        }
        PD_STAMP(12);
        ll_stage<256, 8>(fcrs, sF, B * G, 4, 4, tag, watch, tid);
        __syncthreads();
        // ================= S6: every workgroup combines the G triples of every row: same word everywhere
        {   // (resident variant: one row per wave, no loop — see the attention phase)
            auto row_work = [&](const int b) {
                float best = -INFINITY, tot = 0.f;
                int bi = 0x7fffffff;
                float pm[4], ps[4];
                int pi[4];
    #pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int wi = lane + 64 * i;
                    const bool have = wi < G;
                    const f32x4 e4 = have ? *reinterpret_cast<const f32x4*>(sF + (b * G + wi) * 4) : zero4;
                    pm[i] = have ? e4[0] : -INFINITY;
                    pi[i] = have ? __float_as_int(e4[1]) : 0x7fffffff;
                    ps[i] = have ? e4[2] : 0.f;
                }
    #pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (pm[i] > best || (pm[i] == best && pi[i] < bi)) { best = pm[i]; bi = pi[i]; }
                    pw_wargmax(best, bi);
    #pragma unroll
                for (int i = 0; i < 4; ++i) tot += (pm[i] == -INFINITY) ? ps[i] : ps[i] * expf(pm[i] - best);
                tot = pw_wsum(tot);
                if (lane == 0) {
                    float logp = (best - best) - logf(tot);           // log_softmax at the arg-max, as greedy_pick_k writes it
                    if (bi == 0x7fffffff) { bi = 0; logp = __builtin_nanf(""); }   // all-NaN row: word 0 and a NaN log-prob
                    long long it = bi;
                    if (it == P.end_idx) it = 0;
                    const int unf = (t == 0) ? (it > 0) : (sUnf[b] && it > 0);
                    it = unf ? it : 0;
                    if (wg == 0) {
                        P.seq[(long long)b * P.max_len + t] = it;
                        P.seq_logp[(long long)b * P.max_len + t] = logp;
                        P.unfinished[b] = unf;
                        P.it[b] = it;
                    }
                    sTok[b] = it;
                    sUnf[b] = unf;
                }
            };
            if constexpr (RES) { if (kq < B) row_work(kq); }
            else { for (int b = kq; b < B; b += 4) row_work(b); }
        }
        __syncthreads();
        PD_STAMP(13);
        int alive = 0;
        for (int b = 0; b < B; ++b) alive += sUnf[b];
        if (wg == 0 && tid == 0) P.alive[t] = alive;
        if (alive == 0) break;                                    // dcnet_rl.py:341-342: every caption has ended
    }
    // ---- an exchange of this launch timed out: never hand the result out as a decode (see encoder_persistent.hip)
    __shared__ unsigned s_bad;
    if (tid == 0) s_bad = __hip_atomic_load(P.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_bad) {
        const float qnan = __builtin_nanf("");
        if (P.caps) {
            // teacher-forced: EVERY score this workgroup wrote (its vocabulary rows, all rows and timesteps) — each workgroup
            // poisons its own region after its own loop, so no later store of another workgroup can undo it
            const int row0 = wg * P.rpw;
            for (int i = tid; i < B * P.max_len * P.rpw; i += PDEC_THREADS) {
                const int row = row0 + i % P.rpw, bt_ = i / P.rpw;
                if (row < V) P.predictions[(long long)(bt_ / P.max_len) * P.ld_pred_b + (long long)(bt_ % P.max_len) * V + row] = qnan;
            }
        } else if (wg == 0) {
            for (int i = tid; i < B * P.max_len; i += PDEC_THREADS) { P.seq_logp[i] = qnan; P.seq[i] = 0; }
        }
    }
}

static int g_pdec_capacity[64][2] = {};
static int g_pdec_capacity_lds[64][2] = {};
static int pdec_lds_floats(int B, int D, int A) {
    return 2 * B * (D + 4) + 4 * 3 * 256 + PDEC_MAXB * PDEC_TMAX + PDEC_MAXB * 16 + B * A + B * (D / 4) * 4 + 2 * A + B * 16 * (PDEC_TMAX + 1);
}


bool dcnet_persistent_ok(const SetDcnetDims* d, int max_len) {
    const int on = env_int("SET_DEC_PERSISTENT", 1);                 // (read per call: tests and A/B runs flip it inside one process)
    const int maxb = env_int("SET_DEC_PERSISTENT_MAXB", PDEC_MAXB);
    if (!on || d->B > maxb || d->B > PDEC_MAXB || max_len < 1) return false;
    if (d->D != 64 * PDEC_KB || d->E != d->D || 2 * d->C != d->D || d->A != 512 || d->T > PDEC_TMAX) return false;   // A = 2 rows of cap_decoder_att per workgroup, 512 scores per wave pass
    const int G = d->D / 4;
    if ((d->V + G - 1) / G > 16 * PDEC_FC_TILES) return false;
    if (pdec_lds_floats(PDEC_MAXB, d->D, d->A) * (int)sizeof(float) > persistent_lds_limit()) return false;   // (a 64-KB-LDS device)
    return !persistent_disabled();
}

// exchange region of one decode: [status line | h1 | h2 | cap_decoder_att(h1) | fc triples] as 8-byte flag-in-data words
size_t dcnet_persistent_xbytes(int B, int D, int A) {
    if (B > PDEC_MAXB) return 0;
    return 128 + (size_t)B * D * 8 * 2 + (size_t)B * A * 8 + (size_t)B * (D / 4) * 32;
}

// the greedy loop after set_dcnet_begin's prologue.  `pc` = the hoisted context products (B, T, 4D), `xbuf` = exchange region
// (dcnet_persistent_xbytes).  SET_ERR_UNSUPPORTED: nothing was touched, the caller runs the per-step loop.
int dcnet_persistent_greedy(const SetDcnetWeights* w, const SetDcnetDims* d, const float* pre1, const float* att1_c,
                            const float* mask, const float* pc, void* xbuf, long long* it, int* unfinished, int* alive,
                            long long start_idx, long long end_idx, int max_len, long long* seq, float* seq_logp,
                            hipStream_t s, const PDecTeacher* teach) {
    if (!dcnet_persistent_ok(d, max_len)) return SET_ERR_UNSUPPORTED;
    const int B = d->B, D = d->D, E = d->E, C = d->C, G = D / 4;
    PDecDcnetArgs P{};
    P.al_wih_h2 = w->al_wih + E + 2 * C; P.ld_al = 3LL * E;
    P.al_whh = w->al_whh; P.ll_whh = w->ll_whh; P.ll_wih = w->ll_wih; P.ld_ll = 2LL * E;
    P.ll_bih = w->ll_bih; P.ll_bhh = w->ll_bhh;
    P.ca_dec_w = w->ca_dec_w; P.ca_dec_b = w->ca_dec_b; P.ca_full_w = w->ca_full_w; P.ca_full_b = w->ca_full_b;
    P.fc_w = w->fc_w; P.fc_b = w->fc_b; P.tok_table = w->tok_table; P.ld_tab = 4LL * D + 8LL * C;
    P.pre1 = pre1; P.att1_c = att1_c; P.mask = mask; P.pc = pc;
    {
        char* x = (char*)xbuf;
        P.status = (unsigned*)x; x += 128;
        P.x_h1 = x; x += (size_t)B * D * 8;
        P.x_h2 = x; x += (size_t)B * D * 8;
        P.x_att2 = x; x += (size_t)B * d->A * 8;
        P.x_fc = x;
    }
    P.it = it; P.unfinished = unfinished; P.alive = alive; P.seq = seq; P.seq_logp = seq_logp;
    P.B = B; P.D = D; P.T = d->T; P.A = d->A; P.V = d->V; P.max_len = max_len; P.rpw = (d->V + G - 1) / G;
    P.start_idx = start_idx; P.end_idx = end_idx;
    if (teach) {
        P.caps = (const long long*)teach->caps; P.caps_stride = teach->caps_stride;
        P.predictions = teach->predictions; P.ld_pred_b = (long long)max_len * d->V;
        for (int b = 0; b < B; ++b) P.dlen[b] = teach->host_decode_lengths[b];
    }
    const int lds = pdec_lds_floats(B, D, d->A) * (int)sizeof(float);
    PersistentGuard guard;
    if (guard.rc != SET_OK) return guard.rc;
    const int dev = guard.dev;
    P.spin_limit = guard.spin_limit();              // bound of one wait, ticks of the 100-MHz counter
    P.test_stall = guard.test_stall(); P.fault = guard.fault;
    // residency: every workgroup must be on the chip at once (see encoder_persistent.hip penc_fits)
    // (function attributes are per device; a device whose LDS limit is below the request, e.g. a 64-KB part, is answered with
    // SET_ERR_UNSUPPORTED — the caller's per-step loop — never with a HIP error)
    static bool configured[2][64] = {};
    const int lds_max = pdec_lds_floats(PDEC_MAXB, D, d->A) * (int)sizeof(float);
    if (guard.set_lds(reinterpret_cast<const void*>(&dcnet_persistent_k<true>), lds_max, configured[0]) != SET_OK ||
        guard.set_lds(reinterpret_cast<const void*>(&dcnet_persistent_k<false>), lds_max, configured[1]) != SET_OK)
        return SET_ERR_UNSUPPORTED;
    const bool res = B <= 4 && d->T <= PDEC_TREG;
    // resident workgroups the device admits, asked with the LDS size of THIS batch (re-asked when a larger one comes along)
    int& cap = g_pdec_capacity[dev][res ? 1 : 0];
    int& cap_lds = g_pdec_capacity_lds[dev][res ? 1 : 0];
    if (cap == 0 || lds > cap_lds) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, res ? reinterpret_cast<const void*>(&dcnet_persistent_k<true>)
                                                                       : reinterpret_cast<const void*>(&dcnet_persistent_k<false>), PDEC_THREADS,
                                                         (size_t)lds) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
            (void)hipGetLastError();
            return SET_ERR_UNSUPPORTED;
        }
        cap = per_cu * cus;
        cap_lds = lds;
        if (cap <= 0) cap = -1;
        const int forced = env_int("SET_PENC_TEST_CAPACITY", 0);
        if (forced > 0) cap = forced;
    }
    if (G > cap) return SET_ERR_UNSUPPORTED;
    const double wbytes = 4.0 * ((double)d->V * D + 4.0 * 4 * D * D + (double)d->A * D);
    ProfScope ps("persistent_decode", s, 2.0 * B * wbytes / 4.0 * max_len, wbytes * max_len);
    SET_TRY(guard.serialise(s));
    SET_HIP_TRY(hipMemsetAsync(xbuf, 0, dcnet_persistent_xbytes(B, D, d->A), s));    // no word of an earlier decode may carry a tag of this one
    SET_TRY(pd_stamps_begin(&P.stamps, &P.stamp_wg, s));
    if (res) hipLaunchKernelGGL(dcnet_persistent_k<true>, dim3(G), dim3(PDEC_THREADS), lds, s, P);
    else hipLaunchKernelGGL(dcnet_persistent_k<false>, dim3(G), dim3(PDEC_THREADS), lds, s, P);
    SET_LAUNCH_CHECK();
    SET_TRY(guard.launched(s));
    SET_TRY(pd_stamps_report(P.stamps, P.stamp_wg, 13, max_len, s));
    return SET_OK;
}

}  // namespace set
