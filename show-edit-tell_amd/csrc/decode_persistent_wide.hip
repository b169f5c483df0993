// Persistent greedy decode of EditNet for 1 .. 16 rows (editnet_rl.py:485-549): ONE launch of D / 4 workgroups for the whole
// free-running loop, the construction of decode_persistent.hip (DCNet) (a workgroup OWNS four hidden units, streams their
// weight rows as B operands of v_mfma_f32_16x16x4_f32 — the tile has 16 batch rows, all of them used here — and finishes every
// pointwise stage itself; what other workgroups need travels as flag-in-data words, grid_barrier.h) re-cut for the things
// that grow with the batch:
//   * ONE activation buffer in LDS instead of two.  attention_lstm's W_hh h1 product of timestep t + 1 is contracted right
//     after h1 arrives (S2, with the other two products that read h1) instead of at the end of the timestep, so h1 is dead
//     after S2 and one (B, D) buffer carries h1 -> attend_cap -> c_new -> h2 -> the fc triples in turn (a wave polls the two
//     projection rows it scores with straight into registers): 66 KB at 16 rows, where two buffers + the projections + the
//     triples would need 260 KB;
//   * the caption attention is NOT recomputed by every workgroup (256 x B x T x A tanh per timestep and B x T x A floats of
//     cap_features_att from L2 per workgroup: fine at 4 rows, 10 us of a timestep at 8): its B x T scores are spread over the
//     grid exactly like the B x R visual scores — at most one of each per wave, whose cap_features_att / features_att row
//     stays in registers — and travel in one more exchange; every workgroup then runs the two (tiny) softmaxes itself;
//   * wave reductions (two softmaxes, three arg-max / sum-exp passes for four rows per wave) run on DPP row permutations +
//     four v_readlane instead of ds_bpermute butterflies (25 dependent LDS round trips per softmax: 8 us of a timestep at
//     16 rows), and the hoisted-product tables in LDS have odd row strides (no bank conflicts in the 256-thread gathers).
// (Tried and removed: "courier" waves 4-5 that do every poll / LDS fill so that the compute waves' weight requests stay in
// flight across LDS-only barriers — six waves leave 256 registers per wave, the compute path spills and it measured 20 %
// slower, EXPERIMENTS.md 5.3.)
// Per timestep (eval mode, token table):
//   S1   attention_lstm cell from the products of S2(t-1) + S1'(t-1) and the token-table row      -> h1            X1 (B, D)
//   S2   copy_lstm.x2h[:, :D] h1; the mixed tile (4 rows each of context_gate, tc_affine, [cap_decoder_att ; decoder_att]);
//        attention_lstm.W_hh h1 for timestep t + 1                                               -> projections   X2 (B, 2A)
//   S3a  this wave's caption score e_c[b, t] and visual score e[b, r]                            -> scores        X3a (B, T + R)
//   S3b  both softmaxes of every row, SelectC's arg-max; context gate of the owned columns from the hoisted P; sel /
//        gate_cmem(sel) from Mem / Q rows j*                                                     -> attend_cap    X3b (B, D)
//   S4   copy_lstm.x2h[:, D:2D] attend_cap + sum_r alpha_r Pv[b, r]                              -> c_new         X4 (B, D)
//   S5   gate_cnew rows of the owned units, copy gate                                            -> h2            X5 (B, D)
//   S6   fc rows of this workgroup, (max, first arg-max, sum exp) per batch row                   -> triples       X6 (B, G)
//   S1'  attention_lstm.W_ih[:, h2] h2 and copy_lstm.h2h h2 for timestep t + 1, while the triples travel
//   S7   every workgroup combines the G triples: same word everywhere; workgroup 0 writes seq / seq_logp
// Buffer reuse without a barrier (grid_barrier.h): every workgroup contributes to X1 and consumes exchanges in order, so a
// word of any exchange of timestep t is overwritten (timestep t + 1) only after every workgroup consumed it.
// Teacher-forced mode (set_editnet_xe_forward): words from the captions, scores written by the owners of the vocabulary rows.
#include "decode_persistent.h"

namespace set {

namespace {

constexpr int PW_U = 16;           // 16-byte requests a lane keeps in flight while it fills LDS from an exchange buffer
constexpr int PW_BEAM_K = 4;       // beam mode: hypotheses (= rows) at most
constexpr int PW_BEAM_W = 12;      // ... words a workgroup publishes per row: max, sum exp, 4 x (value, index), 2 pads
constexpr int PW_RS = PDEC_RREG + 1;   // row strides of the hoisted-product tables in LDS: odd, so that the (thread, index) gathers
constexpr int PW_TS = PDEC_TMAX + 1;   // of 256 threads spread over all banks

}  // namespace

// BEAM: the rows are the k <= PW_BEAM_K hypotheses of ONE image and the loop is the reference's beam search (editnet.py:643-713)
// instead of the greedy loop — see "beam mode" below.
template <bool BEAM>
__global__ void __launch_bounds__(PDEC_THREADS, 1) editnet_persistent_wide_k(const PDecEditArgs P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ long long sTok[PDW_MAXB];
    __shared__ int sUnf[PDW_MAXB], sJs[PDW_MAXB];
    __shared__ float sWj[PDW_MAXB];
    const int tid = threadIdx.x, lane = tid & 63, kq = tid >> 6, r = lane & 15, g = lane >> 4;
    constexpr bool worker = true;
    const int B = P.B, D = P.D, T = P.T, R = P.R, A = P.A, V = P.V;
    const int KQ = D >> 2, LDH = D + 4;
    const int wg = (int)blockIdx.x, u0 = wg * 4, G = (int)gridDim.x;
    float* sX = smem;                                    // (B, LDH): h1 -> attend_cap -> c_new -> h2 -> triples (B, G, 4) in turn
    float* sRed = sX + B * LDH;                          // [4 waves][3 tiles][16][16]
    float* sAlc = sRed + 4 * 3 * 256;                    // (B, TMAX) caption scores -> weights
    float* sAlv = sAlc + PDW_MAXB * PDEC_TMAX;           // (B, 64) visual scores -> weights
    float* sG = sAlv + PDW_MAXB * 64;                    // (B, 16) copy_lstm gate pre-activations
    float* sZ = sG + PDW_MAXB * 16;                      // (B, 8) [sum alpha P_z (4) | sum alpha P_s (4)] of the owned columns
    float* sM = sZ + PDW_MAXB * 8;                       // (B, 8) [context_gate.W h1 (4) | tc_affine.W h1 (4)]
    float* sPv = sM + PDW_MAXB * 8;                      // (B, 16, RREG) hoisted region products of the owned gate rows
    float* sPz = sPv + B * 16 * PW_RS;                   // (B, 8, TMAX) hoisted caption-context products of the owned columns
    float* sCon = sPz + B * 8 * PW_TS;               // [cap_decoder_att.b | cap_full_att.w | decoder_att.b | full_att.w] (4, A)
    const float* sF = sX;
    // beam mode only: every workgroup's per-slice candidates, the previous timestep's h2h products (added through the parent
    // map), the cell states on their way through the parent map, the per-row candidate lists of the pick
    float* sFB = sCon + 4 * A;                           // (B, G, PW_BEAM_W)
    float* sRedP = sFB + (BEAM ? B * G * PW_BEAM_W : 0); // [4 waves][16][16]
    float* sCst = sRedP + (BEAM ? 4 * 256 : 0);          // (2, B, 4) c1 | c2 of the owned units
    float* sCand = sCst + (BEAM ? 2 * PW_BEAM_K * 4 : 0);// (B, K, 2) (score, flat index) of every row's best K candidates
    __shared__ int sPar[PW_BEAM_K];                      // parent slot of every slot (identity before the first pick)
    __shared__ float sScore[PW_BEAM_K];                  // running scores of the slots (-inf = dead)
    __shared__ int sKleft;
    __shared__ float sBest;                              // best completed hypothesis so far
    const LLWatch watch{P.status, P.fault, P.spin_limit};

#define PW_SYNC() __syncthreads()
#define PW_STAGE(RS, DST, ROWS, COLS, LD) ll_stage<PDEC_THREADS, PW_U>(RS, DST, ROWS, COLS, LD, tag, watch, tid)

    // ---- weight tiles of this lane (workers)
    const long long grow = (long long)(r >> 2) * D + u0 + (r & 3);       // gate row of the 4D-row matrices
    const int kcol = (kq & 3) * KQ + 4 * g;
    const float* pT0 = P.al_wih + grow * P.ld_ih + 2 * D + kcol;
    const float* pT1 = P.al_whh + grow * D + kcol;
    const float* pT2 = P.cl_h2h_w + grow * D + kcol;
    const float* pT3 = P.cl_x2h_w + grow * P.ld_x2h + kcol;
    const float* pT5 = P.cl_x2h_w + grow * P.ld_x2h + D + kcol;
    const float* pT4;                                    // mixed tile: rows 0-3 context_gate, 4-7 tc_affine, 8-11 attention projections
    {
        const int j = wg * 4 + (r & 3);                  // row of the stacked [cap_decoder_att ; decoder_att] (2A rows)
        pT4 = r < 4 ? P.ca_gate_w + (long long)(u0 + r) * 3 * D + D + kcol
            : r < 8 ? P.ca_tc_w + (long long)(u0 + r - 4) * 2 * D + D + kcol
                    : (j < A ? P.ca_dec_w + (long long)j * D : P.va_dec_w + (long long)(j - A) * D) + kcol;
    }
    const bool v4 = r < 12;
    const bool v6 = r < 4;
    const float* pT6 = P.cl_cnew_w + (long long)(u0 + (r & 3)) * D + kcol;
    const int row0 = wg * P.rpw;
    const float* pF[PDEC_FC_TILES];
    bool vF[PDEC_FC_TILES];
#pragma unroll
    for (int j = 0; j < PDEC_FC_TILES; ++j) {
        const int row = row0 + 16 * j + r;
        vF[j] = (16 * j + r < P.rpw) && row < V;
        pF[j] = P.fc_w + (long long)(vF[j] ? row : 0) * D + kcol;
    }
    const float fcb_lane = (lane < 16 * PDEC_FC_TILES && lane < P.rpw && row0 + lane < V) ? P.fc_b[row0 + lane] : 0.f;
    const int arow = (r < B ? r : B - 1) * LDH;          // rows >= B repeat the last one: their outputs are never read
    const float* aX = sX + arow + kcol;

    // ---- thread roles (workers) and their loop-invariant operands
    const bool pair = worker && tid < B * 4;             // (batch row, owned unit): the two cells, the context gate, the copy gate
    const int pb = tid >> 2, pu = tid & 3, pd = u0 + pu;
    const bool gcol = worker && tid < B * 16;            // (batch row, gate row) of copy_lstm's hoisted region products
    const int cb = tid >> 4, crr = tid & 15;
    const long long ccol = (long long)(crr >> 2) * D + u0 + (crr & 3);
    const bool zrole = worker && tid < B * 8;            // (batch row, [z | s] column) of the hoisted caption-context products
    const int zb = tid >> 3, zc8 = tid & 7;
    float c1 = 0.f, c2 = 0.f, pre[4] = {0.f, 0.f, 0.f, 0.f};
    float bg = 0.f, bsc = 0.f, btc = 0.f, bcn = 0.f, bcm = 0.f, b2 = 0.f;
    if (pair) {
#pragma unroll
        for (int q = 0; q < 4; ++q) pre[q] = P.pre1[(long long)pb * 4 * D + (long long)q * D + pd];
        bg = P.ca_gate_b[pd]; bsc = P.ca_sc_b[pd]; btc = P.ca_tc_b[pd]; bcn = P.cl_cnew_b[pd]; bcm = P.cl_cmem_b[pd];
    }
    if (gcol) {
        for (int rr = 0; rr < PDEC_RREG; ++rr) sPv[tid * PW_RS + rr] = rr < R ? P.pv[((long long)cb * R + rr) * 4 * D + ccol] : 0.f;
        b2 = P.cl_x2h_b[ccol] + P.cl_h2h_b[ccol];
    }
    if (zrole)
        for (int tt = 0; tt < PDEC_TMAX; ++tt)
            sPz[tid * PW_TS + tt] = tt < T ? P.capP[((long long)zb * T + tt) * 2 * D + (zc8 < 4 ? u0 + zc8 : D + u0 + zc8 - 4)] : 0.f;
    if (worker)
        for (int i = tid; i < A; i += PDEC_THREADS) {
            sCon[i] = P.ca_dec_b[i]; sCon[A + i] = P.ca_full_w[i]; sCon[2 * A + i] = P.va_dec_b[i]; sCon[3 * A + i] = P.va_full_w[i];
        }
    const int a_lo = lane * 4, a_hi = lane * 4 + 256;
    const float cbf = P.ca_full_b[0], vbf = P.va_full_b[0];
    // the ONE caption score and the ONE visual score this wave owns: index s = wg + G * wave over (b, t) / (b, r); the
    // cap_features_att / features_att rows they need are loop-invariant and stay in registers
    const int s_idx = wg + G * (kq & 3);
    const bool cs_on = worker && s_idx < B * T, vs_on = worker && s_idx < B * R;
    const int cs_b = cs_on ? s_idx / T : 0, cs_t = cs_on ? s_idx % T : 0;
    const int vs_b = vs_on ? s_idx / R : 0, vs_r = vs_on ? s_idx % R : 0;
    const f32x4 ca1_0 = *reinterpret_cast<const f32x4*>(P.att1_c + ((long long)cs_b * T + cs_t) * A + a_lo);
    const f32x4 ca1_1 = *reinterpret_cast<const f32x4*>(P.att1_c + ((long long)cs_b * T + cs_t) * A + a_hi);
    const float cs_mask = P.mask[(long long)cs_b * T + cs_t];
    const f32x4 va1_0 = *reinterpret_cast<const f32x4*>(P.att1 + ((long long)vs_b * R + vs_r) * A + a_lo);
    const f32x4 va1_1 = *reinterpret_cast<const f32x4*>(P.att1 + ((long long)vs_b * R + vs_r) * A + a_hi);
    // exchange buffers
    const __amdgpu_buffer_rsrc_t h1rs = __builtin_amdgcn_make_buffer_rsrc(P.x_h1, 0, B * D * 8, 0x00027000);
    const __amdgpu_buffer_rsrc_t a2rs = __builtin_amdgcn_make_buffer_rsrc(P.x_a2, 0, B * 2 * A * 8, 0x00027000);
    const __amdgpu_buffer_rsrc_t gtrs = __builtin_amdgcn_make_buffer_rsrc(P.x_gt, 0, B * D * 8, 0x00027000);
    const __amdgpu_buffer_rsrc_t csrs = __builtin_amdgcn_make_buffer_rsrc(P.x_cs, 0, B * PDEC_TMAX * 8, 0x00027000);
    const __amdgpu_buffer_rsrc_t vsrs = __builtin_amdgcn_make_buffer_rsrc(P.x_vs, 0, B * 64 * 8, 0x00027000);
    const __amdgpu_buffer_rsrc_t cnrs = __builtin_amdgcn_make_buffer_rsrc(P.x_cn, 0, B * D * 8, 0x00027000);
    const __amdgpu_buffer_rsrc_t h2rs = __builtin_amdgcn_make_buffer_rsrc(P.x_h2, 0, B * D * 8, 0x00027000);
    const __amdgpu_buffer_rsrc_t fcrs = __builtin_amdgcn_make_buffer_rsrc(P.x_fc, 0, B * G * 32, 0x00027000);
    const __amdgpu_buffer_rsrc_t fbrs = __builtin_amdgcn_make_buffer_rsrc(BEAM ? P.x_fcb : P.x_fc, 0, BEAM ? B * G * PW_BEAM_W * 8 : 32, 0x00027000);

    // ---- initial state (init_hidden_state, editnet.py:494-495): zeros; every row is fed <start>
    if (tid < B) { sTok[tid] = P.start_idx; sUnf[tid] = 1; }
    if (BEAM && tid < PW_BEAM_K) { sPar[tid] = tid; sScore[tid] = tid == 0 ? 0.f : -INFINITY; }   // step 1: all rows are identical, only row 0 counts
    if (BEAM) for (int i = tid; i < 4 * 256; i += PDEC_THREADS) sRedP[i] = 0.f;                   // h2h h2 of the zero initial state
    if (BEAM && tid == 0) {
        sKleft = B; sBest = -INFINITY;
        if (wg == 0) { P.bm_best_score[0] = -INFINITY; P.bm_best_word[0] = 0; P.bm_result[0] = -1; P.bm_result[1] = -1; P.bm_result[2] = B; P.bm_result[3] = 0; }
    }
    __syncthreads();

    // weight tiles rotate through three register buffers (one workgroup per CU: 512 registers per lane):
    //   X1: wb<-T3 wa<-T4 wc<-T1 | after the projections' poll: wb<-T5 | after the scores' poll: wa<-T6 wc<-F0 | S4: wb<-F1 |
    //   S5: wa<-F2 | S6: wc<-T0' wb<-T2'
    f32x4 wa[PDEC_KB], wb[PDEC_KB], wc[PDEC_KB];
    unsigned tag = 0;
    {
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // gate products of the NEXT timestep that do not depend on the word: acc1n = W_hh h1 (S2), acc1 = acc1n + W_ih[:, h2] h2
    // and acc2 = h2h h2 (S1', while the fc triples travel).  t = 0: h1 = h2 = 0.
    f32x4 acc1 = zero4, acc2 = zero4, acc1n = zero4;
    for (int t = 0; t < P.max_len; ++t) {
        // ================= S1: attention_lstm cell (h1)
        PD_STAMP(0);
        float tg[4] = {0.f, 0.f, 0.f, 0.f}, ttc = 0.f, tcg = 0.f;
        int bt = B;                                              // teacher-forced: rows whose caption is still running (sorted batch)
        if (P.caps) {
            bt = 0;
            for (int b = 0; b < B; ++b) bt += P.dlen[b] > t ? 1 : 0;
            if (bt == 0) break;
        }
        if (pair) {
            long long tok = P.caps ? P.caps[(long long)pb * P.caps_stride + t] : sTok[pb];
            tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
            const float* trow = P.tok_table + tok * P.ld_tab + pd;
#pragma unroll
            for (int q = 0; q < 4; ++q) tg[q] = trow[(long long)q * D];
            ttc = trow[4LL * D];
            tcg = trow[5LL * D];
        }
        if (worker) {
#pragma unroll
            for (int e = 0; e < 4; ++e) sRed[(kq * 3 + 0) * 256 + (4 * g + e) * 16 + r] = acc1[e];
        }
        if (BEAM && pair) { sCst[pb * 4 + pu] = c1; sCst[PW_BEAM_K * 4 + pb * 4 + pu] = c2; }
        PW_SYNC();
        // beam mode: slot pb continues hypothesis sPar[pb] of the previous timestep — its cell states and the gate products
        // that were contracted before the pick (S2 / S1' of the previous timestep) are read through the parent map
        const int par = BEAM ? sPar[pair ? pb : 0] : pb;
        if (BEAM && pair) { c1 = sCst[par * 4 + pu]; c2 = sCst[PW_BEAM_K * 4 + par * 4 + pu]; }
        ++tag;                                                   // X1: h1
        if (pair && !(P.test_stall && wg == 0)) {
            float gq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int o = par * 16 + q * 4 + pu;
                gq[q] = ((((sRed[o] + sRed[3 * 256 + o]) + sRed[6 * 256 + o]) + sRed[9 * 256 + o]) + pre[q]) + tg[q];
            }
            const float ai = pd_sigm(gq[0]), af = pd_sigm(gq[1]), ag = tanhf(gq[2]), ao = pd_sigm(gq[3]);
            c1 = af * c1 + ai * ag;
            ll_put(h1rs, pb * D + pd, ao * tanhf(c1), tag);
        }
        if (worker) {
            pd_load(wb, pT3);
            pd_load_if(wa, pT4, v4);
            pd_load(wc, pT1);
        }
        PD_STAMP(1);
        PW_STAGE(h1rs, sX, B, D, LDH);
        PW_SYNC();
        // ================= S2: copy_lstm.x2h[:, :D] h1; context_gate / tc_affine rows of the owned columns, 4 projection rows;
        // attention_lstm.W_hh h1 for the next timestep (h1's last reader: the buffer is free after this)
        PD_STAMP(2);
        if (worker) {
            pd_mma(acc2, wb, aX);
            f32x4 accm = zero4;
            pd_mma(accm, wa, aX);
            acc1n = zero4;
            pd_mma(acc1n, wc, aX);
#pragma unroll
            for (int e = 0; e < 4; ++e) sRed[(kq * 3 + 1) * 256 + (4 * g + e) * 16 + r] = accm[e];
        }
        PW_SYNC();
        PD_STAMP(3);
        ++tag;                                                   // X2: [cap_decoder_att(h1) | decoder_att(h1)]
        if (gcol && crr < 12) {
            const int o = cb * 16 + crr;
            const float v = ((sRed[1 * 256 + o] + sRed[4 * 256 + o]) + sRed[7 * 256 + o]) + sRed[10 * 256 + o];
            if (crr < 8) sM[cb * 8 + crr] = v;
            else ll_put(a2rs, cb * 2 * A + wg * 4 + crr - 8, v, tag);
        }
        // ================= S3a: this wave's caption score (editnet.py:370-374) and visual score (:443-445).  The two projection
        // rows it needs (8 floats per lane each) are polled straight into registers — nothing of this wave is in flight ahead
        // of the poll — and only then are the next weight tiles requested: they stream under X3a / X3b
        f32x4 pc0 = zero4, pc1 = zero4, pv0 = zero4, pv1 = zero4;
        if (cs_on || vs_on) {
            const int wc_ = cs_b * 2 * A, wv_ = vs_b * 2 * A + A;
            unsigned spins = 0;
            unsigned long long t0 = 0;
            gb_u32x4 q[8];
            for (;;) {
                asm volatile("" ::: "memory");
                q[0] = ll_req2(a2rs, wc_ + a_lo); q[1] = ll_req2(a2rs, wc_ + a_lo + 2);
                q[2] = ll_req2(a2rs, wc_ + a_hi); q[3] = ll_req2(a2rs, wc_ + a_hi + 2);
                q[4] = ll_req2(a2rs, wv_ + a_lo); q[5] = ll_req2(a2rs, wv_ + a_lo + 2);
                q[6] = ll_req2(a2rs, wv_ + a_hi); q[7] = ll_req2(a2rs, wv_ + a_hi + 2);
                bool ok = true;
#pragma unroll
                for (int i = 0; i < 8; ++i) ok = ok && ll_ok2(q[i], tag);
                if (__all(ok) || ll_giveup(spins, t0, watch)) break;
            }
            pc0 = (f32x4){__uint_as_float(q[0].x), __uint_as_float(q[0].z), __uint_as_float(q[1].x), __uint_as_float(q[1].z)};
            pc1 = (f32x4){__uint_as_float(q[2].x), __uint_as_float(q[2].z), __uint_as_float(q[3].x), __uint_as_float(q[3].z)};
            pv0 = (f32x4){__uint_as_float(q[4].x), __uint_as_float(q[4].z), __uint_as_float(q[5].x), __uint_as_float(q[5].z)};
            pv1 = (f32x4){__uint_as_float(q[6].x), __uint_as_float(q[6].z), __uint_as_float(q[7].x), __uint_as_float(q[7].z)};
        }
        pd_load(wb, pT5);                                        // S4's tile: one tile ahead of the (small) score exchange
        PD_STAMP(4);
        float cs_val = 0.f, vs_val = 0.f;
        if (cs_on) {
            const f32x4 x0 = ca1_0 + (pc0 + *reinterpret_cast<const f32x4*>(sCon + a_lo));
            const f32x4 x1 = ca1_1 + (pc1 + *reinterpret_cast<const f32x4*>(sCon + a_hi));
            const f32x4 cfw0 = *reinterpret_cast<const f32x4*>(sCon + A + a_lo), cfw1 = *reinterpret_cast<const f32x4*>(sCon + A + a_hi);
            const float sc = pw_wsum(pd_score8(x0, x1, cfw0, cfw1));
            cs_val = (cs_mask == 0.f) ? -1e10f : (sc + cbf);
        }
        if (vs_on) {
            const f32x4 x0 = va1_0 + (pv0 + *reinterpret_cast<const f32x4*>(sCon + 2 * A + a_lo));
            const f32x4 x1 = va1_1 + (pv1 + *reinterpret_cast<const f32x4*>(sCon + 2 * A + a_hi));
            const f32x4 vfw0 = *reinterpret_cast<const f32x4*>(sCon + 3 * A + a_lo), vfw1 = *reinterpret_cast<const f32x4*>(sCon + 3 * A + a_hi);
            float sc = vfw0[0] * fmaxf(x0[0], 0.f) + vfw0[1] * fmaxf(x0[1], 0.f) + vfw0[2] * fmaxf(x0[2], 0.f) + vfw0[3] * fmaxf(x0[3], 0.f);
            sc += vfw1[0] * fmaxf(x1[0], 0.f) + vfw1[1] * fmaxf(x1[1], 0.f) + vfw1[2] * fmaxf(x1[2], 0.f) + vfw1[3] * fmaxf(x1[3], 0.f);
            vs_val = pw_wsum(sc) + vbf;
        }
        ++tag;                                                   // X3a: caption scores + visual scores
        if (cs_on && lane == 0) ll_put(csrs, cs_b * T + cs_t, cs_val, tag);
        if (vs_on && lane == 0) ll_put(vsrs, vs_b * R + vs_r, vs_val, tag);
        PD_STAMP(5);
        PW_STAGE(csrs, sAlc, B, T, PDEC_TMAX);                   // (T and R are even: editnet_persistent_wide_ok)
        PW_STAGE(vsrs, sAlv, B, R, 64);
        pd_load_if(wa, pT6, v6);                                 // S5's (short) tile and fc's first one stream under the softmaxes
        pd_load_if(wc, pF[0], vF[0]);                            // and the attend_cap exchange
        PW_SYNC();
        PD_STAMP(6);
        // ================= S3b: both softmaxes of every row, SelectC's arg-max (editnet.py:375-376, :409-416, :446)
        if (worker) {
            for (int b = kq; b < B; b += 4) {
                {
                    const float sc = lane < T ? sAlc[b * PDEC_TMAX + lane] : -INFINITY;
                    const float m = pw_wmax(sc);
                    const float ex = lane < T ? expf(sc - m) : 0.f;
                    const float sum = pw_wsum(ex);
                    const float al = ex / sum;
                    if (lane < T) sAlc[b * PDEC_TMAX + lane] = al;
                    // first arg-max of the weights (block_softmax in attention.hip): SelectC's hard choice
                    float best = lane < T ? al : -1.f;
                    int bi = lane < T ? lane : 0x7fffffff;
                    if (!(best > -1.f)) bi = 0x7fffffff;         // a NaN weight never wins a comparison
                    pw_wargmax(best, bi);
                    const int js = bi == 0x7fffffff ? 0 : bi;
                    const float aj = pw_lane(al, js);
                    if (lane == 0) { sJs[b] = js; sWj[b] = aj * 1.f + (1.f - aj); }   // the reference's fp32 expression (editnet.py:417-418)
                }
                {
                    const float sc = lane < R ? sAlv[b * 64 + lane] : -INFINITY;
                    const float m = pw_wmax(sc);
                    const float ex = lane < R ? expf(sc - m) : 0.f;
                    const float sum = pw_wsum(ex);
                    if (lane < R) sAlv[b * 64 + lane] = ex / sum;
                }
            }
        }
        PW_SYNC();
        if (zrole) {
            float s = 0.f;
            for (int tt = 0; tt < T; ++tt) s += sAlc[zb * PDEC_TMAX + tt] * sPz[tid * PW_TS + tt];
            sZ[zb * 8 + zc8] = s;
        }
        PW_SYNC();
        PD_STAMP(7);
        ++tag;                                                   // X3b: attend_cap columns
        float selv = 0.f, cmemv = 0.f;
        if (pair) {
            // context gate of column pd (editnet.py:378-380; operand order as caption_attention_body in attention.hip)
            const float z = ((sM[pb * 8 + pu] + tcg) + sZ[pb * 8 + pu]) + bg;
            const float zt = pd_sigm(z);
            const float o = zt * tanhf(sZ[pb * 8 + 4 + pu] + bsc) + (1.f - zt) * tanhf((sM[pb * 8 + 4 + pu] + ttc) + btc);
            ll_put(gtrs, pb * D + pd, o, tag);
            const int js = sJs[pb];
            const float wj = sWj[pb];
            selv = P.Mem[((long long)pb * T + js) * D + pd] * wj;
            cmemv = P.memQ[((long long)pb * T + js) * D + pd] * wj;
        }
        PW_STAGE(gtrs, sX, B, D, LDH);
        PW_SYNC();
        PD_STAMP(8);
        // ================= S4: copy_lstm.x2h[:, D:2D] attend_cap; hoisted region products -> c_new
        if (worker) {
            pd_mma(acc2, wb, aX);
            pd_load_if(wb, pF[1], vF[1]);
#pragma unroll
            for (int e = 0; e < 4; ++e) sRed[(kq * 3 + 2) * 256 + (4 * g + e) * 16 + r] = acc2[e];
        }
        PW_SYNC();
        if (gcol) {
            const int o = cb * 16 + crr;
            float s = 0.f;
            for (int rr = 0; rr < R; ++rr) s += sAlv[cb * 64 + rr] * sPv[tid * PW_RS + rr];
            float g2 = ((sRed[2 * 256 + o] + sRed[5 * 256 + o]) + sRed[8 * 256 + o]) + sRed[11 * 256 + o];
            if (BEAM) {                                          // + copy_lstm.h2h h2 of the PARENT hypothesis (S1' of the previous timestep)
                const int op = sPar[cb] * 16 + crr;
                g2 += ((sRedP[op] + sRedP[256 + op]) + sRedP[512 + op]) + sRedP[768 + op];
            }
            sG[o] = (g2 + s) + b2;
        }
        PW_SYNC();
        PD_STAMP(9);
        ++tag;                                                   // X4: c_new
        float cnv = 0.f, ogv = 0.f;
        if (pair) {
            const float* gp = sG + pb * 16 + pu;
            const float ai = pd_sigm(gp[0]), af = pd_sigm(gp[4]), ag = tanhf(gp[8]);
            ogv = pd_sigm(gp[12]);
            cnv = af * c2 + ai * ag;
            ll_put(cnrs, pb * D + pd, cnv, tag);
        }
        PW_STAGE(cnrs, sX, B, D, LDH);
        PW_SYNC();
        PD_STAMP(10);
        // ================= S5: gate_cnew rows of the owned units, copy gate (editnet.py:281-283) -> c2, h2
        if (worker) {
            f32x4 acc5 = zero4;
            pd_mma(acc5, wa, aX);
            pd_load_if(wa, pF[2], vF[2]);
#pragma unroll
            for (int e = 0; e < 4; ++e) sRed[(kq * 3 + 0) * 256 + (4 * g + e) * 16 + r] = acc5[e];
        }
        PW_SYNC();
        ++tag;                                                   // X5: h2
        if (pair) {
            const int o = pb * 16 + pu;
            const float a = (((sRed[o] + sRed[3 * 256 + o]) + sRed[6 * 256 + o]) + sRed[9 * 256 + o]) + bcn;
            const float bq = cmemv + bcm;
            const float cg = pd_sigm(a + bq);
            c2 = cg * selv + (1.f - cg) * cnv;
            ll_put(h2rs, pb * D + pd, ogv * tanhf(c2), tag);
        }
        PD_STAMP(11);
        PW_STAGE(h2rs, sX, B, D, LDH);
        PW_SYNC();
        PD_STAMP(12);
        // ================= S6: fc over this workgroup's vocabulary rows, local (max, first arg-max, sum exp) per batch row
        const bool more = t + 1 < P.max_len;
        if (worker) {
            f32x4 accf0 = zero4, accf1 = zero4, accf2 = zero4;
            pd_mma(accf0, wc, aX);
            if (more) pd_load(wc, pT0);
            pd_mma(accf1, wb, aX);
            if (more) pd_load(wb, pT2);
            pd_mma(accf2, wa, aX);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sRed[(kq * 3 + 0) * 256 + (4 * g + e) * 16 + r] = accf0[e];
                sRed[(kq * 3 + 1) * 256 + (4 * g + e) * 16 + r] = accf1[e];
                sRed[(kq * 3 + 2) * 256 + (4 * g + e) * 16 + r] = accf2[e];
            }
        }
        PW_SYNC();
        PD_STAMP(13);
        if (P.caps) {
            // teacher-forced: the scores themselves, rows 0 .. bt - 1 (editnet.py:546: predictions[:batch_size_t, t, :] = preds)
            if (worker) {
                for (int b = kq; b < bt; b += 4) {
                    const int j = lane >> 4, rr = lane & 15, row = row0 + lane;
                    if (lane < 16 * PDEC_FC_TILES && lane < P.rpw && row < V) {
                        const int o = j * 256 + b * 16 + rr;
                        P.predictions[(long long)b * P.ld_pred_b + (long long)t * V + row] =
                            (((sRed[o] + sRed[3 * 256 + o]) + sRed[6 * 256 + o]) + sRed[9 * 256 + o]) + fcb_lane;
                    }
                }
                if (more) {                                      // S1' (see below)
                    acc1 = acc1n;
                    pd_mma(acc1, wc, aX);
                    acc2 = zero4;
                    pd_mma(acc2, wb, aX);
                }
            }
            PW_SYNC();                                           // sRed is rewritten by the next timestep's S1
            continue;
        }
        if constexpr (BEAM) {
            // ================= beam mode (editnet.py:654-699; the bookkeeping of csrc/beam.hip beam_pick_k for ONE image):
            // X6 carries, per row and vocabulary slice, (max, sum exp) and the slice's B best (score, word) pairs — a global
            // top-B over B x V candidates takes at most B from one slice
            ++tag;
            if (kq < B) {
                const int b = kq;
                const int j = lane >> 4, rr = lane & 15, row = row0 + lane;
                const bool ok = lane < 16 * PDEC_FC_TILES && lane < P.rpw && row < V;
                float x = -INFINITY;
                if (ok) {
                    const int o = j * 256 + b * 16 + rr;
                    x = (((sRed[o] + sRed[3 * 256 + o]) + sRed[6 * 256 + o]) + sRed[9 * 256 + o]) + fcb_lane;
                }
                float cvv[PW_BEAM_K];
                int cii[PW_BEAM_K];
                float xx = x;
#pragma unroll
                for (int q = 0; q < PW_BEAM_K; ++q) {
                    float bv = -INFINITY;
                    int bix = 0x7fffffff;
                    if (q < B) {
                        if (xx > -INFINITY) { bv = xx; bix = row; }
                        pw_wargmax(bv, bix);
                        if (ok && row == bix) xx = -INFINITY;
                    }
                    cvv[q] = bv; cii[q] = bix;
                }
                const float mx = cvv[0];
                float se = (ok && mx > -INFINITY) ? expf(x - mx) : 0.f;
                se = pw_wsum(se);
                if (lane < PW_BEAM_W) {
                    float v = 0.f;
                    if (lane == 0) v = mx;
                    else if (lane == 1) v = se;
                    else if (lane < 2 + 2 * PW_BEAM_K) {
                        const int q = (lane - 2) >> 1;
                        float cv_ = cvv[0]; int ci_ = cii[0];
#pragma unroll
                        for (int u = 1; u < PW_BEAM_K; ++u) if (q == u) { cv_ = cvv[u]; ci_ = cii[u]; }
                        v = (lane & 1) ? __int_as_float(ci_) : cv_;
                    }
                    ll_put(fbrs, (b * G + wg) * PW_BEAM_W + lane, v, tag);
                }
            }
            // S1' (see the greedy path); copy_lstm.h2h h2 goes to LDS: the next timestep adds it through the parent map
            if (more) {
                acc1 = acc1n;
                pd_mma(acc1, wc, aX);
                f32x4 accp = zero4;
                pd_mma(accp, wb, aX);
#pragma unroll
                for (int e = 0; e < 4; ++e) sRedP[kq * 256 + (4 * g + e) * 16 + r] = accp[e];
                acc2 = zero4;
            }
            PW_SYNC();
            PW_STAGE(fbrs, sFB, B * G, PW_BEAM_W, PW_BEAM_W);
            PW_SYNC();
            // ---- every workgroup runs the same pick.  Wave j: log-sum-exp of row j and its B best candidates
            if (kq < B) {
                const int j = kq;
                const float scj = sScore[j];
                float ov[PW_BEAM_K];
                int oi[PW_BEAM_K];
#pragma unroll
                for (int q = 0; q < PW_BEAM_K; ++q) { ov[q] = -INFINITY; oi[q] = 0x7fffffff; }
                if (scj > -INFINITY) {                           // (uniform in the wave; dead slots take no part)
                    float cv[16];
                    int ci[16];
                    float pm[4], ps[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float* e = sFB + ((j * G + lane + 64 * i) * PW_BEAM_W);
                        pm[i] = e[0]; ps[i] = e[1];
#pragma unroll
                        for (int q = 0; q < 4; ++q) { cv[4 * i + q] = e[2 + 2 * q]; ci[4 * i + q] = __float_as_int(e[3 + 2 * q]); }
                    }
                    const float m = pw_wmax(fmaxf(fmaxf(pm[0], pm[1]), fmaxf(pm[2], pm[3])));
                    float ssum = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) ssum += (pm[i] == -INFINITY) ? 0.f : ps[i] * expf(pm[i] - m);
                    ssum = pw_wsum(ssum);
                    const float lse = m + logf(ssum);
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        const bool have = ci[c] != 0x7fffffff;
                        cv[c] = have ? scj + (cv[c] - lse) : -INFINITY;       // score + log-prob, as beam_pick_k forms it
                        ci[c] = have ? j * V + ci[c] : 0x7fffffff;
                    }
#pragma unroll
                    for (int q = 0; q < PW_BEAM_K; ++q) {
                        if (q < B) {
                            float bv = -INFINITY;
                            int bix = 0x7fffffff;
#pragma unroll
                            for (int c = 0; c < 16; ++c)
                                if (cv[c] > bv || (cv[c] == bv && ci[c] < bix)) { bv = cv[c]; bix = ci[c]; }
                            if (!(bv > -INFINITY)) bix = 0x7fffffff;
                            pw_wargmax(bv, bix);
#pragma unroll
                            for (int c = 0; c < 16; ++c) if (ci[c] == bix) cv[c] = -INFINITY;
                            ov[q] = bv; oi[q] = bix;
                        }
                    }
                }
                if (lane == 0) {
#pragma unroll
                    for (int q = 0; q < PW_BEAM_K; ++q) { sCand[(j * PW_BEAM_K + q) * 2] = ov[q]; sCand[(j * PW_BEAM_K + q) * 2 + 1] = __int_as_float(oi[q]); }
                }
            }
            PW_SYNC();
            if (tid == 0) {
                // the B best of the B x B candidates (ties: lowest flat index), then beam_pick_k's bookkeeping
                const int k = B, kl = sKleft;
                float pv_[PW_BEAM_K];
                int pi_[PW_BEAM_K];
                unsigned taken = 0u;
                for (int rr_ = 0; rr_ < k; ++rr_) {
                    float bv = -INFINITY;
                    int bix = 0x7fffffff, bc = -1;
                    for (int c = 0; c < k * PW_BEAM_K; ++c) {
                        if ((taken >> c) & 1u) continue;
                        if ((c % PW_BEAM_K) >= k) continue;
                        const float v = sCand[c * 2];
                        const int ix = __float_as_int(sCand[c * 2 + 1]);
                        if (ix == 0x7fffffff) continue;
                        if (v > bv || (v == bv && ix < bix) || bc < 0) { bv = v; bix = ix; bc = c; }
                    }
                    if (bc >= 0) taken |= 1u << bc;
                    pv_[rr_] = bc >= 0 ? bv : -INFINITY;
                    pi_[rr_] = bc >= 0 ? bix : 0x7fffffff;
                }
                int n_end = 0, c_arg = -1, slot = 0;
                float c_best = -INFINITY;
                bool live[PW_BEAM_K];
                for (int rr_ = 0; rr_ < k; ++rr_) {
                    const int flat = pi_[rr_];
                    const bool okp = flat != 0x7fffffff && rr_ < kl;          // only the first k_left picks count
                    const long long word = okp ? flat % V : 0;
                    const bool is_end = okp && word == P.end_idx;
                    live[rr_] = okp && !is_end;
                    if (is_end) {
                        ++n_end;
                        if (pv_[rr_] > c_best) { c_best = pv_[rr_]; c_arg = rr_; }   // first maximum
                    }
                }
                if (c_arg >= 0 && c_best > sBest) {
                    sBest = c_best;
                    if (wg == 0) {
                        P.bm_best_score[0] = c_best;
                        P.bm_best_word[0] = pi_[c_arg] % V;
                        P.bm_result[0] = t;                              // pick index of the best completed hypothesis
                        P.bm_result[1] = pi_[c_arg] / V;                 // its parent slot (numbering before this pick)
                    }
                }
                sKleft = kl - n_end;
                for (int pass = 0; pass < 2; ++pass)
                    for (int rr_ = 0; rr_ < k; ++rr_) {
                        if ((pass == 0) != live[rr_]) continue;
                        const int flat = pi_[rr_];
                        const int parent = flat != 0x7fffffff ? flat / V : 0;
                        const long long word = flat != 0x7fffffff ? flat % V : 0;
                        sScore[slot] = live[rr_] ? pv_[rr_] : -INFINITY;
                        sTok[slot] = live[rr_] ? word : 0;
                        sPar[slot] = parent;
                        if (wg == 0) {
                            P.bm_hist_par[t * PW_BEAM_K + slot] = parent;
                            P.bm_hist_word[t * PW_BEAM_K + slot] = word;
                        }
                        ++slot;
                    }
                if (wg == 0) { P.bm_result[2] = sKleft; P.bm_result[3] = t + 1; }
            }
            PW_SYNC();
            PD_STAMP(16);
            if (sKleft == 0) break;                              // every hypothesis has ended (editnet.py:700-701)
            continue;
        }
        ++tag;                                                   // X6: triples
        if (worker) {
            for (int b = kq; b < B; b += 4) {
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
            }
            // ================= S1': attention_lstm.W_ih[:, h2] h2 (+ W_hh h1 from S2) and copy_lstm.h2h h2 for timestep t + 1,
            // while the triples travel: nothing here waits for the word
            if (more) {
                acc1 = acc1n;
                pd_mma(acc1, wc, aX);
                acc2 = zero4;
                pd_mma(acc2, wb, aX);
            }
        }
        PW_SYNC();                                               // (h2's last readers are done: the buffer takes the triples)
        PD_STAMP(14);
        PW_STAGE(fcrs, sX, B * G, 4, 4);
        PW_SYNC();
        PD_STAMP(15);
        // ================= S7: every workgroup combines the G triples of every row: same word everywhere
        if (worker) {
            for (int b = kq; b < B; b += 4) {
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
            }
        }
        PW_SYNC();
        PD_STAMP(16);
        int alive = 0;
        for (int b = 0; b < B; ++b) alive += sUnf[b];
        if (wg == 0 && tid == 0) P.alive[t] = alive;
        if (alive == 0) break;                                    // editnet_rl.py:546: every caption has ended
    }
    }
#undef PW_STAGE
#undef PW_SYNC
    __shared__ unsigned s_bad;
    if (tid == 0) s_bad = __hip_atomic_load(P.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_bad && worker) {                                        // an exchange timed out: never hand this out as a decode
        const float qnan = __builtin_nanf("");
        if (P.caps) {                                              // teacher-forced: every score this workgroup wrote (see decode_persistent.hip)
            const int row0_ = wg * P.rpw;
            for (int i = tid; i < B * P.max_len * P.rpw; i += PDEC_THREADS) {
                const int row = row0_ + i % P.rpw, bt_ = i / P.rpw;
                if (row < V) P.predictions[(long long)(bt_ / P.max_len) * P.ld_pred_b + (long long)(bt_ % P.max_len) * V + row] = qnan;
            }
        } else if (BEAM) {
            if (wg == 0 && tid == 0) { P.bm_best_score[0] = qnan; P.bm_result[2] = -1; P.bm_result[3] = -1; }   // never a search result
        } else if (wg == 0) {
            for (int i = tid; i < B * P.max_len; i += PDEC_THREADS) { P.seq_logp[i] = qnan; P.seq[i] = 0; }
        }
    }
}

static int pwide_lds_floats(int B, int D, int A, bool beam = false) {
    const int base = B * (D + 4) + 4 * 3 * 256 + PDW_MAXB * (PDEC_TMAX + 64 + 16 + 8 + 8) + B * 16 * PW_RS + B * 8 * PW_TS + 4 * A;
    return base + (beam ? B * (D / 4) * PW_BEAM_W + 4 * 256 + 2 * PW_BEAM_K * 4 + PW_BEAM_K * PW_BEAM_K * 2 : 0);
}

// [status line | h1 | attend_cap | c_new | h2 | projections | caption scores | visual scores | fc triples] as flag-in-data words
size_t editnet_persistent_wide_xbytes(int B, int D, int A) {
    if (B > PDW_MAXB) return 0;
    return 128 + (size_t)B * D * 8 * 4 + (size_t)B * 2 * A * 8 + (size_t)B * PDEC_TMAX * 8 + (size_t)B * 64 * 8 + (size_t)B * (D / 4) * 32 +
           (B <= PW_BEAM_K ? (size_t)B * (D / 4) * PW_BEAM_W * 8 : 0);       // + the beam mode's candidate words
}

bool editnet_persistent_wide_ok(int B, int D, int A, int T, int R, int V) {
    if (B < 1 || B > PDW_MAXB) return false;
    if (D != 64 * PDEC_KB || A != 512 || 2 * A != D || T > PDEC_TMAX || R > PDEC_RREG || R > 64 || (R & 1) || (T & 1)) return false;
    const int G = D / 4;
    if ((V + G - 1) / G > 16 * PDEC_FC_TILES) return false;
    if (B * R > 4 * G || B * T > 4 * G) return false;               // one score of each kind per wave
    if (4 * G != 2 * A) return false;                                // the triples (B, G, 4) take the projections' place
    const int lds = pwide_lds_floats(B, D, A) * (int)sizeof(float);
    if (lds > 156 * 1024 || lds + 4096 > persistent_lds_limit()) return false;
    return true;
}

static int g_pwide_capacity[2][64] = {};
static int g_pwide_capacity_lds[2][64] = {};

int editnet_persistent_wide_launch(PDecEditArgs& P, void* xbuf, PersistentGuard& guard, hipStream_t s, bool* unsupported, bool beam) {
    *unsupported = true;
    const int B = P.B, D = P.D, A = P.A, G = D / 4;
    if (!editnet_persistent_wide_ok(B, D, A, P.T, P.R, P.V)) return SET_OK;
    if (beam && (B > PW_BEAM_K || P.caps || (long long)B * P.V >= 0x7fffffffLL)) return SET_OK;
    {
        char* x = (char*)xbuf;
        P.status = (unsigned*)x; x += 128;
        P.x_h1 = x; x += (size_t)B * D * 8;
        P.x_gt = x; x += (size_t)B * D * 8;
        P.x_cn = x; x += (size_t)B * D * 8;
        P.x_h2 = x; x += (size_t)B * D * 8;
        P.x_a2 = x; x += (size_t)B * 2 * A * 8;
        P.x_cs = x; x += (size_t)B * PDEC_TMAX * 8;
        P.x_vs = x; x += (size_t)B * 64 * 8;
        P.x_fc = x; x += (size_t)B * G * 32;
        P.x_fcb = x;
    }
    const void* kern = beam ? reinterpret_cast<const void*>(&editnet_persistent_wide_k<true>)
                            : reinterpret_cast<const void*>(&editnet_persistent_wide_k<false>);
    const int lds = pwide_lds_floats(B, D, A, beam) * (int)sizeof(float);
    static bool configured[2][64] = {};
    int lds_max = pwide_lds_floats(beam ? PW_BEAM_K : PDW_MAXB, D, A, beam) * (int)sizeof(float);
    if (lds_max > 156 * 1024) lds_max = 156 * 1024;
    if (lds > lds_max || guard.set_lds(kern, lds_max, configured[beam ? 1 : 0]) != SET_OK) return SET_OK;
    int& cap = g_pwide_capacity[beam ? 1 : 0][guard.dev];
    int& cap_lds = g_pwide_capacity_lds[beam ? 1 : 0][guard.dev];
    if (cap == 0 || lds > cap_lds) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, PDEC_THREADS, (size_t)lds) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, guard.dev) != hipSuccess) {
            (void)hipGetLastError();
            return SET_OK;
        }
        cap = per_cu * cus;
        cap_lds = lds;
        if (cap <= 0) cap = -1;
        const int forced = env_int("SET_PENC_TEST_CAPACITY", 0);
        if (forced > 0) cap = forced;
    }
    if (G > cap) return SET_OK;
    *unsupported = false;
    const double wbytes = 4.0 * ((double)P.V * D + 5.0 * 4 * D * D + 3.0 * D * D + 2.0 * A * D);
    ProfScope ps(beam ? "persistent_beam" : "persistent_decode", s, 2.0 * B * wbytes / 4.0 * P.max_len, wbytes * P.max_len);
    SET_TRY(guard.serialise(s));
    SET_HIP_TRY(hipMemsetAsync(xbuf, 0, editnet_persistent_wide_xbytes(B, D, A), s));    // no word of an earlier decode may carry a tag of this one
    SET_TRY(pd_stamps_begin(&P.stamps, &P.stamp_wg, s));
    if (beam) hipLaunchKernelGGL(editnet_persistent_wide_k<true>, dim3(G), dim3(PDEC_THREADS), lds, s, P);
    else hipLaunchKernelGGL(editnet_persistent_wide_k<false>, dim3(G), dim3(PDEC_THREADS), lds, s, P);
    SET_LAUNCH_CHECK();
    SET_TRY(guard.launched(s));
    SET_TRY(pd_stamps_report(P.stamps, P.stamp_wg, 16, P.max_len, s));
    return SET_OK;
}


// ---------------------------------------------------------------------------------------------
// Host side of EditNet's persistent greedy / teacher-forced / beam decode: the argument block of one launch and the dispatch to
// the kernel above, which since round 5 serves every batch of 1 .. 16 rows (it measured faster than the <= 8-row kernel of round
// 4 at every row count — B = 4: 1.24 vs 1.28 ms, B = 8: 1.41 vs 1.67 ms — so that kernel was removed).
// ---------------------------------------------------------------------------------------------
// exchange region of one decode (laid out by decode_persistent_wide.hip)
size_t editnet_persistent_xbytes(int B, int D, int A) { return B > PDW_MAXB ? 0 : editnet_persistent_wide_xbytes(B, D, A); }

bool editnet_persistent_ok(const SetEditNetDims* d, int max_len) {
    const int on = env_int("SET_DEC_PERSISTENT", 1);                 // (read per call: tests and A/B runs flip it inside one process)
    const int maxb = env_int("SET_DEC_PERSISTENT_MAXB", PDW_MAXB);
    if (!on || d->B > maxb || max_len < 1 || d->adaptive) return false;
    return editnet_persistent_wide_ok(d->B, d->D, d->A, d->T, d->R, d->V) && !persistent_disabled();
}

// the greedy loop after the prologue (set_editnet_begin).  pv = X x2h[:, 2D:]^T (B, R, 4D), xbuf = exchange region.
// SET_ERR_UNSUPPORTED: nothing was touched, the caller runs the per-step loop.
int editnet_persistent_greedy(const SetEditNetWeights* w, const SetEditNetDims* d, const float* pre1, const float* att1,
                              const float* att1_c, const float* mask, const float* capP, const float* memQ, const float* Mem,
                              const float* pv, void* xbuf, long long* it, int* unfinished, int* alive, long long start_idx,
                              long long end_idx, int max_len, long long* seq, float* seq_logp, hipStream_t s,
                              const PDecTeacher* teach, const PDecBeam* beam) {
    if (beam) {         // beam mode lives in the wide variant, whatever the row count
        if (!env_int("SET_DEC_PERSISTENT", 1) || max_len < 1 || d->adaptive || persistent_disabled() ||
            !editnet_persistent_wide_ok(d->B, d->D, d->A, d->T, d->R, d->V)) return SET_ERR_UNSUPPORTED;
    } else if (!editnet_persistent_ok(d, max_len)) return SET_ERR_UNSUPPORTED;
    const int B = d->B, D = d->D, A = d->A, F = d->F, G = D / 4;
    PDecEditArgs P{};
    P.al_wih = w->al_wih; P.ld_ih = 3LL * D + F; P.al_whh = w->al_whh; P.cl_h2h_w = w->cl_h2h_w;
    P.cl_x2h_w = w->cl_x2h_w; P.ld_x2h = 2LL * D + F; P.cl_x2h_b = w->cl_x2h_b; P.cl_h2h_b = w->cl_h2h_b;
    P.ca_gate_w = w->ca_gate_w; P.ca_gate_b = w->ca_gate_b; P.ca_tc_w = w->ca_tc_w; P.ca_tc_b = w->ca_tc_b; P.ca_sc_b = w->ca_sc_b;
    P.ca_dec_w = w->ca_dec_w; P.ca_dec_b = w->ca_dec_b; P.ca_full_w = w->ca_full_w; P.ca_full_b = w->ca_full_b;
    P.va_dec_w = w->va_dec_w; P.va_dec_b = w->va_dec_b; P.va_full_w = w->va_full_w; P.va_full_b = w->va_full_b;
    P.cl_cnew_w = w->cl_cnew_w; P.cl_cnew_b = w->cl_cnew_b; P.cl_cmem_b = w->cl_cmem_b;
    P.fc_w = w->fc_w; P.fc_b = w->fc_b; P.tok_table = w->tok_table; P.ld_tab = 10LL * D;
    P.pre1 = pre1; P.att1 = att1; P.att1_c = att1_c; P.mask = mask; P.capP = capP; P.memQ = memQ; P.Mem = Mem; P.pv = pv;
    P.it = it; P.unfinished = unfinished; P.alive = alive; P.seq = seq; P.seq_logp = seq_logp;
    P.B = B; P.D = D; P.T = d->T; P.R = d->R; P.A = A; P.V = d->V; P.max_len = max_len; P.rpw = (d->V + G - 1) / G;
    P.start_idx = start_idx; P.end_idx = end_idx;
    if (teach) {
        P.caps = (const long long*)teach->caps; P.caps_stride = teach->caps_stride;
        P.predictions = teach->predictions; P.ld_pred_b = (long long)max_len * d->V;
        for (int b = 0; b < B; ++b) P.dlen[b] = teach->host_decode_lengths[b];
    }
    PersistentGuard guard;
    if (guard.rc != SET_OK) return guard.rc;
    P.spin_limit = guard.spin_limit();
    P.test_stall = guard.test_stall(); P.fault = guard.fault;
    if (beam) {
        P.bm_hist_par = beam->hist_par; P.bm_hist_word = (long long*)beam->hist_word; P.bm_best_score = beam->best_score;
        P.bm_best_word = (long long*)beam->best_word; P.bm_result = beam->result;
    }
    bool unsupported = true;
    const int rc = editnet_persistent_wide_launch(P, xbuf, guard, s, &unsupported, beam != nullptr);      // (lays out the exchange region)
    return rc != SET_OK ? rc : (unsupported ? SET_ERR_UNSUPPORTED : SET_OK);
}

}  // namespace set
