// Persistent greedy decode of EditNet for small batches (B <= 8; editnet_rl.py:485-549): the whole free-running loop as ONE
// launch of D / 4 workgroups, same construction as the DCNet kernel (decode_persistent.hip): a workgroup OWNS four hidden
// units, streams the weight rows of those units as B operands of v_mfma_f32_16x16x4_f32 and finishes every pointwise
// stage itself; what other workgroups need travels as flag-in-data words (grid_barrier.h).
//
// Per timestep (eval mode, token table):
//   S1  attention_lstm cell from the gate products [W_ih[:, h2] | W_hh] (and copy_lstm.h2h), which were contracted at the
//       END of the previous timestep (S1', while its fc triples travelled: they do not depend on the word) -> h1  X1 (B, D)
//   S2  copy_lstm.x2h[:, :D] h1, and one mixed tile: 4 rows of context_gate.W[:, D:2D], 4 rows of tc_affine.W[:, D:],
//       4 rows of [cap_decoder_att ; decoder_att] (2A = 4 x 256 rows)                     -> projections   X2 (B, 2A)
//   S3  caption attention of every row in every workgroup (its cap_features_att rows stay in registers at B <= 4) ->
//       alpha_c, first arg-max j*; the context gate of the OWNED four columns from the hoisted P = [W_g H | W_sc H]
//       (loop-invariant, in registers) -> attend_cap columns; sel / gate_cmem(sel) columns from Mem / Q rows j*;
//       ONE visual score e[b, r] per wave (B R <= 4 x 256 scores over the grid)            -> gated, scores X3 (B, D + 64)
//   S4  visual softmax; copy_lstm.x2h[:, D:2D] gated + sum_r alpha_r Pv[b, r] (Pv = X W_x2h[:, 2D:]^T hoisted in the
//       prologue: the 33 MB of x2h's region columns are never streamed, attend_img is never formed) -> c_new  X4 (B, D)
//   S5  gate_cnew rows of the owned units, copy gate                                       -> h2            X5 (B, D)
//   S6  fc rows of this workgroup, (max, first arg-max, sum exp) per batch row             -> triples       X6 (B, G)
//   S1' the next timestep's h-dependent gate products, between publishing the triples and polling them
//   S7  every workgroup combines the G triples: same word everywhere; workgroup 0 writes seq / seq_logp
// 139 MB of weights per timestep instead of 263 MB, six exchanges instead of seven launches.  Request placement, the
// loop-free B <= 4 variant and the teacher-forced mode (set_editnet_xe_forward): as in decode_persistent.hip.
#include "decode_persistent.h"

namespace set {

template <bool RES>
__global__ void __launch_bounds__(PDEC_THREADS, 1) editnet_persistent_k(const PDecEditArgs P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ long long sTok[PDEC_MAXB];
    __shared__ int sUnf[PDEC_MAXB], sJs[PDEC_MAXB];
    __shared__ float sWj[PDEC_MAXB];
    const int tid = threadIdx.x, lane = tid & 63, kq = tid >> 6, r = lane & 15, g = lane >> 4;
    const int B = P.B, D = P.D, T = P.T, R = P.R, A = P.A, V = P.V;
    const int KQ = D >> 2, LDH = D + 4;
    const int wg = (int)blockIdx.x, u0 = wg * 4, G = (int)gridDim.x;
    float* sH2 = smem;                                   // (B, LDH): h2 from its exchange to the end of the timestep (fc, S1') ...
    float* sX = sH2;                                     // ... and before that attend_cap, then c_new (one K operand at a time)
    float* sH1 = sH2 + B * LDH;                          // h1: from its exchange to S1' at the end of the timestep
    float* sRed = sH1 + B * LDH;                         // [4 waves][3 tiles][16][16]
    float* sAlc = sRed + 4 * 3 * 256;                    // (B, TMAX) caption attention weights
    float* sAlv = sAlc + PDEC_MAXB * PDEC_TMAX;          // (B, 64) visual scores -> weights
    float* sG = sAlv + PDEC_MAXB * 64;                   // (B, 16) copy_lstm gate pre-activations
    float* sZ = sG + PDEC_MAXB * 16;                     // (B, 8) [sum alpha P_z (4) | sum alpha P_s (4)] of the owned columns
    float* sM = sZ + PDEC_MAXB * 8;                      // (B, 8) [context_gate.W h1 (4) | tc_affine.W h1 (4)]
    float* sA2 = sM + PDEC_MAXB * 8;                     // (B, 2A) [cap_decoder_att(h1) | decoder_att(h1)], biases not added
    float* sF = sA2;                                     // (B, G, 4) fc triples: same size (2A == 4G), disjoint lifetime (S3 / S7)
    // loop-invariant operands that would otherwise take ~90 registers per thread next to the 160 of the resident attention rows
    float* sPv = sA2 + B * 2 * A;                        // (B, 16, RREG) hoisted region products of the owned gate rows
    float* sPz = sPv + B * 16 * PDEC_RREG;               // (B, 8, TREG|TMAX) hoisted caption-context products of the owned columns
    float* sCon = sPz + B * 8 * PDEC_TMAX;               // [cap_decoder_att.b | cap_full_att.w | decoder_att.b | full_att.w] (4, A)
    const LLWatch watch{P.status, P.fault, P.spin_limit};

    // ---- weight tiles of this lane
    const long long grow = (long long)(r >> 2) * D + u0 + (r & 3);       // gate row of the 4D-row matrices
    const int kcol = kq * KQ + 4 * g;
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
    const float fcb_lane = (lane < 16 * PDEC_FC_TILES && lane < P.rpw && row0 + lane < V) ? P.fc_b[row0 + lane] : 0.f;   // fc.bias of the row lane l scores
    const int arow = (r < B ? r : B - 1) * LDH;
    const float* aX = sX + arow + kcol;
    const float* aH2 = sH2 + arow + kcol;
    const float* aH1 = sH1 + arow + kcol;

    // ---- thread roles and their loop-invariant operands
    const bool pair = tid < B * 4;                       // (batch row, owned unit): the two cells, the context gate, the copy gate
    const int pb = tid >> 2, pu = tid & 3, pd = u0 + pu;
    const bool gcol = tid < B * 16;                      // (batch row, gate row) of copy_lstm's hoisted region products
    const int cb = tid >> 4, crr = tid & 15;
    const long long ccol = (long long)(crr >> 2) * D + u0 + (crr & 3);
    const bool zrole = tid < B * 8;                      // (batch row, [z | s] column) of the hoisted caption-context products
    const int zb = tid >> 3, zc8 = tid & 7;
    float c1 = 0.f, c2 = 0.f, pre[4] = {0.f, 0.f, 0.f, 0.f};
    float bg = 0.f, bsc = 0.f, btc = 0.f, bcn = 0.f, bcm = 0.f, b2 = 0.f;
    if (pair) {
#pragma unroll
        for (int q = 0; q < 4; ++q) pre[q] = P.pre1[(long long)pb * 4 * D + (long long)q * D + pd];
        bg = P.ca_gate_b[pd]; bsc = P.ca_sc_b[pd]; btc = P.ca_tc_b[pd]; bcn = P.cl_cnew_b[pd]; bcm = P.cl_cmem_b[pd];
    }
    if (gcol) {
        for (int rr = 0; rr < PDEC_RREG; ++rr) sPv[tid * PDEC_RREG + rr] = rr < R ? P.pv[((long long)cb * R + rr) * 4 * D + ccol] : 0.f;
        b2 = P.cl_x2h_b[ccol] + P.cl_h2h_b[ccol];
    }
    if (zrole)
        for (int tt = 0; tt < PDEC_TMAX; ++tt)
            sPz[tid * PDEC_TMAX + tt] = tt < T ? P.capP[((long long)zb * T + tt) * 2 * D + (zc8 < 4 ? u0 + zc8 : D + u0 + zc8 - 4)] : 0.f;
    for (int i = tid; i < A; i += PDEC_THREADS) {
        sCon[i] = P.ca_dec_b[i]; sCon[A + i] = P.ca_full_w[i]; sCon[2 * A + i] = P.va_dec_b[i]; sCon[3 * A + i] = P.va_full_w[i];
    }
    const int a_lo = lane * 4, a_hi = lane * 4 + 256;
    const float cbf = P.ca_full_b[0], vbf = P.va_full_b[0];
    // ... the ONE visual score this wave owns: index s = wg + G * wave over (b, r), its features_att row in registers
    const int vs_idx = wg + G * kq;
    const bool vs_on = vs_idx < B * R;
    const int vs_b = vs_on ? vs_idx / R : 0, vs_r = vs_on ? vs_idx % R : 0;
    const f32x4 va1_0 = *reinterpret_cast<const f32x4*>(P.att1 + ((long long)vs_b * R + vs_r) * A + a_lo);
    const f32x4 va1_1 = *reinterpret_cast<const f32x4*>(P.att1 + ((long long)vs_b * R + vs_r) * A + a_hi);
    // ... and (RES) this wave's row of cap_features_att
    f32x4 a1r[RES ? PDEC_TREG : 1][2];
    float mk_res = 1.f;                                  // (RES) ... and its mask word
    if constexpr (RES) {
        const int brow = kq < B ? kq : B - 1;
        if (lane < T) mk_res = P.mask[(long long)brow * T + lane];
#pragma unroll
        for (int tt = 0; tt < PDEC_TREG; ++tt) {
            const int t2 = tt < T ? tt : T - 1;
            a1r[tt][0] = *reinterpret_cast<const f32x4*>(P.att1_c + ((long long)brow * T + t2) * A + a_lo);
            a1r[tt][1] = *reinterpret_cast<const f32x4*>(P.att1_c + ((long long)brow * T + t2) * A + a_hi);
        }
    }
    // exchange buffers
    const __amdgpu_buffer_rsrc_t h1rs = __builtin_amdgcn_make_buffer_rsrc(P.x_h1, 0, B * D * 8, 0x00027000);
    const __amdgpu_buffer_rsrc_t a2rs = __builtin_amdgcn_make_buffer_rsrc(P.x_a2, 0, B * 2 * A * 8, 0x00027000);
    const __amdgpu_buffer_rsrc_t gtrs = __builtin_amdgcn_make_buffer_rsrc(P.x_gt, 0, B * D * 8, 0x00027000);
    const __amdgpu_buffer_rsrc_t vsrs = __builtin_amdgcn_make_buffer_rsrc(P.x_vs, 0, B * 64 * 8, 0x00027000);
    const __amdgpu_buffer_rsrc_t cnrs = __builtin_amdgcn_make_buffer_rsrc(P.x_cn, 0, B * D * 8, 0x00027000);
    const __amdgpu_buffer_rsrc_t h2rs = __builtin_amdgcn_make_buffer_rsrc(P.x_h2, 0, B * D * 8, 0x00027000);
    const __amdgpu_buffer_rsrc_t fcrs = __builtin_amdgcn_make_buffer_rsrc(P.x_fc, 0, B * G * 32, 0x00027000);

    // ---- initial state (init_hidden_state, editnet.py:494-495): zeros; every row is fed <start>
    for (int i = tid; i < 2 * B * LDH; i += PDEC_THREADS) smem[i] = 0.f;
    if (tid < B) { sTok[tid] = P.start_idx; sUnf[tid] = 1; }
    __syncthreads();

    // weight tiles rotate through two register buffers: X1: wb<-T3 wa<-T4 | X2: wb<-T5 wa<-T6 | S4 (wb<-F0) | S5 (wa<-F1) |
    // S6 (wb<-F2, wa<-T0', wb<-T1') | S1' (wa<-T2')
    f32x4 wa[PDEC_KB], wb[PDEC_KB];
    unsigned tag = 0;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // the h-dependent gate products of timestep t do not depend on the word chosen at t - 1: they are contracted at the END
    // of timestep t - 1, while its fc triples travel (S1' below; the F/A merge of the per-step loop).  t = 0: h1 = h2 = 0.
    f32x4 acc1 = zero4, acc2 = zero4;
    for (int t = 0; t < P.max_len; ++t) {
        // ================= S1: attention_lstm cell (h1); copy_lstm.h2h h2
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
            ll_put(h1rs, pb * D + pd, ao * tanhf(c1), tag);
        }
        pd_load(wb, pT3);
        pd_load_if(wa, pT4, v4);
        PD_STAMP(2);
        ll_stage<256, 8>(h1rs, sH1, B, D, LDH, tag, watch, tid);
        __syncthreads();
        // ================= S2: copy_lstm.x2h[:, :D] h1; context_gate / tc_affine rows of the owned columns, 4 projection rows
        PD_STAMP(3);
        pd_mma(acc2, wb, aH1);
        f32x4 accm = zero4;
        pd_mma(accm, wa, aH1);
        PD_STAMP(4);
#pragma unroll
        for (int e = 0; e < 4; ++e) sRed[(kq * 3 + 1) * 256 + (4 * g + e) * 16 + r] = accm[e];
        __syncthreads();
        ++tag;                                                   // X2: [cap_decoder_att(h1) | decoder_att(h1)]
        if (gcol && crr < 12) {
            const int o = cb * 16 + crr;
            const float v = ((sRed[1 * 256 + o] + sRed[4 * 256 + o]) + sRed[7 * 256 + o]) + sRed[10 * 256 + o];
            if (crr < 8) sM[cb * 8 + crr] = v;
            else ll_put(a2rs, cb * 2 * A + wg * 4 + crr - 8, v, tag);
        }
        PD_STAMP(5);
        ll_stage<256, 8>(a2rs, sA2, B, 2 * A, 2 * A, tag, watch, tid);
        __syncthreads();
        pd_load(wb, pT5);                                        // S4's tile streams under the attention's arithmetic
        PD_STAMP(6);
        // ================= S3: caption attention of every row (editnet.py:370-376), SelectC's arg-max (:409-416) ...
        // (RES: one row per wave, no loop and no branch around the scores — S4's weight tile, requested just above, streams
        // under this arithmetic; the compiler would drain it at a loop header or a branch join)
        auto attend = [&](const int b) {
            f32x4 a2[2];
            a2[0] = *reinterpret_cast<const f32x4*>(sA2 + b * 2 * A + a_lo) + *reinterpret_cast<const f32x4*>(sCon + a_lo);
            a2[1] = *reinterpret_cast<const f32x4*>(sA2 + b * 2 * A + a_hi) + *reinterpret_cast<const f32x4*>(sCon + a_hi);
            const f32x4 cfw0 = *reinterpret_cast<const f32x4*>(sCon + A + a_lo), cfw1 = *reinterpret_cast<const f32x4*>(sCon + A + a_hi);
            const float mk = RES ? mk_res : (lane < T ? P.mask[(long long)b * T + lane] : 1.f);
            float mine = 0.f;
            const unsigned long long live = __ballot(lane < T && mk != 0.f);
            if constexpr (RES) {
#pragma unroll
                for (int tt = 0; tt < PDEC_TREG; ++tt) {
                    const float sc = pd_wsum(pd_score8(a1r[tt][0] + a2[0], a1r[tt][1] + a2[1], cfw0, cfw1));
                    if (lane == tt) mine = sc;
                }
            } else {
                constexpr int RB = 10;
                const float* a1 = P.att1_c + (long long)b * T * A;
                for (int t0 = 0; t0 < T; t0 += RB) {
                    f32x4 v[RB][2];
#pragma unroll
                    for (int u = 0; u < RB; ++u) {
                        const int tt = t0 + u < T ? t0 + u : T - 1;
                        v[u][0] = *reinterpret_cast<const f32x4*>(a1 + (long long)tt * A + a_lo);
                        v[u][1] = *reinterpret_cast<const f32x4*>(a1 + (long long)tt * A + a_hi);
                    }
#pragma unroll
                    for (int u = 0; u < RB; ++u) {
                        if (!((live >> (t0 + u)) & 1ull)) continue;
                        const float sc = pd_wsum(pd_score8(v[u][0] + a2[0], v[u][1] + a2[1], cfw0, cfw1));
                        if (lane == t0 + u) mine = sc;
                    }
                }
            }
            const float sc = lane < T ? ((mk == 0.f) ? -1e10f : (mine + cbf)) : -INFINITY;
            const float m = pd_wmax(sc);
            const float ex = lane < T ? expf(sc - m) : 0.f;
            const float sum = pd_wsum(ex);
            const float al = ex / sum;
            if (lane < T) sAlc[b * PDEC_TMAX + lane] = al;
            // first arg-max of the weights (block_softmax in attention.hip): SelectC's hard choice
            float best = lane < T ? al : -1.f;
            int bi = lane < T ? lane : 0x7fffffff;
            if (!(best > -1.f)) bi = 0x7fffffff;                 // a NaN weight never wins a comparison
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ob = __shfl_xor(best, o);
                const int oi = __shfl_xor(bi, o);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            const int js = bi == 0x7fffffff ? 0 : bi;          // all-NaN weights: a valid row index (the outputs are NaN anyway)
            const float aj = __shfl(al, js);
            if (lane == 0) { sJs[b] = js; sWj[b] = aj * 1.f + (1.f - aj); }   // the reference's fp32 expression (editnet.py:417-418)
        };
        if constexpr (RES) { if (kq < B) attend(kq); }
        else { for (int b = kq; b < B; b += 4) attend(b); }
        pd_load_if(wa, pT6, v6);                                 // S5's (short) tile: under the context gate and the exchange
        // ... and this wave's visual score e[b, r] = w . relu(att1[b, r] + decoder_att(h1) + bias) + b (editnet.py:443-445)
        ++tag;                                                   // X3: attend_cap columns + visual scores
        if (vs_on) {
            const f32x4 x0 = va1_0 + (*reinterpret_cast<const f32x4*>(sA2 + vs_b * 2 * A + A + a_lo) + *reinterpret_cast<const f32x4*>(sCon + 2 * A + a_lo));
            const f32x4 x1 = va1_1 + (*reinterpret_cast<const f32x4*>(sA2 + vs_b * 2 * A + A + a_hi) + *reinterpret_cast<const f32x4*>(sCon + 2 * A + a_hi));
            const f32x4 vfw0 = *reinterpret_cast<const f32x4*>(sCon + 3 * A + a_lo), vfw1 = *reinterpret_cast<const f32x4*>(sCon + 3 * A + a_hi);
            float sc = vfw0[0] * fmaxf(x0[0], 0.f) + vfw0[1] * fmaxf(x0[1], 0.f) + vfw0[2] * fmaxf(x0[2], 0.f) + vfw0[3] * fmaxf(x0[3], 0.f);
            sc += vfw1[0] * fmaxf(x1[0], 0.f) + vfw1[1] * fmaxf(x1[1], 0.f) + vfw1[2] * fmaxf(x1[2], 0.f) + vfw1[3] * fmaxf(x1[3], 0.f);
            sc = pd_wsum(sc);
            if (lane == 0) ll_put(vsrs, vs_b * R + vs_r, sc + vbf, tag);
        }
        __syncthreads();
        PD_STAMP(7);
        if (zrole) {
            float s = 0.f;
            for (int tt = 0; tt < T; ++tt) s += sAlc[zb * PDEC_TMAX + tt] * sPz[tid * PDEC_TMAX + tt];
            sZ[zb * 8 + zc8] = s;
        }
        __syncthreads();
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
        PD_STAMP(8);
        ll_stage<256, 8>(gtrs, sX, B, D, LDH, tag, watch, tid);
        ll_stage<256, 8>(vsrs, sAlv, B, R, 64, tag, watch, tid);     // (only the R published words of a row; R is even)
        __syncthreads();
        PD_STAMP(9);
        // ================= S4: copy_lstm.x2h[:, D:2D] attend_cap; visual softmax; hoisted region products -> c_new
        pd_mma(acc2, wb, aX);
        pd_load_if(wb, pF[0], vF[0]);
        {   // (resident variant: one row per wave, no loop — see the attention phase)
            auto row_work = [&](const int b) {
                const float sc = lane < R ? sAlv[b * 64 + lane] : -INFINITY;
                const float m = pd_wmax(sc);
                const float ex = lane < R ? expf(sc - m) : 0.f;
                const float sum = pd_wsum(ex);
                if (lane < R) sAlv[b * 64 + lane] = ex / sum;
            };
            if constexpr (RES) { if (kq < B) row_work(kq); }
            else { for (int b = kq; b < B; b += 4) row_work(b); }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) sRed[(kq * 3 + 2) * 256 + (4 * g + e) * 16 + r] = acc2[e];
        __syncthreads();
        PD_STAMP(10);
        if (gcol) {
            const int o = cb * 16 + crr;
            float s = 0.f;
            for (int rr = 0; rr < R; ++rr) s += sAlv[cb * 64 + rr] * sPv[tid * PDEC_RREG + rr];
            const float g2 = ((sRed[2 * 256 + o] + sRed[5 * 256 + o]) + sRed[8 * 256 + o]) + sRed[11 * 256 + o];
            sG[o] = (g2 + s) + b2;
        }
        __syncthreads();
        ++tag;                                                   // X4: c_new
        float cnv = 0.f, ogv = 0.f;
        if (pair) {
            const float* gp = sG + pb * 16 + pu;
            const float ai = pd_sigm(gp[0]), af = pd_sigm(gp[4]), ag = tanhf(gp[8]);
            ogv = pd_sigm(gp[12]);
            cnv = af * c2 + ai * ag;
            ll_put(cnrs, pb * D + pd, cnv, tag);
        }
        PD_STAMP(11);
        ll_stage<256, 8>(cnrs, sX, B, D, LDH, tag, watch, tid);
        __syncthreads();
        PD_STAMP(12);
        // ================= S5: gate_cnew rows of the owned units, copy gate (editnet.py:281-283) -> c2, h2
        f32x4 acc5 = zero4;
        pd_mma(acc5, wa, aX);
        pd_load_if(wa, pF[1], vF[1]);
#pragma unroll
        for (int e = 0; e < 4; ++e) sRed[(kq * 3 + 0) * 256 + (4 * g + e) * 16 + r] = acc5[e];
        __syncthreads();
        ++tag;                                                   // X5: h2
        if (pair) {
            const int o = pb * 16 + pu;
            const float a = (((sRed[o] + sRed[3 * 256 + o]) + sRed[6 * 256 + o]) + sRed[9 * 256 + o]) + bcn;
            const float bq = cmemv + bcm;
            const float cg = pd_sigm(a + bq);
            c2 = cg * selv + (1.f - cg) * cnv;
            ll_put(h2rs, pb * D + pd, ogv * tanhf(c2), tag);
        }
        PD_STAMP(13);
        ll_stage<256, 8>(h2rs, sH2, B, D, LDH, tag, watch, tid);
        __syncthreads();
        PD_STAMP(14);
        // ================= S6: fc over this workgroup's vocabulary rows, local (max, first arg-max, sum exp) per batch row
        f32x4 accf0 = zero4, accf1 = zero4, accf2 = zero4;
        const bool more = t + 1 < P.max_len;
        pd_mma(accf0, wb, aH2);
        pd_load_if(wb, pF[2], vF[2]);
        pd_mma(accf1, wa, aH2);
        if (more) pd_load(wa, pT0);
        pd_mma(accf2, wb, aH2);
        if (more) pd_load(wb, pT1);
        PD_STAMP(15);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sRed[(kq * 3 + 0) * 256 + (4 * g + e) * 16 + r] = accf0[e];
            sRed[(kq * 3 + 1) * 256 + (4 * g + e) * 16 + r] = accf1[e];
            sRed[(kq * 3 + 2) * 256 + (4 * g + e) * 16 + r] = accf2[e];
        }
        __syncthreads();
        if (P.caps) {
            // teacher-forced: the scores themselves, rows 0 .. bt - 1 (editnet.py:546: predictions[:batch_size_t, t, :] = preds)
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
                pd_mma(acc1, wa, aH2);
                pd_load(wa, pT2);
                pd_mma(acc1, wb, aH1);
                pd_mma(acc2, wa, aH2);
            }
            __syncthreads();                                     // sRed is rewritten by the next timestep's S1
            continue;
        }
        ++tag;                                                   // X6: triples
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
    #pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const float ob = __shfl_xor(best, o);
                    const int oi = __shfl_xor(bi, o);
                    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
                }
                float se = ok ? expf(x - best) : 0.f;
                if (best == -INFINITY) se = ok ? x : 0.f;
                se = pd_wsum(se);
                if (lane < 4) ll_put(fcrs, (b * G + wg) * 4 + lane, lane == 0 ? best : (lane == 1 ? __int_as_float(bi) : (lane == 2 ? se : 0.f)), tag);
            };
            if constexpr (RES) { if (kq < B) row_work(kq); }
            else { for (int b = kq; b < B; b += 4) row_work(b); }
        }
        // ================= S1': attention_lstm's [W_ih[:, h2] | W_hh] and copy_lstm.h2h for timestep t + 1, while the triples
        // travel: nothing here waits for the word
        if (more) {
            acc1 = zero4; acc2 = zero4;
            pd_mma(acc1, wa, aH2);
            pd_load(wa, pT2);
            pd_mma(acc1, wb, aH1);
            pd_mma(acc2, wa, aH2);
        }
        PD_STAMP(16);
        ll_stage<256, 8>(fcrs, sF, B * G, 4, 4, tag, watch, tid);
        __syncthreads();
        PD_STAMP(17);
        // ================= S7: every workgroup combines the G triples of every row: same word everywhere
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
    #pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const float ob = __shfl_xor(best, o);
                    const int oi = __shfl_xor(bi, o);
                    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
                }
    #pragma unroll
                for (int i = 0; i < 4; ++i) tot += (pm[i] == -INFINITY) ? ps[i] : ps[i] * expf(pm[i] - best);
                tot = pd_wsum(tot);
                if (lane == 0) {
                    float logp = (best - best) - logf(tot);
                    if (bi == 0x7fffffff) { bi = 0; logp = __builtin_nanf(""); }
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
        PD_STAMP(18);
        int alive = 0;
        for (int b = 0; b < B; ++b) alive += sUnf[b];
        if (wg == 0 && tid == 0) P.alive[t] = alive;
        if (alive == 0) break;                                    // editnet_rl.py:546: every caption has ended
    }
    __shared__ unsigned s_bad;
    if (tid == 0) s_bad = __hip_atomic_load(P.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_bad) {                                                  // an exchange timed out: never hand this out as a decode
        const float qnan = __builtin_nanf("");
        if (P.caps) {                                              // teacher-forced: every score this workgroup wrote (see decode_persistent.hip)
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

static int g_pedit_capacity[64][2] = {};
static int g_pedit_capacity_lds[64][2] = {};
static int pedit_lds_floats(int B, int D, int A) {
    return 2 * B * (D + 4) + 4 * 3 * 256 + PDEC_MAXB * (PDEC_TMAX + 64 + 16 + 8 + 8) + B * 2 * A +
           B * 16 * PDEC_RREG + B * 8 * PDEC_TMAX + 4 * A;
}

// [status line | h1 | attend_cap | c_new | h2 | projections | visual scores | fc triples] as 8-byte flag-in-data words
size_t editnet_persistent_xbytes(int B, int D, int A) {
    if (B > PDW_MAXB) return 0;
    const size_t wide = editnet_persistent_wide_xbytes(B, D, A);
    if (B > PDEC_MAXB) return wide;
    const size_t narrow = 128 + (size_t)B * D * 8 * 4 + (size_t)B * 2 * A * 8 + (size_t)B * 64 * 8 + (size_t)B * (D / 4) * 32;
    return narrow > wide ? narrow : wide;
}

// rows from which the wide variant (decode_persistent_wide.hip) takes over (SET_DEC_WIDE_MINB; read per call)
static int wide_minb() { return env_int("SET_DEC_WIDE_MINB", 5); }

bool editnet_persistent_ok(const SetEditNetDims* d, int max_len) {
    const int on = env_int("SET_DEC_PERSISTENT", 1);                 // (read per call: tests and A/B runs flip it inside one process)
    const int maxb = env_int("SET_DEC_PERSISTENT_MAXB", PDW_MAXB);
    if (!on || d->B > maxb || max_len < 1 || d->adaptive) return false;
    if (d->B >= wide_minb() || d->B > PDEC_MAXB)                     // 5 .. 16 rows: the wide variant
        return editnet_persistent_wide_ok(d->B, d->D, d->A, d->T, d->R, d->V) && !persistent_disabled();
    // one mixed tile serves 4 + 4 + 4 rows: 2A attention-projection rows over D / 4 workgroups; 8 score columns per lane
    if (d->D != 64 * PDEC_KB || d->A != 512 || 2 * d->A != d->D || d->T > PDEC_TMAX || d->R > PDEC_RREG || d->R > 64 || (d->R & 1)) return false;
    const int G = d->D / 4;
    if ((d->V + G - 1) / G > 16 * PDEC_FC_TILES) return false;
    if (d->B * d->R > 4 * G) return false;                           // one visual score per wave
    if (pedit_lds_floats(d->B, d->D, d->A) * (int)sizeof(float) > 156 * 1024) return false;      // (160 KB per CU, a little of it static)
    if (pedit_lds_floats(d->B, d->D, d->A) * (int)sizeof(float) + 4096 > persistent_lds_limit()) return false;   // the DEVICE's limit (a 64-KB-LDS part)
    return !persistent_disabled();
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
    const bool wide = beam || B >= wide_minb() || B > PDEC_MAXB;
    PDecEditArgs P{};
    P.al_wih = w->al_wih; P.ld_ih = 3LL * D + F; P.al_whh = w->al_whh; P.cl_h2h_w = w->cl_h2h_w;
    P.cl_x2h_w = w->cl_x2h_w; P.ld_x2h = 2LL * D + F; P.cl_x2h_b = w->cl_x2h_b; P.cl_h2h_b = w->cl_h2h_b;
    P.ca_gate_w = w->ca_gate_w; P.ca_gate_b = w->ca_gate_b; P.ca_tc_w = w->ca_tc_w; P.ca_tc_b = w->ca_tc_b; P.ca_sc_b = w->ca_sc_b;
    P.ca_dec_w = w->ca_dec_w; P.ca_dec_b = w->ca_dec_b; P.ca_full_w = w->ca_full_w; P.ca_full_b = w->ca_full_b;
    P.va_dec_w = w->va_dec_w; P.va_dec_b = w->va_dec_b; P.va_full_w = w->va_full_w; P.va_full_b = w->va_full_b;
    P.cl_cnew_w = w->cl_cnew_w; P.cl_cnew_b = w->cl_cnew_b; P.cl_cmem_b = w->cl_cmem_b;
    P.fc_w = w->fc_w; P.fc_b = w->fc_b; P.tok_table = w->tok_table; P.ld_tab = 10LL * D;
    P.pre1 = pre1; P.att1 = att1; P.att1_c = att1_c; P.mask = mask; P.capP = capP; P.memQ = memQ; P.Mem = Mem; P.pv = pv;
    {
        char* x = (char*)xbuf;
        P.status = (unsigned*)x; x += 128;
        P.x_h1 = x; x += (size_t)B * D * 8;
        P.x_gt = x; x += (size_t)B * D * 8;
        P.x_cn = x; x += (size_t)B * D * 8;
        P.x_h2 = x; x += (size_t)B * D * 8;
        P.x_a2 = x; x += (size_t)B * 2 * A * 8;
        P.x_vs = x; x += (size_t)B * 64 * 8;
        P.x_fc = x;
    }
    P.it = it; P.unfinished = unfinished; P.alive = alive; P.seq = seq; P.seq_logp = seq_logp;
    P.B = B; P.D = D; P.T = d->T; P.R = d->R; P.A = A; P.V = d->V; P.max_len = max_len; P.rpw = (d->V + G - 1) / G;
    P.start_idx = start_idx; P.end_idx = end_idx;
    if (teach) {
        P.caps = (const long long*)teach->caps; P.caps_stride = teach->caps_stride;
        P.predictions = teach->predictions; P.ld_pred_b = (long long)max_len * d->V;
        for (int b = 0; b < B; ++b) P.dlen[b] = teach->host_decode_lengths[b];
    }
    const int lds = pedit_lds_floats(B, D, A) * (int)sizeof(float);
    PersistentGuard guard;
    if (guard.rc != SET_OK) return guard.rc;
    const int dev = guard.dev;
    P.spin_limit = guard.spin_limit();
    P.test_stall = guard.test_stall(); P.fault = guard.fault;
    if (beam) {
        P.bm_hist_par = beam->hist_par; P.bm_hist_word = (long long*)beam->hist_word; P.bm_best_score = beam->best_score;
        P.bm_best_word = (long long*)beam->best_word; P.bm_result = beam->result;
    }
    if (wide) {
        bool unsupported = true;
        const int rc = editnet_persistent_wide_launch(P, xbuf, guard, s, &unsupported, beam != nullptr);
        return rc != SET_OK ? rc : (unsupported ? SET_ERR_UNSUPPORTED : SET_OK);
    }
    static bool configured[2][64] = {};
    int lds_max = pedit_lds_floats(PDEC_MAXB, D, A) * (int)sizeof(float);
    if (lds_max > 156 * 1024) lds_max = 156 * 1024;             // (editnet_persistent_ok refuses batches that need more)
    if (guard.set_lds(reinterpret_cast<const void*>(&editnet_persistent_k<true>), lds_max, configured[0]) != SET_OK ||
        guard.set_lds(reinterpret_cast<const void*>(&editnet_persistent_k<false>), lds_max, configured[1]) != SET_OK)
        return SET_ERR_UNSUPPORTED;
    const bool res = B <= 4 && d->T <= PDEC_TREG;
    int& cap = g_pedit_capacity[dev][res ? 1 : 0];
    int& cap_lds = g_pedit_capacity_lds[dev][res ? 1 : 0];
    if (cap == 0 || lds > cap_lds) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, res ? reinterpret_cast<const void*>(&editnet_persistent_k<true>)
                                                                       : reinterpret_cast<const void*>(&editnet_persistent_k<false>),
                                                         PDEC_THREADS, (size_t)lds) != hipSuccess ||
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
    const double wbytes = 4.0 * ((double)d->V * D + 5.0 * 4 * D * D + 3.0 * D * D + 2.0 * A * D);
    ProfScope ps("persistent_decode", s, 2.0 * B * wbytes / 4.0 * max_len, wbytes * max_len);
    SET_TRY(guard.serialise(s));
    SET_HIP_TRY(hipMemsetAsync(xbuf, 0, editnet_persistent_xbytes(B, D, A), s));    // no word of an earlier decode may carry a tag of this one
    SET_TRY(pd_stamps_begin(&P.stamps, &P.stamp_wg, s));
    if (res) hipLaunchKernelGGL(editnet_persistent_k<true>, dim3(G), dim3(PDEC_THREADS), lds, s, P);
    else hipLaunchKernelGGL(editnet_persistent_k<false>, dim3(G), dim3(PDEC_THREADS), lds, s, P);
    SET_LAUNCH_CHECK();
    SET_TRY(guard.launched(s));
    SET_TRY(pd_stamps_report(P.stamps, P.stamp_wg, 18, max_len, s));
    return SET_OK;
}

}  // namespace set
