// Beam-search step epilogue for MANY images at once (the reference's evaluate() is
// batch 1: editnet.py:654-699, dcnet.py:450-500, eval_full.py:150-200).
//
// One workgroup per image.  The k hypotheses of image i are rows i*k .. i*k+k-1 of the logits.
//   1. log-softmax statistics of every live row (block-wide online max / sum-exp), or for the ensemble
//      log((softmax_e + softmax_d) / 2) (eval_full.py:151-153);
//   2. flat top-k over the k*V candidates  score[j] + logp[j][v]  (editnet.py:659-664): per-thread sorted
//      top-KMAX lists, then k rounds of block arg-max that pop the winner (ties: lowest flat index);
//   3. the bookkeeping the reference does on the host (editnet.py:666-699): parent / word split, completed
//      hypotheses (best completed score, first maximum over time), k shrinks by the number of <end>s, live
//      hypotheses compacted to the front in pick order, sequences re-indexed by parent and extended,
//      next input words, and the parent row map that beam_gather_k uses to re-index the LSTM states.
// Nothing is synchronised with the host; an image whose k has reached 0 is a no-op.
#include "set_common.h"

namespace set {

constexpr int BEAM_KMAX = 8;

struct BeamArgs {
    const float* logits;
    const float* logits2;        // second model of the ensemble, or nullptr
    long long ld;
    int k, V, cur_len, Lmax;
    long long end_idx;
    float* scores;               // (NI, k) running hypothesis scores, -inf = dead slot
    int* k_left;                 // (NI)
    const long long* seqs_in;    // (NI, k, Lmax), cur_len tokens valid
    long long* seqs_out;
    float* best_score;           // (NI) best completed hypothesis so far
    long long* best_seq;         // (NI, Lmax)
    int* best_len;               // (NI)
    long long* words;            // (NI*k) next input token per hypothesis row
    int* rows;                   // (NI*k) source row of each hypothesis row's recurrent state
};

__device__ __forceinline__ bool beam_better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

__global__ void __launch_bounds__(256) beam_pick_k(const BeamArgs a) {
    __shared__ float s_m[4][2], s_s[4][2];
    __shared__ float s_lse[BEAM_KMAX][2];
    __shared__ float s_rv[4];
    __shared__ int s_ri[4];
    __shared__ float s_pick_v[BEAM_KMAX];
    __shared__ int s_pick_i[BEAM_KMAX];
    __shared__ int s_src[BEAM_KMAX];            // per output slot: parent hypothesis
    __shared__ long long s_word[BEAM_KMAX];     // per output slot: appended word
    __shared__ int s_best_parent;
    __shared__ long long s_best_word;
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = a.k, V = a.V;
    const int kl = a.k_left[img];
    if (kl <= 0) {                               // finished image: identity state map, nothing else changes
        if (tid < k) { a.rows[img * k + tid] = img * k + tid; a.words[img * k + tid] = 0; }
        return;
    }
    float sc[BEAM_KMAX];
#pragma unroll
    for (int j = 0; j < BEAM_KMAX; ++j) sc[j] = j < k ? a.scores[img * k + j] : -INFINITY;

    // ---- 1. log-sum-exp of every live row (both models for the ensemble)
    const int nmodel = a.logits2 ? 2 : 1;
    for (int j = 0; j < k; ++j) {
        if (sc[j] == -INFINITY) continue;        // uniform across the block
        for (int mdl = 0; mdl < nmodel; ++mdl) {
            const float* row = (mdl ? a.logits2 : a.logits) + (long long)(img * k + j) * a.ld;
            float m = -INFINITY, s = 0.f;
            for (int v = tid; v < V; v += 256) {
                const float x = row[v];
                if (x > m) { s = s * expf(m - x) + 1.f; m = x; }
                else s += expf(x - m);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float om = __shfl_xor(m, off), os = __shfl_xor(s, off);
                const float nm = fmaxf(m, om);
                s = (m == -INFINITY ? 0.f : s * expf(m - nm)) + (om == -INFINITY ? 0.f : os * expf(om - nm));
                m = nm;
            }
            if (lane == 0) { s_m[wave][mdl] = m; s_s[wave][mdl] = s; }
        }
        __syncthreads();
        if (tid < nmodel) {
            float m = s_m[0][tid], s = s_s[0][tid];
            for (int w = 1; w < 4; ++w) {
                const float om = s_m[w][tid], os = s_s[w][tid], nm = fmaxf(m, om);
                s = s * expf(m - nm) + os * expf(om - nm);
                m = nm;
            }
            s_lse[j][tid] = m + logf(s);
        }
        __syncthreads();
    }

    // ---- 2a. per-thread sorted top-KMAX of the candidates it owns (ascending flat index per thread)
    float lv[BEAM_KMAX];
    int li[BEAM_KMAX];
#pragma unroll
    for (int q = 0; q < BEAM_KMAX; ++q) { lv[q] = -INFINITY; li[q] = 0x7fffffff; }
    for (int j = 0; j < k; ++j) {
        if (sc[j] == -INFINITY) continue;
        const float* row = a.logits + (long long)(img * k + j) * a.ld;
        const float* row2 = a.logits2 ? a.logits2 + (long long)(img * k + j) * a.ld : nullptr;
        const float l0 = s_lse[j][0], l1 = row2 ? s_lse[j][1] : 0.f;
        for (int v = tid; v < V; v += 256) {
            float lp;
            if (row2) lp = logf((expf(row[v] - l0) + expf(row2[v] - l1)) * 0.5f);
            else lp = row[v] - l0;
            const float x = sc[j] + lp;
            if (x > lv[BEAM_KMAX - 1]) {
                lv[BEAM_KMAX - 1] = x; li[BEAM_KMAX - 1] = j * V + v;
#pragma unroll
                for (int q = BEAM_KMAX - 1; q > 0; --q)
                    if (lv[q] > lv[q - 1]) {
                        const float tv = lv[q]; lv[q] = lv[q - 1]; lv[q - 1] = tv;
                        const int ti = li[q]; li[q] = li[q - 1]; li[q - 1] = ti;
                    }
            }
        }
    }
    // ---- 2b. k rounds of block arg-max over the list heads; the winner pops its head
    for (int r = 0; r < k; ++r) {
        float bv = lv[0];
        int bi = li[0];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(bv, off);
            const int oi = __shfl_xor(bi, off);
            if (beam_better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { s_rv[wave] = bv; s_ri[wave] = bi; }
        __syncthreads();
        bv = s_rv[0]; bi = s_ri[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (beam_better(s_rv[w], s_ri[w], bv, bi)) { bv = s_rv[w]; bi = s_ri[w]; }
        if (li[0] == bi && bi != 0x7fffffff) {
#pragma unroll
            for (int q = 0; q + 1 < BEAM_KMAX; ++q) { lv[q] = lv[q + 1]; li[q] = li[q + 1]; }
            lv[BEAM_KMAX - 1] = -INFINITY; li[BEAM_KMAX - 1] = 0x7fffffff;
        }
        if (tid == 0) { s_pick_v[r] = bv; s_pick_i[r] = bi; }
        __syncthreads();
    }

    // ---- 3. bookkeeping of one image (editnet.py:666-699)
    if (tid == 0) {
        int n_end = 0, c_arg = -1, slot = 0;
        float c_best = -INFINITY;
        bool live[BEAM_KMAX];
        for (int r = 0; r < k; ++r) {
            const int flat = s_pick_i[r];
            const bool ok = flat != 0x7fffffff && r < kl;               // only the first k_left picks count
            const long long word = ok ? flat % V : 0;
            const bool is_end = ok && word == a.end_idx;
            live[r] = ok && !is_end;
            if (is_end) {
                ++n_end;
                if (s_pick_v[r] > c_best) { c_best = s_pick_v[r]; c_arg = r; }     // first maximum
            }
        }
        s_best_parent = -1;
        if (c_arg >= 0 && c_best > a.best_score[img]) {
            a.best_score[img] = c_best;
            a.best_len[img] = a.cur_len + 1;
            s_best_parent = s_pick_i[c_arg] / V;
            s_best_word = s_pick_i[c_arg] % V;
        }
        a.k_left[img] = kl - n_end;
        // live picks first, in pick order; the other slots die
        for (int pass = 0; pass < 2; ++pass)
            for (int r = 0; r < k; ++r) {
                if ((pass == 0) != live[r]) continue;
                const int flat = s_pick_i[r];
                const int parent = flat != 0x7fffffff ? flat / V : 0;
                const long long word = flat != 0x7fffffff ? flat % V : 0;
                a.scores[img * k + slot] = live[r] ? s_pick_v[r] : -INFINITY;
                a.words[img * k + slot] = live[r] ? word : 0;
                a.rows[img * k + slot] = img * k + parent;
                s_src[slot] = parent;
                s_word[slot] = word;
                ++slot;
            }
    }
    __syncthreads();
    const int L = a.cur_len;
    const long long* sin = a.seqs_in + (long long)img * k * a.Lmax;
    long long* sout = a.seqs_out + (long long)img * k * a.Lmax;
    for (int e = tid; e < k * (L + 1); e += 256) {
        const int slot = e / (L + 1), p = e - slot * (L + 1);
        sout[(long long)slot * a.Lmax + p] = p < L ? sin[(long long)s_src[slot] * a.Lmax + p] : s_word[slot];
    }
    if (s_best_parent >= 0)
        for (int p = tid; p <= L; p += 256)
            a.best_seq[(long long)img * a.Lmax + p] = p < L ? sin[(long long)s_best_parent * a.Lmax + p] : s_best_word;
}

// state[s][r] <- state[s][rows[r]] for the k rows of one image, in place (all parents are rows of the same image)
struct BeamStates {
    float* p[4];
    int n;
};
__global__ void __launch_bounds__(256) beam_gather_k(BeamStates st, const int* rows, int k, int D) {
    extern __shared__ float lds[];
    const int img = blockIdx.x, tid = threadIdx.x;
    float* base = st.p[blockIdx.y] + (long long)img * k * D;
    bool identity = true;
    for (int j = 0; j < k; ++j) identity = identity && rows[img * k + j] == img * k + j;
    if (identity) return;
    for (int e = tid; e < k * D; e += 256) lds[e] = base[e];
    __syncthreads();
    for (int e = tid; e < k * D; e += 256) {
        const int j = e / D, c = e - j * D;
        base[e] = lds[(rows[img * k + j] - img * k) * D + c];
    }
}

}  // namespace set

using namespace set;

extern "C" {

int set_beam_pick_f32(const float* logits, const float* logits2, int64_t ld, int NI, int k, int V, int64_t end_idx,
                      int cur_len, int Lmax, float* scores, int32_t* k_left, const int64_t* seqs_in, int64_t* seqs_out,
                      float* best_score, int64_t* best_seq, int32_t* best_len, int64_t* words, int32_t* rows,
                      void* stream) {
    if (!logits || !scores || !k_left || !seqs_in || !seqs_out || !best_score || !best_seq || !best_len || !words || !rows)
        return SET_ERR_ARG;
    if (NI <= 0 || k <= 0 || V <= 0 || cur_len < 1 || cur_len + 1 > Lmax || ld < V) return SET_ERR_ARG;
    if (k > BEAM_KMAX || (long long)k * V >= 0x7fffffffLL) return SET_ERR_UNSUPPORTED;
    BeamArgs a;
    a.logits = logits; a.logits2 = logits2; a.ld = ld; a.k = k; a.V = V; a.cur_len = cur_len; a.Lmax = Lmax;
    a.end_idx = end_idx; a.scores = scores; a.k_left = k_left;
    a.seqs_in = (const long long*)seqs_in; a.seqs_out = (long long*)seqs_out;
    a.best_score = best_score; a.best_seq = (long long*)best_seq; a.best_len = best_len;
    a.words = (long long*)words; a.rows = rows;
    ProfScope ps("beam_pick", (hipStream_t)stream, 0.0, 8.0 * NI * k * V * (logits2 ? 2.0 : 1.0));
    hipLaunchKernelGGL(beam_pick_k, dim3(NI), dim3(256), 0, (hipStream_t)stream, a);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

int set_beam_gather_f32(float* s0, float* s1, float* s2, float* s3, const int32_t* rows, int NI, int k, int D,
                        void* stream) {
    if (!s0 || !rows || NI <= 0 || k <= 0 || D <= 0) return SET_ERR_ARG;
    if (k > BEAM_KMAX || (size_t)k * D * sizeof(float) > 64 * 1024) return SET_ERR_UNSUPPORTED;
    BeamStates st;
    st.p[0] = s0; st.p[1] = s1; st.p[2] = s2; st.p[3] = s3;
    st.n = s3 ? 4 : (s2 ? 3 : (s1 ? 2 : 1));
    ProfScope ps("beam_gather", (hipStream_t)stream, 0.0, 8.0 * NI * k * D * st.n);
    hipLaunchKernelGGL(beam_gather_k, dim3(NI, st.n), dim3(256), (size_t)k * D * sizeof(float), (hipStream_t)stream, st,
                       rows, k, D);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

}  // extern "C"
