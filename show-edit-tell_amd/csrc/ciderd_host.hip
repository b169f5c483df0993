// Host-side CIDEr-D for the self-critical reward (editnet_rl.py:611-646 calls an external,
// un-vendored Python scorer per batch).  No device code here: scoring is per-sample string/n-gram work that shards
// with the batch, but at 5 samples per image it was HALF of the SCST step's wall time in Python (74 of 147 ms at
// B = 64), so the hot part — n-gram counting, tf-idf vectors, clipped cosine with the length penalty — is native:
//
//   g_n(s)      = tf(ngram) * (log N_docs - log max(1, df(ngram)))                      n = 1..4
//   sim_n(c, r) = sum_ngram min(g(c), g(r)) * g(r) / (|g(c)| |g(r)|) * exp(-(l_c - l_r)^2 / (2 sigma^2))
//   CIDEr-D(c)  = 10 / (n |refs|) * sum_n sum_r sim_n(c, r),      l = number of bigrams
//
// Sentences arrive as int64 token ids (the Python wrapper interns token strings); an n-gram is a (length, 4 ids) key.
// The Python implementation in ciderd.py stays as the readable statement of the metric and as the cross-check.
#include <cmath>
#include <cstring>
#include <unordered_map>
#include <vector>
#include <algorithm>
#include "set_common.h"

namespace {

struct Gram {
    int64_t t[4];
    int32_t n;
    bool operator==(const Gram& o) const { return n == o.n && t[0] == o.t[0] && t[1] == o.t[1] && t[2] == o.t[2] && t[3] == o.t[3]; }
    bool operator<(const Gram& o) const {
        if (n != o.n) return n < o.n;
        for (int i = 0; i < 4; ++i) if (t[i] != o.t[i]) return t[i] < o.t[i];
        return false;
    }
};
struct GramHash {
    size_t operator()(const Gram& g) const {
        uint64_t h = 0x9E3779B97F4A7C15ull * (uint64_t)(g.n + 1);
        for (int i = 0; i < 4; ++i) {
            h ^= (uint64_t)g.t[i] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
            h *= 0xBF58476D1CE4E5B9ull;
        }
        return (size_t)(h ^ (h >> 31));
    }
};

struct Vec {                                   // tf-idf vector of one sentence, per n-gram order
    std::vector<std::pair<Gram, double>> w[4]; // sorted by key
    double norm[4];
    long length;                               // number of bigrams (the length the public implementations use)
};

struct Scorer {
    std::unordered_map<Gram, double, GramHash> df;
    double log_ref_len, sigma;
    int n;

    void vectorise(const int64_t* tok, int64_t len, Vec& v) const {
        std::unordered_map<Gram, int, GramHash> counts;
        counts.reserve((size_t)len * 4 + 8);
        for (int k = 1; k <= n; ++k)
            for (int64_t i = 0; i + k <= len; ++i) {
                Gram g{{0, 0, 0, 0}, k};
                for (int j = 0; j < k; ++j) g.t[j] = tok[i + j];
                ++counts[g];
            }
        for (int k = 0; k < 4; ++k) { v.w[k].clear(); v.norm[k] = 0.0; }
        v.length = 0;
        for (const auto& kv : counts) {
            const int k = kv.first.n - 1;
            auto it = df.find(kv.first);
            const double d = it == df.end() ? 0.0 : it->second;
            const double w = (double)kv.second * (log_ref_len - std::log(std::max(1.0, d)));
            v.w[k].emplace_back(kv.first, w);
            if (k == 1) v.length += kv.second;
        }
        for (int k = 0; k < n; ++k) {
            std::sort(v.w[k].begin(), v.w[k].end(), [](const auto& a, const auto& b) { return a.first < b.first; });
            double s = 0.0;
            for (const auto& e : v.w[k]) s += e.second * e.second;      // fixed (key) order: deterministic
            v.norm[k] = std::sqrt(s);
        }
    }

    void similarity(const Vec& h, const Vec& r, double* out) const {
        const double delta = (double)(h.length - r.length);
        const double penalty = std::exp(-(delta * delta) / (2.0 * sigma * sigma));
        for (int k = 0; k < n; ++k) {
            double s = 0.0;
            size_t i = 0, j = 0;
            const auto& a = h.w[k];
            const auto& b = r.w[k];
            while (i < a.size() && j < b.size()) {                      // both sorted by key: merge join
                if (a[i].first < b[j].first) ++i;
                else if (b[j].first < a[i].first) ++j;
                else { s += std::min(a[i].second, b[j].second) * b[j].second; ++i; ++j; }
            }
            if (h.norm[k] != 0.0 && r.norm[k] != 0.0) s /= h.norm[k] * r.norm[k];
            out[k] += s * penalty;
        }
    }
};

}  // namespace

extern "C" {

void* set_ciderd_create(const int64_t* tokens, const int32_t* lens, const double* df, int64_t n_entries, double ref_len,
                        int n, double sigma) {
    if (n < 1 || n > 4 || ref_len <= 0.0 || n_entries < 0 || (n_entries > 0 && (!tokens || !lens || !df))) return nullptr;
    Scorer* s = new Scorer();
    s->n = n;
    s->sigma = sigma;
    s->log_ref_len = std::log(ref_len);
    s->df.reserve((size_t)n_entries * 2 + 16);
    for (int64_t i = 0; i < n_entries; ++i) {
        if (lens[i] < 1 || lens[i] > 4) continue;
        Gram g{{0, 0, 0, 0}, lens[i]};
        for (int j = 0; j < lens[i]; ++j) g.t[j] = tokens[4 * i + j];
        s->df[g] = df[i];
    }
    return s;
}

void set_ciderd_destroy(void* h) { delete static_cast<Scorer*>(h); }

// hypothesis i = hyp_tokens[hyp_off[i] .. hyp_off[i+1]); its reference set = set_of_hyp[i]; set s holds the references
// r in [ref_set_off[s], ref_set_off[s+1]); reference r = ref_tokens[ref_off[r] .. ref_off[r+1]).  scores (n_hyp) doubles.
int set_ciderd_score(void* h, const int64_t* hyp_tokens, const int64_t* hyp_off, int n_hyp, const int32_t* set_of_hyp,
                     const int64_t* ref_tokens, const int64_t* ref_off, const int64_t* ref_set_off, int n_sets,
                     double* scores) {
    if (!h || n_hyp < 0 || n_sets < 0 || (n_hyp > 0 && (!hyp_off || !set_of_hyp || !scores)) ||
        (n_sets > 0 && (!ref_off || !ref_set_off)))
        return SET_ERR_ARG;
    const Scorer& S = *static_cast<Scorer*>(h);
    const int64_t n_refs = n_sets > 0 ? ref_set_off[n_sets] : 0;
    std::vector<Vec> rv((size_t)n_refs);
    for (int64_t r = 0; r < n_refs; ++r) S.vectorise(ref_tokens + ref_off[r], ref_off[r + 1] - ref_off[r], rv[(size_t)r]);
    Vec hv;
    for (int i = 0; i < n_hyp; ++i) {
        const int s = set_of_hyp[i];
        if (s < 0 || s >= n_sets) return SET_ERR_ARG;
        S.vectorise(hyp_tokens + hyp_off[i], hyp_off[i + 1] - hyp_off[i], hv);
        double tot[4] = {0.0, 0.0, 0.0, 0.0};
        const int64_t r0 = ref_set_off[s], r1 = ref_set_off[s + 1];
        for (int64_t r = r0; r < r1; ++r) S.similarity(hv, rv[(size_t)r], tot);
        double mean = 0.0;
        for (int k = 0; k < S.n; ++k) mean += tot[k];
        mean /= (double)S.n;
        scores[i] = mean / (double)std::max<int64_t>(1, r1 - r0) * 10.0;
    }
    return SET_OK;
}

}  // extern "C"
