// Microbenchmark of the decode step's three grouped fp32-MFMA GEMM launches (F/A, B, D of csrc/editnet.hip step_impl)
// with the library's own kernel source, outside Python:  build with tools/ubench/build_gemm_steps.sh, run on the GPU box.
//   gemm_steps [iters] [mode]     mode 0: each launch shape timed in its own loop + the F/A,B,D rotation
// Kernel variants are selected with -D flags on the included gemm_f32.hip (same flags as tools/ab.sh).
#include "../../show-edit-tell_amd/csrc/gemm_f32.hip"
#include <cstdio>
#include <cmath>
#include <vector>
using namespace set;
// self-contained: the few host helpers gemm_f32.hip expects from the rest of the library (linking libset_hip.so as well
// would register a second copy of the same kernel symbols and the runtime may launch the library's instead of this build's)
namespace set {
thread_local int g_last_hip_error = 0;
thread_local RowGate g_row_gate;
int env_int(const char* name, int dflt) { const char* v = getenv(name); return (v && *v) ? atoi(v) : dflt; }
ProfScope::ProfScope(const char*, hipStream_t s, double, double) : idx(-1), st(s) {}
ProfScope::~ProfScope() {}
}

typedef float f32x16_ __attribute__((ext_vector_type(16)));
// effective shader clock over a kernel: s_memtime (shader cycles) against s_memrealtime (100 MHz), block 0 / thread 0
__device__ unsigned long long g_raw_clock[2];
#define RAW_CLOCK_BEGIN() const unsigned long long cy0_ = clock64(), wl0_ = wall_clock64()
#define RAW_CLOCK_END() if (blockIdx.x == 0 && threadIdx.x == 0) { g_raw_clock[0] = clock64() - cy0_; g_raw_clock[1] = wall_clock64() - wl0_; }
__global__ void __launch_bounds__(256) raw_mfma_k(float* out, int iters, float a0, float b0) {
    RAW_CLOCK_BEGIN();
    f32x16_ c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
        }
    }
    float s = 0;
    for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    RAW_CLOCK_END();
}

__global__ void __launch_bounds__(256) empty_k(float* out, int flag) { if (flag) out[threadIdx.x] = 0.f; }

// the same issue pattern on random operands (8 operand pairs per lane, rotated) instead of constants: what the matrix pipe
// sustains when its inputs toggle like real weights / activations do
__global__ void __launch_bounds__(256) raw_mfma_rand_k(float* out, const float* src, int iters) {
    RAW_CLOCK_BEGIN();
    f32x16_ c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {          // pseudo-random mantissas, no memory access (same start-up as the constant kernel)
        const unsigned ha = (threadIdx.x * 2654435761u + i * 40503u + blockIdx.x * 97u) * 2246822519u;
        const unsigned hb = (threadIdx.x * 3266489917u + i * 668265263u + blockIdx.x * 31u) * 374761393u;
        a[i] = (__uint_as_float(0x3f800000u | (ha >> 9)) - 1.5f) * 0.2f + src[0] * 0.f;
        b[i] = (__uint_as_float(0x3f800000u | (hb >> 9)) - 1.5f) * 0.2f;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * r], b[2 * r], c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * r + 1], b[2 * r + 1], c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * r + 1], b[2 * r], c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * r], b[2 * r + 1], c3, 0, 0, 0);
        }
    }
    float s = 0;
    for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    RAW_CLOCK_END();
}

static void print_raw_clock(const char* what) {
    unsigned long long h[2];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_raw_clock), sizeof h);
    printf("     %s: %.0f shader cycles in %.2f us -> effective shader clock %.3f GHz (%.1f cycles per MFMA per SIMD at 3 waves)\n", what,
           (double)h[0], h[1] / 100.0, (double)h[0] / (h[1] * 10.0), (double)h[0] / (512.0 * 16 * 3));
}

static float* dev_rand(size_t n, unsigned seed) {
    std::vector<float> h(n);
    static const bool zero = getenv("ZERO_INPUT") != nullptr;      // power / clock probe: operands that never toggle
    if (zero) { float* d; hipMalloc(&d, n * sizeof(float)); hipMemset(d, 0, n * sizeof(float)); return d; }
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.1f; }
    float* d; hipMalloc(&d, n * sizeof(float)); hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
    return d;
}
static GemmProb slab(float* buf, int M, int N) {
    GemmProb p; p.C = buf; p.M = M; p.N = N; p.ldc = N; p.slab_stride = (long long)M * N; return p;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    const int B = 128, D = 1024, F = 2048, A = 512, V = 10000;
    const int tgt = 512;
    float *h1 = dev_rand((size_t)B * D, 1), *h2 = dev_rand((size_t)B * D, 2), *cap = dev_rand((size_t)B * D, 3),
          *img = dev_rand((size_t)B * F, 4);
    float* fc_w = dev_rand((size_t)V * D, 5); float* fc_b = dev_rand(V, 6);
    float* al_wih = dev_rand((size_t)4 * D * (3 * D + F), 7); float* al_whh = dev_rand((size_t)4 * D * D, 8);
    float* cl_h2h = dev_rand((size_t)4 * D * D, 9); float* cl_x2h = dev_rand((size_t)4 * D * (2 * D + F), 10);
    float* ca_dec = dev_rand((size_t)A * D, 11); float* va_dec = dev_rand((size_t)A * D, 12);
    float* ca_tc = dev_rand((size_t)D * 2 * D, 13); float* ca_gate = dev_rand((size_t)D * 3 * D, 14);
    float* slabs; hipMalloc(&slabs, (size_t)160 << 22);      // 160 M floats of slab space
    float* logits; hipMalloc(&logits, (size_t)B * 10048 * 4);
    size_t off = 0;
    auto take = [&](size_t n) { float* p = slabs + off; off += n; return p; };
    const long long ld_ih = 3LL * D + F, ld_x2h = 2LL * D + F;

    GemmProb fa[3];
    fa[0] = slab(take((size_t)8 * B * 10048), B, V); fa[0].ldc = 10048; fa[0].slab_stride = (long long)B * 10048;
    fa[0].add(h2, D, fc_w, D, D);
    fa[1] = slab(take((size_t)8 * B * 4 * D), B, 4 * D); fa[1].add(h2, D, al_wih + 2 * D, ld_ih, D); fa[1].add(h1, D, al_whh, D, D);
    fa[2] = slab(take((size_t)8 * B * 4 * D), B, 4 * D); fa[2].add(h2, D, cl_h2h, D, D);
    plan_ksplit(fa, 3, tgt);
    if (fa[0].ksplit == 1) { fa[0].C = logits; fa[0].slab_stride = 0; fa[0].bias = fc_b; }
    GemmProb b[5];
    b[0] = slab(take((size_t)8 * B * A), B, A); b[0].add(h1, D, ca_dec, D, D);
    b[1] = slab(take((size_t)8 * B * A), B, A); b[1].add(h1, D, va_dec, D, D);
    b[2] = slab(take((size_t)8 * B * D), B, D); b[2].add(h1, D, ca_tc + D, 2 * D, D);
    b[3] = slab(take((size_t)8 * B * D), B, D); b[3].add(h1, D, ca_gate + D, 3 * D, D);
    b[4] = slab(take((size_t)8 * B * 4 * D), B, 4 * D); b[4].add(h1, D, cl_x2h, ld_x2h, D);
    plan_ksplit(b, 5, tgt);
    GemmProb dd = slab(take((size_t)8 * B * 4 * D), B, 4 * D);
    dd.add(cap, D, cl_x2h + D, ld_x2h, D); dd.add(img, F, cl_x2h + 2 * D, ld_x2h, F);
    plan_ksplit(&dd, 1, tgt);
    // long-K reference problem (steady state): M = 128, N = 4096, K = 3 x 8192 re-reading the x2h weights
    GemmProb big = slab(take((size_t)2560 * 4 * D), 2560, 4 * D);
    float* bigA = dev_rand((size_t)2560 * D, 20);
    big.add(bigA, D, cl_h2h, D, D); big.max_ksplit = 1;
    plan_ksplit(&big, 1, tgt);

    auto flops = [](const GemmProb* p, int n) { double f = 0; for (int i = 0; i < n; ++i) f += 2.0 * p[i].M * p[i].N * p[i].ktiles() * 32.0; return f; };
    auto wgs = [](const GemmProb* p, int n) { int w = 0; for (int i = 0; i < n; ++i) w += cdiv(p[i].M, tile_m_of(p[0])) * cdiv(p[i].N, launch_tile_n(p, n)) * p[i].ksplit; return w; };
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto fn, double fl, int n_launch) {
        for (int i = 0; i < 20; ++i) fn();
        hipEventRecord(e0, st);
        for (int i = 0; i < iters; ++i) fn();
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %8.2f us/iter  %7.2f TFLOP/s  (%d launches/iter, %.2f us/launch)\n", name, 1e3 * ms / iters,
               fl / (ms / iters * 1e-3) / 1e12, n_launch, 1e3 * ms / iters / n_launch);
    };
    if (getenv("ONLY_RAW")) {          // long raw-MFMA runs for clock / power sampling with rocm-smi
        float* out; hipMalloc(&out, 768 * 256 * 4);
        float* src = dev_rand(8192, 77);
        if (atoi(getenv("ONLY_RAW")) == 1)
            run("raw MFMA, random operands", [&] { hipLaunchKernelGGL(raw_mfma_rand_k, dim3(768), dim3(256), 0, st, out, src, 512); }, 768.0 * 4 * 512 * 16 * 4096.0, 1);
        else
            run("raw MFMA, constant operands", [&] { hipLaunchKernelGGL(raw_mfma_k, dim3(768), dim3(256), 0, st, out, 512, 1.f, 1.f); }, 768.0 * 4 * 512 * 16 * 4096.0, 1);
        return 0;
    }
    printf("F/A: %d WGs ksplit %d/%d/%d   B: %d WGs ksplit %d/%d/%d/%d/%d   D: %d WGs ksplit %d\n", wgs(fa, 3), fa[0].ksplit,
           fa[1].ksplit, fa[2].ksplit, wgs(b, 5), b[0].ksplit, b[1].ksplit, b[2].ksplit, b[3].ksplit, b[4].ksplit, wgs(&dd, 1), dd.ksplit);
#ifdef SET_EXP_STAMPS
    {
        unsigned long long* dst; hipMalloc(&dst, 1024 * 8 * 8);
        std::vector<unsigned long long> hs(1024 * 8);
        const GemmProb* groups[3] = {fa, b, &dd}; const int gn[3] = {3, 5, 1}; const char* gname[3] = {"F/A", "B", "D"};
        for (int g = 0; g < 3; ++g) {
            for (int rep = 0; rep < 3; ++rep) {            // warm, then stamp the third launch of a back-to-back run
                hipMemsetAsync(dst, 0, 1024 * 64, st);
                g_gemm_stamps = nullptr;
                for (int i = 0; i < 5; ++i) gemm_group(groups[g], gn[g], st, nullptr);
                g_gemm_stamps = dst;
                gemm_group(groups[g], gn[g], st, nullptr);
                g_gemm_stamps = nullptr;
                hipStreamSynchronize(st);
            }
            hipMemcpy(hs.data(), dst, 1024 * 64, hipMemcpyDeviceToHost);
            const int nw = wgs(groups[g], gn[g]);
            unsigned long long t0 = ~0ull, t_end = 0;
            for (int w = 0; w < nw; ++w) if (hs[w * 8]) { if (hs[w * 8] < t0) t0 = hs[w * 8]; if (hs[w * 8 + 6] > t_end) t_end = hs[w * 8 + 6]; }
            double s[7] = {0}; double smax[7] = {0}; int cnt = 0;
            for (int w = 0; w < nw; ++w) {
                if (!hs[w * 8]) continue;
                ++cnt;
                for (int i = 0; i < 7; ++i) {
                    const double v = i == 0 ? (double)(hs[w * 8] - t0) : (double)(hs[w * 8 + i] - hs[w * 8 + i - 1]);
                    s[i] += v; if (v > smax[i]) smax[i] = v;
                }
            }
            printf("%-4s stamps over %d WGs (us, mean / max): start-skew %.2f/%.2f  args %.2f/%.2f  tile0+lds %.2f/%.2f  tile2-issue+barrier %.2f/%.2f  "
                   "k-tile0 %.2f/%.2f  rest-of-loop %.2f/%.2f  epilogue %.2f/%.2f   first-start -> last-end %.2f\n", gname[g], cnt,
                   s[0] / cnt / 100, smax[0] / 100, s[1] / cnt / 100, smax[1] / 100, s[2] / cnt / 100, smax[2] / 100, s[3] / cnt / 100, smax[3] / 100,
                   s[4] / cnt / 100, smax[4] / 100, s[5] / cnt / 100, smax[5] / 100, s[6] / cnt / 100, smax[6] / 100, (double)(t_end - t0) / 100);
            // shader clock over the k loop: cycles / wall time, and the MFMA pipe's share (3 resident waves x 16 x 64 cycles per k-tile)
            double cyc = 0, wall = 0;
            for (int w = 0; w < nw; ++w) if (hs[w * 8]) { cyc += (double)(hs[w * 8 + 7] & 0xfffffffffffffffull); wall += (double)(hs[w * 8 + 5] - hs[w * 8 + 3]); }
            printf("     k loop: %.0f cycles mean, clock %.3f GHz;  xcc of WG 0..9:", cyc / cnt, cyc / (wall * 10.0));
            for (int w = 0; w < 10; ++w) printf(" %llu", hs[w * 8 + 7] >> 60);
            printf("\n");
        }
    }
#endif
    if (getenv("CHECK_WREG") || getenv("CHECK_ASM")) {
        const bool chk_asm = getenv("CHECK_ASM") != nullptr;
        // parity of gemm_nt_f32_wreg against gemm_nt_f32<64,64>: same problems, slabs summed on the host in index order
        const GemmProb* groups[3] = {fa, b, &dd}; const int gn[3] = {3, 5, 1}; const char* gname[3] = {"F/A", "B", "D"};
        for (int g = 0; g < 3; ++g) {
            std::vector<std::vector<double>> res[2];
            for (int v = 0; v < 2; ++v) {
                if (chk_asm) g_gemm_asm_force = v; else g_gemm_wreg_force = v;
                hipMemsetAsync(slabs, 0xff, (size_t)off * 4, st);      // NaN-fill: an unwritten element shows
                gemm_group(groups[g], gn[g], st, nullptr);
                hipStreamSynchronize(st);
                for (int i = 0; i < gn[g]; ++i) {
                    const GemmProb& p = groups[g][i];
                    std::vector<double> acc((size_t)p.M * p.N, 0.0);
                    std::vector<float> h((size_t)p.M * p.ldc);
                    for (int s2 = 0; s2 < p.ksplit; ++s2) {
                        hipMemcpy(h.data(), p.C + (size_t)s2 * p.slab_stride, h.size() * 4, hipMemcpyDeviceToHost);
                        for (int r = 0; r < p.M; ++r) for (int c = 0; c < p.N; ++c) acc[(size_t)r * p.N + c] += h[(size_t)r * p.ldc + c];
                    }
                    res[v].push_back(acc);
                }
            }
            g_gemm_wreg_force = -1; g_gemm_asm_force = -1;
            for (int i = 0; i < gn[g]; ++i) {
                double md = 0, mx = 0; size_t bad = 0;
                for (size_t k = 0; k < res[0][i].size(); ++k) {
                    const double a = res[0][i][k], c = res[1][i][k];
                    if (!(a == a) || !(c == c)) { ++bad; continue; }
                    md = fmax(md, fabs(a - c)); mx = fmax(mx, fabs(a));
                }
                printf("check %-4s prob %d (M %d N %d ksplit %d): max |diff| %.3e of max |ref| %.3e, %zu NaN\n", gname[g], i,
                       groups[g][i].M, groups[g][i].N, groups[g][i].ksplit, md, mx, bad);
            }
        }
    }
    run("F/A  fc+gates1+h2h", [&] { gemm_group(fa, 3, st, nullptr); }, flops(fa, 3), 1);
    run("B    att2,tc,cg,x2h_h1", [&] { gemm_group(b, 5, st, nullptr); }, flops(b, 5), 1);
    run("D    x2h ctx", [&] { gemm_group(&dd, 1, st, nullptr); }, flops(&dd, 1), 1);
    run("rotation F/A,B,D", [&] { gemm_group(fa, 3, st, nullptr); gemm_group(b, 5, st, nullptr); gemm_group(&dd, 1, st, nullptr); },
        flops(fa, 3) + flops(b, 5) + flops(&dd, 1), 3);
    run("big  M=2560 N=4096 K=1024", [&] { gemm_group(&big, 1, st, nullptr); }, flops(&big, 1), 1);
    // fixed cost vs per-k-tile cost: 768 workgroups (N = 4096, ksplit 6) running 8 / 16 / 32 / 64 k-tiles each
    {
        float* Wl = dev_rand((size_t)4096 * 12288, 30); float* Al = dev_rand((size_t)B * 12288, 31);
        float* sl = take((size_t)6 * B * 4096);
        for (int kper : {1, 2, 4, 8, 16, 32, 64}) {
            GemmProb p = slab(sl, B, 4096); p.add(Al, 12288, Wl, 12288, 6 * kper * 32); p.ksplit = 6;
            char nm[64]; snprintf(nm, sizeof nm, "768 WGs x %d k-tiles", kper);
            run(nm, [&] { gemm_group(&p, 1, st, nullptr); }, flops(&p, 1), 1);
        }
    }
    {
        float* out; hipMalloc(&out, 768 * 256 * 4);
        run("empty kernel 768 WGs", [&] { hipLaunchKernelGGL(empty_k, dim3(768), dim3(256), 0, st, out, 0); }, 0.0, 1);
    }
    // raw MFMA issue rate, sustained, 3 waves per SIMD (the GEMM's residency), 4 independent accumulators per wave
    {
        float* out; hipMalloc(&out, 768 * 256 * 4);
        const int inner = 512;      // 512 x 16 MFMAs per wave per launch = the F/A launch's MFMA count per wave x 16
        run("raw MFMA 768 WGs", [&] { hipLaunchKernelGGL(raw_mfma_k, dim3(768), dim3(256), 0, st, out, inner, 1.f, 1.f); },
            768.0 * 4 * inner * 16 * 4096.0, 1);
        print_raw_clock("constant operands");
        float* src = dev_rand(8192, 77);
        run("raw MFMA, random operands", [&] { hipLaunchKernelGGL(raw_mfma_rand_k, dim3(768), dim3(256), 0, st, out, src, inner); },
            768.0 * 4 * inner * 16 * 4096.0, 1);
        print_raw_clock("random operands");
    }
    return 0;
}
