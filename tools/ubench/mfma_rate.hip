// Microbenchmark: issue rate of v_mfma_f32_32x32x2_f32 under the dependency patterns used by
// gemm_nt_f32 (hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int PATTERN>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a0, float b0) {
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
        if (PATTERN == 0) {        // 2 accumulators, runs of 4 dependent (as in the GEMM)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
            }
        } else if (PATTERN == 1) { // 2 accumulators alternating
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
            }
        } else {                   // 4 accumulators round robin
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
            }
        }
    }
    float s = 0;
    for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int P>
void run(const char* name, int blocks, int threads) {
    float* out; hipMalloc(&out, blocks * threads * sizeof(float));
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<P>, dim3(blocks), dim3(threads), 0, 0, out, 10, 1.f, 1.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<P>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.f, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mf = (double)blocks * (threads / 64) * iters * 32.0;
    double tf = mf * 4096.0 / (ms * 1e-3) / 1e12;
    double cyc = ms * 1e-3 * 2.4e9 / (iters * 32.0) / ((double)(threads / 64) * blocks / 1024.0);
    printf("%-34s blocks=%4d thr=%3d  %8.3f ms  %7.2f TFLOP/s  (%.1f cyc/MFMA/SIMD @2.4GHz)\n", name, blocks, threads, ms, tf, cyc);
    hipFree(out);
}

int main() {
    run<0>("2 acc, runs of 4 dependent", 256, 256);
    run<1>("2 acc, alternating", 256, 256);
    run<2>("4 acc, round robin", 256, 256);
    run<0>("2 acc, runs of 4 (2 waves/SIMD)", 512, 256);
    run<2>("4 acc, round robin (2 waves/SIMD)", 512, 256);
    return 0;
}
