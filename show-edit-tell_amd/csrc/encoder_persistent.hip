// Weights-stationary persistent caption encoder: the whole recurrence of CaptionEncoderC.forward (editnet.py:331-338) —
// T dependent steps of  gates = h W_hh^T + (hoisted x W_xh^T + b),  LSTM cell update, H / Mem stores — in ONE launch.
//
// The per-step launch (gemm_fused_k<..., ENCLSTM>) costs ~19.5 us for 1.07 GFLOP (6.8 us of matrix-pipe time): every
// launch re-streams the 16.8 MB of W_hh through LDS and pays a launch boundary + start-up + cross-wave reduction.  Here
//   * W_hh is split over G = D / 4 workgroups (256 at D = 1024, one per CU): a workgroup owns 4 hidden units = 16 gate rows
//     of W_hh (64 KB at D = 1024) and keeps them IN REGISTERS for all T steps — each of its 4 waves holds the 16 rows over
//     one quarter of K (64 VGPRs per lane) as ready-made B operands of v_mfma_f32_16x16x4_f32;
//   * per step a wave contracts every 16-row tile of h (read as 16-byte pieces straight from L2: one load per lane feeds
//     four MFMAs, K permuted identically in A and B) against its K quarter, the four partial tiles are added through LDS
//     in fixed wave order, and 2 (row, unit) pairs per thread run the cell update with c and h kept in registers;
//   * rows are visited longest first (perm / nactive, as in the per-step path): 16-row tiles whose rows have all finished
//     skip their loads and MFMAs, finished rows carry their state;
//   * the new h (2 KB per workgroup) is published with write-through (sc1) stores and the steps are separated by a grid
//     barrier: per-shard arrival counters (blockIdx % 8), a top counter, per-shard generation words polled by ONE lane with
//     relaxed loads + s_sleep.  h is READ with sc1 loads as well (raw buffer loads, aux = sc1), so both sides of the exchange
//     go to the coherence point and neither an L2 write-back nor an invalidate is needed — only the drain of a wave's own
//     sc1 stores before it arrives (MI355X_MICROARCH.md, barrier-xcd / Guideline 16).  Sharding is by block id, not by XCC
//     id: correct for ANY placement; faster when block b runs on XCD b % 8, which is what the dispatcher does.
// Residency: the barrier needs all G workgroups resident at once.  G <= 256 workgroups of 256 threads, <= 32 KB of LDS; the
// default small-batch instantiation (B <= 32) stays within 256 registers (__launch_bounds__(256, 2)), so any two instances
// fit the chip together (two processes sharing one GPU, e.g. the world-size-2 tests, cannot starve each other; the opt-in
// B <= 128 instantiation takes a wave slot per SIMD alone); the host side serialises instances of this kernel across
// streams with an event chain (at most one runs at a time in this process), other kernels sharing the chip always finish on
// their own, and every spin is bounded (a timeout raises the status word instead of hanging the queue).
#include <fcntl.h>
#include <sys/file.h>
#include <unistd.h>
#include <cstdio>
#include <mutex>
#include "set_common.h"
#include "grid_barrier.h"

namespace set {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const f32x4 __attribute__((address_space(1)))* gptr4;

// one recurrence ("direction"): EditNet's encoder has one, DCNet's bidirectional encoder two that share the launch
struct PEncDir {
    const float* w_hh;               // (4D, D), gate q of unit u at row q*D + u
    const float* xg;                 // hoisted input projection: token table rows (seq != NULL) or (B, T, 4D)
    const float* b_extra;            // (4D) added to xg, or NULL
    float* hbuf0; float* hbuf1;      // (B, D) ping-pong; hbuf0 = initial state (zeros)
    int out_col0;                    // first output column of this direction in H / Mem rows
    int reverse;                     // 1: step t visits position len - 1 - t (nn.LSTM's backward direction on a packed batch)
};
struct PEncArgs {
    PEncDir dir[2];
    int ndir;
    long long ld_xg_row, ld_xg_t;
    const int64_t* lens;             // (B)
    const int64_t* seq;              // (B, seq_T) token ids when xg is a token table
    int seq_T, seq_V;
    float* H; float* Mem;            // (B, T, ld_out_t) outputs (zero-initialised by the caller); Mem may be NULL
    long long ld_out_b, ld_out_t;
    const int* perm; const int* nactive;
    unsigned* bar;                   // GBAR_WORDS zeroed words
    unsigned* status;                // set to 1 on a barrier timeout
    unsigned* fault;                 // host-mapped, sticky: a timeout of ANY launch of this process on this device (never cleared by a launch)
    unsigned spin_limit;             // polls of one barrier wait before it gives up
    int test_stall;                  // test hook (SET_PENC_TEST_STALL): workgroup 0 never arrives at a barrier
    int B, D, T;
};

__device__ __forceinline__ float psigm(float x) { return 1.f / (1.f + expf(-x)); }

// acc[0..N) += h tiles (rows arow[rt], this wave's K quarter at hk) x the stationary W operands; loads run one k-block ahead
template <int NT, int KB, int N>
__device__ __forceinline__ void penc_contract(f32x4 (&acc)[NT], __amdgpu_buffer_rsrc_t hrs, int koff, const int (&aoff)[NT],
                                              const f32x4 (&wreg)[KB]) {
    f32x4 a0[N], a1[N];
#pragma unroll
    for (int rt = 0; rt < N; ++rt) a0[rt] = ld_sc1(hrs, aoff[rt] + koff);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        f32x4(&cur)[N] = (kb & 1) ? a1 : a0;
        f32x4(&nxt)[N] = (kb & 1) ? a0 : a1;
        if (kb + 1 < KB) {
#pragma unroll
            for (int rt = 0; rt < N; ++rt) nxt[rt] = ld_sc1(hrs, aoff[rt] + koff + 64 * (kb + 1));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int rt = 0; rt < N; ++rt)
                acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[rt][j], wreg[kb][j], acc[rt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);           // keep the loads one k-block ahead, not sixteen (register budget)
    }
}

// NT = 16-row tiles held (B <= 16 NT), KB = 16-wide k-blocks per wave (D = 64 KB)
template <int NT, int KB>
__global__ void __launch_bounds__(256, (NT <= 2 ? 2 : 1)) encoder_persistent_k(const PEncArgs P) {
    constexpr int PAIRS = NT >= 4 ? NT / 4 : 1;      // (row, unit) pairs per thread: 16 NT rows x 4 units over 256 threads
    extern __shared__ __attribute__((aligned(16))) float red[];       // [4 waves][16 NT rows][16 cols]
    const int tid = threadIdx.x, lane = tid & 63, kq = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int D = P.D, B = P.B, K4 = D >> 2;
    const int per_dir = D >> 2;                      // workgroups per direction
    const PEncDir& Q = P.dir[(int)blockIdx.x >= per_dir ? 1 : 0];
    const int unit0 = ((int)blockIdx.x % per_dir) * 4;
    const unsigned G = gridDim.x, ns = G < 8u ? G : 8u;
    const int shard = (int)(blockIdx.x % ns);
    const unsigned pop = G / ns + ((unsigned)shard < G % ns ? 1u : 0u);

    // ---- stationary B operands: W_hh[gate row of column r][kq*K4 + 16 kb + 4 g .. +3]
    f32x4 wreg[KB];
    {
        const float* wrow = Q.w_hh + ((long long)(r >> 2) * D + unit0 + (r & 3)) * D + kq * K4 + 4 * g;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) wreg[kb] = *(gptr4)(wrow + 16 * kb);
    }
    // rows of this lane's A fragments (sorted position 16 rt + r -> batch row), clamped
    int aoff[NT];                                    // byte offset of the row inside an h buffer (B * D * 4 < 2^31: B <= 256)
#pragma unroll
    for (int rt = 0; rt < NT; ++rt) { const int p = 16 * rt + r; aoff[rt] = P.perm[p < B ? p : B - 1] * D * 4; }
    const int hbytes = B * D * 4;
    const __amdgpu_buffer_rsrc_t hrs0 = __builtin_amdgcn_make_buffer_rsrc((void*)Q.hbuf0, 0, hbytes, 0x00027000);
    const __amdgpu_buffer_rsrc_t hrs1 = __builtin_amdgcn_make_buffer_rsrc((void*)Q.hbuf1, 0, hbytes, 0x00027000);
    const int koff = (kq * K4 + 4 * g) * 4;
    // (row, unit) pairs of this thread for the cell update: pair p = tid + 256 i -> sorted row p >> 2, unit p & 3
    int prow[PAIRS], plen[PAIRS];
    float creg[PAIRS], hreg[PAIRS];
#pragma unroll
    for (int i = 0; i < PAIRS; ++i) {
        const int rs = (tid + 256 * i) >> 2;
        prow[i] = rs < B ? P.perm[rs] : -1;
        plen[i] = rs < B ? (int)P.lens[prow[i]] : 0;
        creg[i] = 0.f;
        hreg[i] = rs < B ? Q.hbuf0[(long long)prow[i] * D + unit0 + ((tid + 256 * i) & 3)] : 0.f;
    }

    // the hoisted input projection of step t for this thread's pairs: it does not depend on the recurrence, so step t + 1's
    // is requested between the arrival at step t's barrier and the wait (off the dependent chain)
    float eg[PAIRS][4];
#define PENC_GATHER(TT)                                                                                   \
    _Pragma("unroll") for (int i = 0; i < PAIRS; ++i) {                                                   \
        const int u_ = unit0 + ((tid + 256 * i) & 3);                                                     \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) eg[i][q] = 0.f;                                     \
        if (prow[i] >= 0 && (TT) < plen[i]) {                                                             \
            const float* xr_;                                                                             \
            const int pos_ = Q.reverse ? plen[i] - 1 - (TT) : (TT);                                       \
            if (P.seq) {                                                                                  \
                long long tok_ = P.seq[(long long)prow[i] * P.seq_T + pos_];                              \
                tok_ = tok_ < 0 ? 0 : (tok_ >= P.seq_V ? P.seq_V - 1 : tok_);   /* same clamp as embed_relu_k */ \
                xr_ = Q.xg + tok_ * P.ld_xg_row;                                                          \
            } else {                                                                                      \
                xr_ = Q.xg + (long long)prow[i] * P.ld_xg_row + (long long)pos_ * P.ld_xg_t;              \
            }                                                                                             \
            _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                 \
                eg[i][q] = xr_[q * D + u_] + (Q.b_extra ? Q.b_extra[q * D + u_] : 0.f);                   \
        }                                                                                                 \
    }
    PENC_GATHER(0);
    for (int t = 0; t < P.T; ++t) {
        const int nact = P.nactive[t];
        const int nt_act = (nact + 15) >> 4;                       // wave-uniform: tiles with at least one live row
        const __amdgpu_buffer_rsrc_t hrs = (t & 1) ? hrs1 : hrs0;
        float* hout = (t & 1) ? Q.hbuf0 : Q.hbuf1;
        // ---- contraction of the live row tiles against this wave's K quarter (static tile counts: 2, 4, 6, ... NT; the
        // tiles beyond nt_act inside a variant hold finished rows only and their results are never read)
        f32x4 acc[NT];
#pragma unroll
        for (int rt = 0; rt < NT; ++rt) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (nt_act > 0) {
            if constexpr (NT >= 16) { if (nt_act > 12) { penc_contract<NT, KB, 16>(acc, hrs, koff, aoff, wreg); goto contracted; } }
            if constexpr (NT >= 12) { if (nt_act > 8) { penc_contract<NT, KB, 12>(acc, hrs, koff, aoff, wreg); goto contracted; } }
            if constexpr (NT >= 8) { if (nt_act > 6) { penc_contract<NT, KB, 8>(acc, hrs, koff, aoff, wreg); goto contracted; } }
            if constexpr (NT >= 6) { if (nt_act > 4) { penc_contract<NT, KB, 6>(acc, hrs, koff, aoff, wreg); goto contracted; } }
            if constexpr (NT >= 4) { if (nt_act > 2) { penc_contract<NT, KB, 4>(acc, hrs, koff, aoff, wreg); goto contracted; } }
            if (nt_act > 1) penc_contract<NT, KB, 2>(acc, hrs, koff, aoff, wreg);
            else penc_contract<NT, KB, 1>(acc, hrs, koff, aoff, wreg);
        }
    contracted:
        // ---- partial tiles -> LDS: C/D map of 16x16 MFMA: col = lane & 15, row = 4 (lane >> 4) + reg
#pragma unroll
        for (int rt = 0; rt < NT; ++rt)
            if (rt < nt_act) {
#pragma unroll
                for (int e = 0; e < 4; ++e) red[((kq * NT + rt) * 16 + 4 * g + e) * 16 + r] = acc[rt][e];
            }
        __syncthreads();
        // ---- cell update of this thread's pairs; h published write-through first, the barrier arrival next, and only then
        // the H / Mem stores (plain: read after the launch) and the gather of the next step's input projection
        float hn_[PAIRS], cn_[PAIRS];
        bool live_[PAIRS];
#pragma unroll
        for (int i = 0; i < PAIRS; ++i) {
            const int p = tid + 256 * i, rs = p >> 2, u = p & 3;
            live_[i] = prow[i] >= 0 && t < plen[i];
            if (prow[i] < 0) continue;
            if (live_[i]) {
                float gq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int o = rs * 16 + q * 4 + u;
                    gq[q] = (((red[o] + red[NT * 256 + o]) + red[2 * NT * 256 + o]) + red[3 * NT * 256 + o]) + eg[i][q];
                }
                const float ai = psigm(gq[0]), af = psigm(gq[1]), ag = tanhf(gq[2]), ao = psigm(gq[3]);
                const float cn = af * creg[i] + ai * ag;
                const float hn = ao * tanhf(cn);
                creg[i] = cn;
                hreg[i] = hn;
            }
            hn_[i] = hreg[i]; cn_[i] = creg[i];
            __hip_atomic_store(hout + (long long)prow[i] * D + unit0 + u, hreg[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const bool more = t + 1 < P.T;
        if (more && !(P.test_stall && blockIdx.x == 0)) gbar_arrive(P.bar, (unsigned)(t + 1), shard, pop, ns);
#pragma unroll
        for (int i = 0; i < PAIRS; ++i)
            if (live_[i]) {
                const int pos = Q.reverse ? plen[i] - 1 - t : t;
                const long long o = (long long)prow[i] * P.ld_out_b + (long long)pos * P.ld_out_t + Q.out_col0 + unit0 + ((tid + 256 * i) & 3);
                P.H[o] = hn_[i];
                if (P.Mem) P.Mem[o] = cn_[i];
            }
        if (more) {
            PENC_GATHER(t + 1);
            gbar_wait(P.bar, (unsigned)(t + 1), shard, P.status, P.fault, P.spin_limit);
        }
    }
#undef PENC_GATHER
    // ---- a barrier of this launch timed out (not every workgroup was resident, or one was held up beyond the bound): the
    // recurrence ran on stale h somewhere.  Never hand that out as a result: every workgroup that sees the status word
    // overwrites its units of both h buffers (-> final_hidden -> every later gate product) and of H / Mem at position 0
    // with NaN, so the decode that consumes this encoder returns NaN log-probabilities / scores instead of plausible
    // numbers; the host raises SET_ERR_FAULT at its next call (sticky host-mapped fault word) and stops using this kernel.
    __shared__ unsigned s_bad;
    if (tid == 0) s_bad = __hip_atomic_load(P.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_bad) {
        const float qnan = __builtin_nanf("");
        for (int i = tid; i < B * 4; i += 256) {
            const int b = i >> 2, u = unit0 + (i & 3);
            Q.hbuf0[(long long)b * D + u] = qnan;
            Q.hbuf1[(long long)b * D + u] = qnan;
            const long long o = (long long)b * P.ld_out_b + Q.out_col0 + u;
            P.H[o] = qnan;
            if (P.Mem) P.Mem[o] = qnan;
        }
    }
}

static std::mutex g_penc_mutex;
static hipEvent_t g_penc_event[64] = {};
// sticky fault word per device: host-mapped pinned memory the kernel writes on a barrier timeout and the host reads (without
// any synchronisation) at its next call.  g_penc_disabled: a fault was seen -> the per-step path from then on.
static unsigned* g_penc_fault_host[64] = {};
static unsigned* g_penc_fault_dev[64] = {};
static bool g_penc_disabled[64] = {};
static int g_penc_capacity[64][9] = {};          // resident workgroups the device admits per kernel instantiation (0 = not asked yet)

// ---- one owner PROCESS per device.  A persistent grid needs all of its workgroups resident at once; the mutex + event chain
// above order the launches of one process, but two processes sharing a GPU (several ranks on one device, a second job) can
// each get half a grid resident and wait for the rest until the 1-s time-out poisons both.  The first process that launches a
// persistent kernel on a device takes an advisory lock on a per-device file (named after the PCI bus id, so ordinals remapped by
// HIP_VISIBLE_DEVICES agree) and keeps it until it exits (the kernel releases it even when the process dies); every other
// process gets SET_ERR_UNSUPPORTED from the guard — the per-step kernels, no fault, nothing disabled — and asks again at its
// next call, so it takes over once the owner is gone.  SET_PERSISTENT_IPC_LOCK=0 switches the check off (exclusive devices);
// a lock directory that cannot be written is treated the same way.
static int g_penc_lock_fd[64];                   // 0 = not asked yet, > 0 = owner (open descriptor + 1), -1 = no locking
static bool penc_process_owns(int dev) {
    static const int on = env_int("SET_PERSISTENT_IPC_LOCK", 1);
    if (!on || g_penc_lock_fd[dev] == -1) return true;
    if (g_penc_lock_fd[dev] > 0) return true;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus) - 1, dev) != hipSuccess) { (void)hipGetLastError(); snprintf(bus, sizeof(bus), "ord%d", dev); }
    for (char* c = bus; *c; ++c) if (*c == ':' || *c == '/' || *c == '.') *c = '_';
    const char* dir = getenv("SET_PERSISTENT_LOCK_DIR");
    char path[256];
    snprintf(path, sizeof(path), "%s/set_hip_persistent_%s.lock", dir && *dir ? dir : "/tmp", bus);
    int fd = open(path, O_CREAT | O_RDWR | O_CLOEXEC, 0666);
    if (fd < 0) fd = open(path, O_RDONLY | O_CLOEXEC);        // another user's lock file (created 0644 under the usual umask): flock needs no write access
    if (fd < 0) { g_penc_lock_fd[dev] = -1; return true; }
    if (flock(fd, LOCK_EX | LOCK_NB) != 0) { close(fd); return false; }       // another process owns the device's persistent launches
    g_penc_lock_fd[dev] = fd + 1;
    return true;
}

PersistentGuard::PersistentGuard() : rc(SET_OK), dev(0), fault(nullptr), locked(false) {
    if (hipGetDevice(&dev) != hipSuccess) { rc = SET_ERR_HIP; return; }
    dev &= 63;
    g_penc_mutex.lock();
    locked = true;
    if (!penc_process_owns(dev)) { rc = SET_ERR_UNSUPPORTED; return; }
    if (!g_penc_fault_host[dev]) {
        void* hp = nullptr; void* dp = nullptr;
        if (hipHostMalloc(&hp, 64, hipHostMallocMapped) != hipSuccess) { rc = SET_ERR_HIP; return; }
        *(volatile unsigned*)hp = 0u;
        if (hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) { rc = SET_ERR_HIP; return; }
        g_penc_fault_host[dev] = (unsigned*)hp; g_penc_fault_dev[dev] = (unsigned*)dp;
    }
    if (*(volatile unsigned*)g_penc_fault_host[dev]) {
        // an earlier launch's barrier timed out (its outputs were poisoned with NaN on the device): say so ONCE, loudly, and
        // never launch a persistent kernel again in this process on this device — the caller's retry runs the per-step kernels
        *(volatile unsigned*)g_penc_fault_host[dev] = 0u;
        g_penc_disabled[dev] = true;
        rc = SET_ERR_FAULT;
        return;
    }
    if (g_penc_disabled[dev]) { rc = SET_ERR_UNSUPPORTED; return; }
    fault = g_penc_fault_dev[dev];
}
PersistentGuard::~PersistentGuard() { if (locked) g_penc_mutex.unlock(); }
static int penc_serialise() { static const int v = env_int("SET_ENC_PERSISTENT_SERIALISE", 1); return v; }
int PersistentGuard::serialise(hipStream_t s) {
    if (!penc_serialise()) return SET_OK;
    if (!g_penc_event[dev]) SET_HIP_TRY(hipEventCreateWithFlags(&g_penc_event[dev], hipEventDisableTiming));
    else SET_HIP_TRY(hipStreamWaitEvent(s, g_penc_event[dev], 0));
    return SET_OK;
}
int PersistentGuard::launched(hipStream_t s) {
    if (penc_serialise() && g_penc_event[dev]) SET_HIP_TRY(hipEventRecord(g_penc_event[dev], s));
    return SET_OK;
}
unsigned PersistentGuard::spin_limit() const {
    static const int us = env_int("SET_PENC_TIMEOUT_US", (int)GBAR_TIMEOUT_US);
    const unsigned v = us > 0 ? (unsigned)us : GBAR_TIMEOUT_US;
    return v > 40000000u ? 4000000000u : v * 100u;             // ticks of the 100-MHz counter
}
int persistent_lds_limit() {
    static int limit[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    dev &= 63;
    if (limit[dev] == 0) {
        int v = 0;
        limit[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && v > 0) ? v : -1;
    }
    return limit[dev] > 0 ? limit[dev] : 0;
}
int PersistentGuard::set_lds(const void* kernel, int bytes, bool (&done)[64]) {
    if (done[dev]) return SET_OK;
    const int lim = persistent_lds_limit();
    if (lim <= 0 || bytes > lim) return SET_ERR_UNSUPPORTED;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return SET_ERR_UNSUPPORTED;
    }
    done[dev] = true;
    return SET_OK;
}
int PersistentGuard::test_stall() const { static const int v = env_int("SET_PENC_TEST_STALL", 0); return v; }
// persistent launches are not to be tried on the current device: a barrier timed out earlier in this process, or another
// process owns them (penc_process_owns; asked again at every call, so a device whose owner exited is taken over).  Callers ask
// BEFORE they prepare anything for a persistent launch (hoisted products, the beam entry point's prologue).
bool persistent_disabled() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    dev &= 63;
    std::lock_guard<std::mutex> lk(g_penc_mutex);
    return g_penc_disabled[dev] || !penc_process_owns(dev);
}

size_t persistent_encoder_bar_bytes() { return sizeof(unsigned) * (GBAR_WORDS + GBAR_STRIDE); }

bool persistent_encoder_ok(int B, int D, int T) {
    // default: small batches only (SET_ENC_PERSISTENT_MAXB rows).  Measured (round 3): a step of the dependent chain —
    // drain the sc1 stores, two-level arrival, poll, re-read h from L2 — costs ~6-9 us, the per-step launch ~14 us + its
    // matrix work: at B = 4 the recurrence takes 174 us instead of 280, at B = 128 385 us instead of 400 while the spinning
    // workgroups take CU slots from the other batches in flight (headline -4 %)
    static const int on = env_int("SET_ENC_PERSISTENT", 1);
    static const int maxb = env_int("SET_ENC_PERSISTENT_MAXB", 32);
    if (B > maxb) return false;
    if (persistent_disabled()) return false;                 // a barrier timed out earlier
    // 16-row tiles held in registers: 8 (B <= 128) fit the 256-register budget of two co-resident instances at D = 512 / 1024;
    // 16 tiles would spill there (only the reduced test dimension takes them)
    return on && B >= 1 && T >= 1 && ((B <= 128 && (D == 512 || D == 1024)) || (B <= 256 && D == 64));
}

// slot of an instantiation in g_penc_capacity
template <int NT, int KB> constexpr int penc_slot() { return (NT == 2 ? 0 : NT == 8 ? 1 : 2) * 3 + (KB == 16 ? 0 : KB == 8 ? 1 : 2); }

// configure the instantiation once and say whether the device admits `grid` workgroups of it at the same time.
// The grid barrier needs every workgroup resident: ask the runtime how many workgroups of THIS instantiation one CU admits
// (registers, LDS) and how many CUs the device has (a partitioned or smaller part reports fewer); when the grid does not fit
// the caller takes the per-step kernels.  (MI355X_MICROARCH.md: the API can be one high per CU only at >= 7 workgroups per
// CU; these kernels sit at 1-2 by their register budget.)
template <int NT, int KB>
static int penc_fits(PersistentGuard& guard, int grid, int dev, bool* fits) {
    const int lds = 4 * NT * 256 * (int)sizeof(float);
    static bool configured[64] = {};
    *fits = false;
    if (guard.set_lds(reinterpret_cast<const void*>(&encoder_persistent_k<NT, KB>), lds, configured) != SET_OK) return SET_OK;
    static_assert(penc_slot<NT, KB>() < 9, "capacity table");
    int& cap = g_penc_capacity[dev][penc_slot<NT, KB>()];
    if (cap == 0) {
        int per_cu = 0, cus = 0;
        SET_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&encoder_persistent_k<NT, KB>),
                                                                 256, (size_t)lds));
        SET_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        if (per_cu > 8) per_cu = 8;
        cap = per_cu * cus;
        if (cap <= 0) cap = -1;
        const int forced = env_int("SET_PENC_TEST_CAPACITY", 0);      // test hook: pretend the device admits this many
        if (forced > 0) cap = forced;
    }
    *fits = grid <= cap;
    return SET_OK;
}

template <int NT, int KB>
static int launch_penc(const PEncArgs& P, int grid, hipStream_t s) {
    const int lds = 4 * NT * 256 * (int)sizeof(float);
    hipLaunchKernelGGL((encoder_persistent_k<NT, KB>), dim3(grid), dim3(256), lds, s, P);
    SET_LAUNCH_CHECK();
    return SET_OK;
}

// the whole recurrence(s); every direction's hbuf0 must hold its initial state (zeros), H / Mem must be zero-filled, bar is
// scratch of persistent_encoder_bar_bytes().  A direction's final state is left in (T & 1) ? hbuf1 : hbuf0.
int persistent_encoder_dirs(const PEncDirHost* dirs, int ndir, long long ld_xg_row, long long ld_xg_t, const int64_t* lens,
                            const int64_t* seq, int seq_T, int seq_V, float* H, float* Mem, long long ld_out_b,
                            long long ld_out_t, const int* perm, const int* nactive, void* bar, int B, int D, int T,
                            hipStream_t s) {
    if (!persistent_encoder_ok(B, D, T) || !perm || !nactive || !bar || ndir < 1 || ndir > 2) return SET_ERR_UNSUPPORTED;
    PEncArgs P{};
    for (int i = 0; i < ndir; ++i) {
        P.dir[i].w_hh = dirs[i].w_hh; P.dir[i].xg = dirs[i].xg; P.dir[i].b_extra = dirs[i].b_extra;
        P.dir[i].hbuf0 = dirs[i].hbuf0; P.dir[i].hbuf1 = dirs[i].hbuf1; P.dir[i].out_col0 = dirs[i].out_col0;
        P.dir[i].reverse = dirs[i].reverse;
    }
    if (ndir == 1) P.dir[1] = P.dir[0];
    P.ndir = ndir; P.ld_xg_row = ld_xg_row; P.ld_xg_t = ld_xg_t; P.lens = lens;
    P.seq = seq; P.seq_T = seq_T; P.seq_V = seq_V > 0 ? seq_V : 1; P.H = H; P.Mem = Mem;
    P.ld_out_b = ld_out_b; P.ld_out_t = ld_out_t; P.perm = perm; P.nactive = nactive;
    P.bar = (unsigned*)bar; P.status = (unsigned*)bar + GBAR_WORDS; P.B = B; P.D = D; P.T = T;
    // at most ONE persistent kernel runs at a time in this process (grid_barrier.h PersistentGuard: lock, sticky fault word,
    // event chain); see the residency note at the top of the file
    PersistentGuard guard;
    if (guard.rc != SET_OK) return guard.rc;
    const int dev = guard.dev;
    P.spin_limit = guard.spin_limit();
    P.test_stall = guard.test_stall();
    P.fault = guard.fault;
    const int nt = (B + 15) / 16, grid = ndir * (D / 4);
    // one of nine instantiations: 0/1/2 = <2,*>/<8,*>/<16,*>, KB by D
    const int inst = D == 1024 ? (nt <= 2 ? 0 : 1) : D == 512 ? (nt <= 2 ? 2 : 3) : (nt <= 2 ? 4 : (nt <= 8 ? 5 : 6));
    bool fits = false;
    switch (inst) {
        case 0: SET_TRY((penc_fits<2, 16>(guard, grid, dev, &fits))); break;
        case 1: SET_TRY((penc_fits<8, 16>(guard, grid, dev, &fits))); break;
        case 2: SET_TRY((penc_fits<2, 8>(guard, grid, dev, &fits))); break;
        case 3: SET_TRY((penc_fits<8, 8>(guard, grid, dev, &fits))); break;
        case 4: SET_TRY((penc_fits<2, 1>(guard, grid, dev, &fits))); break;
        case 5: SET_TRY((penc_fits<8, 1>(guard, grid, dev, &fits))); break;
        default: SET_TRY((penc_fits<16, 1>(guard, grid, dev, &fits))); break;
    }
    if (!fits) return SET_ERR_UNSUPPORTED;          // (nothing has been touched: the caller runs the per-step kernels)
    ProfScope ps("persistent_encoder", s, 8.0 * ndir * B * D * D * T, 4.0 * ndir * (4.0 * D * D + 8.0 * B * D * T));
    SET_TRY(guard.serialise(s));
    SET_HIP_TRY(hipMemsetAsync(bar, 0, persistent_encoder_bar_bytes(), s));
    int rc;
    switch (inst) {
        case 0: rc = launch_penc<2, 16>(P, grid, s); break;
        case 1: rc = launch_penc<8, 16>(P, grid, s); break;
        case 2: rc = launch_penc<2, 8>(P, grid, s); break;
        case 3: rc = launch_penc<8, 8>(P, grid, s); break;
        case 4: rc = launch_penc<2, 1>(P, grid, s); break;
        case 5: rc = launch_penc<8, 1>(P, grid, s); break;
        default: rc = launch_penc<16, 1>(P, grid, s); break;
    }
    if (rc == SET_OK) SET_TRY(guard.launched(s));
    return rc;
}

int persistent_encoder(const float* w_hh, const float* xg, long long ld_xg_row, long long ld_xg_t, const float* b_extra,
                       const int64_t* lens, const int64_t* seq, int seq_T, int seq_V, float* hbuf0, float* hbuf1, float* H,
                       float* Mem, long long ld_out_b, long long ld_out_t, const int* perm, const int* nactive, void* bar,
                       int B, int D, int T, hipStream_t s) {
    PEncDirHost d{w_hh, xg, b_extra, hbuf0, hbuf1, 0, 0};
    return persistent_encoder_dirs(&d, 1, ld_xg_row, ld_xg_t, lens, seq, seq_T, seq_V, H, Mem, ld_out_b, ld_out_t, perm, nactive,
                                   bar, B, D, T, s);
}

}  // namespace set
