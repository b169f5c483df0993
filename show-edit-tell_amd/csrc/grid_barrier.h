// Grid barrier of the persistent kernels (encoder_persistent.hip, decode_persistent.hip) and the host-side guard that
// keeps them safe: at most one persistent launch at a time in this process, a sticky host-mapped fault word, a device that
// once timed out is never given another persistent launch.
//
// Barrier: per-shard arrival counters (blockIdx % 8), a top counter, per-shard generation words polled by ONE lane per
// workgroup with relaxed loads + s_sleep.  Data crossing a barrier is stored AND loaded at the coherence point (sc1 both
// sides: MI355X_MICROARCH.md, barrier-xcd / Guideline 16), so no release or acquire fence is needed — only the drain of a
// wave's own sc1 stores before it arrives.  Sharding is by block id, not by XCC id: correct for ANY placement; faster when
// block b runs on XCD b % 8, which is what the dispatcher does.  Every spin is bounded: a timeout raises the launch's status
// word and the process-wide fault word instead of hanging the queue.
#pragma once
#include "set_common.h"

namespace set {

constexpr int GBAR_STRIDE = 32;                     // unsigned words between two barrier words (128 bytes)
constexpr int GBAR_WORDS = (8 + 1 + 8) * GBAR_STRIDE;
// Every wait is bounded in WALL-CLOCK time (s_memrealtime, the constant 100-MHz counter), not in poll iterations: a poll is a
// memory round trip whose duration depends on what else runs, a bound in iterations is not a bound.
constexpr unsigned GBAR_TIMEOUT_US = 1000000u;      // default bound of one wait: 1 s (SET_PENC_TIMEOUT_US overrides it)
__device__ __forceinline__ unsigned long long gb_now() { return __builtin_amdgcn_s_memrealtime(); }
inline size_t grid_barrier_bytes() { return sizeof(unsigned) * (GBAR_WORDS + GBAR_STRIDE); }   // + the status word's line

typedef float gb_f32x4 __attribute__((ext_vector_type(4)));

// epoch = 1, 2, ... within the launch; `pop` = workgroups of this shard, `ns` = shards.  Split in two so that work which does
// not depend on the other workgroups (stores read after the launch, requests for loop-invariant operands) sits between
// the arrival and the wait.
__device__ __forceinline__ void gbar_arrive(unsigned* bar, unsigned epoch, int shard, unsigned pop, unsigned ns) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every storing wave: its write-through stores are out
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned* mine = bar + shard * GBAR_STRIDE;
        const unsigned prev = __hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev + 1u == epoch * pop) {                           // last arrival of this shard
            const unsigned p2 = __hip_atomic_fetch_add(bar + 8 * GBAR_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (p2 + 1u == epoch * ns)                            // last shard: release every shard's pollers
                for (unsigned s = 0; s < ns; ++s)
                    __hip_atomic_store(bar + (9 + s) * GBAR_STRIDE, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
__device__ __forceinline__ void gbar_wait(unsigned* bar, unsigned epoch, int shard, unsigned* status, unsigned* fault,
                                          unsigned limit) {
    if (threadIdx.x == 0) {
        unsigned* gen = bar + (9 + shard) * GBAR_STRIDE;
        unsigned spins = 0;
        unsigned long long t0 = 0;
        // bounded spin (`limit` = ticks of the 100-MHz counter); a timeout is sticky (later barriers of this launch do not wait
        // again) and raises the status word
        while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
            __builtin_amdgcn_s_sleep(1);
            if (spins == 0u) t0 = gb_now();
            if ((++spins & 1023u) == 0u && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
            if ((spins & 63u) == 0u && gb_now() - t0 > (unsigned long long)limit) {
                __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // the host reads this one at its next call
                break;
            }
        }
    }
    __syncthreads();
}

// 16-byte sc1 load (L1 bypassed, served by L2 / memory): the reading side of a write-through exchange
__device__ __forceinline__ gb_f32x4 ld_sc1(__amdgpu_buffer_rsrc_t rs, int byte_off) {
    return __builtin_bit_cast(gb_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 16));
}

// ---------------------------------------------------------------------------------------------
// Flag-in-data exchange (decode_persistent.hip): every exchanged float travels as ONE naturally aligned 8-byte word
// {value, tag}.  An aligned 8-byte store is single-copy atomic, so a reader that sees the expected tag sees the value that
// was stored with it — no counter, no release / acquire, no ordering between different words: the data IS the
// synchronisation.  tag = the exchange's sequence number inside the launch (1, 2, ...); the buffers are zero-filled before
// the launch, so a word of an earlier launch or an earlier step never carries the expected tag.  Both sides use sc1
// accesses (coherence point).  Buffer reuse is safe without a barrier when EVERY workgroup contributes to EVERY exchange and
// consumes every exchange completely before contributing to the next one: a word of exchange k can only be overwritten by
// exchange k + n after its writer consumed exchange k + n - 1, which contains a contribution every other workgroup made
// after it had consumed exchange k.  Polls are bounded like the barrier's.
// ---------------------------------------------------------------------------------------------
typedef unsigned gb_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned gb_u32x4 __attribute__((ext_vector_type(4)));
struct LLWatch { unsigned* status; unsigned* fault; unsigned limit; };

__device__ __forceinline__ void ll_put(__amdgpu_buffer_rsrc_t rs, int idx, float v, unsigned tag) {
    const gb_u32x2 w = {__float_as_uint(v), tag};
    __builtin_amdgcn_raw_buffer_store_b64(w, rs, idx * 8, 0, 16);
}
// words idx, idx + 1 (idx even): one 16-byte request
__device__ __forceinline__ gb_u32x4 ll_req2(__amdgpu_buffer_rsrc_t rs, int idx) {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, idx * 8, 0, 16);
}
__device__ __forceinline__ bool ll_ok2(const gb_u32x4& w, unsigned tag) { return w.y == tag && w.w == tag; }
// after a failed poll: back off; true = give up (this wait or an earlier one of the launch timed out).  f.limit = ticks of the
// 100-MHz counter one wait may take; t0 = the time of the wait's first failed poll
__device__ __forceinline__ bool ll_giveup(unsigned& spins, unsigned long long& t0, const LLWatch& f) {
    __builtin_amdgcn_s_sleep(2);
    if (spins == 0u) t0 = gb_now();
    ++spins;
    if ((spins & 255u) == 0u && __hip_atomic_load(f.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return true;
    if ((spins & 31u) == 0u && gb_now() - t0 > (unsigned long long)f.limit) {
        __hip_atomic_store(f.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(f.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return true;
    }
    return false;
}
// rows x cols floats (cols even) published as LL words -> LDS rows of stride ld floats, by NT threads (tid = 0 .. NT - 1) with U
// 16-byte requests in flight each
template <int NT, int U>
__device__ __forceinline__ void ll_stage(__amdgpu_buffer_rsrc_t rs, float* lds, int rows, int cols, int ld, unsigned tag,
                                         const LLWatch& f, int tid) {
    const int half = cols >> 1, n2 = rows * half;
    for (int c0 = tid; c0 < n2; c0 += NT * U) {
        gb_u32x4 w[U];
        unsigned spins = 0;
        unsigned long long t0 = 0;
        for (;;) {
            bool ok = true;
            asm volatile("" ::: "memory");                        // a poll re-reads memory: nothing may be hoisted out of the loop
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i2 = c0 + NT * u;
                if (i2 < n2) w[u] = ll_req2(rs, 2 * i2);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i2 = c0 + NT * u;
                if (i2 < n2) ok = ok && ll_ok2(w[u], tag);
            }
            if (ok || ll_giveup(spins, t0, f)) break;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i2 = c0 + NT * u;
            if (i2 < n2) {
                const int b = i2 / half, c = (i2 % half) * 2;
                float* o = lds + b * ld + c;
                o[0] = __uint_as_float(w[u].x);
                o[1] = __uint_as_float(w[u].z);
            }
        }
    }
}

// ---- host side (encoder_persistent.hip)
// One persistent launch: construct (takes the process-wide lock, reads the device's sticky fault word), check rc, make the
// stream wait for the previous persistent launch (`serialise`), launch, `launched()`.  rc:
//   SET_OK              go ahead; fault = device pointer of the fault word for the kernel arguments
//   SET_ERR_FAULT       an earlier persistent launch on this device timed out at a barrier (its outputs were poisoned with NaN
//                       on the device): reported ONCE, the device is then disabled for persistent launches in this process
//   SET_ERR_UNSUPPORTED the device is disabled: the caller takes the per-step kernels
struct PersistentGuard {
    int rc;
    int dev;
    unsigned* fault;
    PersistentGuard();
    ~PersistentGuard();
    int serialise(hipStream_t s);         // wait for the previous persistent launch's completion event
    int launched(hipStream_t s);          // record this launch's completion event
    unsigned spin_limit() const;          // bound of one wait in ticks of the 100-MHz counter (SET_PENC_TIMEOUT_US)
    int test_stall() const;
    // raise a kernel's dynamic-LDS cap to `bytes` on THIS device (function attributes are per device; `done` = the caller's
    // per-device flags).  SET_ERR_UNSUPPORTED when the device's LDS limit is below `bytes` or the runtime refuses: the caller
    // takes the per-step kernels, no HIP error leaves the library
    int set_lds(const void* kernel, int bytes, bool (&done)[64]);
  private:
    bool locked;
};
bool persistent_disabled();               // a barrier timed out earlier on the current device
int persistent_lds_limit();               // hipDeviceAttributeMaxSharedMemoryPerBlock of the current device (cached; 0 = unknown)

}  // namespace set
