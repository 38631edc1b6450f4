// mpb_stomp_fused.h -- what the persistent STOMP kernels (mpb_stomp_fused.hip: the H = 64 kernel; mpb_stomp_fused_hx.hip:
// any horizon up to 128, any channel count, any number of sample batches) share: the workspace header, the ticket pools,
// the agent-scope accessors and the tagged granules of the exchange between the workgroups of a particle.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#define FUSED_WAVES 16
#define FUSED_THREADS (64 * FUSED_WAVES)
#define FUSED_LD 68                        // padded row (floats) of the Sigma image and of the transposed delta tile
#define FUSED_XCHG 912                     // granules per published partial: m, z, then H*d <= 896 values, padded
#define FUSED_MAX_CHUNKS 4                 // S <= 64
#define FUSED_TIMEOUT_TICKS 200000000ull   // 2 s of s_memrealtime (100 MHz) + FUSED_TIMEOUT_PER_ITER per iteration of the call
#define FUSED_TIMEOUT_PER_ITER 10000ull    // 100 us: a workgroup whose partner starts a whole round of workgroups later waits that long
// workspace header (16 words, zero before the first use: mpb_stomp_workspace_init; maintained by the kernel afterwards)
#define FUSED_HDR_ERR 0      // tag of the call in which a workgroup gave up waiting (or found the header uninitialised)
#define FUSED_HDR_TAG 1      // tag of the last call
#define FUSED_HDR_DONE 2     // workgroups of the running call that have left; 0 between calls
#define FUSED_HDR_WHY 3      // why FUSED_HDR_ERR was raised: 1 = partner timed out, 2 = header not initialised
// Every workgroup hits the start stamp and a ticket pool with an atomic when it starts: in one 64-byte line (rounds 2-4) those
// 512 read-modify-writes queued behind each other at one L2 channel -- a workgroup had its ticket 5.7 us after it entered the
// kernel (scripts/stamps_launch.py).  The stamp and each pool now sit in a 128-byte line of their own.
#define FUSED_HDR_BEGIN 32   // words 32, 33 (one 8-byte word): ~(earliest start of a workgroup of the running call); 0 between calls
#define FUSED_HDR_TICKET 64  // word 64 + 32 k: next ticket of unit pool k (8 pools) of the running call; 0 between calls
#define FUSED_HDR_POOL_STRIDE 32
#define FUSED_HDR_WORDS 320  // the header: 1 280 bytes, zero before the first call (mpb_stomp_workspace_init)
// unit pools: pool x owns the particles p = x (mod FUSED_POOLS); a workgroup draws from the pool of the XCD it runs on
// first (partners then share an L2: speed only), from the next pools once that one is exhausted
#ifndef FUSED_POOLS
#define FUSED_POOLS 8
#endif

__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent_u(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld_agent_u(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_system_u(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// one naturally aligned 8-byte granule {value, tag}, written by ONE agent-scope (sc1) store and read by ONE sc1 load: the
// tag tells the reader which iteration of which call the value belongs to, so the payload needs no separate flag, no
// drain of the stores and no fence (MI355X_MICROARCH.md: "handoff-1to1, data-tagged granules"; observed untorn)
typedef unsigned long long granule_t;
__device__ __forceinline__ void st_granule(granule_t* p, float v, unsigned tag) {
    __hip_atomic_store(p, ((granule_t)tag << 32) | (granule_t)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ granule_t ld_granule(const granule_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }


// Measurement aid (mpb_stomp_run_timed): while these are set, the persistent launch records the pair on the dispatch
// itself (hipExtLaunchKernelGGL: kernel begin / end timestamps, the quantity rocprofv3 --kernel-trace reports).
struct FusedProfile { hipEvent_t start, stop; };
#define MPB_FUSED_LAUNCH(prof, kernel, grid, block, st, ...)                                                        \
    do {                                                                                                            \
        if ((prof) != nullptr) hipExtLaunchKernelGGL(kernel, grid, block, 0, st, (prof)->start, (prof)->stop, 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel, grid, block, 0, st, __VA_ARGS__);                                           \
    } while (0)

// unit of a workgroup of the exchange layout, by ticket (see the header comment of mpb_stomp_fused.hip): returns the unit
// index u = particle * nc + chunk, or sets `why` (1: an earlier workgroup of this call gave up; 2: header not zeroed)
__device__ __forceinline__ unsigned fused_draw_unit(unsigned* wsu, int P, int nc, uint32_t tag0, int& why) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    const unsigned lost = ld_agent_u(wsu + FUSED_HDR_ERR);     // (in flight together with the first ticket draw)
    unsigned u = 0;
    bool got = false;
    for (unsigned k = 0; k < FUSED_POOLS && !got; ++k) {
        const unsigned pool = (xcc + k) % FUSED_POOLS;
        const unsigned size = ((unsigned)P + FUSED_POOLS - 1u - pool) / FUSED_POOLS * (unsigned)nc;
        if (size == 0u) continue;
        const unsigned t = __hip_atomic_fetch_add(wsu + FUSED_HDR_TICKET + FUSED_HDR_POOL_STRIDE * pool, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t < size) {
            got = true;
            u = (FUSED_POOLS * (t / (unsigned)nc) + pool) * (unsigned)nc + t % (unsigned)nc;
        }
    }
    why = 0;
    if (lost == tag0) why = 1;                 // the call is lost already: leave at once
    if (!got) {                                // (as many units as workgroups: only a header that was not zero gets here)
        why = 2;
        u = 0;
    }
    return u;
}

// every workgroup stamps the device's real-time counter (100 MHz) into header words 4, 5 when it starts -- an atomic MAX of
// the INVERTED time, so that the zeroed header is the neutral element and the word ends up as ~(earliest start); the last
// workgroup out copies it to words 4, 5 of the status block next to its own time (words 6, 7): the span of the launch as
// the device saw it, from the FIRST workgroup to start (whichever unit it owns) to the last one to leave
__device__ __forceinline__ void fused_stamp_begin(unsigned* wsu, unsigned* status_host) {
    if (status_host) {
        const unsigned long long t = __builtin_amdgcn_s_memrealtime();
        __hip_atomic_fetch_max(reinterpret_cast<unsigned long long*>(wsu + FUSED_HDR_BEGIN), ~t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// leaving: the error word (device header + the caller's host-visible status block), then the head count; the last
// workgroup out re-arms the header for the next call and reports the call as completed.  One thread per workgroup.
__device__ __forceinline__ void fused_leave(unsigned* wsu, unsigned* status_host, uint32_t tag0, int aborted) {
    if (aborted) {
        st_agent_u(wsu + FUSED_HDR_WHY, (unsigned)aborted);
        st_agent_u(wsu + FUSED_HDR_ERR, tag0);   // == header word 1 of THIS call: "lost"
        if (status_host) {
            st_system_u(status_host + 2, (unsigned)aborted);
            st_system_u(status_host + 1, tag0);
            __threadfence_system();              // (rare path) visible to the host before the head count says "completed"
        }
    }
    const unsigned left = __hip_atomic_fetch_add(wsu + FUSED_HDR_DONE, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (left == gridDim.x - 1u && aborted != 2) {
#pragma unroll
        for (int k = 0; k < 8; ++k) st_agent_u(wsu + FUSED_HDR_TICKET + FUSED_HDR_POOL_STRIDE * k, 0u);
        st_agent_u(wsu + FUSED_HDR_DONE, 0u);
        if (status_host) {
            unsigned long long* b64 = reinterpret_cast<unsigned long long*>(wsu + FUSED_HDR_BEGIN);
            // (every workgroup's stamp precedes its own release increment of the head count, which this one has acquired)
            const unsigned long long t0 = ~__hip_atomic_load(b64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(b64, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long t = __builtin_amdgcn_s_memrealtime();     // words 6, 7: when the last workgroup left
            st_system_u(status_host + 4, (unsigned)t0);
            st_system_u(status_host + 5, (unsigned)(t0 >> 32));
            st_system_u(status_host + 6, (unsigned)t);
            st_system_u(status_host + 7, (unsigned)(t >> 32));
            // "completed" is published last, with release order: a host that reads the tag sees the stamps of THIS call
            __hip_atomic_store(status_host + 0, tag0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
