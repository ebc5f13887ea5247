// segk_chain_barrier.h -- the grid barrier of the persistent sequential chains (segk_seq_chain.hip: k-means;
// segk_fbgmm.hip: FBGMM).  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CH_SPIN_LIMIT (1 << 22)
#define CH_FLAGS 16               /* release words of the barrier, one 128-byte line each */

// Grid barrier number `phase` (1, 2, ...): false when the spin limit was hit or another workgroup reported an error.  The
// arrivals go to ONE counter (low 30 bits; bit 30: a component emptied, stop after this utterance); the workgroup whose add
// came last learns it from the value the add returned and releases the others through CH_FLAGS words on cache lines of
// their own (one store instruction, one lane per word), each polled by G / CH_FLAGS workgroups: with every workgroup
// polling the counter itself the last arriver's own add queued behind 124 pollers (5.9 us per barrier for the LAST one).
// ctl: [0] counter, [1] stop (for the host), [2] utterances completed, [3] error, [32 * (1 + f)] release word f.
#define CH_STOP_BIT (1 << 30)
static __device__ __forceinline__ bool chain_barrier(int32_t *ctl, int phase, int *sh_flag, unsigned long long *dbg = nullptr)
{
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // this wave's atomics and stores have been performed
    __syncthreads();
    if (dbg && threadIdx.x == 0) *dbg = wall_clock64();
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const int target = phase * (int)gridDim.x;
        int seen = 0;
        if (lane == 0) seen = __hip_atomic_fetch_add(&ctl[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
        seen = __shfl(seen, 0);
        int ok = 1, word;
        if ((seen & (CH_STOP_BIT - 1)) >= target) {                 // the last one in: release the others
            word = (seen & CH_STOP_BIT) | phase;
            if (lane < CH_FLAGS) __hip_atomic_store(&ctl[32 * (1 + lane)], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            int *flag = &ctl[32 * (1 + (int)(blockIdx.x % CH_FLAGS))];
            int spins = 0;
            for (;;) {
                word = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((word & (CH_STOP_BIT - 1)) >= phase) break;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > CH_SPIN_LIMIT || ((spins & 1023) == 0 && __hip_atomic_load(&ctl[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                    __hip_atomic_fetch_or(&ctl[3], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = 0;
                    break;
                }
            }
        }
        // bit 0: passed; bit 1: a component emptied during the previous utterance
        if (lane == 0) *sh_flag = ok | ((word & CH_STOP_BIT) ? 2 : 0);
    }
    __syncthreads();
    return (*sh_flag & 1) != 0;
}

