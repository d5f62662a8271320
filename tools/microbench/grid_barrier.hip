// What does a device-wide barrier between the phases of ONE persistent launch cost on MI355X (256 CUs, 8 XCDs with their own
// L2)?  Decides whether the reference-default batch-32 step can run as one kernel with 4 - 5 barriers instead of 8 launches
// (each launch boundary costs ~5 us of drain + ramp at this size).  Each workgroup also passes DATA across every barrier (a
// word written before, read by its neighbour after) so that the measured barrier is one that actually orders memory.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/grid_barrier tools/microbench/grid_barrier.hip && /tmp/grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

// one counter, monotone: barrier number b is passed when the counter reaches b * nwg
__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned& target, unsigned nwg) {
    __syncthreads();
    if (threadIdx.x == 0) {
        target += nwg;
        __atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE);                    // agent scope: prior writes visible device-wide
        while (__atomic_load_n(ctr, __ATOMIC_ACQUIRE) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k(unsigned* ctr, unsigned* mail, int nb, unsigned* bad, int dirty_words, float* dirty) {
    unsigned target = 0;
    const unsigned nwg = gridDim.x, me = blockIdx.x;
    unsigned errs = 0;
    for (int b = 0; b < nb; ++b) {
        if (dirty_words) {                                                // (a phase that leaves dirty lines in this XCD's L2)
            for (int i = threadIdx.x; i < dirty_words; i += 256) dirty[(long)me * dirty_words + i] = (float)(b + i);
        }
        if (threadIdx.x == 0) __atomic_store_n(&mail[me], (unsigned)(b + 1), __ATOMIC_RELAXED);
        grid_barrier(ctr, target, nwg);
        // the word of a workgroup that (round robin) lives on ANOTHER XCD
        if (threadIdx.x == 0) {
            const unsigned got = __atomic_load_n(&mail[(me + 1) % nwg], __ATOMIC_RELAXED);
            if (got != (unsigned)(b + 1)) ++errs;
        }
        grid_barrier(ctr, target, nwg);                                   // (nobody overwrites its word before it was read)
    }
    if (threadIdx.x == 0 && errs) atomicAdd(bad, errs);
}

int main() {
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int nwg = pr.multiProcessorCount;
    unsigned *ctr, *mail, *bad; float* dirty;
    const int maxdirty = 32768;
    hipMalloc(&ctr, 4); hipMalloc(&mail, 4 * nwg); hipMalloc(&bad, 4); hipMalloc(&dirty, 4L * nwg * maxdirty);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int dw : {0, 4096, 32768}) {
        for (int nb : {1, 8, 64}) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipMemset(ctr, 0, 4); hipMemset(bad, 0, 4); hipMemset(mail, 0, 4 * nwg);
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(k, dim3(nwg), dim3(256), 0, 0, ctr, mail, nb, bad, dw, dirty);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            unsigned hb; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
            printf("%d workgroups, %6d dirty words per workgroup and round, %2d rounds (2 barriers each): %8.2f us  -> %6.2f us per barrier  stale reads %u\n",
                   nwg, dw, nb, best * 1e3, best * 1e3 / (2 * nb), hb);
        }
    }
    return 0;
}
