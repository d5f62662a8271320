// K-PEER: the small exchanges of the data-parallel step (SyncBN statistics: 2 h floats per BatchNormalization layer, forward
// and backward -- dca_amd/dist.py items 2 and 3) WITHOUT a library call: every rank stores its vector into a slot of every
// peer's exchange buffer (mapped through hipIpc: peer stores over xGMI), raises a flag there, waits for the flags in its own
// buffer and reads its own slots.  One launch of one workgroup per exchange; no host involvement, capturable into the step's
// hipGraph (the epoch counter lives in device memory).  RCCL spends 17-20 us per call on these <= 4 KB messages before a byte
// moves (profiles/r04_dp_one_rank.txt: six calls = 0.13 ms of a 1.24 ms step).
//
// Protocol, epoch e = 1, 2, ...: parity p = e & 1 selects one of two slot / flag sets.
//   write:   slots[q][p][rank][0 .. n) = local          for every rank q (its own included)
//            system-scope fence, then  flags[q][p][rank] = e   (release, system scope)
//   wait:    until flags[rank][p][q] == e for every q (acquire, system scope), for at most timeout_us microseconds of the
//            100 MHz wall clock (the product passes a minute -- rank skew from checkpoint writes, first captures or page
//            faults is tolerated the way RCCL tolerates it; the init-time self check of dca_amd/peer.py a second)
//   read:    out = slots[rank][p][*]  as the concatenation (gather) or the sum in rank order (reduce: identical on all ranks)
//   A wait that times out sets *status |= 1 AND writes NaN over the whole of `out`: stale statistics never enter a step
//   silently -- the batch loss of that step and every later one is NaN, and the fit raises at its next host synchronisation.
//
// Memory: the slot and flag buffers MUST be fine-grained (hipExtMallocWithFlags(hipDeviceMallocFinegrained), dca_amd/peer.py).
// A peer's stores arrive over xGMI on the fabric side of the owner's memory; in a coarse-grained (plain hipMalloc)
// allocation the owner's L2 may keep serving its spinning loads from a line it cached before the store landed -- system-scope
// atomics order the accesses, they do not make MTYPE_RW lines coherent with fabric-side writes.  Fine-grained memory is
// mapped uncached at device scope for exactly this use (RCCL allocates its flag / LL buffers the same way).
// Two sets suffice: a rank writes epoch e + 2 into set p only after it finished epoch e + 1, which needed every peer's e + 1
// flag, i.e. every peer had started its e + 1 exchange -- in stream order behind its complete e exchange (the reads of set p).
// Reference: none (the reference is single-process; SURVEY 8e adds data parallelism); replaces two torch.distributed calls.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dcahip.h"

namespace {

struct PeerArgs {
    const float* local; int n;
    float* const* slots; unsigned* const* flags;
    int rank, world, nmax;
    unsigned long long* epoch;
    float* out; int reduce;
    int* status; long timeout_us;
};

__global__ __launch_bounds__(256) void peer_exchange_kernel(PeerArgs a) {
    const int tid = threadIdx.x;
    const unsigned e = (unsigned)(*a.epoch) + 1u;
    const long set = (long)(e & 1u) * a.world;
    for (int q = 0; q < a.world; ++q) {
        float* dst = a.slots[q] + (set + a.rank) * a.nmax;
        for (int i = tid; i < a.n; i += 256) __hip_atomic_store(dst + i, a.local[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __threadfence_system();
    __syncthreads();
    if (tid < a.world) __hip_atomic_store(a.flags[tid] + set + a.rank, e, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    __shared__ int timed_out;
    if (tid == 0) timed_out = 0;
    __syncthreads();
    if (tid < a.world) {
        const unsigned* f = a.flags[a.rank] + set + tid;
        const unsigned long long t0 = wall_clock64(), limit = (unsigned long long)a.timeout_us * 100ull;     // 100 MHz ticks
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != e) {
            if (wall_clock64() - t0 > limit) { atomicOr(a.status, 1); timed_out = 1; break; }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
    __threadfence_system();
    const float* mine = a.slots[a.rank] + set * a.nmax;
    if (timed_out) {
        const int total = a.reduce ? a.n : a.n * a.world;
        for (int i = tid; i < total; i += 256) a.out[i] = __builtin_nanf("");
    } else if (a.reduce) {
        for (int i = tid; i < a.n; i += 256) {
            float v = 0.f;
            for (int q = 0; q < a.world; ++q) v += __hip_atomic_load(mine + (long)q * a.nmax + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            a.out[i] = v;
        }
    } else {
        for (int i = tid; i < a.n * a.world; i += 256) {
            const int q = i / a.n, j = i - q * a.n;
            a.out[i] = __hip_atomic_load(mine + (long)q * a.nmax + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __syncthreads();
    if (tid == 0) *a.epoch = (unsigned long long)e;
}

}  // namespace

extern "C" long dcahip_peer_slot_bytes(int world, int nmax) { return world > 0 && nmax > 0 ? 2L * world * nmax * (long)sizeof(float) : 0; }
extern "C" long dcahip_peer_flag_bytes(int world) { return world > 0 ? 2L * world * (long)sizeof(unsigned) : 0; }

extern "C" int dcahip_peer_exchange(const float* local, int n, float* const* slots, unsigned* const* flags, int rank, int world,
                                    int nmax, unsigned long long* epoch, float* out, int reduce, int* status, long timeout_us,
                                    void* stream) {
    if (!local || !slots || !flags || !epoch || !out || !status || n <= 0 || n > nmax || world <= 0 || world > 256 || rank < 0 ||
        rank >= world || timeout_us <= 0)
        return DCAHIP_EINVAL;
    PeerArgs a{local, n, slots, flags, rank, world, nmax, epoch, out, reduce, status, timeout_us};
    hipLaunchKernelGGL(peer_exchange_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}
