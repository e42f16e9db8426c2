// What does a hand-off between two workgroups on different CUs cost inside ONE launch?  (The question behind "compute the next
// selection's candidates on an idle CU while the trunk of this simulation runs": the tree lives in global memory, one workgroup
// writes it, the other must see it.)  Pairs of workgroups (one per CU: 150 KB of LDS each) play ping-pong through global memory:
// the producer stores `bytes` of payload with plain stores, then a RELEASE store of a sequence number at agent scope; the consumer
// spins with ACQUIRE loads, reads and checks the payload, answers the same way.  Every spin is bounded (a lost partner ends the
// kernel with an error count, never a hang).
//   hipcc --offload-arch=gfx950 -O2 profiles/microbench/wg_handoff.hip -o /tmp/wg_handoff && /tmp/wg_handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int kSpinMax = 1 << 22;

__device__ __forceinline__ bool wait_for(const unsigned *p, unsigned want) {
    for (int i = 0; i < kSpinMax; ++i) {
        if (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == want) {   // (polls: sc1 loads; ONE invalidate behind the hit)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            return true;
        }
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

// role: blocks [0, n) produce, blocks [cons0, cons0 + n) consume for producer (block - cons0); everything else leaves at once
template <int MODE>
__global__ __launch_bounds__(256) void k_pingpong(unsigned *seq, unsigned *ack, unsigned *payload, int words, int n, int cons0, int iters,
                                                  unsigned long long *ticks, unsigned *errors, unsigned *xcc) {
    extern __shared__ unsigned char lds[];
    const int b = blockIdx.x, lane = threadIdx.x;
    const bool producer = b < n, consumer = b >= cons0 && b < cons0 + n;
    if (!producer && !consumer) return;
    const int p = producer ? b : b - cons0;
    if (lane == 0) xcc[b] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20) & 15;
    if (threadIdx.x >= 64) return;   // one wave plays (the others would be the trunk's)
    unsigned *pay = payload + (size_t)p * words;
    unsigned bad = 0;
    const unsigned long long t0 = wall_clock64();
    for (int it = 1; it <= iters; ++it) {
        if (producer) {
            for (int i = lane; i < words; i += 64) pay[i] = (unsigned)it * 2654435761u + i;
            if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_store(seq + p, (unsigned)it, MODE == 0 ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool ok = true;
            if (lane == 0) ok = wait_for(ack + p, (unsigned)it);
            ok = __shfl(ok, 0);
            if (!ok) { bad |= 1u << 31; break; }
        } else {
            bool ok = true;
            if (lane == 0) ok = wait_for(seq + p, (unsigned)it);
            ok = __shfl(ok, 0);
            if (!ok) { bad |= 1u << 31; break; }
            // (the acquire was lane 0's: the wave's later loads are behind it in program order)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            for (int i = lane; i < words; i += 64) bad += pay[i] != (unsigned)it * 2654435761u + i;
            if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_store(ack + p, (unsigned)it, MODE == 0 ? __ATOMIC_RELEASE : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const unsigned long long t1 = wall_clock64();
    for (int off = 32; off >= 1; off >>= 1) bad |= __shfl_xor(bad, off);
    if (lane == 0) {
        ticks[b] = t1 - t0;
        if (bad) atomicAdd(errors, 1u);
        if (bad & 0x7fffffffu) atomicAdd(errors + 1, 1u);
    }
    (void)lds;
}

int main() {
    const int n = 64, iters = 2000;
    unsigned *seq, *ack, *payload, *errors, *xcc;
    unsigned long long *ticks;
    hipMalloc(&seq, 4096 * 4);
    hipMalloc(&ack, 4096 * 4);
    hipMalloc(&payload, (size_t)n * 16384 * 4);
    hipMalloc(&errors, 8);
    hipMalloc(&xcc, 1024 * 4);
    hipMalloc(&ticks, 1024 * 8);
    hipFuncSetAttribute((const void *)k_pingpong<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipFuncSetAttribute((const void *)k_pingpong<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    for (int mode : {0, 1})
    for (int shift : {0, 1}) {
        for (int words : {16, 256, 2048}) {
            const int cons0 = 64 + shift, grid = cons0 + n;
            hipMemset(seq, 0, 4096 * 4);
            hipMemset(ack, 0, 4096 * 4);
            hipMemset(errors, 0, 8);
            hipMemset(ticks, 0, 1024 * 8);
            if (mode == 0) k_pingpong<0><<<grid, 256, 150 * 1024>>>(seq, ack, payload, words, n, cons0, iters, ticks, errors, xcc);
            else k_pingpong<1><<<grid, 256, 150 * 1024>>>(seq, ack, payload, words, n, cons0, iters, ticks, errors, xcc);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
            std::vector<unsigned long long> t(1024);
            std::vector<unsigned> x(1024), e(2);
            hipMemcpy(t.data(), ticks, 1024 * 8, hipMemcpyDeviceToHost);
            hipMemcpy(x.data(), xcc, 1024 * 4, hipMemcpyDeviceToHost);
            hipMemcpy(e.data(), errors, 8, hipMemcpyDeviceToHost);
            double mean = 0;
            int same = 0;
            for (int p = 0; p < n; ++p) { mean += (double)t[p]; same += x[p] == x[cons0 + p]; }
            mean /= n;
            printf("%s consumer = producer + %d: payload %5d B: %.0f ns per round trip (two hand-offs), pairs on the same XCD: %d of %d, waves in error %u (payload mismatches in %u)\n",
                   mode == 0 ? "agent-scope release:" : "stores waited for, relaxed flag (one L2 only):", cons0, words * 4, mean * 10.0 / iters, same, n, e[0], e[1]);
        }
    }
    return 0;
}
