// microbenchmark 2: which instruction classes overlap with a wave's own MFMAs (and with another wave's)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
enum { PK_FMA, FMA, ADD, MOV, DS, PK_ADD };
template <int KIND> __device__ __forceinline__ void op(f32x2 &v, f32x2 c, int lds_off) {
    if constexpr (KIND == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(c));
    if constexpr (KIND == PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v) : "v"(c));
    if constexpr (KIND == FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v.x) : "v"(c.x));
    if constexpr (KIND == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v.x) : "v"(c.x));
    if constexpr (KIND == MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(v.x) : "v"(c.x));
    if constexpr (KIND == DS) asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(lds_off));
}
template <int KIND, int NV, int WAVES, int MF>
__global__ __launch_bounds__(64 * WAVES) void k(float *out, int iters) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
    __syncthreads();
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = f32x2{a + i, b - i};
    const f32x2 c = {1.0001f, 0.9999f};
    const int lds_off = (threadIdx.x & 63) * 8;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if constexpr (MF) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(a), "v"(b));
#pragma unroll
            for (int j = 0; j < NV; ++j) op<KIND>(v[(m + j) & 7], c, lds_off);
        }
        if constexpr (KIND == DS) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + v[i].x + v[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND, int NV, int WAVES, int MF>
float run() {
    float *d; (void)hipMalloc(&d, 256 * 64 * WAVES * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    k<KIND, NV, WAVES, MF><<<256, 64 * WAVES>>>(d, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<KIND, NV, WAVES, MF><<<256, 64 * WAVES>>>(d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipFree(d);
    return ms * 1e6 / (iters * 8.0);
}
template <int KIND> void sweep(const char *name) {
    printf("%-10s  1 wave/SIMD: mfma only %.1f | +2 ops %.1f | +4 ops %.1f | +8 ops %.1f | 8 ops alone %.1f   "
           "2 waves/SIMD: mfma only %.1f | +4 ops %.1f | +8 ops %.1f | 8 ops alone %.1f   (ns per slot per wave)\n", name,
           run<KIND, 0, 4, 1>(), run<KIND, 2, 4, 1>(), run<KIND, 4, 4, 1>(), run<KIND, 8, 4, 1>(), run<KIND, 8, 4, 0>(),
           run<KIND, 0, 8, 1>(), run<KIND, 4, 8, 1>(), run<KIND, 8, 8, 1>(), run<KIND, 8, 8, 0>());
}
int main() {
    sweep<PK_FMA>("v_pk_fma"); sweep<PK_ADD>("v_pk_add"); sweep<FMA>("v_fma"); sweep<ADD>("v_add"); sweep<MOV>("v_mov"); sweep<DS>("ds_read64");
    return 0;
}
