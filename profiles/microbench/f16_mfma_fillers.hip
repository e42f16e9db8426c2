// does VALU / LDS work placed between a wave's own 32x32x16 f16 MFMAs hide? wall time and s_memtime, few vs all CUs
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
enum { FMA, PK, ACCRD, DS };
template <int KIND, int NV>
__global__ __launch_bounds__(256) void k(float *out, long long *ticks, int iters) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
    __syncthreads();
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (_Float16)(0.01f * ((threadIdx.x * 7 + j * 13) % 97) - 0.4f);
        b[j] = (_Float16)(0.02f * ((threadIdx.x * 5 + j * 11) % 89) - 0.7f);
    }
    f32x16 acc[4], spare;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    for (int r = 0; r < 16; ++r) spare[r] = r;
    asm volatile("" : "+a"(spare));
    f32x2 v[8];
    for (int i = 0; i < 8; ++i) v[i] = f32x2{(float)threadIdx.x + i, 1.0f - i};
    const f32x2 c = {1.0001f, 0.9999f};
    const int lds_off = (threadIdx.x & 63) * 16;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(a), "v"(b));
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                f32x2 &x = v[(m * NV + j) & 7];
                if constexpr (KIND == FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x.x) : "v"(c.x));
                if constexpr (KIND == PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(c));
                if constexpr (KIND == ACCRD) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x.x) : "a"(spare[0]));
                if constexpr (KIND == DS) asm volatile("ds_read_b128 %0, %1" : "=v"(*(float __attribute__((ext_vector_type(4))) *)&v[(j & 3) * 2]) : "v"(lds_off));
            }
        }
        if constexpr (KIND == DS) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][9];
    for (int i = 0; i < 8; ++i) s += v[i].x + v[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int KIND, int NV>
void run(const char *name, int grid) {
    float *d; long long *t; (void)hipMalloc(&d, 256 * 256 * 4); (void)hipMalloc(&t, 256 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    k<KIND, NV><<<grid, 256>>>(d, t, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<KIND, NV><<<grid, 256>>>(d, t, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[256]; (void)hipMemcpy(h, t, grid * 8, hipMemcpyDeviceToHost);
    printf("%-8s x%d per MFMA, grid %3d: %.2f ns and %.1f memtime ticks per MFMA\n", name, NV, grid, ms * 1e6 / (iters * 4.0), (double)h[0] / (iters * 4.0));
    (void)hipFree(d); (void)hipFree(t);
}
int main() {
    for (int grid : {16, 224}) {
        run<FMA, 0>("none", grid); run<FMA, 2>("v_fma", grid); run<FMA, 4>("v_fma", grid); run<FMA, 6>("v_fma", grid); run<FMA, 8>("v_fma", grid);
        run<PK, 2>("v_pk_fma", grid); run<PK, 4>("v_pk_fma", grid);
        run<ACCRD, 2>("accread", grid); run<ACCRD, 4>("accread", grid);
        run<DS, 1>("ds_b128", grid); run<DS, 2>("ds_b128", grid);
    }
    return 0;
}
