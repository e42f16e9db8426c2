// sustained rate of v_mfma_f32_32x32x16_f16 / 16x16x32_f16 on every CU, one wave per SIMD, random operands
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int BIG, int ZERO>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = ZERO ? (_Float16)0 : (_Float16)(0.01f * ((threadIdx.x * 7 + j * 13) % 97) - 0.4f);
        b[j] = ZERO ? (_Float16)0 : (_Float16)(0.02f * ((threadIdx.x * 5 + j * 11) % 89) - 0.7f);
    }
    float s = 0;
    if constexpr (BIG) {
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][7];
    } else {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 8; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[m], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int BIG, int ZERO>
void run(int grid) {
    float *d; (void)hipMalloc(&d, 256 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 40000;
    k<BIG, ZERO><<<grid, 256>>>(d, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<BIG, ZERO><<<grid, 256>>>(d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / (iters * 8.0);
    const double flops = (BIG ? 32.0 * 32 * 16 : 16.0 * 16 * 32) * 2 * 4 * grid / (ns * 1e-9);
    printf("%s %s grid %d: %.2f ns per MFMA per wave, %.0f TFLOP/s\n", BIG ? "32x32x16" : "16x16x32", ZERO ? "zeros " : "random", grid, ns, flops / 1e12);
    (void)hipFree(d);
}
int main() {
    run<1, 1>(256); run<1, 0>(256); run<1, 0>(224); run<1, 0>(32); run<0, 1>(256); run<0, 0>(256);
    return 0;
}
