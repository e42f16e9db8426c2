// micro-latency probe (development aid): cycles per operation for one wave on an otherwise idle CU
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang fp contract(off)
__global__ void k(double *out, long long *cyc, int active, double seed) {
    __shared__ double lds[1024];
    __shared__ int chase[1024];
    const int l = threadIdx.x;
    for (int i = l; i < 1024; i += 64) { lds[i] = 1.0 + i * 1e-3; chase[i] = (i * 37 + 11) & 1023; }
    __syncthreads();
    if (l >= active) return;
    double a = seed + l * 1e-9, b = 1.000001, c = 0.5;
    long long t0, t1;
    // 1: dependent fp64 fma chain
    t0 = clock64();
    for (int i = 0; i < 256; ++i) a = __builtin_fma(a, b, c);
    t1 = clock64(); if (l == 0) cyc[0] = (t1 - t0) / 256;
    // 2: dependent division chain
    double x = a;
    t0 = clock64();
    for (int i = 0; i < 64; ++i) x = (x + 3.0) / (b + 1.0);
    t1 = clock64(); if (l == 0) cyc[1] = (t1 - t0) / 64;
    // 3: four independent division chains (compiler order)
    double y0 = x, y1 = x + 1, y2 = x + 2, y3 = x + 3;
    t0 = clock64();
    for (int i = 0; i < 64; ++i) { y0 = (y0 + 3.0) / (b + 1.0); y1 = (y1 + 3.0) / (b + 2.0); y2 = (y2 + 3.0) / (b + 3.0); y3 = (y3 + 3.0) / (b + 4.0); }
    t1 = clock64(); if (l == 0) cyc[2] = (t1 - t0) / 64;
    // 4: LDS pointer chase (dependent ds_read_b32)
    int p = l;
    t0 = clock64();
    for (int i = 0; i < 256; ++i) p = chase[p];
    t1 = clock64(); if (l == 0) cyc[3] = (t1 - t0) / 256;
    // 5: dependent LDS b128 + b64 read of a 32-byte record then index
    t0 = clock64();
    double acc = 0;
    for (int i = 0; i < 256; ++i) { const double2 v = *reinterpret_cast<const double2 *>(&lds[(p & 511) * 2]); acc += v.x; p = chase[(p + (int)v.y) & 1023]; }
    t1 = clock64(); if (l == 0) cyc[4] = (t1 - t0) / 256;
    // 6: dependent fp64 add chain
    t0 = clock64();
    for (int i = 0; i < 256; ++i) a = a + c;
    t1 = clock64(); if (l == 0) cyc[5] = (t1 - t0) / 256;
    // 7: dependent fp32 fma chain
    float f = (float)a;
    t0 = clock64();
    for (int i = 0; i < 256; ++i) f = __builtin_fmaf(f, 1.0001f, 0.5f);
    t1 = clock64(); if (l == 0) cyc[6] = (t1 - t0) / 256;
    // 8: sqrt f64 dependent
    double sq = a;
    t0 = clock64();
    for (int i = 0; i < 64; ++i) sq = sqrt(sq + 2.0);
    t1 = clock64(); if (l == 0) cyc[7] = (t1 - t0) / 64;
    // 9: global load pointer chase (L2 hit)
    out[l] = a + x + y0 + y1 + y2 + y3 + p + acc + f + sq;
}
int main() {
    double *out; long long *cyc;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 16 * 8);
    for (int active : {64, 16}) {
        hipMemset(cyc, 0, 128);
        k<<<1, 64>>>(out, cyc, active, 1.25);
        hipDeviceSynchronize();
        long long h[16]; hipMemcpy(h, cyc, 128, hipMemcpyDeviceToHost);
        printf("active=%d  fma64=%lld  div64=%lld  4xdiv64=%lld  lds_chase=%lld  lds_rec=%lld  add64=%lld  fma32=%lld  sqrt64=%lld\n", active, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    }
    return 0;
}
