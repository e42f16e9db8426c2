// v_mfma_scale_f32_16x16x128_f8f6f4 with A = e4m3, B = e5m2: the operand / scale layout that rz_net.hip's FP8 cross terms assume,
// checked against a host evaluation:  D[i][j] = sum_k A[i][k] B[k][j] 2^(sa[i][k / 32] - 127) 2^(sb[j][k / 32] - 127),
// lane l holds A[l % 16][32 (l / 16) .. + 31] (32 bytes, K ascending), B[32 (l / 16) .. + 31][l % 16] likewise, and
// D[4 (l / 16) + r][l % 16], r = 0 .. 3.  The kernel uses ONE scale for all lanes (byte 0 of the scale registers, op_sel 0), which is
// what is checked here; modes 3 / 4 show that "lane l carries the scale of its own row and K block" is NOT the layout of per-block
// scales (informational: the mapping was not needed and not established).  The products of a K = 128 block are summed at less than
// f32 precision inside the pipe (a few 1e-5 of the largest sum on bytes spread over all exponents, i.e. a grid about 2^-13 below the
// block's largest product): the tolerance below.  In rz_net.hip's use this adds 15 .. 20 % to the error the 8-bit operands cause.
// hipcc --offload-arch=gfx950 -O3 -o fp8_scaled_mfma_check fp8_scaled_mfma_check.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int FB>
__global__ void k(const i32x8 *a, const i32x8 *b, const int *sa, const int *sb, f32x4 *d) {
    const int l = threadIdx.x;
    f32x4 acc = d[l];   // the accumulator that comes in
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], acc, 0 /* A: e4m3 */, FB /* B: 0 e4m3, 1 e5m2 */, 0, sa[l], 0, sb[l]);
    d[l] = acc;
}
static double e4m3(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    const double x = e ? std::ldexp(1.0 + m / 8.0, e - 7) : std::ldexp(m / 8.0, -6);
    return s ? -x : x;
}
static double e5m2(unsigned char v) {
    const int s = v >> 7, e = (v >> 2) & 31, m = v & 3;
    const double x = e ? std::ldexp(1.0 + m / 4.0, e - 15) : std::ldexp(m / 4.0, -14);
    return s ? -x : x;
}
// mode 5: as mode 2 with a LARGE accumulator coming in (2^11 x the block's sum, all 24 bits in use): it must come out at f32 precision
// mode 0: all scales 1, B e4m3; 1: all scales 1, B e5m2; 2: one scale for all lanes; 3: a scale per lane (A only); 4: per lane, both
static int run(int mode) {
    unsigned char ha[64][32], hb[64][32];
    int hsa[64], hsb[64];
    srand(3 + mode);
    const bool b52 = mode >= 1;
    for (int l = 0; l < 64; ++l) {
        for (int j = 0; j < 32; ++j) {
            unsigned char x = rand() & 255;
            if ((x & 0x7f) == 0x7f) x ^= 1;   // e4m3fn: S.1111.111 is NaN
            ha[l][j] = x;
            unsigned char y = rand() & 255;
            if (b52) {
                if (((y >> 2) & 31) == 31) y ^= 4;   // e5m2: exponent 31 is inf / nan
                if (((y >> 2) & 31) > 20) y &= ~0x40;   // keep the products moderate
            } else if ((y & 0x7f) == 0x7f) y ^= 1;
            hb[l][j] = y;
        }
        hsa[l] = mode < 2 ? 127 : (mode == 2 || mode == 5) ? 124 : 120 + rand() % 12 + ((rand() & 0xffff) << 8);   // (the upper bytes must not matter with op_sel 0)
        hsb[l] = mode < 2 ? 127 : mode == 2 ? 129 : mode == 5 ? 119 : mode == 3 ? 127 : 122 + rand() % 9;
    }
    void *da, *db, *dsa, *dsb, *dd;
    (void)hipMalloc(&da, sizeof ha); (void)hipMalloc(&db, sizeof hb); (void)hipMalloc(&dsa, sizeof hsa); (void)hipMalloc(&dsb, sizeof hsb);
    (void)hipMalloc(&dd, 64 * 16);
    float hc[64][4];
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) hc[l][r] = mode == 5 ? (float)((rand() % 2000001 - 1000000) * 1.0000001) : 0.0f;
    (void)hipMemcpy(dd, hc, sizeof hc, hipMemcpyHostToDevice);
    (void)hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    (void)hipMemcpy(dsa, hsa, sizeof hsa, hipMemcpyHostToDevice); (void)hipMemcpy(dsb, hsb, sizeof hsb, hipMemcpyHostToDevice);
    if (b52) k<1><<<1, 64>>>((const i32x8 *)da, (const i32x8 *)db, (const int *)dsa, (const int *)dsb, (f32x4 *)dd);
    else k<0><<<1, 64>>>((const i32x8 *)da, (const i32x8 *)db, (const int *)dsa, (const int *)dsb, (f32x4 *)dd);
    float hd[64][4];
    (void)hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
    double worst = 0.0, top = 0.0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double ref = 0.0;
            for (int g = 0; g < 4; ++g) {
                double part = 0.0;
                for (int kk = 0; kk < 32; ++kk) part += e4m3(ha[16 * g + i][kk]) * (b52 ? e5m2(hb[16 * g + j][kk]) : e4m3(hb[16 * g + j][kk]));
                ref += std::ldexp(part, (hsa[16 * g + i] & 255) - 127 + (hsb[16 * g + j] & 255) - 127);
            }
            ref += hc[16 * (i / 4) + j][i % 4];
            const double got = hd[16 * (i / 4) + j][i % 4];
            if (i == 5 && j < 3) printf("    D[5][%d]: device %.6e host %.6e\n", j, got, ref);
            worst = std::fmax(worst, std::fabs(got - ref));
            top = std::fmax(top, std::fabs(ref));
        }
    const bool ok = worst <= (mode == 5 ? 3e-7 : 2e-4) * top;
    printf("mode %d: max |device - host| = %.3e of max |host| = %.3e -> %s\n", mode, worst, top, ok ? "the assumed layout holds" : "MISMATCH");
    return ok ? 0 : 1;
}
int main() {
    int bad = 0;
    for (int mode : {0, 1, 2, 5}) bad += run(mode);
    printf("informational (per-lane scales, expected to differ):\n");
    for (int mode = 3; mode < 5; ++mode) (void)run(mode);
    return bad;
}
