// Where do the waves of co-resident workgroups land?  k_mz_search runs two workgroups of four waves per CU and wave 0 of each walks
// the trees (the serial part of a simulation): if both walker waves sit on the same SIMD they share its issue slots.
//   hipcc --offload-arch=gfx950 -O2 profiles/microbench/wave_placement.hip -o /tmp/wave_placement && /tmp/wave_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256, 2) void k(unsigned *out, int spin) {
    extern __shared__ unsigned char lds[];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID, all 32 bits
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
        out[(blockIdx.x * 4 + w) * 2] = hw;
        out[(blockIdx.x * 4 + w) * 2 + 1] = xcc;
    }
    // stay resident long enough for every workgroup of the grid to be placed beside its neighbour
    volatile unsigned char *p = lds;
    for (int i = 0; i < spin; ++i) p[threadIdx.x] = (unsigned char)(p[threadIdx.x] + i);
}
int main() {
    const int n_wg = 512;
    unsigned *d;
    hipMalloc(&d, n_wg * 4 * 2 * 4);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 81 * 1024);
    k<<<n_wg, 256, 81 * 1024>>>(d, 20000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(n_wg * 8);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    // HW_ID (gfx9): WAVE_ID [3:0], SIMD_ID [5:4], PIPE_ID [7:6], CU_ID [11:8], SH_ID [12], SE_ID [15:13], TG_ID [19:16]
    std::map<unsigned long long, std::vector<int>> by_cu;
    int distinct4 = 0;
    for (int b = 0; b < n_wg; ++b) {
        unsigned simds = 0;
        for (int w = 0; w < 4; ++w) simds |= 1u << ((h[(b * 4 + w) * 2] >> 4) & 3);
        distinct4 += simds == 15u;
        const unsigned hw = h[b * 8], xcc = h[b * 8 + 1] & 15;
        const unsigned long long cu = ((unsigned long long)xcc << 16) | ((hw >> 8) & 0xff);   // XCC, SE, SH, CU
        by_cu[cu].push_back(b);
    }
    int pairs = 0, same_simd = 0;
    int hist[4][4] = {{0}};
    for (auto &kv : by_cu) {
        if (kv.second.size() != 2) continue;
        ++pairs;
        const int a = kv.second[0], b = kv.second[1];
        const int sa = (h[a * 8] >> 4) & 3, sb = (h[b * 8] >> 4) & 3;
        same_simd += sa == sb;
        hist[sa][sb]++;
    }
    printf("workgroups whose four waves sit on four different SIMDs: %d of %d\n", distinct4, n_wg);
    printf("CUs seen: %zu, with exactly two workgroups: %d, wave 0 of both on the SAME SIMD: %d\n", by_cu.size(), pairs, same_simd);
    for (int i = 0; i < 4; ++i) printf("  wave 0 of the first on SIMD %d: second's wave 0 on SIMD 0..3: %d %d %d %d\n", i, hist[i][0], hist[i][1], hist[i][2], hist[i][3]);
    for (int b = 0; b < 4; ++b) {
        printf("  block %d: xcc %u", b, h[b * 8 + 1] & 15);
        for (int w = 0; w < 4; ++w) printf("  w%d: simd %u wave %u tg %u cu %u se %u", w, (h[(b * 4 + w) * 2] >> 4) & 3, h[(b * 4 + w) * 2] & 15, (h[(b * 4 + w) * 2] >> 16) & 15, (h[(b * 4 + w) * 2] >> 8) & 15, (h[(b * 4 + w) * 2] >> 13) & 7);
        printf("\n");
    }
    return 0;
}
