// conv3 (64 -> 128 channels, 3x3, one 15x15 board per pass, hi + lo f16 operands, 3 MFMAs per product) as the trunk's
// MFMA loop in four shapes, on every CU, random data -- wall time, shader cycles and the clock the chip holds:
//   P32  the shipped loop: v_mfma_f32_32x32x16_f16, wave = 4 board rows (2 N-tiles of 2 rows x 16) x all 128 channels
//   C16  v_mfma_f32_16x16x32_f16, N-tile = one board row, wave = 32 output channels x all 15 rows      (channel split)
//   H16  the same MFMA, wave = 64 output channels x 8 / 7 rows                                         (2 x 2 split)
//   P16  the same MFMA, wave = all 128 output channels x 4 / 3 rows                                    (row split)
// Weights stream from L2 by buffer loads (fragment order), activations from LDS by ds_read_b128.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -o conv3_shapes conv3_shapes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) f16x8 *lds_frag;
constexpr int kRowW = 18;

__device__ __forceinline__ f16x8 load_w(__amdgpu_buffer_rsrc_t rsrc, int lane_off, int uniform_off) {
    return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, uniform_off, 0));
}

// ---------------------------------------------------------------- P32: the shipped loop (rz_net.hip, sp::conv<64, 4, 2>)
namespace p32 {
constexpr int pos_bytes = 144, piece_bytes = 324 * pos_bytes, chunks = 4, steps = 36, TM = 4, TN = 2, D = 3, DB = 2;
template <int S, int I>
__device__ __forceinline__ void slot(f32x16 (&acc)[TM][TN], f16x8 (&a)[D][TM][2], f16x8 (&b)[DB][TN][2], lds_frag q0, lds_frag q1,
                                     __amdgpu_buffer_rsrc_t w_rsrc, int w_lane) {
    constexpr int combo = I / (TM * TN), m = (I / TN) % TM, n = I % TN;
    constexpr int pa = combo == 2 ? 1 : 0, pb = combo == 1 ? 1 : 0;
    if constexpr (S == 0 && combo == 0) {
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[S % D][m][pa], b[S % DB][n][pb], zero, 0, 0, 0);
    } else {
        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[S % D][m][pa], b[S % DB][n][pb], acc[m][n], 0, 0, 0);
    }
    if constexpr (I < 2 * TN) {
        if constexpr (S + DB - 1 < steps) {
            constexpr int s1 = S + DB - 1, tap = s1 / chunks, c = s1 % chunks, nn = I / 2, piece = I % 2;
            constexpr int off = ((2 * nn + tap / 3) * kRowW + tap % 3) * pos_bytes + c * 32;
            b[s1 % DB][nn][piece] = (piece ? q1 : q0)[off / 16];
        }
    } else if constexpr (I < 2 * TN + 2 * TM) {
        if constexpr (S + D - 1 < steps) {
            constexpr int s2 = S + D - 1, j = I - 2 * TN, mm = j / 2, piece = j % 2;
            a[s2 % D][mm][piece] = load_w(w_rsrc, w_lane, ((mm * steps + s2) * 2 + piece) * 1024);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}
template <int S, int... Is>
__device__ __forceinline__ void step(std::integer_sequence<int, Is...>, f32x16 (&acc)[TM][TN], f16x8 (&a)[D][TM][2], f16x8 (&b)[DB][TN][2],
                                     lds_frag q0, lds_frag q1, __amdgpu_buffer_rsrc_t w_rsrc, int w_lane) {
    (slot<S, Is>(acc, a, b, q0, q1, w_rsrc, w_lane), ...);
}
template <int... Ss>
__device__ __forceinline__ void steps_(std::integer_sequence<int, Ss...>, f32x16 (&acc)[TM][TN], f16x8 (&a)[D][TM][2], f16x8 (&b)[DB][TN][2],
                                       lds_frag q0, lds_frag q1, __amdgpu_buffer_rsrc_t w_rsrc, int w_lane) {
    (step<Ss>(std::make_integer_sequence<int, 3 * TM * TN>{}, acc, a, b, q0, q1, w_rsrc, w_lane), ...);
}
__device__ __forceinline__ float board(const char *in, const void *wts, int wave, int lane) {
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(wts), 0, 0x7fffffff, 0x00020000);
    f16x8 a[D][TM][2];
#pragma unroll
    for (int s = 0; s < D - 1; ++s)
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int p = 0; p < 2; ++p) a[s][m][p] = load_w(w_rsrc, lane * 16, ((m * steps + s) * 2 + p) * 1024);
    const int n = lane & 31, h = lane >> 5, ry = n >> 4, x = n & 15, row0 = 4 * wave;
    const int lane_byte = ((row0 + ry) * kRowW + x) * pos_bytes + h * 16;
    const lds_frag q0 = (lds_frag)(in + lane_byte), q1 = (lds_frag)(in + lane_byte + piece_bytes);
    f16x8 b[DB][TN][2];
#pragma unroll
    for (int nn = 0; nn < TN; ++nn) {
        b[0][nn][0] = q0[(2 * nn * kRowW * pos_bytes) / 16];
        b[0][nn][1] = q1[(2 * nn * kRowW * pos_bytes) / 16];
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc[TM][TN];
    steps_(std::make_integer_sequence<int, steps>{}, acc, a, b, q0, q1, w_rsrc, lane * 16);
    float s = 0.0f;
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[m][t][r];
    return s;
}
constexpr int lds_bytes = 2 * piece_bytes;
}  // namespace p32

// ---------------------------------------------------------------- 16x16x32: N-tile = one board row, K-step = 32 channels of one tap
// LDS position record: [hi: 64 channels][lo: 64 channels][32 bytes of padding] = 288 bytes: with lane = 16 * (k block) + column,
// a ds_read_b128's 16-lane groups then hit 16 different 16-byte slots (stride / 16 = 18 = 2 mod 4)
namespace s16 {
constexpr int CIN = 64, pos_bytes = 2 * CIN * 2 + 32, chunks = CIN / 32, steps = 9 * chunks, LA = 3, AD = 3;   // B fragments requested LA rows ahead
// slot JP = TP consecutive (K-step, row) pairs J = TP * JP ..: their MFMAs interleaved so that an accumulator meets its next MFMA
// TP * TM MFMAs later; B fragments of pair J + LA requested first, A fragments of the next K-step behind its first rows
template <int TM, int NT, int TP, int JP>
__device__ __forceinline__ void slot(f32x4 (&acc)[TM][NT], f16x8 (&a)[AD][TM][2], f16x8 (&b)[TP + LA][2], lds_frag q,
                                     __amdgpu_buffer_rsrc_t w_rsrc, int w_lane) {
#pragma unroll
    for (int u = 0; u < TP; ++u) {
        constexpr int PD = TP + LA;
        const int J = JP * TP + u, J2 = J + LA;
        if (J < steps * NT && J2 < steps * NT) {
            const int s2 = J2 / NT, t2 = J2 % NT, tap = s2 / chunks, c = s2 % chunks;
            const int row = t2 + tap / 3, far = row >= 8;   // (a second base 8 rows down: the immediate offset is 16 bits)
            const int off = ((row - 8 * far) * kRowW + tap % 3) * pos_bytes + c * 64;
            const lds_frag qq = far ? q + 8 * kRowW * pos_bytes / 16 : q;
            b[J2 % PD][0] = qq[off / 16];
            b[J2 % PD][1] = qq[(off + CIN * 2) / 16];
        }
        const int s = J / NT, t = J % NT;
        if (J < steps * NT && s + 1 < steps) {
#pragma unroll
            for (int mm = 0; mm < TM; ++mm)
                if (mm % NT == t) {
#pragma unroll
                    for (int p = 0; p < 2; ++p) a[(s + 1) % AD][mm][p] = load_w(w_rsrc, w_lane, ((mm * steps + s + 1) * 2 + p) * 1024);
                }
        }
    }
#pragma unroll
    for (int combo = 0; combo < 3; ++combo)
#pragma unroll
        for (int u = 0; u < TP; ++u)
#pragma unroll
            for (int m = 0; m < TM; ++m) {
                constexpr int PD = TP + LA;
                const int J = JP * TP + u, s = J / NT, t = J % NT;
                const int pa = combo == 2 ? 1 : 0, pb = combo == 1 ? 1 : 0;
                if (J < steps * NT) {
                    if (s == 0 && combo == 0) {
                        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[s % AD][m][pa], b[J % PD][pb], zero, 0, 0, 0);
                    } else {
                        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[s % AD][m][pa], b[J % PD][pb], acc[m][t], 0, 0, 0);
                    }
                }
            }
    __builtin_amdgcn_sched_barrier(0);
}
template <int TM, int NT, int TP, int... Js>
__device__ __forceinline__ void slots(std::integer_sequence<int, Js...>, f32x4 (&acc)[TM][NT], f16x8 (&a)[AD][TM][2], f16x8 (&b)[TP + LA][2],
                                      lds_frag q, __amdgpu_buffer_rsrc_t w_rsrc, int w_lane) {
    (slot<TM, NT, TP, Js>(acc, a, b, q, w_rsrc, w_lane), ...);
}
// M-tiles m0 .. m0 + TM - 1 (16 output channels each), board rows row0 .. row0 + NT - 1
template <int TM, int NT, int TP>
__device__ __forceinline__ float board(const char *in, const char *wts, int m0, int row0, int lane) {
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(wts + (size_t)m0 * steps * 2 * 1024), 0, 0x7fffffff, 0x00020000);
    f16x8 a[AD][TM][2];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int p = 0; p < 2; ++p) a[0][m][p] = load_w(w_rsrc, lane * 16, ((m * steps) * 2 + p) * 1024);
    const int n = lane & 15, g = lane >> 4;
    const lds_frag q = (lds_frag)(in + (row0 * kRowW + n) * pos_bytes + g * 16);
    f16x8 b[TP + LA][2];
    static_assert(LA <= 3, "the first fragments: rows of step 0");
#pragma unroll
    for (int j = 0; j < LA; ++j) {
        b[j][0] = q[(j * kRowW * pos_bytes) / 16];
        b[j][1] = q[(j * kRowW * pos_bytes + CIN * 2) / 16];
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[TM][NT];
    slots<TM, NT, TP>(std::make_integer_sequence<int, (steps * NT + TP - 1) / TP>{}, acc, a, b, q, w_rsrc, lane * 16);
    float s = 0.0f;
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += acc[m][t][r];
    return s;
}
constexpr int lds_bytes = 324 * pos_bytes;
}  // namespace s16

// ---------------------------------------------------------------- R16: the C16 split with the K loop turned inside out: for
// every (tap column dx, channel chunk) the 17 halo rows are read ONCE each and a row's fragment meets the three kernel rows
// (output rows r, r - 1, r - 2): a third of C16's LDS reads, the same weight traffic, 18 MFMAs per pair of ds_read_b128
namespace r16 {
using s16::CIN; using s16::pos_bytes; using s16::chunks;
constexpr int NT = 15, TM = 2, LA = 3, PD = LA + 1, combos = 3 * chunks, rows = NT + 2;   // combo = dx * chunks + chunk
// weight fragment of (M-tile m, tap (dy, dx), chunk c): the layout of s16 (step = tap * chunks + c)
__device__ __forceinline__ constexpr int w_off(int m, int dy, int combo, int p) {
    return ((m * s16::steps + (dy * 3 + combo / chunks) * chunks + combo % chunks) * 2 + p) * 1024;
}
template <int J>
__device__ __forceinline__ void slot(f32x4 (&acc)[TM][NT], f16x8 (&a)[2][3][TM][2], f16x8 (&b)[PD][2], lds_frag q,
                                     __amdgpu_buffer_rsrc_t w_rsrc, int w_lane) {
    constexpr int cb = J / rows, r = J % rows, J2 = J + LA;
    if constexpr (J2 < combos * rows) {
        constexpr int cb2 = J2 / rows, r2 = J2 % rows, dx = cb2 / chunks, c = cb2 % chunks, far = r2 >= 8;
        constexpr int off = ((r2 - 8 * far) * kRowW + dx) * pos_bytes + c * 64;
        const lds_frag qq = far ? q + 8 * kRowW * pos_bytes / 16 : q;
        b[J2 % PD][0] = qq[off / 16];
        b[J2 % PD][1] = qq[(off + CIN * 2) / 16];
    }
    if constexpr (cb + 1 < combos && r < 3 * TM) {   // the next combo's 12 weight fragments behind this combo's first rows
        constexpr int dy = r / TM, m = r % TM;
#pragma unroll
        for (int p = 0; p < 2; ++p) a[(cb + 1) % 2][dy][m][p] = load_w(w_rsrc, w_lane, w_off(m, dy, cb + 1, p));
    }
#pragma unroll
    for (int combo = 0; combo < 3; ++combo)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int m = 0; m < TM; ++m) {
                const int t = r - dy, pa = combo == 2 ? 1 : 0, pb = combo == 1 ? 1 : 0;
                if (t >= 0 && t < NT) {
                    if (cb == 0 && dy == 0 && combo == 0) {   // the first MFMA of a tile (every row meets dy = 0 in combo 0 first)
                        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[cb % 2][dy][m][pa], b[J % PD][pb], zero, 0, 0, 0);
                    } else {
                        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[cb % 2][dy][m][pa], b[J % PD][pb], acc[m][t], 0, 0, 0);
                    }
                }
            }
    __builtin_amdgcn_sched_barrier(0);
}
template <int... Js>
__device__ __forceinline__ void slots(std::integer_sequence<int, Js...>, f32x4 (&acc)[TM][NT], f16x8 (&a)[2][3][TM][2], f16x8 (&b)[PD][2],
                                      lds_frag q, __amdgpu_buffer_rsrc_t w_rsrc, int w_lane) {
    (slot<Js>(acc, a, b, q, w_rsrc, w_lane), ...);
}
__device__ __forceinline__ float board(const char *in, const char *wts, int m0, int lane) {
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(wts + (size_t)m0 * s16::steps * 2 * 1024), 0, 0x7fffffff, 0x00020000);
    f16x8 a[2][3][TM][2];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int p = 0; p < 2; ++p) a[0][dy][m][p] = load_w(w_rsrc, lane * 16, w_off(m, dy, 0, p));
    const int n = lane & 15, g = lane >> 4;
    const lds_frag q = (lds_frag)(in + n * pos_bytes + g * 16);
    f16x8 b[PD][2];
#pragma unroll
    for (int j = 0; j < LA; ++j) {
        b[j][0] = q[(j * kRowW * pos_bytes) / 16];
        b[j][1] = q[(j * kRowW * pos_bytes + CIN * 2) / 16];
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[TM][NT];
    slots(std::make_integer_sequence<int, combos * rows>{}, acc, a, b, q, w_rsrc, lane * 16);
    float s = 0.0f;
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += acc[m][t][r];
    return s;
}
}  // namespace r16


// ---------------------------------------------------------------- F8: the C16 split with the CROSS TERMS on the block-scaled FP8 pipe
// (DESIGN / TRIED "FP8 cross terms"): per tap and N-tile two v_mfma_f32_16x16x32_f16 (hi x hi, 2 x 32 channels) and ONE
// v_mfma_scale_f32_16x16x128_f8f6f4 whose K = 128 is [hi8 x (w_lo 2^k)8 | (lo 2^k)8 x w_hi8] of the tap's 64 channels -- 2 f16-MFMA
// equivalents instead of 3 per product.  LDS: one region of hi16 records, one of fp8 records (hi8 | lo8), 144 bytes each; the same
// bytes per position and per weight fragment as C16.  A timing gate only: the bytes are random.
namespace f8 {
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(3))) i32x4 *lds_q;
constexpr int CIN = 64, rec = 144, region = 324 * rec, steps = 9, LA = 3, AD = 2;
struct Frag { f16x8 h[2]; i32x8 q; };
__device__ __forceinline__ i32x8 cat(i32x4 lo, i32x4 hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }
template <int TM, int NT, int TP, int JP>
__device__ __forceinline__ void slot(f32x4 (&acc)[TM][NT], Frag (&a)[AD][TM], Frag (&b)[TP + LA], lds_frag q16, lds_q q8,
                                     __amdgpu_buffer_rsrc_t w_rsrc, int w_lane, int sa, int sb) {
    constexpr int PD = TP + LA;
#pragma unroll
    for (int u = 0; u < TP; ++u) {
        const int J = JP * TP + u, J2 = J + LA;
        if (J < steps * NT && J2 < steps * NT) {
            const int tap = J2 / NT, t2 = J2 % NT;
            const int row = t2 + tap / 3, far = row >= 8;
            const int off = ((row - 8 * far) * kRowW + tap % 3) * rec;
            const lds_frag p16 = far ? q16 + 8 * kRowW * rec / 16 : q16;
            const lds_q p8 = far ? q8 + 8 * kRowW * rec / 16 : q8;
            b[J2 % PD].h[0] = p16[off / 16];
            b[J2 % PD].h[1] = p16[(off + 64) / 16];
            b[J2 % PD].q = cat(p8[off / 16], p8[(off + 16) / 16]);
        }
        const int s = J / NT, t = J % NT;
        if (J < steps * NT && s + 1 < steps) {
#pragma unroll
            for (int mm = 0; mm < TM; ++mm)
                if (mm % NT == t) {
                    const int base = (mm * steps + s + 1) * 4 * 1024;
                    a[(s + 1) % AD][mm].h[0] = load_w(w_rsrc, w_lane, base);
                    a[(s + 1) % AD][mm].h[1] = load_w(w_rsrc, w_lane, base + 1024);
                    a[(s + 1) % AD][mm].q = cat(__builtin_bit_cast(i32x4, load_w(w_rsrc, w_lane, base + 2048)),
                                                __builtin_bit_cast(i32x4, load_w(w_rsrc, w_lane, base + 3072)));
                }
        }
    }
#pragma unroll
    for (int combo = 0; combo < 3; ++combo)
#pragma unroll
        for (int u = 0; u < TP; ++u)
#pragma unroll
            for (int m = 0; m < TM; ++m) {
                const int J = JP * TP + u, s = J / NT, t = J % NT;
                if (J < steps * NT) {
                    if (combo < 2) {
                        if (s == 0 && combo == 0) {
                            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                            acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[s % AD][m].h[0], b[J % PD].h[0], zero, 0, 0, 0);
                        } else {
                            acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[s % AD][m].h[combo], b[J % PD].h[combo], acc[m][t], 0, 0, 0);
                        }
                    } else {
                        acc[m][t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[s % AD][m].q, b[J % PD].q, acc[m][t], 0, 0, 0, sa, 0, sb);
                    }
                }
            }
    __builtin_amdgcn_sched_barrier(0);
}
template <int TM, int NT, int TP, int... Js>
__device__ __forceinline__ void slots(std::integer_sequence<int, Js...>, f32x4 (&acc)[TM][NT], Frag (&a)[AD][TM], Frag (&b)[TP + LA],
                                      lds_frag q16, lds_q q8, __amdgpu_buffer_rsrc_t w_rsrc, int w_lane, int sa, int sb) {
    (slot<TM, NT, TP, Js>(acc, a, b, q16, q8, w_rsrc, w_lane, sa, sb), ...);
}
template <int TM, int NT, int TP>
__device__ __forceinline__ float board(const char *in, const char *wts, int m0, int row0, int lane, int sa, int sb) {
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(wts + (size_t)m0 * steps * 4 * 1024), 0, 0x7fffffff, 0x00020000);
    Frag a[AD][TM];
#pragma unroll
    for (int m = 0; m < TM; ++m) {
        a[0][m].h[0] = load_w(w_rsrc, lane * 16, m * steps * 4 * 1024);
        a[0][m].h[1] = load_w(w_rsrc, lane * 16, m * steps * 4 * 1024 + 1024);
        a[0][m].q = cat(__builtin_bit_cast(i32x4, load_w(w_rsrc, lane * 16, m * steps * 4 * 1024 + 2048)),
                        __builtin_bit_cast(i32x4, load_w(w_rsrc, lane * 16, m * steps * 4 * 1024 + 3072)));
    }
    const int n = lane & 15, g = lane >> 4;
    const lds_frag q16 = (lds_frag)(in + (row0 * kRowW + n) * rec + g * 16);
    const lds_q q8 = (lds_q)(in + region + (row0 * kRowW + n) * rec + g * 32);
    Frag b[TP + LA];
#pragma unroll
    for (int j = 0; j < LA; ++j) {
        b[j].h[0] = q16[(j * kRowW * rec) / 16];
        b[j].h[1] = q16[(j * kRowW * rec + 64) / 16];
        b[j].q = cat(q8[(j * kRowW * rec) / 16], q8[(j * kRowW * rec + 16) / 16]);
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[TM][NT];
    slots<TM, NT, TP>(std::make_integer_sequence<int, (steps * NT + TP - 1) / TP>{}, acc, a, b, q16, q8, w_rsrc, lane * 16, sa, sb);
    float s = 0.0f;
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += acc[m][t][r];
    return s;
}
constexpr int lds_bytes = 2 * region;
}  // namespace f8

enum { P32, C16, H16, P16, C16x1, C16x3, R16, F8x1, F8x2 };
template <int KIND>
__global__ __launch_bounds__(256) void k(const _Float16 *__restrict__ act, const char *__restrict__ wts, float *out, long long *ticks,
                                         int boards) {
    constexpr int kLds = KIND == P32 ? p32::lds_bytes : s16::lds_bytes;
    __shared__ __attribute__((aligned(16))) char lds[kLds];
    for (int i = threadIdx.x; i < kLds / 2; i += 256) reinterpret_cast<_Float16 *>(lds)[i] = act[i];
    __syncthreads();
    int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float s = 0.0f;
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int bd = 0; bd < boards; ++bd) {
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63;
        if constexpr (KIND == P32) s += p32::board(lds, wts, wave, lane);
        if constexpr (KIND == C16) s += s16::board<2, 15, 2>(lds, wts, 2 * wave, 0, lane);
        if constexpr (KIND == C16x1) s += s16::board<2, 15, 1>(lds, wts, 2 * wave, 0, lane);
        if constexpr (KIND == C16x3) s += s16::board<2, 15, 3>(lds, wts, 2 * wave, 0, lane);
        if constexpr (KIND == R16) s += r16::board(lds, wts, 2 * wave, lane);
        if constexpr (KIND == F8x1) s += f8::board<2, 15, 1>(lds, wts, 2 * wave, 0, lane, boards + 115, 127);
        if constexpr (KIND == F8x2) s += f8::board<2, 15, 2>(lds, wts, 2 * wave, 0, lane, boards + 115, 127);
        if constexpr (KIND == H16) {
            if (wave < 2) s += s16::board<4, 8, 1>(lds, wts, 4 * wave, 0, lane);
            else s += s16::board<4, 7, 1>(lds, wts, 4 * (wave - 2), 8, lane);
        }
        if constexpr (KIND == P16) {
            if (wave < 3) s += s16::board<8, 4, 1>(lds, wts, 0, 4 * wave, lane);
            else s += s16::board<8, 3, 1>(lds, wts, 0, 12, lane);
        }
        __syncthreads();
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        ticks[2 * blockIdx.x] = t1 - t0;
        ticks[2 * blockIdx.x + 1] = r1 - r0;
    }
}

template <int KIND>
void run(const char *name, int grid, const _Float16 *act, const char *wts, float *out, long long *ticks, int boards) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) k<KIND><<<grid, 256>>>(act, wts, out, ticks, boards);   // warm the clocks
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<KIND><<<grid, 256>>>(act, wts, out, ticks, boards);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(2 * grid);
    (void)hipMemcpy(h.data(), ticks, grid * 16, hipMemcpyDeviceToHost);
    double cyc = 0, real = 0;
    for (int i = 0; i < grid; ++i) { cyc += h[2 * i]; real += h[2 * i + 1]; }
    printf("%-24s grid %3d: %7.2f us and %7.0f cycles per board (conv3 only), clock %.2f GHz\n", name, grid, ms * 1e3 / boards,
           cyc / grid / boards, cyc / real * 0.1);
    fflush(stdout);
}

int main() {
    const size_t act_n = 100 * 1024, w_bytes = 8 * 18 * 2 * 1024 + 4096;   // 8 M-tiles of 16 (= 4 of 32 x 36 steps) x hi / lo
    std::vector<_Float16> ha(act_n), hw(w_bytes / 2);
    srand(1);
    for (auto &v : ha) v = (_Float16)((rand() % 2001 - 1000) * 0.004f);
    for (auto &v : hw) v = (_Float16)((rand() % 2001 - 1000) * 0.01f);
    _Float16 *act; char *wts; float *out; long long *ticks;
    (void)hipMalloc(&act, act_n * 2); (void)hipMalloc(&wts, w_bytes); (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&ticks, 256 * 16);
    (void)hipMemcpy(act, ha.data(), act_n * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(wts, hw.data(), w_bytes, hipMemcpyHostToDevice);
    const int boards = 400;
    for (int rep = 0; rep < 2; ++rep)
        for (int grid : {256, 32}) {
            run<P32>("P32", grid, act, wts, out, ticks, boards);
            run<C16>("C16", grid, act, wts, out, ticks, boards);
            run<H16>("H16", grid, act, wts, out, ticks, boards);
            run<P16>("P16", grid, act, wts, out, ticks, boards);
            run<C16x1>("C16 one row per slot", grid, act, wts, out, ticks, boards);
            run<C16x3>("C16 three rows per slot", grid, act, wts, out, ticks, boards);
            run<R16>("R16 halo rows read once", grid, act, wts, out, ticks, boards);
            run<F8x1>("F8 cross terms, 1 row/slot", grid, act, wts, out, ticks, boards);
            run<F8x2>("F8 cross terms, 2 rows/slot", grid, act, wts, out, ticks, boards);
        }
    return 0;
}
