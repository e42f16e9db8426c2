// rz_net.hip -- fused forward of the AlphaZero policy-value network on MI355X (gfx950), f32 results.
//
// Replaces PolicyValueNet.forward (rlzero/games/gomoku/policy_value_net.py:34-52) for the
// batch of MCTS leaves: the one dense contraction of the path (42.8 MFLOP per
// 15x15 position, MFMA-bound).  The default trunk carries f32 operands as hi + lo f16 pairs on the f16 matrix
// pipe (three MFMAs per product, f32 accumulation: as accurate as the exact-f32 kernel) and so does the FC GEMM
// behind it; the other trunks and their FC GEMM use v_mfma_f32_16x16x4_f32, a k-ordered fmaf chain (tolerance vs
// the reference's CPU output: 1e-4).
//
// Kernels (one workgroup owns a board; its activations never leave the CU: input planes, conv1 output (32 ch)
// and conv2 output (64 ch) live in LDS as halo-padded planes [channel][18 rows][18 cols]; conv3's 128 channels
// stay in registers and are consumed by the two 1x1 head convolutions there):
//   k_trunk_rows<NT> (default,    conv1..conv3 as direct convolutions on the f16 matrix pipe, every f32 operand a hi + lo pair of
//     boards of 11 .. 16 rows     f16 values: v_mfma_f32_16x16x32_f16, one N-tile per board row, the four waves split the OUTPUT
//     and columns)                CHANNELS, rows innermost in the K loop; persistent workgroups (layouts in LDS: see the kernel)
//   k_trunk_split<TN, MS>         the same arithmetic on v_mfma_f32_32x32x16_f16 tiles of whole rows, the waves split the rows
//     (smaller boards)            (TN = 1: one tile per wave, MS: idle waves take channel shares); optionally the FC layers of
//                                 its own board behind it (RZ_NET_HEADS_IN_TRUNK); the checker of k_trunk_rows
//   k_trunk_wino_f4<4>            conv2 / conv3 as Winograd F(4x4,3x3) on the f32-input MFMA, persistent workgroups, 4 waves
//   k_trunk                       direct implicit GEMM (bit-for-bit a k-ordered fmaf chain): M = output channels
//                                 (A = weights, pre-packed on the host in fragment order, streamed from L2), N =
//                                 the 16 columns of a board row (B = ds_read_b32 from the halo planes), K =
//                                 (group of 4 input channels, tap); wave w = 4*rh + q4 owns channel quarter q4
//                                 and row half rh
//   k_heads_split                 the first FC layers of both heads on the f16 pipe (hi + lo pairs), fed by the f16
//                                 feature pieces k_trunk_split writes in MFMA fragment order (default behind it)
//   k_heads_gemm, k_heads_finish  the same layers on the f32-input MFMA (32 x 32 output blocks); log_softmax and tanh
// Every kernel and its design notes are described where it is defined.

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <utility>
#include <vector>

#include "rlzero_hip.h"
#include "rz_trace.h"
#include "rz_tree.h"

void rz_set_error(const char *msg);  // rz_engine.hip

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kRowW = 18;     // halo row width (x = -1 .. 16)
// halo rows: 18 (y = -1 .. 16), 18*18 = 324 floats per plane before padding
// plane stride (floats): 18*18 = 324 padded.  Direct kernel: 336 = 16 mod 32, so the 4 channel
// sub-groups of a fragment read hit disjoint banks.  Winograd kernel: 337 (odd) -- see wino_conv.
constexpr int kPlaneDirect = 336, kPlaneWino = 337;
constexpr int kPlanesIn = 4, kPlanesC1 = 32, kPlanesC2 = 64, kPlanes = kPlanesIn + kPlanesC1 + kPlanesC2;
constexpr int kTrunkThreads = 512;

struct NetDev {
    const f32x4 *w1, *w2, *w3;   // packed [tile][cin_step][3][64 lanes] x 4 taps
    const f32x4 *u2f, *u3f;      // F(4x4,3x3): [tile][pass][cin_step][3][64 lanes] x 4 components (pack_wino_f4)
    const f32x4 *s1;             // conv1 for k_trunk_split: [kernel row][hi | lo][64 lanes] x 8 f16 (pack_split1)
    const f32x4 *s2, *s3;        // split f16 weights: [32-channel tile][tap][16-channel chunk][hi | lo][64 lanes] x 8 f16
    const f32x4 *t2, *t3;        // the same for k_trunk_rows: [16-channel tile][tap][32-channel chunk][hi | lo][64 lanes] x 8 f16 (pack_rows)
    const f32x4 *t3f;            // conv3 for the FP8 cross terms (RZ_NET_SPLIT_F16_FP8): [16-channel tile][tap][part][half][64 lanes] x 16 bytes
                                 // (pack_rows_f8: part 0 = the hi f16 pieces of the tap's two chunks, part 1 = e4m3 bytes [lo 2^5 | hi 2^-6])
    const float *s_inv;          // [8] in device memory (a captured launch must see a reload's values), with a1, a2, a3 =
                                 // the activation scales of conv1's / conv2's outputs and of the head features (powers of
                                 // two from rz_net_load's activation bounds), sw* the weight scales:
                                 // [0] a2 / (a1 sw2), [1] 1 / (a2 sw3), [2] a1 / (16 sw1), [3] 1 / (a3 sw_act_fc1),
                                 // [4] 1 / (a3 sw_val_fc1), [5] a1, [6] a2, [7] a3
    const f32x4 *fs_act, *fs_val;  // split f16 FC weights: [32-output tile][K-step of 16][hi | lo][64 lanes] x 8 f16 (+ a zero step)
    const float *b1, *b2, *b3;   // conv biases
    const float *wh;             // [6][128]: act_conv1 (4 rows) then val_conv1 (2 rows)
    const float *whp;            // the same, [128][6] (k_trunk_split)
    const float *bh;             // [6]
    const float *fc_act_w;       // act_fc1.weight [out][in], zero padded to [Npad][16*groups_act]
    const float *fc_act_b;       // [Npad]
    const float *fc_val1_w;      // val_fc1.weight [out][in], zero padded to [64][16*groups_val]
    const float *fc_val1_b;      // [64]
    const float *fc_val2_w;      // [64]
    const float *fc_val2_b;      // [1]
    // head features of a board: policy inputs at [0, 4S), value inputs at [feat_val_off, +2S) of a row of
    // feat_ld floats.  A caller's buffer is the natural [board][6S]; the internal one pads both ranges
    // to multiples of 16 (zero filled) so the FC GEMM reads aligned 16-byte fragments.
    int feat_ld, feat_val_off;
    int BH, BW, S, A, Npad, groups_act, groups_val;  // A policy outputs (Npad: padded to 32); groups_*: K / 16
    // k_trunk_split: an N-tile (32 MFMA columns) = tile_rows board rows x tile_cols columns, position n of a tile =
    // (n / tile_cols, n % tile_cols) with n / tile_cols = (n * tile_rcp) >> 16; (2, 16) for boards that need 5 .. 8
    // tiles, (32 / width, width) when 4 tiles of that shape cover the board (9x9: 3 x 9, Connect4: 4 x 7)
    int tile_rows, tile_cols, tile_rcp;
};

// Operand fragments of one input-channel group (4 channels x 9 taps) for a wave that owns TM
// output-channel tiles and NR board rows: TM x 3 packed weight vectors and the (NR+2) x 3
// distinct (row, dx) activation fragments that the 9 taps x NR rows reuse.
template <int TM, int NR>
struct Frags {
    f32x4 a[TM][3];
    float b[NR + 2][3];
};

template <int PL, int TM, int NR, int STEPS>
__device__ __forceinline__ void load_frags(Frags<TM, NR> &f, const float *__restrict__ base,
                                           const f32x4 *__restrict__ wbase, int s) {
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int tg = 0; tg < 3; ++tg) f.a[m][tg] = wbase[((size_t)(m * STEPS + s) * 3 + tg) * 64];
    const float *p = base + (4 * s) * PL;
#pragma unroll
    for (int ro = 0; ro < NR + 2; ++ro)
#pragma unroll
        for (int dxi = 0; dxi < 3; ++dxi) f.b[ro][dxi] = p[ro * kRowW + dxi];
}

template <int TM, int NR>
__device__ __forceinline__ void mfma_group(const Frags<TM, NR> &f, f32x4 (&acc)[TM][8]) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const int dyi = tap / 3, dxi = tap % 3;
#pragma unroll
        for (int m = 0; m < TM; ++m) {
            const float av = f.a[m][tap / 4][tap % 4];
#pragma unroll
            for (int t = 0; t < NR; ++t)
                acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, f.b[t + dyi][dxi], acc[m][t], 0, 0, 0);
        }
    }
}

// acc[m][t] += W(tile tile0+m) x in(rows row0+t) over all CIN input channels and the 9 taps.
// Software pipelined: the fragments of channel group s+1 are fetched (weights: 16-byte loads
// from L2, activations: ds_read_b32) while the 9*TM*NR MFMAs of group s issue.
template <int PL, int CIN, int TM, int NR>
__device__ __forceinline__ void conv_accumulate(const float *__restrict__ in, const f32x4 *__restrict__ wp,
                                                int tile0, int row0, int lane, f32x4 (&acc)[TM][8]) {
    constexpr int kSteps = CIN / 4;
    const int x = lane & 15, kq = lane >> 4;
    const float *base = in + kq * PL + row0 * kRowW + x;  // in[(4s+kq)][row0 + ro][x + dxi]
    const f32x4 *wbase = wp + (size_t)tile0 * kSteps * 3 * 64 + lane;
    Frags<TM, NR> f0, f1;
    load_frags<PL, TM, NR, kSteps>(f0, base, wbase, 0);
#pragma unroll 1
    // sched_barrier(0) pins "issue every load of the next group, THEN the MFMAs of this one":
    // left alone, hipcc sinks each load next to its first use and the MFMAs wait on it.
    for (int s = 0; s < kSteps; s += 2) {
        load_frags<PL, TM, NR, kSteps>(f1, base, wbase, s + 1 < kSteps ? s + 1 : kSteps - 1);
        __builtin_amdgcn_sched_barrier(0);
        mfma_group<TM, NR>(f0, acc);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (kSteps > 1) {  // kSteps is 1 (conv1) or even
            load_frags<PL, TM, NR, kSteps>(f0, base, wbase, s + 2 < kSteps ? s + 2 : kSteps - 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_group<TM, NR>(f1, acc);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int PL, int CIN, int TM>
__device__ __forceinline__ void conv_rows(const float *__restrict__ in, const f32x4 *__restrict__ wp, int tile0,
                                          int row0, int n_rows, int lane, f32x4 (&acc)[TM][8]) {
    if (n_rows == 7) conv_accumulate<PL, CIN, TM, 7>(in, wp, tile0, row0, lane, acc);
    else conv_accumulate<PL, CIN, TM, 8>(in, wp, tile0, row0, lane, acc);
}

// out[cout][y+1][x+1] = relu(acc + bias[cout]) for the lane's 4 channels of every tile/row.
template <int PL, int TM>
__device__ __forceinline__ void store_relu(float *__restrict__ out, const float *__restrict__ bias, int tile0,
                                           int row0, int lane, int BH, int BW, const f32x4 (&acc)[TM][8],
                                           int n_rows = 8) {
    const int x = lane & 15, q = lane >> 4;
    if (x >= BW) return;
#pragma unroll
    for (int m = 0; m < TM; ++m) {
        const int c0 = (tile0 + m) * 16 + 4 * q;
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + c0);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int y = row0 + t;
            if (y >= BH || t >= n_rows) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                out[(c0 + j) * PL + (y + 1) * kRowW + (x + 1)] = fmaxf(acc[m][t][j] + bv[j], 0.0f);
        }
    }
}

template <int TM>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[TM][8]) {
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// helpers shared by the Winograd F(4x4,3x3) trunk
typedef const __attribute__((address_space(3))) float *lds_cptr;
typedef int i32x4 __attribute__((ext_vector_type(4)));

// U fragment load: buffer addressing = scalar resource + the lane's 32-bit offset + a scalar byte
// offset, so the address of every load of the stream costs SALU only.
__device__ __forceinline__ f32x4 load_u(__amdgpu_buffer_rsrc_t rsrc, int lane_off, int uniform_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, uniform_off, 0));
}

// =====================================================================================================
// Winograd F(4x4,3x3) trunk (RZ_NET_WINOGRAD_F4): 36 element-wise products per 4x4 output tile instead of
// 144 multiply-adds -- 4x fewer MFMAs than the direct form, 1.78x fewer than F(2x2,3x3).  fp32
// throughout; the larger transform constants cost about one decimal digit (|error| ~1e-6 on the conv3
// activations against fp64, ~5e-7 on the log-probabilities: tests/test_gpu_parity.py).
//   * the MFMA N dimension is the WHOLE board: 16 tiles of 4x4 outputs (lane & 15 = 4*ty + tx); K = input
//     channels (4 per step, lane >> 4); M = 16 output channels;
//   * 4 waves per workgroup, one per SIMD (up to 512 registers): a wave owns TM output-channel tiles (conv3: 2,
//     conv2: 1) and ALL 36 components, taken in 3 passes over the channel groups, two transform rows (12
//     components) per pass -- {1,2}, {3,4}, {0,5} -- so 12*TM accumulators are live, and each pass folds
//     its rows into the 4x4 outputs (the output transform A^T M A is linear in M);
//   * per channel group a lane reads its 6x6 input patch rows from the halo planes (4 rows for the
//     first two passes, 6 for the last), forms the 12 transformed values (row stage then column stage,
//     52..64 fused multiply-adds) and issues 12*TM MFMAs against U fragments streamed from L2 with buffer
//     loads; the software pipeline of wino_block is kept: the transform of group g+1, the LDS reads of
//     g+2 and the U loads of g+3 are threaded between the MFMAs of group g.
namespace f4 {

// Everything below is indexed by template parameters and expanded with fold expressions (not loops the
// unroller may decline to unroll: the op lists are long), so every register-array index is a constant.

template <int P> struct Pass {  // pass P handles transform rows i' = a, b
    static constexpr int a = (P == 0) ? 1 : (P == 1) ? 3 : 0;
    static constexpr int b = (P == 0) ? 2 : (P == 1) ? 4 : 5;
    static constexpr int first = (P == 2) ? 0 : 1;   // patch rows it reads: 1..4, or all six for rows 0 / 5
    static constexpr int count = (P == 2) ? 6 : 4;
    static constexpr int n_ld = 3 * count;           // ds_read2 per group
    static constexpr int row_ops = (P == 2) ? 4 : 3; // row-stage instructions per patch column
    static constexpr int n_xf = 6 * row_ops + 14;    // transform instructions per group (packed: 2 floats each)
};

// LDS read O of a group of pass P: patch row first + O / 3, column pair O % 3
template <int P, int O>
__device__ __forceinline__ void ld_op(float (&d)[6][6], lds_cptr q) {
    constexpr int row = Pass<P>::first + O / 3, c = 2 * (O % 3);
    d[row][c] = q[row * kRowW + c];
    d[row][c + 1] = q[row * kRowW + c + 1];
}

// Transform instruction O of a group of pass P; every case is ONE instruction, most of them PACKED fp32
// (v_pk_add_f32 / v_pk_fma_f32 on register pairs, with half selects / negations as operand modifiers):
// the two transform rows a, b of the pass go through identical arithmetic, so the pair (row a, row b) is
// the natural vector.  Row stage, column by column: t[c] = (rows a, b of B^T d)[c]; then the column stage
// v[j'] = (t B)[j'] on those pairs,
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1].
template <int P, int O>
__device__ __forceinline__ void xf_op(const float (&d)[6][6], f32x2 (&w)[2], f32x2 (&t)[6], f32x2 (&u)[8], f32x2 (&v)[6]) {
    constexpr int kRowOps = Pass<P>::row_ops;
    if constexpr (O < 6 * kRowOps) {
        constexpr int c = O / kRowOps, k = O % kRowOps;
        if constexpr (P == 0) {  // rows 1, 2: -4 (d1 + d2) + (d3 + d4) ; 4 (d1 - d2) - (d3 - d4)
            if constexpr (k == 0) w[0] = f32x2{d[1][c], d[1][c]} + f32x2{d[2][c], -d[2][c]};
            else if constexpr (k == 1) w[1] = f32x2{d[4][c], d[4][c]} + f32x2{d[3][c], -d[3][c]};
            else t[c] = __builtin_elementwise_fma(f32x2{-4.0f, 4.0f}, w[0], w[1]);
        } else if constexpr (P == 1) {  // rows 3, 4: +-2 (d3 - d1) + (d4 - d2)
            if constexpr (k == 0) w[0].x = d[3][c] - d[1][c];
            else if constexpr (k == 1) w[0].y = d[4][c] - d[2][c];
            else t[c] = __builtin_elementwise_fma(f32x2{2.0f, -2.0f}, f32x2{w[0].x, w[0].x}, f32x2{w[0].y, w[0].y});
        } else {  // rows 0, 5: 4 d0 - 5 d2 + d4 ; 4 d1 - 5 d3 + d5
            if constexpr (k == 0) w[0].x = fmaf(-5.0f, d[2][c], d[4][c]);
            else if constexpr (k == 1) t[c].x = fmaf(4.0f, d[0][c], w[0].x);
            else if constexpr (k == 2) w[0].y = fmaf(-5.0f, d[3][c], d[5][c]);
            else t[c].y = fmaf(4.0f, d[1][c], w[0].y);
        }
    } else {
        constexpr int k = O - 6 * kRowOps;
        const f32x2 c4 = {4.0f, 4.0f}, c2 = {2.0f, 2.0f}, c5 = {-5.0f, -5.0f};
        if constexpr (k == 0) u[0] = t[1] + t[2];
        else if constexpr (k == 1) u[1] = t[3] + t[4];
        else if constexpr (k == 2) u[2] = t[1] - t[2];
        else if constexpr (k == 3) u[3] = t[3] - t[4];
        else if constexpr (k == 4) v[1] = __builtin_elementwise_fma(-c4, u[0], u[1]);
        else if constexpr (k == 5) v[2] = __builtin_elementwise_fma(c4, u[2], -u[3]);
        else if constexpr (k == 6) u[4] = t[3] - t[1];
        else if constexpr (k == 7) u[5] = t[4] - t[2];
        else if constexpr (k == 8) v[3] = __builtin_elementwise_fma(c2, u[4], u[5]);
        else if constexpr (k == 9) v[4] = __builtin_elementwise_fma(-c2, u[4], u[5]);
        else if constexpr (k == 10) u[6] = __builtin_elementwise_fma(c5, t[2], t[4]);
        else if constexpr (k == 11) v[0] = __builtin_elementwise_fma(c4, t[0], u[6]);
        else if constexpr (k == 12) u[7] = __builtin_elementwise_fma(c5, t[3], t[5]);
        else v[5] = __builtin_elementwise_fma(c4, t[1], u[7]);
    }
}
template <int P, int BASE, int... Os>
__device__ __forceinline__ void xf_ops(std::integer_sequence<int, Os...>, const float (&d)[6][6], f32x2 (&w)[2],
                                       f32x2 (&t)[6], f32x2 (&u)[8], f32x2 (&v)[6]) {
    (xf_op<P, BASE + Os>(d, w, t, u, v), ...);
}
template <int P, int BASE, int... Os>
__device__ __forceinline__ void ld_ops(std::integer_sequence<int, Os...>, float (&d)[6][6], lds_cptr q) {
    (ld_op<P, BASE + Os>(d, q), ...);
}

// Slot I of the MFMA block of one channel group: MFMA I (component k = I / TM = rr*6 + j', tile m = I % TM) and its
// slice of the next groups' work: first third of the slots = LDS reads of the group two ahead (pass LP) into
// d_ld and the U loads two groups ahead; the other two thirds = transform of the next group (pass XP), whose
// patch rows d_xf were fetched during the PREVIOUS block (the patch buffer is double buffered: with one wave
// per SIMD nothing else covers the LDS latency), into v_nxt.
template <int TM, int XP, int LP, int I>
__device__ __forceinline__ void slot(f32x4 (&acc)[TM][12], const f32x4 (&a_cur)[TM][3], const f32x2 (&v_cur)[6],
                                     f32x2 (&v_nxt)[6], const float (&d_xf)[6][6], float (&d_ld)[6][6], f32x2 (&w)[2],
                                     f32x2 (&t)[6], f32x2 (&u)[8], lds_cptr q_ld, f32x4 (&a_ld)[TM][3],
                                     __amdgpu_buffer_rsrc_t u_rsrc, int u_off, int u_lane, int u_stride) {
    constexpr int NS = 12 * TM, LD_SLOTS = NS / 3, XF_SLOTS = NS - LD_SLOTS;
    constexpr int kXf = Pass<XP>::n_xf, kLd = Pass<LP>::n_ld;
    constexpr int XF_PER = (kXf + XF_SLOTS - 1) / XF_SLOTS, LD_PER = (kLd + LD_SLOTS - 1) / LD_SLOTS;
    constexpr int k = I / TM, m = I % TM;
    // In-place accumulation on accumulation registers, written as inline assembly: with more than 256
    // registers per wave the compiler otherwise stages every accumulator through a[0:3] and copies it to
    // and from ordinary registers around each MFMA (8 extra instructions per MFMA).  No hazard handling is
    // lost: consecutive MFMAs use different accumulators (the same one recurs 12*TM MFMAs later) and the
    // operands were produced in the previous block; the fold waits explicitly.
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0"
                 : "+a"(acc[m][k])
                 : "v"(a_cur[m][k >> 2][k & 3]), "v"(v_cur[k % 6][k / 6]));
    if constexpr (I < LD_SLOTS) {
        constexpr int lo = I * LD_PER, hi = (lo + LD_PER < kLd) ? lo + LD_PER : kLd;
        if constexpr (hi > lo) ld_ops<LP, lo>(std::make_integer_sequence<int, hi - lo>{}, d_ld, q_ld);
        if constexpr (I < 3) {
#pragma unroll
            for (int m2 = 0; m2 < TM; ++m2) a_ld[m2][I] = load_u(u_rsrc, u_lane, u_off + m2 * u_stride + I * 1024);
        }
    } else {
        constexpr int j = I - LD_SLOTS;
        constexpr int lo = j * XF_PER, hi = (lo + XF_PER < kXf) ? lo + XF_PER : kXf;
        if constexpr (hi > lo) xf_ops<XP, lo>(std::make_integer_sequence<int, hi - lo>{}, d_xf, w, t, u, v_nxt);
    }
    __builtin_amdgcn_sched_barrier(0);
}
template <int TM, int XP, int LP, int... Is>
__device__ __forceinline__ void block(std::integer_sequence<int, Is...>, f32x4 (&acc)[TM][12], const f32x4 (&a_cur)[TM][3],
                                      const f32x2 (&v_cur)[6], f32x2 (&v_nxt)[6], const float (&d_xf)[6][6],
                                      float (&d_ld)[6][6], lds_cptr q_ld, f32x4 (&a_ld)[TM][3],
                                      __amdgpu_buffer_rsrc_t u_rsrc, int u_off, int u_lane, int u_stride) {
    f32x2 w[2], t[6], u[8];
    (slot<TM, XP, LP, Is>(acc, a_cur, v_cur, v_nxt, d_xf, d_ld, w, t, u, q_ld, a_ld, u_rsrc, u_off, u_lane, u_stride), ...);
}

// output transform of one transform row: w[q] = sum_j' A^T[q][j'] M[j'],  A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0;
// 0 1 1 4 4 0; 0 1 -1 8 -8 1]
__device__ __forceinline__ void out_row(const f32x4 *M, f32x4 (&w)[4]) {
    const f32x4 s12 = M[1] + M[2], d12 = M[1] - M[2], s34 = M[3] + M[4], d34 = M[3] - M[4];
    w[0] = M[0] + s12 + s34;
    w[1] = d12 + 2.0f * d34;
    w[2] = s12 + 4.0f * s34;
    w[3] = d12 + 8.0f * d34 + M[5];
}

template <int CIN, int TM>
__device__ __forceinline__ void preload_u(const f32x4 *__restrict__ up, int tile0, int lane, f32x4 (&a)[4][TM][3]) {
    constexpr int kSteps = CIN / 4, kG = 3 * 1024, kUStride = 3 * kSteps * kG;
    const __amdgpu_buffer_rsrc_t u_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<f32x4 *>(up), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                a[g][m][j] = load_u(u_rsrc, lane * 16, tile0 * kUStride + m * kUStride + g * kG + j * 1024);
}

// One pass (transform rows Pass<P>::a, b) over the kSteps channel groups.  Linear group index g = P*kSteps + s;
// block g multiplies group g, transforms g+1, reads the patch of g+2 and loads the U fragments of g+2 -- at the end
// of the pass those belong to pass P+1 (after the last pass the indices are clamped: fetched again, unused).
template <int PL, int CIN, int TM, int P>
__device__ __forceinline__ void pass(lds_cptr base, __amdgpu_buffer_rsrc_t u_rsrc, int ubase, int u_lane,
                                     f32x4 (&a)[4][TM][3], float (&d)[2][6][6], f32x2 (&vb)[2][6], f32x4 (&Y)[TM][16]) {
    constexpr int kSteps = CIN / 4, kGroups = 3 * kSteps, kG = 3 * 1024, kUStride = 3 * kSteps * kG;
    constexpr int NP = P < 2 ? P + 1 : 2;
    constexpr auto seq = std::make_integer_sequence<int, 12 * TM>{};
    f32x4 acc[TM][12];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int k = 0; k < 12; ++k) acc[m][k] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the compiler cannot see that the assembly below is an MFMA reading these registers as its C operand:
    // ALL clearing writes are forced to precede this point (every accumulator is an operand) and the required
    // distance to the first MFMA is kept by hand
    if constexpr (TM == 1) asm volatile("s_nop 15" : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[0][4]), "+a"(acc[0][5]), "+a"(acc[0][6]), "+a"(acc[0][7]), "+a"(acc[0][8]), "+a"(acc[0][9]), "+a"(acc[0][10]), "+a"(acc[0][11]));
    else asm volatile("s_nop 15" : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[0][4]), "+a"(acc[0][5]), "+a"(acc[0][6]), "+a"(acc[0][7]), "+a"(acc[0][8]), "+a"(acc[0][9]), "+a"(acc[0][10]), "+a"(acc[0][11]), "+a"(acc[1][0]), "+a"(acc[1][1]), "+a"(acc[1][2]), "+a"(acc[1][3]), "+a"(acc[1][4]), "+a"(acc[1][5]), "+a"(acc[1][6]), "+a"(acc[1][7]), "+a"(acc[1][8]), "+a"(acc[1][9]), "+a"(acc[1][10]), "+a"(acc[1][11]));
    auto patch = [&](int g2) {  // LDS base of group g2's patch, opaque so that the reads use immediate offsets
        lds_cptr q = base + (4 * (g2 % kSteps)) * PL;
        asm volatile("" : "+v"(q));
        return q;
    };
#pragma unroll 1
    for (int s0 = 0; s0 < kSteps - 4; s0 += 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int g2 = P * kSteps + s0 + r + 2;
            block<TM, P, P>(seq, acc, a[r], vb[r & 1], vb[(r + 1) & 1], d[(r + 1) & 1], d[r & 1], patch(g2), a[(r + 2) & 3],
                            u_rsrc, ubase + g2 * kG, u_lane, kUStride);
        }
    }
    {   // last four groups of the pass
        constexpr int g0 = P * kSteps + kSteps - 4;
        constexpr int c2 = (g0 + 4 < kGroups) ? g0 + 4 : kGroups - 1, c3 = (g0 + 5 < kGroups) ? g0 + 5 : kGroups - 1;
        block<TM, P, P>(seq, acc, a[0], vb[0], vb[1], d[1], d[0], patch(g0 + 2), a[2], u_rsrc, ubase + (g0 + 2) * kG, u_lane,
                        kUStride);
        block<TM, P, P>(seq, acc, a[1], vb[1], vb[0], d[0], d[1], patch(g0 + 3), a[3], u_rsrc, ubase + (g0 + 3) * kG, u_lane,
                        kUStride);
        block<TM, P, NP>(seq, acc, a[2], vb[0], vb[1], d[1], d[0], patch(c2), a[0], u_rsrc, ubase + c2 * kG, u_lane, kUStride);
        block<TM, NP, NP>(seq, acc, a[3], vb[1], vb[0], d[0], d[1], patch(c3), a[1], u_rsrc, ubase + c3 * kG, u_lane, kUStride);
    }
    // the last MFMAs (8 passes = 32 cycles) must have left the matrix pipe before the fold reads them
    // (every accumulator is an operand, so no read of one can be moved above the wait)
    if constexpr (TM == 1) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[0][4]), "+a"(acc[0][5]), "+a"(acc[0][6]), "+a"(acc[0][7]), "+a"(acc[0][8]), "+a"(acc[0][9]), "+a"(acc[0][10]), "+a"(acc[0][11]));
    else asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[0][4]), "+a"(acc[0][5]), "+a"(acc[0][6]), "+a"(acc[0][7]), "+a"(acc[0][8]), "+a"(acc[0][9]), "+a"(acc[0][10]), "+a"(acc[0][11]), "+a"(acc[1][0]), "+a"(acc[1][1]), "+a"(acc[1][2]), "+a"(acc[1][3]), "+a"(acc[1][4]), "+a"(acc[1][5]), "+a"(acc[1][6]), "+a"(acc[1][7]), "+a"(acc[1][8]), "+a"(acc[1][9]), "+a"(acc[1][10]), "+a"(acc[1][11]));
    // fold: Y[p][q] (+)= A^T[p][i'] w_i'[q]; columns of A^T: i'=1: 1 1 1 1, 2: 1 -1 1 -1, 3: 1 2 4 8,
    // 4: 1 -2 4 -8, 0: 1 0 0 0, 5: 0 0 0 1.  Pass 0 writes Y for the first time.
#pragma unroll
    for (int m = 0; m < TM; ++m) {
        f32x4 wa[4], wb[4];
        out_row(&acc[m][0], wa);
        out_row(&acc[m][6], wb);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if constexpr (P == 0) {
                const f32x4 sum = wa[q] + wb[q], dif = wa[q] - wb[q];
                Y[m][0 + q] = sum;
                Y[m][4 + q] = dif;
                Y[m][8 + q] = sum;
                Y[m][12 + q] = dif;
            } else if constexpr (P == 1) {
                const f32x4 sum = wa[q] + wb[q], dif = wa[q] - wb[q];
                Y[m][0 + q] += sum;
                Y[m][4 + q] += 2.0f * dif;
                Y[m][8 + q] += 4.0f * sum;
                Y[m][12 + q] += 8.0f * dif;
            } else {
                Y[m][0 + q] += wa[q];
                Y[m][12 + q] += wb[q];
            }
        }
    }
}

// Y[m][p*4 + q] = conv output (no bias) of output-channel tile tile0 + m at board row 4*ty + p, column 4*tx + q
// (ty = (lane >> 2) & 3, tx = lane & 3) for the lane's 4 channels.
template <int PL, int CIN, int TM>
__device__ __forceinline__ void conv(const float *__restrict__ in, const f32x4 *__restrict__ up, int tile0, int lane,
                                     f32x4 (&a)[4][TM][3], f32x4 (&Y)[TM][16]) {
    constexpr int kSteps = CIN / 4, kG = 3 * 1024, kUStride = 3 * kSteps * kG;
    const int kq = lane >> 4, ty = (lane >> 2) & 3, tx = lane & 3;
    // top-left of the lane's 6x6 patch in halo coordinates: row 4*ty, column 4*tx, plane kq of the group
    const lds_cptr base = (lds_cptr)(in + kq * PL + (4 * ty) * kRowW + 4 * tx);
    const __amdgpu_buffer_rsrc_t u_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<f32x4 *>(up), 0, 0x7fffffff, 0x00020000);
    const int ubase = tile0 * kUStride, u_lane = lane * 16;
    float d[2][6][6];  // patch rows, double buffered: a group's rows are read one block before they are transformed
    f32x2 vb[2][6];    // transformed fragment, double buffered: vb[.][j'] = (row a, row b) of column j'
    {   // pipeline prologue: group 0 transformed, patch rows of group 1 in flight
        f32x2 w[2], t[6], u[8];
        ld_ops<0, 0>(std::make_integer_sequence<int, Pass<0>::n_ld>{}, d[0], base);
        ld_ops<0, 0>(std::make_integer_sequence<int, Pass<0>::n_ld>{}, d[1], base + 4 * PL);
        xf_ops<0, 0>(std::make_integer_sequence<int, Pass<0>::n_xf>{}, d[0], w, t, u, vb[0]);
        __builtin_amdgcn_sched_barrier(0);
    }
    pass<PL, CIN, TM, 0>(base, u_rsrc, ubase, u_lane, a, d, vb, Y);
    pass<PL, CIN, TM, 1>(base, u_rsrc, ubase, u_lane, a, d, vb, Y);
    pass<PL, CIN, TM, 2>(base, u_rsrc, ubase, u_lane, a, d, vb, Y);
}

// Sum `vals` over the 4 lanes {n, n+16, n+32, n+48}: every lane ends with 24 of the 96 sums,
// out[i] = sum of vals[(q & 1) * 48 + (q >> 1) * 24 + i], q = lane >> 4.
__device__ __forceinline__ void reduce_scatter_96(const float (&vals)[96], float (&out)[24]) {
    float r1[48];
#pragma unroll
    for (int i = 0; i < 48; ++i) {
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(vals[i]), __float_as_uint(vals[48 + i]),
                                                         false, false);
        r1[i] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
#pragma unroll
    for (int i = 0; i < 24; ++i) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(r1[i]), __float_as_uint(r1[24 + i]),
                                                         false, false);
        out[i] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
}

}  // namespace f4

// W waves per workgroup: wave w owns the 8 / W output-channel tiles {w*8/W ..} of conv3 for the whole board;
// conv2 (4 tiles) runs on waves 0..3.
//   W = 4 (default): two tiles per wave, one wave per SIMD with up to 512 registers.  A wave does not overlap
//     its own MFMAs with its own vector work (measured: time = MFMA + VALU), so the transform is exposed --
//     but it is done once per SIMD.
//   W = 8: one tile per wave, two waves per SIMD (<= 256 registers each, spills) that do overlap -- but every
//     wave forms the transformed input of the whole board itself, and the doubled vector work makes it 40 %
//     slower than W = 4 (kept selectable as RZ_NET_WINOGRAD_F4_8W).
// Persistent workgroups, LDS layout, conv1, observation prefetch and the feature epilogue as in k_trunk_wino.
template <int W>
__global__ __launch_bounds__(64 * W) void k_trunk_wino_f4(NetDev nd, const float *__restrict__ obs,
                                                          float *__restrict__ feat, int n_boards) {
    constexpr int PL = kPlaneWino;
    constexpr int kLdsFloats = kPlanes * PL;
    constexpr int kThreads = 64 * W;
    constexpr int TM3 = 8 / W;
    constexpr int kPartialFloats = 4 * 6 * 256;
    __shared__ __attribute__((aligned(16))) float lds[kLdsFloats + kPartialFloats];
    float *in0 = lds;
    float *c1 = in0 + kPlanesIn * PL;
    float *c2 = c1 + kPlanesC1 * PL;
    float *partial = c2 + kPlanesC2 * PL;  // [wave][o][y][x]
    const int tid0 = threadIdx.x;
    const int BH = nd.BH, BW = nd.BW, S = nd.S;
    {
        f32x4 *z = reinterpret_cast<f32x4 *>(lds);
        for (int i = tid0; i < kLdsFloats / 4; i += kThreads) z[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    constexpr int kObsPer = (4 * RZ_MAX_BOARD_SIZE * RZ_MAX_BOARD_SIZE + kThreads - 1) / kThreads;
    float ob[kObsPer];
    auto load_obs = [&](int board, int tid) {
        const float *src = obs + (size_t)board * 4 * S;
#pragma unroll
        for (int k = 0; k < kObsPer; ++k) {
            const int i = tid + k * kThreads;
            ob[k] = i < 4 * S ? src[i] : 0.0f;
        }
    };
    // element tid + k*kThreads of a board's observation planes / head features -> where it lives in LDS /
    // in the feature row: the same for every board, so the integer divisions are done once per thread
    int obs_off[kObsPer];
#pragma unroll
    for (int k = 0; k < kObsPer; ++k) {
        const int i = tid0 + k * kThreads;
        const int c = i / S, r = i - c * S, y = r / BW, x = r - y * BW;
        obs_off[k] = i < 4 * S ? c * PL + (y + 1) * kRowW + (x + 1) : -1;
    }
    constexpr int kFeatPer = (6 * RZ_MAX_BOARD_SIZE * RZ_MAX_BOARD_SIZE + kThreads - 1) / kThreads;
    int feat_src[kFeatPer], feat_dst[kFeatPer];
    float feat_bias[kFeatPer];
#pragma unroll
    for (int k = 0; k < kFeatPer; ++k) {
        const int i = tid0 + k * kThreads;
        const int o = i / S, r = i - o * S, y = r / BW, x = r - y * BW;
        feat_src[k] = (o * 16 + y) * 16 + x;
        feat_dst[k] = i < 6 * S ? (i < 4 * S ? i : i - 4 * S + nd.feat_val_off) : -1;
        feat_bias[k] = i < 6 * S ? nd.bh[o] : 0.0f;
    }
    auto store_obs = [&](int) {
#pragma unroll
        for (int k = 0; k < kObsPer; ++k)
            if (obs_off[k] >= 0) in0[obs_off[k]] = ob[k];
    };
    __syncthreads();
    if ((int)blockIdx.x < n_boards) {
        load_obs(blockIdx.x, tid0);
        store_obs(tid0);
    }
    __syncthreads();
    for (int board = blockIdx.x; board < n_boards; board += gridDim.x) {
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int next_board = board + (int)gridDim.x;
    {   // conv1: 4 -> 32 direct: output tile (wave & 1), board rows (32 / W) * (wave >> 1) ..
        constexpr int kRowsPer = 32 / W;
        const int tile = wave & 1, row0 = kRowsPer * (wave >> 1);
        if (row0 < BH) {
            f32x4 acc[1][8];
            zero_acc<1>(acc);
            if (kRowsPer == 4) conv_accumulate<PL, 4, 1, 4>(in0, nd.w1, tile, row0, lane, acc);
            else conv_rows<PL, 4, 1>(in0, nd.w1, tile, row0, (BH - row0 == 7) ? 7 : 8, lane, acc);
            store_relu<PL, 1>(c1, nd.b1, tile, row0, lane, BH, BW, acc, kRowsPer);
        }
    }
    __syncthreads();
    if (next_board < n_boards) load_obs(next_board, tid);
    const int q = lane >> 4, ty = (lane >> 2) & 3, tx = lane & 3;
    if (wave < 4) {  // conv2: 32 -> 64, one output-channel tile per wave (waves 0..3)
        f32x4 a2[4][1][3];
        f32x4 Y[1][16];
        f4::preload_u<32, 1>(nd.u2f, wave, lane, a2);
        f4::conv<PL, 32, 1>(c1, nd.u2f, wave, lane, a2, Y);
        const int c0 = wave * 16 + 4 * q;
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(nd.b2 + c0);
#pragma unroll
        for (int pq = 0; pq < 16; ++pq) {
            const int y = 4 * ty + (pq >> 2), x = 4 * tx + (pq & 3);
            if (y < BH && x < BW) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    c2[(c0 + j) * PL + (y + 1) * kRowW + (x + 1)] = fmaxf(Y[0][pq][j] + bv[j], 0.0f);
            }
        }
    }
    if (next_board < n_boards) store_obs(tid);
    __syncthreads();
    {   // conv3: 64 -> 128, TM3 tiles per wave; the ReLU'd output feeds the two 1x1 head convolutions
        // head partial sums, packed in pairs of outputs (o, o+1): index pos*3 + o/2, pos = p*4 + q
        f32x2 vals2[48];
#pragma unroll
        for (int i = 0; i < 48; ++i) vals2[i] = f32x2{0.0f, 0.0f};
        {
            f32x4 Y[TM3][16];
            f32x4 a3[4][TM3][3];
            f4::preload_u<64, TM3>(nd.u3f, TM3 * wave, lane, a3);
            f4::conv<PL, 64, TM3>(c2, nd.u3f, TM3 * wave, lane, a3, Y);
#pragma unroll
            for (int m = 0; m < TM3; ++m) {
                const int c0 = (TM3 * wave + m) * 16 + 4 * q;
                const f32x4 bv = *reinterpret_cast<const f32x4 *>(nd.b3 + c0);
                f32x4 wv[6];
#pragma unroll
                for (int o = 0; o < 6; ++o) wv[o] = *reinterpret_cast<const f32x4 *>(nd.wh + o * 128 + c0);
#pragma unroll
                for (int pq = 0; pq < 16; ++pq)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float hv = fmaxf(Y[m][pq][j] + bv[j], 0.0f);
#pragma unroll
                        for (int o2 = 0; o2 < 3; ++o2)  // one v_pk_fma_f32 per pair of head outputs
                            vals2[pq * 3 + o2] = __builtin_elementwise_fma(f32x2{wv[2 * o2][j], wv[2 * o2 + 1][j]},
                                                                           f32x2{hv, hv}, vals2[pq * 3 + o2]);
                    }
            }
        }
        float vals[96];
#pragma unroll
        for (int i = 0; i < 96; ++i) vals[i] = vals2[i >> 1][i & 1];
        float sums[24];
        f4::reduce_scatter_96(vals, sums);
        // partial sums of the waves: [wave & 3][o][y][x]; with 8 waves, wave w + 4 stores first and wave w adds
        // its own on top (fixed order: the result does not depend on timing)
        const int off = (q & 1) * 48 + (q >> 1) * 24;
        if (W == 4 || wave >= 4) {
#pragma unroll
            for (int i = 0; i < 24; ++i) {
                const int vi = off + i, pq = vi / 6, o = vi - 6 * pq;
                const int y = 4 * ty + (pq >> 2), x = 4 * tx + (pq & 3);
                partial[(((wave & 3) * 6 + o) * 16 + y) * 16 + x] = sums[i];
            }
        }
        if (W == 8) {
            __syncthreads();
            if (wave < 4) {
#pragma unroll
                for (int i = 0; i < 24; ++i) {
                    const int vi = off + i, pq = vi / 6, o = vi - 6 * pq;
                    const int y = 4 * ty + (pq >> 2), x = 4 * tx + (pq & 3);
                    partial[((wave * 6 + o) * 16 + y) * 16 + x] += sums[i];
                }
            }
        }
    }
    __syncthreads();
    {
        float *dst = feat + (size_t)board * nd.feat_ld;
#pragma unroll
        for (int k = 0; k < kFeatPer; ++k) {
            if (feat_dst[k] < 0) continue;
            float v = feat_bias[k];
#pragma unroll
            for (int g = 0; g < 4; ++g) v += partial[g * 6 * 256 + feat_src[k]];
            dst[feat_dst[k]] = fmaxf(v, 0.0f);
        }
    }
    }  // boards
}

// ------------------------------------------------------------------ split-operand direct convolution
// conv1 .. conv3 as DIRECT 3x3 convolutions on the f16 matrix pipe (v_mfma_f32_32x32x16_f16: 16x the rate of
// the f32-input MFMA, which runs at the vector rate and does not overlap with vector work at all -- DESIGN.md
// section 5), with every f32 operand carried as an unevaluated sum of two f16 values:
//     x * s = hi + lo,  hi = f16(x * s),  lo = f16(x * s - hi)           (s: a power of two, see below)
// and every product formed as hi*hi + hi*lo + lo*hi on three MFMAs that accumulate in f32.  hi + lo holds 22
// significant bits of x and the dropped lo*lo term is below 2^-22 of the product, so the result is within a
// few 1e-7 (relative) of the f32 kernels -- the level of their own accumulation rounding (measured in
// tests/test_gpu_parity.py; the tolerance of the path is 1e-4).  Scales keep the lo halves out of f16's
// subnormal range: the activations of a layer are stored times a power of two <= 16 that rz_net_load derives from a
// bound on that layer's activations (bias + positive weights x input bounds, observation planes in [0, 1]), so a
// network of ANY weight scale stays inside the f16 range on the 0 / 1 planes of the MCTS leaves -- no fallback is
// needed for trained weights (inputs beyond [0, 1] through rz_net_trunk / rz_net_forward can still overflow: that
// raises RZ_NET_FLAG_F16_RANGE); the weights of a layer are stored times the power of two that brings their
// largest magnitude into [2^13, 2^14); the accumulator is rescaled (exactly) in the epilogue.
//   * LDS: conv1's and conv2's outputs as [piece][18 rows][18 cols][channels + 8] f16, channels innermost, so
//     the B fragment of a lane (8 consecutive input channels of one position) is ONE ds_read_b128; position
//     strides of 80 / 144 bytes spread the 8 lanes of an LDS cycle over all 64 banks.  147.4 KB + 3.5 KB of
//     head weights and conv3 biases, staged once per persistent workgroup.
//   * MFMA tile: M = 32 output channels, N = 32 positions = two board rows x 16 columns, K = 16 input channels
//     of one tap.  Wave w owns board rows 4w .. 4w+3 (2 N-tiles) and ALL M-tiles of a layer (conv2: 2, conv3:
//     4), so one K-step is 6*TM MFMAs on 2*TM weight fragments (buffer loads from L2, packed on the host in
//     fragment order, two steps ahead) and 4 activation fragments (one step ahead); one load is pinned behind
//     each of the first MFMAs of the step.  The chip is at its power limit in these loops (DESIGN.md section
//     5): what counts is the amount of work, not where it is placed.
//   * conv1 (4 -> 32): the observation planes live in LDS as [piece][position][4 planes] f16, K-step = one kernel
//     row (4 columns x 4 planes, the 4th column meeting zero weights), its 6 weight fragments and biases stay in
//     registers across boards; conv3's output feeds the two 1x1 head convolutions from registers in f32.
namespace sp {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(3))) f16x8 *lds_frag;

constexpr float kObsScale = 16.0f;  // observation planes (0 / 1) are stored times 16
constexpr float kMaxActScale = 16.0f, kF16Room = 60000.0f;  // activation scales: powers of two <= 16 that keep bound * scale < 60000
constexpr int kGridPos = 18 * 18;
// POS: positions of the halo grid (18 x 18 in general; the compact layout of small boards: see k_trunk_split's RW / RH)
template <int CIN, int POS = kGridPos> struct Geo {
    static constexpr int pos_bytes = (CIN + 8) * 2;          // 80 / 144
    static constexpr int piece_bytes = POS * pos_bytes;      // 25 920 / 46 656 on the 18 x 18 grid
    static constexpr int chunks = CIN / 16, steps = 9 * chunks;
};
constexpr int kC1Bytes = 2 * Geo<32>::piece_bytes, kC2Bytes = 2 * Geo<64>::piece_bytes;
// observation planes: [hi | lo][18 rows][20 cols][4 planes] f16 -- the 4 planes of a position are 8 contiguous bytes,
// so the 16 K-values of conv1's step "kernel row ky" (4 columns x 4 planes, the 4th column meeting zero weights)
// are two 16-byte runs
constexpr int kInCols = 20, kInPieceBytes = 18 * kInCols * 8, kInBytes = 2 * kInPieceBytes;
constexpr int kHeadFloats = 128 * 7 + 8;  // head weights [128][6] + conv3 biases [128] + head biases [6] (+ 2 pad)
constexpr int kLdsBytes = kInBytes + kC1Bytes + kC2Bytes + kHeadFloats * 4;
static_assert(kLdsBytes <= 160 * 1024, "LDS budget");
static_assert(kInBytes % 16 == 0, "piece alignment");

__device__ __forceinline__ f16x8 load_w(__amdgpu_buffer_rsrc_t rsrc, int lane_off, int uniform_off) {
    return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane_off, uniform_off, 0));
}

// x (already scaled) -> hi, lo in 8 vector instructions per 4 values: two packed conversions for the hi pieces
// (v_cvt_pk_f16_f32, round to nearest even like the scalar conversion), the residuals z - hi as v_fma_mix_f32 with the f16
// operand widened inside the instruction (the same single rounding as convert + subtract; hipcc folds fma(x, -1, z) back
// into the two instructions, hence the asm), two packed conversions for lo.
__device__ __forceinline__ float resid_lo(float z, unsigned pair) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pair), "v"(z));
    return r;
}
__device__ __forceinline__ float resid_hi(float z, unsigned pair) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(pair), "v"(z));
    return r;
}
__device__ __forceinline__ void split4(const float (&z)[4], f16x4 &hi, f16x4 &lo) {
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const f32x4v zv = {z[0], z[1], z[2], z[3]};
    hi = __builtin_convertvector(zv, f16x4);
    const u32x2 pairs = __builtin_bit_cast(u32x2, hi);
    const f32x4v r = {resid_lo(z[0], pairs[0]), resid_hi(z[1], pairs[0]), resid_lo(z[2], pairs[1]), resid_hi(z[3], pairs[1])};
    lo = __builtin_convertvector(r, f16x4);
}

// slot I of K-step S: MFMA I of the step plus (behind the first MFMAs) one load of a coming step
// (D = depth of the ring of weight fragments: a step's fragments are requested D - 1 steps ahead -- 2 where a step has
// 6 or more MFMAs to cover the L2 round trip, 4 for the small tiles of the channel-split variants)
constexpr int ring_depth(int tm, int tn) { return tm * tn >= 3 ? 3 : 5; }
// (and of the ring of activation fragments: read from LDS one step ahead)
constexpr int act_depth(int, int) { return 2; }   // (3 for the small tiles was tried: no gain, their steps are bound by the accumulator chain)
template <int CIN, int TM, int TN, int RPT, int RW, int S, int I>
__device__ __forceinline__ void slot(f32x16 (&acc)[TM][TN], f16x8 (&a)[ring_depth(TM, TN)][TM][2], f16x8 (&b)[act_depth(TM, TN)][TN][2], lds_frag q0,
                                     lds_frag q1, __amdgpu_buffer_rsrc_t w_rsrc, int w_base, int w_lane) {
    using G = Geo<CIN>;
    constexpr int D = ring_depth(TM, TN);
    constexpr int combo = I / (TM * TN), m = (I / TN) % TM, n = I % TN;
    constexpr int pa = combo == 2 ? 1 : 0, pb = combo == 1 ? 1 : 0;
    constexpr int DB = act_depth(TM, TN);
    if constexpr (S == 0 && combo == 0) {   // the first MFMA of a tile starts from the constant 0: no zeroing of 16 registers per tile
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[S % D][m][pa], b[S % DB][n][pb], zero, 0, 0, 0);
    } else {
        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[S % D][m][pa], b[S % DB][n][pb], acc[m][n], 0, 0, 0);
    }
    if constexpr (I < 2 * TN) {
        if constexpr (S + DB - 1 < G::steps) {
            constexpr int s1 = S + DB - 1, tap = s1 / G::chunks, c = s1 % G::chunks, nn = I / 2, piece = I % 2;
            constexpr int off = ((RPT * nn + tap / 3) * RW + tap % 3) * G::pos_bytes + c * 32;
            static_assert(off % 16 == 0 && off < 65536, "ds_read_b128 immediate");
            b[s1 % DB][nn][piece] = (piece ? q1 : q0)[off / 16];
        }
    } else if constexpr (I < 2 * TN + 2 * TM) {
        if constexpr (S + D - 1 < G::steps) {
            constexpr int s2 = S + D - 1, j = I - 2 * TN, mm = j / 2, piece = j % 2;
            a[s2 % D][mm][piece] = load_w(w_rsrc, w_lane, w_base + ((mm * G::steps + s2) * 2 + piece) * 1024);
            // one M-tile x one N-tile: three MFMAs per step but four fragments to fetch -- the last slot takes two
            if constexpr (TM == 1 && TN == 1 && I == 2)
                a[s2 % D][0][1] = load_w(w_rsrc, w_lane, w_base + (s2 * 2 + 1) * 1024);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

template <int CIN, int TM, int TN, int RPT, int RW, int S, int... Is>
__device__ __forceinline__ void step(std::integer_sequence<int, Is...>, f32x16 (&acc)[TM][TN], f16x8 (&a)[ring_depth(TM, TN)][TM][2],
                                     f16x8 (&b)[act_depth(TM, TN)][TN][2], lds_frag q0, lds_frag q1, __amdgpu_buffer_rsrc_t w_rsrc,
                                     int w_base, int w_lane) {
    (slot<CIN, TM, TN, RPT, RW, S, Is>(acc, a, b, q0, q1, w_rsrc, w_base, w_lane), ...);
}

template <int CIN, int TM, int TN, int RPT, int RW, int... Ss>
__device__ __forceinline__ void steps(std::integer_sequence<int, Ss...>, f32x16 (&acc)[TM][TN], f16x8 (&a)[ring_depth(TM, TN)][TM][2],
                                      f16x8 (&b)[act_depth(TM, TN)][TN][2], lds_frag q0, lds_frag q1, __amdgpu_buffer_rsrc_t w_rsrc,
                                      int w_base, int w_lane) {
    (step<CIN, TM, TN, RPT, RW, Ss>(std::make_integer_sequence<int, 3 * TM * TN>{}, acc, a, b, q0, q1, w_rsrc, w_base, w_lane), ...);
}

// The weight fragments of the first K-steps (M-tiles 0 .. TM-1): no dependence on LDS, so a layer's first
// fragments are requested while the previous layer is still being reduced.
template <int CIN, int TM, int TN>
__device__ __forceinline__ void preload_w(f16x8 (&a)[ring_depth(TM, TN)][TM][2], const void *wts, int lane) {
    using G = Geo<CIN>;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(wts), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int s = 0; s < ring_depth(TM, TN) - 1; ++s)
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int p = 0; p < 2; ++p) a[s][m][p] = load_w(w_rsrc, lane * 16, ((m * G::steps + s) * 2 + p) * 1024);
}

// acc[m][n] = sum over taps and input channels for M-tiles 0 .. TM-1 (all output channels of the layer) and
// N-tiles nt0 .. nt0 + TN - 1 (`in` = piece 0 of the layer's input in LDS, `a` primed by preload_w).
// The lane's MFMA column is the position (row0 + ry, x) of the wave's first N-tile; a further tile of the wave (TN = 2
// only) lies two rows below (RPT rows in general: the 3 + 1 variant runs three 3-row tiles in one wave).
template <int CIN, int TM, int TN, int RPT = 2, int RW = kRowW, int POS = kGridPos>
__device__ __forceinline__ void conv(const char *in, const void *wts, int row0, int ry, int x, int lane,
                                     f16x8 (&a)[ring_depth(TM, TN)][TM][2], f32x16 (&acc)[TM][TN]) {
    using G = Geo<CIN, POS>;
    const int h = lane >> 5;
    // halo position (row0 + ry, x) = the top-left tap of output (row0 + ry, x)
    const int lane_byte = ((row0 + ry) * RW + x) * G::pos_bytes + h * 16;
    const lds_frag q0 = (lds_frag)(in + lane_byte), q1 = (lds_frag)(in + lane_byte + G::piece_bytes);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(wts), 0, 0x7fffffff, 0x00020000);
    f16x8 b[act_depth(TM, TN)][TN][2];
#pragma unroll
    for (int s0 = 0; s0 < act_depth(TM, TN) - 1; ++s0) {   // the fragments of the first step(s): tap = s0 / chunks, chunk = s0 % chunks
        const int tap = s0 / G::chunks, c = s0 % G::chunks;
#pragma unroll
        for (int nn = 0; nn < TN; ++nn) {
            const int off = ((RPT * nn + tap / 3) * RW + tap % 3) * G::pos_bytes + c * 32;
            b[s0][nn][0] = q0[off / 16];
            b[s0][nn][1] = q1[off / 16];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    steps<CIN, TM, TN, RPT, RW>(std::make_integer_sequence<int, G::steps>{}, acc, a, b, q0, q1, w_rsrc, 0, lane * 16);
}

}  // namespace sp

// Wave w owns board rows 4w .. 4w+3 (N-tiles 2w, 2w+1) and ALL output channels of conv2 and of conv3, so the
// 1x1 head convolutions see every channel of a position in one wave (two lane halves, one shuffle) and the head
// features go from registers to memory: two barriers per board.  TN = N-tiles per wave: 2 (tiles of 2 rows x 16
// columns) in general; when four tiles of 32 / width rows x width columns cover the board (9x9: 3 x 9, 10x10: 3 x 10,
// 8x8: 4 x 8, Connect4: 4 x 7, 6x6: 5 x 6) a wave owns ONE such tile (TN = 1) and issues half the MFMAs or fewer; a
// wave whose rows lie below the board skips its MFMA loops.
// The leaf positions themselves (rz_net_trunk_leaves): bitboards [board][2 colours][4 words], side to move and last
// cell, exactly what the tree kernels keep per leaf.  The trunk then builds the four observation planes of
// GomokuEnv.current_state (gomoku_env.py:95-114) itself -- thread t = cell t: stones of the side to move, of the other
// side, the last move (if any stone is on the board), ones if the stone count is even -- so the tree kernel need not
// write, and this kernel need not read, 16 S bytes of 0.0 / 1.0 floats per leaf.
struct LeafBits {
    const uint64_t *stones;
    const int32_t *to_move;
    const int32_t *last;
};

// Development aid (not built by default): -DRZ_NET_PROFILE accumulates the shader-clock cycles wave 0 of workgroup 0 spends in
// each phase of a board in k_trunk_split into net_prof[] (rz_net_debug_profile).
#ifdef RZ_NET_PROFILE
__device__ long long net_prof[24];   // [16 .. 19]: the resident search's tree phases (value head, expand / backup, selection, planes)
#define NET_TICK(i) do { __builtin_amdgcn_sched_barrier(0); const long long now_ = __builtin_readcyclecounter(); prof_acc[i] += now_ - prof_t; prof_t = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define NET_TICK(i)
#endif

// MS (with TN = 1): when the board needs only 2 (MS = 2) or 1 (MS = 4) of the four N-tiles, the waves that would idle
// take a share of the OUTPUT CHANNELS instead: wave = part * (4 / MS) + tile, part p computes M-tiles p * TM / MS .. of
// conv2 and conv3 for its tile (a 6x7 Connect4 board: 6 instead of 12 MFMAs per K-step and wave; a 3x3 board: 3).  The
// 1x1 head convolutions then sum over the channels of MS waves: partial sums meet in LDS (in the 16 padding bytes of
// conv1's positions, which nothing else touches), part 0 adds them in part order and stores the features.
// the value of lane l ^ 32 (h = l >> 5): v_permlane32_swap, two vector instructions instead of a trip through the LDS crossbar
__device__ __forceinline__ float other_half(float x, int h) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(h ? r[0] : r[1]);
}

// RESIDENT SEARCH (RES instantiations of the trunk kernels; rz_net_search_resident).  For a batch of at most one game per CU the whole
// chain of a search lives in ONE workgroup per game and ONE launch: trunk -> value head -> expand / backup -> next selection, n_sims
// times, the leaf handed from the tree code to the trunk through LDS, the value head's inputs never leaving the CU, the trunk's
// prologue (weights into registers, LDS zeroing) paid once per launch instead of once per simulation, no kernel boundary inside a
// search.  The tree code is the engine's own (rz_tree.h: the bodies of k_tree_step_def), the policy features go to the deferred
// store like in the two-launch step, so trees, priors and values are those of that route bit for bit.
#ifndef RZ_SPLIT_TREE_PRIO
#define RZ_SPLIT_TREE_PRIO 1   // k_trunk_split<RES> on the compact grid (two games per CU): issue priority of the tree phase
#endif
template <bool RES> struct ResArgs {};
template <> struct ResArgs<true> {
    rzt::Dev E;          // the engine's device view (rz_device_view)
    rz_value_head vh;    // valfeat unused: the inputs stay in LDS
    int n_sims;          // simulations of this launch: n_sims x (trunk, expand / backup), a selection between two of them
    int select_first;    // != 0: the launch begins with the selection of the first leaf itself (no rz_select_step before it)
};
__device__ __forceinline__ int res_sims(const ResArgs<false> &) { return 0; }
__device__ __forceinline__ int res_sims(const ResArgs<true> &r) { return r.n_sims; }

// Deferred priors (rz_value_head, include/rlzero_hip.h): where a board's features go when no FC GEMM follows the trunk -- the policy
// pieces into slot slot_of[board] of a store of `slot_halfs` f16 values per slot (tiles of groups_act K-steps), the value head's
// inputs as f32 rows of vf_ld floats.  slot_of == nullptr: the ordinary route.
struct DeferredOut {
    const int32_t *slot_of;
    long long slot_halfs;
    float *valfeat;
    int vf_ld;
    unsigned long long *trace;   // rz_trace.h (NULL: none)
    int n_slots;                 // slots of the store: a leaf whose slot lies beyond it is NOT stored (the tree step flags the game)
};

// FC_HERE (small boards, TN = 1; `raw` / `hid` given): the workgroup also runs the first FC layers of both heads on ITS OWN
// board, behind the feature stage -- the arithmetic of k_heads_split (the same MFMA on the same K quarters, one per wave, the
// quarters summed in wave order, fmaf(sum, scale, bias)): the same bits, one launch and one kernel boundary less in the chain
// trunk -> FC -> tree step of a small batch.  A board is row 0 of the MFMA's 32 (the other rows are zero: rows do not mix), its
// features never leave the CU (f16 pieces in LDS), the weights stream from L2.
// RW x RH: the halo grid of the activations in LDS.  18 x 18 (a board of up to 16 x 16) in general: 154 KB, one workgroup per CU.
// COMPACT grids for small boards (RW = width + 2, RH such that RW x RH >= 128 positions: the channel-split variants park their
// partial head sums in the padding of positions 0 .. 127) cut that to 60-70 KB -- TWO workgroups per CU: a board of 6 x 7 is a
// latency chain of small MFMA loops, and two such chains interleave on a CU where one leaves the pipes idle most of the time.
// Tile rows beyond the board still read positions past its ring (MFMA columns that are stored nowhere): inside the grid or
// in the bytes behind it, always inside the workgroup's LDS.  (FC_HERE needs the big grid's spare rows: 18 x 18 only.)
template <int TN, int MS = 1, bool RES = false, int RW = kRowW, int RH = 18>
__global__ __launch_bounds__(256, (RW == kRowW ? 1 : 2)) void k_trunk_split(NetDev nd, const float *__restrict__ obs, LeafBits leaves,
                                                     float *__restrict__ feat, _Float16 *__restrict__ feat16,
                                                     int n_boards, unsigned *__restrict__ flags,
                                                     float *__restrict__ raw = nullptr, float *__restrict__ hid = nullptr,
                                                     DeferredOut later = DeferredOut{nullptr, 0, nullptr, 0, nullptr},
                                                     ResArgs<RES> res = ResArgs<RES>{}) {
#ifdef RZ_NET_PROFILE
    const long long prof_k0 = __builtin_readcyclecounter();
    long long prof_acc[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, prof_t = prof_k0;
#endif
    constexpr int kThreads = 256;
    // the resident search's tree code (rz_tree.h: W): a board of one N-tile (MS = 4) or two (MS = 2) has at most 64 cells = one word
    // of a bitboard, every board of this kernel (four tiles of 32 positions) at most 128 = two
    constexpr int kResWords = (MS == 4 || MS == 2) ? 1 : 2;
    // RES (the resident search, see ResArgs): the value head's input row (boards of up to 10 rows and columns: 2 S <= 256 with
    // the padding), the K-quarter sums of its first layer, the next leaf
    __shared__ float res_vrow[RES ? 256 : 1];
    __shared__ float res_part[RES ? rzt::kDefWaves : 1][RES ? rzt::kWave : 1];
    __shared__ __attribute__((aligned(16))) uint64_t res_leaf[RES ? 2 * RZ_BOARD_WORDS + 1 : 1];
    int res_slot0 = 0;
    if constexpr (RES) {
        if ((int)blockIdx.x >= n_boards || res.E.active[blockIdx.x] == 0) return;   // (uniform: before any barrier)
        res_slot0 = res.E.pend[blockIdx.x];
        res_vrow[threadIdx.x] = 0.0f;
    }
    constexpr int POS = RW * RH, IC = RW + 2;   // positions of the halo grid; columns of the observation planes' grid
    constexpr int kInPiece = RH * IC * 8, kInB = 2 * kInPiece, kC1B = 2 * sp::Geo<32, POS>::piece_bytes, kC2B = 2 * sp::Geo<64, POS>::piece_bytes;
    static_assert(kInB % 16 == 0 && POS >= 128, "the grid: 16-byte pieces, 128 positions for the channel-split variants' partial sums");
    __shared__ __attribute__((aligned(16))) char lds_raw[kInB + kC1B + kC2B + sp::kHeadFloats * 4];
    char *in0 = lds_raw;                      // observation planes, pieces hi | lo
    char *c1 = lds_raw + kInB;                // conv1 output, pieces hi | lo
    char *c2 = c1 + kC1B;                     // conv2 output, pieces hi | lo
    float *hw = reinterpret_cast<float *>(c2 + kC2B);  // head weights [128][6], then conv3 biases [128]
    const int tid0 = threadIdx.x;
    const int BH = nd.BH, BW = nd.BW, S = nd.S;
    float zmax = 0.0f;  // largest scaled value this thread stored as f16 pieces
    constexpr int kObsPer = (4 * RZ_MAX_BOARD_SIZE * RZ_MAX_BOARD_SIZE + kThreads - 1) / kThreads;
    float ob[kObsPer];
    auto load_obs = [&](int board, int tid) {
        const float *src = obs + (size_t)board * 4 * S;
#pragma unroll
        for (int k = 0; k < kObsPer; ++k) {
            const int i = tid + k * kThreads;
            ob[k] = i < 4 * S ? src[i] : 0.0f;
        }
    };
    const bool from_bits = leaves.stones != nullptr;
    int obs_off[kObsPer];
#pragma unroll
    for (int k = 0; k < kObsPer; ++k) obs_off[k] = -1;
    if (!from_bits) {  // (two integer divisions per element: ~1.5 k cycles of the prologue that the bitboard route does not need)
#pragma unroll
        for (int k = 0; k < kObsPer; ++k) {
            const int i = tid0 + k * kThreads;
            const int c = i / S, r = i - c * S, y = r / BW, x = r - y * BW;
            obs_off[k] = i < 4 * S ? ((y + 1) * IC + (x + 1)) * 8 + c * 2 : -1;
        }
    }
    // bit mode: thread t owns cell t (S <= 256 = threads); its 4 plane values as f16 (x 16: exact, the lo piece is 0)
    const int cell_y = tid0 / BW, cell_x = tid0 - cell_y * BW;
    const int cell_off = tid0 < S ? ((cell_y + 1) * IC + (cell_x + 1)) * 8 : -1;
    sp::f16x4 cell_planes = {(_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
    auto load_bits = [&](int board, int tid) {
        const uint64_t *sb = leaves.stones + (size_t)board * 8;
        const int tm = leaves.to_move[board], lc = leaves.last[board];
        int nst = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) nst += __popcll(sb[q]);  // (uniform address: scalar loads)
        const int word = (tid >> 6) & 3, bit = tid & 63;
        const uint64_t w0 = sb[word], w1 = sb[4 + word];
        const bool s0 = (w0 >> bit) & 1ull, s1 = (w1 >> bit) & 1ull;
        const bool mine = tm == 0 ? s0 : s1, theirs = tm == 0 ? s1 : s0;
        const _Float16 one = (_Float16)sp::kObsScale, zero = (_Float16)0.0f;
        cell_planes[0] = mine ? one : zero;
        cell_planes[1] = theirs ? one : zero;
        cell_planes[2] = (nst > 0 && tid == lc) ? one : zero;
        cell_planes[3] = (nst & 1) ? zero : one;
    };
    auto store_obs = [&](int) {
        if (from_bits) {
            if (cell_off >= 0) {
                *reinterpret_cast<sp::f16x4 *>(in0 + cell_off) = cell_planes;
                *reinterpret_cast<sp::f16x4 *>(in0 + kInPiece + cell_off) =
                    sp::f16x4{(_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < kObsPer; ++k)
            if (obs_off[k] >= 0) {
                const float z = ob[k] * sp::kObsScale;
                const _Float16 hi = (_Float16)z;
                zmax = fmaxf(zmax, fabsf(z));
                *reinterpret_cast<_Float16 *>(in0 + obs_off[k]) = hi;
                *reinterpret_cast<_Float16 *>(in0 + kInPiece + obs_off[k]) = (_Float16)(z - (float)hi);
            }
    };
    // Prologue of a persistent workgroup.  Every global load it needs -- head weights and conv3 biases (3.5 KB, bound for
    // LDS), the rescaling factors and activation scales, conv1's weights and biases (registers), the first board -- is
    // ISSUED first, the parts of the LDS that no board writes are zeroed under their latency, and only then the values are stored: with one board
    // per workgroup (256 boards per launch) the prologue is not amortised, and two lanes alternate such launches.
    constexpr int kHwPer = (128 * 7 + kThreads - 1) / kThreads;
    float hw_reg[kHwPer];
#pragma unroll
    for (int k = 0; k < kHwPer; ++k) {
        const int i = tid0 + k * kThreads;
        hw_reg[k] = i < 768 ? nd.whp[i] : (i < 128 * 7 ? nd.b3[i - 768] : 0.0f);
    }
    // the 6 head biases too: a global load in the epilogue would sit between the feature stores, and its
    // s_waitcnt vmcnt(0) also waits for the stores before it -- six store round trips per board
    const float bh_reg = tid0 < 6 ? nd.bh[tid0] : 0.0f;
    // the rescaling factors and the activation scales of the layers (powers of two chosen by rz_net_load from
    // bounds on the activations) once per workgroup: a load placed behind a layer's MFMA loop is exposed in full
    const float k1 = nd.s_inv[2], k2 = nd.s_inv[0], k3 = nd.s_inv[1];
    const float act1 = nd.s_inv[5], act2 = nd.s_inv[6], act3 = nd.s_inv[7];
    // conv1's weights (3 kernel rows x hi / lo, 6 KB per workgroup) and biases stay in registers for all boards
    sp::f16x8 a1[3][2];
    f32x4 bias1[4];
    {
        const int lane0 = tid0 & 63;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int p_ = 0; p_ < 2; ++p_) a1[ky][p_] = __builtin_bit_cast(sp::f16x8, nd.s1[(ky * 2 + p_) * 64 + lane0]);
#pragma unroll
        for (int g = 0; g < 4; ++g) bias1[g] = *reinterpret_cast<const f32x4 *>(nd.b1 + 8 * g + 4 * (lane0 >> 5)) * act1;
    }
    const bool first = (int)blockIdx.x < n_boards;
    bool sel_first = false;   // RES: the first leaf is selected by this launch (below, behind the zeroing)
    if constexpr (RES) sel_first = res.select_first != 0;
    if (first && !sel_first) {
        if (from_bits) load_bits(blockIdx.x, tid0); else load_obs(blockIdx.x, tid0);
    }
    // the planes of a leaf handed over through LDS by the tree code of this workgroup (select_body's lds_leaf): what load_bits forms
    auto planes_from_lds = [&](int tid) {
        int nst = 0;
#pragma unroll
        for (int q8 = 0; q8 < 8; ++q8) nst += __popcll(res_leaf[q8]);
        const int tm = reinterpret_cast<const int *>(res_leaf + 2 * RZ_BOARD_WORDS)[0], lc = reinterpret_cast<const int *>(res_leaf + 2 * RZ_BOARD_WORDS)[1];
        const int word = (tid >> 6) & 3, bit = tid & 63;
        const uint64_t w0 = res_leaf[word], w1 = res_leaf[4 + word];
        const bool s0 = (w0 >> bit) & 1ull, s1 = (w1 >> bit) & 1ull;
        const bool mine_ = tm == 0 ? s0 : s1, theirs = tm == 0 ? s1 : s0;
        const _Float16 one = (_Float16)sp::kObsScale, zero = (_Float16)0.0f;
        cell_planes[0] = mine_ ? one : zero;
        cell_planes[1] = theirs ? one : zero;
        cell_planes[2] = (nst > 0 && tid == lc) ? one : zero;
        cell_planes[3] = (nst & 1) ? zero : one;
    };
    __builtin_amdgcn_sched_barrier(0);  // the loads above stay above the zeroing
    NET_TICK(11);
    {
        // What a VALID position reads and no board writes must be zero: the observation planes' halo (all of in0: 5.8 KB)
        // and, in c1 / c2, the ring of positions around the board (row 0, row BH + 1, column 0, column BW + 1 of the halo
        // grid).  Positions further out are read only by MFMA columns that are not positions of the board (tile padding:
        // a column's garbage stays in that column and is never stored), and the board's own positions are overwritten by
        // every board: 64 positions x 2 pieces on a 15x15 board instead of 147 KB, numbered densely (an LDS store costs
        // its issue whatever the number of active lanes).
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        f32x4 *z = reinterpret_cast<f32x4 *>(lds_raw);
        for (int i = tid0; i < kInB / 16; i += kThreads) z[i] = zero;
        const int n_ring = 2 * (BW + 2) + 2 * BH;
        for (int it = tid0; it < 2 * n_ring; it += kThreads) {
            const int piece = it >= n_ring, idx = it - piece * n_ring;
            int py, px;
            if (idx < 2 * (BW + 2)) {
                const int bottom = idx >= BW + 2;
                py = bottom ? BH + 1 : 0;
                px = idx - bottom * (BW + 2);
            } else {
                const int j = idx - 2 * (BW + 2);
                py = 1 + (j >> 1);
                px = (j & 1) ? BW + 1 : 0;
            }
            const int pos = py * RW + px;
            f32x4 *q1 = reinterpret_cast<f32x4 *>(c1 + piece * sp::Geo<32, POS>::piece_bytes + pos * sp::Geo<32>::pos_bytes);
#pragma unroll
            for (int i = 0; i < sp::Geo<32>::pos_bytes / 16; ++i) q1[i] = zero;
            f32x4 *q2 = reinterpret_cast<f32x4 *>(c2 + piece * sp::Geo<64, POS>::piece_bytes + pos * sp::Geo<64>::pos_bytes);
#pragma unroll
            for (int i = 0; i < sp::Geo<64>::pos_bytes / 16; ++i) q2[i] = zero;
        }
    }
    NET_TICK(12);
    __syncthreads();
    NET_TICK(13);
    // (the head weights go to LDS behind the first board's conv1: they are first read two barriers later, and their
    // loads need not be waited for here)
    bool hw_pending = true;
    if constexpr (RES) {
        if (sel_first) {   // AlphaZeroMCTS._playout's select loop for the first simulation of the search (rz_select_step's work)
            if ((tid0 >> 6) == 0) rzt::select_body<false, kResWords>(res.E, nullptr, blockIdx.x, tid0 & 63, 0, res_leaf);
            __syncthreads();
            planes_from_lds(tid0);
        }
    }
    if (first) store_obs(tid0);
    NET_TICK(14);
    __syncthreads();
#ifdef RZ_NET_PROFILE
    NET_TICK(15);
    prof_acc[9] = prof_t - prof_k0;   // the prologue
#endif
    for (int board = blockIdx.x, sim = 0; RES ? sim < res_sims(res) : board < n_boards; RES ? (void)++sim : (void)(board += gridDim.x)) {
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int next_board = RES ? n_boards : board + (int)gridDim.x;   // (RES: the next leaf does not exist yet)
    // MS = 3 (three N-tiles of 3 rows: 9x9): waves 0 .. 2 = the tiles with M-tiles 0 .. 2 of conv3 (and all of conv1 / conv2),
    // wave 3 = part 1 = conv3's M-tile 3 for ALL three tiles: 9 MFMAs per K-step in every wave instead of 12 in three
    constexpr int kTiles = MS == 3 ? 3 : 4 / MS;          // waves side by side over the board's rows
    const int tile = MS == 1 ? wave : (MS == 3 ? (wave < 3 ? wave : 0) : wave % kTiles);
    const int part = MS == 1 ? 0 : (MS == 3 ? (wave == 3 ? 1 : 0) : wave / kTiles);
    constexpr int TM2 = (MS == 1 || MS == 3) ? 2 : 1, TM3 = MS == 3 ? 3 : 4 / MS;   // M-tiles of conv2 / conv3 per wave
    const int m2 = (MS == 1 || MS == 3) ? 0 : (part & 1), m3 = MS == 3 ? 0 : part * TM3;   // ... starting at
    const bool conv2_mine = MS == 3 ? part == 0 : (MS < 4 || part < 2);   // (conv2 has two M-tiles: with MS = 4 parts 2, 3 sit it out)
    const char *s2p = reinterpret_cast<const char *>(nd.s2) + (size_t)m2 * sp::Geo<32>::steps * 2 * 1024;
    const char *s3p = reinterpret_cast<const char *>(nd.s3) + (size_t)m3 * sp::Geo<64>::steps * 2 * 1024;
    sp::f16x8 a2[sp::ring_depth(TM2, TN)][TM2][2];
    sp::preload_w<32, TM2, TN>(a2, s2p, lane);
    // the lane's column of an N-tile: position (ry, x) of a tile of RT rows (TN = 2: always 2 x 16); a lane past the
    // tile's positions computes position (0, 0) again and stores nothing
    const int RT = TN == 2 ? 2 : nd.tile_rows, CT = TN == 2 ? 16 : nd.tile_cols;
    const int n = lane & 31, h = lane >> 5;
    const int ry_raw = TN == 2 ? n >> 4 : (n * nd.tile_rcp) >> 16;
    const bool col_ok = ry_raw < RT;
    const int ry = col_ok ? ry_raw : 0, x = col_ok ? n - ry_raw * CT : 0;
    const int row0 = RT * TN * tile;       // first board row of this wave
    // a wave below the board skips its MFMA loops (it still meets the barriers); with TN = 2 its rows stay inside the
    // halo grid and it computes them unconditionally (a branch around the loops costs the accumulators their registers)
    const bool busy = row0 < BH;
    if (busy && part == 0) {   // conv1: 4 -> 32 (one M-tile), N-tiles TN*wave ..; K-step = kernel row ky
        typedef const __attribute__((address_space(3))) sp::f16x4 *lds_half;
        const lds_half q = (lds_half)(in0 + ((row0 + ry) * IC + x + 2 * h) * 8);
        sp::f16x8 b1[3][TN][2];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int t = 0; t < TN; ++t)
#pragma unroll
                for (int p_ = 0; p_ < 2; ++p_) {
                    const int o = ((2 * t + ky) * IC * 8 + p_ * kInPiece) / 8;
                    const sp::f16x4 lo4 = q[o], hi4 = q[o + 1];
                    b1[ky][t][p_] = __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
                }
        sp::f32x16 acc1[TN];
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[t][r] = 0.0f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int combo = 0; combo < 3; ++combo)
#pragma unroll
                for (int t = 0; t < TN; ++t)
                    acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[ky][combo == 2], b1[ky][t][combo == 1], acc1[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const int y = row0 + 2 * t + ry;
            if (col_ok && y < BH && x < BW) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float z[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) z[j] = fmaxf(fmaf(acc1[t][4 * g + j], k1, bias1[g][j]), 0.0f);
                    zmax = fmaxf(fmaxf(zmax, fmaxf(z[0], z[1])), fmaxf(z[2], z[3]));
                    sp::f16x4 hi, lo;
                    sp::split4(z, hi, lo);
                    char *dst = c1 + ((y + 1) * RW + (x + 1)) * sp::Geo<32>::pos_bytes + (8 * g + 4 * h) * 2;
                    *reinterpret_cast<sp::f16x4 *>(dst) = hi;
                    *reinterpret_cast<sp::f16x4 *>(dst + sp::Geo<32, POS>::piece_bytes) = lo;
                }
            }
        }
    }
    if (hw_pending) {
#pragma unroll
        for (int k = 0; k < kHwPer; ++k) {
            const int i = tid0 + k * kThreads;
            if (i < 128 * 7) hw[i] = hw_reg[k];
        }
        if (tid0 < 8) hw[128 * 7 + tid0] = bh_reg;
        hw_pending = false;
    }
    NET_TICK(0);
    __syncthreads();
    NET_TICK(1);
    if (next_board < n_boards) {
        if (from_bits) load_bits(next_board, tid); else load_obs(next_board, tid);
    }
    sp::f16x8 a3[sp::ring_depth(TM3, TN)][TM3][2];
    {   // conv2: 32 -> 64
        sp::f32x16 acc[TM2][TN];
        f32x4 bias2[TM2][4];  // fetched before the MFMA loop
#pragma unroll
        for (int m = 0; m < TM2; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g) bias2[m][g] = *reinterpret_cast<const f32x4 *>(nd.b2 + (m2 + m) * 32 + 8 * g + 4 * h) * act2;
        if (TN == 2 || (busy && conv2_mine)) sp::conv<32, TM2, TN, 2, RW, POS>(c1, s2p, row0, ry, x, lane, a2, acc);
        NET_TICK(2);
        sp::preload_w<64, TM3, TN>(a3, s3p, lane);
        // (position outermost: ONE guarded region per N-tile instead of one per group of 4 channels)
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const int y = row0 + 2 * t + ry;
            if (busy && conv2_mine && col_ok && y < BH && x < BW) {
                char *pos = c2 + ((y + 1) * RW + (x + 1)) * sp::Geo<64>::pos_bytes + (m2 * 32 + 4 * h) * 2;
#pragma unroll
                for (int m = 0; m < TM2; ++m)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 bv = bias2[m][g];
                        float z[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) z[j] = fmaxf(fmaf(acc[m][t][4 * g + j], k2, bv[j]), 0.0f);
                        zmax = fmaxf(fmaxf(zmax, fmaxf(z[0], z[1])), fmaxf(z[2], z[3]));
                        sp::f16x4 hi, lo;
                        sp::split4(z, hi, lo);
                        char *dst = pos + (m * 32 + 8 * g) * 2;
                        *reinterpret_cast<sp::f16x4 *>(dst) = hi;
                        *reinterpret_cast<sp::f16x4 *>(dst + sp::Geo<64, POS>::piece_bytes) = lo;
                    }
            }
        }
    }
    if (next_board < n_boards) store_obs(tid);
    NET_TICK(3);
    __syncthreads();
    NET_TICK(4);
    {   // conv3: 64 -> 128; its ReLU'd output feeds the two 1x1 head convolutions from registers
        f32x2 vals2[TN][3];  // [position][pair of head outputs]
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int o2 = 0; o2 < 3; ++o2) vals2[t][o2] = f32x2{0.0f, 0.0f};
        f32x2 vals3[3][3];   // MS = 3, wave 3: [tile][pair of head outputs] over the channels of M-tile 3
        if (MS == 3 && part == 1) {
            constexpr int kM = 3;   // the M-tile
            const char *s3q = reinterpret_cast<const char *>(nd.s3) + (size_t)kM * sp::Geo<64>::steps * 2 * 1024;
            sp::f16x8 a3w[sp::ring_depth(1, 3)][1][2];
            sp::preload_w<64, 1, 3>(a3w, s3q, lane);
            sp::f32x16 accw[1][3];
            sp::conv<64, 1, 3, 3, RW, POS>(c2, s3q, 0, ry, x, lane, a3w, accw);
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int o2 = 0; o2 < 3; ++o2) vals3[t][o2] = f32x2{0.0f, 0.0f};
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = kM * 32 + 8 * g + 4 * h;
                f32x4 wc[7];
#pragma unroll
                for (int i = 0; i < 6; ++i) wc[i] = *reinterpret_cast<const f32x4 *>(hw + c0 * 6 + 4 * i);
                wc[6] = *reinterpret_cast<const f32x4 *>(hw + 768 + c0);
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float hv = fmaxf(fmaf(accw[0][t][4 * g + j], k3, wc[6][j]), 0.0f);
#pragma unroll
                        for (int o2 = 0; o2 < 3; ++o2) {
                            const int e = 6 * j + 2 * o2;
                            vals3[t][o2] = __builtin_elementwise_fma(f32x2{wc[e >> 2][e & 3], wc[e >> 2][(e & 3) + 1]},
                                                                     f32x2{hv, hv}, vals3[t][o2]);
                        }
                    }
            }
        }
        if (MS != 3 || part == 0) {
            sp::f32x16 acc[TM3][TN];
            if (TN == 2 || busy) sp::conv<64, TM3, TN, 2, RW, POS>(c2, s3p, row0, ry, x, lane, a3, acc);
            NET_TICK(5);
            // per (m, g): the lane's channels c0 .. c0+3 = 32*m + 8*g + 4*h ..: 24 head weights [j][output] and 4
            // biases from LDS, fetched one group ahead (the fences keep hipcc from hoisting all 16 groups' reads)
            f32x4 w[2][7];
            auto load_group = [&](int mg, f32x4 (&dstw)[7]) {
                const int c0 = (m3 + (mg >> 2)) * 32 + 8 * (mg & 3) + 4 * h;
#pragma unroll
                for (int i = 0; i < 6; ++i) dstw[i] = *reinterpret_cast<const f32x4 *>(hw + c0 * 6 + 4 * i);
                dstw[6] = *reinterpret_cast<const f32x4 *>(hw + 768 + c0);
            };
            load_group(0, w[0]);
#pragma unroll
            for (int mg = 0; mg < 4 * TM3; ++mg) {
                const int m = mg >> 2, g = mg & 3;
                if (mg + 1 < 4 * TM3) load_group(mg + 1, w[(mg + 1) & 1]);
                const f32x4(&wc)[7] = w[mg & 1];
#pragma unroll
                for (int t = 0; t < TN; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float hv = fmaxf(fmaf(acc[m][t][4 * g + j], k3, wc[6][j]), 0.0f);
#pragma unroll
                        for (int o2 = 0; o2 < 3; ++o2) {
                            const int e = 6 * j + 2 * o2;  // float index of (channel j, outputs 2*o2, 2*o2 + 1)
                            vals2[t][o2] = __builtin_elementwise_fma(f32x2{wc[e >> 2][e & 3], wc[e >> 2][(e & 3) + 1]},
                                                                     f32x2{hv, hv}, vals2[t][o2]);
                        }
                    }
                // pin the partial sums here: their only use is the guarded store below, and hipcc otherwise sinks
                // the whole chains of multiply-adds into that block (every weight and activation kept alive)
                if constexpr (TN == 2)
                    asm volatile("" : "+v"(vals2[0][0]), "+v"(vals2[0][1]), "+v"(vals2[0][2]), "+v"(vals2[TN - 1][0]),
                                 "+v"(vals2[TN - 1][1]), "+v"(vals2[TN - 1][2]));
                else
                    asm volatile("" : "+v"(vals2[0][0]), "+v"(vals2[0][1]), "+v"(vals2[0][2]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        NET_TICK(6);
        // the two lane halves hold different channels of the same TN positions: with TN = 2 lane half h stores
        // position h, with TN = 1 half 0 stores the one position
        float *dst = feat ? feat + (size_t)board * nd.feat_ld : nullptr;  // null: only the f16 pieces are wanted
        const int y = row0 + (TN == 2 ? 2 * h : 0) + ry;
        const bool mine = busy && part == 0 && col_ok && (TN == 2 || h == 0);
        // MS > 1: the 6 sums of a position are spread over MS waves (their shares of the 128 channels): they meet in LDS
        float *pad_a = reinterpret_cast<float *>(c1 + ((part * kTiles + tile) * 32 + n) * sp::Geo<32>::pos_bytes + 64);
        float *pad_b = reinterpret_cast<float *>(reinterpret_cast<char *>(pad_a) + sp::Geo<32, POS>::piece_bytes);
        if (MS == 3) {   // wave 3 leaves its share of every tile's sums in slot (tile, n)
            if (part == 1) {
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    float *qa = reinterpret_cast<float *>(c1 + (t * 32 + n) * sp::Geo<32>::pos_bytes + 64);
                    float *qb = reinterpret_cast<float *>(reinterpret_cast<char *>(qa) + sp::Geo<32, POS>::piece_bytes);
#pragma unroll
                    for (int o = 0; o < 6; ++o) {
                        float v0 = vals3[t][o >> 1][o & 1];
                        v0 += other_half(v0, h);
                        if (h == 0) (o < 4 ? qa[o] : qb[o - 4]) = v0;
                    }
                }
            }
            __syncthreads();
        } else if (MS > 1) {
#pragma unroll
            for (int o = 0; o < 6; ++o) {
                float v0 = vals2[0][o >> 1][o & 1];
                v0 += other_half(v0, h);
                if (h == 0) (o < 4 ? pad_a[o] : pad_b[o - 4]) = v0;
            }
            __syncthreads();
        }
        // the same features as hi + lo f16 pieces for the A fragments of k_heads_split:
        // [32-board tile][K-step][hi | lo][board % 32][k % 16] -- the 16 values of a board and K-step are one 32-byte
        // sector (written whole by neighbouring lanes of this wave), a wave of the GEMM reads the 1 KB of a piece
        _Float16 *dst16 = feat16 ? feat16 + ((size_t)(board >> 5) * (nd.groups_act + nd.groups_val) * 1024 + (board & 31) * 16)
                                 : nullptr;
        const bool deferred = RES || later.slot_of != nullptr;   // (DeferredOut: the policy pieces wait in the store, the value inputs go on as f32)
        float *vdst = nullptr;
        if constexpr (RES) {   // the game's slot advances by one per simulation; the value inputs stay in LDS
            dst16 = res_slot0 + sim < later.n_slots ? feat16 + (size_t)(res_slot0 + sim) * later.slot_halfs + (size_t)(board >> 5) * nd.groups_act * 1024 + (board & 31) * 16 : nullptr;
            vdst = res_vrow;
        } else if (deferred) {
            const int slot_ = later.slot_of[board];   // (uniform; beyond the store: nothing is written, expand_backup_body<DEF> flags the game)
            dst16 = slot_ < later.n_slots ? feat16 + (size_t)slot_ * later.slot_halfs + (size_t)(board >> 5) * nd.groups_act * 1024 + (board & 31) * 16 : nullptr;
            vdst = later.valfeat + (size_t)board * later.vf_ld;
        }
        // the six sums of the lane's position first (the other lane half's share by v_permlane32_swap, the biases in one
        // go), then the stores: nothing in the store sequence waits for a cross-lane or LDS round trip
        float vsum[6];
#pragma unroll
        for (int o = 0; o < 6; ++o) {
            float v0 = vals2[0][o >> 1][o & 1], v1 = vals2[TN - 1][o >> 1][o & 1];
            if (MS == 3) {  // channels 0 .. 95 (this wave) + 96 .. 127 (wave 3's slot of this tile)
                const float *q = o < 4 ? pad_a + o : pad_b + (o - 4);   // (part 0: slot (tile, n))
                v0 += other_half(v0, h);
                v0 += q[0];
            } else if (MS > 1) {  // the parts' shares, in part order (every lane reads: part 0's result is the one stored)
                const int stride = kTiles * 32 * sp::Geo<32>::pos_bytes / 4;   // floats from one part's slot to the next
                const float *q = (o < 4 ? pad_a + o : pad_b + (o - 4)) - part * stride;
                v0 = q[0];
#pragma unroll
                for (int p_ = 1; p_ < MS; ++p_) v0 += q[p_ * stride];
            } else {
                v0 += other_half(v0, h);
            }
            if (TN == 2) v1 += other_half(v1, h);
            vsum[o] = (TN == 2 && h) ? v1 : v0;
        }
        float hb[6];
#pragma unroll
        for (int o = 0; o < 6; ++o) hb[o] = hw[128 * 7 + o];
        // FC_HERE: the board's f16 feature pieces [K-step][hi | lo][16] in LDS, inside halo rows 12 .. of conv2's region (a
        // board of up to 10 rows never reads them); zeroed K tail
        const bool fc_here = TN == 1 && raw != nullptr;
        _Float16 *fa_lds = reinterpret_cast<_Float16 *>(c2 + 12 * RW * sp::Geo<64>::pos_bytes);
        const int fc_steps = nd.groups_act + nd.groups_val;
        if (fc_here) {
            for (int i = tid; i < fc_steps * 8; i += kThreads) reinterpret_cast<f32x2 *>(fa_lds)[i] = f32x2{0.0f, 0.0f};
            __syncthreads();
        }
        if (mine && y < BH && x < BW) {
            const int cell = y * BW + x;
#pragma unroll
            for (int o = 0; o < 6; ++o) {
                const float v = fmaxf(vsum[o] + hb[o], 0.0f);
                if (dst) dst[(o < 4 ? o * S : nd.feat_val_off + (o - 4) * S) + cell] = v;
                if (fc_here) {
                    const int k = (o < 4 ? o : o - 4) * S + cell;
                    const int step = (o < 4 ? 0 : nd.groups_act) + (k >> 4);
                    const float z = v * act3;
                    const _Float16 zh = (_Float16)z;
                    zmax = fmaxf(zmax, z);
                    fa_lds[step * 32 + (k & 15)] = zh;
                    fa_lds[step * 32 + 16 + (k & 15)] = (_Float16)(z - (float)zh);
                } else if (deferred && o >= 4) {
                    vdst[(o - 4) * S + cell] = v;
                } else if (dst16) {
                    const int k = (o < 4 ? o : o - 4) * S + cell;
                    const int step = (o < 4 ? 0 : nd.groups_act) + (k >> 4);
                    const float z = v * act3;
                    const _Float16 zh = (_Float16)z;
                    zmax = fmaxf(zmax, z);
                    _Float16 *q = dst16 + (size_t)step * 1024 + (k & 15);
                    q[0] = zh;
                    q[512] = (_Float16)(z - (float)zh);
                }
            }
        }
    }
    NET_TICK(7);
    if constexpr (TN == 1) {
        if (raw != nullptr) {
            // ---- the first FC layers of both heads on this board (k_heads_split's arithmetic: see the kernel's header)
            __syncthreads();   // the feature pieces are in LDS
            const _Float16 *fa_lds = reinterpret_cast<const _Float16 *>(c2 + 12 * RW * sp::Geo<64>::pos_bytes);
            float *ps = reinterpret_cast<float *>(c2 + 15 * RW * sp::Geo<64>::pos_bytes);   // [K quarter][tile][32 outputs]
            const int n_act_tiles = nd.Npad / 32, n_tiles = n_act_tiles + 2;
            const f32x4 *zero_frag = nd.fs_act + (size_t)n_act_tiles * nd.groups_act * 128;
            const int half = lane >> 5;
            const bool row0 = (lane & 31) == 0;   // the lanes that hold MFMA row 0 of the A operand
            for (int tile = 0; tile < n_tiles; ++tile) {
                const bool is_act = tile < n_act_tiles;
                const f32x4 *fb = is_act ? nd.fs_act + (size_t)tile * nd.groups_act * 128
                                         : nd.fs_val + (size_t)(tile - n_act_tiles) * nd.groups_val * 128;
                const int K = is_act ? nd.groups_act : nd.groups_val, a0 = is_act ? 0 : nd.groups_act;
                const int k0 = wave * K / 4, k1 = (wave + 1) * K / 4;
                sp::f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
                constexpr int D = 4;   // weight fragments in flight (every load unconditional: see fs_load)
                sp::f16x8 bw[D][2];
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const f32x4 *src = k0 + d < k1 ? fb + (size_t)(k0 + d) * 128 : zero_frag;
#pragma unroll
                    for (int p_ = 0; p_ < 2; ++p_) bw[d][p_] = __builtin_bit_cast(sp::f16x8, src[p_ * 64 + lane]);
                }
                for (int k = k0; k < k1; k += D) {
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        if (k + d < k1) {   // (wave-uniform)
                            const sp::f16x8 zero8 = {(_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f,
                                                     (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
                            const _Float16 *src = fa_lds + (a0 + k + d) * 32 + 8 * half;
                            const sp::f16x8 ah = row0 ? *reinterpret_cast<const sp::f16x8 *>(src) : zero8;
                            const sp::f16x8 al = row0 ? *reinterpret_cast<const sp::f16x8 *>(src + 16) : zero8;
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bw[d][0], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bw[d][1], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bw[d][0], acc, 0, 0, 0);
                        }
                        const f32x4 *src2 = k + d + D < k1 ? fb + (size_t)(k + d + D) * 128 : zero_frag;
#pragma unroll
                        for (int p_ = 0; p_ < 2; ++p_) bw[d][p_] = __builtin_bit_cast(sp::f16x8, src2[p_ * 64 + lane]);
                    }
                }
                if (half == 0) ps[(wave * n_tiles + tile) * 32 + (lane & 31)] = acc[0];   // D row 0 = this board, column = output
            }
            __syncthreads();
            const float sc_act = nd.s_inv[3], sc_val = nd.s_inv[4];
            for (int i = tid; i < n_tiles * 32; i += kThreads) {
                const int tile = i >> 5, col = i & 31;
                float v = ps[(0 * n_tiles + tile) * 32 + col];
#pragma unroll
                for (int q = 1; q < 4; ++q) v += ps[(q * n_tiles + tile) * 32 + col];
                if (tile < n_act_tiles) {
                    const int c = 32 * tile + col;
                    raw[(size_t)board * nd.Npad + c] = fmaf(v, sc_act, nd.fc_act_b[c]);
                } else {
                    const int c = 32 * (tile - n_act_tiles) + col;
                    hid[(size_t)board * 64 + c] = fmaxf(fmaf(v, sc_val, nd.fc_val1_b[c]), 0.0f);
                }
            }
            __syncthreads();   // (a further board of this workgroup reuses the pieces and the partial sums)
        }
    }
    if constexpr (RES) {
        // ---- the rest of the simulation, by the same workgroup (see k_trunk_rows: the body of k_tree_step_def, rz_tree.h)
        __syncthreads();   // the value head's inputs are in LDS
        const int game = blockIdx.x;
        // (two workgroups per CU -- the compact grid --: the serial part of a simulation ahead of the other game's trunk waves at issue,
        // as in k_delta_res)
        if (RW != kRowW && RZ_SPLIT_TREE_PRIO) __builtin_amdgcn_s_setprio(RZ_SPLIT_TREE_PRIO);
        if (res.vh.groups == 64) rzt::value_quarter_lds<8>(res.vh, res_vrow, lane, wave, res_part);
        else if (res.vh.groups == 32) rzt::value_quarter_lds<4>(res.vh, res_vrow, lane, wave, res_part);
        else rzt::value_quarter_lds<2>(res.vh, res_vrow, lane, wave, res_part);
        NET_TICK(16);
        if (wave == 0) rzt::expand_backup_body<float, false, false, false, true, kResWords>(res.E, nullptr, nullptr, game, lane, rz_raw_heads(), 0, res.vh, res_part);
        else __syncthreads();   // (the barrier inside the body, where the quarters meet)
        __syncthreads();        // the tree's updates before the selection's loads
        NET_TICK(17);
        const bool more = sim + 1 < res_sims(res);
        if (wave == 0 && more) rzt::select_body<false, kResWords>(res.E, nullptr, game, lane, 0, res_leaf);
        if (RW != kRowW && RZ_SPLIT_TREE_PRIO) __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        NET_TICK(18);
        if (more) {   // the planes of the next leaf, from LDS: what load_bits forms from the leaf arrays
            planes_from_lds(tid);
            store_obs(tid);
            __syncthreads();
        }
        NET_TICK(19);
    }
    }  // boards
#ifdef RZ_NET_PROFILE
    if (blockIdx.x == 0 && tid0 == 0) {
        for (int i = 0; i < 24; ++i) net_prof[i] = prof_acc[i];
        net_prof[10] = __builtin_readcyclecounter() - prof_k0;
    }
#endif
    if (!(zmax <= 65504.0f)) atomicOr(flags, (unsigned)RZ_NET_FLAG_F16_RANGE);
}

// ------------------------------------------------------------------ row-tile trunk (wide boards: 11 .. 16 columns)
// The arithmetic of k_trunk_split (hi + lo f16 operands, three MFMAs per product, f32 accumulation) on
// v_mfma_f32_16x16x32_f16 with the work split over the waves by OUTPUT CHANNEL instead of by board row:
//   * N-tile = ONE board row (16 columns), K-step = 32 input channels of one tap, M-tile = 16 output channels.  A 15 x 15
//     board is 15 N-tiles (240 MFMA columns for 225 positions) where 2-row x 16-column tiles of the 32 x 32 MFMA need 8 x 32 =
//     256, and no wave owns a row that does not exist.
//   * wave w owns output channels 16 w .. 16 w + 15 of conv2 and 32 w .. 32 w + 31 of conv3 for ALL rows: every weight
//     fragment is fetched by one wave instead of four (L2 -> CU traffic of conv3: 295 KB per board instead of 1.18 MB), the
//     activation fragments (LDS) by all four.
//   * the chip holds a higher clock in this MFMA shape (profiles/r03/conv3_shapes.txt: the conv3 loop on every CU, random
//     data: 1.75 GHz against 1.52 GHz and 6 % fewer cycles: 17.1 against 20.5 us per board).
// LDS: a position's record is [hi: CIN f16][lo: CIN f16][32 bytes of padding] (160 / 288 bytes): lane = 16 * (k block) +
// column reads the 8 channels of its k block with one ds_read_b128, and record size / 16 = 2 (mod 4) puts the 16 lanes of
// every LDS cycle on 16 different 16-byte slots.  The head convolutions: a lane holds 8 of the 128 channels of a position,
// so the 6 sums of a position are spread over 4 lanes x 4 waves: lanes meet by v_permlane16/32_swap (reduce-scatter), waves
// in LDS (inside the board positions of conv1's region, which the next board overwrites anyway); thread = cell then adds
// the four waves' shares in wave order and stores the features.
namespace rt {

using sp::f16x4;
using sp::f16x8;
using sp::lds_frag;
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int CIN> struct Geo {
    static constexpr int pos_bytes = 4 * CIN + 32;               // 160 / 288
    static constexpr int grid_bytes = sp::kGridPos * pos_bytes;  // 51 840 / 93 312
    static constexpr int chunks = CIN / 32, steps = 9 * chunks;
    static_assert((pos_bytes / 16) % 4 == 2, "conflict-free ds_read_b128");
};
constexpr int kLA = 3;   // activation fragments are requested kLA rows ahead
constexpr int kLdsBytes = sp::kInBytes + Geo<32>::grid_bytes + Geo<64>::grid_bytes;
static_assert(kLdsBytes <= 160 * 1024, "LDS budget");

// The K loop, rows innermost: for every (tap column dx, channel chunk) -- a "combo" -- the NT live halo rows 1 .. NT are read
// ONCE each and a row's fragment meets the three kernel rows (output rows r - dy): 9 * TM MFMAs per pair of ds_read_b128, a
// third of the LDS reads of a (tap, row) order (profiles/microbench/conv3_shapes.hip: R16 against C16; +2.1 .. 2.7 % on the whole
// bench); the 3 * TM weight fragments of a combo's three kernel rows sit in registers, the next combo's arrive meanwhile.
// Halo rows 0 and NT + 1 are the zero ring above / below the board (the kernel runs boards of exactly NT rows): never read,
// their products never formed (2 of 45 (row, kernel row) pairs = 4.4 % of a layer's MFMAs; +2.3 %).
// F8 (conv3 of RZ_NET_SPLIT_F16_FP8; CIN = 64): a position's record is [hi: 64 f16][hi8: 64 e5m2 of the value][lo8: 64 e5m2 of
// (value - hi) 2^11][pad] and a "combo" is (tap column dx, part): part 0 = the hi x hi products of the tap's two 32-channel chunks (two
// v_mfma_f32_16x16x32_f16 per kernel row and M-tile), part 1 = BOTH cross terms of the tap's 64 channels in one
// v_mfma_scale_f32_16x16x128_f8f6f4 (A = e4m3 weights, B = e5m2 activations; lane group g: K block = [hi8 x (w_lo 2^5)8 | lo8 x
// (w_hi 2^-6)8] of channels 16 g .. 16 g + 15, one scale 2^-5 for the block).  The same fragment addresses, loads per slot and
// registers as the three-MFMA loop; 2 f16-MFMA equivalents per product instead of 3 (profiles/microbench/conv3_shapes.hip: F8).
__device__ __forceinline__ i32x8 cat8(f16x8 lo, f16x8 hi) {
    return __builtin_shufflevector(__builtin_bit_cast(i32x4, lo), __builtin_bit_cast(i32x4, hi), 0, 1, 2, 3, 4, 5, 6, 7);
}
constexpr int kF8ScaleA = 127 - 5, kF8ScaleB = 127;   // E8M0: the weights' bytes carry 2^5 (pack_rows_f8), the activations' 2^0
template <int CIN, int TM, int NT, int J, bool F8 = false>
__device__ __forceinline__ void slot_r(f32x4 (&acc)[TM][NT], f16x8 (&a)[2][3][TM][2], f16x8 (&b)[kLA + 1][2], lds_frag q, lds_frag qf,
                                       __amdgpu_buffer_rsrc_t w_rsrc, int w_lane) {
    using G = Geo<CIN>;
    static_assert(!F8 || CIN == 64, "the FP8 cross terms: one tap of 64 channels = one K = 128 block");
    constexpr int PD = kLA + 1, combos = 3 * G::chunks, cb = J / NT, r = J % NT + 1, J2 = J + kLA;   // r: halo row
    constexpr int second = F8 ? 64 : CIN * 2;   // the second fragment of a slot: the other chunk (F8) / the lo piece
    if constexpr (J2 < combos * NT) {
        constexpr int cb2 = J2 / NT, r2 = J2 % NT + 1, dx = cb2 / G::chunks, c = cb2 % G::chunks, far = r2 >= 8;
        constexpr int off = ((r2 - 8 * far) * kRowW + dx) * G::pos_bytes + c * (F8 ? 128 : 64);
        static_assert(off % 16 == 0 && off + second < 65536, "ds_read_b128 immediate");
        b[J2 % PD][0] = (far ? qf : q)[off / 16];
        b[J2 % PD][1] = (far ? qf : q)[(off + second) / 16];
    }
    if constexpr (cb + 1 < combos && r - 1 < 3 * TM) {   // the next combo's weight fragments behind this combo's first rows
        constexpr int dy = (r - 1) / TM, m = (r - 1) % TM, dx = (cb + 1) / G::chunks, c = (cb + 1) % G::chunks;
#pragma unroll
        for (int p = 0; p < 2; ++p)
            a[(cb + 1) % 2][dy][m][p] = sp::load_w(w_rsrc, w_lane, ((m * G::steps + (dy * 3 + dx) * G::chunks + c) * 2 + p) * 1024);
    }
    if constexpr (F8) {
        constexpr int part = cb % 2;
#pragma unroll
        for (int c = 0; c < (part ? 1 : 2); ++c)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int m = 0; m < TM; ++m) {
                    const int t = r - dy;
                    if (t >= 0 && t < NT) {
                        if (part) {
                            acc[m][t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(cat8(a[cb % 2][dy][m][0], a[cb % 2][dy][m][1]),
                                                                                         cat8(b[J % PD][0], b[J % PD][1]), acc[m][t],
                                                                                         0 /* A: e4m3 */, 1 /* B: e5m2 */, 0, kF8ScaleA, 0, kF8ScaleB);
                        } else if (cb == 0 && c == 0 && (dy == 0 || (t == 0 && dy == 1))) {   // a row's first product (as below)
                            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                            acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0][dy][m][0], b[J % PD][0], zero, 0, 0, 0);
                        } else {
                            acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[cb % 2][dy][m][c], b[J % PD][c], acc[m][t], 0, 0, 0);
                        }
                    }
                }
        __builtin_amdgcn_sched_barrier(0);
        return;
    }
#pragma unroll
    for (int combo = 0; combo < 3; ++combo)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int m = 0; m < TM; ++m) {
                const int t = r - dy, pa = combo == 2 ? 1 : 0, pb = combo == 1 ? 1 : 0;   // halo row r = board row r - 1 = tap row dy of output row r - dy
                if (t >= 0 && t < NT) {
                    // a row's first product: combo 0 of (cb = 0, dy = 0) at halo row t -- board row 0 would meet dy = 0 at halo
                    // row 0 (the zero ring, never read): its first is dy = 1 at halo row 1
                    if (cb == 0 && combo == 0 && (dy == 0 || (t == 0 && dy == 1))) {
                        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[cb % 2][dy][m][pa], b[J % PD][pb], zero, 0, 0, 0);
                    } else {
                        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[cb % 2][dy][m][pa], b[J % PD][pb], acc[m][t], 0, 0, 0);
                    }
                }
            }
    __builtin_amdgcn_sched_barrier(0);
}

template <int CIN, int TM, int NT, bool F8, int... Js>
__device__ __forceinline__ void slots_r(std::integer_sequence<int, Js...>, f32x4 (&acc)[TM][NT], f16x8 (&a)[2][3][TM][2],
                                        f16x8 (&b)[kLA + 1][2], lds_frag q, lds_frag qf, __amdgpu_buffer_rsrc_t w_rsrc, int w_lane) {
    (slot_r<CIN, TM, NT, Js, F8>(acc, a, b, q, qf, w_rsrc, w_lane), ...);
}

// the weight fragments of combo 0 (tap column 0, chunk 0; kernel rows 0 .. 2): requested while the previous layer is reduced
template <int CIN, int TM>
__device__ __forceinline__ void preload_w_r(f16x8 (&a)[2][3][TM][2], const void *wts, int lane) {
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(wts), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                a[0][dy][m][p] = sp::load_w(w_rsrc, lane * 16, ((m * Geo<CIN>::steps + dy * 3 * Geo<CIN>::chunks) * 2 + p) * 1024);
}

template <int CIN, int TM, int NT, bool F8 = false>
__device__ __forceinline__ void conv_r(const char *in, const void *wts, int lane, f16x8 (&a)[2][3][TM][2], f32x4 (&acc)[TM][NT]) {
    using G = Geo<CIN>;
    static_assert(3 * TM <= NT && kLA <= NT, "loads are spread over a combo's first rows");
    const int n = lane & 15, g = lane >> 4;
    const lds_frag q = (lds_frag)(in + n * G::pos_bytes + g * 16), qf = q + 8 * kRowW * G::pos_bytes / 16;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(wts), 0, 0x7fffffff, 0x00020000);
    f16x8 b[kLA + 1][2];
#pragma unroll
    for (int j = 0; j < kLA; ++j) {   // slots 0 .. kLA - 1: combo 0 (dx = 0, chunk 0), halo rows 1 ..
        b[j][0] = q[((j + 1) * kRowW * G::pos_bytes) / 16];
        b[j][1] = q[((j + 1) * kRowW * G::pos_bytes + (F8 ? 64 : CIN * 2)) / 16];
    }
    __builtin_amdgcn_sched_barrier(0);
    slots_r<CIN, TM, NT, F8>(std::make_integer_sequence<int, 3 * G::chunks * NT>{}, acc, a, b, q, qf, w_rsrc, lane * 16);
}

}  // namespace rt

#include "rz_delta.h"

// Register budget: 200 VGPRs + 200 accumulation registers.  Two lanes of games overlap because a wave of the other lane's tree
// step or FC GEMM (112 / 104 registers) fits beside a trunk wave on the same SIMD (512 registers): at 400 registers or fewer the
// trunk leaves that room, at 408 it does not and the lanes' kernels take turns (measured: 10.7 -> 9.5 M sims/s from 8 registers).
// Left alone hipcc allocates 396 .. 420 here depending on details of the prologue; the cap holds it at 372, no scratch.
// TRACE (rz_trace.h): instantiated for the 15-row bitboard kernel only -- the layout whose schedule profiles/lane_timeline.py reads;
// the production kernels carry nothing of it (its live values cost ten registers of a budget that is pinned).
// F8: conv3 with its cross terms on the block-scaled FP8 pipe (RZ_NET_SPLIT_F16_FP8, opt-in: narrower arithmetic than the reference's
// f32 -- rt::slot_r): conv2's epilogue stores the e5m2 pieces where the lo f16 pieces stood, conv3 reads nd.t3f.
// The body is shared by two kernels: k_trunk_rows (the launches of a lane's step: the register cap above) and k_trunk_rows_res (the
// resident search: a workgroup has its CU to itself for a whole search, no other lane's waves to make room for -- no cap, so the
// tree code's registers beside the trunk's need no scratch).
template <int NT, bool BITS, bool TRACE, bool RES, bool F8>
__device__ __forceinline__ void trunk_rows_body(const NetDev &nd, const float *__restrict__ obs, LeafBits leaves,
                                                float *__restrict__ feat, _Float16 *__restrict__ feat16,
                                                int n_boards, unsigned *__restrict__ flags, const DeferredOut &later, const ResArgs<RES> &res) {
    static_assert(!RES || (BITS && !TRACE), "the resident search reads positions");
    static_assert(!F8 || (BITS && !TRACE), "the FP8 cross terms: position-fed launches only");
    // RES: the value head's input row (zero padded to 4 x groups floats), the K-quarter sums of its first layer, the next leaf
    __shared__ float res_vrow[RES ? 512 : 1];
    __shared__ float res_part[RES ? rzt::kDefWaves : 1][RES ? rzt::kWave : 1];
    __shared__ __attribute__((aligned(16))) uint64_t res_leaf[RES ? 2 * RZ_BOARD_WORDS + 1 : 1];
    int res_slot0 = 0;
    if constexpr (RES) {
        if ((int)blockIdx.x >= n_boards || res.E.active[blockIdx.x] == 0) return;   // (uniform: before any barrier)
        res_slot0 = res.E.pend[blockIdx.x];
        for (int i = threadIdx.x; i < 512; i += 256) res_vrow[i] = 0.0f;
    }
#ifdef RZ_NET_PROFILE
    const long long prof_k0 = __builtin_readcyclecounter();
    long long prof_acc[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, prof_t = prof_k0;
#endif
    constexpr int kThreads = 256;
    constexpr int P1 = rt::Geo<32>::pos_bytes, P2 = rt::Geo<64>::pos_bytes;
    __shared__ __attribute__((aligned(16))) char lds_raw[rt::kLdsBytes];
    char *in0 = lds_raw;                      // observation planes, pieces hi | lo (as k_trunk_split)
    char *c1 = lds_raw + sp::kInBytes;        // conv1 output, records [hi 32 | lo 32 | pad]
    char *c2 = c1 + rt::Geo<32>::grid_bytes;  // conv2 output, records [hi 64 | lo 64 | pad]
    const int tid0 = threadIdx.x;
    __shared__ unsigned long long trace_t0;   // (parked in LDS: the register budget below is pinned)
    if (TRACE && tid0 == 0) trace_t0 = rz_trace_now();
    const int BH = nd.BH, BW = nd.BW, S = nd.S;
    float zmax = 0.0f;  // largest scaled value this thread stored as f16 pieces (float planes only: bitboard planes are 0 / 1)
    constexpr int kObsPer = BITS ? 1 : (4 * RZ_MAX_BOARD_SIZE * RZ_MAX_BOARD_SIZE + kThreads - 1) / kThreads;
    float ob[kObsPer];
    int obs_off[kObsPer];
    if constexpr (!BITS) {
#pragma unroll
        for (int k = 0; k < kObsPer; ++k) {
            const int i = tid0 + k * kThreads;
            const int c = i / S, r = i - c * S, y = r / BW, x = r - y * BW;
            obs_off[k] = i < 4 * S ? ((y + 1) * sp::kInCols + (x + 1)) * 8 + c * 2 : -1;
        }
    }
    auto load_obs = [&](int board, int tid) {
        if constexpr (!BITS) {
            const float *src = obs + (size_t)board * 4 * S;
#pragma unroll
            for (int k = 0; k < kObsPer; ++k) {
                const int i = tid + k * kThreads;
                ob[k] = i < 4 * S ? src[i] : 0.0f;
            }
        }
    };
    // thread t owns cell t (S <= 256 = threads): the planes of the bitboard route, and the cell whose features it stores
    const int cell_y = tid0 / BW, cell_x = tid0 - cell_y * BW;
    const int cell_off = tid0 < S ? ((cell_y + 1) * sp::kInCols + (cell_x + 1)) * 8 : -1;
    sp::f16x4 cell_planes = {(_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
    auto load_bits = [&](int board, int tid) {
        const uint64_t *sb = leaves.stones + (size_t)board * 8;
        const int tm = leaves.to_move[board], lc = leaves.last[board];
        int nst = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) nst += __popcll(sb[q]);  // (uniform address: scalar loads)
        const int word = (tid >> 6) & 3, bit = tid & 63;
        const uint64_t w0 = sb[word], w1 = sb[4 + word];
        const bool s0 = (w0 >> bit) & 1ull, s1 = (w1 >> bit) & 1ull;
        const bool mine = tm == 0 ? s0 : s1, theirs = tm == 0 ? s1 : s0;
        const _Float16 one = (_Float16)sp::kObsScale, zero = (_Float16)0.0f;
        cell_planes[0] = mine ? one : zero;
        cell_planes[1] = theirs ? one : zero;
        cell_planes[2] = (nst > 0 && tid == lc) ? one : zero;
        cell_planes[3] = (nst & 1) ? zero : one;
    };
    auto load_board = [&](int board, int tid) {
        if constexpr (BITS) load_bits(board, tid); else load_obs(board, tid);
    };
    auto store_obs = [&]() {
        if constexpr (BITS) {
            if (cell_off >= 0) {
                *reinterpret_cast<sp::f16x4 *>(in0 + cell_off) = cell_planes;
                *reinterpret_cast<sp::f16x4 *>(in0 + sp::kInPieceBytes + cell_off) =
                    sp::f16x4{(_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
            }
        } else {
#pragma unroll
            for (int k = 0; k < kObsPer; ++k)
                if (obs_off[k] >= 0) {
                    const float z = ob[k] * sp::kObsScale;
                    const _Float16 hi = (_Float16)z;
                    zmax = fmaxf(zmax, fabsf(z));
                    *reinterpret_cast<_Float16 *>(in0 + obs_off[k]) = hi;
                    *reinterpret_cast<_Float16 *>(in0 + sp::kInPieceBytes + obs_off[k]) = (_Float16)(z - (float)hi);
                }
        }
    };
    // Prologue of a persistent workgroup: every global load is issued first, the LDS is zeroed under their latency.
    const int lane0 = tid0 & 63, wave0 = tid0 >> 6, g0 = lane0 >> 4;
    // the 1x1 head convolutions: the lane's 8 channels of conv3 (32 wave + 16 m + 4 g + j) meet 6 outputs each -- 48 weights
    // and 8 biases that never change: registers for the whole launch
    f32x4 hwr[2][6], b3r[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int c0 = 32 * wave0 + 16 * m + 4 * g0;
#pragma unroll
        for (int i = 0; i < 6; ++i) hwr[m][i] = *reinterpret_cast<const f32x4 *>(nd.whp + c0 * 6 + 4 * i);
        b3r[m] = *reinterpret_cast<const f32x4 *>(nd.b3 + c0);
    }
    float hb[6];
#pragma unroll
    for (int o = 0; o < 6; ++o) hb[o] = nd.bh[o];
    const float k1 = nd.s_inv[2], k2 = nd.s_inv[0], k3 = nd.s_inv[1];
    const float act1 = nd.s_inv[5], act2 = nd.s_inv[6], act3 = nd.s_inv[7];
    // conv1's weights (3 kernel rows x hi / lo) and biases stay in registers for all boards (32 x 32 x 16 tiles, as k_trunk_split)
    sp::f16x8 a1[3][2];
    f32x4 bias1[4];
    {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int p_ = 0; p_ < 2; ++p_) a1[ky][p_] = __builtin_bit_cast(sp::f16x8, nd.s1[(ky * 2 + p_) * 64 + lane0]);
#pragma unroll
        for (int g = 0; g < 4; ++g) bias1[g] = *reinterpret_cast<const f32x4 *>(nd.b1 + 8 * g + 4 * (lane0 >> 5)) * act1;
    }
    const f32x4 bias2 = *reinterpret_cast<const f32x4 *>(nd.b2 + 16 * wave0 + 4 * g0) * act2;
    const bool first = (int)blockIdx.x < n_boards;
    bool sel_first = false;   // RES: the first leaf is selected by this launch (below, behind the zeroing)
    if constexpr (RES) sel_first = res.select_first != 0;
    if (first && !sel_first) load_board(blockIdx.x, tid0);
    // the planes of a leaf handed over through LDS by the tree code of this workgroup (select_body's lds_leaf): what load_bits forms
    auto planes_from_lds = [&](int tid) {
        int nst = 0;
#pragma unroll
        for (int q8 = 0; q8 < 8; ++q8) nst += __popcll(res_leaf[q8]);
        const int tm = reinterpret_cast<const int *>(res_leaf + 2 * RZ_BOARD_WORDS)[0], lc = reinterpret_cast<const int *>(res_leaf + 2 * RZ_BOARD_WORDS)[1];
        const int word = (tid >> 6) & 3, bit = tid & 63;
        const uint64_t w0 = res_leaf[word], w1 = res_leaf[4 + word];
        const bool s0 = (w0 >> bit) & 1ull, s1 = (w1 >> bit) & 1ull;
        const bool mine = tm == 0 ? s0 : s1, theirs = tm == 0 ? s1 : s0;
        const _Float16 one = (_Float16)sp::kObsScale, zero = (_Float16)0.0f;
        cell_planes[0] = mine ? one : zero;
        cell_planes[1] = theirs ? one : zero;
        cell_planes[2] = (nst > 0 && tid == lc) ? one : zero;
        cell_planes[3] = (nst & 1) ? zero : one;
    };
    __builtin_amdgcn_sched_barrier(0);  // the loads above stay above the zeroing
    NET_TICK(11);
    {
        // what a valid position reads and no board writes must be zero: the observation planes' halo (all of in0) and, in
        // c1 / c2, the ring of positions around the board (whole records)
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        f32x4 *z = reinterpret_cast<f32x4 *>(lds_raw);
        for (int i = tid0; i < sp::kInBytes / 16; i += kThreads) z[i] = zero;
        const int n_ring = 2 * (BW + 2) + 2 * BH;
        for (int idx = tid0; idx < n_ring; idx += kThreads) {
            int py, px;
            if (idx < 2 * (BW + 2)) {
                const int bottom = idx >= BW + 2;
                py = bottom ? BH + 1 : 0;
                px = idx - bottom * (BW + 2);
            } else {
                const int j = idx - 2 * (BW + 2);
                py = 1 + (j >> 1);
                px = (j & 1) ? BW + 1 : 0;
            }
            const int pos = py * kRowW + px;
            f32x4 *q1 = reinterpret_cast<f32x4 *>(c1 + pos * P1);
#pragma unroll
            for (int i = 0; i < P1 / 16; ++i) q1[i] = zero;
            f32x4 *q2 = reinterpret_cast<f32x4 *>(c2 + pos * P2);
#pragma unroll
            for (int i = 0; i < P2 / 16; ++i) q2[i] = zero;
        }
    }
    NET_TICK(12);
    __syncthreads();
    NET_TICK(13);
    if constexpr (RES) {
        if (sel_first) {   // AlphaZeroMCTS._playout's select loop for the first simulation of the search (rz_select_step's work)
            if (wave0 == 0) rzt::select_body<false>(res.E, nullptr, blockIdx.x, lane0, 0, res_leaf);
            __syncthreads();
            planes_from_lds(tid0);
        }
    }
    if (first) store_obs();
    NET_TICK(14);
    __syncthreads();
#ifdef RZ_NET_PROFILE
    NET_TICK(15);
    prof_acc[9] = prof_t - prof_k0;   // the prologue
#endif
    // (RES: the "boards" of this workgroup are the leaves of its game's simulations, one after the other)
    for (int board = blockIdx.x, sim = 0; RES ? sim < res_sims(res) : board < n_boards; RES ? (void)++sim : (void)(board += gridDim.x)) {
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int next_board = RES ? n_boards : board + (int)gridDim.x;   // (RES: the next leaf does not exist yet)
    const int n = lane & 15, g = lane >> 4;
    const char *t2p = reinterpret_cast<const char *>(nd.t2) + (size_t)wave * rt::Geo<32>::steps * 2 * 1024;
    const char *t3p = reinterpret_cast<const char *>(F8 ? nd.t3f : nd.t3) + (size_t)(2 * wave) * rt::Geo<64>::steps * 2 * 1024;
    sp::f16x8 a2[2][3][1][2];
    rt::preload_w_r<32, 1>(a2, t2p, lane);
    {   // conv1: 4 -> 32 on 32 x 32 x 16 tiles of 2 rows x 16 columns, wave w = rows 4 w .. 4 w + 3; K-step = kernel row
        const int n32 = lane & 31, h = lane >> 5, ry = n32 >> 4, x = n32 & 15, row0 = 4 * wave;
        if (row0 < BH) {
            typedef const __attribute__((address_space(3))) sp::f16x4 *lds_half;
            const lds_half q = (lds_half)(in0 + ((row0 + ry) * sp::kInCols + x + 2 * h) * 8);
            sp::f16x8 b1[3][2][2];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int p_ = 0; p_ < 2; ++p_) {
                        const int o = ((2 * t + ky) * sp::kInCols * 8 + p_ * sp::kInPieceBytes) / 8;
                        const sp::f16x4 lo4 = q[o], hi4 = q[o + 1];
                        b1[ky][t][p_] = __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
                    }
            sp::f32x16 acc1[2];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc1[t][r] = 0.0f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int combo = 0; combo < 3; ++combo) {
                    if (BITS && combo == 1) continue;   // the lo pieces of 0 / 1 planes are zero
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        acc1[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[ky][combo == 2], b1[ky][t][combo == 1], acc1[t], 0, 0, 0);
                }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int y = row0 + 2 * t + ry;
                if (y < BH && x < BW) {
                    char *pos = c1 + ((y + 1) * kRowW + (x + 1)) * P1 + 4 * h * 2;
#pragma unroll
                    for (int gg = 0; gg < 4; ++gg) {
                        float z[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) z[j] = fmaxf(fmaf(acc1[t][4 * gg + j], k1, bias1[gg][j]), 0.0f);
                        if constexpr (!BITS) zmax = fmaxf(fmaxf(zmax, fmaxf(z[0], z[1])), fmaxf(z[2], z[3]));
                        sp::f16x4 hi, lo;
                        sp::split4(z, hi, lo);
                        *reinterpret_cast<sp::f16x4 *>(pos + 8 * gg * 2) = hi;
                        *reinterpret_cast<sp::f16x4 *>(pos + 8 * gg * 2 + 64) = lo;
                    }
                }
            }
        }
    }
    NET_TICK(0);
    __syncthreads();
    NET_TICK(1);
    if (next_board < n_boards) load_board(next_board, tid);
    sp::f16x8 a3[2][3][2][2];
    {   // conv2: 32 -> 64, wave w = output channels 16 w .. 16 w + 15
        f32x4 acc[1][NT];
        rt::conv_r<32, 1, NT>(c1, t2p, lane, a2, acc);
        NET_TICK(2);
        rt::preload_w_r<64, 2>(a3, t3p, lane);
        if (n < BW) {
            char *pos = c2 + (kRowW + n + 1) * P2 + (16 * wave + 4 * g) * 2;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float z[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) z[j] = fmaxf(fmaf(acc[0][t][j], k2, bias2[j]), 0.0f);
                if constexpr (!BITS) zmax = fmaxf(fmaxf(zmax, fmaxf(z[0], z[1])), fmaxf(z[2], z[3]));
                if constexpr (F8) {   // [hi f16 | e5m2 of the value | e5m2 of (value - hi) 2^11]: the lane's 4 channels, 8 + 4 + 4 bytes
                    typedef float f32x4v __attribute__((ext_vector_type(4)));
                    const sp::f16x4 hi = __builtin_convertvector((f32x4v){z[0], z[1], z[2], z[3]}, sp::f16x4);
                    *reinterpret_cast<sp::f16x4 *>(pos + t * kRowW * P2) = hi;
                    int h8 = __builtin_amdgcn_cvt_pk_bf8_f32(z[0], z[1], 0, false);
                    h8 = __builtin_amdgcn_cvt_pk_bf8_f32(z[2], z[3], h8, true);
                    float d[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) d[j] = (z[j] - (float)hi[j]) * 2048.0f;
                    int l8 = __builtin_amdgcn_cvt_pk_bf8_f32(d[0], d[1], 0, false);
                    l8 = __builtin_amdgcn_cvt_pk_bf8_f32(d[2], d[3], l8, true);
                    char *p8 = c2 + (kRowW + n + 1) * P2 + 128 + 16 * wave + 4 * g + t * kRowW * P2;
                    *reinterpret_cast<int *>(p8) = h8;
                    *reinterpret_cast<int *>(p8 + 64) = l8;
                } else {
                sp::f16x4 hi, lo;
                sp::split4(z, hi, lo);
                *reinterpret_cast<sp::f16x4 *>(pos + t * kRowW * P2) = hi;
                *reinterpret_cast<sp::f16x4 *>(pos + t * kRowW * P2 + 128) = lo;
                }
            }
        }
    }
    if (next_board < n_boards) store_obs();
    NET_TICK(3);
    __syncthreads();
    NET_TICK(4);
    // the waves' shares of the head sums: rows of 16 floats [wave][output], inside the board positions of halo row t + 1 of c1
    constexpr int kShare = 6 * 64;   // bytes of a wave's share of one board row
    {   // conv3: 64 -> 128, wave w = output channels 32 w .. 32 w + 31; its ReLU'd output feeds the two 1x1 head convolutions
        float vals[96];   // [row t][output o]: the lane's 8 channels of position (t, n)
        {
            f32x4 acc[2][NT];
            rt::conv_r<64, 2, NT, F8>(c2, t3p, lane, a3, acc);
            NET_TICK(5);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                f32x2 v2[3] = {f32x2{0.0f, 0.0f}, f32x2{0.0f, 0.0f}, f32x2{0.0f, 0.0f}};
                if (t < NT) {
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float hv = fmaxf(fmaf(acc[m][t < NT ? t : 0][j], k3, b3r[m][j]), 0.0f);
#pragma unroll
                            for (int o2 = 0; o2 < 3; ++o2) {
                                const int e = 6 * j + 2 * o2;  // float index of (channel j, outputs 2 o2, 2 o2 + 1)
                                v2[o2] = __builtin_elementwise_fma(f32x2{hwr[m][e >> 2][e & 3], hwr[m][e >> 2][(e & 3) + 1]},
                                                                   f32x2{hv, hv}, v2[o2]);
                            }
                        }
                }
#pragma unroll
                for (int o = 0; o < 6; ++o) vals[t * 6 + o] = v2[o >> 1][o & 1];
            }
        }
        // sum over the 4 k blocks (lanes n, n + 16, n + 32, n + 48); lane group g is left with rows 8 (g & 1) + 4 (g >> 1) + 0 .. 3
        float mine[24];
        f4::reduce_scatter_96(vals, mine);
        const int t0 = 8 * (g & 1) + 4 * (g >> 1);
        float *share = reinterpret_cast<float *>(c1 + ((t0 + 1) * kRowW + 1) * P1 + wave * kShare) + n;
#pragma unroll
        for (int i = 0; i < 24; ++i)
            if (t0 + i / 6 < NT) share[(i / 6) * (kRowW * P1 / 4) + (i % 6) * 16] = mine[i];
    }
    NET_TICK(6);
    __syncthreads();
    {
        float *dst = feat ? feat + (size_t)board * nd.feat_ld : nullptr;  // null: only the f16 pieces are wanted
        _Float16 *dst16 = feat16 ? feat16 + ((size_t)(board >> 5) * (nd.groups_act + nd.groups_val) * 1024 + (board & 31) * 16)
                                 : nullptr;
        const bool deferred = RES || later.slot_of != nullptr;
        float *vdst = nullptr;
        if constexpr (RES) {   // the game's slot advances by one per simulation (expand_backup_body<DEF>); the value inputs stay in LDS
            dst16 = res_slot0 + sim < later.n_slots ? feat16 + (size_t)(res_slot0 + sim) * later.slot_halfs + (size_t)(board >> 5) * nd.groups_act * 1024 + (board & 31) * 16 : nullptr;
            vdst = res_vrow;
        } else if (deferred) {   // the policy pieces wait in the store (tiles of groups_act K-steps), the value inputs go on as f32
            const int slot_ = later.slot_of[board];   // (uniform; beyond the store: nothing is written, expand_backup_body<DEF> flags the game)
            dst16 = slot_ < later.n_slots ? feat16 + (size_t)slot_ * later.slot_halfs + (size_t)(board >> 5) * nd.groups_act * 1024 + (board & 31) * 16 : nullptr;
            vdst = later.valfeat + (size_t)board * later.vf_ld;
        }
        if (tid < S) {
            const float *share = reinterpret_cast<const float *>(c1 + ((cell_y + 1) * kRowW + 1) * P1) + cell_x;
            float vsum[6];
#pragma unroll
            for (int o = 0; o < 6; ++o) {
                float v = share[o * 16];
#pragma unroll
                for (int w = 1; w < 4; ++w) v += share[w * (kShare / 4) + o * 16];
                vsum[o] = v;
            }
            const int cell = tid;
#pragma unroll
            for (int o = 0; o < 6; ++o) {
                const float v = fmaxf(vsum[o] + hb[o], 0.0f);
                if (dst) dst[(o < 4 ? o * S : nd.feat_val_off + (o - 4) * S) + cell] = v;
                if (deferred && o >= 4) {
                    vdst[(o - 4) * S + cell] = v;
                } else if (dst16) {
                    const int k = (o < 4 ? o : o - 4) * S + cell;
                    const int step = (o < 4 ? 0 : nd.groups_act) + (k >> 4);
                    const float z = v * act3;
                    const _Float16 zh = (_Float16)z;
                    if constexpr (!BITS) zmax = fmaxf(zmax, z);
                    _Float16 *q = dst16 + (size_t)step * 1024 + (k & 15);
                    q[0] = zh;
                    q[512] = (_Float16)(z - (float)zh);
                }
            }
        }
    }
    NET_TICK(7);
    __syncthreads();   // the shares are read: the next board's conv1 may overwrite them
    NET_TICK(8);
    if constexpr (RES) {
        // ---- the rest of the simulation, by the same workgroup (k_tree_step_def's body: rz_tree.h): the value head's first layer
        // by K-quarters from the row in LDS, then wave 0 -- the game's wave -- finishes the value, reserves the prior block, backs
        // up and selects the next leaf, which comes back through LDS
        const int game = blockIdx.x;
        if (res.vh.groups == 128) rzt::value_quarter_lds<16>(res.vh, res_vrow, lane, wave, res_part);
        else rzt::value_quarter_lds<8>(res.vh, res_vrow, lane, wave, res_part);
        NET_TICK(16);
        if (wave == 0) rzt::expand_backup_body<float, false, false, false, true>(res.E, nullptr, nullptr, game, lane, rz_raw_heads(), 0, res.vh, res_part);
        else __syncthreads();   // (the barrier inside the body, where the quarters meet)
        __syncthreads();        // the tree's updates before the selection's loads
        NET_TICK(17);
        const bool more = sim + 1 < res_sims(res);
        if (wave == 0 && more) rzt::select_body<false>(res.E, nullptr, game, lane, 0, res_leaf);
        __syncthreads();
        NET_TICK(18);
        if (more) {   // the planes of the next leaf, from LDS: what load_bits forms from the leaf arrays
            planes_from_lds(tid);
            // conv1's weight fragments again (6 KB, L2-resident; their latency passes under the barrier below): carried in
            // registers ACROSS the tree code above they cost the resident kernels a spill that was reloaded in every simulation
            asm volatile("" ::: "memory");
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int p_ = 0; p_ < 2; ++p_) a1[ky][p_] = __builtin_bit_cast(sp::f16x8, nd.s1[(ky * 2 + p_) * 64 + lane]);
            store_obs();
            __syncthreads();
        }
        NET_TICK(19);
    }
    }  // boards
#ifdef RZ_NET_PROFILE
    if (blockIdx.x == 0 && tid0 == 0) {
        for (int i = 0; i < 24; ++i) net_prof[i] = prof_acc[i];
        net_prof[10] = __builtin_readcyclecounter() - prof_k0;
    }
#endif
    if constexpr (!BITS)
        if (!(zmax <= 65504.0f)) atomicOr(flags, (unsigned)RZ_NET_FLAG_F16_RANGE);
    if (TRACE && tid0 == 0 && (int)blockIdx.x < n_boards)
        rz_trace_write(later.trace, RZ_TRACE_TRUNK, later.slot_of ? later.slot_of[blockIdx.x] : 0, blockIdx.x, trace_t0);
}

template <int NT, bool BITS, bool TRACE = false, bool F8 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(200))) void k_trunk_rows(NetDev nd, const float *__restrict__ obs, LeafBits leaves,
                                                    float *__restrict__ feat, _Float16 *__restrict__ feat16,
                                                    int n_boards, unsigned *__restrict__ flags, DeferredOut later) {
    trunk_rows_body<NT, BITS, TRACE, false, F8>(nd, obs, leaves, feat, feat16, n_boards, flags, later, ResArgs<false>{});
}

template <int NT, bool F8 = false>
__global__ __launch_bounds__(256) void k_trunk_rows_res(NetDev nd, LeafBits leaves, _Float16 *__restrict__ feat16, int n_boards,
                                                        unsigned *__restrict__ flags, DeferredOut later, ResArgs<true> res) {
    trunk_rows_body<NT, true, false, true, F8>(nd, nullptr, leaves, nullptr, feat16, n_boards, flags, later, res);
}

// Direct path: wave w = 4*rh + q4 owns output-channel quarter q4 (the two waves of a quarter share
// a SIMD, waves are dealt to SIMDs cyclically) and row half rh (rows 0-7 / 8-15; on a 15x15 board
// the second half computes 7 rows, so every SIMD carries exactly 15 row-units of each layer).
__global__ __launch_bounds__(kTrunkThreads) void k_trunk(NetDev nd, const float *__restrict__ obs,
                                                         float *__restrict__ feat, int n_boards) {
    constexpr int PL = kPlaneDirect;
    constexpr int kLdsFloats = kPlanes * PL;  // 131.25 KiB
    __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
    float *in0 = lds;
    float *c1 = in0 + kPlanesIn * PL;
    float *c2 = c1 + kPlanesC1 * PL;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q4 = wave & 3, rh = wave >> 2;
    const int BH = nd.BH, BW = nd.BW, S = nd.S;
    const int board = blockIdx.x;
    if (board >= n_boards) return;
    const int row0 = 8 * rh;
    const int n_rows = (BH - row0 == 7) ? 7 : 8;
    const bool busy = row0 < BH;  // a wave whose rows are all outside the board only hits barriers

    // zero the halo planes (interiors are overwritten below), then stage the observation
    {
        f32x4 *z = reinterpret_cast<f32x4 *>(lds);
        for (int i = tid; i < kLdsFloats / 4; i += kTrunkThreads) z[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    {
        const float *src = obs + (size_t)board * 4 * S;
        for (int i = tid; i < 4 * S; i += kTrunkThreads) {
            const int c = i / S, r = i - c * S, y = r / BW, x = r - y * BW;
            in0[c * PL + (y + 1) * kRowW + (x + 1)] = src[i];
        }
    }
    __syncthreads();

    if (busy && q4 < 2) {   // conv1: 4 -> 32 = two 16-channel tiles
        f32x4 acc[1][8];
        zero_acc<1>(acc);
        conv_rows<PL, 4, 1>(in0, nd.w1, q4, row0, n_rows, lane, acc);
        store_relu<PL, 1>(c1, nd.b1, q4, row0, lane, BH, BW, acc);
    }
    __syncthreads();
    if (busy) {   // conv2: 32 -> 64 = one tile per quarter
        f32x4 acc[1][8];
        zero_acc<1>(acc);
        conv_rows<PL, 32, 1>(c1, nd.w2, q4, row0, n_rows, lane, acc);
        store_relu<PL, 1>(c2, nd.b2, q4, row0, lane, BH, BW, acc);
    }
    __syncthreads();
    // conv3: 64 -> 128 (two tiles per quarter), kept in registers and fed to the 1x1 head convs
    float part[8][6];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int o = 0; o < 6; ++o) part[t][o] = 0.0f;
    if (busy) {
        const int q = lane >> 4;
        f32x4 acc[2][8];
        zero_acc<2>(acc);
        conv_rows<PL, 64, 2>(c2, nd.w3, 2 * q4, row0, n_rows, lane, acc);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int c0 = (2 * q4 + m) * 16 + 4 * q;
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(nd.b3 + c0);
            f32x4 wv[6];
#pragma unroll
            for (int o = 0; o < 6; ++o) wv[o] = *reinterpret_cast<const f32x4 *>(nd.wh + o * 128 + c0);
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float hv = fmaxf(acc[m][t][j] + bv[j], 0.0f);
#pragma unroll
                    for (int o = 0; o < 6; ++o) part[t][o] = fmaf(wv[o][j], hv, part[t][o]);
                }
        }
    }
    // sum over the 4 channel sub-groups held by lanes x, x+16, x+32, x+48
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int o = 0; o < 6; ++o) {
            float v = part[t][o];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            part[t][o] = v;
        }
    float *partial = c1;  // [q4][o][y][x]: c1 is free now
    if (lane < 16) {
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int o = 0; o < 6; ++o) partial[((q4 * 6 + o) * 16 + (row0 + t)) * 16 + lane] = part[t][o];
    }
    __syncthreads();
    {
        float *dst = feat + (size_t)board * nd.feat_ld;
        for (int i = tid; i < 6 * S; i += kTrunkThreads) {
            const int o = i / S, r = i - o * S, y = r / BW, x = r - y * BW;
            float v = nd.bh[o];
#pragma unroll
            for (int k = 0; k < 4; ++k) v += partial[((k * 6 + o) * 16 + y) * 16 + x];
            dst[i < 4 * S ? i : i - 4 * S + nd.feat_val_off] = fmaxf(v, 0.0f);
        }
    }
}

// ------------------------------------------------------------------ heads (FC layers)
// k_heads_gemm: the two first FC layers as ONE fp32-MFMA GEMM: M = boards, N = outputs (policy
// logits padded to a multiple of 32, then the 64 hidden units of the value head), K = 4S (policy) /
// 2S (value), both padded to multiples of 16.  One workgroup per (32 boards) x (32 outputs); its 4
// waves split K and each accumulates the whole 2 x 2 block of 16 x 16 tiles over its slice, so a
// fragment pair feeds four independent MFMA chains.  Operands stream straight from L2 in their
// natural row-major layouts: lane (row r, quarter q) loads 16 bytes = k 16g + 4q .. + 3 of its row,
// the 4 values being the lane's operand of the 4 MFMA steps of group g (the order in which K is
// consumed is free as long as both operands agree).  A ring of kHeadDepth groups is kept in flight.
// The 4 partial blocks are summed through LDS.  k_heads_finish: log_softmax / fc2 + tanh.
constexpr int kHeadDepth = 4;
constexpr int kHeadWaves = 4;  // K split

__global__ __launch_bounds__(64 * kHeadWaves) void k_heads_gemm(NetDev nd, const float *__restrict__ feat,
                                                    float *__restrict__ raw, float *__restrict__ hid,
                                                    int n_boards) {
    __shared__ f32x4 part[kHeadWaves][4][64];
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b0 = blockIdx.x * 32;
    const int n_act_tiles = nd.Npad / 32;
    const int ot = blockIdx.y;
    const bool is_act = ot < n_act_tiles;
    const int m = lane & 15, kq = lane >> 4;
    const int groups = is_act ? nd.groups_act : nd.groups_val;
    const int g0 = wave * groups / kHeadWaves, g1 = (wave + 1) * groups / kHeadWaves;
    const size_t ldw = (size_t)16 * groups, lda = (size_t)nd.feat_ld;
    const int n0 = 32 * (is_act ? ot : ot - n_act_tiles);
    const float *w = (is_act ? nd.fc_act_w : nd.fc_val1_w) + (size_t)(n0 + m) * ldw + 4 * kq;
    const float *a = feat + (size_t)(b0 + m) * lda + (is_act ? 0 : nd.feat_val_off) + 4 * kq;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 ra[kHeadDepth][2], rw[kHeadDepth][2];
#pragma unroll
    for (int d = 0; d < kHeadDepth; ++d) {
        const int g = g0 + d;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ra[d][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            rw[d][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (g < g1) {
                ra[d][i] = *reinterpret_cast<const f32x4 *>(a + (size_t)(16 * i) * lda + 16 * g);
                rw[d][i] = *reinterpret_cast<const f32x4 *>(w + (size_t)(16 * i) * ldw + 16 * g);
            }
        }
    }
    for (int g = g0; g < g1; g += kHeadDepth) {
#pragma unroll
        for (int d = 0; d < kHeadDepth; ++d) {
            if (g + d >= g1) break;
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[d][i][u], rw[d][j][u], acc[i][j], 0, 0, 0);
            const int gn = g + d + kHeadDepth;
            if (gn < g1) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ra[d][i] = *reinterpret_cast<const f32x4 *>(a + (size_t)(16 * i) * lda + 16 * gn);
                    rw[d][i] = *reinterpret_cast<const f32x4 *>(w + (size_t)(16 * i) * ldw + 16 * gn);
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) part[wave][2 * i + j][lane] = acc[i][j];
    __syncthreads();
    if (wave >= 4) return;
    // wave t finishes tile t = 2 i + j.  D: column = output (lane & 15), rows = boards 4 * (lane >> 4) + e
    const int ti = wave >> 1, tj = wave & 1;
    f32x4 v = part[0][wave][lane];
#pragma unroll
    for (int k = 1; k < kHeadWaves; ++k) {
        const f32x4 p = part[k][wave][lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += p[e];
    }
    const int col = n0 + 16 * tj + m;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int b = b0 + 16 * ti + 4 * kq + e;
        if (b >= n_boards) continue;
        if (is_act) raw[(size_t)b * nd.Npad + col] = v[e] + nd.fc_act_b[col];
        else hid[(size_t)b * 64 + col] = fmaxf(v[e] + nd.fc_val1_b[col], 0.0f);
    }
}

// k_heads_split: the same two FC layers on the f16 matrix pipe, operands as hi + lo f16 pairs (three
// v_mfma_f32_32x32x16_f16 per product, f32 accumulation -- the arithmetic of k_trunk_split, which also writes the
// features as f16 pieces in A-fragment order; weights packed by pack_split_fc).  M = boards, N = outputs, so a
// lane's accumulator registers are boards of ONE output column and the stores of a tile row are 128 contiguous
// bytes.  A workgroup = TM 32-board tiles x TN policy N-tiles (group blockIdx.y) or x one of the two value N-tiles;
// its 4 waves split K into quarters (policy: 4S/16 steps, value: 2S/16) and every wave carries the whole block, so
// a fragment it loads (1 KB, one 16-byte load per lane, fully coalesced) feeds 3*TN or 3*TM MFMAs; DEPTH K-steps
// are in flight per wave.  The four partial blocks are summed through LDS in wave order.  The K quarters and the
// order of the sums do not depend on the shape, so every instantiation gives the same bits:
//   <2, 4, 3, true>   64 boards x half of the policy outputs AND one value tile per workgroup (policy first, then
//                     the value tile in the same LDS): 22 workgroups for 672 boards -- for the 32 CUs a capped
//                     trunk leaves free; the loads stay below the ~64 B/clk of a CU's vector memory path
//   <1, 2, 5, false>  32 boards x 2 policy tiles, the value tiles in workgroups of their own: 96 small workgroups
//                     for 512 boards, deep prefetch -- for the whole chip (the kernel is load-latency bound)
template <int TM, int TN>
struct FsFrags {
    sp::f16x8 a[TM][2], b[TN][2];
};

// Fragments of K-step `step` (A: TM feature tiles, steps_a K-steps apart; B: TN weight tiles at fb[n]).  Every load
// is unconditional -- a load under a branch makes hipcc wait for ALL outstanding loads (vmcnt(0)) before each use,
// which serialises the ring: a step past the end of the wave's K range re-reads the last feature step against the
// all-zero weight fragment `zero` (pack_split_fc appends one), and so does an N-tile past the last output tile.
template <int TM, int TN>
__device__ __forceinline__ void fs_load(FsFrags<TM, TN> &f, const f32x4 *__restrict__ fa, const f32x4 *const (&fb)[TN],
                                        const f32x4 *__restrict__ zero, int steps_a, int a_step0, int step, int k1, int lane) {
    const bool live = step < k1;
    const int sa = a_step0 + (live ? step : k1 - 1);
    const int lane_a = ((lane & 31) << 1) | (lane >> 5);  // features: [board % 32][k / 8 % 2] x 8 f16 (k_trunk_split)
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int p = 0; p < 2; ++p)
            f.a[m][p] = __builtin_bit_cast(sp::f16x8, fa[(((size_t)m * steps_a + sa) * 2 + p) * 64 + lane_a]);
#pragma unroll
    for (int n = 0; n < TN; ++n) {
        // (a tile that does not exist has fb[n] == zero: every step of it reads the one zero fragment)
        const f32x4 *src = live && fb[n] != zero ? fb[n] + (size_t)step * 128 : zero;
#pragma unroll
        for (int p = 0; p < 2; ++p) f.b[n][p] = __builtin_bit_cast(sp::f16x8, src[p * 64 + lane]);
    }
}

// acc[m][n] += sum over K-steps [k0, k1) of A-tile m (features) x B-tile n (weights, fb[n] -> its step 0, or the zero
// fragment for a tile that does not exist)
template <int TM, int TN, int DEPTH>
__device__ __forceinline__ void fs_gemm(sp::f32x16 (&acc)[TM][TN], const f32x4 *__restrict__ fa, const f32x4 *const (&fb)[TN],
                                        const f32x4 *__restrict__ zero, int steps_a, int a_step0, int k0, int k1, int lane) {
    if (k0 >= k1) return;
    FsFrags<TM, TN> ring[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) fs_load<TM, TN>(ring[d], fa, fb, zero, steps_a, a_step0, k0 + d, k1, lane);
    __builtin_amdgcn_sched_barrier(0);
    for (int k = k0; k < k1; k += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int combo = 0; combo < 3; ++combo)
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int n = 0; n < TN; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ring[d].a[m][combo == 2], ring[d].b[n][combo == 1],
                                                                          acc[m][n], 0, 0, 0);
            fs_load<TM, TN>(ring[d], fa, fb, zero, steps_a, a_step0, k + d + DEPTH, k1, lane);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// PAIRED (the GEMM over the deferred store: policy outputs only, hundreds of MB of features read once per move): a 1-D grid in
// which the workgroups of the SAME board tiles and different output groups are 8 apart -- workgroups go to the 8 XCDs round robin,
// so the two share an L2 and run together: the second one's feature reads hit there instead of HBM (a (tiles, groups) grid
// dispatches them thousands of workgroups apart: the store came from HBM twice, profiles/r04/pmc_traffic.json).
template <int TM, int TN, int DEPTH, bool VAL_FUSED, bool PAIRED = false>
__global__ __launch_bounds__(256) void k_heads_split(NetDev nd, const f32x4 *__restrict__ feat16,
                                                     float *__restrict__ raw, float *__restrict__ hid, int n_boards) {
    __shared__ sp::f32x16 part[4][TM * TN][64];  // [K quarter][tile][lane]
    const int n_act_tiles = nd.Npad / 32, n_groups = (n_act_tiles + TN - 1) / TN;
    int block_x = blockIdx.x, block_y = blockIdx.y;
    if constexpr (PAIRED) {
        static_assert(!VAL_FUSED, "policy outputs only");
        if (n_groups == 2) {   // (at most 256 outputs = 8 tiles = 2 groups of TN = 4)
            block_y = (block_x >> 3) & 1;
            block_x = ((block_x >> 4) << 3) | (block_x & 7);
        } else {
            block_y = 0;
        }
        if (block_x * TM * 32 >= n_boards) return;   // (the grid is rounded up to whole groups of 16; uniform, before any barrier)
    }
    __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mt0 = block_x * TM;
    const int steps_all = nd.groups_act + nd.groups_val;
    const f32x4 *fa = feat16 + (size_t)mt0 * steps_all * 128;  // 128 f32x4 = one K-step (hi | lo) of one tile
    const f32x4 *zero = nd.fs_act + (size_t)n_act_tiles * nd.groups_act * 128;  // one all-zero K-step behind the weights
    const int col = lane & 31, h = lane >> 5;
    const int group = block_y;                                       // policy outputs 32 * TN * group ..
    const int vtile = VAL_FUSED ? block_y : block_y - n_groups;  // value hidden units 32 * vtile ..
    if (group < n_groups) {
        const f32x4 *fb[TN];
#pragma unroll
        for (int n = 0; n < TN; ++n)
            fb[n] = TN * group + n < n_act_tiles ? nd.fs_act + (size_t)(TN * group + n) * nd.groups_act * 128 : zero;
        sp::f32x16 acc[TM][TN];
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;
        const int K = nd.groups_act, k0 = wave * K / 4, k1 = (wave + 1) * K / 4;
        fs_gemm<TM, TN, DEPTH>(acc, fa, fb, zero, steps_all, 0, k0, k1, lane);
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n) part[wave][m * TN + n][lane] = acc[m][n];
        __syncthreads();
        const float scale = nd.s_inv[3];
        // wave w finishes tiles w, w + 4, ...: D column = output (lane & 31), rows = boards 8g + 4h + j
#pragma unroll
        for (int t = wave; t < TM * TN; t += 4) {
            const int m = t / TN, n = t % TN;
            if (TN * group + n >= n_act_tiles) continue;
            sp::f32x16 v = part[0][t][lane];
#pragma unroll
            for (int q = 1; q < 4; ++q) {
                const sp::f32x16 pq = part[q][t][lane];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] += pq[r];
            }
            const int c = 32 * (TN * group + n) + col;
            const float bias = nd.fc_act_b[c];
            // raw / hid have rows for whole 64-board tiles (rz_net_reserve): the stores need no bounds test -- under a
            // branch each one would wait for the previous store to be acknowledged (vmcnt(0) per basic block)
            float *dst = raw + (size_t)(32 * (mt0 + m) + 4 * h) * nd.Npad + c;
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[(size_t)(8 * (r >> 2) + (r & 3)) * nd.Npad] = fmaf(v[r], scale, bias);
        }
        if (VAL_FUSED) __syncthreads();
    }
    if (vtile >= 0 && vtile < 2) {
        const f32x4 *fb[1] = {nd.fs_val + (size_t)vtile * nd.groups_val * 128};
        sp::f32x16 acc[TM][1];
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][0][r] = 0.0f;
        const int K = nd.groups_val, k0 = wave * K / 4, k1 = (wave + 1) * K / 4;
        fs_gemm<TM, 1, DEPTH>(acc, fa, fb, zero, steps_all, nd.groups_act, k0, k1, lane);
#pragma unroll
        for (int m = 0; m < TM; ++m) part[wave][m][lane] = acc[m][0];
        __syncthreads();
        if (wave < TM) {
            const int m = wave;
            sp::f32x16 v = part[0][m][lane];
#pragma unroll
            for (int q = 1; q < 4; ++q) {
                const sp::f32x16 pq = part[q][m][lane];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] += pq[r];
            }
            const float scale = nd.s_inv[4];
            const int c = 32 * vtile + col;
            const float bias = nd.fc_val1_b[c];
            float *dst = hid + (size_t)(32 * (mt0 + m) + 4 * h) * 64 + c;
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[(8 * (r >> 2) + (r & 3)) * 64] = fmaxf(fmaf(v[r], scale, bias), 0.0f);
        }
    }
}

// k_heads_part: the arithmetic of k_heads_split with the reduction left to the consumer.  One single-wave workgroup
// per (32-board tile, 32-output tile, K quarter): no LDS, no barrier, ~170 registers -- a wave fits on a SIMD beside a
// wave of a resident k_trunk_split workgroup (344 registers of 512, 151 KB of LDS), so with two lanes of games this
// GEMM runs UNDER the other lane's trunk on all CUs instead of waiting for it (or for CUs reserved for it).  Part q of
// tile (m, n) goes to raw + q * raw_stride (policy) / hid + q * hid_stride (value) un-scaled; the consumer adds the four
// parts in the order k_heads_split does, ((p0 + p1) + p2) + p3, then fmaf(sum, scale, bias): the same bits.
// blockIdx = (board tile, output tile: policy tiles then the two value tiles, K quarter).
template <int DEPTH>
__global__ __launch_bounds__(64) void k_heads_part(NetDev nd, const f32x4 *__restrict__ feat16, float *__restrict__ raw,
                                                   float *__restrict__ hid, long long raw_stride, long long hid_stride) {
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x;
    const int mt = blockIdx.x, q = blockIdx.z;
    const int steps_all = nd.groups_act + nd.groups_val;
    const int n_act_tiles = nd.Npad / 32;
    const f32x4 *fa = feat16 + (size_t)mt * steps_all * 128;
    const f32x4 *zero = nd.fs_act + (size_t)n_act_tiles * nd.groups_act * 128;
    const int col = lane & 31, h = lane >> 5;
    const int tile = blockIdx.y;
    const bool is_act = tile < n_act_tiles;
    const int vtile = tile - n_act_tiles;
    const f32x4 *fb[1] = {is_act ? nd.fs_act + (size_t)tile * nd.groups_act * 128 : nd.fs_val + (size_t)vtile * nd.groups_val * 128};
    sp::f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.0f;
    const int K = is_act ? nd.groups_act : nd.groups_val, k0 = q * K / 4, k1 = (q + 1) * K / 4;
    fs_gemm<1, 1, DEPTH>(acc, fa, fb, zero, steps_all, is_act ? 0 : nd.groups_act, k0, k1, lane);
    // rows for whole 64-board tiles exist in every part (rz_net_reserve): unconditional stores
    float *dst = is_act ? raw + (size_t)q * raw_stride + (size_t)(32 * mt + 4 * h) * nd.Npad + 32 * tile + col
                        : hid + (size_t)q * hid_stride + (size_t)(32 * mt + 4 * h) * 64 + 32 * vtile + col;
    const size_t ld = is_act ? (size_t)nd.Npad : (size_t)64;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[(size_t)(8 * (r >> 2) + (r & 3)) * ld] = acc[0][0][r];
}

__global__ __launch_bounds__(64) void k_heads_finish(NetDev nd, const float *__restrict__ raw,
                                                     const float *__restrict__ hid, float *__restrict__ logp,
                                                     float *__restrict__ value, int n_boards, int n_parts,
                                                     long long raw_stride, long long hid_stride) {
    __builtin_amdgcn_s_setprio(3);
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= n_boards) return;
    const int S = nd.A;  // number of policy outputs
    const float *r = raw + (size_t)b * nd.Npad;
    const float act_scale = nd.s_inv[3], val_scale = nd.s_inv[4];
    float v[4];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = lane + 64 * i;
        v[i] = -INFINITY;
        if (j < S) {
            if (n_parts == 4)  // k_heads_part left the four K-quarter sums: finish them as k_heads_split does
                v[i] = fmaf(((r[j] + r[j + raw_stride]) + r[j + 2 * raw_stride]) + r[j + 3 * raw_stride], act_scale, nd.fc_act_b[j]);
            else
                v[i] = r[j];
        }
        mx = fmaxf(mx, v[i]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) sum += (lane + 64 * i < S) ? expf(v[i] - mx) : 0.0f;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    const float lse = mx + logf(sum);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = lane + 64 * i;
        if (j < S) logp[(size_t)b * S + j] = v[i] - lse;
    }
    const float *hp = hid + (size_t)b * 64 + lane;
    float hv = hp[0];
    if (n_parts == 4)
        hv = fmaxf(fmaf(((hp[0] + hp[hid_stride]) + hp[2 * hid_stride]) + hp[3 * hid_stride], val_scale, nd.fc_val1_b[lane]), 0.0f);
    float h = hv * nd.fc_val2_w[lane];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) h += __shfl_xor(h, off);
    if (lane == 0) value[b] = tanhf(h + nd.fc_val2_b[0]);
}

}  // namespace

struct rz_net {
    int board_size = 0, device = 0;
    bool loaded = false;
    bool split_ok = true;        // rz_net_load found finite activation bounds: the split-f16 trunk cannot overflow
    float range_info[8] = {0};   // rz_net_range_info
    int algo = RZ_NET_SPLIT_F16;
    bool fp8_cross = false;      // RZ_NET_SPLIT_F16_FP8: algo stays RZ_NET_SPLIT_F16, conv3's cross terms run on the FP8 pipe (position-fed launches)
    int n_cus = 256;
    int max_wgs = 0;  // rz_net_set_max_workgroups: 0 = one persistent trunk workgroup per CU
    // small boards: k_trunk_split on the compact LDS grid, two workgroups per CU, for launches of more boards than HALF the CUs -- more
    // boards than CUs run in one round instead of two, and of two lanes with up to a CU's worth of boards each both trunks are on the
    // chip together (Connect4 512 games on two lanes 21.1 -> 21.8 M, 6 x 6 +7 %; four lanes of 128 boards: 3 % slower, hence the half).
    // RZ_NET_COMPACT=0: never, 2: whenever the geometry allows
    bool compact_grid = true;
    bool compact_always = false;
    NetDev dev;
    std::vector<void *> allocs;
    std::vector<size_t> alloc_bytes;
    size_t upload_cursor = 0;
    float *d_feat = nullptr, *d_raw = nullptr, *d_hid = nullptr;
    _Float16 *d_feat16 = nullptr;  // the features as hi + lo f16 pieces in fragment order (k_trunk_split -> k_heads_split)
    bool feat16_valid = false;     // the last trunk launch into the internal buffer wrote d_feat16
    bool feat32_valid = false;     // ... wrote d_feat (the split-f16 trunk skips it when the GEMM reads the f16 pieces)
    int heads_algo = RZ_NET_HEADS_AUTO;
    int raw_parts = 1;           // what the last launch_heads_gemm left in d_raw / d_hid: 1 = final, 4 = K-quarter sums
    bool raw_from_trunk = false; // the last trunk launch ran the FC layers itself (k_trunk_split, FC_HERE): d_raw / d_hid are final
    size_t raw_part_floats = 0, hid_part_floats = 0;  // stride between the parts
    unsigned *d_flags = nullptr;
    long long feat_boards = 0;
    size_t feat_floats = 0;
    // deferred priors (rz_net_deferred_reserve): the policy-feature store, the logits of a flush, the value head's inputs
    _Float16 *d_store16 = nullptr;
    float *d_store_raw = nullptr, *d_valfeat = nullptr;
    const float *d_w1t = nullptr;        // val_fc1.weight as [groups][64][4] (rz_value_head)
    int vf_groups = 0;                    // K / 4 of the value head's first layer, padded to a multiple of 4
    int store_slots = 0, store_tiles = 0; // slots x 32-board tiles per slot
    unsigned long long *d_trace = nullptr; // rz_net_trace_attach
    long long store_boards = 0;
    // receptive-field leaf evaluation (rz_delta.h; rz_net_delta_*): the base cache of `base_games` games
    dl::BaseHdr *d_base_hdr = nullptr;
    char *d_base_recs = nullptr;
    unsigned *d_delta_stats = nullptr;
    uint8_t *d_base_ones = nullptr;   // [base_games] of 1: the `active` flags of a caller that has none
    int base_games = 0;
    bool delta_resident = true;       // rz_net_delta_resident: rz_net_search_resident runs k_delta_res where the cache allows
};

namespace {

int net_fail(int code, const char *msg, const char *detail = "") {
    char buf[480];
    snprintf(buf, sizeof(buf), "%s%s", msg, detail);
    rz_set_error(buf);
    return code;
}

// Parameter buffers are allocated by the first rz_net_load and REUSED by later ones (same shapes, same
// order), so device pointers captured in hipGraphs stay valid across weight updates.
template <typename T>
int net_upload(rz_net *net, const std::vector<T> &host, const T **out) {
    void *p = nullptr;
    const size_t bytes = host.size() * sizeof(T);
    if (net->upload_cursor < net->allocs.size()) {
        if (net->alloc_bytes[net->upload_cursor] != bytes) return net_fail(RZ_ERR_ARG, "parameter size changed between loads");
        p = net->allocs[net->upload_cursor];
    } else {
        if (hipMalloc(&p, bytes) != hipSuccess) return net_fail(RZ_ERR_OOM, "hipMalloc failed (net)");
        net->allocs.push_back(p);
        net->alloc_bytes.push_back(bytes);
    }
    net->upload_cursor += 1;
    if (hipMemcpy(p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess)
        return net_fail(RZ_ERR_HIP, "hipMemcpy failed (net)");
    *out = (const T *)p;
    return RZ_OK;
}

// weight [cout][cin][3][3] -> [tile][cin_step][3][lane][4]: lane = kq*16 + m holds
// W[16*tile + m][4*step + kq][tap = 4*tg + e] (taps 9..11 are zero padding)
std::vector<f32x4> pack_conv(const float *w, int cout, int cin) {
    const int tiles = cout / 16, steps = cin / 4;
    std::vector<f32x4> out((size_t)tiles * steps * 3 * 64);
    for (int t = 0; t < tiles; ++t)
        for (int s = 0; s < steps; ++s)
            for (int tg = 0; tg < 3; ++tg)
                for (int lane = 0; lane < 64; ++lane) {
                    const int m = lane & 15, kq = lane >> 4;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    for (int e = 0; e < 4; ++e) {
                        const int tap = 4 * tg + e;
                        if (tap < 9) v[e] = w[((size_t)(16 * t + m) * cin + (4 * s + kq)) * 9 + tap];
                    }
                    out[(((size_t)t * steps + s) * 3 + tg) * 64 + lane] = v;
                }
    return out;
}

// U = G g G^T for F(4x4,3x3) (G: 6x3), packed [tile][pass][cin_step][3][64 lanes] x 4: lane = kq*16 + m
// holds components k = 4*j + e (j = 0..2) of pass p for U[16*tile + m][4*step + kq], component k =
// (transform row i' = rows[p][k / 6], column j' = k % 6).  fp64, rounded once.
std::vector<f32x4> pack_wino_f4(const float *w, int cout, int cin) {
    static const double G[6][3] = {{1.0 / 4, 0, 0},          {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                   {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6},  {0, 0, 1}};
    static const int rows[3][2] = {{1, 2}, {3, 4}, {0, 5}};
    const int tiles = cout / 16, steps = cin / 4;
    std::vector<f32x4> out((size_t)tiles * 3 * steps * 3 * 64);
    for (int t = 0; t < tiles; ++t)
        for (int s = 0; s < steps; ++s)
            for (int lane = 0; lane < 64; ++lane) {
                const int m = lane & 15, kq = lane >> 4;
                const float *g = w + ((size_t)(16 * t + m) * cin + (4 * s + kq)) * 9;
                double tmp[6][3], U[6][6];
                for (int i = 0; i < 6; ++i)
                    for (int c = 0; c < 3; ++c)
                        tmp[i][c] = G[i][0] * g[0 * 3 + c] + G[i][1] * g[1 * 3 + c] + G[i][2] * g[2 * 3 + c];
                for (int i = 0; i < 6; ++i)
                    for (int j = 0; j < 6; ++j) U[i][j] = tmp[i][0] * G[j][0] + tmp[i][1] * G[j][1] + tmp[i][2] * G[j][2];
                for (int p = 0; p < 3; ++p)
                    for (int j = 0; j < 3; ++j) {
                        f32x4 v;
                        for (int e = 0; e < 4; ++e) {
                            const int k = 4 * j + e;
                            v[e] = (float)U[rows[p][k / 6]][k % 6];
                        }
                        out[((((size_t)t * 3 + p) * steps + s) * 3 + j) * 64 + lane] = v;
                    }
            }
    return out;
}

// Split f16 weights (k_trunk_split): w * scale = hi + lo with scale = the power of two that brings the
// largest |w| of the layer into [2^13, 2^14).  Packed [tile of 32 cout][step = tap * chunks + chunk][piece][lane]
// x 8 f16: lane = h*32 + r holds W[32*tile + r][16*chunk + 8*h + j][tap], j = 0..7 (the A fragment of
// v_mfma_f32_32x32x16_f16).
std::vector<f32x4> pack_split(const float *w, int cout, int cin, float *scale_out) {
    float wmax = 0.0f;
    for (size_t i = 0; i < (size_t)cout * cin * 9; ++i) wmax = std::fmax(wmax, std::fabs(w[i]));
    int e = 0;
    if (wmax > 0.0f && std::isfinite(wmax)) {
        (void)std::frexp(wmax, &e);  // wmax = f * 2^e, f in [0.5, 1)
        e = 14 - e;                  // wmax * 2^e in [2^13, 2^14)
    }
    const float scale = std::ldexp(1.0f, e);
    *scale_out = scale;
    const int tiles = cout / 32, chunks = cin / 16, steps = 9 * chunks;
    std::vector<f32x4> out((size_t)tiles * steps * 2 * 64);
    _Float16 *o = reinterpret_cast<_Float16 *>(out.data());
    for (int t = 0; t < tiles; ++t)
        for (int tap = 0; tap < 9; ++tap)
            for (int c = 0; c < chunks; ++c)
                for (int lane = 0; lane < 64; ++lane) {
                    const int r = lane & 31, h = lane >> 5, s = tap * chunks + c;
                    for (int j = 0; j < 8; ++j) {
                        const float v = w[((size_t)(32 * t + r) * cin + (16 * c + 8 * h + j)) * 9 + tap] * scale;
                        const _Float16 hi = (_Float16)v;
                        const _Float16 lo = (_Float16)(v - (float)hi);
                        o[((((size_t)t * steps + s) * 2 + 0) * 64 + lane) * 8 + j] = hi;
                        o[((((size_t)t * steps + s) * 2 + 1) * 64 + lane) * 8 + j] = lo;
                    }
                }
    return out;
}

// The same weights for k_trunk_rows (scale as pack_split: the two kernels share the rescaling factors).  Packed
// [tile of 16 cout][step = tap * chunks + chunk of 32 cin][piece][lane] x 8 f16: lane = g*16 + r holds
// W[16*tile + r][32*chunk + 8*g + j][tap], j = 0..7 (the A fragment of v_mfma_f32_16x16x32_f16).
std::vector<f32x4> pack_rows(const float *w, int cout, int cin, float scale) {
    const int tiles = cout / 16, chunks = cin / 32, steps = 9 * chunks;
    std::vector<f32x4> out((size_t)tiles * steps * 2 * 64);
    _Float16 *o = reinterpret_cast<_Float16 *>(out.data());
    for (int t = 0; t < tiles; ++t)
        for (int tap = 0; tap < 9; ++tap)
            for (int c = 0; c < chunks; ++c)
                for (int lane = 0; lane < 64; ++lane) {
                    const int r = lane & 15, g = lane >> 4, s = tap * chunks + c;
                    for (int j = 0; j < 8; ++j) {
                        const float v = w[((size_t)(16 * t + r) * cin + (32 * c + 8 * g + j)) * 9 + tap] * scale;
                        const _Float16 hi = (_Float16)v;
                        o[((((size_t)t * steps + s) * 2 + 0) * 64 + lane) * 8 + j] = hi;
                        o[((((size_t)t * steps + s) * 2 + 1) * 64 + lane) * 8 + j] = (_Float16)(v - (float)hi);
                    }
                }
    return out;
}

// OCP e4m3fn (1.4.3, bias 7, no infinities, largest finite 448) of x, round to nearest even, saturating
unsigned char to_e4m3(float x) {
    const unsigned char sign = std::signbit(x) ? 0x80 : 0;
    double a = std::fabs((double)x);
    if (!(a == a)) return 0x7f;
    if (a >= 448.0) return sign | 0x7e;
    if (a < std::ldexp(1.0, -10)) return sign;   // below half the smallest subnormal (2^-9): zero
    int e = 0;
    (void)std::frexp(a, &e);   // a = f 2^e, f in [0.5, 1)
    int ex = e - 1;            // a = 1.m x 2^ex
    if (ex < -6) ex = -6;      // subnormals share the exponent of the smallest normal
    const double step = std::ldexp(1.0, ex - 3);
    double qv = std::nearbyint(a / step);   // (the default rounding mode: to nearest even)
    int m = (int)qv;   // 0 .. 16 in units of step
    if (ex == -6 && m < 8) return sign | (unsigned char)m;   // subnormal
    if (m == 16) { m = 8; ++ex; }
    if (ex > 8 || (ex == 8 && m - 8 > 6)) return sign | 0x7e;
    return sign | (unsigned char)(((ex + 7) << 3) | (m - 8));
}

// conv3 for the FP8 cross terms of k_trunk_rows (rt::slot_r, F8).  v = w * scale as in pack_rows (|v| < 2^14), hi = f16(v),
// lo = f16(v - hi).  Packed [tile of 16 cout][tap][part][half][lane] x 16 bytes with lane = g*16 + r:
//   part 0, half c: the hi f16 pieces of W[16 tile + r][32 c + 8 g + j][tap], j = 0..7 (pack_rows' hi fragment of chunk c);
//   part 1: the lane's 32 bytes of the K = 128 block of the scaled MFMA, input channels 16 g + j, j = 0..15:
//           half 0 = e4m3(lo * 2^5) (meets the activations' e5m2 value), half 1 = e4m3(hi * 2^-6) (meets e5m2((value - hi) 2^11));
//           with the block's scale 2^-5 both products come out in the units of hi x hi.  |lo| <= 4 and |hi| <= 2^14: 128 and 256 of 448.
std::vector<f32x4> pack_rows_f8(const float *w, int cout, int cin, float scale) {
    const int tiles = cout / 16;
    std::vector<f32x4> out((size_t)tiles * 9 * 2 * 2 * 64);
    unsigned char *o = reinterpret_cast<unsigned char *>(out.data());
    for (int t = 0; t < tiles; ++t)
        for (int tap = 0; tap < 9; ++tap)
            for (int lane = 0; lane < 64; ++lane) {
                const int r = lane & 15, g = lane >> 4;
                auto piece = [&](int ci, _Float16 *hi, _Float16 *lo) {
                    const float v = w[((size_t)(16 * t + r) * cin + ci) * 9 + tap] * scale;
                    *hi = (_Float16)v;
                    *lo = (_Float16)(v - (float)*hi);
                };
                const size_t base = ((size_t)t * 9 + tap) * 4 * 1024 + (size_t)lane * 16;
                for (int c = 0; c < 2; ++c)
                    for (int j = 0; j < 8; ++j) {
                        _Float16 hi, lo;
                        piece(32 * c + 8 * g + j, &hi, &lo);
                        memcpy(o + base + c * 1024 + j * 2, &hi, 2);
                    }
                for (int j = 0; j < 16; ++j) {
                    _Float16 hi, lo;
                    piece(16 * g + j, &hi, &lo);
                    o[base + 2048 + j] = to_e4m3((float)lo * 32.0f);
                    o[base + 3072 + j] = to_e4m3((float)hi * (1.0f / 64.0f));
                }
            }
    return out;
}

// conv1 (32 x 4 x 3 x 3) for k_trunk_split: K-step = kernel row ky, k = 4 * kx + plane for kx = 0..3 (kx = 3: zero
// padding); lane = h*32 + r holds k = 8*h .. 8*h + 7 of output channel r.  [ky][hi | lo][lane] x 8 f16.
std::vector<f32x4> pack_split1(const float *w, float *scale_out) {
    float wmax = 0.0f;
    for (int i = 0; i < 32 * 4 * 9; ++i) wmax = std::fmax(wmax, std::fabs(w[i]));
    int e = 0;
    if (wmax > 0.0f && std::isfinite(wmax)) {
        (void)std::frexp(wmax, &e);
        e = 14 - e;
    }
    const float scale = std::ldexp(1.0f, e);
    *scale_out = scale;
    std::vector<f32x4> out((size_t)3 * 2 * 64);
    _Float16 *o = reinterpret_cast<_Float16 *>(out.data());
    for (int ky = 0; ky < 3; ++ky)
        for (int lane = 0; lane < 64; ++lane) {
            const int r = lane & 31, h = lane >> 5;
            for (int j = 0; j < 8; ++j) {
                const int kx = 2 * h + (j >> 2), c = j & 3;
                const float v = kx < 3 ? w[((r * 4 + c) * 3 + ky) * 3 + kx] * scale : 0.0f;
                const _Float16 hi = (_Float16)v;
                o[(((size_t)ky * 2 + 0) * 64 + lane) * 8 + j] = hi;
                o[(((size_t)ky * 2 + 1) * 64 + lane) * 8 + j] = (_Float16)(v - (float)hi);
            }
        }
    return out;
}

// FC weights for k_heads_split: w [n_out][k_in] row-major * scale = hi + lo (scale as in pack_split), packed
// [32-output tile][K-step][hi | lo][lane] x 8 f16: lane = h*32 + c holds W[32*tile + c][16*step + 8*h + j], the B
// fragment of v_mfma_f32_32x32x16_f16; zero beyond n_out / k_in.
std::vector<f32x4> pack_split_fc(const float *w, int n_out, int k_in, int tiles, int steps, float *scale_out) {
    float wmax = 0.0f;
    for (size_t i = 0; i < (size_t)n_out * k_in; ++i) wmax = std::fmax(wmax, std::fabs(w[i]));
    int e = 0;
    if (wmax > 0.0f && std::isfinite(wmax)) {
        (void)std::frexp(wmax, &e);
        e = 14 - e;
    }
    const float scale = std::ldexp(1.0f, e);
    *scale_out = scale;
    std::vector<f32x4> out(((size_t)tiles * steps + 1) * 2 * 64, f32x4{0.f, 0.f, 0.f, 0.f});  // + one all-zero K-step
    _Float16 *o = reinterpret_cast<_Float16 *>(out.data());
    for (int t = 0; t < tiles; ++t)
        for (int st = 0; st < steps; ++st)
            for (int lane = 0; lane < 64; ++lane) {
                const int c = lane & 31, h = lane >> 5, row = 32 * t + c;
                for (int j = 0; j < 8; ++j) {
                    const int k = 16 * st + 8 * h + j;
                    const float v = (row < n_out && k < k_in) ? w[(size_t)row * k_in + k] * scale : 0.0f;
                    const _Float16 hi = (_Float16)v;
                    o[((((size_t)t * steps + st) * 2 + 0) * 64 + lane) * 8 + j] = hi;
                    o[((((size_t)t * steps + st) * 2 + 1) * 64 + lane) * 8 + j] = (_Float16)(v - (float)hi);
                }
            }
    return out;
}

}  // namespace

template <int NT>
static void launch_trunk_rows(bool bits, dim3 grid, hipStream_t stream, const NetDev &nd, const float *d_obs, LeafBits leaves, float *f32,
                              _Float16 *f16, int n_boards, unsigned *flags, DeferredOut later = DeferredOut{nullptr, 0, nullptr, 0, nullptr},
                              bool fp8 = false) {
    if (fp8 && bits) {   // (the schedule trace reads the default arithmetic's kernel)
        k_trunk_rows<NT, true, false, true><<<grid, dim3(256), 0, stream>>>(nd, d_obs, leaves, f32, f16, n_boards, flags, later);
        return;
    }
    if constexpr (NT == 15) {
        if (bits && later.trace) {
            k_trunk_rows<NT, true, true><<<grid, dim3(256), 0, stream>>>(nd, d_obs, leaves, f32, f16, n_boards, flags, later);
            return;
        }
    }
    if (bits) k_trunk_rows<NT, true><<<grid, dim3(256), 0, stream>>>(nd, d_obs, leaves, f32, f16, n_boards, flags, later);
    else k_trunk_rows<NT, false><<<grid, dim3(256), 0, stream>>>(nd, d_obs, leaves, f32, f16, n_boards, flags, later);
}

template <int NT>
static void launch_search_rows(dim3 grid, hipStream_t stream, const NetDev &nd, LeafBits leaves, _Float16 *store, int n_games, unsigned *flags,
                               DeferredOut later, const ResArgs<true> &res, bool fp8) {
    if (fp8) k_trunk_rows_res<NT, true><<<grid, dim3(256), 0, stream>>>(nd, leaves, store, n_games, flags, later, res);
    else k_trunk_rows_res<NT><<<grid, dim3(256), 0, stream>>>(nd, leaves, store, n_games, flags, later, res);
}

extern "C" {

int rz_net_create(int32_t height, int32_t width, int32_t n_actions, int32_t device, rz_net **out) {
    if (out == nullptr) return net_fail(RZ_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (height < 1 || height > RZ_MAX_BOARD_SIZE || width < 1 || width > RZ_MAX_BOARD_SIZE)
        return net_fail(RZ_ERR_ARG, "board dimensions out of range");
    if (n_actions < 1 || n_actions > RZ_MAX_BOARD_SIZE * RZ_MAX_BOARD_SIZE)
        return net_fail(RZ_ERR_ARG, "n_actions out of range");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev)
        return net_fail(RZ_ERR_ARG, "bad device ordinal");
    rz_net *net = new (std::nothrow) rz_net();
    if (!net) return net_fail(RZ_ERR_OOM, "host allocation failed");
    net->board_size = height;
    net->device = device;
    if (const char *v = getenv("RZ_NET_COMPACT")) {
        net->compact_grid = v[0] != '0';
        net->compact_always = v[0] == '2';
    }
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
            net->n_cus = prop.multiProcessorCount;
    }
    if (hipSetDevice(device) != hipSuccess || hipMalloc((void **)&net->d_flags, sizeof(unsigned)) != hipSuccess ||
        hipMemset(net->d_flags, 0, sizeof(unsigned)) != hipSuccess) {
        delete net;
        return net_fail(RZ_ERR_OOM, "hipMalloc failed (net flags)");
    }
    memset(&net->dev, 0, sizeof(net->dev));
    net->dev.BH = height;
    net->dev.BW = width;
    net->dev.S = height * width;
    net->dev.A = n_actions;
    {
        NetDev &D = net->dev;
        D.Npad = (D.A + 31) / 32 * 32;
        D.groups_act = (4 * D.S + 15) / 16;
        D.groups_val = (2 * D.S + 15) / 16;
        // N-tile geometry of k_trunk_split: rows x width tiles if four of them cover the board, else 2 x 16
        const int rows = 32 / D.BW < 16 ? 32 / D.BW : 16;  // rows + 2 halo rows stay inside the 18-row grid
        if (rows >= 1 && (D.BH + rows - 1) / rows <= 4) {
            D.tile_rows = rows;
            D.tile_cols = D.BW;
        } else {
            D.tile_rows = 2;
            D.tile_cols = 16;
        }
        D.tile_rcp = (65536 + D.tile_cols - 1) / D.tile_cols;  // (n * rcp) >> 16 == n / cols for n < 32
    }
    *out = net;
    return RZ_OK;
}

int rz_net_destroy(rz_net *net) {
    if (!net) return RZ_OK;
    (void)hipSetDevice(net->device);
    (void)hipDeviceSynchronize();
    for (void *p : net->allocs) (void)hipFree(p);
    if (net->d_feat) (void)hipFree(net->d_feat);
    if (net->d_feat16) (void)hipFree(net->d_feat16);
    if (net->d_raw) (void)hipFree(net->d_raw);
    if (net->d_hid) (void)hipFree(net->d_hid);
    if (net->d_flags) (void)hipFree(net->d_flags);
    if (net->d_store16) (void)hipFree(net->d_store16);
    if (net->d_store_raw) (void)hipFree(net->d_store_raw);
    if (net->d_valfeat) (void)hipFree(net->d_valfeat);
    if (net->d_base_hdr) (void)hipFree(net->d_base_hdr);
    if (net->d_base_recs) (void)hipFree(net->d_base_recs);
    if (net->d_delta_stats) (void)hipFree(net->d_delta_stats);
    if (net->d_base_ones) (void)hipFree(net->d_base_ones);
    delete net;
    return RZ_OK;
}

int rz_net_error_flags(rz_net *net, uint32_t *h_flags) {
    if (!net || !h_flags) return net_fail(RZ_ERR_ARG, "NULL argument");
    if (hipSetDevice(net->device) != hipSuccess) return net_fail(RZ_ERR_HIP, "hipSetDevice failed");
    unsigned v = 0;
    if (hipMemcpy(&v, net->d_flags, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess)
        return net_fail(RZ_ERR_HIP, "hipMemcpy failed (net flags)");
    if (v != 0 && hipMemset(net->d_flags, 0, sizeof(unsigned)) != hipSuccess)
        return net_fail(RZ_ERR_HIP, "hipMemset failed (net flags)");
    *h_flags = v;
    return RZ_OK;
}

int rz_net_load(rz_net *net, const float *const *h_params, int32_t n_params) {
    if (!net || !h_params) return net_fail(RZ_ERR_ARG, "NULL argument");
    if (n_params != 16) return net_fail(RZ_ERR_ARG, "expected the 16 tensors of PolicyValueNet.state_dict()");
    for (int i = 0; i < 16; ++i)
        if (!h_params[i]) return net_fail(RZ_ERR_ARG, "a parameter pointer is NULL");
    if (hipSetDevice(net->device) != hipSuccess) return net_fail(RZ_ERR_HIP, "hipSetDevice failed");
    (void)hipDeviceSynchronize();
    net->upload_cursor = 0;
    const int S = net->dev.S;
    NetDev &D = net->dev;
    int rc = RZ_OK;
    auto up_vec4 = [&](const std::vector<f32x4> &v, const f32x4 **dst) { if (rc == RZ_OK) rc = net_upload(net, v, dst); };
    auto up_f = [&](const float *src, size_t count, const float **dst) {
        if (rc == RZ_OK) rc = net_upload(net, std::vector<float>(src, src + count), dst);
    };
    // order of PolicyValueNet.state_dict(): conv1.w,b conv2.w,b conv3.w,b act_conv1.w,b
    // act_fc1.w,b val_conv1.w,b val_fc1.w,b val_fc2.w,b
    up_vec4(pack_conv(h_params[0], 32, 4), &D.w1);
    up_f(h_params[1], 32, &D.b1);
    up_vec4(pack_conv(h_params[2], 64, 32), &D.w2);
    up_f(h_params[3], 64, &D.b2);
    up_vec4(pack_conv(h_params[4], 128, 64), &D.w3);
    up_vec4(pack_wino_f4(h_params[2], 64, 32), &D.u2f);
    up_vec4(pack_wino_f4(h_params[4], 128, 64), &D.u3f);
    {
        float sw2 = 1.0f, sw3 = 1.0f;
        up_vec4(pack_split(h_params[2], 64, 32, &sw2), &D.s2);
        up_vec4(pack_split(h_params[4], 128, 64, &sw3), &D.s3);
        up_vec4(pack_rows(h_params[2], 64, 32, sw2), &D.t2);
        up_vec4(pack_rows(h_params[4], 128, 64, sw3), &D.t3);
        up_vec4(pack_rows_f8(h_params[4], 128, 64, sw3), &D.t3f);
        float sw1 = 1.0f;
        up_vec4(pack_split1(h_params[0], &sw1), &D.s1);
        float sfa = 1.0f, sfv = 1.0f;
        up_vec4(pack_split_fc(h_params[8], D.A, 4 * S, D.Npad / 32, D.groups_act, &sfa), &D.fs_act);
        up_vec4(pack_split_fc(h_params[12], 64, 2 * S, 2, D.groups_val, &sfv), &D.fs_val);
        // Activation bounds for observation planes in [0, 1] (the MCTS leaves: 0 / 1): a ReLU output is at most its
        // bias plus the positive weights times the bounds of their inputs.  Each layer's f16 pieces are stored times
        // the largest power of two <= 16 that keeps bound * scale below 60000, so NO activation of such an input can
        // leave the f16 range (the pieces of a value of size z carry an absolute error of max(2^-22 z, 2^-25): a
        // bound 1000x above the real activations still leaves the error below f32 rounding).  Without finite
        // bounds (inf / nan weights) the net runs on the exact-f32 direct trunk instead.
        double b1v[32], b2v[64], b3v[128], bfv[6];
        auto layer_bound = [](const float *w, const float *bias, int cout, int cin, int taps, const double *in, double *out) {
            double top = 0.0;
            for (int c = 0; c < cout; ++c) {
                double acc = bias[c] > 0.0f ? (double)bias[c] : 0.0;
                for (int i = 0; i < cin; ++i)
                    for (int t = 0; t < taps; ++t) {
                        const double wv = w[((size_t)c * cin + i) * taps + t];
                        if (wv > 0.0) acc += wv * (in ? in[i] : 1.0);
                        else if (!(wv <= 0.0)) acc = INFINITY;  // nan
                    }
                out[c] = acc;
                top = std::fmax(top, acc);
                if (!(acc >= 0.0)) top = INFINITY;
            }
            return top;
        };
        const double t1 = layer_bound(h_params[0], h_params[1], 32, 4, 9, nullptr, b1v);
        const double t2 = layer_bound(h_params[2], h_params[3], 64, 32, 9, b1v, b2v);
        (void)layer_bound(h_params[4], h_params[5], 128, 64, 9, b2v, b3v);
        double tf = layer_bound(h_params[6], h_params[7], 4, 128, 1, b3v, bfv);
        tf = std::fmax(tf, layer_bound(h_params[10], h_params[11], 2, 128, 1, b3v, bfv + 4));
        auto act_scale = [](double bound) {
            if (!(bound * sp::kMaxActScale >= sp::kF16Room)) return sp::kMaxActScale;  // also bound == 0
            int e = 0;
            (void)std::frexp(sp::kF16Room / bound, &e);   // kF16Room / bound = f * 2^e, f in [0.5, 1)
            return std::ldexp(1.0f, e - 1);                // the largest power of two <= kF16Room / bound
        };
        net->split_ok = std::isfinite(t1) && std::isfinite(t2) && std::isfinite(tf) && t1 < 1e30 && t2 < 1e30 && tf < 1e30;
        const float a1 = net->split_ok ? act_scale(t1) : sp::kMaxActScale, a2 = net->split_ok ? act_scale(t2) : sp::kMaxActScale,
                    a3 = net->split_ok ? act_scale(tf) : sp::kMaxActScale;
        const float info[8] = {(float)t1, (float)t2, (float)tf, a1, a2, a3, net->split_ok ? 1.0f : 0.0f, 0.0f};
        memcpy(net->range_info, info, sizeof(info));
        up_f(std::vector<float>{a2 / (a1 * sw2), 1.0f / (a2 * sw3), a1 / (sp::kObsScale * sw1),
                                1.0f / (a3 * sfa), 1.0f / (a3 * sfv), a1, a2, a3}.data(), 8, &D.s_inv);
    }
    up_f(h_params[5], 128, &D.b3);
    {
        std::vector<float> wh(6 * 128), bh(6);
        memcpy(wh.data(), h_params[6], 4 * 128 * sizeof(float));
        memcpy(wh.data() + 4 * 128, h_params[10], 2 * 128 * sizeof(float));
        memcpy(bh.data(), h_params[7], 4 * sizeof(float));
        memcpy(bh.data() + 4, h_params[11], 2 * sizeof(float));
        if (rc == RZ_OK) rc = net_upload(net, wh, &D.wh);
        std::vector<float> whp(128 * 6);
        for (int c = 0; c < 128; ++c)
            for (int o = 0; o < 6; ++o) whp[c * 6 + o] = wh[o * 128 + c];
        if (rc == RZ_OK) rc = net_upload(net, whp, &D.whp);
        if (rc == RZ_OK) rc = net_upload(net, bh, &D.bh);
    }
    {
        const size_t ld = (size_t)16 * D.groups_act;
        std::vector<float> t((size_t)D.Npad * ld, 0.0f), bias((size_t)D.Npad, 0.0f);
        for (int j = 0; j < D.A; ++j) {
            bias[j] = h_params[9][j];
            memcpy(&t[(size_t)j * ld], h_params[8] + (size_t)j * 4 * S, (size_t)4 * S * sizeof(float));
        }
        if (rc == RZ_OK) rc = net_upload(net, t, &D.fc_act_w);
        if (rc == RZ_OK) rc = net_upload(net, bias, &D.fc_act_b);
    }
    {
        const size_t ld = (size_t)16 * D.groups_val;
        std::vector<float> t((size_t)64 * ld, 0.0f);
        for (int j = 0; j < 64; ++j)
            memcpy(&t[(size_t)j * ld], h_params[12] + (size_t)j * 2 * S, (size_t)2 * S * sizeof(float));
        if (rc == RZ_OK) rc = net_upload(net, t, &D.fc_val1_w);
        up_f(h_params[13], 64, &D.fc_val1_b);
    }
    up_f(h_params[14], 64, &D.fc_val2_w);
    up_f(h_params[15], 1, &D.fc_val2_b);
    {   // the value head's first layer for the tree step of the deferred route: [group of 4 inputs][hidden unit][4]
        // four waves x two halves x PER groups of 4 inputs, PER = 2, 4, 8 or 16 (k_tree_step_def): 64 .. 512 inputs
        const int need = (2 * S + 3) / 4;
        net->vf_groups = need <= 16 ? 16 : need <= 32 ? 32 : need <= 64 ? 64 : 128;
        std::vector<float> t((size_t)net->vf_groups * 64 * 4, 0.0f);
        for (int j = 0; j < 64; ++j)
            for (int k = 0; k < 2 * S; ++k) t[((size_t)(k / 4) * 64 + j) * 4 + k % 4] = h_params[12][(size_t)j * 2 * S + k];
        if (rc == RZ_OK) rc = net_upload(net, t, &net->d_w1t);
    }
    net->loaded = rc == RZ_OK;
    return rc;
}

static int net_ready(rz_net *net, int32_t n) {
    if (!net) return net_fail(RZ_ERR_ARG, "net handle is NULL");
    if (!net->loaded) return net_fail(RZ_ERR_ARG, "rz_net_load has not been called");
    if (n < 0) return net_fail(RZ_ERR_ARG, "negative batch");
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) return net_fail(RZ_ERR_HIP, "hipGetDevice failed");
    if (cur != net->device && hipSetDevice(net->device) != hipSuccess) return net_fail(RZ_ERR_HIP, "hipSetDevice failed");
    return RZ_OK;
}

int rz_net_reserve(rz_net *net, int32_t max_boards) {
    int rc = net_ready(net, max_boards);
    if (rc != RZ_OK) return rc;
    if (max_boards <= net->feat_boards) return RZ_OK;
    (void)hipDeviceSynchronize();
    if (net->d_feat) (void)hipFree(net->d_feat);
    if (net->d_feat16) (void)hipFree(net->d_feat16);
    if (net->d_raw) (void)hipFree(net->d_raw);
    if (net->d_hid) (void)hipFree(net->d_hid);
    net->d_feat = net->d_raw = net->d_hid = nullptr;
    net->d_feat16 = nullptr;
    net->feat16_valid = net->feat32_valid = false;
    net->feat_boards = 0;
    // internal features: [boards padded to 32][16 * (groups_act + groups_val)], zero filled once
    const size_t pad_boards = ((size_t)max_boards + 31) / 32 * 32;
    net->feat_floats = pad_boards * 16 * (size_t)(net->dev.groups_act + net->dev.groups_val);
    if (hipMalloc((void **)&net->d_feat, net->feat_floats * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&net->d_raw, 4 * (((size_t)max_boards + 63) / 64 * 64) * net->dev.Npad * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&net->d_hid, 4 * (((size_t)max_boards + 63) / 64 * 64) * 64 * sizeof(float)) != hipSuccess)
        return net_fail(RZ_ERR_OOM, "hipMalloc failed (feature buffers)");
    // padded boards and the K tail must read as finite values (they meet zero weights)
    if (hipMemset(net->d_feat, 0, net->feat_floats * sizeof(float)) != hipSuccess)
        return net_fail(RZ_ERR_HIP, "hipMemset failed (feature buffer)");
    {   // f16 pieces: [boards padded to 64][K-steps][hi | lo][16 x f16]; the K tail and padded boards stay zero
        const size_t bytes = (((size_t)max_boards + 63) / 64 * 64) * (size_t)(net->dev.groups_act + net->dev.groups_val) * 64;
        if (hipMalloc((void **)&net->d_feat16, bytes) != hipSuccess)
            return net_fail(RZ_ERR_OOM, "hipMalloc failed (f16 feature buffer)");
        if (hipMemset(net->d_feat16, 0, bytes) != hipSuccess)
            return net_fail(RZ_ERR_HIP, "hipMemset failed (f16 feature buffer)");
    }
    net->raw_part_floats = (((size_t)max_boards + 63) / 64 * 64) * net->dev.Npad;  // room for four K-quarter parts
    net->hid_part_floats = (((size_t)max_boards + 63) / 64 * 64) * 64;
    net->feat_boards = max_boards;
    return RZ_OK;
}

// k_trunk_rows is instantiated for boards of 11 .. 16 rows (the 15 x 15 Gomoku board of the BASELINE configuration and its kin:
// the boards k_trunk_split needs more than four tiles for), 11 .. 16 columns wide (narrower ones leave too many of a row tile's
// 16 MFMA columns empty)
static bool rows_kernel_covers(int bh, int bw) { return bh >= 11 && bh <= 16 && bw >= 11 && bw <= 16; }

static void launch_trunk(rz_net *net, const float *d_obs, float *d_feat, int32_t n_boards, void *stream,
                         LeafBits leaves = LeafBits{nullptr, nullptr, nullptr}, DeferredOut later = DeferredOut{nullptr, 0, nullptr, 0, nullptr}) {
    const dim3 grid((unsigned)n_boards);
    // the internal buffer uses the padded layout of the FC GEMM, a caller's buffer the natural one
    const bool internal = d_feat == net->d_feat;
    // the split-f16 trunk writes what the FC GEMM behind it reads: the f16 pieces, and the f32 features only for a
    // caller's buffer or when the f32 GEMM is forced
    // a net whose weights give no finite activation bound (rz_net_load) never runs on the f16 pipe
    const bool split_algo = net->algo == RZ_NET_SPLIT_F16 || net->algo == RZ_NET_SPLIT_F16_TILES;
    const int algo = (split_algo && !net->split_ok) ? RZ_NET_DIRECT : (split_algo ? RZ_NET_SPLIT_F16 : net->algo);
    const bool split = algo == RZ_NET_SPLIT_F16;
    const bool want_f32 = !split || !internal || net->heads_algo == RZ_NET_HEADS_F32;
    if (internal) {
        net->feat16_valid = split;
        net->feat32_valid = want_f32;
    }
    net->dev.feat_ld = internal ? 16 * (net->dev.groups_act + net->dev.groups_val) : 6 * net->dev.S;
    net->dev.feat_val_off = internal ? 16 * net->dev.groups_act : 4 * net->dev.S;
    // Winograd kernels are persistent: one workgroup per CU (LDS bound) loops over its boards
    const int wg_cap = net->max_wgs > 0 ? net->max_wgs : net->n_cus;
    const dim3 pgrid((unsigned)(n_boards < wg_cap ? n_boards : wg_cap));
    net->raw_from_trunk = false;
    if (algo == RZ_NET_WINOGRAD_F4)
        k_trunk_wino_f4<4><<<pgrid, dim3(256), 0, (hipStream_t)stream>>>(net->dev, d_obs, d_feat, n_boards);
    else if (algo == RZ_NET_SPLIT_F16)
    {
        _Float16 *f16 = later.slot_of ? net->d_store16 : internal ? net->d_feat16 : nullptr;
        float *f32 = later.slot_of ? nullptr : want_f32 ? d_feat : nullptr;
        if (later.slot_of) net->feat16_valid = net->feat32_valid = false;   // (nothing for rz_net_heads_gemm)
        const int tiles = (net->dev.BH + net->dev.tile_rows - 1) / net->dev.tile_rows;
        const bool bits = leaves.stones != nullptr;
        // Small batches of small boards (every board has a workgroup of its own, nothing of another lane to overlap with): the
        // trunk's workgroups run the FC layers on their boards themselves -- no FC launch, no kernel boundary (same bits).
        // Every workgroup then streams ALL FC weights from L2 for its one board (the GEMM launch reads them once per 32
        // boards): AUTO takes this route while the weights are at most 40 KB (6 x 6: 39 KB, TicTacToe one game +6 %, 16 games +3 %;
        // Connect4: 26 KB); at 9 x 9 (146 KB) the
        // launch it saves is cheaper than the stream it costs (64 games -5 %, profiles/r03/in_trunk_fc.txt)
        const bool fc_here = !later.slot_of && internal && tiles <= 4 && net->dev.BH <= 10 && !rows_kernel_covers(net->dev.BH, net->dev.BW) &&
                             (net->heads_algo == RZ_NET_HEADS_IN_TRUNK ||
                              (net->heads_algo == RZ_NET_HEADS_AUTO && net->max_wgs == 0 && n_boards <= wg_cap &&
                               ((size_t)net->dev.A * 4 * net->dev.S + (size_t)64 * 2 * net->dev.S) * 4 <= 40 * 1024));
        float *raw = fc_here ? net->d_raw : nullptr, *hid = fc_here ? net->d_hid : nullptr;
        net->raw_from_trunk = fc_here;
        if (fc_here) net->feat16_valid = false;   // (the pieces stayed in LDS)
        if (net->algo == RZ_NET_SPLIT_F16 && rows_kernel_covers(net->dev.BH, net->dev.BW)) {   // wide boards: one N-tile per row
            const hipStream_t st = (hipStream_t)stream;
            const bool fp8 = net->fp8_cross;   // (float planes in this mode were refused by the callers)
            switch (net->dev.BH) {
                case 11: launch_trunk_rows<11>(bits, pgrid, st, net->dev, d_obs, leaves, f32, f16, n_boards, net->d_flags, later, fp8); break;
                case 12: launch_trunk_rows<12>(bits, pgrid, st, net->dev, d_obs, leaves, f32, f16, n_boards, net->d_flags, later, fp8); break;
                case 13: launch_trunk_rows<13>(bits, pgrid, st, net->dev, d_obs, leaves, f32, f16, n_boards, net->d_flags, later, fp8); break;
                case 14: launch_trunk_rows<14>(bits, pgrid, st, net->dev, d_obs, leaves, f32, f16, n_boards, net->d_flags, later, fp8); break;
                case 15: launch_trunk_rows<15>(bits, pgrid, st, net->dev, d_obs, leaves, f32, f16, n_boards, net->d_flags, later, fp8); break;
                default: launch_trunk_rows<16>(bits, pgrid, st, net->dev, d_obs, leaves, f32, f16, n_boards, net->d_flags, later, fp8); break;
            }
        } else if (tiles <= 2 && !fc_here && net->compact_grid && (2 * n_boards > wg_cap || net->compact_always) && net->dev.BW <= 7 && tiles * net->dev.tile_rows + 2 <= 15) {
            // small boards, more boards than CUs: the compact LDS grid (9 x 15 positions, 67 KB): two workgroups per CU
            const dim3 cgrid((unsigned)(n_boards < 2 * wg_cap ? n_boards : 2 * wg_cap));
            if (tiles <= 1) k_trunk_split<1, 4, false, 9, 15><<<cgrid, dim3(256), 0, (hipStream_t)stream>>>(net->dev, d_obs, leaves, f32, f16, n_boards, net->d_flags, nullptr, nullptr, later);
            else k_trunk_split<1, 2, false, 9, 15><<<cgrid, dim3(256), 0, (hipStream_t)stream>>>(net->dev, d_obs, leaves, f32, f16, n_boards, net->d_flags, nullptr, nullptr, later);
        } else if (tiles <= 1)        // one tile: the four waves share the output channels
            k_trunk_split<1, 4><<<pgrid, dim3(256), 0, (hipStream_t)stream>>>(net->dev, d_obs, leaves, f32, f16, n_boards, net->d_flags, raw, hid, later);
        else if (tiles <= 2)   // two tiles x two channel halves
            k_trunk_split<1, 2><<<pgrid, dim3(256), 0, (hipStream_t)stream>>>(net->dev, d_obs, leaves, f32, f16, n_boards, net->d_flags, raw, hid, later);
        else if (tiles == 3 && net->dev.tile_rows == 3)   // 9x9: three tiles + the fourth wave on a quarter of conv3's channels
            k_trunk_split<1, 3><<<pgrid, dim3(256), 0, (hipStream_t)stream>>>(net->dev, d_obs, leaves, f32, f16, n_boards, net->d_flags, raw, hid, later);
        else if (tiles <= 4)   // four tiles cover the board: one per wave
            k_trunk_split<1><<<pgrid, dim3(256), 0, (hipStream_t)stream>>>(net->dev, d_obs, leaves, f32, f16, n_boards, net->d_flags, raw, hid, later);
        else
            k_trunk_split<2><<<pgrid, dim3(256), 0, (hipStream_t)stream>>>(net->dev, d_obs, leaves, f32, f16, n_boards, net->d_flags, nullptr, nullptr, later);
    }
    else
        k_trunk<<<grid, dim3(kTrunkThreads), 0, (hipStream_t)stream>>>(net->dev, d_obs, d_feat, n_boards);
}

// The FC GEMM of the heads on the internal features -> net->d_raw (policy logits) / net->d_hid (value hidden layer).
static void launch_heads_gemm(rz_net *net, const float *d_feat, int32_t n_boards, void *stream) {
    net->dev.feat_ld = 16 * (net->dev.groups_act + net->dev.groups_val);
    net->dev.feat_val_off = 16 * net->dev.groups_act;
    if (net->raw_from_trunk && d_feat == net->d_feat) {   // the trunk's workgroups ran these layers on their boards
        net->raw_parts = 1;
        return;
    }
    int algo = net->heads_algo;
    // after the split-f16 trunk: the f16 pipe.  Beside a capped trunk the GEMM has 32 CUs (64-board workgroups keep
    // the loads of a CU below its vector memory rate), alone it has the chip (many small workgroups hide the load
    // latency); up to 256 boards the single-wave K-quarter workgroups are the shortest launch (+4 % whole-step at 64 and
    // 256 boards, level above that); every shape gives the same bits (profiles/r01/sweep_heads.txt, r02/heads_small.txt)
    if (algo == RZ_NET_HEADS_AUTO || algo == RZ_NET_HEADS_IN_TRUNK)   // (IN_TRUNK on a board size the trunk does not do it for)
        algo = net->max_wgs > 0 ? RZ_NET_HEADS_SPLIT_64 : n_boards <= 256 ? RZ_NET_HEADS_SPLIT_PARTS : RZ_NET_HEADS_SPLIT_32;
    if (d_feat != net->d_feat || !net->feat16_valid) algo = RZ_NET_HEADS_F32;
    else if (algo == RZ_NET_HEADS_F32 && !net->feat32_valid)  // F32 chosen after a trunk that wrote only the f16 pieces
        algo = net->max_wgs > 0 ? RZ_NET_HEADS_SPLIT_64 : RZ_NET_HEADS_SPLIT_32;
    const f32x4 *f16 = reinterpret_cast<const f32x4 *>(net->d_feat16);
    const int n_act_tiles = net->dev.Npad / 32;
    net->raw_parts = 1;
    if (algo == RZ_NET_HEADS_SPLIT_PARTS) {
        net->raw_parts = 4;
        const dim3 grid((unsigned)((n_boards + 31) / 32), (unsigned)(n_act_tiles + 2), 4);
        k_heads_part<5><<<grid, dim3(64), 0, (hipStream_t)stream>>>(net->dev, f16, net->d_raw, net->d_hid,
                                                                     (long long)net->raw_part_floats, (long long)net->hid_part_floats);
    } else if (algo == RZ_NET_HEADS_SPLIT_64) {
        // y = policy half (4 N-tiles) + value tile y: both halves exist even when the second has no policy tile
        const dim3 grid((unsigned)((n_boards + 63) / 64), 2);
        k_heads_split<2, 4, 3, true><<<grid, dim3(256), 0, (hipStream_t)stream>>>(net->dev, f16, net->d_raw, net->d_hid, n_boards);
    } else if (algo == RZ_NET_HEADS_SPLIT_32) {
        const dim3 grid((unsigned)((n_boards + 31) / 32), (unsigned)((n_act_tiles + 1) / 2 + 2));
        k_heads_split<1, 2, 5, false><<<grid, dim3(256), 0, (hipStream_t)stream>>>(net->dev, f16, net->d_raw, net->d_hid, n_boards);
    } else {
        const dim3 grid((unsigned)((n_boards + 31) / 32), (unsigned)(net->dev.Npad / 32 + 2));
        k_heads_gemm<<<grid, dim3(64 * kHeadWaves), 0, (hipStream_t)stream>>>(net->dev, d_feat, net->d_raw, net->d_hid, n_boards);
    }
}

static int launch_heads(rz_net *net, const float *d_feat, int32_t n_boards, float *d_logp, float *d_value,
                        void *stream) {
    launch_heads_gemm(net, d_feat, n_boards, stream);
    k_heads_finish<<<dim3((unsigned)n_boards), dim3(64), 0, (hipStream_t)stream>>>(
        net->dev, net->d_raw, net->d_hid, d_logp, d_value, n_boards, net->raw_parts, (long long)net->raw_part_floats,
        (long long)net->hid_part_floats);
    if (hipGetLastError() != hipSuccess) return net_fail(RZ_ERR_HIP, "launch of k_heads_* failed");
    return RZ_OK;
}

int rz_net_trunk(rz_net *net, const float *d_obs, int32_t n_boards, float *d_feat, void *stream) {
    int rc = net_ready(net, n_boards);
    if (rc != RZ_OK) return rc;
    if (n_boards == 0) return RZ_OK;  // an empty batch is a no-op (its tensors have no storage)
    if (!d_obs) return net_fail(RZ_ERR_ARG, "NULL device pointer");
    if (net->fp8_cross)
        return net_fail(RZ_ERR_ARG, "RZ_NET_SPLIT_F16_FP8 evaluates positions (rz_net_trunk_leaves*, rz_net_search_resident): float planes are refused, "
                                    "not computed in another arithmetic");
    if (!d_feat) {  // internal feature buffer (the input of rz_net_heads)
        if (n_boards > net->feat_boards) return net_fail(RZ_ERR_ARG, "batch larger than rz_net_reserve()d");
        d_feat = net->d_feat;
    }
    launch_trunk(net, d_obs, d_feat, n_boards, stream);
    if (hipGetLastError() != hipSuccess) return net_fail(RZ_ERR_HIP, "launch of k_trunk failed");
    return RZ_OK;
}

int rz_net_trunk_leaves(rz_net *net, const uint64_t *d_stones, const int32_t *d_to_move, const int32_t *d_last_cell,
                        int32_t n_boards, void *stream) {
    int rc = net_ready(net, n_boards);
    if (rc != RZ_OK) return rc;
    if (n_boards == 0) return RZ_OK;
    if (!d_stones || !d_to_move || !d_last_cell) return net_fail(RZ_ERR_ARG, "NULL device pointer");
    if (n_boards > net->feat_boards) return net_fail(RZ_ERR_ARG, "batch larger than rz_net_reserve()d");
    if ((net->algo != RZ_NET_SPLIT_F16 && net->algo != RZ_NET_SPLIT_F16_TILES) || !net->split_ok)
        return net_fail(RZ_ERR_ARG, "rz_net_trunk_leaves needs the RZ_NET_SPLIT_F16 trunk (the others read float planes: rz_net_trunk)");
    launch_trunk(net, nullptr, net->d_feat, n_boards, stream, LeafBits{d_stones, d_to_move, d_last_cell});
    if (hipGetLastError() != hipSuccess) return net_fail(RZ_ERR_HIP, "launch of k_trunk_split failed");
    return RZ_OK;
}

static bool deferred_trunk_covers(const rz_net *net) {   // the split-f16 trunks (both kernels), fed positions
    return (net->algo == RZ_NET_SPLIT_F16 || net->algo == RZ_NET_SPLIT_F16_TILES) && net->split_ok;
}

int rz_net_deferred_reserve(rz_net *net, int32_t max_boards, int32_t slots) {
    int rc = net_ready(net, max_boards);
    if (rc != RZ_OK) return rc;
    if (slots < 1 || max_boards < 1) return net_fail(RZ_ERR_ARG, "rz_net_deferred_reserve: slots and max_boards must be positive");
    if (max_boards <= net->store_boards && slots <= net->store_slots) return RZ_OK;
    // grow to the larger of what is held and what is asked in BOTH directions: an engine that reserved more boards (or slots)
    // earlier keeps fitting when another one asks for more of the other
    if (max_boards < net->store_boards) max_boards = net->store_boards;
    if (slots < net->store_slots) slots = net->store_slots;
    (void)hipDeviceSynchronize();
    if (net->d_store16) (void)hipFree(net->d_store16);
    if (net->d_store_raw) (void)hipFree(net->d_store_raw);
    if (net->d_valfeat) (void)hipFree(net->d_valfeat);
    net->d_store16 = nullptr;
    net->d_store_raw = net->d_valfeat = nullptr;
    net->store_boards = 0;
    net->store_slots = 0;
    const int tiles = ((max_boards + 63) / 64) * 2;   // whole 64-board blocks: the GEMM's workgroups take two tiles
    const size_t store_bytes = (size_t)slots * tiles * net->dev.groups_act * 2048;
    const size_t raw_bytes = (size_t)slots * tiles * 32 * net->dev.Npad * sizeof(float);
    const size_t val_bytes = (size_t)tiles * 32 * net->vf_groups * 4 * sizeof(float);
    if (hipMalloc((void **)&net->d_store16, store_bytes) != hipSuccess || hipMalloc((void **)&net->d_store_raw, raw_bytes) != hipSuccess ||
        hipMalloc((void **)&net->d_valfeat, val_bytes) != hipSuccess)
        return net_fail(RZ_ERR_OOM, "hipMalloc failed (deferred-priors store)");
    // the K tail of a tile's last K-step, the rows of boards that do not exist and the padding of the value rows are never
    // written: they must read as finite values (they meet zero weights, or rows nobody reads)
    if (hipMemset(net->d_store16, 0, store_bytes) != hipSuccess || hipMemset(net->d_valfeat, 0, val_bytes) != hipSuccess)
        return net_fail(RZ_ERR_HIP, "hipMemset failed (deferred-priors store)");
    net->store_tiles = tiles;
    net->store_slots = slots;
    net->store_boards = max_boards;
    return RZ_OK;
}

int rz_net_trunk_leaves_deferred(rz_net *net, const uint64_t *d_stones, const int32_t *d_to_move, const int32_t *d_last_cell,
                                 int32_t n_boards, const int32_t *d_slot_of_board, rz_value_head *out, void *stream) {
    int rc = net_ready(net, n_boards);
    if (rc != RZ_OK) return rc;
    if (!out) return net_fail(RZ_ERR_ARG, "NULL output pointer");
    if (!deferred_trunk_covers(net))
        return net_fail(RZ_ERR_ARG, "the deferred-priors route needs the RZ_NET_SPLIT_F16 trunk (a net with finite activation bounds)");
    if (n_boards > net->store_boards) return net_fail(RZ_ERR_ARG, "batch larger than rz_net_deferred_reserve()d");
    if (n_boards > 0) {
        if (!d_stones || !d_to_move || !d_last_cell || !d_slot_of_board) return net_fail(RZ_ERR_ARG, "NULL device pointer");
        const DeferredOut later{d_slot_of_board, (long long)net->store_tiles * net->dev.groups_act * 1024, net->d_valfeat, net->vf_groups * 4, net->d_trace, net->store_slots};
        launch_trunk(net, nullptr, net->d_feat, n_boards, stream, LeafBits{d_stones, d_to_move, d_last_cell}, later);
        if (hipGetLastError() != hipSuccess) return net_fail(RZ_ERR_HIP, "launch of the split-f16 trunk failed");
    }
    memset(out, 0, sizeof(*out));
    out->valfeat = net->d_valfeat;
    out->w1t = net->d_w1t;
    out->b1 = net->dev.fc_val1_b;
    out->w2 = net->dev.fc_val2_w;
    out->b2 = net->dev.fc_val2_b;
    out->ld = net->vf_groups * 4;
    out->groups = net->vf_groups;
    return RZ_OK;
}

int rz_net_search_resident(rz_net *net, rz_engine *engine, int32_t n_sims, int32_t select_first, void *stream) {
    rzt::Dev dev;
    int rc = rz_device_view(engine, &dev, (int64_t)sizeof(dev));
    if (rc != RZ_OK) return rc;
    if ((rc = net_ready(net, dev.n_games)) != RZ_OK) return rc;
    if (n_sims < 1) return net_fail(RZ_ERR_ARG, "rz_net_search_resident: n_sims must be positive");
    // a game's leaves go to store slots pend[g] .. pend[g] + n_sims - 1: more simulations than either side has slots can never fit
    // (and a leaf whose slot lies beyond the store is not written: the tree code flags its game RZ_FLAG_INTERNAL)
    if (n_sims > net->store_slots || n_sims > dev.pend_cap) {
        char detail[160];
        snprintf(detail, sizeof(detail), ": %d simulations, %d / %d slots (rz_net_deferred_reserve / rz_deferred_reserve)", n_sims, net->store_slots, dev.pend_cap);
        return net_fail(RZ_ERR_ARG, "rz_net_search_resident: more simulations than the store has slots", detail);
    }
    const bool rows = net->algo == RZ_NET_SPLIT_F16 && rows_kernel_covers(net->dev.BH, net->dev.BW);
    const int tiles = (net->dev.BH + net->dev.tile_rows - 1) / net->dev.tile_rows;
    if (!deferred_trunk_covers(net) || (!rows && (tiles > 4 || net->dev.BH > 10 || net->dev.BW > 10)))
        return net_fail(RZ_ERR_ARG, "the resident search needs the RZ_NET_SPLIT_F16 trunk on a board of 11 .. 16 rows and columns (the row-tile "
                                    "kernel) or of up to 10 x 10 (k_trunk_split with one N-tile per wave)");
    if (dev.K != 1 || dev.score_mode != RZ_SCORE_UCT_REF || dev.pend_cap <= 0)
        return net_fail(RZ_ERR_ARG, "the resident search is the deferred-priors route: RZ_SCORE_UCT_REF, one simulation in flight, rz_deferred_reserve first");
    if (dev.BH != net->dev.BH || dev.BW != net->dev.BW || dev.A != net->dev.A) return net_fail(RZ_ERR_ARG, "engine and network disagree on the board");
    // receptive-field evaluation (rz_net_delta_reserve for this many games, the default trunk on a board of 11 .. 16 rows and columns):
    // k_delta_res, TWO workgroups per CU; rz_net_delta_resident(net, 0) keeps k_trunk_rows_res
    const bool delta_res = net->delta_resident && rows && !net->fp8_cross && net->split_ok && net->base_games >= dev.n_games;
    // k_trunk_rows_res holds a CU (151 KB of LDS) for a whole search: at most one game per CU.  k_delta_res holds half a CU and its
    // workgroups depend on nothing outside their game: a batch beyond two per CU runs in ROUNDS, the dispatcher handing a CU's free half
    // to the next game of the grid as a search ends (1024 / 1536 games: two / three rounds of 512, the chip full throughout)
    // (small boards on the compact LDS grid, launch_trunk's condition: 69 KB, two workgroups per CU -- the same freedom)
    const bool compact_res = !rows && tiles <= 2 && net->compact_grid && net->dev.BW <= 7 && tiles * net->dev.tile_rows + 2 <= 15;
    if (dev.n_games > net->store_boards || (!delta_res && !compact_res && dev.n_games > net->n_cus))
        return net_fail(RZ_ERR_ARG, "the resident search runs one workgroup per game, at most one per CU (any number with rz_net_delta_reserve, or on a board of "
                                    "the compact grid) and rz_net_deferred_reserve()d");
    if (rows ? (net->vf_groups != 64 && net->vf_groups != 128) : net->vf_groups > 64) return net_fail(RZ_ERR_INTERNAL, "value head groups");
    ResArgs<true> res;
    res.E = dev;
    memset(&res.vh, 0, sizeof(res.vh));
    res.vh.w1t = net->d_w1t;
    res.vh.b1 = net->dev.fc_val1_b;
    res.vh.w2 = net->dev.fc_val2_w;
    res.vh.b2 = net->dev.fc_val2_b;
    res.vh.ld = net->vf_groups * 4;
    res.vh.groups = net->vf_groups;
    res.n_sims = n_sims;
    res.select_first = select_first ? 1 : 0;
    const DeferredOut later{dev.pend, (long long)net->store_tiles * net->dev.groups_act * 1024, nullptr, 0, nullptr, net->store_slots};
    const LeafBits leaves{dev.leaf_stones, dev.leaf_to_move, dev.leaf_last};
    const dim3 grid((unsigned)dev.n_games);
    const hipStream_t st = (hipStream_t)stream;
    net->feat16_valid = net->feat32_valid = false;
    if (delta_res) {
        if (select_first) {   // a search begins: the bases of its roots (a continued search finds them, or takes the route without)
            if ((rc = rz_net_delta_bases(net, dev.root_stones, dev.root_to_move, dev.n_games, stream)) != RZ_OK) return rc;
        }
        dl::DeltaArgs da{net->d_base_hdr, net->d_base_recs, net->d_base_ones, nullptr, net->d_delta_stats, 0, (65536 + net->dev.BW - 1) / net->dev.BW};
        dl::k_delta_res<<<grid, dim3(256), 0, st>>>(net->dev, net->d_store16, later, da, res);
        if (hipGetLastError() != hipSuccess) return net_fail(RZ_ERR_HIP, "launch of the resident search (k_delta_res) failed");
        return RZ_OK;
    }
    if (!rows) {   // (the launches of launch_trunk for these boards, RES instantiations)
        _Float16 *store = net->d_store16;
        const int ng = dev.n_games;
        if (compact_res) {   // two games per CU -- and for fewer games too: the smaller grid is 1-2 % ahead even with ONE game on the chip (TicTacToe 93.3 -> 94.6 k)
            if (tiles <= 1) k_trunk_split<1, 4, true, 9, 15><<<grid, dim3(256), 0, st>>>(net->dev, nullptr, leaves, nullptr, store, ng, net->d_flags, nullptr, nullptr, later, res);
            else k_trunk_split<1, 2, true, 9, 15><<<grid, dim3(256), 0, st>>>(net->dev, nullptr, leaves, nullptr, store, ng, net->d_flags, nullptr, nullptr, later, res);
            if (hipGetLastError() != hipSuccess) return net_fail(RZ_ERR_HIP, "launch of the resident search (compact grid) failed");
            return RZ_OK;
        }
        if (tiles <= 1) k_trunk_split<1, 4, true><<<grid, dim3(256), 0, st>>>(net->dev, nullptr, leaves, nullptr, store, ng, net->d_flags, nullptr, nullptr, later, res);
        else if (tiles <= 2) k_trunk_split<1, 2, true><<<grid, dim3(256), 0, st>>>(net->dev, nullptr, leaves, nullptr, store, ng, net->d_flags, nullptr, nullptr, later, res);
        else if (tiles == 3 && net->dev.tile_rows == 3) k_trunk_split<1, 3, true><<<grid, dim3(256), 0, st>>>(net->dev, nullptr, leaves, nullptr, store, ng, net->d_flags, nullptr, nullptr, later, res);
        else k_trunk_split<1, 1, true><<<grid, dim3(256), 0, st>>>(net->dev, nullptr, leaves, nullptr, store, ng, net->d_flags, nullptr, nullptr, later, res);
        if (hipGetLastError() != hipSuccess) return net_fail(RZ_ERR_HIP, "launch of the resident search failed");
        return RZ_OK;
    }
    switch (net->dev.BH) {
        case 11: launch_search_rows<11>(grid, st, net->dev, leaves, net->d_store16, dev.n_games, net->d_flags, later, res, net->fp8_cross); break;
        case 12: launch_search_rows<12>(grid, st, net->dev, leaves, net->d_store16, dev.n_games, net->d_flags, later, res, net->fp8_cross); break;
        case 13: launch_search_rows<13>(grid, st, net->dev, leaves, net->d_store16, dev.n_games, net->d_flags, later, res, net->fp8_cross); break;
        case 14: launch_search_rows<14>(grid, st, net->dev, leaves, net->d_store16, dev.n_games, net->d_flags, later, res, net->fp8_cross); break;
        case 15: launch_search_rows<15>(grid, st, net->dev, leaves, net->d_store16, dev.n_games, net->d_flags, later, res, net->fp8_cross); break;
        default: launch_search_rows<16>(grid, st, net->dev, leaves, net->d_store16, dev.n_games, net->d_flags, later, res, net->fp8_cross); break;
    }
    if (hipGetLastError() != hipSuccess) return net_fail(RZ_ERR_HIP, "launch of the resident search failed");
    return RZ_OK;
}

static bool delta_covers(const rz_net *net) {   // k_trunk_rows' boards, its arithmetic, positions
    return net->algo == RZ_NET_SPLIT_F16 && !net->fp8_cross && net->split_ok && rows_kernel_covers(net->dev.BH, net->dev.BW);
}

int rz_net_delta_reserve(rz_net *net, int32_t n_games) {
    int rc = net_ready(net, n_games);
    if (rc != RZ_OK) return rc;
    if (!delta_covers(net))
        return net_fail(RZ_ERR_ARG, "receptive-field evaluation exists for the RZ_NET_SPLIT_F16 trunk on boards of 11 .. 16 rows and columns");
    if (n_games < 1) return net_fail(RZ_ERR_ARG, "rz_net_delta_reserve: n_games must be positive");
    if (n_games <= net->base_games) return RZ_OK;
    (void)hipDeviceSynchronize();
    if (net->d_base_hdr) (void)hipFree(net->d_base_hdr);
    if (net->d_base_recs) (void)hipFree(net->d_base_recs);
    if (net->d_base_ones) (void)hipFree(net->d_base_ones);
    net->d_base_ones = nullptr;
    net->d_base_hdr = nullptr;
    net->d_base_recs = nullptr;
    net->base_games = 0;
    const size_t hdr_bytes = (size_t)n_games * sizeof(dl::BaseHdr), rec_bytes = (size_t)n_games * 2 * dl::kBaseBytes;
    if (hipMalloc((void **)&net->d_base_hdr, hdr_bytes) != hipSuccess || hipMalloc((void **)&net->d_base_recs, rec_bytes) != hipSuccess ||
        hipMalloc((void **)&net->d_base_ones, (size_t)n_games) != hipSuccess)
        return net_fail(RZ_ERR_OOM, "hipMalloc failed (base cache)");
    if (hipMemset(net->d_base_ones, 1, (size_t)n_games) != hipSuccess) return net_fail(RZ_ERR_HIP, "hipMemset failed (base cache)");
    if (!net->d_delta_stats && hipMalloc((void **)&net->d_delta_stats, 8 * sizeof(unsigned)) != hipSuccess)
        return net_fail(RZ_ERR_OOM, "hipMalloc failed (delta counters)");
    // valid = 0: a leaf of a game without bases takes the four passes without a base
    if (hipMemset(net->d_base_hdr, 0, hdr_bytes) != hipSuccess || hipMemset(net->d_delta_stats, 0, 8 * sizeof(unsigned)) != hipSuccess)
        return net_fail(RZ_ERR_HIP, "hipMemset failed (base cache)");
    net->base_games = n_games;
    return RZ_OK;
}

int rz_net_delta_resident(rz_net *net, int32_t on) {
    if (!net) return net_fail(RZ_ERR_ARG, "net handle is NULL");
    net->delta_resident = on != 0;
    return RZ_OK;
}

int rz_net_delta_invalidate(rz_net *net, void *stream) {
    if (!net) return net_fail(RZ_ERR_ARG, "net handle is NULL");
    if (net->base_games > 0 &&
        hipMemsetAsync(net->d_base_hdr, 0, (size_t)net->base_games * sizeof(dl::BaseHdr), (hipStream_t)stream) != hipSuccess)
        return net_fail(RZ_ERR_HIP, "hipMemsetAsync failed (base cache)");
    return RZ_OK;
}

int rz_net_delta_bases(rz_net *net, const uint64_t *d_root_stones, const int32_t *d_root_to_move, int32_t n_games, void *stream) {
    int rc = net_ready(net, n_games);
    if (rc != RZ_OK) return rc;
    if (n_games == 0) return RZ_OK;
    if (!delta_covers(net)) return net_fail(RZ_ERR_ARG, "receptive-field evaluation: RZ_NET_SPLIT_F16 on boards of 11 .. 16 rows and columns");
    if (n_games > net->base_games) return net_fail(RZ_ERR_ARG, "more games than rz_net_delta_reserve()d");
    if (!d_root_stones || !d_root_to_move) return net_fail(RZ_ERR_ARG, "NULL device pointer");
    dl::DeltaArgs da{net->d_base_hdr, net->d_base_recs, net->d_base_ones, nullptr, nullptr, 1, (65536 + net->dev.BW - 1) / net->dev.BW};
    const DeferredOut none{nullptr, 0, nullptr, 0, nullptr, 0};
    dl::k_trunk_delta<false><<<dim3((unsigned)(2 * n_games)), dim3(256), 0, (hipStream_t)stream>>>(
        net->dev, LeafBits{d_root_stones, d_root_to_move, nullptr}, nullptr, 2 * n_games, none, da);
    if (hipGetLastError() != hipSuccess) return net_fail(RZ_ERR_HIP, "launch of k_trunk_delta (bases) failed");
    return RZ_OK;
}

int rz_net_delta_leaves(rz_net *net, const uint64_t *d_stones, const int32_t *d_to_move, const int32_t *d_last_cell, int32_t n_boards,
                        const int32_t *d_slot_of_board, const uint8_t *d_active, float *d_feat32, int32_t without_base, rz_value_head *out,
                        void *stream) {
    int rc = net_ready(net, n_boards);
    if (rc != RZ_OK) return rc;
    if (!delta_covers(net)) return net_fail(RZ_ERR_ARG, "receptive-field evaluation: RZ_NET_SPLIT_F16 on boards of 11 .. 16 rows and columns");
    if (without_base && n_boards > net->base_games && (rc = rz_net_delta_reserve(net, n_boards)) != RZ_OK) return rc;   // (the kernel reads a header per board in every mode)
    if (n_boards > net->base_games) return net_fail(RZ_ERR_ARG, "more boards than rz_net_delta_reserve()d games");
    if (d_slot_of_board && n_boards > net->store_boards) return net_fail(RZ_ERR_ARG, "batch larger than rz_net_deferred_reserve()d");
    if (n_boards > 0) {
        if (!d_stones || !d_to_move || !d_last_cell) return net_fail(RZ_ERR_ARG, "NULL device pointer");
        if (!d_slot_of_board && !d_feat32) return net_fail(RZ_ERR_ARG, "rz_net_delta_leaves: neither a store slot nor an f32 buffer to write to");
        const DeferredOut later{d_slot_of_board, (long long)net->store_tiles * net->dev.groups_act * 1024, net->d_valfeat, net->vf_groups * 4,
                                net->d_trace, net->store_slots};
        dl::DeltaArgs da{net->d_base_hdr, net->d_base_recs, d_active ? d_active : net->d_base_ones, d_feat32, net->d_delta_stats, without_base ? 2 : 0, (65536 + net->dev.BW - 1) / net->dev.BW};
        const dim3 grid((unsigned)n_boards);
        _Float16 *store = d_slot_of_board ? net->d_store16 : nullptr;
        if (d_slot_of_board) net->feat16_valid = net->feat32_valid = false;
        if (net->d_trace && d_slot_of_board)
            dl::k_trunk_delta<true><<<grid, dim3(256), 0, (hipStream_t)stream>>>(net->dev, LeafBits{d_stones, d_to_move, d_last_cell}, store, n_boards, later, da);
        else
            dl::k_trunk_delta<false><<<grid, dim3(256), 0, (hipStream_t)stream>>>(net->dev, LeafBits{d_stones, d_to_move, d_last_cell}, store, n_boards, later, da);
        if (hipGetLastError() != hipSuccess) return net_fail(RZ_ERR_HIP, "launch of k_trunk_delta failed");
    }
    if (out) {
        memset(out, 0, sizeof(*out));
        out->valfeat = net->d_valfeat;
        out->w1t = net->d_w1t;
        out->b1 = net->dev.fc_val1_b;
        out->w2 = net->dev.fc_val2_w;
        out->b2 = net->dev.fc_val2_b;
        out->ld = net->vf_groups * 4;
        out->groups = net->vf_groups;
    }
    return RZ_OK;
}

// the two calls above on an engine's own arrays (rz_device_view): the bases from its ROOT positions, the step on its leaves (store
// slot pend[g], games whose active flag is 0 skipped)
int rz_net_delta_bases_engine(rz_net *net, rz_engine *engine, void *stream) {
    rzt::Dev dev;
    int rc = rz_device_view(engine, &dev, (int64_t)sizeof(dev));
    if (rc != RZ_OK) return rc;
    if (dev.BH != net->dev.BH || dev.BW != net->dev.BW) return net_fail(RZ_ERR_ARG, "engine and network disagree on the board");
    return rz_net_delta_bases(net, dev.root_stones, dev.root_to_move, dev.n_games, stream);
}

int rz_net_delta_step(rz_net *net, rz_engine *engine, rz_value_head *out, void *stream) {
    rzt::Dev dev;
    int rc = rz_device_view(engine, &dev, (int64_t)sizeof(dev));
    if (rc != RZ_OK) return rc;
    if (!out) return net_fail(RZ_ERR_ARG, "NULL output pointer");
    if (dev.K != 1 || dev.pend_cap <= 0 || dev.pend == nullptr)
        return net_fail(RZ_ERR_ARG, "rz_net_delta_step is the deferred-priors route: one simulation in flight, rz_deferred_reserve first");
    if (dev.BH != net->dev.BH || dev.BW != net->dev.BW) return net_fail(RZ_ERR_ARG, "engine and network disagree on the board");
    return rz_net_delta_leaves(net, dev.leaf_stones, dev.leaf_to_move, dev.leaf_last, dev.n_games, dev.pend, dev.active, nullptr, 0, out, stream);
}

// rz_net_trunk_leaves on the engine's leaves through the receptive-field kernel: the features go to the internal buffer's f16 tiles
// (policy and value K-steps of the FC GEMM: rz_net_heads_gemm next) -- the three-launch step (PUCT) on boards of 11 .. 16 rows.
int rz_net_delta_trunk_engine(rz_net *net, rz_engine *engine, void *stream) {
    rzt::Dev dev;
    int rc = rz_device_view(engine, &dev, (int64_t)sizeof(dev));
    if (rc != RZ_OK) return rc;
    if ((rc = net_ready(net, dev.n_games)) != RZ_OK) return rc;
    if (!delta_covers(net)) return net_fail(RZ_ERR_ARG, "receptive-field evaluation: RZ_NET_SPLIT_F16 on boards of 11 .. 16 rows and columns");
    if (dev.K != 1) return net_fail(RZ_ERR_ARG, "rz_net_delta_trunk_engine: one simulation in flight per tree (a base per game)");
    if (dev.BH != net->dev.BH || dev.BW != net->dev.BW) return net_fail(RZ_ERR_ARG, "engine and network disagree on the board");
    if (dev.n_games > net->base_games) return net_fail(RZ_ERR_ARG, "more games than rz_net_delta_reserve()d");
    if (dev.n_games > net->feat_boards) return net_fail(RZ_ERR_ARG, "batch larger than rz_net_reserve()d");
    if (net->heads_algo == RZ_NET_HEADS_F32) return net_fail(RZ_ERR_ARG, "rz_net_delta_trunk_engine writes the f16 tiles only (RZ_NET_HEADS_F32 reads f32 features: rz_net_trunk_leaves)");
    net->feat16_valid = true;
    net->feat32_valid = false;
    net->raw_from_trunk = false;
    net->dev.feat_ld = 16 * (net->dev.groups_act + net->dev.groups_val);
    net->dev.feat_val_off = 16 * net->dev.groups_act;
    const DeferredOut later{nullptr, 0, nullptr, 0, nullptr, 0};
    dl::DeltaArgs da{net->d_base_hdr, net->d_base_recs, dev.active, nullptr, net->d_delta_stats, 0, (65536 + net->dev.BW - 1) / net->dev.BW};
    dl::k_trunk_delta<false><<<dim3((unsigned)dev.n_games), dim3(256), 0, (hipStream_t)stream>>>(
        net->dev, LeafBits{dev.leaf_stones, dev.leaf_to_move, dev.leaf_last}, net->d_feat16, dev.n_games, later, da);
    if (hipGetLastError() != hipSuccess) return net_fail(RZ_ERR_HIP, "launch of k_trunk_delta failed");
    return RZ_OK;
}

int rz_net_delta_stats(rz_net *net, uint32_t *h_out8, int32_t reset) {
    if (!net || !h_out8) return net_fail(RZ_ERR_ARG, "NULL argument");
    memset(h_out8, 0, 8 * sizeof(uint32_t));
    if (!net->d_delta_stats) return RZ_OK;
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h_out8, net->d_delta_stats, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess)
        return net_fail(RZ_ERR_HIP, "hipMemcpy failed (delta counters)");
    if (reset && hipMemset(net->d_delta_stats, 0, 8 * sizeof(uint32_t)) != hipSuccess) return net_fail(RZ_ERR_HIP, "hipMemset failed (delta counters)");
    return RZ_OK;
}

int rz_net_trace_attach(rz_net *net, void *d_trace) {
    if (!net) return net_fail(RZ_ERR_ARG, "net handle is NULL");
    net->d_trace = (unsigned long long *)d_trace;
    return RZ_OK;
}

int rz_net_deferred_gemm(rz_net *net, int32_t n_boards, int32_t n_slots, rz_deferred_logits *out, void *stream) {
    int rc = net_ready(net, n_boards);
    if (rc != RZ_OK) return rc;
    if (!out) return net_fail(RZ_ERR_ARG, "NULL output pointer");
    if (n_slots < 0 || n_slots > net->store_slots || n_boards > net->store_boards)
        return net_fail(RZ_ERR_ARG, "more slots / boards than rz_net_deferred_reserve()d");
    if (n_slots > 0 && n_boards > 0) {
        // the store is a list of n_slots * store_tiles tiles of groups_act K-steps each: k_heads_split's policy groups over all of
        // them (64 boards x 128 outputs per workgroup, the K quarters over its four waves: the bits of every other shape)
        NetDev nd = net->dev;
        nd.groups_val = 0;   // a tile of the store holds the policy K-steps only
        const int n_act_tiles = nd.Npad / 32, n_groups = (n_act_tiles + 3) / 4;
        if (n_groups > 2) return net_fail(RZ_ERR_INTERNAL, "more than 256 policy outputs");
        const size_t pairs = (size_t)n_slots * net->store_tiles / 2;   // workgroups per output group: two board tiles each
        const dim3 grid((unsigned)(n_groups == 2 ? 2 * ((pairs + 7) / 8 * 8) : pairs));
        k_heads_split<2, 4, 3, false, true><<<grid, dim3(256), 0, (hipStream_t)stream>>>(
            nd, reinterpret_cast<const f32x4 *>(net->d_store16), net->d_store_raw, nullptr, n_slots * net->store_tiles * 32);
        if (hipGetLastError() != hipSuccess) return net_fail(RZ_ERR_HIP, "launch of k_heads_split failed");
    }
    out->raw = net->d_store_raw;
    out->ld = net->dev.Npad;
    out->rows_per_slot = net->store_tiles * 32;
    return RZ_OK;
}

int rz_net_set_algo(rz_net *net, int32_t algo) {
    if (!net) return net_fail(RZ_ERR_ARG, "net handle is NULL");
    if (algo != RZ_NET_DIRECT && algo != RZ_NET_WINOGRAD_F4 && algo != RZ_NET_SPLIT_F16 && algo != RZ_NET_SPLIT_F16_TILES &&
        algo != RZ_NET_SPLIT_F16_FP8)
        return net_fail(RZ_ERR_ARG, "unknown algorithm");
    if (algo == RZ_NET_SPLIT_F16_FP8 && !rows_kernel_covers(net->dev.BH, net->dev.BW))
        return net_fail(RZ_ERR_ARG, "RZ_NET_SPLIT_F16_FP8 exists for boards of 11 .. 16 rows and columns (k_trunk_rows)");
    net->fp8_cross = algo == RZ_NET_SPLIT_F16_FP8;
    net->algo = net->fp8_cross ? RZ_NET_SPLIT_F16 : algo;
    return RZ_OK;
}

int rz_net_range_info(rz_net *net, float *h_info8) {
    if (!net || !h_info8) return net_fail(RZ_ERR_ARG, "NULL argument");
    if (!net->loaded) return net_fail(RZ_ERR_ARG, "rz_net_load has not been called");
    memcpy(h_info8, net->range_info, sizeof(net->range_info));
    return RZ_OK;
}

#ifdef RZ_NET_PROFILE
int rz_net_debug_profile(long long *h_out16) {
    return hipDeviceSynchronize() == hipSuccess && hipMemcpyFromSymbol(h_out16, HIP_SYMBOL(net_prof), 24 * sizeof(long long)) == hipSuccess ? RZ_OK : RZ_ERR_HIP;
}
#endif

int rz_net_set_heads_algo(rz_net *net, int32_t heads_algo) {
    if (!net) return net_fail(RZ_ERR_ARG, "net handle is NULL");
    if (heads_algo < RZ_NET_HEADS_AUTO || heads_algo > RZ_NET_HEADS_IN_TRUNK) return net_fail(RZ_ERR_ARG, "unknown heads algorithm");
    net->heads_algo = heads_algo;
    return RZ_OK;
}

int rz_net_set_max_workgroups(rz_net *net, int32_t max_workgroups) {
    if (!net) return net_fail(RZ_ERR_ARG, "net handle is NULL");
    if (max_workgroups < 0) return net_fail(RZ_ERR_ARG, "max_workgroups must be >= 0");
    net->max_wgs = max_workgroups;
    return RZ_OK;
}

int rz_net_heads_gemm(rz_net *net, int32_t n_boards, rz_raw_heads *out, void *stream) {
    int rc = net_ready(net, n_boards);
    if (rc != RZ_OK) return rc;
    if (!out) return net_fail(RZ_ERR_ARG, "NULL output pointer");
    if (n_boards > net->feat_boards) return net_fail(RZ_ERR_ARG, "batch larger than rz_net_reserve()d");
    if (n_boards > 0) {
        launch_heads_gemm(net, net->d_feat, n_boards, stream);
        if (hipGetLastError() != hipSuccess) return net_fail(RZ_ERR_HIP, "launch of k_heads_gemm failed");
    }
    memset(out, 0, sizeof(*out));
    out->raw = net->d_raw;
    out->hid = net->d_hid;
    out->w2 = net->dev.fc_val2_w;
    out->b2 = net->dev.fc_val2_b;
    out->act_scale = net->dev.s_inv + 3;
    out->act_bias = net->dev.fc_act_b;
    out->val_scale = net->dev.s_inv + 4;
    out->val_bias = net->dev.fc_val1_b;
    out->raw_part_stride = (int64_t)net->raw_part_floats;
    out->hid_part_stride = (int64_t)net->hid_part_floats;
    out->ld = net->dev.Npad;
    out->n_parts = n_boards > 0 ? net->raw_parts : 1;
    return RZ_OK;
}

int rz_net_heads(rz_net *net, int32_t n_boards, float *d_logp, float *d_value, void *stream) {
    int rc = net_ready(net, n_boards);
    if (rc != RZ_OK) return rc;
    if (n_boards == 0) return RZ_OK;
    if (!d_logp || !d_value) return net_fail(RZ_ERR_ARG, "NULL device pointer");
    if (n_boards > net->feat_boards) return net_fail(RZ_ERR_ARG, "batch larger than rz_net_reserve()d");
    return launch_heads(net, net->d_feat, n_boards, d_logp, d_value, stream);
}

int rz_net_forward(rz_net *net, const float *d_obs, int32_t n_boards, float *d_logp, float *d_value, void *stream) {
    int rc = net_ready(net, n_boards);
    if (rc != RZ_OK) return rc;
    if (n_boards == 0) return RZ_OK;
    if (!d_obs || !d_logp || !d_value) return net_fail(RZ_ERR_ARG, "NULL device pointer");
    if (net->fp8_cross)
        return net_fail(RZ_ERR_ARG, "RZ_NET_SPLIT_F16_FP8 evaluates positions (rz_net_trunk_leaves*, rz_net_search_resident): float planes are refused, "
                                    "not computed in another arithmetic");
    if (n_boards > net->feat_boards)
        return net_fail(RZ_ERR_ARG, "batch larger than rz_net_reserve()d (no allocation on the launch path)");
    launch_trunk(net, d_obs, net->d_feat, n_boards, stream);
    if (hipGetLastError() != hipSuccess) return net_fail(RZ_ERR_HIP, "launch of k_trunk failed");
    return launch_heads(net, net->d_feat, n_boards, d_logp, d_value, stream);
}

}  // extern "C"
