// Device-side tree code shared by the engine's kernels (rz_engine.hip) and the resident search kernels of the evaluator
// (rz_net.hip): the engine's device view (Dev), bitboard rules, the selection and expand / backup bodies, the value head of the
// deferred-priors route.  Everything lives in namespace rzt with internal linkage; both translation units are compiled with
// -ffp-contract=off (see rz_engine.hip: the tree arithmetic must round like CPython does).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <stdint.h>

#include "rlzero_hip.h"
#include "rz_trace.h"

#pragma clang fp contract(off)

namespace rzt {
namespace {

constexpr int kWave = 64;
constexpr int kWords = RZ_BOARD_WORDS;

struct Dev {
    int kind, BH, BW, S, A, n_row, n_games, score_mode, n_playout;
    int K;  // simulations in flight per tree (1 = the reference's sequential search; > 1: opt-in virtual-loss mode)
    int path_stride, qcap;
    long long cap, pcap, logtab_n;  // record slots / prior floats per arena
    double c_puct;
    int4 *R;   // node records, two int4 per slot
    float *P;  // prior blocks
    int32_t *cur_arena, *top, *ptop, *nblk;
    uint64_t *root_stones;
    int32_t *root_to_move, *root_last;
    uint8_t *active;
    int32_t *path, *leaf_node, *leaf_depth, *leaf_fresh, *leaf_term;
    double *leaf_tval;
    uint64_t *leaf_stones;
    int32_t *leaf_to_move, *leaf_last;
    int32_t *queue;
    int32_t *err, *err_any, *reuse_drops;
    const double *logtab;
    int32_t *noise_ctr;
    uint64_t *noise_key;   // per game: the key of its Dirichlet stream (default: noise_seed ^ game << 20; rz_set_noise_keys)
    uint64_t noise_seed;
    int add_noise;
    // deferred priors (rz_deferred_reserve): pend[g] = records of game g since the last flush = the store slot of its next leaf;
    // record (slot, g): the prior block reserved for the node expanded in that step (-1: none), the counter of its noise
    // stream, the leaf's board
    int pend_cap;
    int32_t *pend, *pend_pb, *pend_ctr;
    uint64_t *pend_stones;
    unsigned long long *trace;   // rz_trace.h (NULL: none)
    uint64_t valid[kWords];
    // x / BW and x / n_row for x < 4096 as (x * rcp) >> 16 (rcp = ceil(65536 / d): exact while x * (rcp * d - 65536) < 65536): the
    // rule checks divide cell numbers and lane numbers, and an integer division is ~35 instructions of a latency-bound wave
    int bw_rcp, n_rcp;
    // Connect4 on a one-word board: the cells of column 0 (bits y * BW, y < BH); << c = the cells of column c (legal_of<1>)
    uint64_t col0;
    // a board of one or two words: per LANE l < 4 n of line_through the n cells of its window, relative to the window's first cell (bits j * stride
    // of direction l / n: right, down, down-right, down-left); a table in device memory -- four more 64-bit kernel arguments cost the
    // tree step scalar registers it does not have (68 B of scratch)
    const uint64_t *line_tab;
    int line_masks;   // four-word boards: != 0 when the first n - 1 cells of every window fit the 64 bits of its mask (else cell by cell)
};

// the packed node record
constexpr int kFirstCap = 4;  // child records reserved at a node's first visited child
__device__ __forceinline__ int rec_k(const int4 &lo) { return lo.w & 0xffff; }
__device__ __forceinline__ int rec_cap(const int4 &lo) { return (int)((unsigned)lo.w >> 16); }
__device__ __forceinline__ int pack_kc(int k, int cap) { return k | (cap << 16); }
__device__ __forceinline__ double rec_w(const int4 &hi) { return __hiloint2double(hi.y, hi.x); }
__device__ __forceinline__ int4 make_hi(double w, int pb, float prior) {
    return make_int4(__double2loint(w), __double2hiint(w), pb, __float_as_int(prior));
}
__device__ __forceinline__ int4 *arena_records(const Dev &E, int g, int arena) {
    return E.R + 2 * (((long long)g * 2 + arena) * E.cap);
}
__device__ __forceinline__ float *arena_priors(const Dev &E, int g, int arena) {
    return E.P + ((long long)g * 2 + arena) * E.pcap;
}
__device__ __forceinline__ int32_t *rec_n(int4 *R, int slot) { return reinterpret_cast<int32_t *>(R + 2 * slot); }
__device__ __forceinline__ double *rec_wsum(int4 *R, int slot) { return reinterpret_cast<double *>(R + 2 * slot + 1); }
__device__ __forceinline__ int32_t *rec_pb(int4 *R, int slot) { return reinterpret_cast<int32_t *>(R + 2 * slot + 1) + 2; }

// ------------------------------------------------------------------ bitboard helpers
// a[j] for a lane-dependent j, as pure ALU on the four VALUES (masks, no selects of array elements): hipcc turns a
// chain of `j == i ? a[i] : r` into ONE load with a selected address, which pins the whole board array in scratch
// memory (a memory round trip inside the dependent chain of the tree kernels; 106 scratch instructions before)
// W (template parameter of the helpers below and of the tree bodies): the 64-bit words of a colour's bitboard that the board can
// use -- 1 for boards of up to 64 cells, 2 up to 128, kWords in general.  The arrays keep kWords words (the memory layout does not
// change, the words past W are zero); a kernel that KNOWS its boards are small (its value head's width, its trunk's tiling) passes W
// and the masks over four words become a shift: the tree code is one wave's instruction count (C1 +7 %, Connect4 +5 %).
template <int W = kWords>
__device__ __forceinline__ uint64_t word_of(const uint64_t *a, int j) {
    if constexpr (W == 1) return a[0];
    if constexpr (W == 2) {   // (masks, not a select of elements: see above)
        const uint64_t m0 = j == 0 ? ~0ull : 0ull;
        return (a[0] & m0) | (a[1] & ~m0);
    }
    const uint64_t m0 = j == 0 ? ~0ull : 0ull, m1 = j == 1 ? ~0ull : 0ull, m2 = j == 2 ? ~0ull : 0ull,
                   m3 = j == 3 ? ~0ull : 0ull;
    return (a[0] & m0) | (a[1] & m1) | (a[2] & m2) | (a[3] & m3);
}
template <int W = kWords>
__device__ __forceinline__ bool test_bit(const uint64_t *a, int c) {
    return (word_of<W>(a, c >> 6) >> (c & 63)) & 1ull;
}
template <int W = kWords>
__device__ __forceinline__ void set_bit(uint64_t *a, int c) {
    const uint64_t m = 1ull << (c & 63);
    if constexpr (W == 1) {
        a[0] |= m;
    } else {
        const int j = c >> 6;
        a[0] |= (j == 0) ? m : 0ull;
        a[1] |= (j == 1) ? m : 0ull;
        if constexpr (W > 2) {
            a[2] |= (j == 2) ? m : 0ull;
            a[3] |= (j == 3) ? m : 0ull;
        }
    }
}
template <int W = kWords>
__device__ __forceinline__ int count_bits(const uint64_t *a) {
    int n = 0;
#pragma unroll
    for (int j = 0; j < W; ++j) n += __popcll(a[j]);
    return n;
}
template <int W = kWords>
__device__ __forceinline__ void load_board(const uint64_t *src, int g, uint64_t (&st)[2][kWords]) {
#pragma unroll
    for (int j = 0; j < kWords; ++j) {
        st[0][j] = j < W ? src[((long long)g * 2 + 0) * kWords + j] : 0ull;
        st[1][j] = j < W ? src[((long long)g * 2 + 1) * kWords + j] : 0ull;
    }
}
// Lane c < 2 writes colour c: its four words as two 16-byte stores (the position is wave-uniform: four selects by the lane's colour,
// no word picked by a lane-dependent index -- see word_of).  `dst` is 16-byte aligned (the engine's arrays; the resident kernels' LDS).
template <int W = kWords>
__device__ __forceinline__ void store_board(uint64_t *dst, int g, const uint64_t (&st)[2][kWords], int lane) {
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    if (lane < 2) {
        const bool c1 = lane == 1;
        u64x2 lo, hi;
        lo[0] = c1 ? st[1][0] : st[0][0];
        lo[1] = W >= 2 ? (c1 ? st[1][1] : st[0][1]) : 0ull;
        hi[0] = W > 2 ? (c1 ? st[1][2] : st[0][2]) : 0ull;   // (the words past W are zero)
        hi[1] = W > 2 ? (c1 ? st[1][3] : st[0][3]) : 0ull;
        u64x2 *p = reinterpret_cast<u64x2 *>(dst + ((long long)g * 2 + lane) * kWords);
        p[0] = lo;
        p[1] = hi;
    }
}

// ------------------------------------------------------------------ legal moves
// Wave-uniform description of the legal actions of a position.  Gomoku: every empty cell.
// Connect4: every column whose top cell is empty; lane c < BW keeps the column's height.
struct Legal {
    int k;                     // number of legal actions
    unsigned long long cols;   // Connect4: bit c = column c playable
    int height;                // Connect4: stones in column `lane` (per lane)
};

template <int W = kWords>
__device__ __forceinline__ Legal legal_of(const Dev &E, const uint64_t *occ, int lane) {
    Legal L;
    L.cols = 0ull;
    L.height = 0;
    if (E.kind == RZ_GAME_CONNECT4) {
        int h = 0;
        if constexpr (W == 1) {   // a column's stones are one mask of the board's word
            h = lane < E.BW ? __popcll(occ[0] & (E.col0 << lane)) : 0;
        } else {
            if (lane < E.BW)
                for (int y = 0; y < E.BH; ++y) h += test_bit<W>(occ, y * E.BW + lane) ? 1 : 0;
        }
        L.height = h;
        L.cols = __ballot(lane < E.BW && h < E.BH);
        L.k = __popcll(L.cols);
    } else {
        L.k = E.S - count_bits<W>(occ);
    }
    return L;
}

// r-th legal action in ascending order -> (action, cell it occupies).  Wave-uniform.
template <int W = kWords>
__device__ __forceinline__ bool nth_legal(const Dev &E, const uint64_t *occ, const Legal &L, int r, int lane,
                                          int &action, int &cell) {
    if (E.kind == RZ_GAME_CONNECT4) {
        const bool mine = (L.cols >> lane) & 1ull;
        const bool hit = mine && __popcll(L.cols & ((1ull << lane) - 1ull)) == r;
        const unsigned long long m = __ballot(hit);
        if (m == 0ull) return false;
        action = __ffsll((long long)m) - 1;
        cell = __shfl(L.height, action) * E.BW + action;
        return true;
    }
    const uint64_t below = (1ull << lane) - 1ull;
    int before = 0, found = -1;
#pragma unroll
    for (int j = 0; j < W; ++j) {
        const uint64_t e = ~occ[j] & E.valid[j];
        const bool mine = (e >> lane) & 1ull;
        if (mine && before + __popcll(e & below) == r) found = 64 * j + lane;
        before += __popcll(e);
    }
    const unsigned long long m = __ballot(found >= 0);
    if (m == 0ull) return false;
    action = cell = __shfl(found, __ffsll((long long)m) - 1);
    return true;
}

// action -> (legal?, rank among the legal actions, cell).  Wave-uniform.
__device__ __forceinline__ bool locate_action(const Dev &E, const uint64_t *occ, const Legal &L, int a,
                                              int &rank, int &cell) {
    if (a < 0 || a >= E.A) return false;
    if (E.kind == RZ_GAME_CONNECT4) {
        if (!((L.cols >> a) & 1ull)) return false;
        rank = __popcll(L.cols & ((1ull << a) - 1ull));
        cell = __shfl(L.height, a) * E.BW + a;
        return true;
    }
    if (test_bit(occ, a)) return false;
    const int i = a >> 6;
    const uint64_t below = (1ull << (a & 63)) - 1ull;
    int r = 0;
#pragma unroll
    for (int j = 0; j < kWords; ++j) {
        const uint64_t e = ~occ[j] & E.valid[j];
        r += (j < i) ? __popcll(e) : ((j == i) ? __popcll(e & below) : 0);
    }
    rank = r;
    cell = a;
    return true;
}

// Per-lane view of "my" legal action for scatter/gather over the action space: lane handles
// action 64*j + lane (j < kWords; Connect4 only j == 0).  Returns rank or -1.
__device__ __forceinline__ int lane_action_rank(const Dev &E, const uint64_t *occ, const Legal &L, int j,
                                                int lane, int &before) {
    if (E.kind == RZ_GAME_CONNECT4) {
        if (j != 0 || !((L.cols >> lane) & 1ull)) return -1;
        return __popcll(L.cols & ((1ull << lane) - 1ull));
    }
    const uint64_t e = ~occ[j] & E.valid[j];
    const int r = ((e >> lane) & 1ull) ? before + __popcll(e & ((1ull << lane) - 1ull)) : -1;
    before += __popcll(e);
    return r;
}

// n-in-row through `last` only: lane l < 4n tests the window of direction l/n that starts
// l%n steps before `last`.  Equivalent to the reference's whole-board scan
// (gomoku_env.py:136-168) when the position before `last` had no line.
template <int W = kWords>
__device__ __forceinline__ bool line_through(const uint64_t *x, int last, int BH, int BW, int n, int lane, int bw_rcp, int n_rcp,
                                             uint64_t window = 0, uint64_t window_hi = 0, bool masks = false) {
    // straight-line: every lane runs the window test on registers (test_bit reads no memory; a cell number outside the board tests a
    // bit that the conditions below discard) -- short-circuit tests cost this single wave an exec-mask round per cell
    const int d = (lane * n_rcp) >> 16, t = lane - d * n;
    const int stride = (d == 0) ? 1 : (d == 1) ? BW : (d == 2) ? BW + 1 : BW - 1;
    const int start = last - t * stride, s0 = start < 0 ? 0 : start;
    const int h = (s0 * bw_rcp) >> 16, w = s0 - h * BW;
    const bool right = w <= BW - n, down = h <= BH - n, left = w >= n - 1;
    const bool ok = (d == 0) ? right : (d == 1) ? down : (d == 2) ? (right & down) : (left & down);
    bool all = true;
    if constexpr (W == 1) {   // the window's n cells are one mask of the board's word (`window` = Dev::line_tab[lane])
        all = ((x[0] >> s0) & window) == window;
    } else if constexpr (W == 2) {
        // ... of the 128 bits of the two words that begin at the window's first cell: a low and a high mask (the high one is empty
        // unless n is 7 or more on a wide board: `window_hi` = Dev::line_tab[64 + lane])
        const int sh = s0 & 63;
        const uint64_t from0 = (x[0] >> sh) | ((x[1] << 1) << (63 - sh)), from1 = x[1] >> sh;
        const uint64_t v_lo = s0 < 64 ? from0 : from1, v_hi = s0 < 64 ? from1 : 0ull;
        all = ((v_lo & window) == window) & ((v_hi & window_hi) == window_hi);
    } else if (!masks) {   // (wave-uniform: a window longer than 64 cell numbers -- n of 6 or more on a 16-wide board -- is tested cell by cell)
        for (int j = 0; j < n; ++j) all = all & test_bit<W>(x, s0 + j * stride);
    } else {
        // four words: the first n - 1 cells as a mask of the 64 bits from the window's first cell (3 x 17 < 64 on a 16-wide board with
        // n = 5), the last cell by itself (`window` = those n - 1 cells here)
        const int j0 = s0 >> 6, sh = s0 & 63;
        const uint64_t w_lo = word_of<W>(x, j0), w_hi = word_of<W>(x, j0 + 1);   // (j0 + 1 == 4: no word, zero)
        const uint64_t v = (w_lo >> sh) | ((w_hi << 1) << (63 - sh));
        all = ((v & window) == window) & test_bit<W>(x, s0 + (n - 1) * stride);
    }
    const bool hit = (lane < 4 * n) & (start >= 0) & ok & all;
    return __ballot(hit) != 0ull;
}

// Whole-board n-in-row scan of one colour (gomoku_env.py:136-168), lanes over start cells.
template <int W = kWords>
__device__ __forceinline__ bool line_anywhere(const uint64_t *x, int S, int BH, int BW, int n, int lane, int bw_rcp) {
    bool hit = false;
    for (int m = lane; m < S; m += kWave) {
        if (!test_bit<W>(x, m)) continue;
        const int h = (m * bw_rcp) >> 16, w = m - h * BW;
        const bool right = w <= BW - n, down = h <= BH - n, left = w >= n - 1;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int stride = (d == 0) ? 1 : (d == 1) ? BW : (d == 2) ? BW + 1 : BW - 1;
            const bool ok = (d == 0) ? right : (d == 1) ? down : (d == 2) ? (right && down)
                                                                          : (left && down);
            if (!ok) continue;
            bool all = true;
            for (int j = 1; j < n; ++j) all = all && test_bit<W>(x, m + j * stride);
            hit = hit || all;
        }
    }
    return __ballot(hit) != 0ull;
}

// GomokuEnv.current_state (gomoku_env.py:95-114): 4 planes, [4][S] floats, coalesced.
__device__ __forceinline__ void write_obs(float *out, const uint64_t *mine, const uint64_t *theirs,
                                          int last, int nst, int S, int lane) {
    const float colour = (nst & 1) ? 0.0f : 1.0f;
#pragma unroll
    for (int j = 0; j < kWords; ++j) {
        const int c = 64 * j + lane;
        if (c < S) {
            out[c] = (float)((mine[j] >> lane) & 1ull);
            out[S + c] = (float)((theirs[j] >> lane) & 1ull);
            out[2 * S + c] = (nst > 0 && c == last) ? 1.0f : 0.0f;
            out[3 * S + c] = colour;
        }
    }
}

__device__ __forceinline__ void flag(const Dev &E, int g, int bits, int lane) {
    if (lane == 0) {
        atomicOr(&E.err[g], bits);
        atomicOr(E.err_any, bits);
    }
}

// node.py:83-87: exploration_score + c_puct * exploitation_score, each op rounded once.
__device__ __forceinline__ double uct_ref(double w, int n, double ln_parent, double c) {
    const double nd = (double)n;
    const double q = w / nd;
    const double u = sqrt(ln_parent / nd);
    const double cu = c * u;
    return q + cu;
}

// Opt-in PUCT (node.py:105-117 with Q = 0 at N = 0 instead of the reference's division by zero):
// exploration_score + c_puct * (prior * sqrt(parent N) / (N + 1)), evaluated in this order in fp64.
__device__ __forceinline__ double puct(double w, int n, float prior, double sqrt_parent, double c) {
    const double q = n > 0 ? w / (double)n : 0.0;
    const double u = ((double)prior * sqrt_parent) / (double)(n + 1);
    const double cu = c * u;
    return q + cu;
}

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// 32-bit integer hash (two multiplies, three xor-shifts): the uniforms of the noise come from a counter hashed with it
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

// One Gamma(0.3, 1) sample (the marginal of numpy's dirichlet(0.3 * ones(k)), node.py:65) from a
// counter-based stream: Marsaglia-Tsang for shape 1.3, boosted by U^(1/0.3).  The sample is noise: its
// transcendentals are the hardware ones (v_log_f32 / v_exp_f32 / v_cos_f32 / v_sqrt_f32, ~1 ulp) and its uniforms
// 24-bit fractions of a 32-bit hash of (key, draw index) -- the whole sample is ~60 instructions per rejection round
// (the 64-bit mixer of the first version cost ~150: a third of the instructions of an expansion, which counts double
// when the tree step shares a SIMD with a trunk wave).  `key` = 32 bits derived per (seed, game, expansion, child).
__device__ __forceinline__ float gamma03(uint32_t key) {
    const float d = 1.3f - 1.0f / 3.0f, c = 0.33903103f;  // 1 / sqrt(9 d)
    const float kLn2 = 0.69314718f, k2m24 = 1.0f / 16777216.0f;
    float g = d;
    for (int t = 0; t < 8; ++t) {
        const uint32_t h1 = hash32(key + 3u * t), h2 = hash32(key + 3u * t + 1u), h3 = hash32(key + 3u * t + 2u);
        const float u1 = (float)((h1 >> 8) + 1u) * k2m24;   // (0, 1]
        const float u2 = (float)(h2 >> 8) * k2m24;           // [0, 1): a turn of the cosine
        const float u3 = (float)((h3 >> 8) + 1u) * k2m24;   // (0, 1]
        // Box-Muller: sqrt(-2 ln u1) cos(2 pi u2); v_cos_f32 takes its argument in turns
        const float x = __builtin_amdgcn_sqrtf(-2.0f * kLn2 * __builtin_amdgcn_logf(u1)) * __builtin_amdgcn_cosf(u2);
        float v = 1.0f + c * x;
        if (v <= 0.0f) continue;
        v = v * v * v;
        if (kLn2 * __builtin_amdgcn_logf(u3) < 0.5f * x * x + d - d * v + d * kLn2 * __builtin_amdgcn_logf(v)) {
            g = d * v;
            break;
        }
    }
    const float ub = (float)((hash32(key + 0x5bd1e995u) >> 8) + 1u) * k2m24;
    return g * __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(ub) * (1.0f / 0.3f));  // ub^(1/0.3)
}

// wave arg-max of (score, index) with the LOWEST index winning ties (Python's max keeps the first).  Six exchange
// stages, none through the LDS crossbar (a ds_bpermute round trip per stage was ~600 cycles of waiting per tree level):
// lanes 1, 2 apart by DPP quad_perm, the other half of a row of 8 / 16 by row_half_mirror / row_mirror (a reduction needs
// a partner from the other group, not a particular one), rows 16 / 32 apart by v_permlane16_swap / v_permlane32_swap.
template <int CTRL>
__device__ __forceinline__ int dpp_move(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true); }
template <int STAGE>   // 0 .. 5: partner 1, 2, (4), (8), 16, 32 lanes away
__device__ __forceinline__ int partner_of(int x, int lane) {
    if (STAGE == 0) return dpp_move<0xB1>(x);    // quad_perm [1, 0, 3, 2]
    if (STAGE == 1) return dpp_move<0x4E>(x);    // quad_perm [2, 3, 0, 1]
    if (STAGE == 2) return dpp_move<0x141>(x);   // row_half_mirror
    if (STAGE == 3) return dpp_move<0x140>(x);   // row_mirror
    if (STAGE == 4) {
        const auto r = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);
        return (int)((lane & 16) ? r[0] : r[1]);
    }
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);
    return (int)((lane & 32) ? r[0] : r[1]);
}
// The value of lane l ^ OFF, the partner of __shfl_xor(x, OFF), without the LDS crossbar -- for the wave sums whose ORDER
// of additions is part of the result (the log_softmax of the heads must give the bits of k_heads_finish): 32 / 16 by
// v_permlane32/16_swap, 8 = a rotation of the row by 8, 4 = row_shl:4 / row_shr:4 by bit 2 of the lane, 2 / 1 by quad_perm.
template <int OFF>
__device__ __forceinline__ float xor_partner(float x, int lane) {
    const int v = __float_as_int(x);
    if (OFF == 32) return __int_as_float(partner_of<5>(v, lane));
    if (OFF == 16) return __int_as_float(partner_of<4>(v, lane));
    if (OFF == 8) return __int_as_float(dpp_move<0x128>(v));   // row_ror:8
    if (OFF == 4) {
        const int up = dpp_move<0x104>(v), down = dpp_move<0x114>(v);   // row_shl:4 (from lane + 4), row_shr:4 (from lane - 4)
        return __int_as_float((lane & 4) ? down : up);
    }
    if (OFF == 2) return __int_as_float(dpp_move<0x4E>(v));
    return __int_as_float(dpp_move<0xB1>(v));
}
__device__ __forceinline__ float wave_sum(float x, int lane) {   // x + partner, offsets 32, 16, .. 1: the order of the shuffle loop
    x += xor_partner<32>(x, lane);
    x += xor_partner<16>(x, lane);
    x += xor_partner<8>(x, lane);
    x += xor_partner<4>(x, lane);
    x += xor_partner<2>(x, lane);
    x += xor_partner<1>(x, lane);
    return x;
}
__device__ __forceinline__ float wave_max(float x, int lane) {
    x = fmaxf(x, xor_partner<32>(x, lane));
    x = fmaxf(x, xor_partner<16>(x, lane));
    x = fmaxf(x, xor_partner<8>(x, lane));
    x = fmaxf(x, xor_partner<4>(x, lane));
    x = fmaxf(x, xor_partner<2>(x, lane));
    x = fmaxf(x, xor_partner<1>(x, lane));
    return x;
}

// The wave's maximum first (a stage = two moves and a compare-select on the score alone), then the FIRST index that has it: the
// lanes whose own best equals the maximum vote by slot (index / 64) in ascending order, the lowest lane of the first non-empty
// vote wins -- index = 64 * slot + lane.  (-0.0 == 0.0 as in Python's `>`; a lane's own best is already the first of its slots.)
// The (score, index) pairs used to travel together: three moves, two fp64 compares, an integer compare and three selects per stage.
template <int STAGE>
__device__ __forceinline__ void max_stage(double &mx, int lane) {
    const long long b = __double_as_longlong(mx);
    const int olo = partner_of<STAGE>((int)b, lane), ohi = partner_of<STAGE>((int)(b >> 32), lane);
    const double ob = __longlong_as_double(((long long)ohi << 32) | (unsigned int)olo);
    mx = ob > mx ? ob : mx;
}
// (`wide` == false: only lanes 0 .. 15 hold candidates -- a node of at most 16 children -- and four stages give their maximum to
// every lane of their row)
template <int W = kWords>
__device__ __forceinline__ int wave_first_max(double best, int besti, bool wide = true) {
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    double mx = best;
    max_stage<0>(mx, lane);
    max_stage<1>(mx, lane);
    max_stage<2>(mx, lane);
    max_stage<3>(mx, lane);
    if (wide) {
        max_stage<4>(mx, lane);
        max_stage<5>(mx, lane);
    }
    const bool hit = besti != 0x7fffffff && best == mx;
    constexpr int kSlots = W == 1 ? 1 : (W == 2 ? 2 : 4);
#pragma unroll
    for (int s_ = 0; s_ < kSlots; ++s_) {
        const unsigned long long m = __ballot(hit && (besti >> 6) == s_);
        if (m != 0ull) return 64 * s_ + __ffsll((long long)m) - 1;
    }
    return 0x7fffffff;
}

// ------------------------------------------------------------------ SELECT + STEP
// Score every child of a fully visited (or dense) node, lane r0 = lane + 64 j takes child r0, and
// return the first maximum; the winner's record is broadcast so the descent needs no reload.
// (W: a board of W words has at most 64 W children per node -- W of the four child slots of a lane exist)
template <bool PUCT, int W = kWords>
__device__ __forceinline__ int scan_children(const Dev &E, const int4 *R, const float *P, const int4 &lo,
                                             const int4 &hi, double parent_term, int lane, int4 &clo, int4 &chi) {
    const int k = rec_k(lo), fc = lo.y, pb = hi.z;
    // the lane's (up to) four children r0 = lane + 64 j: all loads first, then the scores; the lane keeps the RECORD
    // of its best child in registers (no array survives the loop: an array of records selected by a run-time index
    // ends up in scratch memory), and the wave's winner is the best child of the lane that owns it
    int4 l0 = make_int4(0, -1, 0, 0), l1 = l0, l2 = l0, l3 = l0;
    int4 h0 = make_int4(0, 0, -1, 0), h1 = h0, h2 = h0, h3 = h0;
    float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
    if (lane < k) { l0 = R[2 * (fc + lane)]; h0 = R[2 * (fc + lane) + 1]; if (PUCT) p0 = P[pb + lane]; }
    if constexpr (W >= 2) {
        if (lane + 64 < k) { l1 = R[2 * (fc + lane + 64)]; h1 = R[2 * (fc + lane + 64) + 1]; if (PUCT) p1 = P[pb + lane + 64]; }
    }
    if constexpr (W > 2) {
        if (lane + 128 < k) { l2 = R[2 * (fc + lane + 128)]; h2 = R[2 * (fc + lane + 128) + 1]; if (PUCT) p2 = P[pb + lane + 128]; }
        if (lane + 192 < k) { l3 = R[2 * (fc + lane + 192)]; h3 = R[2 * (fc + lane + 192) + 1]; if (PUCT) p3 = P[pb + lane + 192]; }
    }
    double best = -INFINITY;
    int besti = 0x7fffffff;
    int4 blo = l0, bhi = h0;
    auto consider = [&](const int4 &cl, const int4 &ch, float prior, int r0) {
        if (r0 < k) {
            const double sc = PUCT ? puct(rec_w(ch), cl.x, prior, parent_term, E.c_puct)
                                   : uct_ref(rec_w(ch), cl.x, parent_term, E.c_puct);
            if (sc > best) {
                best = sc;
                besti = r0;
                blo = cl;
                bhi = ch;
            }
        }
    };
    consider(l0, h0, p0, lane);
    if constexpr (W >= 2) consider(l1, h1, p1, lane + 64);
    if constexpr (W > 2) {
        consider(l2, h2, p2, lane + 128);
        consider(l3, h3, p3, lane + 192);
    }
    // (TicTacToe, Connect4, the end of a 6 x 6 game: at most 16 children, two stages of the arg-max less; a scalar branch)
    const int r = __builtin_amdgcn_readfirstlane(wave_first_max<W>(best, besti, W > 2 || __builtin_amdgcn_readfirstlane(k) > 16));
    if (r >= k) return r;
    const int l = r & 63;
    clo = make_int4(__builtin_amdgcn_readlane(blo.x, l), __builtin_amdgcn_readlane(blo.y, l),
                    __builtin_amdgcn_readlane(blo.z, l), __builtin_amdgcn_readlane(blo.w, l));
    chi = make_int4(__builtin_amdgcn_readlane(bhi.x, l), __builtin_amdgcn_readlane(bhi.y, l),
                    __builtin_amdgcn_readlane(bhi.z, l), __builtin_amdgcn_readlane(bhi.w, l));
    return r;
}

// VL = false: the reference's search, ONE simulation in flight per tree (bit-exact).  VL = true (opt-in, E.K > 1,
// never used by a parity test): slot j of K simulations in flight; the slots of a game are selected one after the
// other by the game's wave, and every node on a selected path -- the leaf included -- gets a VIRTUAL LOSS (N += 1,
// W -= 1: "one more visit, lost") that steers the following slots elsewhere until the backup of the slot replaces it
// by the real value.  A first-visit child gets a placeholder record (N = 1, W = -1) at once, so the prefix invariant
// and the scans hold while its evaluation is pending; two slots may end in the same unexpanded leaf (both are
// evaluated, the first backup expands it).  Per-leaf state is indexed by g * K + j.
// `lds_leaf` (the resident search kernels, rz_net.hip): the leaf position also goes to LDS -- uint64 [8] stones, then side to move
// and last cell as two int32 -- where the trunk of the SAME workgroup reads it behind a barrier (a scalar load of the leaf arrays
// could hit the scalar cache's copy of the previous simulation's leaf).
template <bool VL, int W = kWords>
__device__ __forceinline__ void select_body(const Dev &E, float *obs, int g, int lane, int j = 0, uint64_t *lds_leaf = nullptr) {
    const int gk = VL ? g * E.K + j : g;
    // every load that does not depend on another one is issued before `active` is tested: a kernel of dependent
    // round trips (an inactive game's slots exist, reading them is harmless)
    const int arena = E.cur_arena[g];
    const int top0 = E.top[g];
    int to_move = E.root_to_move[g];
    int last = E.root_last[g];
    const bool act = E.active[g] != 0;
    const int S = E.S;
    uint64_t st[2][kWords];
    load_board<W>(E.root_stones, g, st);
    uint64_t window = 0;   // (line_through<1>: asked for with the first loads, used at the end)
    window = E.line_tab[lane];
    uint64_t window_hi = 0;
    if constexpr (W == 2) window_hi = E.line_tab[kWave + lane];
    if (!act) return;
    int4 *R = arena_records(E, g, arena);
    const float *P = arena_priors(E, g, arena);
    const bool use_puct = E.score_mode == RZ_SCORE_PUCT;
    int top = top0;
    int nst = count_bits<W>(st[0]) + count_bits<W>(st[1]);

    int32_t *path = E.path + (long long)gk * E.path_stride;
    int node = 0, depth = 0, fresh = 0;
    if (lane == 0) path[0] = 0;

    int4 lo = R[0], hi = R[1];  // the record of `node`: loaded for the root, broadcast by the scans below
    for (int it = 0; it <= S; ++it) {
        const int k = rec_k(lo);
        if (k == 0) break;  // leaf: never expanded, or a terminal position
        int fc = lo.y;
        int r;
        int4 clo = make_int4(0, -1, 0, 0), chi = make_int4(0, 0, -1, 0);
        if (use_puct) {
            // every child was initialised at expansion (N = 0, W = 0): scan all k
            r = scan_children<true, W>(E, R, P, lo, hi, sqrt((double)lo.x), lane, clo, chi);
        } else if (lo.z < k) {
            // some child still has N == 0 -> score +inf, first such child wins; its record is
            // written by the backup of this simulation, here only the slot is reserved
            const int nv = lo.z;
            int cap = rec_cap(lo);
            if (nv == cap) {
                const int ncap = cap == 0 ? (k < kFirstCap ? k : kFirstCap) : (2 * cap < k ? 2 * cap : k);
                if ((long long)top + ncap > E.cap) {
                    flag(E, g, RZ_FLAG_ARENA_FULL, lane);
                    fresh = 2;  // stop at this (expanded) node; the backup must not expand it again
                    break;
                }
                for (int i = lane; i < nv; i += kWave) {
                    const int4 a = R[2 * (fc + i)], b = R[2 * (fc + i) + 1];
                    R[2 * (top + i)] = a;
                    R[2 * (top + i) + 1] = b;
                }
                if (VL) {
                    // the paths of the slots selected earlier in this step are still pending and hold slot indices:
                    // entries inside the block that has just moved follow it
                    for (int jj = 0; jj < j; ++jj) {
                        const int gj = g * E.K + jj;
                        int32_t *pj = E.path + (long long)gj * E.path_stride;
                        const int dj = E.leaf_depth[gj];
                        for (int d = lane; d <= dj; d += kWave) {
                            const int q = pj[d];
                            if (q >= fc && q < fc + nv) pj[d] = q - fc + top;
                        }
                        if (lane == 0) {
                            const int q = E.leaf_node[gj];
                            if (q >= fc && q < fc + nv) E.leaf_node[gj] = q - fc + top;
                        }
                    }
                }
                fc = top;
                top += ncap;
                cap = ncap;
            }
            r = nv;
            fresh = 1;
            if (lane == 0) {
                R[2 * node] = make_int4(lo.x + (VL ? 1 : 0), fc, nv + 1, pack_kc(k, cap));
                if (VL) {
                    *rec_wsum(R, node) = rec_w(hi) - 1.0;
                    R[2 * (fc + r)] = make_int4(1, -1, 0, 0);          // placeholder of the pending child:
                    R[2 * (fc + r) + 1] = make_hi(-1.0, -1, 0.0f);     // visited once, lost
                }
            }
        } else {
            const int pn = lo.x;
            if (pn < 1 || pn >= E.logtab_n) {
                flag(E, g, RZ_FLAG_LOGTAB, lane);
                break;
            }
            r = scan_children<false, W>(E, R, P, lo, hi, E.logtab[pn], lane, clo, chi);
        }
        if (r >= k) {
            flag(E, g, RZ_FLAG_INTERNAL, lane);
            break;
        }
        if (VL && !fresh && lane == 0) {  // virtual loss on an inner node of the path
            *rec_n(R, node) = lo.x + 1;
            *rec_wsum(R, node) = rec_w(hi) - 1.0;
        }
        uint64_t occ[kWords];
#pragma unroll
        for (int j = 0; j < kWords; ++j) occ[j] = st[0][j] | st[1][j];
        const Legal L = legal_of<W>(E, occ, lane);
        int action, cell;
        if (!nth_legal<W>(E, occ, L, r, lane, action, cell)) {
            flag(E, g, RZ_FLAG_INTERNAL, lane);
            break;
        }
        if (to_move == 0) set_bit<W>(st[0], cell); else set_bit<W>(st[1], cell);
        last = cell;
        to_move ^= 1;
        nst += 1;
        node = fc + r;
        depth += 1;
        if (lane == 0) path[depth] = node;
        if (fresh) break;  // a first-visit child has no statistics and no children yet
        lo = clo;
        hi = chi;
    }
    if (lane == 0 && top != top0) E.top[g] = top;
    if (VL && fresh == 0 && lane == 0) {  // the path ends in an existing leaf (unexpanded or terminal): virtual loss on it
        *rec_n(R, node) = lo.x + 1;
        *rec_wsum(R, node) = rec_w(hi) - 1.0;
    }

    // game_end_winner on the leaf (gomoku_env.py:196-203)
    int term = 0;
    double tval = 0.0;
    {
        int winner = -1;
        if (depth == 0) {
            if (line_anywhere<W>(st[0], S, E.BH, E.BW, E.n_row, lane, E.bw_rcp)) winner = 0;
            else if (line_anywhere<W>(st[1], S, E.BH, E.BW, E.n_row, lane, E.bw_rcp)) winner = 1;
        } else {
            const int mover = to_move ^ 1;
            if (line_through<W>(mover == 0 ? st[0] : st[1], last, E.BH, E.BW, E.n_row, lane, E.bw_rcp, E.n_rcp, window, window_hi, E.line_masks != 0)) winner = mover;
        }
        if (winner >= 0) {
            term = 2;
            tval = (winner == to_move) ? 1.0 : -1.0;  // alphazero_mcts.py:66-68
        } else if (nst == S) {
            term = 1;
            tval = 0.0;  // alphazero_mcts.py:64-65
        }
    }
    if (lane == 0) {
        E.leaf_node[gk] = node;
        E.leaf_depth[gk] = depth;
        E.leaf_fresh[gk] = fresh;
        E.leaf_term[gk] = term;
        E.leaf_tval[gk] = tval;
        E.leaf_to_move[gk] = to_move;
        E.leaf_last[gk] = last;
    }
    store_board<W>(E.leaf_stones, gk, st, lane);
    if (lds_leaf != nullptr) {
        store_board<W>(lds_leaf, 0, st, lane);
        if (lane == 0) {
            reinterpret_cast<int *>(lds_leaf + 2 * kWords)[0] = to_move;
            reinterpret_cast<int *>(lds_leaf + 2 * kWords)[1] = last;
        }
    }
    if (obs != nullptr)
        write_obs(obs + (long long)gk * 4 * S, to_move == 0 ? st[0] : st[1],
                  to_move == 0 ? st[1] : st[0], last, nst, S, lane);
}

// Un-normalised outputs of the evaluator's last GEMM (rz_raw_heads, include/rlzero_hip.h), finished inside the tree
// kernel (same wave per game): with n_parts == 4 the sum of the four K-quarter partial sums of the FC GEMM + scale +
// bias (and ReLU for the value head's hidden units) -- the operations of k_heads_split's epilogue, in its order --
// then log_softmax over the A policy logits and value = tanh(hid . w2 + b2), the work of k_heads_finish
// (rz_net.hip), operation for operation, so every route gives identical bits.
typedef rz_raw_heads RawHeads;
typedef rz_value_head ValueHead;
constexpr int kDefWaves = 4;   // waves of a game's workgroup: all sum a quarter of the value head's first layer (k_tree_step_def), wave 0 is the game's

// ------------------------------------------------------------------ EXPAND + BACKUP
// PROBS: `logp` already holds probabilities (host evaluators hand over the callable's exact
// numbers); otherwise log-probabilities from the network (prior = exp, alphazero_agent.py:44).
// Policy arrays are indexed by ACTION: [n_games][A].
// VL (see select_body): the backup of slot j of a game with K simulations in flight.  Every node of the path already
// counts this visit (N += 1 at selection) and carries its virtual loss, so the backup adds x + 1 to W and leaves N
// alone; whether the leaf is expanded is decided from its CURRENT record (an earlier slot of the same step may have
// expanded it already).
// DEF (deferred priors, rz_value_head in include/rlzero_hip.h; RZ_SCORE_UCT_REF, K = 1): `def_value` is the leaf value the
// game's workgroup has just finished (value_head_def below); an expansion reserves its prior block and leaves a record
// (block, noise counter, board) in slot pend[g] -- the priors themselves are written by k_deferred_priors at the next flush
// -- and every active game moves on to the next slot.
template <typename VT, bool PROBS = false, bool RAW = false, bool VL = false, bool DEF = false, int W = kWords>
__device__ __forceinline__ void expand_backup_body(const Dev &E, const float *logp, const VT *value, int g,
                                                   int lane, RawHeads rh = RawHeads(), int j = 0, ValueHead vh = ValueHead(),
                                                   float (*part)[kWave] = nullptr) {
    const int gk = VL ? g * E.K + j : g;
    const int slot = DEF ? E.pend[g] : 0;
    // loads first, the test of `active` after them (see select_body)
    const bool act = E.active[g] != 0;
    const int arena = E.cur_arena[g];
    const int depth = E.leaf_depth[gk];
    const int fresh = E.leaf_fresh[gk];
    const int term = E.leaf_term[gk];
    const double leaf_tval = E.leaf_tval[gk];
    const int ptop = E.ptop[g];
    const int nblk = E.nblk[g];
    const int top_now = E.top[g];
    const int noise_ctr = E.noise_ctr[g];
    const int32_t *path = E.path + (long long)gk * E.path_stride;
    const int path_lane = path[lane < E.path_stride ? lane : 0];  // the node of path level `lane` (if that level exists)
    uint64_t st[2][kWords];
    load_board<W>(E.leaf_stones, gk, st);
    float lse = 0.0f, raw_value = 0.0f;
    float x[kWords] = {0.f, 0.f, 0.f, 0.f};  // RAW: the lane's policy logits, kept for the priors below
    if (RAW) {
        const float *r = rh.raw + (size_t)gk * rh.ld;
        const float *hp = rh.hid + (size_t)gk * 64 + lane;
        const float w2 = rh.w2[lane], b2 = rh.b2[0];
        float hid = hp[0];
        float mx = -INFINITY;
        if (rh.n_parts == 4) {
            const long long rs = rh.raw_part_stride, hs = rh.hid_part_stride;
            const float act_scale = rh.act_scale[0], val_scale = rh.val_scale[0];
            float part[kWords][4], bias[kWords];
#pragma unroll
            for (int i = 0; i < kWords; ++i) {
                const int j = lane + 64 * i;
                const bool in = j < E.A;
#pragma unroll
                for (int q = 0; q < 4; ++q) part[i][q] = in ? r[j + q * rs] : 0.0f;
                bias[i] = in ? rh.act_bias[j] : 0.0f;
            }
            const float h1 = hp[hs], h2 = hp[2 * hs], h3 = hp[3 * hs], hb = rh.val_bias[lane];
#pragma unroll
            for (int i = 0; i < kWords; ++i) {
                const float sum4 = ((part[i][0] + part[i][1]) + part[i][2]) + part[i][3];
                x[i] = lane + 64 * i < E.A ? fmaf(sum4, act_scale, bias[i]) : -INFINITY;
                mx = fmaxf(mx, x[i]);
            }
            hid = fmaxf(fmaf(((hid + h1) + h2) + h3, val_scale, hb), 0.0f);
        } else {
#pragma unroll
            for (int i = 0; i < kWords; ++i) {
                const int j = lane + 64 * i;
                x[i] = j < E.A ? r[j] : -INFINITY;
                mx = fmaxf(mx, x[i]);
            }
        }
        mx = wave_max(mx, lane);
        float sum = 0.0f;
#pragma unroll
        for (int i = 0; i < kWords; ++i) sum += (lane + 64 * i < E.A) ? expf(x[i] - mx) : 0.0f;
        sum = wave_sum(sum, lane);
        lse = mx + logf(sum);
        float h = hid * w2;
        h = wave_sum(h, lane);
        raw_value = tanhf(h + b2);
    }
    float def_value = 0.0f;
    if (DEF) {
        // the worker waves' slices of val_fc1 (value_slice_def) meet here, behind this wave's own first loads: added in slice
        // order, then the bias, ReLU, val_fc2 over the wave (the order of wave_sum) and tanh (policy_value_net.py:47-51)
        const float b1 = vh.b1[lane], w2 = vh.w2[lane], b2 = vh.b2[0];
        __syncthreads();
        float hid = part[0][lane];
#pragma unroll
        for (int q = 1; q < kDefWaves; ++q) hid += part[q][lane];
        hid = fmaxf(hid + b1, 0.0f);
        def_value = tanhf(wave_sum(hid * w2, lane) + b2);
    }
    if (!act) return;
    int4 *R = arena_records(E, g, arena);
    float *P = arena_priors(E, g, arena);

    // the reference evaluates terminal leaves too and discards the result (:59-68)
    const double v = term ? leaf_tval : (DEF ? (double)def_value : RAW ? (double)raw_value : (double)value[gk]);
    if (DEF && slot >= E.pend_cap) {   // the host flushes before the slots run out: never reached
        flag(E, g, RZ_FLAG_INTERNAL, lane);
        return;
    }
    const long long rec = DEF ? (long long)slot * E.n_games + g : 0;

    int new_fc = -1, new_nv = 0, new_k = 0, new_cap = 0, new_pb = -1;
    bool expand_now = !term && fresh != 2;
    if (VL && expand_now) {  // pending in several slots: only the first backup expands the leaf
        const int leaf = __shfl(path_lane, depth < kWave ? depth : 0);
        expand_now = rec_k(R[2 * (depth < kWave ? leaf : path[depth])]) == 0;
    }
    if (expand_now) {
        uint64_t occ[kWords];
#pragma unroll
        for (int j = 0; j < kWords; ++j) occ[j] = st[0][j] | st[1][j];
        const Legal L = legal_of<W>(E, occ, lane);
        const int k = L.k;
        const bool dense = !DEF && E.score_mode == RZ_SCORE_PUCT;
        const int top = dense ? top_now : 0;
        if ((long long)ptop + k > E.pcap || nblk >= E.qcap) {
            flag(E, g, RZ_FLAG_BLOCKS_FULL, lane);
        } else if (dense && (long long)top + k > E.cap) {
            flag(E, g, RZ_FLAG_ARENA_FULL, lane);
        } else {
            new_pb = ptop;
            new_k = k;
            if (dense) {  // PUCT: all k child records are reserved and initialised below
                new_fc = top;
                new_nv = k;
                new_cap = k;
            }
            if (lane == 0) {
                E.ptop[g] = ptop + k;
                E.nblk[g] = nblk + 1;
                if (dense) E.top[g] = top + k;
            }
            if (DEF) {
                // the block is reserved, its priors come at the next flush: what k_deferred_priors needs to write them
                if (lane == 0) {
                    E.pend_pb[rec] = ptop;
                    E.pend_ctr[rec] = noise_ctr;
                    if (E.add_noise) E.noise_ctr[g] = noise_ctr + 1;
                }
                store_board<W>(E.pend_stones, (int)rec, st, lane);
            }
            // TreeNode.expand: one child per legal move, prior from the policy head; in self-play
            // mixed with Dirichlet(0.3) noise at EVERY expanded node (node.py:63-69)
            const float uniform = 1.0f / (float)k;
            if (!DEF) {
            int ranks[kWords];
            int before = 0;
#pragma unroll
            for (int j = 0; j < kWords; ++j) ranks[j] = lane_action_rank(E, occ, L, j, lane, before);
            float noise[kWords] = {0.f, 0.f, 0.f, 0.f};
            float noise_sum = 1.0f;
            if (E.add_noise) {
                const int ctr = noise_ctr;
                const uint64_t key = mix64(mix64(E.noise_key[g]) ^ (uint64_t)ctr);
                float local = 0.0f;
#pragma unroll
                for (int j = 0; j < kWords; ++j)
                    if (ranks[j] >= 0) {
                        noise[j] = gamma03(hash32((uint32_t)key ^ (uint32_t)(key >> 32)) + 0x9E3779B9u * (uint32_t)(64 * j + lane + 1));
                        local += noise[j];
                    }
                local = wave_sum(local, lane);
                noise_sum = local > 0.0f ? local : 1.0f;
                if (lane == 0) E.noise_ctr[g] = ctr + 1;
            }
#pragma unroll
            for (int j = 0; j < kWords; ++j) {
                const int r = ranks[j];
                if (r < 0) continue;
                const int a = 64 * j + lane;
                float prior = uniform;
                // = exp(log_softmax); the logit is the one loaded above (a second load here would sit between the
                // prior stores, and its s_waitcnt vmcnt(0) also waits for the stores before it)
                if (RAW) prior = expf(x[j] - lse);
                else if (logp) prior = PROBS ? logp[(long long)gk * E.A + a] : expf(logp[(long long)gk * E.A + a]);
                if (E.add_noise) prior = 0.75f * prior + 0.25f * (noise[j] / noise_sum);
                P[ptop + r] = prior;
                if (dense) {
                    R[2 * (top + r)] = make_int4(0, -1, 0, 0);
                    R[2 * (top + r) + 1] = make_hi(0.0, -1, prior);
                }
            }
            }  // !DEF
        }
    }
    if (DEF && lane == 0) {
        if (new_pb < 0) E.pend_pb[rec] = -1;   // nothing expanded in this step (a terminal leaf, a full arena)
        E.pend[g] = slot + 1;
    }

    // TreeNode.update_recursive(-leaf_value): leaf gets -v, its parent +v, ... (node.py:135-144)
    for (int d = lane; d <= depth; d += kWave) {
        const int node = d == lane ? path_lane : path[d];
        const double x = ((depth - d) & 1) ? v : -v;
        if (VL) {
            // N was counted at selection; W trades the virtual loss for the value; a leaf expanded now gets its blocks
            if (d == depth && new_k > 0) {
                const int4 m = R[2 * node];
                R[2 * node] = make_int4(m.x, new_fc, new_nv, pack_kc(new_k, new_cap));
                *rec_pb(R, node) = new_pb;
            }
            *rec_wsum(R, node) += x + 1.0;
        } else if (d == depth) {
            // the leaf: a first-visit slot gets its whole record here; an old leaf that is now
            // expanded gets its prior block (and, dense, its child block)
            if (fresh == 1) {
                R[2 * node] = make_int4(1, new_fc, new_nv, pack_kc(new_k, new_cap));
                R[2 * node + 1] = make_hi(0.0 + x, new_pb, 0.0f);  // int 0 + float in the reference (node.py:29,133)
            } else {
                const int4 m = R[2 * node];
                R[2 * node] = (new_k > 0) ? make_int4(m.x + 1, new_fc, new_nv, pack_kc(new_k, new_cap))
                                          : make_int4(m.x + 1, m.y, m.z, m.w);
                *rec_wsum(R, node) += x;
                if (new_k > 0) *rec_pb(R, node) = new_pb;
            }
        } else {
            *rec_n(R, node) += 1;
            *rec_wsum(R, node) += x;
        }
    }
}

// ------------------------------------------------------------------ deferred priors
// The first layer of the value head of the game's leaf (val_fc1, policy_value_net.py:48) by the four waves of the game's
// workgroup -- one per SIMD, each small enough (<= 112 registers) to sit beside a wave of the trunk, whose workgroup holds
// 400 of a SIMD's 512: wave q sums K-quarter q (2 x PER groups of 4 inputs) into hidden unit `lane` in f32, k ascending, even
// and odd k in two chains.  PER 16-byte weight loads of a wave are in flight together (4 x PER x 1 KB: the layer in TWO memory
// round trips); the quarters meet in LDS (expand_backup_body<DEF>), where wave 0 -- the game's wave -- goes on alone.  The input
// row is uniform over the wave: the constant address space makes its loads scalar.
template <int PER>
__device__ __forceinline__ void value_quarter_def(const ValueHead &vh, int gk, int lane, int q, float (*part)[kWave]) {
    typedef float vec4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) vec4 *uniform_row;
    const vec4 *w = reinterpret_cast<const vec4 *>(vh.w1t) + (size_t)q * 2 * PER * kWave + lane;
    const uniform_row f = (uniform_row)(reinterpret_cast<const vec4 *>(vh.valfeat + (size_t)gk * vh.ld) + q * 2 * PER);
    float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        vec4 wv[PER], fv[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            wv[i] = w[(size_t)(half * PER + i) * kWave];
            fv[i] = f[half * PER + i];
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            a0 = fmaf(fv[i].x, wv[i].x, a0);
            a1 = fmaf(fv[i].y, wv[i].y, a1);
            a0 = fmaf(fv[i].z, wv[i].z, a0);
            a1 = fmaf(fv[i].w, wv[i].w, a1);
        }
        __builtin_amdgcn_sched_barrier(0);   // (the second half's loads are not hoisted over the first half's sums: registers)
    }
    part[q][lane] = a0 + a1;
}

// the same quarter with the input row in LDS (`row`: 8 x PER x 4 floats, zero padded; every lane reads the same address: a
// broadcast) -- the resident search kernels, whose trunk leaves the value head's inputs on the CU.  Same operations, same order.
template <int PER>
__device__ __forceinline__ void value_quarter_lds(const ValueHead &vh, const float *row, int lane, int q, float (*part)[kWave]) {
    typedef float vec4 __attribute__((ext_vector_type(4)));
    const vec4 *w = reinterpret_cast<const vec4 *>(vh.w1t) + (size_t)q * 2 * PER * kWave + lane;
    const vec4 *f = reinterpret_cast<const vec4 *>(row) + q * 2 * PER;
    float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        vec4 wv[PER], fv[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            wv[i] = w[(size_t)(half * PER + i) * kWave];
            fv[i] = f[half * PER + i];
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            a0 = fmaf(fv[i].x, wv[i].x, a0);
            a1 = fmaf(fv[i].y, wv[i].y, a1);
            a0 = fmaf(fv[i].z, wv[i].z, a0);
            a1 = fmaf(fv[i].w, wv[i].w, a1);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    part[q][lane] = a0 + a1;
}

}  // namespace
}  // namespace rzt
