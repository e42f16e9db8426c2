// rz_delta.h -- receptive-field ("delta") leaf evaluation for boards of 11 .. 16 rows and columns; included by rz_net.hip inside its
// anonymous namespace, behind namespace rt (it is the arithmetic of k_trunk_rows, cell by cell).
//
// Why.  The reference's search (alphazero_mcts.py:42-71 under node.py:32-42's UCT rule) is near breadth-first: at 15 x 15 / 800
// simulations a leaf is the root position plus ONE or TWO stones (mean depth 1.74).  PolicyValueNet.forward
// (policy_value_net.py:34-52) is three zero-padded 3 x 3 convolutions and two 1 x 1 head convolutions, so a changed input cell moves
// conv1's output only in the 3 x 3 window around it, conv2's in the 5 x 5, conv3's and the head features in the 7 x 7 window.  Per
// game and move the trunk is therefore evaluated ONCE on two pseudo-positions of the root ("bases": the root's stones seen by the
// side to move / by the other side, no last-move plane, the stone-count plane of that parity) with conv1's and conv2's outputs and the
// six head-feature planes kept in HBM (BaseCache); a leaf then recomputes only the cells inside the windows of the cells where its
// planes differ from the base of its parity (the added stones; the last move is one of them).
//
// Same bits.  An MFMA column (a cell) never mixes with another column, an accumulator sees the products of its cell in the order
// (tap column, channel chunk) -> kernel row -> (hi x hi, hi x lo, lo x hi) of rt::slot_r, the epilogues are the element-wise code of
// trunk_rows_body, the 1 x 1 head sums keep its tree ((lane groups 0 + 1) + (2 + 3), then the waves in wave order): every value a
// leaf's window holds is what k_trunk_rows computes for that cell, bit for bit (tests/test_delta_trunk.py); a cell outside the
// windows takes the base's value, which is k_trunk_rows' as well.  The one formal difference: k_trunk_rows never multiplies the zero
// ring above / below the board, here an off-board tap meets an all-zero record (acc + 0 = acc).
//
// One kernel, three uses (DeltaArgs::mode).  A workgroup owns a leaf.  The cells are not board rows but LISTS: thread = cell finds its
// distance to the changed cells, ballots compact the cells of every layer's window into tiles of 16 (conv2, conv3) or 32 (conv1) MFMA
// columns, a map [halo position] -> LDS record gives every tap its operand (records of cells outside a layer's window are copied from
// the base, off-board taps read the zero record).  LDS: 77 KB -- TWO workgroups per CU, so one leaf's gathers and vector phases run
// under the other's MFMAs.  A leaf whose windows exceed the budget (more than kMaxD changed cells: deep paths late in a game; or a
// base that is not a subset of the leaf) is evaluated by the same code WITHOUT a base, in four passes over the board's quadrants
// (each pass: conv3 on <= 8 x 8 cells, conv2 on the 9 x 9 and conv1 on the 10 x 10 around them) -- no second kernel, no host decision.
// mode 1 builds the bases that way and writes the cache.
// (issue priorities of the resident search's phases, s_setprio: its serial tree phase -- one wave, ~20 k cycles a simulation -- ahead of
// the OTHER game's trunk waves on the same SIMDs: +2.5 % on the whole line, profiles/r06/ab_prio.txt; PRO: the trunk's prologue likewise)
#ifndef RZ_DELTA_TREE_PRIO
#define RZ_DELTA_TREE_PRIO 1
#endif
#ifndef RZ_DELTA_PRO_PRIO
#define RZ_DELTA_PRO_PRIO 0
#endif

namespace dl {

using sp::f16x4;
using sp::f16x8;
typedef const __attribute__((address_space(3))) f16x8 *lds_frag;
typedef const __attribute__((address_space(3))) uint32_t *lds_u32;
typedef __attribute__((address_space(3))) f16x4 *lds_h4;
typedef __attribute__((address_space(3))) f32x4 *lds_v4;

constexpr int kC1Slots = 128, kC2Slots = 164;   // records of conv1's / conv2's output a pass may hold
constexpr int kT1 = 4, kT2 = 8, kT3 = 8;        // tiles per pass: conv1 (32 cells each), conv2 / conv3 (16 cells each)
constexpr int kMaxD = 4;                        // changed cells a delta pass handles
constexpr int P1 = rt::Geo<32>::pos_bytes, P2 = rt::Geo<64>::pos_bytes;   // 160 / 288: k_trunk_rows' records
constexpr int kGrid = kRowW * kRowW;            // halo positions (18 x 18), position of cell (y, x) = (y + 1) * 18 + x + 1
constexpr int kShareFloats = kT3 * 4 * 96;      // the waves' shares of the head sums: [tile][wave][output][column]
constexpr int kHeadW = 128 * 6 * 4 + 128 * 4 + 32;   // the 1 x 1 head convolutions' weights [128][6], conv3's biases [128], the head biases [6 + 2], staged once
// (the planes: the hi pieces only -- the lo pieces of 0 / 1 planes are zero and conv1 skips their products)
constexpr int kOffC1 = sp::kInPieceBytes, kOffC2 = kOffC1 + kC1Slots * P1, kOffZero = kOffC2 + kC2Slots * P2, kOffHead = kOffZero + P2,
              kOffMap1 = kOffHead + kHeadW, kOffMap2 = kOffMap1 + kGrid * 4, kOffList = kOffMap2 + kGrid * 4, kOffCnt = kOffList + 3 * 128 * 2,
              kLdsBytes = kOffCnt + 5 * 4 * 4;
static_assert(kShareFloats * 4 <= kC1Slots * P1, "the shares lie inside conv1's records (dead behind conv2)");
static_assert(2 * (kLdsBytes + 512 * 4 + 4 * 64 * 4 + 80 + 512) <= 160 * 1024, "two workgroups per CU, also of the resident search (value row, K-quarter sums, leaf)");
static_assert(kOffC1 % 16 == 0 && kOffZero % 16 == 0 && kOffHead % 16 == 0 && kOffMap1 % 16 == 0, "alignment");

// The base cache: per game a header (the stones the two bases were computed from) and per (game, parity) the records.
constexpr int kCells = RZ_MAX_BOARD_SIZE * RZ_MAX_BOARD_SIZE;
struct BaseHdr {
    uint64_t stones[2 * RZ_BOARD_WORDS];
    int32_t to_move, valid, pad[2];
};
constexpr size_t kBaseC2 = (size_t)kCells * 128, kBaseV = kBaseC2 + (size_t)kCells * 256, kBaseBytes = kBaseV + (size_t)6 * kCells * 4;

struct DeltaArgs {
    BaseHdr *hdr;            // [games]
    char *recs;              // [games][2 parities][kBaseBytes]: conv1 records [cell][hi 32 | lo 32] f16, conv2 records [cell][hi 64 | lo 64],
                             // head features [6][kCells] f32 (behind their ReLU)
    const uint8_t *active;   // mode 0: games whose flag is 0 are skipped (NULL: none is)
    float *feat32;           // the features as f32 [board][6 S] (tests; NULL otherwise)
    unsigned *stats;         // [8] counters (NULL: none): delta leaves, leaves without a base, conv3 tiles, changed cells, conv2 tiles, -, and of
                             // workgroup 0 of the LAST resident launch: shader-clock cycles >> 8, ticks of the constant 100 MHz clock
    int mode;                // 0: the leaves of `leaves` against the cache; 1: build the cache from the ROOT positions in `leaves`
                             // (board = 2 game + parity); 2: the leaves without a base (the four-pass route alone: a checker)
    int bw_rcp;              // ceil(65536 / width): cell / width = (cell * bw_rcp) >> 16 for cell < 4096
};

__device__ __forceinline__ uint32_t lds_addr(const void *p) {
    return (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char *)p);
}

// ---- the convolutions at tiles of gathered cells.  pos19[t] = the lane's cell of tile t as halo position - 19 (every tap's map entry
// then lies at a non-negative offset); rec[t][dy] = the LDS record of tile t's tap (dy, current tap column) + the lane's k block.
// Per accumulator the order of rt::slot_r: (tap column, chunk) -> kernel row -> hi x hi, hi x lo, lo x hi.  The work of a step (tap
// column, chunk, kernel row) is cut into UNITS of at most three tiles: a unit's fragments are read from LDS while the unit before it
// multiplies (two sets of registers), and inside a unit the products go combo -> tile -> M-tile, so that an accumulator meets its next
// product 2 x (tiles of the unit) MFMAs later.
constexpr int unit_tiles(int nt) { return nt <= 3 ? nt : (nt + 1) / 2; }
constexpr int units_of(int nt) { return (nt + unit_tiles(nt) - 1) / unit_tiles(nt); }

template <int CIN, int NT, int UT>
__device__ __forceinline__ void read_unit(f16x8 (&b)[UT][2], const uint32_t (&rec)[NT][3], int h, int dy, int c) {
#pragma unroll
    for (int i = 0; i < UT; ++i) {
        const int t = h * UT + i;
        if (t < NT) {
            const uint32_t r = rec[t][dy] + (uint32_t)(c * 64);
            b[i][0] = *(lds_frag)(uintptr_t)r;
            b[i][1] = *(lds_frag)(uintptr_t)(r + CIN * 2);
        }
    }
}
template <int TM, int NT, int UT>
__device__ __forceinline__ void mfma_unit(f32x4 (&acc)[TM][NT], const f16x8 (&a)[TM][2], const f16x8 (&b)[UT][2], int h) {
#pragma unroll
    for (int combo = 0; combo < 3; ++combo)
#pragma unroll
        for (int i = 0; i < UT; ++i) {
            const int t = h * UT + i;
            if (t < NT) {
#pragma unroll
                for (int m = 0; m < TM; ++m)
                    acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m][combo == 2], b[i][combo == 1], acc[m][t], 0, 0, 0);
            }
        }
}
template <int NT>
__device__ __forceinline__ void lookup(uint32_t (&rec)[NT][3], lds_u32 map, const int (&pos19)[NT], int dx, int g) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) rec[t][dy] = map[pos19[t] + dx + dy * kRowW] + (uint32_t)(g * 16);
}

// conv2 (32 -> 64): the wave's 16 output channels at NT <= 4 tiles; ALL its weight fragments (9 taps x hi / lo: 72 registers) are in
// registers.  A step = (tap column, kernel row) is one unit.
template <int NT>
__device__ __forceinline__ void conv2_g(lds_u32 map, const int (&pos19)[NT], int lane, const f16x8 (&a2)[9][1][2], f32x4 (&acc)[1][NT]) {
    static_assert(NT <= 4, "one unit per step");
    const int g = lane >> 4;
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[0][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int AHEAD = NT <= 2 ? 2 : 1;   // units read ahead: a unit of one or two tiles (3 - 6 MFMAs) is shorter than an LDS round trip
    uint32_t rec[3][NT][3];
    f16x8 b[AHEAD + 1][NT][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) lookup<NT>(rec[dx], map, pos19, dx, g);
#pragma unroll
    for (int u = 0; u < AHEAD; ++u) read_unit<32, NT, NT>(b[u], rec[u / 3], 0, u % 3, 0);
#pragma unroll
    for (int u = 0; u < 9; ++u) {   // u = 3 dx + dy
        const int dx = u / 3, dy = u % 3;
        if (u + AHEAD < 9) read_unit<32, NT, NT>(b[(u + AHEAD) % (AHEAD + 1)], rec[(u + AHEAD) / 3], 0, (u + AHEAD) % 3, 0);
        mfma_unit<1, NT, NT>(acc, a2[dy * 3 + dx], b[u % (AHEAD + 1)], 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// conv3 (64 -> 128): the wave's 32 output channels (two M-tiles) at NT <= 6 tiles.  A step (tap column, chunk, kernel row) needs four
// weight fragments; they come through a ring of R steps, requested R - 1 steps ahead: R = 6 where a step is short (up to four tiles:
// 24 MFMAs a step do not cover an L2 round trip two steps ahead), R = 3 above.  The first R - 1 steps are requested BEFORE the barrier
// behind conv2, which the waves meet in here (SYNC; the ring as a kernel-scope array handed through the barrier and the switch over the
// tile count cost 200 - 400 spilled registers).  The tap-column loop is rolled: 6 steps a turn.
#ifndef RZ_DELTA_RING6_UPTO
#define RZ_DELTA_RING6_UPTO 2
#endif
constexpr int ring3(int nt) { return nt <= RZ_DELTA_RING6_UPTO ? 6 : 3; }
__device__ __forceinline__ void load_a3(f16x8 (&a)[2][2], __amdgpu_buffer_rsrc_t w_rsrc, int lane, int dx, int j) {
    using G = rt::Geo<64>;
    const int c = j / 3, dy = j % 3;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int p = 0; p < 2; ++p)
            a[m][p] = sp::load_w(w_rsrc, lane * 16, dx * (G::chunks * 2048) + ((m * G::steps + dy * 3 * G::chunks + c) * 2 + p) * 1024);
}
template <int NT>
__device__ __forceinline__ void conv3_g(lds_u32 map, const uint16_t *list3, const void *wts, int lane, bool sync, f32x4 (&acc)[2][NT]) {
    static_assert(NT <= 6, "two units per step");
    constexpr int R = ring3(NT), H = units_of(NT);
    const int n = lane & 15, g = lane >> 4;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(wts), 0, 0x7fffffff, 0x00020000);
    f16x8 a3[R][2][2];
#pragma unroll
    for (int j = 0; j < R - 1; ++j) load_a3(a3[j], w_rsrc, lane, 0, j);
    if (sync) __syncthreads();   // conv2's records are complete (the barrier behind conv2, met behind the first requests)
    int pos19[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) pos19[t] = (int)list3[16 * t + n] - 19;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint32_t rec[NT][3], recn[NT][3];
    // units read ahead: a unit of one or two tiles (6 - 12 MFMAs) is shorter than an LDS round trip under load
#ifndef RZ_DELTA_AHEAD3_SMALL
#define RZ_DELTA_AHEAD3_SMALL 1
#endif
    constexpr int UT = unit_tiles(NT), AH = UT <= 2 ? RZ_DELTA_AHEAD3_SMALL : 1, NB = AH + 1;
    static_assert((6 * H) % NB == 0 && AH <= H * 6, "a unit's registers are a constant of the unrolled body");
    f16x8 b[NB][UT][2];
    lookup<NT>(rec, map, pos19, 0, g);
#pragma unroll
    for (int u = 0; u < AH; ++u) read_unit<64, NT, UT>(b[u], rec, u % H, (u / H) % 3, u / H / 3);
#pragma unroll 1
    for (int dx = 0; dx < 3; ++dx) {
        lookup<NT>(recn, map, pos19, dx < 2 ? dx + 1 : 2, g);   // the next tap column's records (past the last: unused)
#pragma unroll
        for (int u = 0; u < 6 * H; ++u) {   // unit u = step (u / H) x tiles of group (u % H); 6 H is a multiple of NB: a unit's registers are constants
            const int j = u / H, h = u % H;
            if (h == 0) {   // the fragments of step j + R - 1 (past the last tap column: a reload of the last one's, never used)
                const int jj = j + R - 1, dx2 = jj < 6 ? dx : (dx < 2 ? dx + 1 : 2);
                load_a3(a3[jj % R], w_rsrc, lane, dx2, jj % 6);
            }
            {
                const int v = u + AH;   // the unit read now (past this tap column's last: the next column's first)
                if (v < 6 * H) read_unit<64, NT, UT>(b[v % NB], rec, v % H, (v / H) % 3, v / H / 3);
                else read_unit<64, NT, UT>(b[v % NB], recn, (v - 6 * H) % H, ((v - 6 * H) / H) % 3, (v - 6 * H) / H / 3);
            }
            mfma_unit<2, NT, UT>(acc, a3[j % R], b[u % NB], h);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) rec[t][dy] = recn[t][dy];
    }
}

// the lane's head values [tile][output] of T tiles (T = 4: 24 values, T = 8: 48) -> those of the T / 4 tiles its lane group is left with,
// summed over the four lane groups in f4::reduce_scatter_96's order: (group 0 + group 1) + (group 2 + group 3).  Group g keeps tiles
// (T / 2) (g & 1) + (T / 4) (g >> 1) + ...
template <int T>
__device__ __forceinline__ void reduce_scatter(const float (&vals)[6 * T], float (&out)[6 * T / 4]) {
    float r1[3 * T];
#pragma unroll
    for (int i = 0; i < 3 * T; ++i) {
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(vals[i]), __float_as_uint(vals[3 * T + i]), false, false);
        r1[i] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
#pragma unroll
    for (int i = 0; i < 3 * T / 2; ++i) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(r1[i]), __float_as_uint(r1[3 * T / 2 + i]), false, false);
        out[i] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
}

// conv2 of NT <= 4 tiles -> the cells' conv2 records (bias, ReLU, hi + lo pieces: trunk_rows_body's epilogue); n2 = cells of these tiles
template <int NT>
__device__ __forceinline__ void conv2_tiles(lds_u32 map1, lds_u32 map2, const uint16_t *list2, int n2, int lane, int wave, const f16x8 (&a2)[9][1][2],
                                            f32x4 bias2, float k2) {
    const int n = lane & 15, g = lane >> 4;
    int pos19[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) pos19[t] = (int)list2[16 * t + n] - 19;
    f32x4 acc[1][NT];
    conv2_g<NT>(map1, pos19, lane, a2, acc);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (16 * t + n < n2) {
            const uint32_t rec = map2[pos19[t] + kRowW + 1] + (uint32_t)((16 * wave + 4 * g) * 2);
            float z[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) z[j] = fmaxf(fmaf(acc[0][t][j], k2, bias2[j]), 0.0f);
            f16x4 hi, lo;
            sp::split4(z, hi, lo);
            *(lds_h4)(uintptr_t)rec = hi;
            *(lds_h4)(uintptr_t)(rec + 128) = lo;
        }
    }
}

// conv3 of NT <= 6 tiles + the two 1 x 1 head convolutions -> the wave's shares of the six head sums of these tiles (shares: the first
// tile's).  headw (LDS): [128][6] weights, [128] biases.
template <int NT>
__device__ __forceinline__ void conv3_tiles(lds_u32 map2, const uint16_t *list3, const void *wts, int lane, int wave, bool sync,
                                            const char *headw, float k3, float *shares) {
    constexpr int T = NT <= 4 ? 4 : 8;
    const int n = lane & 15, g = lane >> 4;
    float vals[6 * T];
    {
        f32x4 acc[2][NT];
        conv3_g<NT>(map2, list3, wts, lane, sync, acc);
        // the lane's 8 channels of conv3 (32 wave + 16 m + 4 g + j) meet 6 outputs each (trunk_rows_body: hwr, b3r)
        f32x4 hwr[2][6], b3r[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int c0 = 32 * wave + 16 * m + 4 * g;
#pragma unroll
            for (int i = 0; i < 6; ++i) hwr[m][i] = *reinterpret_cast<const f32x4 *>(headw + (c0 * 6 + 4 * i) * 4);
            b3r[m] = *reinterpret_cast<const f32x4 *>(headw + 128 * 6 * 4 + c0 * 4);
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            f32x2 v2[3] = {f32x2{0.0f, 0.0f}, f32x2{0.0f, 0.0f}, f32x2{0.0f, 0.0f}};
            if (t < NT) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float hv = fmaxf(fmaf(acc[m][t < NT ? t : 0][j], k3, b3r[m][j]), 0.0f);
#pragma unroll
                        for (int o2 = 0; o2 < 3; ++o2) {
                            const int e = 6 * j + 2 * o2;
                            v2[o2] = __builtin_elementwise_fma(f32x2{hwr[m][e >> 2][e & 3], hwr[m][e >> 2][(e & 3) + 1]}, f32x2{hv, hv}, v2[o2]);
                        }
                    }
            }
#pragma unroll
            for (int o = 0; o < 6; ++o) vals[t * 6 + o] = v2[o >> 1][o & 1];
        }
    }
    float mine[6 * T / 4];
    reduce_scatter<T>(vals, mine);
    const int t0 = (T / 2) * (g & 1) + (T / 4) * (g >> 1);
#pragma unroll
    for (int i = 0; i < 6 * T / 4; ++i)
        if (t0 + i / 6 < NT) shares[((t0 + i / 6) * 4 + wave) * 96 + (i % 6) * 16 + n] = mine[i];
}

// ---- what the pass loop of a leaf needs (one struct so that the two kernels below share the loop)
struct Leaf {
    f16x4 cell_planes;        // the four planes of this thread's cell
    const char *base;         // the base of the leaf's parity (mode 1: the base being built)
    int dys[kMaxD], dxs[kMaxD];   // the changed cells (-100: none)
    _Float16 *dst16;          // the policy pieces' place in the store (NULL: none)
    float *vdst;              // the value head's input row (global memory, or LDS in the resident search)
    float *dst32;             // f32 features (tests; NULL otherwise)
    bool deferred, store_head;
};
struct Layers {
    float k1, k2, k3, act1, act2, act3;
    const char *t2p, *t3p;
};
#ifdef RZ_NET_PROFILE
struct Prof { long long acc[24], t; };
#else
struct Prof {};
#endif

// once per launch: every map entry -> the zero record, every list entry -> a valid position (cell 0)
__device__ __forceinline__ void init_maps(char *lds, int tid) {
    uint32_t *map1 = reinterpret_cast<uint32_t *>(lds + kOffMap1);
    const uint32_t zaddr = lds_addr(lds + kOffZero);
    for (int i = tid; i < 2 * kGrid; i += 256) map1[i] = zaddr;    // (map1 and map2 are contiguous)
    if (tid < 3 * 128 / 2) reinterpret_cast<uint32_t *>(lds + kOffList)[tid] = 19u | (19u << 16);
}

// The passes of one leaf: -1 = against the base (use_delta), 0 .. 3 = the board's quadrants without one.  -> tiles computed: conv3 | conv2 << 16.
__device__ __forceinline__ int delta_passes(const NetDev &nd, const DeltaArgs &da, char *lds, int tid0, int wave, const Leaf &leaf, const Layers &ly,
                                            f32x4 headv, bool &use_delta, Prof &prof) {
    char *in0 = lds, *c1 = lds + kOffC1, *c2 = lds + kOffC2, *zrec = lds + kOffZero, *headw = lds + kOffHead;
    uint32_t *map1 = reinterpret_cast<uint32_t *>(lds + kOffMap1), *map2 = reinterpret_cast<uint32_t *>(lds + kOffMap2);
    uint16_t *list1 = reinterpret_cast<uint16_t *>(lds + kOffList), *list2 = list1 + 128, *list3 = list2 + 128;
    int *cnt = reinterpret_cast<int *>(lds + kOffCnt);   // [5 sets][4 waves]
    float *shares = reinterpret_cast<float *>(c1);
    const int mode = da.mode, BH = nd.BH, BW = nd.BW, S = nd.S;
    const float k1 = ly.k1, k2 = ly.k2, k3 = ly.k3, act1 = ly.act1, act2 = ly.act2, act3 = ly.act3;
    const char *t2p = ly.t2p, *t3p = ly.t3p;
    const char *base = leaf.base;
    const f16x4 cell_planes = leaf.cell_planes;
    const int (&dys)[kMaxD] = leaf.dys, (&dxs)[kMaxD] = leaf.dxs;
    _Float16 *dst16 = leaf.dst16;
    float *vdst = leaf.vdst, *dst32 = leaf.dst32;
    const bool deferred = leaf.deferred, store_head = leaf.store_head;
    const uint32_t zaddr = lds_addr(zrec);
    int n_conv3_tiles = 0, n_conv2_tiles = 0;
#ifdef RZ_NET_PROFILE
    long long (&prof_acc)[24] = prof.acc, &prof_t = prof.t;
#endif
    for (int pass = use_delta ? -1 : 0; pass < 4; ++pass) {
        // (the thread's number is opaque per pass: hipcc otherwise hoists every lane-dependent address of the pass -- of all its
        // instantiations -- out of this loop and spills them: trunk_rows_body's board loop does the same)
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, g = lane >> 4;
        const int cy = (tid * da.bw_rcp) >> 16, cx = tid - cy * BW;
        const bool is_cell = tid < S;
        const int mypos = is_cell ? (cy + 1) * kRowW + cx + 1 : 0;
        if (RZ_DELTA_PRO_PRIO) __builtin_amdgcn_s_setprio(RZ_DELTA_PRO_PRIO);
        // ---- the distance of this thread's cell to the changed cells (delta) / to the pass's quadrant
        int dist = 1000;
        if (pass < 0) {
#pragma unroll
            for (int i = 0; i < kMaxD; ++i) {
                const int ady = cy > dys[i] ? cy - dys[i] : dys[i] - cy, adx = cx > dxs[i] ? cx - dxs[i] : dxs[i] - cx;
                const int d = ady > adx ? ady : adx;
                dist = d < dist ? d : dist;
            }
        } else {
            const int y0 = 8 * (pass >> 1), x0 = 8 * (pass & 1);
            const int y1 = (y0 + 8 < BH ? y0 + 8 : BH) - 1, x1 = (x0 + 8 < BW ? x0 + 8 : BW) - 1;   // (inclusive)
            const int ddy = cy < y0 ? y0 - cy : (cy > y1 ? cy - y1 : 0), ddx = cx < x0 ? x0 - cx : (cx > x1 ? cx - x1 : 0);
            dist = ddy > ddx ? ddy : ddx;
        }
        // sets: 0 = conv1 computes, 1 = conv2 computes, 2 = conv3 computes, 3 = conv1 records held, 4 = conv2 records held
        const int th0 = pass < 0 ? 1 : 2, th1 = pass < 0 ? 2 : 1, th2 = pass < 0 ? 3 : 0, th3 = pass < 0 ? 3 : 2, th4 = pass < 0 ? 4 : 1;
        const bool f[5] = {is_cell && dist <= th0, is_cell && dist <= th1, is_cell && dist <= th2, is_cell && dist <= th3, is_cell && dist <= th4};
        // the records conv2 / conv3 read but this leaf does not recompute come from the base: requested now, stored behind the barrier
        // (their places are ranks over the whole workgroup)
        const bool g1 = pass < 0 && f[3] && !f[0], g2 = pass < 0 && f[4] && !f[1];
        // (their lines are TOUCHED now -- one dword of each 128-byte line, so that the copies behind the barrier find them in L2; the
        // copies themselves as registers across the barrier cost 60-110 spilled registers, however the loads were placed)
        // (conv1's weights and the biases: requested per pass, so that nothing of them is live across conv3)
        sp::f16x8 a1[3][2];
        f32x4 bias1[4];
        if (32 * wave < 32 * kT1) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int p_ = 0; p_ < 2; ++p_) a1[ky][p_] = __builtin_bit_cast(sp::f16x8, nd.s1[(ky * 2 + p_) * 64 + lane]);
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) bias1[gg] = *reinterpret_cast<const f32x4 *>(nd.b1 + 8 * gg + 4 * (lane >> 5));
        }
        f32x4 bias2 = *reinterpret_cast<const f32x4 *>(nd.b2 + 16 * wave + 4 * g);
        // (unconditional loads -- every thread's number is a valid index of the base's arrays: a load under a branch makes hipcc wait
        // for every outstanding load first)
        const float touch = *reinterpret_cast<const float *>(base + (size_t)tid * 128) + *reinterpret_cast<const float *>(base + kBaseC2 + (size_t)tid * 256) +
                            *reinterpret_cast<const float *>(base + kBaseC2 + (size_t)tid * 256 + 128);
        float bv[6];   // ... and the features of a cell outside conv3's window
#pragma unroll
        for (int o = 0; o < 6; ++o) bv[o] = reinterpret_cast<const float *>(base + kBaseV)[o * kCells + tid];
        int rank[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const unsigned long long b = __ballot(f[k]);
            rank[k] = __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
            if (lane == 0) cnt[k * 4 + wave] = __popcll(b);
        }
        // (maps and lists: every map entry is the zero record and every list entry a valid position when a pass begins -- set once per
        // launch, init_maps; a pass puts back what it changed, and a lane past a list's count may read any position: its column is dropped)
        NET_TICK(1);   // distances, requests, ballots, map / list defaults
        __syncthreads();
        NET_TICK(2);
        int tot[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            int before = 0, all = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int c = cnt[k * 4 + w];
                before += w < wave ? c : 0;
                all += c;
            }
            rank[k] += before;
            tot[k] = all;
        }
        if (pass < 0 && (tot[0] > 32 * kT1 || tot[1] > 16 * kT2 || tot[2] > 16 * kT3 || tot[3] > kC1Slots || tot[4] > kC2Slots)) {
            use_delta = false;   // (uniform) the windows exceed the budget: the four passes without a base
            __syncthreads();     // (cnt is rewritten)
            continue;
        }
        uint32_t rec1 = 0, rec2 = 0;
        if (f[3]) {
            rec1 = lds_addr(c1 + rank[3] * P1);
            map1[mypos] = rec1;
        }
        if (f[4]) {
            rec2 = lds_addr(c2 + rank[4] * P2);
            map2[mypos] = rec2;
        }
        if (f[0]) list1[rank[0]] = (uint16_t)mypos;
        if (f[1]) list2[rank[1]] = (uint16_t)mypos;
        if (f[2]) list3[rank[2]] = (uint16_t)mypos;
        if (pass <= 0 && is_cell) *reinterpret_cast<f16x4 *>(in0 + ((cy + 1) * sp::kInCols + (cx + 1)) * 8) = cell_planes;
        if (pass <= 0 && store_head && tid < 226) reinterpret_cast<f32x4 *>(headw)[tid] = headv;
        asm volatile("" ::"v"(touch));
        if (g1) {
            const f32x4 *src = reinterpret_cast<const f32x4 *>(base + (size_t)tid * 128);
            f32x4 r1[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) r1[i] = src[i];
#pragma unroll
            for (int i = 0; i < 8; ++i) *(lds_v4)(uintptr_t)(rec1 + 16 * i) = r1[i];
        }
        if (g2) {
            const f32x4 *src = reinterpret_cast<const f32x4 *>(base + kBaseC2 + (size_t)tid * 256);
            f32x4 r2[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) r2[i] = src[i];
#pragma unroll
            for (int i = 0; i < 16; ++i) *(lds_v4)(uintptr_t)(rec2 + 16 * i) = r2[i];
        }
        __builtin_amdgcn_sched_barrier(0);   // (the requests below stay behind the stores of the base's records: their 96 registers are free again)
        f16x8 a2[9][1][2];   // conv2's weight fragments: requested here, conv1 -- one wave's work -- covers their latency
        {
            const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(t2p), 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int p = 0; p < 2; ++p) a2[tap][0][p] = sp::load_w(w_rsrc, lane * 16, (tap * 2 + p) * 1024);
        }
        NET_TICK(3);   // ranks, maps, lists, planes, the base's records
        __syncthreads();
        NET_TICK(4);

        // ---- conv1 (4 -> 32): tile w of 32 cells by wave w (trunk_rows_body's conv1 with the lane's cell taken from the list)
        if (32 * wave < tot[0]) {
            const int n32 = lane & 31, h = lane >> 5, idx = 32 * wave + n32;
            const int pos = list1[idx];
            const int py = (pos * 3641) >> 16, px = pos - py * kRowW;   // halo row / column (board row py - 1, column px - 1)
            typedef const __attribute__((address_space(3))) f16x4 *lds_half;
            const lds_half q = (lds_half)(in0 + ((py - 1) * sp::kInCols + (px - 1) + 2 * h) * 8);
            sp::f16x8 b1[3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int o = ky * sp::kInCols;
                const f16x4 lo4 = q[o], hi4 = q[o + 1];
                b1[ky] = __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            sp::f32x16 acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[r] = 0.0f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int combo = 0; combo < 3; combo += 2)   // (the lo pieces of 0 / 1 planes are zero: no hi x lo product)
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[ky][combo == 2], b1[ky], acc1, 0, 0, 0);
            if (idx < tot[0]) {
                const uint32_t rec = map1[pos] + (uint32_t)(4 * h * 2);
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) {
                    float z[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) z[j] = fmaxf(fmaf(acc1[4 * gg + j], k1, bias1[gg][j] * act1), 0.0f);
                    f16x4 hi, lo;
                    sp::split4(z, hi, lo);
                    *(lds_h4)(uintptr_t)(rec + 8 * gg * 2) = hi;
                    *(lds_h4)(uintptr_t)(rec + 8 * gg * 2 + 64) = lo;
                }
            }
        }
        NET_TICK(5);   // conv1
        __syncthreads();
        NET_TICK(6);
        if (mode == 1 && f[0]) {   // the base keeps conv1's records
            f32x4 *dstp = reinterpret_cast<f32x4 *>(const_cast<char *>(base) + (size_t)tid * 128);
#pragma unroll
            for (int i = 0; i < 8; ++i) dstp[i] = *(lds_v4)(uintptr_t)(rec1 + 16 * i);
        }

        if (RZ_DELTA_PRO_PRIO) __builtin_amdgcn_s_setprio(0);   // (the matrix loops: behind the other workgroup's latency chains at issue)
        // ---- conv2 (32 -> 64): wave w = output channels 16 w .. 16 w + 15 at every tile
        const int nt3 = (tot[2] + 15) >> 4;
        {
            const lds_u32 m1 = (lds_u32)map1, m2 = (lds_u32)map2;
            const int nt2 = (tot[1] + 15) >> 4;
            n_conv2_tiles += nt2;
            for (int t0 = 0; t0 < nt2; t0 += 4) {   // groups of up to four tiles (the weights are in registers: a group costs no fetch)
                const uint16_t *l2 = list2 + 16 * t0;
                const int left = tot[1] - 16 * t0;
                switch (nt2 - t0) {
                    case 1: conv2_tiles<1>(m1, m2, l2, left, lane, wave, a2, bias2 * act2, k2); break;
                    case 2: conv2_tiles<2>(m1, m2, l2, left, lane, wave, a2, bias2 * act2, k2); break;
                    case 3: conv2_tiles<3>(m1, m2, l2, left, lane, wave, a2, bias2 * act2, k2); break;
                    default: conv2_tiles<4>(m1, m2, l2, left, lane, wave, a2, bias2 * act2, k2); break;
                }
            }
        }
        NET_TICK(7);   // conv2

        // ---- conv3 (64 -> 128) + the head convolutions: wave w = output channels 32 w .. 32 w + 31 at every tile
        {
            const lds_u32 m2 = (lds_u32)map2;
            n_conv3_tiles += nt3;
            // up to six tiles in one go; seven or eight as four + the rest (the weights are fetched again: rare, big leaves only)
            const int first = nt3 <= 6 ? nt3 : 4;
            switch (first) {
                case 0: __syncthreads(); break;   // (the barrier behind conv2, which the other cases meet inside conv3_g)
                case 1: conv3_tiles<1>(m2, list3, t3p, lane, wave, true, headw, k3, shares); break;
                case 2: conv3_tiles<2>(m2, list3, t3p, lane, wave, true, headw, k3, shares); break;
                case 3: conv3_tiles<3>(m2, list3, t3p, lane, wave, true, headw, k3, shares); break;
                case 4: conv3_tiles<4>(m2, list3, t3p, lane, wave, true, headw, k3, shares); break;
                case 5: conv3_tiles<5>(m2, list3, t3p, lane, wave, true, headw, k3, shares); break;
                default: conv3_tiles<6>(m2, list3, t3p, lane, wave, true, headw, k3, shares); break;
            }
            if (nt3 == 7) conv3_tiles<3>(m2, list3 + 64, t3p, lane, wave, false, headw, k3, shares + 4 * 4 * 96);
            else if (nt3 >= 8) conv3_tiles<4>(m2, list3 + 64, t3p, lane, wave, false, headw, k3, shares + 4 * 4 * 96);
        }
        NET_TICK(9);   // the barrier behind conv2, conv3 + heads
        __syncthreads();
        NET_TICK(10);
        if (mode == 1 && f[1]) {   // the base keeps conv2's records too
            f32x4 *dstp = reinterpret_cast<f32x4 *>(const_cast<char *>(base) + kBaseC2 + (size_t)tid * 256);
#pragma unroll
            for (int i = 0; i < 16; ++i) dstp[i] = *(lds_v4)(uintptr_t)(rec2 + 16 * i);
        }

        // ---- the features: thread = cell adds the waves' shares in wave order (a cell outside the windows: the base's value)
        if (is_cell && (pass < 0 || f[2])) {
            const float *share = shares + (rank[2] >> 4) * 4 * 96 + (rank[2] & 15);
            float *basev = mode == 1 ? reinterpret_cast<float *>(const_cast<char *>(base) + kBaseV) : nullptr;
#pragma unroll
            for (int o = 0; o < 6; ++o) {
                float v;
                if (f[2]) {
                    float s = share[o * 16];
#pragma unroll
                    for (int w = 1; w < 4; ++w) s += share[w * 96 + o * 16];
                    v = fmaxf(s + reinterpret_cast<const float *>(headw + 128 * 7 * 4)[o], 0.0f);
                } else {
                    v = bv[o];
                }
                if (mode == 1) {
                    basev[o * kCells + tid] = v;
                    continue;
                }
                if (dst32) dst32[o * S + tid] = v;
                if (deferred && o >= 4) {
                    vdst[(o - 4) * S + tid] = v;
                } else if (dst16) {
                    const int k = (o < 4 ? o : o - 4) * S + tid;
                    const int step = (o < 4 ? 0 : nd.groups_act) + (k >> 4);
                    const float z = v * act3;
                    const _Float16 zh = (_Float16)z;
                    _Float16 *q = dst16 + (size_t)step * 1024 + (k & 15);
                    q[0] = zh;
                    q[512] = (_Float16)(z - (float)zh);
                }
            }
        }
        if (f[3]) map1[mypos] = zaddr;   // (every wave is behind conv3: nothing reads the maps any more)
        if (f[4]) map2[mypos] = zaddr;
        NET_TICK(11);   // features
        if (pass < 0) break;
        __syncthreads();   // the next pass rewrites maps, lists and records
    }
    return n_conv3_tiles | (n_conv2_tiles << 16);
}

// the four planes of thread `tid`'s cell (load_bits of trunk_rows_body; a base: no last move, the parity's stone count)
__device__ __forceinline__ f16x4 planes_of(const uint64_t (&ls)[8], int tid, int tm_eff, bool has_last, int lc, int par_count) {
    f16x4 cell_planes;
    const int word = (tid >> 6) & 3, bit = tid & 63;
    uint64_t w0 = 0ull, w1 = 0ull;   // (masks, not a select of array elements: rz_tree.h's word_of)
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const uint64_t mk = word == w ? ~0ull : 0ull;
        w0 |= ls[w] & mk;
        w1 |= ls[4 + w] & mk;
    }
    const bool s0 = (w0 >> bit) & 1ull, s1 = (w1 >> bit) & 1ull;
    const bool mine = tm_eff == 0 ? s0 : s1, theirs = tm_eff == 0 ? s1 : s0;
    const _Float16 one = (_Float16)sp::kObsScale, zero = (_Float16)0.0f;
    cell_planes[0] = mine ? one : zero;
    cell_planes[1] = theirs ? one : zero;
    cell_planes[2] = (has_last && tid == lc) ? one : zero;
    cell_planes[3] = (par_count & 1) ? zero : one;
    return cell_planes;
}

// delta or not: the base's stones (rs) must be a subset of the leaf's (ls), colour by colour; the changed cells D = the added stones and
// the last move (any superset of the cells whose planes differ from the base's is correct).  Selects, no branches.  -> use_delta
__device__ __forceinline__ bool changed_cells(const uint64_t (&ls)[8], const uint64_t (&rs)[8], int nst, int tm, int lc, bool has_last, bool base_ok, int h_tm,
                                              int bw_rcp, int BW, int (&dys)[kMaxD], int (&dxs)[kMaxD], int &parity, int &nD) {
    // ---- delta or not (mode 0): the base's stones must be a subset of the leaf's, colour by colour; the changed cells D = the added
    // stones and the last move (any superset of the cells whose planes differ from the base's is correct).  Selects, no branches.
    bool use_delta = false;
    {
        bool sup = true;
        int nroot = 0;
        uint64_t D[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            sup = sup && (rs[q] & ~ls[q]) == 0ull;
            nroot += __popcll(rs[q]);
            D[q & 3] |= rs[q] ^ ls[q];
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) D[w] |= (has_last && lc >= 0 && (lc >> 6) == w) ? 1ull << (lc & 63) : 0ull;
        parity = (nst - nroot) & 1;
        nD = __popcll(D[0]) + __popcll(D[1]) + __popcll(D[2]) + __popcll(D[3]);
        use_delta = base_ok && sup && tm == (h_tm ^ parity) && nD <= kMaxD;
#pragma unroll
        for (int i = 0; i < kMaxD; ++i) {
            const bool z0 = D[0] == 0ull, z1 = D[1] == 0ull, z2 = D[2] == 0ull;
            const int w = !z0 ? 0 : !z1 ? 1 : !z2 ? 2 : 3;
            const uint64_t word = !z0 ? D[0] : !z1 ? D[1] : !z2 ? D[2] : D[3];
            const bool any = word != 0ull;
            const int cell = 64 * w + __builtin_ctzll(word | (1ull << 63));
            const uint64_t rest = word & (word - 1ull);
#pragma unroll
            for (int k = 0; k < 4; ++k) D[k] = k == w ? rest : D[k];
            const int y = (cell * bw_rcp) >> 16;
            dys[i] = any ? y : -100;
            dxs[i] = any ? cell - y * BW : -100;
        }
    }
    return use_delta;
}

template <bool TRACE>
__global__ __launch_bounds__(256, 2) void k_trunk_delta(NetDev nd, LeafBits leaves, _Float16 *__restrict__ feat16, int n_boards,
                                                        DeferredOut later, DeltaArgs da) {
    __shared__ __attribute__((aligned(16))) char lds[kLdsBytes];
    // (the grid is exactly n_boards workgroups.  Everything up to the first use of a loaded value is ONE block of unconditional
    // requests -- kernel arguments, the position, the header, the constants: a conditional load or an early return costs a memory
    // round trip of its own in the chain kernel arguments -> position -> windows -> base records, which is what a leaf waits for)
    const int board = blockIdx.x;
    const int mode = da.mode;
    const int game = mode == 1 ? board >> 1 : board, par_b = mode == 1 ? board & 1 : 0;
    unsigned long long trace_t0 = 0;
    if (TRACE) trace_t0 = rz_trace_now();
    Prof prof;
#ifdef RZ_NET_PROFILE
    for (int i = 0; i < 24; ++i) prof.acc[i] = 0;
    prof.t = __builtin_readcyclecounter();
    const long long prof_k0 = prof.t;
    long long (&prof_acc)[24] = prof.acc, &prof_t = prof.t;
#endif
    const int tid0 = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int BW = nd.BW, S = nd.S;
    Layers ly;
    ly.t2p = reinterpret_cast<const char *>(nd.t2) + (size_t)wave * rt::Geo<32>::steps * 2 * 1024;
    ly.t3p = reinterpret_cast<const char *>(nd.t3) + (size_t)(2 * wave) * rt::Geo<64>::steps * 2 * 1024;
    // the head convolutions' weights: requested now, stored to LDS behind the first barrier
    f32x4 headv = tid0 < 192 ? reinterpret_cast<const f32x4 *>(nd.whp)[tid0] : reinterpret_cast<const f32x4 *>(nd.b3)[(tid0 - 192) & 31];
    {   // (threads 224 / 225: the six head biases, two floats of padding)
        const float *bh = nd.bh + (tid0 == 225 ? 4 : 0);
        const f32x4 hbv = {bh[0], bh[1], tid0 == 225 ? 0.0f : bh[2], tid0 == 225 ? 0.0f : bh[3]};
        headv = (tid0 == 224 || tid0 == 225) ? hbv : headv;
    }
    ly.k1 = nd.s_inv[2], ly.k2 = nd.s_inv[0], ly.k3 = nd.s_inv[1];
    ly.act1 = nd.s_inv[5], ly.act2 = nd.s_inv[6], ly.act3 = nd.s_inv[7];

    // ---- the position (wave-uniform: scalar loads), the header of the game's bases (read in every mode, used in mode 0), the flag
    const uint64_t *sb = leaves.stones + (size_t)game * 8;
    const BaseHdr *hd = da.hdr + game;
    uint64_t ls[8], rs[8];
    int nst = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        ls[q] = sb[q];
        rs[q] = hd->stones[q];
        nst += __popcll(ls[q]);
    }
    const int tm = leaves.to_move[game];
    const int lc_raw = (mode == 1 ? leaves.to_move : leaves.last)[game];   // (mode 1 has no last-move array: any readable word)
    const int lc = mode == 1 ? -1 : lc_raw;
    const int h_tm = hd->to_move, h_valid = hd->valid;
    const int is_active = da.active[game];
    const bool deferred = later.slot_of != nullptr;
    const int slot_ = (deferred ? later.slot_of : leaves.to_move)[deferred ? board : game];
    const int tm_eff = tm ^ par_b;        // mode 1, parity 1: the root seen by the other side, one stone later
    const int par_count = nst + par_b;
    const bool has_last = mode != 1 && nst > 0;

    Leaf leaf;
    int parity = 0, nD = 0;
    bool use_delta = changed_cells(ls, rs, nst, tm, lc, has_last, mode == 0 && h_valid != 0, h_tm, da.bw_rcp, BW, leaf.dys, leaf.dxs, parity, nD);
    leaf.base = da.recs + ((size_t)game * 2 + (mode == 1 ? par_b : parity)) * kBaseBytes;
    leaf.cell_planes = planes_of(ls, tid0, tm_eff, has_last, lc, par_count);

    // ---- where the features go (trunk_rows_body's feature stage)
    leaf.dst16 = nullptr;
    leaf.vdst = nullptr;
    if (mode != 1 && deferred) {
        leaf.dst16 = (feat16 != nullptr && slot_ < later.n_slots) ? feat16 + (size_t)slot_ * later.slot_halfs + (size_t)(board >> 5) * nd.groups_act * 1024 + (board & 31) * 16 : nullptr;
        leaf.vdst = later.valfeat + (size_t)board * later.vf_ld;
    }
    if (mode != 1 && !deferred && feat16 != nullptr)   // rz_net_delta_trunk_engine: the FC GEMM's own tiles (policy and value K-steps), as trunk_rows_body writes them
        leaf.dst16 = feat16 + ((size_t)(board >> 5) * (nd.groups_act + nd.groups_val) * 1024 + (board & 31) * 16);
    leaf.dst32 = (mode != 1 && da.feat32 != nullptr) ? da.feat32 + (size_t)board * 6 * S : nullptr;
    leaf.deferred = deferred;
    leaf.store_head = true;

    {   // once per leaf: the planes' halo, the zero record; the planes themselves and the head weights behind the first barrier
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        f32x4 *z = reinterpret_cast<f32x4 *>(lds);
        for (int i = tid0; i < sp::kInPieceBytes / 16; i += 256) z[i] = zero;
        if (tid0 < P2 / 16) reinterpret_cast<f32x4 *>(lds + kOffZero)[tid0] = zero;
        init_maps(lds, tid0);
    }
    if (mode == 0 && is_active == 0) return;   // (uniform; before any barrier; behind the stores above so that the flag's load is
                                               // one of the batch, not a round trip of its own at the top)
    NET_TICK(0);   // requests, scalars, header, planes, zeroing
    const int n_tiles = delta_passes(nd, da, lds, tid0, wave, leaf, ly, headv, use_delta, prof);
    const int n_conv3_tiles = n_tiles & 0xffff;

    if (mode == 1 && par_b == 0 && tid0 == 0) {   // the header of this game's bases (the launch behind this one reads it)
        BaseHdr *h = da.hdr + game;
#pragma unroll
        for (int q = 0; q < 8; ++q) h->stones[q] = sb[q];
        h->to_move = tm;
        h->valid = 1;
    }
    if (da.stats != nullptr && tid0 == 0 && mode != 1) {
        atomicAdd(da.stats + (use_delta ? 0 : 1), 1u);
        atomicAdd(da.stats + 2, (unsigned)n_conv3_tiles);
        atomicAdd(da.stats + 3, (unsigned)nD);
        atomicAdd(da.stats + 4, (unsigned)(n_tiles >> 16));
    }
#ifdef RZ_NET_PROFILE
    if (blockIdx.x == 0 && tid0 == 0) {
        for (int i = 0; i < 24; ++i) net_prof[i] = prof_acc[i];
        net_prof[23] = __builtin_readcyclecounter() - prof_k0;
        net_prof[22] = n_conv3_tiles;
    }
#endif
    if (TRACE && tid0 == 0) rz_trace_write(later.trace, RZ_TRACE_TRUNK, deferred ? slot_ : 0, board, trace_t0);
}

// RESIDENT SEARCH with receptive-field evaluation (rz_net_search_resident on boards of 11 .. 16 rows and columns once the base cache
// exists): k_trunk_rows_res's loop -- leaf -> trunk -> value head -> expand / backup -> next selection, n_sims times in ONE launch, one
// workgroup per game, the leaf handed over through LDS, the tree code the engine's own (rz_tree.h) -- with the delta passes as the
// trunk.  82 KB of LDS and 256 registers: TWO games per CU, so the serial tree walk of one game (one wave, ~6 us a simulation) runs
// under the other game's matrix work, and a batch of 2 x CUs games (the 512 per GPU of BASELINE.json configs[3]) is ONE launch per
// search on one stream: no lanes, no hardware-queue layout, no kernel boundary inside a search.  The bases of the roots are built
// by the launch before this one (rz_net_search_resident); the header is compared with the root once per search.
constexpr int kResVrow = 512;
__global__ __launch_bounds__(256, 2) void k_delta_res(NetDev nd, _Float16 *__restrict__ store16, DeferredOut later, DeltaArgs da, ResArgs<true> res) {
    __shared__ __attribute__((aligned(16))) char lds[kLdsBytes];
    __shared__ float res_vrow[kResVrow];
    __shared__ float res_part[rzt::kDefWaves][rzt::kWave];
    __shared__ __attribute__((aligned(16))) uint64_t res_leaf[2 * RZ_BOARD_WORDS + 2];
    const int game = blockIdx.x;
    if (game >= res.E.n_games || res.E.active[game] == 0) return;   // (uniform: before any barrier)
    const unsigned long long clk0 = __builtin_readcyclecounter(), rt0 = __builtin_amdgcn_s_memrealtime();
    const int tid0 = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int BW = nd.BW;
    Prof prof;
#ifdef RZ_NET_PROFILE
    for (int i = 0; i < 24; ++i) prof.acc[i] = 0;
    prof.t = __builtin_readcyclecounter();
    const long long prof_k0 = prof.t;
    long long (&prof_acc)[24] = prof.acc, &prof_t = prof.t;
#endif
    Layers ly;
    ly.t2p = reinterpret_cast<const char *>(nd.t2) + (size_t)wave * rt::Geo<32>::steps * 2 * 1024;
    ly.t3p = reinterpret_cast<const char *>(nd.t3) + (size_t)(2 * wave) * rt::Geo<64>::steps * 2 * 1024;
    f32x4 headv = tid0 < 192 ? reinterpret_cast<const f32x4 *>(nd.whp)[tid0] : reinterpret_cast<const f32x4 *>(nd.b3)[(tid0 - 192) & 31];
    {
        const float *bh = nd.bh + (tid0 == 225 ? 4 : 0);
        const f32x4 hbv = {bh[0], bh[1], tid0 == 225 ? 0.0f : bh[2], tid0 == 225 ? 0.0f : bh[3]};
        headv = (tid0 == 224 || tid0 == 225) ? hbv : headv;
    }
    ly.k1 = nd.s_inv[2], ly.k2 = nd.s_inv[0], ly.k3 = nd.s_inv[1];
    ly.act1 = nd.s_inv[5], ly.act2 = nd.s_inv[6], ly.act3 = nd.s_inv[7];
    const int slot0 = res.E.pend[game];
    // the root (it does not move inside a search) against the header of the game's bases: once
    uint64_t rs[8];
    bool base_ok = da.hdr[game].valid != 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        rs[q] = res.E.root_stones[(size_t)game * 8 + q];
        base_ok = base_ok && da.hdr[game].stones[q] == rs[q];
    }
    const int h_tm = res.E.root_to_move[game];
    base_ok = base_ok && da.hdr[game].to_move == h_tm;
    {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        f32x4 *z = reinterpret_cast<f32x4 *>(lds);
        for (int i = tid0; i < sp::kInPieceBytes / 16; i += 256) z[i] = zero;
        if (tid0 < P2 / 16) reinterpret_cast<f32x4 *>(lds + kOffZero)[tid0] = zero;
        init_maps(lds, tid0);
        for (int i = tid0; i < kResVrow; i += 256) res_vrow[i] = 0.0f;
    }
    if (res.select_first == 0 && tid0 == 0) {   // the first leaf was selected by rz_select_step: from the engine's leaf arrays
#pragma unroll
        for (int q = 0; q < 8; ++q) res_leaf[q] = res.E.leaf_stones[(size_t)game * 8 + q];
        reinterpret_cast<int *>(res_leaf + 2 * RZ_BOARD_WORDS)[0] = res.E.leaf_to_move[game];
        reinterpret_cast<int *>(res_leaf + 2 * RZ_BOARD_WORDS)[1] = res.E.leaf_last[game];
    }
    __syncthreads();
    if (res.select_first != 0) {   // AlphaZeroMCTS._playout's select loop for the first simulation of the search (rz_select_step's work)
        if (wave == 0) rzt::select_body<false>(res.E, nullptr, game, tid0 & 63, 0, res_leaf);
        __syncthreads();
    }
    int tiles_total = 0, tiles2_total = 0, deltas = 0, cells_total = 0;
    for (int sim = 0; sim < res.n_sims; ++sim) {
        // (the thread's number is opaque per simulation: hipcc otherwise hoists the tree code's lane-dependent addresses out of this
        // loop and spills them)
        int tid_s = tid0;
        asm volatile("" : "+v"(tid_s));
        // ---- the leaf, from LDS (select_body's lds_leaf): wave-uniform values
        uint64_t ls[8];
        int nst = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint64_t v = res_leaf[q];
            ls[q] = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v);
            nst += __popcll(ls[q]);
        }
        const int tm = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int *>(res_leaf + 2 * RZ_BOARD_WORDS)[0]);
        const int lc = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int *>(res_leaf + 2 * RZ_BOARD_WORDS)[1]);
        const bool has_last = nst > 0;
        Leaf leaf;
        int parity = 0, nD = 0;
        bool use_delta = changed_cells(ls, rs, nst, tm, lc, has_last, base_ok, h_tm, da.bw_rcp, BW, leaf.dys, leaf.dxs, parity, nD);
        leaf.base = da.recs + ((size_t)game * 2 + parity) * kBaseBytes;
        leaf.cell_planes = planes_of(ls, tid_s, tm, has_last, lc, nst);
        // the game's slot advances by one per simulation (expand_backup_body<DEF>); the value inputs stay in LDS
        leaf.dst16 = slot0 + sim < later.n_slots ? store16 + (size_t)(slot0 + sim) * later.slot_halfs + (size_t)(game >> 5) * nd.groups_act * 1024 + (game & 31) * 16 : nullptr;
        leaf.vdst = res_vrow;
        leaf.dst32 = nullptr;
        leaf.deferred = true;
        leaf.store_head = sim == 0;
        NET_TICK(0);
        const int n_tiles = delta_passes(nd, da, lds, tid_s, wave, leaf, ly, headv, use_delta, prof);
        tiles_total += n_tiles & 0xffff;
        tiles2_total += n_tiles >> 16;
        deltas += use_delta ? 1 : 0;
        cells_total += nD;
        __syncthreads();   // the value head's inputs are complete; the shares are read
        NET_TICK(12);
        // ---- the rest of the simulation, by the same workgroup (k_tree_step_def's body: rz_tree.h), as in trunk_rows_body
        const int lane = tid_s & 63;
        if (RZ_DELTA_TREE_PRIO) __builtin_amdgcn_s_setprio(RZ_DELTA_TREE_PRIO);   // (the serial part of a simulation: ahead of the other game's trunk waves at issue)
        if (res.vh.groups == 128) rzt::value_quarter_lds<16>(res.vh, res_vrow, lane, wave, res_part);
        else rzt::value_quarter_lds<8>(res.vh, res_vrow, lane, wave, res_part);
        NET_TICK(16);
        if (wave == 0) rzt::expand_backup_body<float, false, false, false, true>(res.E, nullptr, nullptr, game, lane, rz_raw_heads(), 0, res.vh, res_part);
        else __syncthreads();   // (the barrier inside the body, where the quarters meet)
        __syncthreads();        // the tree's updates before the selection's loads
        NET_TICK(17);
        const bool more = sim + 1 < res.n_sims;
        if (wave == 0 && more) rzt::select_body<false>(res.E, nullptr, game, lane, 0, res_leaf);
        if (RZ_DELTA_TREE_PRIO) __builtin_amdgcn_s_setprio(0);
        __syncthreads();
        NET_TICK(18);
    }
    if (da.stats != nullptr && tid0 == 0) {
        atomicAdd(da.stats + 0, (unsigned)deltas);
        atomicAdd(da.stats + 1, (unsigned)(res.n_sims - deltas));
        atomicAdd(da.stats + 2, (unsigned)tiles_total);
        atomicAdd(da.stats + 3, (unsigned)cells_total);
        atomicAdd(da.stats + 4, (unsigned)tiles2_total);
        if (game == 0) {   // the clock this search ran at: cycles / (ticks x 10 ns)
            da.stats[6] = (unsigned)((__builtin_readcyclecounter() - clk0) >> 8);
            da.stats[7] = (unsigned)(__builtin_amdgcn_s_memrealtime() - rt0);
        }
    }
#ifdef RZ_NET_PROFILE
    if (blockIdx.x == 0 && tid0 == 0) {
        for (int i = 0; i < 24; ++i) net_prof[i] = prof_acc[i];
        net_prof[23] = __builtin_readcyclecounter() - prof_k0;
        net_prof[22] = tiles_total;
    }
#endif
}

}  // namespace dl
