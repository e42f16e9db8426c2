// rz_engine.hip -- MI355X (gfx950 / CDNA4) AlphaZero self-play MCTS engine.
//
// One 64-lane wavefront per game, one simulation in flight per tree (the reference runs
// its simulations strictly sequentially, rlzero/mcts/alphazero_mcts.py:82-85, so this is
// what bit-exact parity requires); the parallelism is across the lock-stepped games.
//
// Tree layout (HBM, one arena pair per game).  A node is ONE 32-byte record (two int4):
//   lo = {N   visit count                        (TreeNode.explore_count, rlzero/mcts/node.py:28),
//         FC  slot of the first child record, -1 = no child record yet,
//         NV  number of children already visited,
//         K | cap << 16:  K = legal moves at this node = len(TreeNode._children), 0 = not expanded;
//                         cap = child records reserved at FC}
//   hi = {Wsum float64 total value               (TreeNode.total_reward, node.py:29),
//         PB  offset of the node's block of K child priors (TreeNode.prior of each child, node.py:30),
//         the node's own prior (float bits; maintained for roots and for dense blocks)}
// so a level of the descent costs one 32-byte load per lane, all from one cache line.
// The reference's selection rule gives an unvisited child +inf and Python's max() keeps the first
// maximum (node.py:41-42,75-88), so the visited children of every node are always a PREFIX of its
// children in ascending action order: "first unvisited child" is child NV, no scan;
// a scan (coalesced, fp64 score, first-index tie-break across the wave) happens only once all K are
// visited.  Only VISITED children need a record, so the child records of a node are a contiguous
// vector [FC, FC + cap) that grows on demand (4, 8, 16 ... K: the wave copies the visited prefix to a
// new block at the top of the arena; the old block is reclaimed by the next re-root compaction).
// An 800-simulation search then touches ~1.5 k records (~50 KB per game, cache resident) instead of
// one K-slot block per expansion (~5 MB per game): the tree step is latency bound, and that
// latency is dominated by address translation and cache misses, not by instruction count.
// The K priors of an expanded node are written once, to a bump-allocated block of the separate
// prior arena (the reference's rule never reads them; the opt-in PUCT rule does).  PUCT mode
// reserves and initialises all K child records at expansion (cap = K) and always scans.
//
// Boards: two bitboards per game (4 x u64 per colour), cell = h*BW + w.  Gomoku / TicTacToe:
// action = cell (rlzero/games/gomoku/gomoku_env.py:227-234).  Connect4 (no reference
// implementation: build-defined, docs/open-spiel_alphazero.md:58 only mentions it): action =
// column, the stone drops to the lowest empty cell, row 0 is the bottom row.
//
// Bit-exactness: fp64 throughout, this file is compiled with
// -ffp-contract=off (q + c*u is two roundings in CPython), IEEE division and sqrt, and
// ln(parent N) comes from a table filled by the HOST libm -- the function CPython's
// math.log calls -- never from a device logarithm.

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "rlzero_hip.h"
#include "rz_trace.h"

#pragma clang fp contract(off)

#include "rz_tree.h"

using namespace rzt;

namespace {

__global__ __launch_bounds__(kWave) void k_select(Dev E, float *obs) {
    __builtin_amdgcn_s_setprio(3);  // latency-bound: issue ahead of a co-resident MFMA kernel
    select_body<false>(E, obs, blockIdx.x, threadIdx.x);
}

template <typename VT, bool PROBS = false>
__global__ __launch_bounds__(kWave) void k_expand_backup(Dev E, const float *logp, const VT *value) {
    __builtin_amdgcn_s_setprio(3);
    expand_backup_body<VT, PROBS>(E, logp, value, blockIdx.x, threadIdx.x);
}

__global__ __launch_bounds__(kWave) void k_expand_backup_raw(Dev E, RawHeads rh) {
    __builtin_amdgcn_s_setprio(3);
    expand_backup_body<float, false, true>(E, nullptr, nullptr, blockIdx.x, threadIdx.x, rh);
}

__global__ __launch_bounds__(kWave) void k_tree_step_raw(Dev E, RawHeads rh, float *obs) {
    __builtin_amdgcn_s_setprio(3);
    expand_backup_body<float, false, true>(E, nullptr, nullptr, blockIdx.x, threadIdx.x, rh);
    __syncthreads();
    select_body<false>(E, obs, blockIdx.x, threadIdx.x);
}

// ------------------------------------------------------------------ deferred priors (value_quarter_def: rz_tree.h)
// A value head of 4 * 8 * PER inputs covers a board of at most 16 * PER cells (deferred_ok): PER says how many words of the
// bitboards the tree code has to look at (rz_tree.h: W)
template <int PER> constexpr int words_of_per() { return PER <= 4 ? 1 : (PER == 8 ? 2 : kWords); }
template <int PER>
__global__ __launch_bounds__(kWave * kDefWaves) void k_expand_backup_def(Dev E, ValueHead vh) {
    __shared__ float part[kDefWaves][kWave];
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    value_quarter_def<PER>(vh, blockIdx.x, lane, wave, part);
    if (wave != 0) {
        __syncthreads();
        return;
    }
    expand_backup_body<float, false, false, false, true, words_of_per<PER>()>(E, nullptr, nullptr, blockIdx.x, lane, RawHeads(), 0, vh, part);
}

// (`obs` is always NULL on this route -- the trunk reads positions -- but stays a run-time argument: with the constant hipcc
// schedules the selection into 115 registers instead of 107, and 112 is what a SIMD has left beside a wave of the trunk)
// TRACE: the instantiation launched while a trace buffer is attached (rz_trace_attach) leaves a record per game; the production
// one carries nothing of it (the trace's live values cost the latency chain 132 bytes of scratch).
template <int PER, bool TRACE>
__global__ __launch_bounds__(kWave * kDefWaves) void k_tree_step_def(Dev E, ValueHead vh, float *obs) {
    __shared__ float part[kDefWaves][kWave];
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    __shared__ unsigned long long trace_t0;
    if (TRACE && threadIdx.x == 0) trace_t0 = rz_trace_now();
    value_quarter_def<PER>(vh, blockIdx.x, lane, wave, part);
    if (wave != 0) {
        __syncthreads();
        return;
    }
    expand_backup_body<float, false, false, false, true, words_of_per<PER>()>(E, nullptr, nullptr, blockIdx.x, lane, RawHeads(), 0, vh, part);
    __syncthreads();   // (wave 0's alone: the other waves have ended)
    select_body<false, words_of_per<PER>()>(E, obs, blockIdx.x, lane);
    if (TRACE && lane == 0) rz_trace_write(E.trace, RZ_TRACE_TREE, E.pend[blockIdx.x] - 1, blockIdx.x, trace_t0);
}

// The flush: the priors of the node expanded in step `slot` of game g -- exp(log_softmax) of the leaf's logits over its legal
// moves, mixed with the Dirichlet noise of counter pend_ctr: expand_backup_body<RAW>'s operations on the same numbers (the
// logits are final: k_heads_split's epilogue), into the block that step reserved.  grid = (games, slots).
__global__ __launch_bounds__(kWave) void k_deferred_priors(Dev E, const float *__restrict__ raw, int ld, long long rows_per_slot) {
    const int g = blockIdx.x, slot = blockIdx.y, lane = threadIdx.x;
    if (slot >= E.pend[g]) return;
    const long long rec = (long long)slot * E.n_games + g;
    const int pb = E.pend_pb[rec];
    if (pb < 0) return;
    const int ctr = E.pend_ctr[rec];
    uint64_t st[2][kWords];
    load_board(E.pend_stones, (int)rec, st);
    const float *r = raw + ((size_t)slot * rows_per_slot + g) * ld;
    float x[kWords];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < kWords; ++i) {
        const int j = lane + 64 * i;
        x[i] = j < E.A ? r[j] : -INFINITY;
        mx = fmaxf(mx, x[i]);
    }
    mx = wave_max(mx, lane);
    float sum = 0.0f;
#pragma unroll
    for (int i = 0; i < kWords; ++i) sum += (lane + 64 * i < E.A) ? expf(x[i] - mx) : 0.0f;
    sum = wave_sum(sum, lane);
    const float lse = mx + logf(sum);
    float *P = arena_priors(E, g, E.cur_arena[g]);
    uint64_t occ[kWords];
#pragma unroll
    for (int j = 0; j < kWords; ++j) occ[j] = st[0][j] | st[1][j];
    const Legal L = legal_of(E, occ, lane);
    int ranks[kWords];
    int before = 0;
#pragma unroll
    for (int j = 0; j < kWords; ++j) ranks[j] = lane_action_rank(E, occ, L, j, lane, before);
    float noise[kWords] = {0.f, 0.f, 0.f, 0.f};
    float noise_sum = 1.0f;
    if (E.add_noise) {
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
    }
#pragma unroll
    for (int j = 0; j < kWords; ++j) {
        const int rk = ranks[j];
        if (rk < 0) continue;
        float prior = expf(x[j] - lse);
        if (E.add_noise) prior = 0.75f * prior + 0.25f * (noise[j] / noise_sum);
        P[pb + rk] = prior;
    }
}

__global__ void k_deferred_reset(Dev E) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < E.n_games) E.pend[g] = 0;
}

// EXPAND + BACKUP of simulation s and SELECT + STEP of simulation s+1 in one launch (same
// wave, same game): saves a kernel boundary per simulation.  The barrier makes the tree
// updates of the first half visible to the loads of the second.
template <typename VT>
__global__ __launch_bounds__(kWave) void k_tree_step(Dev E, const float *logp, const VT *value, float *obs) {
    __builtin_amdgcn_s_setprio(3);
    expand_backup_body<VT>(E, logp, value, blockIdx.x, threadIdx.x);
    __syncthreads();
    select_body<false>(E, obs, blockIdx.x, threadIdx.x);
}

// K simulations in flight (opt-in): the game's wave backs up the kb pending slots of the previous step one after
// the other, then selects ks new ones; the barrier between two slots orders the tree updates of one before the loads
// of the next (same wave: s_waitcnt + s_barrier).  kb = 0 / ks = 0 give the select-only / backup-only launches.
template <bool RAW>
__global__ __launch_bounds__(kWave) void k_tree_step_vl(Dev E, const float *logp, const float *value, RawHeads rh,
                                                        float *obs, int kb, int ks) {
    __builtin_amdgcn_s_setprio(3);
#pragma unroll 1
    for (int j = 0; j < kb; ++j) {
        expand_backup_body<float, false, RAW, true>(E, logp, value, blockIdx.x, threadIdx.x, rh, j);
        __syncthreads();
    }
#pragma unroll 1
    for (int j = 0; j < ks; ++j) {
        select_body<true>(E, obs, blockIdx.x, threadIdx.x, j);
        __syncthreads();
    }
}

// ------------------------------------------------------------------ K simulations in flight, level-synchronous
// The production kernel of the opt-in virtual-loss mode (k_tree_step_vl above is its sequential restatement: one
// wave, one slot after the other, kept selectable for the tests that compare the two tree for tree).  A game is a
// workgroup of K waves.  Every step of the sequential rule "slot j sees the virtual losses of slots 0 .. j-1" that does
// not depend on an earlier slot is done for all slots at once:
//   * BACKUP of the kb pending leaves: wave j finishes the heads of leaf j (log_softmax, tanh), draws its noise and
//     writes its prior block at an offset taken from a prefix sum over the slots (bump allocator in slot order); then
//     wave 0 trades every virtual loss for the value: W(node) += x + 1 for all (slot, level) pairs with fire-and-forget
//     f64 atomics issued in slot order (same wave, same address: applied in program order), N untouched.
//   * SELECT of ks new leaves, level by level: at level L the slots sit in at most K distinct nodes; wave j stages the
//     child records of slot j's node in LDS (all nodes of a level in ONE memory round trip instead of one per slot and
//     level), then the slots of a node are walked IN SLOT ORDER by one wave (the groups of different nodes side by side): scores from LDS, first maximum, virtual loss into the staged
//     record at once (the next slot at the same node sees it), the slot's own view of the chosen child -- N, W as they
//     were BEFORE its own virtual loss, i.e. with those of the earlier slots only -- goes along to the next level.
//     This is the sequential rule exactly: what slot j sees at a node are the losses of the slots before it, and
//     those were all placed at the same level earlier in the same pass.  A moved child block is written back from the
//     staged copy, so nothing has to be patched.
//   * the tails (terminal test, duplicate-leaf test, leaf arrays, observation planes) again one wave per slot.
struct MlSlot {
    uint64_t st[2][kWords];  // board of the slot's path so far
    int4 lo, hi;             // the slot's VIEW of its current node: record with the virtual losses of earlier slots only
    int4 xlo;                // owner slot only: the node's structure fields (y FC, z NV, w K | cap) as they evolve in a pass
    int node, rank, owner, depth, fresh, active, to_move, last, nst, moved, pad0, pad1;
};

__host__ __device__ inline int ml_prior_bytes(int K, int A, bool puct) { return puct ? ((K * A * 4 + 15) / 16) * 16 : 0; }
__host__ __device__ inline int ml_lds_bytes(int K, int A, bool puct) {
    return K * (int)sizeof(MlSlot) + K * A * 32 + ml_prior_bytes(K, A, puct) + 64 + K * 16;
}

template <bool RAW>
__global__ void k_tree_step_ml(Dev E, const float *logp, const float *value, RawHeads rh, float *obs, int kb, int ks) {
    extern __shared__ __attribute__((aligned(16))) char ml_lds[];
    const int g = blockIdx.x, K = E.K, A = E.A, S = E.S;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool puct_mode = E.score_mode == RZ_SCORE_PUCT;
    MlSlot *SL = reinterpret_cast<MlSlot *>(ml_lds);
    int4 *stage = reinterpret_cast<int4 *>(ml_lds + K * sizeof(MlSlot));          // [K][A][2]
    float *stage_p = reinterpret_cast<float *>(stage + (size_t)K * A * 2);          // [K][A] (PUCT)
    int *misc = reinterpret_cast<int *>(reinterpret_cast<char *>(stage_p) + ml_prior_bytes(K, A, puct_mode));  // [0] any slot active
    double *bval = reinterpret_cast<double *>(misc + 16);                            // [K] leaf values
    int *bdepth = reinterpret_cast<int *>(bval + K);                                 // [K]
    int *bexp = bdepth + K;                                                          // [K] children of an expanded leaf, else 0
    if (!E.active[g]) return;  // the whole workgroup: before any barrier
    const int arena = E.cur_arena[g];
    int4 *R = arena_records(E, g, arena);
    float *P = arena_priors(E, g, arena);

    // ------------------------------------------------------------- BACKUP of the pending leaves
    if (kb > 0) {
        const int j = w, gk = g * K + j;
        const bool mine = j < kb;
        const int ptop0 = E.ptop[g], nblk0 = E.nblk[g], ctr0 = E.noise_ctr[g], top0 = E.top[g];  // before anyone moves them
        int depth = 0, fresh = 0, term = 0, leaf = 0, k = 0;
        double v = 0.0;
        float lse = 0.0f;
        float x[kWords] = {0.f, 0.f, 0.f, 0.f};
        uint64_t st[2][kWords] = {{0, 0, 0, 0}, {0, 0, 0, 0}}, occ[kWords] = {0, 0, 0, 0};
        Legal L;
        L.k = 0; L.cols = 0ull; L.height = 0;
        bool expand = false;
        if (mine) {
            depth = E.leaf_depth[gk];
            fresh = E.leaf_fresh[gk];
            term = E.leaf_term[gk];
            leaf = E.leaf_node[gk];
            const double tval = E.leaf_tval[gk];
            load_board(E.leaf_stones, gk, st);
            float val = 0.0f;
            if (RAW) {  // the heads of leaf j, as in expand_backup_body
                const float *r = rh.raw + (size_t)gk * rh.ld;
                const float *hp = rh.hid + (size_t)gk * 64 + lane;
                const float w2 = rh.w2[lane], b2 = rh.b2[0];
                float hid = hp[0], mx = -INFINITY;
                if (rh.n_parts == 4) {
                    const long long rs = rh.raw_part_stride, hs = rh.hid_part_stride;
                    const float act_scale = rh.act_scale[0], val_scale = rh.val_scale[0];
#pragma unroll
                    for (int i = 0; i < kWords; ++i) {
                        const int a = lane + 64 * i;
                        const bool in = a < A;
                        const float q0 = in ? r[a] : 0.f, q1 = in ? r[a + rs] : 0.f, q2 = in ? r[a + 2 * rs] : 0.f,
                                    q3 = in ? r[a + 3 * rs] : 0.f, bias = in ? rh.act_bias[a] : 0.f;
                        x[i] = in ? fmaf(((q0 + q1) + q2) + q3, act_scale, bias) : -INFINITY;
                        mx = fmaxf(mx, x[i]);
                    }
                    hid = fmaxf(fmaf(((hid + hp[hs]) + hp[2 * hs]) + hp[3 * hs], val_scale, rh.val_bias[lane]), 0.0f);
                } else {
#pragma unroll
                    for (int i = 0; i < kWords; ++i) {
                        const int a = lane + 64 * i;
                        x[i] = a < A ? r[a] : -INFINITY;
                        mx = fmaxf(mx, x[i]);
                    }
                }
                mx = wave_max(mx, lane);
                float sum = 0.0f;
#pragma unroll
                for (int i = 0; i < kWords; ++i) sum += (lane + 64 * i < A) ? expf(x[i] - mx) : 0.0f;
                sum = wave_sum(sum, lane);
                lse = mx + logf(sum);
                float h = hid * w2;
                h = wave_sum(h, lane);
                val = tanhf(h + b2);
            } else {
                val = value[gk];
            }
            v = term ? tval : (double)val;
            expand = !term && fresh != 2 && fresh != 3;
#pragma unroll
            for (int i = 0; i < kWords; ++i) occ[i] = st[0][i] | st[1][i];
            L = legal_of(E, occ, lane);
            k = L.k;
        }
        if (lane == 0 && j < K) {
            bexp[j] = (mine && expand) ? k : 0;
            bval[j] = v;
            bdepth[j] = mine ? depth : -1;
        }
        __syncthreads();
        // bump allocation in slot order: offsets = prefix sums over the slots before j
        int before_k = 0, before_n = 0, total_k = 0, total_n = 0;
        for (int i = 0; i < kb; ++i) {
            const int ki = bexp[i];
            if (i < j) { before_k += ki; before_n += ki > 0 ? 1 : 0; }
            total_k += ki;
            total_n += ki > 0 ? 1 : 0;
        }
        if (w == 0 && lane == 0) {
            if ((long long)ptop0 + total_k > E.pcap || nblk0 + total_n > E.qcap) {
                atomicOr(&E.err[g], RZ_FLAG_BLOCKS_FULL);
                atomicOr(E.err_any, RZ_FLAG_BLOCKS_FULL);
            } else if (puct_mode && (long long)top0 + total_k > E.cap) {
                atomicOr(&E.err[g], RZ_FLAG_ARENA_FULL);
                atomicOr(E.err_any, RZ_FLAG_ARENA_FULL);
            } else {
                E.ptop[g] = ptop0 + total_k;
                E.nblk[g] = nblk0 + total_n;
                if (E.add_noise) E.noise_ctr[g] = ctr0 + total_n;
                if (puct_mode) E.top[g] = top0 + total_k;
            }
        }
        const bool fits = (long long)ptop0 + total_k <= E.pcap && nblk0 + total_n <= E.qcap &&
                          (!puct_mode || (long long)top0 + total_k <= E.cap);
        if (mine && expand && fits) {
            const int pb = ptop0 + before_k, ctop = top0 + before_k;
            const float uniform = 1.0f / (float)k;
            int ranks[kWords];
            int before = 0;
#pragma unroll
            for (int i = 0; i < kWords; ++i) ranks[i] = lane_action_rank(E, occ, L, i, lane, before);
            float noise[kWords] = {0.f, 0.f, 0.f, 0.f};
            float noise_sum = 1.0f;
            if (E.add_noise) {
                const uint64_t key = mix64(mix64(E.noise_key[g]) ^ (uint64_t)(ctr0 + before_n));
                float local = 0.0f;
#pragma unroll
                for (int i = 0; i < kWords; ++i)
                    if (ranks[i] >= 0) {
                        noise[i] = gamma03(hash32((uint32_t)key ^ (uint32_t)(key >> 32)) + 0x9E3779B9u * (uint32_t)(64 * i + lane + 1));
                        local += noise[i];
                    }
                local = wave_sum(local, lane);
                noise_sum = local > 0.0f ? local : 1.0f;
            }
#pragma unroll
            for (int i = 0; i < kWords; ++i) {
                const int r = ranks[i];
                if (r < 0) continue;
                const int a = 64 * i + lane;
                float prior = uniform;
                if (RAW) prior = expf(x[i] - lse);
                else if (logp) prior = expf(logp[(long long)gk * A + a]);
                if (E.add_noise) prior = 0.75f * prior + 0.25f * (noise[i] / noise_sum);
                P[pb + r] = prior;
                if (puct_mode) {
                    R[2 * (ctop + r)] = make_int4(0, -1, 0, 0);
                    R[2 * (ctop + r) + 1] = make_hi(0.0, -1, prior);
                }
            }
            if (lane == 0) {  // the leaf's record: its blocks (N and W stay: counted at selection / traded below)
                int32_t *f = reinterpret_cast<int32_t *>(R + 2 * leaf);
                f[1] = puct_mode ? ctop : -1;
                f[2] = puct_mode ? k : 0;
                f[3] = pack_kc(k, puct_mode ? k : 0);
                *rec_pb(R, leaf) = pb;
            }
        }
        __syncthreads();
        if (w == 0) {
            // W(node) += x + 1 for every (slot, level) pair, slot after slot: fire-and-forget atomics of ONE wave.  The
            // path entries of all slots are fetched first (one round trip), then the atomics go out in slot order.
            int pnode[RZ_MAX_IN_FLIGHT];
#pragma unroll
            for (int i = 0; i < RZ_MAX_IN_FLIGHT; ++i) {
                pnode[i] = 0;
                if (i < kb) {
                    const int di = bdepth[i];
                    if (lane <= di) pnode[i] = E.path[(long long)(g * K + i) * E.path_stride + lane];
                }
            }
#pragma unroll
            for (int i = 0; i < RZ_MAX_IN_FLIGHT; ++i) {
                if (i < kb) {
                    const int di = bdepth[i];
                    const double vi = bval[i];
                    if (lane <= di) unsafeAtomicAdd(rec_wsum(R, pnode[i]), (((di - lane) & 1) ? vi : -vi) + 1.0);
                    if (di >= kWave) {  // (paths longer than a wave: the rest, level by level)
                        const int32_t *path = E.path + (long long)(g * K + i) * E.path_stride;
                        for (int d = lane + kWave; d <= di; d += kWave)
                            unsafeAtomicAdd(rec_wsum(R, path[d]), (((di - d) & 1) ? vi : -vi) + 1.0);
                    }
                }
            }
        }
        // the selection below must read what the stores and atomics above wrote
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    if (ks <= 0) return;

    // ------------------------------------------------------------- SELECT, level by level
    int top = E.top[g];
    const int top_at_start = top;
    {
        const int4 lo0 = R[0], hi0 = R[1];
        uint64_t st0[2][kWords];
        load_board(E.root_stones, g, st0);
        const int to_move0 = E.root_to_move[g], last0 = E.root_last[g];
        const int j = w;
        if (j < ks) {
            double wv = rec_w(hi0);
            for (int i = 0; i < j; ++i) wv = wv - 1.0;  // the root as slot j sees it: the slots before it have passed
            if (lane < 2 * kWords) SL[j].st[lane / kWords][lane % kWords] = word_of(st0[lane / kWords], lane % kWords);
            if (lane == 0) {
                SL[j].lo = make_int4(lo0.x + j, lo0.y, lo0.z, lo0.w);
                SL[j].hi = make_hi(wv, hi0.z, __int_as_float(hi0.w));
                SL[j].xlo = lo0;
                SL[j].node = 0;
                SL[j].rank = -1;
                SL[j].owner = 0;
                SL[j].depth = 0;
                SL[j].fresh = 0;
                SL[j].active = 1;
                SL[j].to_move = to_move0;
                SL[j].last = last0;
                SL[j].nst = count_bits(st0[0]) + count_bits(st0[1]);
                SL[j].moved = 0;
                SL[j].pad0 = -1;
                E.path[(long long)(g * K + j) * E.path_stride] = 0;
            }
        }
        if (w == 0 && lane == 0) {
            misc[0] = 1;
            double wv = rec_w(hi0);
            for (int i = 0; i < ks; ++i) wv = wv - 1.0;
            *rec_n(R, 0) = lo0.x + ks;  // the root carries the virtual loss of every slot
            *rec_wsum(R, 0) = wv;
        }
    }
    for (int pass = 0; pass <= S + 1; ++pass) {
        __syncthreads();
        if (misc[0] == 0) break;
        // phase A: wave j stages the child records of its slot's node (once per distinct node: the owner does it)
        if (w < ks) {
            const int j = w;
            if (SL[j].active && SL[j].owner == j) {
                const int4 xlo = SL[j].xlo;
                const int k = rec_k(xlo), fc = xlo.y, n_load = puct_mode ? k : xlo.z, pb = SL[j].hi.z;
                int4 *dst = stage + (size_t)j * A * 2;
                for (int i = lane; i < n_load; i += kWave) {
                    dst[2 * i] = R[2 * (fc + i)];
                    dst[2 * i + 1] = R[2 * (fc + i) + 1];
                    if (puct_mode) stage_p[(size_t)j * A + i] = P[pb + i];
                }
            }
        }
        __syncthreads();
        // phase B: the owner's wave walks the slots of ITS node in slot order (slots at different nodes do not see each
        // other within a level: the groups run side by side; at the root all slots are one group)
        if (w < ks && SL[w].active && SL[w].owner == w)
        for (int j = w; j < ks; ++j) {
            if (!SL[j].active || SL[j].owner != w) continue;
            const int o = w;
            const int4 xlo = SL[o].xlo;
            const int4 vlo = SL[j].lo;
            const int k = rec_k(xlo);
            if (k == 0) {  // the path ends in an existing leaf (its virtual loss came with the choice of it)
                if (lane == 0) SL[j].active = 0;
                continue;
            }
            int fc = xlo.y, cap = rec_cap(xlo);
            const int nv = xlo.z;
            int4 *stg = stage + (size_t)o * A * 2;
            int r;
            int4 clo, chi;
            bool fresh = false;
            if (!puct_mode && nv < k) {
                if (nv == cap) {  // the child vector grows: its new block is allocated in phase C (in owner order, whatever the
                                  // groups' timing) and written from the staged copy
                    cap = cap == 0 ? (k < kFirstCap ? k : kFirstCap) : (2 * cap < k ? 2 * cap : k);
                    if (lane == 0) SL[o].moved = 1;
                }
                r = nv;
                fresh = true;
                clo = make_int4(1, -1, 0, 0);          // the pending child: visited once, lost
                chi = make_hi(-1.0, -1, 0.0f);
                if (lane == 0) {
                    stg[2 * r] = clo;
                    stg[2 * r + 1] = chi;
                    SL[o].xlo = make_int4(xlo.x, fc, nv + 1, pack_kc(k, cap));
                }
            } else {
                const int pn = vlo.x;
                if (!puct_mode && (pn < 1 || pn >= E.logtab_n)) {
                    flag(E, g, RZ_FLAG_LOGTAB, lane);
                    if (lane == 0) { SL[j].fresh = 2; SL[j].active = 0; }
                    continue;
                }
                const double parent_term = puct_mode ? sqrt((double)pn) : E.logtab[pn];
                double best = -INFINITY;
                int besti = 0x7fffffff;
#pragma unroll
                for (int i = 0; i < kWords; ++i) {
                    const int r0 = lane + 64 * i;
                    if (r0 < k) {
                        const int4 cl = stg[2 * r0], ch = stg[2 * r0 + 1];
                        const double sc = puct_mode ? puct(rec_w(ch), cl.x, stage_p[(size_t)o * A + r0], parent_term, E.c_puct)
                                                    : uct_ref(rec_w(ch), cl.x, parent_term, E.c_puct);
                        if (sc > best) {
                            best = sc;
                            besti = r0;
                        }
                    }
                }
                r = __builtin_amdgcn_readfirstlane(wave_first_max(best, besti));
                if (r >= k) {
                    flag(E, g, RZ_FLAG_INTERNAL, lane);
                    if (lane == 0) { SL[j].fresh = 2; SL[j].active = 0; }
                    continue;
                }
                clo = stg[2 * r];   // the child as THIS slot sees it: before its own virtual loss
                chi = stg[2 * r + 1];
                if (lane == 0) {
                    stg[2 * r] = make_int4(clo.x + 1, clo.y, clo.z, clo.w);
                    stg[2 * r + 1] = make_hi(rec_w(chi) - 1.0, chi.z, __int_as_float(chi.w));
                }
            }
            // (the move on the slot's board -- which cell child r is -- depends on no other slot: phase B2 below, every
            // slot's own wave, instead of inside this slot-after-slot loop)
            if (lane == 0) {
                SL[j].depth += 1;
                SL[j].rank = r;
                SL[j].pad0 = r;   // for phase B2
                SL[j].lo = clo;
                SL[j].hi = chi;
                if (fresh) {
                    SL[j].fresh = 1;
                    SL[j].active = 0;
                }
            }
        }
        __syncthreads();
        // phase B2: wave j plays the chosen child on slot j's board (all slots side by side; wave 0 goes on to phase C,
        // which reads none of what is written here)
        if (w < ks && SL[w].pad0 >= 0) {
            const int j = w, r = SL[j].pad0;
            uint64_t st[2][kWords], occ[kWords];
#pragma unroll
            for (int i = 0; i < kWords; ++i) {
                st[0][i] = SL[j].st[0][i];
                st[1][i] = SL[j].st[1][i];
                occ[i] = st[0][i] | st[1][i];
            }
            const Legal L = legal_of(E, occ, lane);
            int action, cell;
            const bool ok = nth_legal(E, occ, L, r, lane, action, cell);
            if (!ok) flag(E, g, RZ_FLAG_INTERNAL, lane);
            if (lane == 0) {
                SL[j].pad0 = -1;
                if (ok) {
                    const int tm = SL[j].to_move;
                    SL[j].st[tm][cell >> 6] |= 1ull << (cell & 63);
                    SL[j].to_move = tm ^ 1;
                    SL[j].last = cell;
                    SL[j].nst += 1;
                }
            }
        }
        if (w != 0) continue;
        // phase C (wave 0): new blocks for the child vectors that grew, write back what the pass changed, resolve the slots'
        // new nodes, owners of the next pass
        for (int o = 0; o < ks; ++o) {
            if (SL[o].owner != o || !SL[o].moved) continue;
            const int4 xlo = SL[o].xlo;
            const int need = rec_cap(xlo);
            if ((long long)top + need > E.cap) {
                flag(E, g, RZ_FLAG_ARENA_FULL, lane);
                continue;
            }
            if (lane == 0) SL[o].xlo = make_int4(xlo.x, top, xlo.z, xlo.w);
            top += need;
        }
        for (int o = 0; o < ks; ++o) {
            if (SL[o].owner != o || SL[o].rank < -1) continue;
            const int4 xlo = SL[o].xlo;
            const int X = SL[o].node;
            bool touched = false;
            for (int j = o; j < ks; ++j) touched = touched || (SL[j].owner == o && SL[j].rank >= 0);
            if (!touched) continue;
            if (lane == 0) {  // the structure fields of the node (its N / W were written when it was chosen)
                int32_t *f = reinterpret_cast<int32_t *>(R + 2 * X);
                f[1] = xlo.y;
                f[2] = xlo.z;
                f[3] = xlo.w;
            }
            const int4 *stg = stage + (size_t)o * A * 2;
            if (SL[o].moved) {
                for (int i = lane; i < xlo.z; i += kWave) {
                    R[2 * (xlo.y + i)] = stg[2 * i];
                    R[2 * (xlo.y + i) + 1] = stg[2 * i + 1];
                }
            }
        }
        int any = 0;
        if (lane < ks) {
            const int j = lane;
            const int rk = SL[j].rank;
            if (rk >= 0) {
                const int o = SL[j].owner;
                const int4 xlo = SL[o].xlo;
                const int node = xlo.y + rk;
                if (!SL[o].moved) {  // (a moved block was written whole, above)
                    const int4 *stg = stage + (size_t)o * A * 2;
                    R[2 * node] = stg[2 * rk];
                    R[2 * node + 1] = stg[2 * rk + 1];
                }
                E.path[(long long)(g * K + j) * E.path_stride + SL[j].depth] = node;
                SL[j].node = node;
            }
        }
        // (every lane < ks has read its owner's record before any of them is overwritten)
        __builtin_amdgcn_wave_barrier();
        int node_l = 0, act_l = 0;
        if (lane < ks) {
            const int j = lane;
            if (SL[j].rank >= 0) {
                SL[j].xlo = SL[j].lo;
                SL[j].rank = -1;
            }
            SL[j].moved = 0;
            node_l = SL[j].node;
            act_l = SL[j].active;
        }
        int owner_l = lane;
        for (int i = ks - 1; i >= 0; --i) {
            const int ni = __shfl(node_l, i), ai = __shfl(act_l, i);
            if (ai && ni == node_l && i <= lane) owner_l = i;
        }
        if (lane < ks) SL[lane].owner = owner_l;
        any = __ballot(lane < ks && act_l) != 0ull ? 1 : 0;
        if (lane == 0) misc[0] = any;
    }
    __syncthreads();
    if (w == 0 && lane == 0 && top != top_at_start) E.top[g] = top;

    // ------------------------------------------------------------- tails: one wave per slot
    if (w < ks) {
        const int j = w, gk = g * K + j;
        uint64_t st[2][kWords];
#pragma unroll
        for (int i = 0; i < kWords; ++i) {
            st[0][i] = SL[j].st[0][i];
            st[1][i] = SL[j].st[1][i];
        }
        const int to_move = SL[j].to_move, last = SL[j].last, nst = SL[j].nst, depth = SL[j].depth, node = SL[j].node;
        int fresh = SL[j].fresh;
        int term = 0;
        double tval = 0.0;
        int winner = -1;
        if (depth == 0) {
            if (line_anywhere(st[0], S, E.BH, E.BW, E.n_row, lane, E.bw_rcp)) winner = 0;
            else if (line_anywhere(st[1], S, E.BH, E.BW, E.n_row, lane, E.bw_rcp)) winner = 1;
        } else {
            const int mover = to_move ^ 1;
            if (line_through(mover == 0 ? st[0] : st[1], last, E.BH, E.BW, E.n_row, lane, E.bw_rcp, E.n_rcp)) winner = mover;
        }
        if (winner >= 0) {
            term = 2;
            tval = (winner == to_move) ? 1.0 : -1.0;
        } else if (nst == S) {
            term = 1;
        }
        if (fresh == 0 && term == 0)  // an unexpanded leaf an earlier slot of this step ends in too: that one expands it
            for (int i = 0; i < j; ++i)
                if (SL[i].node == node) fresh = 3;
        if (lane == 0) {
            E.leaf_node[gk] = node;
            E.leaf_depth[gk] = depth;
            E.leaf_fresh[gk] = fresh;
            E.leaf_term[gk] = term;
            E.leaf_tval[gk] = tval;
            E.leaf_to_move[gk] = to_move;
            E.leaf_last[gk] = last;
        }
        store_board(E.leaf_stones, gk, st, lane);
        if (obs != nullptr)
            write_obs(obs + (long long)gk * 4 * S, to_move == 0 ? st[0] : st[1], to_move == 0 ? st[1] : st[0], last, nst, S,
                      lane);
    }
}

// ------------------------------------------------------------------ synthetic evaluators
__global__ __launch_bounds__(kWave) void k_eval_synth(Dev E, int kind, float *logp, float *value) {
    const int g = blockIdx.x;  // leaf index: game * K + slot (K = 1: the game)
    const int lane = threadIdx.x;
    if (!E.active[g / E.K]) return;
    const int S = E.S;
    uint64_t st[2][kWords];
    load_board(E.leaf_stones, g, st);
    int acc = 0;
#pragma unroll
    for (int j = 0; j < kWords; ++j) {
        const int c = 64 * j + lane;
        if (c < S) {
            const int a = (int)((st[0][j] >> lane) & 1ull), b = (int)((st[1][j] >> lane) & 1ull);
            acc += (c + 1) * (a + 3 * b);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) {
        float v = 0.0f;
        if (kind == RZ_EVAL_VLIN) {
            const int s = acc + 5 * E.leaf_to_move[g];
            v = (float)((s % 17) - 8) / 8.0f;
        }
        value[g] = v;
    }
    if (logp != nullptr) {
        uint64_t occ[kWords];
#pragma unroll
        for (int j = 0; j < kWords; ++j) occ[j] = st[0][j] | st[1][j];
        const Legal L = legal_of(E, occ, lane);
        const float lp = L.k > 0 ? logf(1.0f / (float)L.k) : 0.0f;
        int before = 0;
#pragma unroll
        for (int j = 0; j < kWords; ++j) {
            const int r = lane_action_rank(E, occ, L, j, lane, before);
            const int a = 64 * j + lane;
            if (a < E.A) logp[(long long)g * E.A + a] = r >= 0 ? lp : -INFINITY;
        }
    }
}

// ------------------------------------------------------------------ random rollouts
// RolloutMCTS._evaluate (rlzero/mcts/rollout_mcts.py:49-74): from the leaf play uniformly random
// legal moves (the reference takes the arg-max of k fresh uniforms, :99-100 = a uniform choice)
// until the game ends or n_limit plies, then value = 0 on a tie / limit, else +1 if the winner
// is the player to move AFTER the rollout else -1 (the reference's perspective quirk, :68-72:
// the winner has just moved, so a decisive rollout always yields -1).  The move index of ply p
// is floor(u * k) with u = the high 32 bits of splitmix64(seed, game, sim, ply) -- reproducible
// on the host (rlzero_amd.mcts.rollout_mcts.rollout_pick) so parity tests can drive the checker
// with the very same choices.
__global__ __launch_bounds__(kWave) void k_eval_rollout(Dev E, uint64_t seed, uint32_t sim, int n_limit,
                                                        float *value) {
    const int g = blockIdx.x;  // leaf index: game * K + slot (K = 1: the game)
    const int lane = threadIdx.x;
    if (!E.active[g / E.K]) return;
    const int S = E.S;
    uint64_t st[2][kWords];
    load_board(E.leaf_stones, g, st);
    int to_move = E.leaf_to_move[g];
    int nst = count_bits(st[0]) + count_bits(st[1]);
    int term = E.leaf_term[g];  // 0 running, 1 tie, 2 won by the player who just moved
    int winner = term == 2 ? (to_move ^ 1) : -1;
    const uint64_t key = mix64(mix64(seed ^ (uint64_t)g) ^ (uint64_t)sim);
    for (int ply = 0; ply < n_limit && term == 0; ++ply) {
        uint64_t occ[kWords];
#pragma unroll
        for (int j = 0; j < kWords; ++j) occ[j] = st[0][j] | st[1][j];
        const Legal L = legal_of(E, occ, lane);
        const uint32_t u = (uint32_t)(mix64(key ^ (uint64_t)ply) >> 32);
        const int r = (int)(((uint64_t)u * (uint64_t)L.k) >> 32);
        int action, cell;
        if (!nth_legal(E, occ, L, r, lane, action, cell)) {
            flag(E, g, RZ_FLAG_INTERNAL, lane);
            break;
        }
        if (to_move == 0) set_bit(st[0], cell); else set_bit(st[1], cell);
        nst += 1;
        if (line_through(to_move == 0 ? st[0] : st[1], cell, E.BH, E.BW, E.n_row, lane, E.bw_rcp, E.n_rcp)) {
            term = 2;
            winner = to_move;
        } else if (nst == S) {
            term = 1;
        }
        to_move ^= 1;
    }
    if (lane == 0) value[g] = (winner < 0) ? 0.0f : (winner == to_move ? 1.0f : -1.0f);
}

// ------------------------------------------------------------------ root read-out
// what: 0 = visits (int32), 1 = W (double), 2 = prior (float); output [n_games][A] by action
__global__ __launch_bounds__(kWave) void k_root_children(Dev E, int what, void *out) {
    const int g = blockIdx.x;
    const int lane = threadIdx.x;
    const int arena = E.cur_arena[g];
    const int4 *R = arena_records(E, g, arena);
    const float *P = arena_priors(E, g, arena);
    uint64_t st[2][kWords], occ[kWords];
    load_board(E.root_stones, g, st);
#pragma unroll
    for (int j = 0; j < kWords; ++j) occ[j] = st[0][j] | st[1][j];
    const Legal L = legal_of(E, occ, lane);
    const int4 lo = R[0], hi = R[1];
    const bool expanded = rec_k(lo) > 0;
    const int fc = lo.y;
    const int nv = expanded ? lo.z : 0;
    int before = 0;
#pragma unroll
    for (int j = 0; j < kWords; ++j) {
        const int r = lane_action_rank(E, occ, L, j, lane, before);
        const int a = 64 * j + lane;
        if (a >= E.A) continue;
        const bool seen = r >= 0 && r < nv;
        const long long o = (long long)g * E.A + a;
        if (what == 0) ((int32_t *)out)[o] = seen ? R[2 * (fc + r)].x : 0;
        else if (what == 1) ((double *)out)[o] = seen ? rec_w(R[2 * (fc + r) + 1]) : 0.0;
        else ((float *)out)[o] = (r >= 0 && expanded) ? P[hi.z + r] : 0.0f;
    }
}

__global__ void k_root_stats(Dev E, int32_t *n, double *w) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.n_games) return;
    const int4 *R = arena_records(E, g, E.cur_arena[g]);
    n[g] = R[0].x;
    w[g] = rec_w(R[1]);
}

// ------------------------------------------------------------------ tree reuse
__device__ __forceinline__ void fresh_root(const Dev &E, int g, int arena, int lane) {
    if (lane == 0) {
        int4 *R = arena_records(E, g, arena);
        R[0] = make_int4(0, -1, 0, 0);
        R[1] = make_hi(0.0, -1, 1.0f);  // TreeNode(None, 1.0), alphazero_mcts.py:35,103
        E.cur_arena[g] = arena;
        E.top[g] = 1;
        E.ptop[g] = 0;
        E.nblk[g] = 0;
    }
}

// AlphaZeroMCTS.update_with_move (alphazero_mcts.py:96-103).  The kept subtree is copied
// breadth-first into the game's other arena (per expanded node: its prior block, and the visited
// prefix of its child records into a block of the same capacity), which also recycles the slots
// of the discarded siblings and of every outgrown child block.  A copied record keeps its SOURCE
// first-child / prior-block offsets until the node itself is taken from the queue, so the queue
// holds destination slots only.
__device__ __forceinline__ void advance_body(const Dev &E, int g, int lane, int mv) {
    if (mv == -2) return;
    const int src_arena = E.cur_arena[g], dst_arena = src_arena ^ 1;
    const int4 *Rs = arena_records(E, g, src_arena);
    int4 *Rd = arena_records(E, g, dst_arena);
    const float *Ps = arena_priors(E, g, src_arena);
    float *Pd = arena_priors(E, g, dst_arena);
    if (mv < 0) {
        if (mv != -1) flag(E, g, RZ_FLAG_ILLEGAL_MOVE, lane);
        fresh_root(E, g, dst_arena, lane);
        return;
    }
    uint64_t st[2][kWords], occ[kWords];
    load_board(E.root_stones, g, st);
#pragma unroll
    for (int j = 0; j < kWords; ++j) occ[j] = st[0][j] | st[1][j];
    const Legal L = legal_of(E, occ, lane);
    int rank = 0, cell = 0;
    if (!locate_action(E, occ, L, mv, rank, cell)) {
        flag(E, g, RZ_FLAG_ILLEGAL_MOVE, lane);
        fresh_root(E, g, dst_arena, lane);
        return;
    }
    const int4 rlo = Rs[0], rhi = Rs[1];
    if (rec_k(rlo) == 0 || rank >= rlo.z) {
        // the chosen child was never visited: it is a TreeNode with N = 0 and no children
        fresh_root(E, g, dst_arena, lane);
        return;
    }
    const int src = rlo.y + rank;
    const int4 slo = Rs[2 * src], shi = Rs[2 * src + 1];
    const float prior = Ps[rhi.z + rank];
    int32_t *queue = E.queue + (long long)g * E.qcap;
    if (lane == 0) {
        Rd[0] = slo;
        Rd[1] = make_int4(shi.x, shi.y, shi.z, __float_as_int(prior));
        queue[0] = 0;
    }
    int q_tail = rec_k(slo) > 0 ? 1 : 0;
    __syncthreads();
    int dtop = 1, dptop = 0, nblk = 0;
    bool full = false;
    for (int q_head = 0; q_head < q_tail; ++q_head) {
        const int dst = queue[q_head];
        const int4 lo = Rd[2 * dst], hi = Rd[2 * dst + 1];
        const int k = rec_k(lo), cap = rec_cap(lo), nv = lo.z, sfc = lo.y, spb = hi.z;
        // the carried subtree must leave room for the n_playout expansions of the coming search
        if ((long long)dptop + k > E.pcap - (long long)(E.n_playout + 1) * E.A ||
            (long long)dtop + cap > E.cap - (long long)(E.n_playout + 1) * 8 - 2 * E.A || nblk >= E.qcap - E.n_playout - 1) {
            full = true;
            break;
        }
        for (int r = lane; r < k; r += kWave) Pd[dptop + r] = Ps[spb + r];
        for (int r0 = 0; r0 < nv; r0 += kWave) {
            const int r = r0 + lane;
            bool expanded = false;
            if (r < nv) {
                const int4 a = Rs[2 * (sfc + r)], b = Rs[2 * (sfc + r) + 1];
                Rd[2 * (dtop + r)] = a;
                Rd[2 * (dtop + r) + 1] = b;
                expanded = rec_k(a) > 0;
            }
            const unsigned long long has = __ballot(expanded);
            if (expanded) {
                const int pos = q_tail + __popcll(has & ((1ull << lane) - 1ull));
                if (pos < E.qcap) queue[pos] = dtop + r;
            }
            q_tail += __popcll(has);
        }
        if (q_tail > E.qcap) {
            full = true;
            break;
        }
        if (lane == 0) {
            Rd[2 * dst] = make_int4(lo.x, cap > 0 ? dtop : -1, nv, lo.w);
            *rec_pb(Rd, dst) = dptop;
        }
        dtop += cap;
        dptop += k;
        nblk += 1;
        __syncthreads();  // records and queue entries written by other lanes are read next iteration
    }
    if (full) {
        // The reference's tree is unbounded; here the kept subtree is limited to pool_factor * n_playout expanded
        // nodes.  A larger one is DROPPED (the search restarts from a fresh root, like update_with_move(-1)):
        // a deviation from the reference that is counted (rz_stats.reuse_dropped) and flagged per game with the
        // non-fatal RZ_FLAG_REUSE_DROPPED, never silent and never fatal for the rest of the batch.
        if (lane == 0) {
            atomicOr(&E.err[g], RZ_FLAG_REUSE_DROPPED);
            atomicAdd(E.reuse_drops, 1);
        }
        fresh_root(E, g, dst_arena, lane);
        return;
    }
    if (lane == 0) {
        E.cur_arena[g] = dst_arena;
        E.top[g] = dtop;
        E.ptop[g] = dptop;
        E.nblk[g] = nblk;
    }
}

__global__ __launch_bounds__(kWave) void k_advance(Dev E, const int32_t *moves) {
    advance_body(E, blockIdx.x, threadIdx.x, moves[blockIdx.x]);
}

// ------------------------------------------------------------------ game step
// -> the winner (player id or -1) and whether the game is over, wave-uniform
__device__ __forceinline__ void step_body(const Dev &E, int g, int lane, int mv, int &who, bool &over) {
    const int S = E.S;
    uint64_t st[2][kWords];
    load_board(E.root_stones, g, st);
    int to_move = E.root_to_move[g];
    if (mv >= 0) {
        uint64_t occ[kWords];
#pragma unroll
        for (int j = 0; j < kWords; ++j) occ[j] = st[0][j] | st[1][j];
        const Legal L = legal_of(E, occ, lane);
        int rank = 0, cell = 0;
        if (!locate_action(E, occ, L, mv, rank, cell)) {
            flag(E, g, RZ_FLAG_ILLEGAL_MOVE, lane);
        } else {
            if (to_move == 0) set_bit(st[0], cell); else set_bit(st[1], cell);
            to_move ^= 1;
            if (lane == 0) {
                E.root_to_move[g] = to_move;
                E.root_last[g] = cell;
            }
            store_board(E.root_stones, g, st, lane);
        }
    }
    who = -1;
    if (line_anywhere(st[0], S, E.BH, E.BW, E.n_row, lane, E.bw_rcp)) who = 0;
    else if (line_anywhere(st[1], S, E.BH, E.BW, E.n_row, lane, E.bw_rcp)) who = 1;
    over = who >= 0 || count_bits(st[0]) + count_bits(st[1]) == S;
}

__global__ __launch_bounds__(kWave) void k_step_games(Dev E, const int32_t *moves, int32_t *winner,
                                                      uint8_t *ended) {
    const int g = blockIdx.x;
    const int lane = threadIdx.x;
    int who = -1;
    bool over = false;
    step_body(E, g, lane, moves[g], who, over);
    if (lane == 0) {
        if (winner) winner[g] = who;
        if (ended) ended[g] = over ? 1 : 0;
    }
}

__global__ void k_set_roots(Dev E, const uint64_t *stones, const int32_t *to_move,
                            const int32_t *last_move, const uint8_t *mask, int reset_trees) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.n_games) return;
    if (mask != nullptr && !mask[g]) return;
    for (int j = 0; j < 2 * kWords; ++j) {
        const uint64_t v = stones[(long long)g * 2 * kWords + j];
        E.root_stones[(long long)g * 2 * kWords + j] = v & E.valid[j % kWords];
    }
    E.root_to_move[g] = to_move[g] & 1;
    E.root_last[g] = last_move[g];
    if (reset_trees) fresh_root(E, g, E.cur_arena[g], 0);
}

__global__ void k_get_roots(Dev E, uint64_t *stones, int32_t *to_move, int32_t *last_move) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.n_games) return;
    for (int j = 0; j < 2 * kWords; ++j)
        stones[(long long)g * 2 * kWords + j] = E.root_stones[(long long)g * 2 * kWords + j];
    to_move[g] = E.root_to_move[g];
    last_move[g] = E.root_last[g];
}

__global__ void k_get_leaves(Dev E, uint64_t *stones, int32_t *to_move, int32_t *last_move,
                             int32_t *terminal) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.n_games) return;
    for (int j = 0; j < 2 * kWords; ++j)
        stones[(long long)g * 2 * kWords + j] = E.leaf_stones[(long long)g * 2 * kWords + j];
    to_move[g] = E.leaf_to_move[g];
    last_move[g] = E.leaf_last[g];
    terminal[g] = E.leaf_term[g];
}

// keys == nullptr: the default keys (noise_seed ^ game << 20) for the selected games
__global__ void k_set_noise_keys(Dev E, const uint64_t *keys, const uint8_t *mask) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.n_games || (mask != nullptr && !mask[g])) return;
    E.noise_key[g] = keys != nullptr ? keys[g] : (E.noise_seed ^ ((uint64_t)g << 20));
    E.noise_ctr[g] = 0;
}

__global__ void k_set_active(Dev E, const uint8_t *active) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < E.n_games) E.active[g] = active ? (active[g] ? 1 : 0) : 1;
}

// which: 0 = leaf boards, 1 = root boards
__global__ __launch_bounds__(kWave) void k_encode(Dev E, int which, float *obs) {
    const int g = blockIdx.x;  // which == 0: leaf index (game * K + slot); which == 1: game
    const int lane = threadIdx.x;
    uint64_t st[2][kWords];
    load_board(which == 0 ? E.leaf_stones : E.root_stones, g, st);
    const int to_move = which == 0 ? E.leaf_to_move[g] : E.root_to_move[g];
    const int last = which == 0 ? E.leaf_last[g] : E.root_last[g];
    const int nst = count_bits(st[0]) + count_bits(st[1]);
    write_obs(obs + (long long)g * 4 * E.S, to_move == 0 ? st[0] : st[1],
              to_move == 0 ? st[1] : st[0], last, nst, E.S, lane);
}

__global__ void k_uct_scores(const double *w, const int32_t *n, const int32_t *np, double c,
                             const double *logtab, long long logtab_n, double *out, long long count) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int pn = np[i];
    if (pn == 0 || n[i] == 0 || pn >= logtab_n) {
        out[i] = INFINITY;
        return;
    }
    out[i] = uct_ref(w[i], n[i], logtab[pn], c);
}

// ------------------------------------------------------------------ the move step on the device (include/rlzero_hip.h: rz_play_*)
struct Play {
    int64_t *game_id;          // [G] -1: idle
    int32_t *ply, *state;      // state: 0 idle, 1 running, 2 stalled
    int32_t *mailbox;          // [G] the host's move for a stalled slot, -1: none
    int32_t *keep, *stepm;     // [G] the moves of this step (update_with_move / env.step), written by k_play_draw
    int32_t *top_hwm;          // [G] the fullest a game's arena has been at the end of a search (read before update_with_move compacts it)
    int32_t *step_ab;          // [2]: k_play_draw reads [0] and writes [1], k_play_apply reads [1] and writes [0] = [1] + 1
    const int64_t *queue_ids;
    int32_t *queue_ctl;        // [0] head, [1] entries valid
    int32_t *log;
    int ring, words;
    uint64_t seed;
    double inv_t, margin;
};
enum { kPlayIdle = 0, kPlayRunning = 1, kPlayStalled = 2 };

// rlzero_amd/selfplay.py: move_uniform(seed, game id, ply) -- 53 high bits of a splitmix64 chain: the same bits
__device__ __forceinline__ double play_uniform(uint64_t seed, uint64_t game, uint64_t ply) {
    uint64_t x = mix64(seed);
    x = mix64(x ^ game);
    x = mix64(x ^ ply);
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}

// One wave per slot: the root's visit counts into the log, then the draw of alphazero_mcts.py:88-92,147-148 in fp64 -- taken only
// when the uniform lies farther than `margin` from both edges of its interval (the host's numpy evaluation is the arbiter).
__global__ __launch_bounds__(kWave) void k_play_draw(Dev E, Play Y) {
    __shared__ double sh_e[kWave * kWords];
    const int g = blockIdx.x;
    const int lane = threadIdx.x;
    const int step = Y.step_ab[0];
    if (g == 0 && lane == 0) Y.step_ab[1] = step;
    int32_t *rec = Y.log + ((long long)(step % Y.ring) * E.n_games + g) * Y.words;
    const int state = Y.state[g];
    if (state == kPlayIdle) {
        if (lane == 0) {
            rec[4] = 0;
            Y.keep[g] = -2;
            Y.stepm[g] = -1;
        }
        return;
    }
    // the root's children by action (k_root_children)
    const int arena = E.cur_arena[g];
    const int4 *R = arena_records(E, g, arena);
    uint64_t st[2][kWords], occ[kWords];
    load_board(E.root_stones, g, st);
#pragma unroll
    for (int j = 0; j < kWords; ++j) occ[j] = st[0][j] | st[1][j];
    const Legal L = legal_of(E, occ, lane);
    const int4 lo = R[0];
    const bool expanded = rec_k(lo) > 0;
    const int fc = lo.y;
    const int nv = expanded ? lo.z : 0;
    int before = 0;
    int cnt[kWords];
    bool legal[kWords];
    double x[kWords];
    double mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < kWords; ++j) {
        const int r = lane_action_rank(E, occ, L, j, lane, before);
        const int a = 64 * j + lane;
        legal[j] = a < E.A && r >= 0;
        cnt[j] = (legal[j] && r < nv) ? R[2 * (fc + r)].x : 0;
        if (a < E.A) rec[RZ_PLAY_RECORD_WORDS + a] = legal[j] ? cnt[j] : -1;
        x[j] = legal[j] ? Y.inv_t * log((double)cnt[j] + 1e-10) : -INFINITY;   // alphazero_mcts.py:91
        mx = fmax(mx, x[j]);
    }
    const int64_t gid = Y.game_id[g];
    const int ply = Y.ply[g];
    if (lane == 0) {
        rec[0] = (int32_t)(uint32_t)(uint64_t)gid;
        rec[1] = (int32_t)((uint64_t)gid >> 32);
        rec[2] = ply;
        rec[5] = lo.x;
        rec[7] = 0;
    }
    if (state == kPlayStalled) {
        const int mv = Y.mailbox[g];
        if (lane == 0) {
            if (mv >= 0) {   // the host has decided (rz_play_resolve)
                rec[3] = mv;
                rec[4] = RZ_PLAY_RUNNING | RZ_PLAY_RESOLVED;
                rec[6] = 0;
                Y.mailbox[g] = -1;
                Y.state[g] = kPlayRunning;
                Y.ply[g] = ply + 1;
                E.active[g] = 1;
                Y.keep[g] = mv;
                Y.stepm[g] = mv;
            } else {
                rec[3] = -1;
                rec[4] = RZ_PLAY_RUNNING | RZ_PLAY_STALLED;
                rec[6] = 0;
                Y.keep[g] = -2;
                Y.stepm[g] = -1;
            }
        }
        return;
    }
    // wave maximum of x (doubles: two 32-bit halves through the shuffle)
    for (int off = 32; off >= 1; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off));
#pragma unroll
    for (int j = 0; j < kWords; ++j) sh_e[64 * j + lane] = legal[j] ? exp(x[j] - mx) : 0.0;   // :12
    __syncthreads();
    if (lane == 0) {
        // cumsum in action order (numpy's cumsum is sequential too), then the first interval whose upper edge exceeds u x total
        double total = 0.0;
        for (int a = 0; a < E.A; ++a) total += sh_e[a];
        const double u = play_uniform(Y.seed, (uint64_t)gid, (uint64_t)ply);
        const double target = u * total;
        double c = 0.0, below = 0.0;
        int chosen = -1;
        for (int a = 0; a < E.A; ++a) {
            const double e = sh_e[a];
            if (e > 0.0 && c + e > target) {
                chosen = a;
                below = c;
                c += e;
                break;
            }
            c += e;
        }
        double rel = 0.0;
        if (chosen >= 0) rel = fmin(target - below, c - target) / total;
        const bool ok = chosen >= 0 && total > 0.0 && rel > Y.margin;
        rec[6] = __float_as_int((float)rel);
        if (ok) {
            rec[3] = chosen;
            rec[4] = RZ_PLAY_RUNNING | RZ_PLAY_SEARCHED;
            Y.ply[g] = ply + 1;
            Y.keep[g] = chosen;
            Y.stepm[g] = chosen;
        } else {
            rec[3] = -1;
            rec[4] = RZ_PLAY_RUNNING | RZ_PLAY_SEARCHED | RZ_PLAY_STALLED;
            Y.state[g] = kPlayStalled;
            E.active[g] = 0;   // the coming searches skip the slot until the host has decided
            Y.keep[g] = -2;
            Y.stepm[g] = -1;
        }
    }
}

// The rest of a move in ONE launch, one wave per slot: update_with_move with the move just drawn (advance_body: before the board
// changes), env.step + game_end_winner (step_body), then the end of a finished game (reset_player, game.py:128) and the refill of an
// idle slot from the queue of game ids (GomokuEnv.reset, gomoku_env.py:33-47: empty board, player 0; a fresh tree; the game's
// noise key).  drawn == 0 (no k_play_draw before it): only the refill.  The pending-priors counter of the game restarts here
// (the flush of the move's search ran just before: rz_deferred_flush leaves its own reset launch out between draw and apply).
__global__ __launch_bounds__(kWave) void k_play_apply(Dev E, Play Y, int drawn) {
    const int g = blockIdx.x, lane = threadIdx.x;
    const int step = Y.step_ab[1];
    if (g == 0 && lane == 0) Y.step_ab[0] = step + 1;
    const int keep = drawn ? Y.keep[g] : -2, mv = drawn ? Y.stepm[g] : -1;
    int state = Y.state[g];
    if (lane == 0 && drawn) {   // (rz_get_stats: after the move the arena holds the kept subtree only)
        const int top = E.top[g];
        if (top > Y.top_hwm[g]) Y.top_hwm[g] = top;
    }
    advance_body(E, g, lane, keep);
    __syncthreads();
    int who = -1;
    bool over = false;
    if (mv >= 0) step_body(E, g, lane, mv, who, over);   // (idle and stalled slots make no move)
    if (lane != 0) return;
    if (drawn && E.pend != nullptr) E.pend[g] = 0;
    if (state == kPlayRunning && mv >= 0 && over) {
        int32_t *rec = Y.log + ((long long)(step % Y.ring) * E.n_games + g) * Y.words;
        rec[4] |= RZ_PLAY_ENDED | ((who + 1) << 16);
        state = kPlayIdle;
        Y.state[g] = state;
        Y.game_id[g] = -1;
        E.active[g] = 0;
        fresh_root(E, g, E.cur_arena[g], 0);
    }
    Y.stepm[g] = -1;
    Y.keep[g] = -2;
    if (state != kPlayIdle) return;
    int head = Y.queue_ctl[0];
    int64_t gid = -1;
    while (head < Y.queue_ctl[1]) {
        const int seen = atomicCAS(&Y.queue_ctl[0], head, head + 1);
        if (seen == head) {
            gid = Y.queue_ids[head];
            break;
        }
        head = seen;
    }
    if (gid < 0) return;
    Y.game_id[g] = gid;
    Y.ply[g] = 0;
    Y.state[g] = kPlayRunning;
    Y.mailbox[g] = -1;
    for (int j = 0; j < 2 * kWords; ++j) E.root_stones[(long long)g * 2 * kWords + j] = 0ull;
    E.root_to_move[g] = 0;
    E.root_last[g] = -1;
    fresh_root(E, g, E.cur_arena[g], 0);
    E.noise_key[g] = mix64(mix64(Y.seed ^ 0x6E6F697365000000ull) ^ (uint64_t)gid);   // rlzero_amd/selfplay.py: _start
    E.noise_ctr[g] = 0;
    E.active[g] = 1;
}

__global__ void k_play_resolve(Play Y, int slot, int move) { Y.mailbox[slot] = move; }
__global__ void k_play_no_draw(Play Y) { Y.step_ab[1] = Y.step_ab[0]; }   // rz_play_apply without a draw: the step k_play_apply reads

__global__ void k_play_stop(Dev E, Play Y) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.n_games) return;
    Y.state[g] = kPlayIdle;
    Y.game_id[g] = -1;
    Y.mailbox[g] = -1;
    Y.keep[g] = -2;
    Y.stepm[g] = -1;
    E.active[g] = 0;
    fresh_root(E, g, E.cur_arena[g], 0);
}

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define RZ_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t err__ = (call);                                                           \
        if (err__ != hipSuccess)                                                             \
            return fail(err__ == hipErrorOutOfMemory ? RZ_ERR_OOM : RZ_ERR_HIP, "%s failed: %s", \
                        #call, hipGetErrorString(err__));                                    \
    } while (0)

}  // namespace

// shared with rz_net.hip: set the thread-local message returned by rz_last_error()
void rz_set_error(const char *msg) { snprintf(g_err, sizeof(g_err), "%s", msg ? msg : ""); }

struct rz_engine {
    rz_config cfg;
    Dev dev;
    std::vector<void *> allocs;
    long long bytes = 0;
    long long n_select = 0;
    int kb = 1, ks = 1;  // rz_set_in_flight: slots the next launches back up / select (sims_in_flight > 1 only)
    bool ml = false;     // sims_in_flight > 1: the level-synchronous kernel (default) instead of the sequential restatement
    int ml_lds = 0;
    double *d_logtab = nullptr;
    uint64_t *d_line_tab = nullptr;   // Dev::line_tab
    Play play = {};      // rz_play_attach
    bool play_on = false, play_drawn = false;
    long long play_steps = 0;   // move steps enqueued (rz_play_apply calls) since rz_play_attach
};

namespace {

template <typename T>
int dev_alloc(rz_engine *e, T **out, long long count) {
    void *p = nullptr;
    const size_t bytes = (size_t)count * sizeof(T);
    hipError_t err = hipMalloc(&p, bytes ? bytes : 8);
    if (err != hipSuccess)
        return fail(RZ_ERR_OOM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(err));
    e->allocs.push_back(p);
    e->bytes += (long long)bytes;
    *out = (T *)p;
    return RZ_OK;
}

int check_engine(rz_engine *e) {
    if (e == nullptr) return fail(RZ_ERR_ARG, "engine handle is NULL");
    int cur = -1;
    RZ_HIP(hipGetDevice(&cur));
    if (cur != e->cfg.device) RZ_HIP(hipSetDevice(e->cfg.device));
    return RZ_OK;
}

inline hipStream_t as_stream(void *s) { return (hipStream_t)s; }

int launched(const char *what) {
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return fail(RZ_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(err));
    return RZ_OK;
}

}  // namespace

extern "C" {

int rz_abi_version(void) { return RZ_ABI_VERSION; }
#ifndef RZ_SOURCE_HASH
#define RZ_SOURCE_HASH "unknown"   /* rlzero_amd/_build.py passes the hash of sources + headers + flags */
#endif
// the marker is what rlzero_amd/_build.py searches the library's bytes for (no need to load it); the accessor returns the hash alone
const char *rz_source_hash(void) {
    static const char marker[] = "RZ_SOURCE_HASH=" RZ_SOURCE_HASH;
    return marker + 15;
}
const char *rz_last_error(void) { return g_err; }

int rz_create(const rz_config *cfg, rz_engine **out) {
    if (cfg == nullptr || out == nullptr) return fail(RZ_ERR_ARG, "cfg/out is NULL");
    *out = nullptr;
    if (cfg->abi_version != RZ_ABI_VERSION)
        return fail(RZ_ERR_ARG, "abi_version %d != library %d", cfg->abi_version, RZ_ABI_VERSION);
    int BH, BW, A, n_row = cfg->n_in_row;
    if (cfg->game_kind == RZ_GAME_GOMOKU) {
        if (cfg->board_size < 1 || cfg->board_size > RZ_MAX_BOARD_SIZE)
            return fail(RZ_ERR_ARG, "board_size %d not in 1..%d", cfg->board_size, RZ_MAX_BOARD_SIZE);
        BH = BW = cfg->board_size;
        A = BH * BW;
    } else if (cfg->game_kind == RZ_GAME_CONNECT4) {
        BH = cfg->board_height > 0 ? cfg->board_height : 6;
        BW = cfg->board_width > 0 ? cfg->board_width : 7;
        if (n_row == 0) n_row = 4;
        if (BH > RZ_MAX_BOARD_SIZE || BW > RZ_MAX_BOARD_SIZE)
            return fail(RZ_ERR_ARG, "board %dx%d exceeds %d", BH, BW, RZ_MAX_BOARD_SIZE);
        A = BW;
    } else {
        return fail(RZ_ERR_ARG, "unknown game_kind %d", cfg->game_kind);
    }
    if (n_row < 1 || (n_row > BH && n_row > BW)) return fail(RZ_ERR_ARG, "n_in_row %d does not fit the board", n_row);
    if (cfg->n_games < 1) return fail(RZ_ERR_ARG, "n_games must be >= 1");
    if (cfg->n_playout < 1) return fail(RZ_ERR_ARG, "n_playout must be >= 1");
    if (cfg->score_mode != RZ_SCORE_UCT_REF && cfg->score_mode != RZ_SCORE_PUCT)
        return fail(RZ_ERR_ARG, "unknown score_mode %d", cfg->score_mode);
    if (!(cfg->c_puct >= 0.0)) return fail(RZ_ERR_ARG, "c_puct must be >= 0");
    if (cfg->sims_in_flight < 0 || cfg->sims_in_flight > RZ_MAX_IN_FLIGHT)
        return fail(RZ_ERR_ARG, "sims_in_flight %d not in 0..%d", cfg->sims_in_flight, RZ_MAX_IN_FLIGHT);
    if (cfg->in_flight_impl != 0 && cfg->in_flight_impl != 1) return fail(RZ_ERR_ARG, "unknown in_flight_impl %d", cfg->in_flight_impl);
    int n_dev = 0;
    RZ_HIP(hipGetDeviceCount(&n_dev));
    if (cfg->device < 0 || cfg->device >= n_dev)
        return fail(RZ_ERR_ARG, "device %d not in 0..%d", cfg->device, n_dev - 1);
    RZ_HIP(hipSetDevice(cfg->device));

    rz_engine *e = new (std::nothrow) rz_engine();
    if (e == nullptr) return fail(RZ_ERR_OOM, "host allocation failed");
    e->cfg = *cfg;
    Dev &D = e->dev;
    memset(&D, 0, sizeof(D));
    const int S = BH * BW;
    const double pf = cfg->pool_factor > 0.0 ? cfg->pool_factor : 2.0;
    D.kind = cfg->game_kind;
    D.BH = BH;
    D.BW = BW;
    D.S = S;
    D.A = A;
    D.n_row = n_row;
    D.bw_rcp = (65536 + BW - 1) / BW;
    for (int y = 0; y < BH && S <= 64; ++y) D.col0 |= 1ull << (y * BW);
    D.n_rcp = (65536 + n_row - 1) / n_row;
    D.n_games = cfg->n_games;
    D.n_playout = cfg->n_playout;
    D.K = cfg->sims_in_flight > 1 ? cfg->sims_in_flight : 1;
    e->kb = e->ks = D.K;
    e->ml = D.K > 1 && cfg->in_flight_impl == 0;
    if (e->ml) {
        e->ml_lds = ml_lds_bytes(D.K, A, cfg->score_mode == RZ_SCORE_PUCT);
        if (e->ml_lds > 160 * 1024) {
            const int need = e->ml_lds;
            delete e;
            return fail(RZ_ERR_ARG, "sims_in_flight %d x %d actions needs %d bytes of LDS (> 160 KB)", cfg->sims_in_flight, A, need);
        }
        // the attribute belongs to the kernel, not to this engine: set to the hardware's 160 KB once and for all, so that a
        // later engine with a smaller K x A cannot lower it under an earlier, larger one
        const int most = 160 * 1024;
        hipError_t a1 = hipFuncSetAttribute((const void *)k_tree_step_ml<true>, hipFuncAttributeMaxDynamicSharedMemorySize, most);
        hipError_t a2 = hipFuncSetAttribute((const void *)k_tree_step_ml<false>, hipFuncAttributeMaxDynamicSharedMemorySize, most);
        if (a1 != hipSuccess || a2 != hipSuccess) {
            delete e;
            return fail(RZ_ERR_HIP, "hipFuncSetAttribute(dynamic LDS %d bytes) failed", most);
        }
    }
    D.score_mode = cfg->score_mode;
    D.add_noise = cfg->add_noise ? 1 : 0;
    D.noise_seed = (uint64_t)(uint32_t)cfg->noise_seed;
    D.c_puct = cfg->c_puct;
    // qcap expanded nodes (= prior blocks) per arena: this move's n_playout plus a carried subtree of
    // up to pf * n_playout.  Record slots: dense (PUCT) = one K-block per expansion; otherwise only
    // visited nodes hold records, at most ~3 slots per simulation with the doubling child vectors
    // (typically ~2), so 16 per expansion leaves a wide margin.
    D.qcap = (int)((pf + 1.0) * (double)cfg->n_playout) + 8;
    D.pcap = (long long)D.qcap * A;
    D.cap = cfg->score_mode == RZ_SCORE_PUCT ? (long long)D.qcap * A + 2
                                             : (long long)D.qcap * 16 + 2 * A + 64;
    D.path_stride = S + 2;
    D.logtab_n = (long long)cfg->n_playout * (S + 1) + 2;
    for (int j = 0; j < kWords; ++j) {
        const int lo = 64 * j;
        D.valid[j] = S >= lo + 64 ? ~0ull : (S > lo ? ((1ull << (S - lo)) - 1ull) : 0ull);
    }
    const long long G = cfg->n_games, GL = G * D.K;  // games, leaves in flight
    const long long slots = G * 2 * D.cap;
    if (D.cap >= (1ll << 30) || D.pcap >= (1ll << 31))
    {
        delete e;
        return fail(RZ_ERR_ARG, "n_playout %d is too large for 32-bit slot indices", cfg->n_playout);
    }
    int rc = RZ_OK;
#define RZ_ALLOC(field, count)                               \
    if (rc == RZ_OK) rc = dev_alloc(e, &D.field, (count))
    RZ_ALLOC(R, slots * 2);
    RZ_ALLOC(P, G * 2 * D.pcap);
    RZ_ALLOC(cur_arena, G);
    RZ_ALLOC(top, G);
    RZ_ALLOC(ptop, G);
    RZ_ALLOC(nblk, G);
    RZ_ALLOC(root_stones, G * 2 * kWords);
    RZ_ALLOC(root_to_move, G);
    RZ_ALLOC(root_last, G);
    RZ_ALLOC(active, G);
    RZ_ALLOC(path, GL * D.path_stride);
    RZ_ALLOC(leaf_node, GL);
    RZ_ALLOC(leaf_depth, GL);
    RZ_ALLOC(leaf_fresh, GL);
    RZ_ALLOC(leaf_term, GL);
    RZ_ALLOC(leaf_tval, GL);
    RZ_ALLOC(leaf_stones, GL * 2 * kWords);
    RZ_ALLOC(leaf_to_move, GL);
    RZ_ALLOC(leaf_last, GL);
    RZ_ALLOC(queue, G * D.qcap);
    RZ_ALLOC(err, G);
    RZ_ALLOC(err_any, 1);
    RZ_ALLOC(reuse_drops, 1);
    RZ_ALLOC(noise_ctr, G);
    RZ_ALLOC(noise_key, G);
    if (rc == RZ_OK) rc = dev_alloc(e, &e->d_logtab, D.logtab_n);
    if (rc == RZ_OK) rc = dev_alloc(e, &e->d_line_tab, 2 * kWave);
#undef RZ_ALLOC
    if (rc != RZ_OK) {
        rz_destroy(e);
        return rc;
    }
    D.logtab = e->d_logtab;
    D.line_tab = e->d_line_tab;
    // zero the small state; arenas need no initialisation beyond the root slot
    hipError_t herr = hipSuccess;
    auto zero = [&](void *p, size_t bytes) { if (herr == hipSuccess) herr = hipMemset(p, 0, bytes); };
    zero(D.cur_arena, G * 4); zero(D.top, G * 4); zero(D.ptop, G * 4); zero(D.nblk, G * 4);
    zero(D.root_stones, G * 2 * kWords * 8); zero(D.root_to_move, G * 4);
    zero(D.leaf_node, GL * 4); zero(D.leaf_depth, GL * 4); zero(D.leaf_fresh, GL * 4);
    zero(D.leaf_term, GL * 4); zero(D.leaf_tval, GL * 8); zero(D.leaf_stones, GL * 2 * kWords * 8);
    zero(D.leaf_to_move, GL * 4); zero(D.path, GL * D.path_stride * 4);
    zero(D.err, G * 4); zero(D.err_any, 4); zero(D.reuse_drops, 4); zero(D.noise_ctr, G * 4);
    if (herr == hipSuccess) herr = hipMemset(D.root_last, 0xff, G * 4);  // -1
    if (herr == hipSuccess) herr = hipMemset(D.leaf_last, 0xff, GL * 4);
    if (herr == hipSuccess) herr = hipMemset(D.active, 1, G);
    if (herr != hipSuccess) {
        rz_destroy(e);
        return fail(RZ_ERR_HIP, "hipMemset failed: %s", hipGetErrorString(herr));
    }
    {
        std::vector<double> tab((size_t)D.logtab_n);
        tab[0] = 0.0;
        for (long long i = 1; i < D.logtab_n; ++i) tab[(size_t)i] = std::log((double)i);
        herr = hipMemcpy(e->d_logtab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice);
        if (herr != hipSuccess) {
            rz_destroy(e);
            return fail(RZ_ERR_HIP, "hipMemcpy(log table) failed: %s", hipGetErrorString(herr));
        }
    }
    {   // Dev::line_tab: the window of lane l -- bits j * stride of its n cells (boards of up to 128 cells: a low word [l] and a high word
        // [64 + l], the high one empty unless a window spans 64 cell numbers or more); boards of more cells: its first n - 1 cells in [l],
        // and line_masks says whether those fit
        uint64_t tab[2 * kWave] = {0};
        const int cells = S <= 128 ? n_row : n_row - 1;
        D.line_masks = (cells - 1) * (BW + 1) < 64 ? 1 : 0;
        for (int l = 0; l < kWave && l < 4 * n_row; ++l) {
            const int d = l / n_row, stride = d == 0 ? 1 : d == 1 ? BW : d == 2 ? BW + 1 : BW - 1;
            for (int j = 0; j < cells; ++j) {
                const int o = j * stride;
                if (o < 64) tab[l] |= 1ull << o;
                else if (o < 128) tab[kWave + l] |= 1ull << (o - 64);
            }
        }
        herr = hipMemcpy(e->d_line_tab, tab, sizeof(tab), hipMemcpyHostToDevice);
        if (herr != hipSuccess) {
            rz_destroy(e);
            return fail(RZ_ERR_HIP, "hipMemcpy(line table) failed: %s", hipGetErrorString(herr));
        }
    }
    {   // fresh trees
        std::vector<int32_t> mv((size_t)G, -1);
        int32_t *d_mv = nullptr;
        herr = hipMalloc((void **)&d_mv, G * 4);
        if (herr == hipSuccess) herr = hipMemcpy(d_mv, mv.data(), G * 4, hipMemcpyHostToDevice);
        if (herr == hipSuccess) {
            k_advance<<<dim3((unsigned)G), dim3(kWave), 0, 0>>>(D, d_mv);
            k_set_noise_keys<<<dim3((unsigned)((G + 255) / 256)), dim3(256), 0, 0>>>(D, nullptr, nullptr);   // the default keys
            herr = hipDeviceSynchronize();
        }
        if (d_mv) (void)hipFree(d_mv);
        if (herr != hipSuccess) {
            rz_destroy(e);
            return fail(RZ_ERR_HIP, "tree initialisation failed: %s", hipGetErrorString(herr));
        }
    }
    *out = e;
    return RZ_OK;
}

int rz_destroy(rz_engine *e) {
    if (e == nullptr) return RZ_OK;
    (void)hipSetDevice(e->cfg.device);
    (void)hipDeviceSynchronize();
    for (void *p : e->allocs) (void)hipFree(p);
    delete e;
    return RZ_OK;
}

int rz_geometry(rz_engine *e, int32_t *height, int32_t *width, int32_t *n_actions) {
    if (e == nullptr) return fail(RZ_ERR_ARG, "engine handle is NULL");
    if (height) *height = e->dev.BH;
    if (width) *width = e->dev.BW;
    if (n_actions) *n_actions = e->dev.A;
    return RZ_OK;
}

int rz_upload_log_table(rz_engine *e, const double *h_table, int64_t count) {
    int rc = check_engine(e);
    if (rc != RZ_OK) return rc;
    if (h_table == nullptr || count < 2) return fail(RZ_ERR_ARG, "table is NULL or too short");
    if (count > e->dev.logtab_n) count = e->dev.logtab_n;
    RZ_HIP(hipDeviceSynchronize());
    RZ_HIP(hipMemcpy(e->d_logtab, h_table, (size_t)count * sizeof(double), hipMemcpyHostToDevice));
    return RZ_OK;
}

int rz_log_table_size(rz_engine *e, int64_t *count) {
    if (e == nullptr || count == nullptr) return fail(RZ_ERR_ARG, "NULL argument");
    *count = e->dev.logtab_n;
    return RZ_OK;
}

#define RZ_ENTER(e)                      \
    do {                                 \
        int rc__ = check_engine(e);      \
        if (rc__ != RZ_OK) return rc__;  \
    } while (0)
#define RZ_NEED(p)                                                       \
    do {                                                                 \
        if ((p) == nullptr) return fail(RZ_ERR_ARG, "%s is NULL", #p);   \
    } while (0)

static inline dim3 per_game(const rz_engine *e) { return dim3((unsigned)e->cfg.n_games); }
static inline dim3 per_leaf(const rz_engine *e) { return dim3((unsigned)(e->cfg.n_games * e->dev.K)); }
static inline dim3 flat_grid(const rz_engine *e) { return dim3((unsigned)((e->cfg.n_games + 255) / 256)); }

int rz_set_roots(rz_engine *e, const uint64_t *d_stones, const int32_t *d_to_move,
                 const int32_t *d_last_move, const uint8_t *d_mask, int reset_trees, void *stream) {
    RZ_ENTER(e);
    RZ_NEED(d_stones); RZ_NEED(d_to_move); RZ_NEED(d_last_move);
    k_set_roots<<<flat_grid(e), dim3(256), 0, as_stream(stream)>>>(e->dev, d_stones, d_to_move,
                                                                   d_last_move, d_mask, reset_trees);
    return launched("k_set_roots");
}

int rz_get_roots(rz_engine *e, uint64_t *d_stones, int32_t *d_to_move, int32_t *d_last_move,
                 void *stream) {
    RZ_ENTER(e);
    RZ_NEED(d_stones); RZ_NEED(d_to_move); RZ_NEED(d_last_move);
    k_get_roots<<<flat_grid(e), dim3(256), 0, as_stream(stream)>>>(e->dev, d_stones, d_to_move, d_last_move);
    return launched("k_get_roots");
}

int rz_set_active(rz_engine *e, const uint8_t *d_active, void *stream) {
    RZ_ENTER(e);
    k_set_active<<<flat_grid(e), dim3(256), 0, as_stream(stream)>>>(e->dev, d_active);
    return launched("k_set_active");
}

int rz_set_noise_keys(rz_engine *e, const uint64_t *d_keys, const uint8_t *d_mask, void *stream) {
    RZ_ENTER(e);
    k_set_noise_keys<<<flat_grid(e), dim3(256), 0, as_stream(stream)>>>(e->dev, d_keys, d_mask);
    return launched("k_set_noise_keys");
}

int rz_set_in_flight(rz_engine *e, int32_t k_backup, int32_t k_select) {
    if (e == nullptr) return fail(RZ_ERR_ARG, "engine handle is NULL");
    if (k_backup < 0 || k_backup > e->dev.K || k_select < 0 || k_select > e->dev.K)
        return fail(RZ_ERR_ARG, "in-flight counts (%d, %d) not in 0..%d", k_backup, k_select, e->dev.K);
    e->kb = k_backup;
    e->ks = k_select;
    return RZ_OK;
}

int rz_select_step(rz_engine *e, float *d_obs, void *stream) {
    RZ_ENTER(e);
    e->n_select += 1;
    if (e->dev.K > 1) {
        if (e->ml) k_tree_step_ml<false><<<per_game(e), dim3(kWave * e->dev.K), e->ml_lds, as_stream(stream)>>>(e->dev, nullptr, nullptr, RawHeads(), d_obs, 0, e->ks);
        else k_tree_step_vl<false><<<per_game(e), dim3(kWave), 0, as_stream(stream)>>>(e->dev, nullptr, nullptr, RawHeads(), d_obs, 0, e->ks);
        return launched("k_tree_step_vl");
    }
    k_select<<<per_game(e), dim3(kWave), 0, as_stream(stream)>>>(e->dev, d_obs);
    return launched("k_select");
}

int rz_leaf_buffers(rz_engine *e, const uint64_t **d_stones, const int32_t **d_to_move, const int32_t **d_last_cell) {
    if (e == nullptr || !d_stones || !d_to_move || !d_last_cell) return fail(RZ_ERR_ARG, "NULL argument");
    *d_stones = e->dev.leaf_stones;
    *d_to_move = e->dev.leaf_to_move;
    *d_last_cell = e->dev.leaf_last;
    return RZ_OK;
}

int rz_encode_leaf_obs(rz_engine *e, float *d_obs, void *stream) {
    RZ_ENTER(e);
    RZ_NEED(d_obs);
    k_encode<<<per_leaf(e), dim3(kWave), 0, as_stream(stream)>>>(e->dev, 0, d_obs);
    return launched("k_encode");
}

int rz_encode_root_obs(rz_engine *e, float *d_obs, void *stream) {
    RZ_ENTER(e);
    RZ_NEED(d_obs);
    k_encode<<<per_game(e), dim3(kWave), 0, as_stream(stream)>>>(e->dev, 1, d_obs);
    return launched("k_encode");
}

int rz_get_leaves(rz_engine *e, uint64_t *d_stones, int32_t *d_to_move, int32_t *d_last_move,
                  int32_t *d_terminal, void *stream) {
    RZ_ENTER(e);
    RZ_NEED(d_stones); RZ_NEED(d_to_move); RZ_NEED(d_last_move); RZ_NEED(d_terminal);
    if (e->dev.K > 1) return fail(RZ_ERR_ARG, "rz_get_leaves serves host evaluators: not available with sims_in_flight > 1");
    k_get_leaves<<<flat_grid(e), dim3(256), 0, as_stream(stream)>>>(e->dev, d_stones, d_to_move,
                                                                    d_last_move, d_terminal);
    return launched("k_get_leaves");
}

int rz_eval_synthetic(rz_engine *e, int kind, float *d_logp, float *d_value, void *stream) {
    RZ_ENTER(e);
    RZ_NEED(d_value);
    if (kind != RZ_EVAL_V0 && kind != RZ_EVAL_VLIN) return fail(RZ_ERR_ARG, "unknown evaluator %d", kind);
    k_eval_synth<<<per_leaf(e), dim3(kWave), 0, as_stream(stream)>>>(e->dev, kind, d_logp, d_value);
    return launched("k_eval_synth");
}

int rz_eval_rollout(rz_engine *e, uint64_t seed, uint32_t sim_index, int32_t n_limit, float *d_value,
                    void *stream) {
    RZ_ENTER(e);
    RZ_NEED(d_value);
    if (n_limit < 0) return fail(RZ_ERR_ARG, "n_limit must be >= 0");
    k_eval_rollout<<<per_leaf(e), dim3(kWave), 0, as_stream(stream)>>>(e->dev, seed, sim_index, n_limit, d_value);
    return launched("k_eval_rollout");
}

int rz_expand_backup(rz_engine *e, const float *d_logp, const float *d_value, void *stream) {
    RZ_ENTER(e);
    RZ_NEED(d_value);
    if (e->dev.K > 1) {
        if (e->ml) k_tree_step_ml<false><<<per_game(e), dim3(kWave * e->dev.K), e->ml_lds, as_stream(stream)>>>(e->dev, d_logp, d_value, RawHeads(), nullptr, e->kb, 0);
        else k_tree_step_vl<false><<<per_game(e), dim3(kWave), 0, as_stream(stream)>>>(e->dev, d_logp, d_value, RawHeads(), nullptr, e->kb, 0);
        return launched("k_tree_step_vl");
    }
    k_expand_backup<float><<<per_game(e), dim3(kWave), 0, as_stream(stream)>>>(e->dev, d_logp, d_value);
    return launched("k_expand_backup");
}

int rz_expand_backup_f64(rz_engine *e, const float *d_logp, const double *d_value, void *stream) {
    RZ_ENTER(e);
    RZ_NEED(d_value);
    if (e->dev.K > 1) return fail(RZ_ERR_ARG, "host evaluators are not available with sims_in_flight > 1");
    k_expand_backup<double><<<per_game(e), dim3(kWave), 0, as_stream(stream)>>>(e->dev, d_logp, d_value);
    return launched("k_expand_backup");
}

int rz_expand_backup_probs(rz_engine *e, const float *d_probs, const double *d_value, void *stream) {
    RZ_ENTER(e);
    RZ_NEED(d_value);
    if (e->dev.K > 1) return fail(RZ_ERR_ARG, "host evaluators are not available with sims_in_flight > 1");
    k_expand_backup<double, true><<<per_game(e), dim3(kWave), 0, as_stream(stream)>>>(e->dev, d_probs, d_value);
    return launched("k_expand_backup");
}

int rz_tree_step(rz_engine *e, const float *d_logp, const float *d_value, float *d_obs, void *stream) {
    RZ_ENTER(e);
    RZ_NEED(d_value);
    e->n_select += 1;
    if (e->dev.K > 1) {
        if (e->ml) k_tree_step_ml<false><<<per_game(e), dim3(kWave * e->dev.K), e->ml_lds, as_stream(stream)>>>(e->dev, d_logp, d_value, RawHeads(), d_obs, e->kb, e->ks);
        else k_tree_step_vl<false><<<per_game(e), dim3(kWave), 0, as_stream(stream)>>>(e->dev, d_logp, d_value, RawHeads(), d_obs, e->kb, e->ks);
        return launched("k_tree_step_vl");
    }
    k_tree_step<float><<<per_game(e), dim3(kWave), 0, as_stream(stream)>>>(e->dev, d_logp, d_value, d_obs);
    return launched("k_tree_step");
}

static int raw_heads_ok(rz_engine *e, const rz_raw_heads *h) {
    if (!h || !h->raw || !h->hid || !h->w2 || !h->b2) return fail(RZ_ERR_ARG, "NULL device pointer");
    if (h->ld < e->dev.A) return fail(RZ_ERR_ARG, "ld %d < n_actions %d", h->ld, e->dev.A);
    if (h->n_parts != 1 && h->n_parts != 4) return fail(RZ_ERR_ARG, "n_parts %d is neither 1 nor 4", h->n_parts);
    if (h->n_parts == 4 && (!h->act_scale || !h->act_bias || !h->val_scale || !h->val_bias))
        return fail(RZ_ERR_ARG, "partial sums need their scales and biases");
    return RZ_OK;
}

int rz_expand_backup_raw(rz_engine *e, const rz_raw_heads *heads, void *stream) {
    RZ_ENTER(e);
    int rc = raw_heads_ok(e, heads);
    if (rc != RZ_OK) return rc;
    const RawHeads rh = *heads;
    if (e->dev.K > 1) {
        if (e->ml) k_tree_step_ml<true><<<per_game(e), dim3(kWave * e->dev.K), e->ml_lds, as_stream(stream)>>>(e->dev, nullptr, nullptr, rh, nullptr, e->kb, 0);
        else k_tree_step_vl<true><<<per_game(e), dim3(kWave), 0, as_stream(stream)>>>(e->dev, nullptr, nullptr, rh, nullptr, e->kb, 0);
        return launched("k_tree_step_vl");
    }
    k_expand_backup_raw<<<per_game(e), dim3(kWave), 0, as_stream(stream)>>>(e->dev, rh);
    return launched("k_expand_backup_raw");
}

static int deferred_ok(rz_engine *e, const rz_value_head *h) {
    if (h == nullptr || !h->valfeat || !h->w1t || !h->b1 || !h->w2 || !h->b2) return fail(RZ_ERR_ARG, "rz_value_head: NULL pointer");
    if ((h->groups != 16 && h->groups != 32 && h->groups != 64 && h->groups != 128) || h->ld != 4 * h->groups)
        return fail(RZ_ERR_ARG, "rz_value_head: groups = %d must be 16, 32, 64 or 128 and ld = 4 * groups", h->groups);
    if (2 * e->dev.S > 4 * h->groups)   // (the kernels also size their bitboard arithmetic by it: words_of_per)
        return fail(RZ_ERR_ARG, "rz_value_head: %d groups of 4 inputs do not hold the 2 x %d inputs of this board", h->groups, e->dev.S);
    if (e->dev.pend_cap <= 0) return fail(RZ_ERR_ARG, "call rz_deferred_reserve first");
    return RZ_OK;
}

int rz_deferred_reserve(rz_engine *e, int32_t slots) {
    int rc = check_engine(e);
    if (rc != RZ_OK) return rc;
    if (slots < 1) return fail(RZ_ERR_ARG, "rz_deferred_reserve: slots must be positive");
    if (e->cfg.score_mode != RZ_SCORE_UCT_REF || e->dev.K != 1)
        return fail(RZ_ERR_ARG, "deferred priors need RZ_SCORE_UCT_REF (the rule that never reads a prior) and sims_in_flight == 1");
    if (slots <= e->dev.pend_cap) return RZ_OK;
    RZ_HIP(hipDeviceSynchronize());
    const long long G = e->cfg.n_games;
    // (the records of a smaller reservation stay allocated until rz_destroy: reservations grow once or twice in a process)
    if ((rc = dev_alloc(e, &e->dev.pend_pb, (long long)slots * G)) != RZ_OK) return rc;
    if ((rc = dev_alloc(e, &e->dev.pend_ctr, (long long)slots * G)) != RZ_OK) return rc;
    if ((rc = dev_alloc(e, &e->dev.pend_stones, (long long)slots * G * 2 * kWords)) != RZ_OK) return rc;
    if (e->dev.pend == nullptr) {
        if ((rc = dev_alloc(e, &e->dev.pend, G)) != RZ_OK) return rc;
    }
    RZ_HIP(hipMemset(e->dev.pend, 0, (size_t)G * sizeof(int32_t)));
    e->dev.pend_cap = slots;
    return RZ_OK;
}

int rz_trace_attach(rz_engine *e, void *d_trace) {
    int rc = check_engine(e);
    if (rc != RZ_OK) return rc;
    e->dev.trace = (unsigned long long *)d_trace;
    return RZ_OK;
}

int rz_device_view(rz_engine *e, void *out, int64_t out_bytes) {
    int rc = check_engine(e);
    if (rc != RZ_OK) return rc;
    if (out == nullptr || out_bytes != (int64_t)sizeof(Dev)) return fail(RZ_ERR_ARG, "rz_device_view: the caller's view is %lld bytes, the engine's %zu", (long long)out_bytes, sizeof(Dev));
    memcpy(out, &e->dev, sizeof(Dev));
    return RZ_OK;
}

int rz_deferred_slots(rz_engine *e, const int32_t **d_slot_of_game) {
    int rc = check_engine(e);
    if (rc != RZ_OK) return rc;
    if (!d_slot_of_game) return fail(RZ_ERR_ARG, "NULL output pointer");
    if (e->dev.pend_cap <= 0) return fail(RZ_ERR_ARG, "call rz_deferred_reserve first");
    *d_slot_of_game = e->dev.pend;
    return RZ_OK;
}

int rz_expand_backup_deferred(rz_engine *e, const rz_value_head *head, void *stream) {
    int rc = check_engine(e);
    if (rc != RZ_OK) return rc;
    if ((rc = deferred_ok(e, head)) != RZ_OK) return rc;
    const dim3 block(kWave * kDefWaves);
    switch (head->groups) {   // = 4 waves x 2 halves x PER
        case 16: k_expand_backup_def<2><<<per_game(e), block, 0, as_stream(stream)>>>(e->dev, *head); break;
        case 32: k_expand_backup_def<4><<<per_game(e), block, 0, as_stream(stream)>>>(e->dev, *head); break;
        case 64: k_expand_backup_def<8><<<per_game(e), block, 0, as_stream(stream)>>>(e->dev, *head); break;
        default: k_expand_backup_def<16><<<per_game(e), block, 0, as_stream(stream)>>>(e->dev, *head); break;
    }
    return launched("k_expand_backup_def");
}

int rz_tree_step_deferred(rz_engine *e, const rz_value_head *head, void *stream) {
    int rc = check_engine(e);
    if (rc != RZ_OK) return rc;
    if ((rc = deferred_ok(e, head)) != RZ_OK) return rc;
    e->n_select += 1;
    const dim3 block(kWave * kDefWaves);
    if (e->dev.trace && head->groups == 128) {   // (the traced instantiation exists for the 15 x 15 board's head: rz_trace_attach)
        k_tree_step_def<16, true><<<per_game(e), block, 0, as_stream(stream)>>>(e->dev, *head, nullptr);
        return launched("k_tree_step_def");
    }
    switch (head->groups) {   // = 4 waves x 2 halves x PER
        case 16: k_tree_step_def<2, false><<<per_game(e), block, 0, as_stream(stream)>>>(e->dev, *head, nullptr); break;
        case 32: k_tree_step_def<4, false><<<per_game(e), block, 0, as_stream(stream)>>>(e->dev, *head, nullptr); break;
        case 64: k_tree_step_def<8, false><<<per_game(e), block, 0, as_stream(stream)>>>(e->dev, *head, nullptr); break;
        default: k_tree_step_def<16, false><<<per_game(e), block, 0, as_stream(stream)>>>(e->dev, *head, nullptr); break;
    }
    return launched("k_tree_step_def");
}

int rz_deferred_flush(rz_engine *e, const rz_deferred_logits *logits, int32_t n_slots, void *stream) {
    int rc = check_engine(e);
    if (rc != RZ_OK) return rc;
    if (e->dev.pend_cap <= 0 || n_slots <= 0) return RZ_OK;
    if (!logits || !logits->raw) return fail(RZ_ERR_ARG, "rz_deferred_logits: NULL pointer");
    if (n_slots > e->dev.pend_cap) return fail(RZ_ERR_ARG, "more slots than rz_deferred_reserve()d");
    if (logits->ld < e->dev.A || logits->rows_per_slot < e->cfg.n_games) return fail(RZ_ERR_ARG, "rz_deferred_logits: rows shorter than the batch");
    k_deferred_priors<<<dim3((unsigned)e->cfg.n_games, (unsigned)n_slots), dim3(kWave), 0, as_stream(stream)>>>(
        e->dev, logits->raw, logits->ld, (long long)logits->rows_per_slot);
    // (between rz_play_draw and rz_play_apply the counters restart in k_play_apply: one launch less in the chain of a move)
    if (!(e->play_on && e->play_drawn)) k_deferred_reset<<<flat_grid(e), dim3(256), 0, as_stream(stream)>>>(e->dev);
    return launched("k_deferred_priors");
}

int rz_tree_step_raw(rz_engine *e, const rz_raw_heads *heads, float *d_obs, void *stream) {
    RZ_ENTER(e);
    int rc = raw_heads_ok(e, heads);
    if (rc != RZ_OK) return rc;
    e->n_select += 1;
    const RawHeads rh = *heads;
    if (e->dev.K > 1) {
        if (e->ml) k_tree_step_ml<true><<<per_game(e), dim3(kWave * e->dev.K), e->ml_lds, as_stream(stream)>>>(e->dev, nullptr, nullptr, rh, d_obs, e->kb, e->ks);
        else k_tree_step_vl<true><<<per_game(e), dim3(kWave), 0, as_stream(stream)>>>(e->dev, nullptr, nullptr, rh, d_obs, e->kb, e->ks);
        return launched("k_tree_step_vl");
    }
    k_tree_step_raw<<<per_game(e), dim3(kWave), 0, as_stream(stream)>>>(e->dev, rh, d_obs);
    return launched("k_tree_step_raw");
}

int rz_root_visits(rz_engine *e, int32_t *d_visits, void *stream) {
    RZ_ENTER(e);
    RZ_NEED(d_visits);
    k_root_children<<<per_game(e), dim3(kWave), 0, as_stream(stream)>>>(e->dev, 0, d_visits);
    return launched("k_root_children");
}

int rz_root_wsum(rz_engine *e, double *d_w, void *stream) {
    RZ_ENTER(e);
    RZ_NEED(d_w);
    k_root_children<<<per_game(e), dim3(kWave), 0, as_stream(stream)>>>(e->dev, 1, d_w);
    return launched("k_root_children");
}

int rz_root_priors(rz_engine *e, float *d_p, void *stream) {
    RZ_ENTER(e);
    RZ_NEED(d_p);
    k_root_children<<<per_game(e), dim3(kWave), 0, as_stream(stream)>>>(e->dev, 2, d_p);
    return launched("k_root_children");
}

int rz_root_stats(rz_engine *e, int32_t *d_n, double *d_w, void *stream) {
    RZ_ENTER(e);
    RZ_NEED(d_n); RZ_NEED(d_w);
    k_root_stats<<<flat_grid(e), dim3(256), 0, as_stream(stream)>>>(e->dev, d_n, d_w);
    return launched("k_root_stats");
}

int rz_advance_roots(rz_engine *e, const int32_t *d_moves, void *stream) {
    RZ_ENTER(e);
    RZ_NEED(d_moves);
    k_advance<<<per_game(e), dim3(kWave), 0, as_stream(stream)>>>(e->dev, d_moves);
    return launched("k_advance");
}

int rz_step_games(rz_engine *e, const int32_t *d_moves, int32_t *d_winner, uint8_t *d_ended, void *stream) {
    RZ_ENTER(e);
    RZ_NEED(d_moves);
    k_step_games<<<per_game(e), dim3(kWave), 0, as_stream(stream)>>>(e->dev, d_moves, d_winner, d_ended);
    return launched("k_step_games");
}

// ------------------------------------------------------------------ the move step on the device
int rz_play_attach(rz_engine *e, const rz_play_config *cfg) {
    int rc = check_engine(e);
    if (rc != RZ_OK) return rc;
    RZ_NEED(cfg);
    if (e->dev.K != 1) return fail(RZ_ERR_ARG, "rz_play_attach: one simulation in flight per tree only");
    if (!cfg->d_queue_ids || !cfg->d_queue_ctl || !cfg->d_log || cfg->ring_steps < 2) return fail(RZ_ERR_ARG, "rz_play_attach: queue, log and a ring of >= 2 steps are needed");
    if (!(cfg->temperature > 0.0)) return fail(RZ_ERR_ARG, "rz_play_attach: temperature must be positive");
    if (cfg->stall_margin < 0.0 || cfg->stall_margin >= 0.5) return fail(RZ_ERR_ARG, "rz_play_attach: stall_margin not in [0, 0.5)");
    RZ_HIP(hipDeviceSynchronize());
    Play &Y = e->play;
    const long long G = e->cfg.n_games;
    if (Y.top_hwm == nullptr) {   // (keyed on the array allocated LAST; an earlier attach that ran out of memory is retried array by array)
        memset(&Y, 0, sizeof(Y));
        if (Y.game_id == nullptr && (rc = dev_alloc(e, &Y.game_id, G)) != RZ_OK) return rc;
        if (Y.ply == nullptr && (rc = dev_alloc(e, &Y.ply, G)) != RZ_OK) return rc;
        if (Y.state == nullptr && (rc = dev_alloc(e, &Y.state, G)) != RZ_OK) return rc;
        if (Y.mailbox == nullptr && (rc = dev_alloc(e, &Y.mailbox, G)) != RZ_OK) return rc;
        if (Y.keep == nullptr && (rc = dev_alloc(e, &Y.keep, G)) != RZ_OK) return rc;
        if (Y.stepm == nullptr && (rc = dev_alloc(e, &Y.stepm, G)) != RZ_OK) return rc;
        if (Y.step_ab == nullptr && (rc = dev_alloc(e, &Y.step_ab, 2)) != RZ_OK) return rc;
        if (Y.top_hwm == nullptr && (rc = dev_alloc(e, &Y.top_hwm, G)) != RZ_OK) return rc;
    }
    Y.queue_ids = cfg->d_queue_ids;
    Y.queue_ctl = cfg->d_queue_ctl;
    Y.log = cfg->d_log;
    {   // a log in HOST memory (pinned: the kernels write it directly) must be addressable from THIS device: asked, not assumed
        hipPointerAttribute_t attr;
        memset(&attr, 0, sizeof(attr));
        if (hipPointerGetAttributes(&attr, cfg->d_log) != hipSuccess) {
            (void)hipGetLastError();   // (an address the runtime does not know: treated as device memory, as before)
        } else if (attr.type == hipMemoryTypeHost) {
            void *dp = nullptr;
            if (hipHostGetDevicePointer(&dp, cfg->d_log, 0) != hipSuccess || dp == nullptr) {
                (void)hipGetLastError();
                return fail(RZ_ERR_ARG, "rz_play_attach: d_log is host memory that device %d cannot address (pass device memory)", e->cfg.device);
            }
            Y.log = static_cast<int32_t *>(dp);
        }
    }
    Y.ring = cfg->ring_steps;
    Y.words = RZ_PLAY_RECORD_WORDS + e->dev.A;
    Y.seed = cfg->seed;
    Y.inv_t = 1.0 / cfg->temperature;
    Y.margin = cfg->stall_margin > 0.0 ? cfg->stall_margin : 1e-10 * (Y.inv_t > 1.0 ? Y.inv_t : 1.0);
    RZ_HIP(hipMemset(Y.step_ab, 0, 8));
    RZ_HIP(hipMemset(Y.ply, 0, (size_t)G * 4));
    RZ_HIP(hipMemset(Y.top_hwm, 0, (size_t)G * 4));
    k_play_stop<<<flat_grid(e), dim3(256), 0, 0>>>(e->dev, Y);
    RZ_HIP(hipDeviceSynchronize());
    e->play_on = true;
    e->play_drawn = false;
    e->play_steps = 0;
    return RZ_OK;
}

int rz_play_draw(rz_engine *e, void *stream) {
    RZ_ENTER(e);
    if (!e->play_on) return fail(RZ_ERR_ARG, "call rz_play_attach first");
    k_play_draw<<<per_game(e), dim3(kWave), 0, as_stream(stream)>>>(e->dev, e->play);
    e->play_drawn = true;
    return launched("k_play_draw");
}

int rz_play_apply(rz_engine *e, void *stream) {
    RZ_ENTER(e);
    if (!e->play_on) return fail(RZ_ERR_ARG, "call rz_play_attach first");
    const Play &Y = e->play;
    if (!e->play_drawn) k_play_no_draw<<<dim3(1), dim3(1), 0, as_stream(stream)>>>(Y);   // (the step counter k_play_draw would have handed on)
    k_play_apply<<<per_game(e), dim3(kWave), 0, as_stream(stream)>>>(e->dev, Y, e->play_drawn ? 1 : 0);
    e->play_drawn = false;
    e->play_steps += 1;
    return launched("k_play_apply");
}

int rz_play_resolve(rz_engine *e, int32_t slot, int32_t move, void *stream) {
    RZ_ENTER(e);
    if (!e->play_on) return fail(RZ_ERR_ARG, "call rz_play_attach first");
    if (slot < 0 || slot >= e->cfg.n_games || move < 0 || move >= e->dev.A) return fail(RZ_ERR_ARG, "rz_play_resolve: slot %d / move %d out of range", slot, move);
    k_play_resolve<<<dim3(1), dim3(1), 0, as_stream(stream)>>>(e->play, slot, move);
    return launched("k_play_resolve");
}

int rz_play_stop(rz_engine *e, void *stream) {
    RZ_ENTER(e);
    if (!e->play_on) return fail(RZ_ERR_ARG, "call rz_play_attach first");
    k_play_stop<<<flat_grid(e), dim3(256), 0, as_stream(stream)>>>(e->dev, e->play);
    return launched("k_play_stop");
}

int rz_play_state(rz_engine *e, int64_t *h_game_id, int32_t *h_ply, int32_t *h_state, int64_t *h_steps) {
    RZ_ENTER(e);
    if (!e->play_on) return fail(RZ_ERR_ARG, "call rz_play_attach first");
    RZ_HIP(hipDeviceSynchronize());
    const size_t G = (size_t)e->cfg.n_games;
    if (h_game_id) RZ_HIP(hipMemcpy(h_game_id, e->play.game_id, G * 8, hipMemcpyDeviceToHost));
    if (h_ply) RZ_HIP(hipMemcpy(h_ply, e->play.ply, G * 4, hipMemcpyDeviceToHost));
    if (h_state) RZ_HIP(hipMemcpy(h_state, e->play.state, G * 4, hipMemcpyDeviceToHost));
    if (h_steps) {
        int32_t ab[2] = {0, 0};
        RZ_HIP(hipMemcpy(ab, e->play.step_ab, 8, hipMemcpyDeviceToHost));
        *h_steps = ab[0];
    }
    return RZ_OK;
}

int rz_get_stats(rz_engine *e, rz_stats *out) {
    RZ_ENTER(e);
    RZ_NEED(out);
    RZ_HIP(hipDeviceSynchronize());
    const size_t G = (size_t)e->cfg.n_games;
    std::vector<int32_t> err(G), top(G), nblk(G);
    int32_t any = 0, drops = 0;
    RZ_HIP(hipMemcpy(&any, e->dev.err_any, 4, hipMemcpyDeviceToHost));
    RZ_HIP(hipMemcpy(&drops, e->dev.reuse_drops, 4, hipMemcpyDeviceToHost));
    RZ_HIP(hipMemcpy(err.data(), e->dev.err, G * 4, hipMemcpyDeviceToHost));
    RZ_HIP(hipMemcpy(top.data(), e->dev.top, G * 4, hipMemcpyDeviceToHost));
    RZ_HIP(hipMemcpy(nblk.data(), e->dev.nblk, G * 4, hipMemcpyDeviceToHost));
    if (e->play_on) {   // device-driven moves: the arenas' tops at the END of the searches (now they hold the kept subtrees)
        std::vector<int32_t> hwm(G);
        RZ_HIP(hipMemcpy(hwm.data(), e->play.top_hwm, G * 4, hipMemcpyDeviceToHost));
        for (size_t g = 0; g < G; ++g)
            if (hwm[g] > top[g]) top[g] = hwm[g];
    }
    memset(out, 0, sizeof(*out));
    out->error_flags = any;
    out->first_bad_game = -1;
    for (size_t g = 0; g < G; ++g) {
        if ((err[g] & ~RZ_FLAG_REUSE_DROPPED) && out->first_bad_game < 0) out->first_bad_game = (int32_t)g;
        if (top[g] > out->max_slots_used) out->max_slots_used = top[g];
        if (nblk[g] > out->max_blocks_used) out->max_blocks_used = nblk[g];
    }
    out->arena_slots = e->dev.cap;
    out->prior_floats = e->dev.pcap;
    out->device_bytes = e->bytes;
    out->n_select_calls = e->n_select;
    out->reuse_dropped = drops;
    return RZ_OK;
}

int rz_poll_errors(rz_engine *e, int32_t *h_flags, int32_t *h_reuse_dropped, void *stream) {
    RZ_ENTER(e);
    RZ_NEED(h_flags);
    RZ_HIP(hipStreamSynchronize(as_stream(stream)));
    RZ_HIP(hipMemcpy(h_flags, e->dev.err_any, 4, hipMemcpyDeviceToHost));
    if (h_reuse_dropped) RZ_HIP(hipMemcpy(h_reuse_dropped, e->dev.reuse_drops, 4, hipMemcpyDeviceToHost));
    return RZ_OK;
}

int rz_clear_errors(rz_engine *e) {
    RZ_ENTER(e);
    RZ_HIP(hipDeviceSynchronize());
    RZ_HIP(hipMemset(e->dev.err, 0, (size_t)e->cfg.n_games * 4));
    RZ_HIP(hipMemset(e->dev.err_any, 0, 4));
    RZ_HIP(hipMemset(e->dev.reuse_drops, 0, 4));
    return RZ_OK;
}

int rz_copy_arena(rz_engine *e, int32_t game, int64_t max_slots, int32_t *h_n, double *h_w,
                  int32_t *h_first_child, int32_t *h_n_visited, int32_t *h_n_children,
                  int32_t *h_prior_block, float *h_root_prior, int32_t *h_top) {
    RZ_ENTER(e);
    RZ_NEED(h_top);
    if (game < 0 || game >= e->cfg.n_games) return fail(RZ_ERR_ARG, "game %d out of range", game);
    RZ_HIP(hipDeviceSynchronize());
    int32_t arena = 0, top = 0;
    RZ_HIP(hipMemcpy(&arena, e->dev.cur_arena + game, 4, hipMemcpyDeviceToHost));
    RZ_HIP(hipMemcpy(&top, e->dev.top + game, 4, hipMemcpyDeviceToHost));
    *h_top = top;
    long long n = top < max_slots ? top : max_slots;
    if (n <= 0) return RZ_OK;
    const long long base = ((long long)game * 2 + arena) * e->dev.cap;
    std::vector<int4> rec((size_t)n * 2);
    RZ_HIP(hipMemcpy(rec.data(), e->dev.R + 2 * base, (size_t)n * 2 * sizeof(int4), hipMemcpyDeviceToHost));
    for (long long i = 0; i < n; ++i) {
        const int4 lo = rec[(size_t)(2 * i)], hi = rec[(size_t)(2 * i + 1)];
        if (h_n) h_n[i] = lo.x;
        if (h_first_child) h_first_child[i] = lo.y;
        if (h_n_visited) h_n_visited[i] = lo.z;
        if (h_n_children) h_n_children[i] = lo.w & 0xffff;
        if (h_prior_block) h_prior_block[i] = hi.z;
        if (h_w) memcpy(&h_w[i], &hi.x, 8);
    }
    if (h_root_prior) memcpy(h_root_prior, &rec[1].w, 4);
    return RZ_OK;
}

int rz_copy_priors(rz_engine *e, int32_t game, int64_t max_floats, float *h_priors, int32_t *h_count) {
    RZ_ENTER(e);
    RZ_NEED(h_count);
    if (game < 0 || game >= e->cfg.n_games) return fail(RZ_ERR_ARG, "game %d out of range", game);
    RZ_HIP(hipDeviceSynchronize());
    int32_t arena = 0, ptop = 0;
    RZ_HIP(hipMemcpy(&arena, e->dev.cur_arena + game, 4, hipMemcpyDeviceToHost));
    RZ_HIP(hipMemcpy(&ptop, e->dev.ptop + game, 4, hipMemcpyDeviceToHost));
    *h_count = ptop;
    const long long n = ptop < max_floats ? ptop : max_floats;
    if (n <= 0 || h_priors == nullptr) return RZ_OK;
    RZ_HIP(hipMemcpy(h_priors, e->dev.P + ((long long)game * 2 + arena) * e->dev.pcap, (size_t)n * 4,
                     hipMemcpyDeviceToHost));
    return RZ_OK;
}

int rz_uct_scores(rz_engine *e, const double *d_w, const int32_t *d_n, const int32_t *d_np,
                  double c_puct, double *d_out, int64_t count, void *stream) {
    RZ_ENTER(e);
    RZ_NEED(d_w); RZ_NEED(d_n); RZ_NEED(d_np); RZ_NEED(d_out);
    if (count <= 0) return RZ_OK;
    const unsigned blocks = (unsigned)((count + 255) / 256);
    k_uct_scores<<<dim3(blocks), dim3(256), 0, as_stream(stream)>>>(d_w, d_n, d_np, c_puct, e->d_logtab,
                                                                    e->dev.logtab_n, d_out, count);
    return launched("k_uct_scores");
}

}  // extern "C"
