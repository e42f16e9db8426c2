// rz_muzero.hip -- MI355X (gfx950) MuZero search tree: latent-state MCTS with pUCT selection and
// min-max value normalisation, for many independent games (environments) at once.
//
// The reference only names MuZero (README.md:3, rlzero/algorithms/rl_args.py:21-24): there is no
// implementation to follow, so the algorithm is the published one -- Schrittwieser et al., "Mastering
// Atari, Go, Chess and Shogi by Planning with a Learned Model" (arXiv:1911.08265v2), appendix
// pseudocode: run_mcts / select_child / ucb_score / expand_node / backpropagate / MinMaxStats --
// which the parity tests restate in CPython and compare against, statistic for statistic.
// Single-player form (to_play is constant, CartPole): no sign flip in the backup.
//
// One THREAD per game: a MuZero tree is tiny (n_sims + 1 expanded nodes, A children each) and its
// walk is a short chain of dependent loads, so the parallelism is across the thousands of games.
// Tree layout (struct of arrays, per game g, node slot i, slot index g * cap + i):
//   one 32-byte record per node: N int32, first_child int32 (-1 = not expanded; the A children of a node are the
//   consecutive slots first_child .. first_child + A - 1), value_sum f64, prior f64, reward f32.
// The learned model stays outside: rz_mz_select reports (parent slot, action, leaf slot) per game;
// the caller gathers the parents' hidden states, runs dynamics + prediction on the batch, stores
// the new hidden states at the leaf slots and hands reward / policy / value to rz_mz_expand_backup.
//
// Arithmetic: fp64, one rounding per operation (-ffp-contract=off), IEEE divide and sqrt; the
// log((N + c2 + 1) / c2) factor of pUCT comes from a table filled by the HOST libm (what CPython's
// math.log calls), so a tree is bit-identical to the CPython restatement.

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <new>
#include <vector>

#include "rlzero_hip.h"

#pragma clang fp contract(off)

void rz_set_error(const char *msg);  // rz_engine.hip

namespace {

// A tree node is ONE 32-byte record (one sector: a level of the walk costs one load for the node and one 64-byte
// transaction for its two children instead of a scattered load per field)
struct __attribute__((aligned(16))) MzNode {
    int32_t N, first_child;   // visit count; slot of the first child (-1 = not expanded)
    double value_sum, prior;
    float reward;             // (the network's float, widened when used)
    int32_t pad;
};
static_assert(sizeof(MzNode) == 32, "MzNode layout");

struct MzDev {
    int n_games, n_actions, n_sims, cap, path_stride;
    double discount, pb_c_init;
    MzNode *nodes;
    int32_t *top, *path, *depth, *err;
    double *vmin, *vmax;
    const double *pb_log;  // [n_sims + 2]: log((n + pb_c_base + 1) / pb_c_base)
};

// Node.value(): value_sum / visit_count, 0 for an unvisited node
__device__ __forceinline__ double node_value(const MzNode &nd) {
    return nd.N > 0 ? nd.value_sum / (double)nd.N : 0.0;
}

// MinMaxStats.normalize
__device__ __forceinline__ double normalize(double v, double lo, double hi) {
    return hi > lo ? (v - lo) / (hi - lo) : v;
}

// expand_node for the roots (from the initial inference), optional Dirichlet mix
// (add_exploration_noise: prior * (1 - frac) + noise * frac), fresh MinMaxStats.
__global__ void k_mz_init(MzDev E, const float *probs, const double *noise, double frac, const uint8_t *mask) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.n_games) return;
    if (mask != nullptr && !mask[g]) return;
    const long long base = (long long)g * E.cap;
    E.nodes[base] = MzNode{0, 1, 0.0, 0.0, 0.0f, 0};
    for (int a = 0; a < E.n_actions; ++a) {
        double p = (double)probs[(long long)g * E.n_actions + a];
        if (noise != nullptr) p = p * (1.0 - frac) + noise[(long long)g * E.n_actions + a] * frac;
        E.nodes[base + 1 + a] = MzNode{0, -1, 0.0, p, 0.0f, 0};
    }
    E.top[g] = 1 + E.n_actions;
    E.vmin[g] = INFINITY;   // MinMaxStats(): minimum = +MAX, maximum = -MAX
    E.vmax[g] = -INFINITY;
    E.depth[g] = 0;
}

// select_child down to the first unexpanded node; max() over (score, action): ties go to the LARGER action
__device__ __forceinline__ void mz_select_one(const MzDev &E, int g, int &par_out, int &act_out, int &leaf_out) {
    const long long base = (long long)g * E.cap;
    int32_t *path = E.path + (long long)g * E.path_stride;
    const double lo = E.vmin[g], hi = E.vmax[g];
    int node = 0, depth = 0, last_action = 0, par = 0;
    path[0] = 0;
    MzNode cur = E.nodes[base];
    while (cur.first_child >= 0 && depth + 1 < E.path_stride) {
        const int fc = cur.first_child;
        const int pn = cur.N;
        const double pb_c0 = E.pb_log[pn <= E.n_sims + 1 ? pn : E.n_sims + 1] + E.pb_c_init;
        const double sq = sqrt((double)pn);
        double best = -INFINITY;
        int besta = 0;
        MzNode bestn = cur;
        for (int a = 0; a < E.n_actions; ++a) {
            const MzNode ch = E.nodes[base + fc + a];
            const int cn = ch.N;
            const double pb_c = pb_c0 * (sq / (double)(cn + 1));
            const double prior_score = pb_c * ch.prior;
            double value_score = 0.0;
            if (cn > 0) value_score = normalize((double)ch.reward + E.discount * node_value(ch), lo, hi);
            const double score = prior_score + value_score;
            if (score >= best) {  // the later (larger) action wins a tie
                best = score;
                besta = a;
                bestn = ch;
            }
        }
        par = node;
        last_action = besta;
        node = fc + besta;
        cur = bestn;
        depth += 1;
        path[depth] = node;
    }
    E.depth[g] = depth;
    par_out = par;
    act_out = last_action;
    leaf_out = node;
}

__global__ void k_mz_select(MzDev E, int32_t *parent, int32_t *action, int32_t *leaf, const uint8_t *mask) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.n_games) return;
    int par = 0, act = 0, lf = 0;
    if (mask == nullptr || mask[g]) mz_select_one(E, g, par, act, lf);
    parent[g] = par;
    action[g] = act;
    leaf[g] = lf;
}

// expand_node(leaf, network_output) + backpropagate(search_path, value, discount, min_max_stats); `probs` = the
// n_actions probabilities of this game
__device__ __forceinline__ void mz_expand_backup_one(const MzDev &E, int g, float reward_g, const float *probs, float value_g) {
    const long long base = (long long)g * E.cap;
    const int32_t *path = E.path + (long long)g * E.path_stride;
    const int depth = E.depth[g];
    const int leaf = path[depth];
    const int top = E.top[g];
    if (top + E.n_actions > E.cap) {
        atomicOr(E.err, RZ_FLAG_ARENA_FULL);
    } else {
        E.nodes[base + leaf].reward = reward_g;
        E.nodes[base + leaf].first_child = top;
        for (int a = 0; a < E.n_actions; ++a) E.nodes[base + top + a] = MzNode{0, -1, 0.0, (double)probs[a], 0.0f, 0};
        E.top[g] = top + E.n_actions;
    }
    double v = (double)value_g;
    double lo = E.vmin[g], hi = E.vmax[g];
    for (int d = depth; d >= 0; --d) {
        MzNode *nd = E.nodes + base + path[d];
        const double sum = nd->value_sum + v;
        const int n = nd->N + 1;
        nd->value_sum = sum;
        nd->N = n;
        const double nv = sum / (double)n;
        hi = nv > hi ? nv : hi;  // MinMaxStats.update
        lo = nv < lo ? nv : lo;
        v = (double)nd->reward + E.discount * v;
    }
    E.vmin[g] = lo;
    E.vmax[g] = hi;
}

__global__ void k_mz_expand_backup(MzDev E, const float *reward, const float *probs, const float *value,
                                   const uint8_t *mask) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.n_games) return;
    if (mask != nullptr && !mask[g]) return;
    mz_expand_backup_one(E, g, reward[g], probs + (long long)g * E.n_actions, value[g]);
}

// ------------------------------------------------------------------ the whole search in ONE launch
// k_mz_search: every simulation of every game's search -- select, gather of the parent's hidden state, recurrent
// inference (dynamics + reward head + prediction of rlzero_amd/muzero/network.py, hidden size 64), scatter of the new
// hidden state, expand + backup -- inside one kernel.  Games are independent, so a workgroup keeps 64 games and the
// model's weights (65 KB, k-major) in LDS for all n_sims simulations: no launch, no global synchronisation and no
// weight traffic between simulations (the step-by-step route replays a hipGraph of ~15 small launches per simulation).
//   * 8 waves; lane = game, wave w owns output units 8w .. 8w+7 of every 64-wide layer: a thread accumulates its 8
//     units over k with the activation of ITS game (one ds_read per k, conflict-free [k][65] layout) and 8 weights
//     broadcast from LDS (two ds_read_b128 per k): plain fp32 FMAs at the vector rate, which on CDNA equals the
//     f32-input MFMA rate (and needs no fragment shuffles for a 64 x 64 layer);
//   * dyn2 and rew1 read the same activations: one pass over k feeds both; the per-sample min-max scaling of the new
//     state and the scalar heads (reward, policy logits, value) reduce over the 8 waves through LDS in wave order;
//   * wave 0 walks and updates the 64 trees (one thread per game, the code of k_mz_select / k_mz_expand_backup), in
//     fp64 with the host's log table: the tree arithmetic is the step-by-step route's, bit for bit, given the same
//     network outputs (the outputs themselves differ from rocBLAS' in the last bits: another summation order).
constexpr int kMzH = 64, kMzLd = 65, kMzWaves = 8, kMzMaxA = 8;

struct MzModel {        // device pointers, weights k-major: w[k][unit]
    const float *dyn1_w, *dyn1_b, *dyn2_w, *dyn2_b, *rew1_w, *rew1_b, *rew2_w, *rew2_b, *pre1_w, *pre1_b, *pol_w, *pol_b,
        *val_w, *val_b;
};

struct MzTrace {        // optional per-simulation outputs for the parity tests: [n_sims][n_games] (probs: x n_actions)
    int32_t *parent, *action, *leaf;
    float *reward, *probs, *value;
};

__host__ __device__ inline int mz_search_lds_floats(int A) {
    return (kMzH + A) * kMzH + 3 * kMzH * kMzH + kMzH * (6 + A) + 8 + A +   // weights and biases
           (kMzH + A) * kMzLd + 2 * kMzH * kMzLd +                           // x, h1, next state
           kMzWaves * 64 * (4 + A) +                                         // reductions: min, max, reward, value, A logits
           64 * (5 + A);                                                     // per game: parent, action, leaf, reward, value, probs
}

__global__ __launch_bounds__(64 * kMzWaves) void k_mz_search(MzDev E, MzModel M, float *hidden, int n_sims, MzTrace T) {
    extern __shared__ __attribute__((aligned(16))) float mz_lds[];
    const int A = E.n_actions, KX = kMzH + A;
    const int tid = threadIdx.x, e = tid & 63, w = tid >> 6, j0 = 8 * w;
    const int g = blockIdx.x * 64 + e;
    const bool live = g < E.n_games;
    float *p = mz_lds;
    float *W1 = p; p += KX * kMzH;
    float *W2 = p; p += kMzH * kMzH;
    float *WR = p; p += kMzH * kMzH;
    float *WP = p; p += kMzH * kMzH;
    float *B1 = p; p += kMzH;
    float *B2 = p; p += kMzH;
    float *BR = p; p += kMzH;
    float *BP = p; p += kMzH;
    float *WR2 = p; p += kMzH;
    float *WPOL = p; p += A * kMzH;
    float *WVAL = p; p += kMzH;
    float *BS = p; p += 8 + A;          // [0] rew2 bias, [1] val bias, [2 .. 2 + A) pol biases
    float *X = p; p += KX * kMzLd;      // parent state + one-hot action, [k][game]
    float *H1 = p; p += kMzH * kMzLd;   // relu(dyn1)
    float *S2 = p; p += kMzH * kMzLd;   // next state (scaled)
    float *RED = p; p += kMzWaves * 64 * (4 + A);
    float *G = p;                        // per game scalars
    int *Gpar = reinterpret_cast<int *>(G), *Gact = Gpar + 64, *Gleaf = Gact + 64;
    // weights into LDS, once
    for (int i = tid; i < KX * kMzH; i += 64 * kMzWaves) W1[i] = M.dyn1_w[i];
    for (int i = tid; i < kMzH * kMzH; i += 64 * kMzWaves) {
        W2[i] = M.dyn2_w[i];
        WR[i] = M.rew1_w[i];
        WP[i] = M.pre1_w[i];
    }
    for (int i = tid; i < kMzH; i += 64 * kMzWaves) {
        B1[i] = M.dyn1_b[i];
        B2[i] = M.dyn2_b[i];
        BR[i] = M.rew1_b[i];
        BP[i] = M.pre1_b[i];
        WR2[i] = M.rew2_w[i];
        WVAL[i] = M.val_w[i];
    }
    for (int i = tid; i < A * kMzH; i += 64 * kMzWaves) WPOL[i] = M.pol_w[i];
    if (tid == 0) {
        BS[0] = M.rew2_b[0];
        BS[1] = M.val_b[0];
    }
    if (tid < A) BS[2 + tid] = M.pol_b[tid];
    const long long hbase = (long long)g * E.cap * kMzH;
    for (int sim = 0; sim < n_sims; ++sim) {
        // S0: wave 0 selects (one thread per game)
        if (w == 0) {
            int par = 0, act = 0, lf = 0;
            if (live) mz_select_one(E, g, par, act, lf);
            Gpar[e] = par;
            Gact[e] = act;
            Gleaf[e] = lf;
        }
        __syncthreads();
        // S1: gather the parents' hidden states (coalesced rows) into X[k][game]; one-hot action rows
        for (int idx = tid; idx < 64 * kMzH; idx += 64 * kMzWaves) {
            const int ee = idx >> 6, k = idx & 63;
            const int gg = blockIdx.x * 64 + ee;
            X[k * kMzLd + ee] = gg < E.n_games ? hidden[(long long)gg * E.cap * kMzH + (long long)Gpar[ee] * kMzH + k] : 0.0f;
        }
        for (int idx = tid; idx < 64 * A; idx += 64 * kMzWaves) {
            const int a = idx >> 6, ee = idx & 63;
            X[(kMzH + a) * kMzLd + ee] = Gact[ee] == a ? 1.0f : 0.0f;
        }
        __syncthreads();
        // S2: h1 = relu(dyn1 [x, onehot(a)])
        {
            float acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = B1[j0 + i];
            for (int k = 0; k < KX; ++k) {
                const float x = X[k * kMzLd + e];
                const float4 wa = *reinterpret_cast<const float4 *>(W1 + k * kMzH + j0);
                const float4 wb = *reinterpret_cast<const float4 *>(W1 + k * kMzH + j0 + 4);
                acc[0] = fmaf(wa.x, x, acc[0]); acc[1] = fmaf(wa.y, x, acc[1]); acc[2] = fmaf(wa.z, x, acc[2]); acc[3] = fmaf(wa.w, x, acc[3]);
                acc[4] = fmaf(wb.x, x, acc[4]); acc[5] = fmaf(wb.y, x, acc[5]); acc[6] = fmaf(wb.z, x, acc[6]); acc[7] = fmaf(wb.w, x, acc[7]);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) H1[(j0 + i) * kMzLd + e] = fmaxf(acc[i], 0.0f);
        }
        __syncthreads();
        // S3: next state (dyn2, before scaling) and the reward head's hidden layer (rew1) from the same activations
        float t[8], rsum = 0.0f;
        {
            float ar[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                t[i] = B2[j0 + i];
                ar[i] = BR[j0 + i];
            }
            for (int k = 0; k < kMzH; ++k) {
                const float x = H1[k * kMzLd + e];
                const float4 wa = *reinterpret_cast<const float4 *>(W2 + k * kMzH + j0);
                const float4 wb = *reinterpret_cast<const float4 *>(W2 + k * kMzH + j0 + 4);
                const float4 ra = *reinterpret_cast<const float4 *>(WR + k * kMzH + j0);
                const float4 rb = *reinterpret_cast<const float4 *>(WR + k * kMzH + j0 + 4);
                t[0] = fmaf(wa.x, x, t[0]); t[1] = fmaf(wa.y, x, t[1]); t[2] = fmaf(wa.z, x, t[2]); t[3] = fmaf(wa.w, x, t[3]);
                t[4] = fmaf(wb.x, x, t[4]); t[5] = fmaf(wb.y, x, t[5]); t[6] = fmaf(wb.z, x, t[6]); t[7] = fmaf(wb.w, x, t[7]);
                ar[0] = fmaf(ra.x, x, ar[0]); ar[1] = fmaf(ra.y, x, ar[1]); ar[2] = fmaf(ra.z, x, ar[2]); ar[3] = fmaf(ra.w, x, ar[3]);
                ar[4] = fmaf(rb.x, x, ar[4]); ar[5] = fmaf(rb.y, x, ar[5]); ar[6] = fmaf(rb.z, x, ar[6]); ar[7] = fmaf(rb.w, x, ar[7]);
            }
            float lo = t[0], hi = t[0];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                lo = fminf(lo, t[i]);
                hi = fmaxf(hi, t[i]);
                rsum = fmaf(fmaxf(ar[i], 0.0f), WR2[j0 + i], rsum);
            }
            RED[(0 * kMzWaves + w) * 64 + e] = lo;
            RED[(1 * kMzWaves + w) * 64 + e] = hi;
            RED[(2 * kMzWaves + w) * 64 + e] = rsum;
        }
        __syncthreads();
        // S4: per-sample min-max scaling of the new state (network.py scale_hidden), stored for the leaf
        {
            float lo = RED[(0 * kMzWaves) * 64 + e], hi = RED[(1 * kMzWaves) * 64 + e];
#pragma unroll
            for (int q = 1; q < kMzWaves; ++q) {
                lo = fminf(lo, RED[(0 * kMzWaves + q) * 64 + e]);
                hi = fmaxf(hi, RED[(1 * kMzWaves + q) * 64 + e]);
            }
            const float inv = fmaxf(hi - lo, 1e-5f);
            float *dst = live ? hidden + hbase + (long long)Gleaf[e] * kMzH + j0 : nullptr;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float sv = (t[i] - lo) / inv;
                S2[(j0 + i) * kMzLd + e] = sv;
                if (dst) dst[i] = sv;
            }
        }
        __syncthreads();
        // S5: prediction: p1 = relu(pre1 s'), partial dot products of the policy and value heads
        {
            float acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = BP[j0 + i];
            for (int k = 0; k < kMzH; ++k) {
                const float x = S2[k * kMzLd + e];
                const float4 wa = *reinterpret_cast<const float4 *>(WP + k * kMzH + j0);
                const float4 wb = *reinterpret_cast<const float4 *>(WP + k * kMzH + j0 + 4);
                acc[0] = fmaf(wa.x, x, acc[0]); acc[1] = fmaf(wa.y, x, acc[1]); acc[2] = fmaf(wa.z, x, acc[2]); acc[3] = fmaf(wa.w, x, acc[3]);
                acc[4] = fmaf(wb.x, x, acc[4]); acc[5] = fmaf(wb.y, x, acc[5]); acc[6] = fmaf(wb.z, x, acc[6]); acc[7] = fmaf(wb.w, x, acc[7]);
            }
            float vsum = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = fmaxf(acc[i], 0.0f);
                vsum = fmaf(acc[i], WVAL[j0 + i], vsum);
            }
            RED[(3 * kMzWaves + w) * 64 + e] = vsum;
            for (int a = 0; a < A; ++a) {
                float ps = 0.0f;
#pragma unroll
                for (int i = 0; i < 8; ++i) ps = fmaf(acc[i], WPOL[a * kMzH + j0 + i], ps);
                RED[((4 + a) * kMzWaves + w) * 64 + e] = ps;
            }
        }
        __syncthreads();
        // S6: wave 0 finishes the heads (sums over the 8 waves in wave order, softmax) and expands + backs up
        if (w == 0) {
            float reward = BS[0], value = BS[1], logit[kMzMaxA], probs[kMzMaxA];
#pragma unroll
            for (int q = 0; q < kMzWaves; ++q) {
                reward += RED[(2 * kMzWaves + q) * 64 + e];
                value += RED[(3 * kMzWaves + q) * 64 + e];
            }
            float mx = -INFINITY;
#pragma unroll
            for (int a = 0; a < kMzMaxA; ++a) {
                logit[a] = -INFINITY;
                if (a < A) {
                    float l = BS[2 + a];
#pragma unroll
                    for (int q = 0; q < kMzWaves; ++q) l += RED[((4 + a) * kMzWaves + q) * 64 + e];
                    logit[a] = l;
                    mx = fmaxf(mx, l);
                }
            }
            float den = 0.0f;
#pragma unroll
            for (int a = 0; a < kMzMaxA; ++a) {
                probs[a] = a < A ? expf(logit[a] - mx) : 0.0f;
                den += probs[a];
            }
#pragma unroll
            for (int a = 0; a < kMzMaxA; ++a) probs[a] = probs[a] / den;
            if (live) {
                if (T.reward != nullptr) {
                    const long long o = (long long)sim * E.n_games + g;
                    T.parent[o] = Gpar[e];
                    T.action[o] = Gact[e];
                    T.leaf[o] = Gleaf[e];
                    T.reward[o] = reward;
                    T.value[o] = value;
#pragma unroll
                    for (int a = 0; a < kMzMaxA; ++a)
                        if (a < A) T.probs[o * A + a] = probs[a];
                }
                mz_expand_backup_one(E, g, reward, probs, value);
            }
        }
        // (no barrier: the next select is wave 0's too, and every LDS buffer wave 0 reads above is rewritten only
        // after the barrier that follows that select)
    }
}

// what: 0 = visit counts (int32 [G][A]), 1 = value sums (f64 [G][A]), 2 = rewards, 3 = priors of the root's children
__global__ void k_mz_root_children(MzDev E, int what, void *out) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.n_games) return;
    const long long base = (long long)g * E.cap;
    const int fc = E.nodes[base].first_child;
    for (int a = 0; a < E.n_actions; ++a) {
        const long long o = (long long)g * E.n_actions + a;
        const MzNode c = fc >= 0 ? E.nodes[base + fc + a] : MzNode{0, -1, 0.0, 0.0, 0.0f, 0};
        if (what == 0) ((int32_t *)out)[o] = c.N;
        else if (what == 1) ((double *)out)[o] = c.value_sum;
        else if (what == 2) ((double *)out)[o] = (double)c.reward;
        else ((double *)out)[o] = c.prior;
    }
}

__global__ void k_mz_root_stats(MzDev E, int32_t *n, double *value_sum, double *vmin, double *vmax) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.n_games) return;
    const long long base = (long long)g * E.cap;
    if (n) n[g] = E.nodes[base].N;
    if (value_sum) value_sum[g] = E.nodes[base].value_sum;
    if (vmin) vmin[g] = E.vmin[g];
    if (vmax) vmax[g] = E.vmax[g];
}

int mz_fail(int code, const char *msg) {
    rz_set_error(msg);
    return code;
}

}  // namespace

struct rz_muzero {
    rz_mz_config cfg;
    MzDev dev;
    std::vector<void *> allocs;
    long long bytes = 0;
    MzModel model = {};          // rz_mz_load_model
    float *d_model = nullptr;    // one allocation behind the pointers above
    bool model_loaded = false;
};

namespace {

template <typename T>
int mz_alloc(rz_muzero *e, T **out, long long count) {
    void *p = nullptr;
    if (hipMalloc(&p, (size_t)count * sizeof(T)) != hipSuccess) return mz_fail(RZ_ERR_OOM, "hipMalloc failed (muzero tree)");
    e->allocs.push_back(p);
    e->bytes += count * (long long)sizeof(T);
    *out = (T *)p;
    return RZ_OK;
}

inline dim3 mz_grid(const rz_muzero *e) { return dim3((unsigned)((e->cfg.n_games + 127) / 128)); }

int mz_ready(rz_muzero *e) {
    if (e == nullptr) return mz_fail(RZ_ERR_ARG, "muzero handle is NULL");
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess) return mz_fail(RZ_ERR_HIP, "hipGetDevice failed");
    if (cur != e->cfg.device && hipSetDevice(e->cfg.device) != hipSuccess) return mz_fail(RZ_ERR_HIP, "hipSetDevice failed");
    return RZ_OK;
}

int mz_launched(const char *what) {
    if (hipGetLastError() != hipSuccess) return mz_fail(RZ_ERR_HIP, what);
    return RZ_OK;
}

}  // namespace

extern "C" {

int rz_mz_create(const rz_mz_config *cfg, rz_muzero **out) {
    if (cfg == nullptr || out == nullptr) return mz_fail(RZ_ERR_ARG, "NULL argument");
    *out = nullptr;
    if (cfg->abi_version != RZ_ABI_VERSION) return mz_fail(RZ_ERR_ARG, "rz_mz_config.abi_version does not match the library");
    if (cfg->n_games < 1 || cfg->n_actions < 1 || cfg->n_actions > 64 || cfg->n_sims < 1)
        return mz_fail(RZ_ERR_ARG, "n_games / n_actions (1..64) / n_sims out of range");
    if (!(cfg->discount > 0.0) || !(cfg->pb_c_base > 0.0)) return mz_fail(RZ_ERR_ARG, "discount and pb_c_base must be > 0");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || cfg->device < 0 || cfg->device >= n_dev)
        return mz_fail(RZ_ERR_ARG, "bad device ordinal");
    if (hipSetDevice(cfg->device) != hipSuccess) return mz_fail(RZ_ERR_HIP, "hipSetDevice failed");
    rz_muzero *e = new (std::nothrow) rz_muzero();
    if (!e) return mz_fail(RZ_ERR_OOM, "host allocation failed");
    e->cfg = *cfg;
    MzDev &D = e->dev;
    D.n_games = cfg->n_games;
    D.n_actions = cfg->n_actions;
    D.n_sims = cfg->n_sims;
    D.cap = 1 + cfg->n_actions * (cfg->n_sims + 1);  // root + one block of children per expansion
    D.path_stride = cfg->n_sims + 2;
    D.discount = cfg->discount;
    D.pb_c_init = cfg->pb_c_init;
    const long long G = cfg->n_games, slots = G * D.cap;
    int rc = RZ_OK;
    double *d_log = nullptr;
#define MZ_ALLOC(field, count) if (rc == RZ_OK) rc = mz_alloc(e, &D.field, (count))
    MZ_ALLOC(nodes, slots);
    MZ_ALLOC(top, G);
    MZ_ALLOC(depth, G);
    MZ_ALLOC(path, G * D.path_stride);
    MZ_ALLOC(vmin, G);
    MZ_ALLOC(vmax, G);
    MZ_ALLOC(err, 1);
#undef MZ_ALLOC
    if (rc == RZ_OK) rc = mz_alloc(e, &d_log, cfg->n_sims + 2);
    if (rc == RZ_OK) {
        std::vector<double> tab((size_t)cfg->n_sims + 2);
        for (int n = 0; n < cfg->n_sims + 2; ++n) tab[(size_t)n] = std::log(((double)n + cfg->pb_c_base + 1.0) / cfg->pb_c_base);
        if (hipMemcpy(d_log, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemset(D.err, 0, 4) != hipSuccess || hipMemset(D.nodes, 0xff, (size_t)slots * sizeof(MzNode)) != hipSuccess ||
            hipMemset(D.top, 0, (size_t)G * 4) != hipSuccess || hipMemset(D.depth, 0, (size_t)G * 4) != hipSuccess)
            rc = mz_fail(RZ_ERR_HIP, "initialisation of the muzero tree failed");
    }
    if (rc != RZ_OK) {
        rz_mz_destroy(e);
        return rc;
    }
    D.pb_log = d_log;
    *out = e;
    return RZ_OK;
}

int rz_mz_destroy(rz_muzero *e) {
    if (e == nullptr) return RZ_OK;
    (void)hipSetDevice(e->cfg.device);
    (void)hipDeviceSynchronize();
    for (void *p : e->allocs) (void)hipFree(p);
    delete e;
    return RZ_OK;
}

int rz_mz_upload_log_table(rz_muzero *e, const double *h_table, int64_t count) {
    int rc = mz_ready(e);
    if (rc != RZ_OK) return rc;
    if (h_table == nullptr || count != e->cfg.n_sims + 2) return mz_fail(RZ_ERR_ARG, "table must hold n_sims + 2 entries");
    if (hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(const_cast<double *>(e->dev.pb_log), h_table, (size_t)count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
        return mz_fail(RZ_ERR_HIP, "hipMemcpy(log table) failed");
    return RZ_OK;
}

int rz_mz_init_roots(rz_muzero *e, const float *d_probs, const double *d_noise, double noise_frac, const uint8_t *d_mask,
                     void *stream) {
    int rc = mz_ready(e);
    if (rc != RZ_OK) return rc;
    if (d_probs == nullptr) return mz_fail(RZ_ERR_ARG, "d_probs is NULL");
    k_mz_init<<<mz_grid(e), dim3(128), 0, (hipStream_t)stream>>>(e->dev, d_probs, d_noise, noise_frac, d_mask);
    return mz_launched("launch of k_mz_init failed");
}

int rz_mz_select(rz_muzero *e, int32_t *d_parent, int32_t *d_action, int32_t *d_leaf, const uint8_t *d_mask, void *stream) {
    int rc = mz_ready(e);
    if (rc != RZ_OK) return rc;
    if (!d_parent || !d_action || !d_leaf) return mz_fail(RZ_ERR_ARG, "NULL output pointer");
    k_mz_select<<<mz_grid(e), dim3(128), 0, (hipStream_t)stream>>>(e->dev, d_parent, d_action, d_leaf, d_mask);
    return mz_launched("launch of k_mz_select failed");
}

int rz_mz_expand_backup(rz_muzero *e, const float *d_reward, const float *d_probs, const float *d_value,
                        const uint8_t *d_mask, void *stream) {
    int rc = mz_ready(e);
    if (rc != RZ_OK) return rc;
    if (!d_reward || !d_probs || !d_value) return mz_fail(RZ_ERR_ARG, "NULL input pointer");
    k_mz_expand_backup<<<mz_grid(e), dim3(128), 0, (hipStream_t)stream>>>(e->dev, d_reward, d_probs, d_value, d_mask);
    return mz_launched("launch of k_mz_expand_backup failed");
}

int rz_mz_root_children(rz_muzero *e, int32_t what, void *d_out, void *stream) {
    int rc = mz_ready(e);
    if (rc != RZ_OK) return rc;
    if (d_out == nullptr || what < 0 || what > 3) return mz_fail(RZ_ERR_ARG, "bad argument");
    k_mz_root_children<<<mz_grid(e), dim3(128), 0, (hipStream_t)stream>>>(e->dev, what, d_out);
    return mz_launched("launch of k_mz_root_children failed");
}

int rz_mz_root_stats(rz_muzero *e, int32_t *d_n, double *d_value_sum, double *d_vmin, double *d_vmax, void *stream) {
    int rc = mz_ready(e);
    if (rc != RZ_OK) return rc;
    k_mz_root_stats<<<mz_grid(e), dim3(128), 0, (hipStream_t)stream>>>(e->dev, d_n, d_value_sum, d_vmin, d_vmax);
    return mz_launched("launch of k_mz_root_stats failed");
}

int rz_mz_load_model(rz_muzero *e, const float *const *h_params, int32_t n_params, int32_t hidden) {
    int rc = mz_ready(e);
    if (rc != RZ_OK) return rc;
    if (h_params == nullptr || n_params != 14) return mz_fail(RZ_ERR_ARG, "expected the 14 tensors of the dynamics / reward / prediction layers");
    if (hidden != kMzH) return mz_fail(RZ_ERR_ARG, "the fused search is built for hidden size 64");
    const int A = e->cfg.n_actions;
    if (A > kMzMaxA) return mz_fail(RZ_ERR_ARG, "the fused search handles up to 8 actions");
    for (int i = 0; i < 14; ++i)
        if (!h_params[i]) return mz_fail(RZ_ERR_ARG, "a parameter pointer is NULL");
    const int KX = kMzH + A;
    // order: dyn1.w [H][H+A], dyn1.b, dyn2.w [H][H], dyn2.b, rew1.w, rew1.b, rew2.w [1][H], rew2.b, pre1.w, pre1.b,
    //        pol.w [A][H], pol.b, val.w [1][H], val.b (torch layout [out][in]); big matrices go k-major [in][out]
    const size_t sizes[14] = {(size_t)KX * kMzH, kMzH, (size_t)kMzH * kMzH, kMzH, (size_t)kMzH * kMzH, kMzH, kMzH, 1,
                              (size_t)kMzH * kMzH, kMzH, (size_t)A * kMzH, (size_t)A, kMzH, 1};
    size_t total = 0, off[14];
    for (int i = 0; i < 14; ++i) {
        off[i] = total;
        total += (sizes[i] + 3) / 4 * 4;
    }
    std::vector<float> host(total, 0.0f);
    auto transpose = [&](int idx, int n_out, int n_in) {
        for (int o = 0; o < n_out; ++o)
            for (int k = 0; k < n_in; ++k) host[off[idx] + (size_t)k * n_out + o] = h_params[idx][(size_t)o * n_in + k];
    };
    transpose(0, kMzH, KX);
    transpose(2, kMzH, kMzH);
    transpose(4, kMzH, kMzH);
    transpose(8, kMzH, kMzH);
    for (int i : {1, 3, 5, 6, 7, 9, 10, 11, 12, 13})
        for (size_t q = 0; q < sizes[i]; ++q) host[off[i] + q] = h_params[i][q];
    if (hipDeviceSynchronize() != hipSuccess) return mz_fail(RZ_ERR_HIP, "hipDeviceSynchronize failed");
    if (e->d_model == nullptr) {
        if (hipMalloc((void **)&e->d_model, total * sizeof(float)) != hipSuccess) return mz_fail(RZ_ERR_OOM, "hipMalloc failed (muzero model)");
        e->allocs.push_back(e->d_model);
        e->bytes += (long long)(total * sizeof(float));
        if (hipFuncSetAttribute((const void *)k_mz_search, hipFuncAttributeMaxDynamicSharedMemorySize,
                                mz_search_lds_floats(A) * (int)sizeof(float)) != hipSuccess)
            return mz_fail(RZ_ERR_HIP, "hipFuncSetAttribute(dynamic LDS) failed");
    }
    if (hipMemcpy(e->d_model, host.data(), total * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        return mz_fail(RZ_ERR_HIP, "hipMemcpy(model) failed");
    const float *b = e->d_model;
    e->model = MzModel{b + off[0], b + off[1], b + off[2], b + off[3], b + off[4], b + off[5], b + off[6], b + off[7],
                       b + off[8], b + off[9], b + off[10], b + off[11], b + off[12], b + off[13]};
    e->model_loaded = true;
    return RZ_OK;
}

int rz_mz_search(rz_muzero *e, float *d_hidden, int32_t n_sims, int32_t *d_trace_parent, int32_t *d_trace_action,
                 int32_t *d_trace_leaf, float *d_trace_reward, float *d_trace_probs, float *d_trace_value, void *stream) {
    int rc = mz_ready(e);
    if (rc != RZ_OK) return rc;
    if (!e->model_loaded) return mz_fail(RZ_ERR_ARG, "rz_mz_load_model has not been called");
    if (d_hidden == nullptr || n_sims < 1 || n_sims > e->cfg.n_sims) return mz_fail(RZ_ERR_ARG, "d_hidden is NULL or n_sims out of range");
    const bool any = d_trace_parent || d_trace_action || d_trace_leaf || d_trace_reward || d_trace_probs || d_trace_value;
    const bool all = d_trace_parent && d_trace_action && d_trace_leaf && d_trace_reward && d_trace_probs && d_trace_value;
    if (any && !all) return mz_fail(RZ_ERR_ARG, "the trace arrays come all together or not at all");
    const MzTrace T = {d_trace_parent, d_trace_action, d_trace_leaf, d_trace_reward, d_trace_probs, d_trace_value};
    const int lds = mz_search_lds_floats(e->cfg.n_actions) * (int)sizeof(float);
    k_mz_search<<<dim3((unsigned)((e->cfg.n_games + 63) / 64)), dim3(64 * kMzWaves), lds, (hipStream_t)stream>>>(
        e->dev, e->model, d_hidden, n_sims, T);
    return mz_launched("launch of k_mz_search failed");
}

int rz_mz_geometry(rz_muzero *e, int32_t *slots_per_game, int64_t *device_bytes) {
    if (e == nullptr) return mz_fail(RZ_ERR_ARG, "muzero handle is NULL");
    if (slots_per_game) *slots_per_game = e->dev.cap;
    if (device_bytes) *device_bytes = e->bytes;
    return RZ_OK;
}

int rz_mz_error_flags(rz_muzero *e, int32_t *flags) {
    int rc = mz_ready(e);
    if (rc != RZ_OK) return rc;
    if (flags == nullptr) return mz_fail(RZ_ERR_ARG, "flags is NULL");
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(flags, e->dev.err, 4, hipMemcpyDeviceToHost) != hipSuccess)
        return mz_fail(RZ_ERR_HIP, "hipMemcpy(err) failed");
    return RZ_OK;
}

}  // extern "C"
