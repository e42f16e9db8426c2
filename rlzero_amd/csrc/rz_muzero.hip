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
// Three routes, one tree arithmetic (mz_descend / mz_grow_backup and their LDS twins give the same bits):
//   * step by step (rz_mz_select / rz_mz_expand_backup, one THREAD per game): the learned model stays outside -- the
//     caller gathers the parents' hidden states, runs dynamics + prediction on the batch, stores the new hidden states
//     at the leaf slots and hands reward / policy / value back;
//   * rz_mz_search: all simulations of a move in ONE launch, the model evaluated inside the kernel (k_mz_search);
//   * rz_mz_play_cartpole: whole MOVES in one launch -- initial inference, root noise, search, action draw, CartPole step,
//     episode history on the device (k_mz_search with its MOVES stages).
// A MuZero tree is tiny (n_sims + 1 expanded nodes, A children each) and its walk is a short chain of dependent steps,
// so the parallelism is across the thousands of games.  Tree layout in HBM (per game g, node slot i, slot index
// g * cap + i): one 32-byte record per node: N int32, first_child int32 (-1 = not expanded; the A children of a node are
// the consecutive slots first_child .. first_child + A - 1), value_sum f64, prior f64, reward f32.
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

// select_child down to the first unexpanded node; max() over (score, action): ties go to the LARGER action.
// ``nodes`` = slot 0 of this game's tree, ``path`` = its path buffer, ``pb_log`` = the host's log table (each in HBM
// for the step-by-step kernels, in LDS inside k_mz_search: the arithmetic is this one function).  -> depth of the leaf
template <typename NodeP, typename PathP, typename LogP>
__device__ __forceinline__ int mz_descend(const MzDev &E, NodeP nodes, PathP path, LogP pb_log, double lo, double hi,
                                          int &par_out, int &act_out, int &leaf_out) {
    int node = 0, depth = 0, last_action = 0, par = 0;
    path[0] = 0;
    MzNode cur = nodes[0];
    while (cur.first_child >= 0 && depth + 1 < E.path_stride) {
        const int fc = cur.first_child;
        const int pn = cur.N;
        const double pb_c0 = pb_log[pn <= E.n_sims + 1 ? pn : E.n_sims + 1] + E.pb_c_init;
        const double sq = sqrt((double)pn);
        double best = -INFINITY;
        int besta = 0;
        MzNode bestn = cur;
        for (int a = 0; a < E.n_actions; ++a) {
            const MzNode ch = nodes[fc + a];
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
    par_out = par;
    act_out = last_action;
    leaf_out = node;
    return depth;
}

__device__ __forceinline__ void mz_select_one(const MzDev &E, int g, int &par_out, int &act_out, int &leaf_out) {
    E.depth[g] = mz_descend(E, E.nodes + (long long)g * E.cap, E.path + (long long)g * E.path_stride, E.pb_log, E.vmin[g],
                            E.vmax[g], par_out, act_out, leaf_out);
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
// n_actions probabilities of this game (MAXA > 0: a register array of MAXA entries, indexed by an unrolled loop)
template <int MAXA, typename NodeP, typename PathP>
__device__ __forceinline__ void mz_grow_backup(const MzDev &E, NodeP nodes, PathP path, int depth, int &top, double &lo,
                                               double &hi, float reward_g, const float *probs, float value_g) {
    const int leaf = path[depth];
    if (top + E.n_actions > E.cap) {
        atomicOr(E.err, RZ_FLAG_ARENA_FULL);
    } else {
        nodes[leaf].reward = reward_g;
        nodes[leaf].first_child = top;
        if (MAXA > 0) {
#pragma unroll
            for (int a = 0; a < (MAXA > 0 ? MAXA : 1); ++a)
                if (a < E.n_actions) nodes[top + a] = MzNode{0, -1, 0.0, (double)probs[a], 0.0f, 0};
        } else {
            for (int a = 0; a < E.n_actions; ++a) nodes[top + a] = MzNode{0, -1, 0.0, (double)probs[a], 0.0f, 0};
        }
        top += E.n_actions;
    }
    double v = (double)value_g;
    for (int d = depth; d >= 0; --d) {
        const int slot = path[d];
        const double sum = nodes[slot].value_sum + v;
        const int n = nodes[slot].N + 1;
        nodes[slot].value_sum = sum;
        nodes[slot].N = n;
        const double nv = sum / (double)n;
        hi = nv > hi ? nv : hi;  // MinMaxStats.update
        lo = nv < lo ? nv : lo;
        v = (double)nodes[slot].reward + E.discount * v;
    }
}

__device__ __forceinline__ void mz_expand_backup_one(const MzDev &E, int g, float reward_g, const float *probs, float value_g) {
    int top = E.top[g];
    double lo = E.vmin[g], hi = E.vmax[g];
    mz_grow_backup<0>(E, E.nodes + (long long)g * E.cap, E.path + (long long)g * E.path_stride, E.depth[g], top, lo, hi,
                      reward_g, probs, value_g);
    E.top[g] = top;
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

// ------------------------------------------------------------------ CartPole-v1 on the device
// Gymnasium's CartPoleEnv (classic_control/cartpole.py v0.29: Euler integrator, tau 0.02 s, force 10 N, episode ends at
// |x| > 2.4 or |theta| > 12 degrees, reward 1 per step) with CartPole-v1's 500-step limit and auto-reset -- the device
// twin of rlzero_amd/muzero/cartpole.py: same constants, same order of operations (fp64, one rounding per operation),
// initial states from the same counter-based stream keyed (seed, environment, episode).
__device__ __forceinline__ unsigned long long mz_splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull;
    unsigned long long z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

struct MzCartPole {
    double x, x_dot, theta, theta_dot;
    long long steps, episode;
};

__device__ __forceinline__ void cartpole_reset(MzCartPole &c, unsigned long long seed, int env) {
    const unsigned long long key = mz_splitmix64(mz_splitmix64(seed ^ ((unsigned long long)env << 24)) ^ (unsigned long long)c.episode);
    double v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const double u = (double)(mz_splitmix64(key ^ (unsigned long long)(j + 1)) >> 11) / 9007199254740992.0;
        v[j] = -0.05 + 0.1 * u;
    }
    c.x = v[0];
    c.x_dot = v[1];
    c.theta = v[2];
    c.theta_dot = v[3];
    c.steps = 0;
}

// one step; -> bit 0 terminated, bit 1 truncated (the state is the terminal one: the caller resets)
__device__ __forceinline__ int cartpole_step(MzCartPole &c, int action) {
    const double gravity = 9.8, masspole = 0.1, total_mass = 0.1 + 1.0, length = 0.5, polemass_length = 0.1 * 0.5, tau = 0.02;
    const double theta_threshold = 12 * 2 * 3.141592653589793 / 360, x_threshold = 2.4;
    const double force = action == 1 ? 10.0 : -10.0;
    const double costheta = cos(c.theta), sintheta = sin(c.theta);
    const double temp = (force + polemass_length * (c.theta_dot * c.theta_dot) * sintheta) / total_mass;
    const double thetaacc = (gravity * sintheta - costheta * temp) / (length * (4.0 / 3.0 - masspole * (costheta * costheta) / total_mass));
    const double xacc = temp - polemass_length * thetaacc * costheta / total_mass;
    c.x = c.x + tau * c.x_dot;
    c.x_dot = c.x_dot + tau * xacc;
    c.theta = c.theta + tau * c.theta_dot;
    c.theta_dot = c.theta_dot + tau * thetaacc;
    c.steps += 1;
    const bool terminated = c.x < -x_threshold || c.x > x_threshold || c.theta < -theta_threshold || c.theta > theta_threshold;
    const bool truncated = c.steps >= 500;
    return (terminated ? 1 : 0) | (truncated ? 2 : 0);
}

__global__ void k_cartpole_step(double *state, long long *steps, long long *episode, const long long *actions, int n_envs,
                                unsigned long long seed, float *obs, float *reward, uint8_t *terminated, uint8_t *truncated) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_envs) return;
    MzCartPole c = {state[4 * g], state[4 * g + 1], state[4 * g + 2], state[4 * g + 3], steps[g], episode[g]};
    const int done = cartpole_step(c, (int)actions[g]);
    if (done) {
        c.episode += 1;
        cartpole_reset(c, seed, g);
    }
    state[4 * g] = c.x;
    state[4 * g + 1] = c.x_dot;
    state[4 * g + 2] = c.theta;
    state[4 * g + 3] = c.theta_dot;
    steps[g] = c.steps;
    episode[g] = c.episode;
    obs[4 * g] = (float)c.x;
    obs[4 * g + 1] = (float)c.x_dot;
    obs[4 * g + 2] = (float)c.theta;
    obs[4 * g + 3] = (float)c.theta_dot;
    reward[g] = 1.0f;
    terminated[g] = (uint8_t)(done & 1);
    truncated[g] = (uint8_t)((done >> 1) & 1);
}

// One Gamma(alpha, 1) sample for the root's Dirichlet noise (add_exploration_noise) from a counter-based stream:
// Marsaglia-Tsang (boosted by U^(1/alpha) below shape 1).  Noise, not parity: hardware transcendentals, 24-bit uniforms
// (the scheme of gamma03 in rz_engine.hip with the shape as an argument).
__device__ __forceinline__ uint32_t mz_hash32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

__device__ __forceinline__ float mz_gamma(float alpha, uint32_t key) {
    const float shape = alpha < 1.0f ? alpha + 1.0f : alpha;
    const float d = shape - 1.0f / 3.0f, c = 1.0f / sqrtf(9.0f * d);
    const float kLn2 = 0.69314718f, k2m24 = 1.0f / 16777216.0f;
    float g = d;
    for (int t = 0; t < 8; ++t) {
        const uint32_t h1 = mz_hash32(key + 3u * t), h2 = mz_hash32(key + 3u * t + 1u), h3 = mz_hash32(key + 3u * t + 2u);
        const float u1 = (float)((h1 >> 8) + 1u) * k2m24;   // (0, 1]
        const float u2 = (float)(h2 >> 8) * k2m24;           // [0, 1): a turn of the cosine
        const float u3 = (float)((h3 >> 8) + 1u) * k2m24;   // (0, 1]
        const float x = __builtin_amdgcn_sqrtf(-2.0f * kLn2 * __builtin_amdgcn_logf(u1)) * __builtin_amdgcn_cosf(u2);
        float v = 1.0f + c * x;
        if (v <= 0.0f) continue;
        v = v * v * v;
        if (kLn2 * __builtin_amdgcn_logf(u3) < 0.5f * x * x + d - d * v + d * kLn2 * __builtin_amdgcn_logf(v)) {
            g = d * v;
            break;
        }
    }
    if (alpha < 1.0f) {
        const float ub = (float)((mz_hash32(key + 0x5bd1e995u) >> 8) + 1u) * k2m24;
        g = g * __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(ub) * (1.0f / alpha));  // ub^(1/alpha)
    }
    return fmaxf(g, 1e-30f);
}

// ------------------------------------------------------------------ the whole search in ONE launch
// k_mz_search: every simulation of every game's search -- select, gather of the parent's hidden state, recurrent
// inference (dynamics + reward head + prediction of rlzero_amd/muzero/network.py, hidden size 64), scatter of the new
// hidden state, expand + backup -- inside one kernel: no launch, no global synchronisation and no weight traffic
// between simulations (the step-by-step route replays a hipGraph of ~15 small launches per simulation).
// A search is a chain of short dependent steps per game, so the kernel is built for LATENCY and for filling the chip
// with few games: a workgroup of 4 waves owns `gpw` <= 16 games (4096 games -> 256 workgroups, one per CU).
//   * the layers are 64 x K GEMMs over the workgroup's games on the matrix pipe, v_mfma_f32_16x16x4_f32 (exact f32
//     products, f32 accumulation): games = the 16 columns, wave w owns output units 16w .. 16w+15.  The WEIGHTS are
//     the A operands and never change: each wave keeps its fragments of all layers in registers for the whole launch;
//     activations go [k][game] through LDS (B operand: one conflict-free ds_read_b32 per MFMA);
//   * dyn2 and rew1 read the same activations: one pass over k feeds both; the per-sample min-max scaling of the new
//     state reduces over the rows with two cross-row steps inside a wave and over the 4 waves through LDS; the scalar
//     heads (reward; policy logits, value) are dot products of the rows a lane already holds: partial sums per wave,
//     summed in wave order by the lane that owns the game;
//   * wave 0 walks and updates the trees, one lane per game, with the code of k_mz_select / k_mz_expand_backup
//     (mz_descend / mz_grow_backup): fp64, the host's log table -- the tree arithmetic is the step-by-step route's,
//     bit for bit, given the same network outputs (the outputs differ from rocBLAS' in the last bits: another
//     summation order).  TREE_LDS: the trees (32 B x slots per game), the paths and the log table live in LDS for the
//     search (a level of the walk costs an LDS round trip instead of an L2 one) and are written back at the end;
//   * MOVES (rz_mz_play_cartpole): whole MOVES of CartPole environments in the launch -- per move the initial inference
//     (representation + prediction on the same tiles), root expansion with Dirichlet noise, the n_sims simulations,
//     the action drawn from the visit counts, one packed record for the host and the environment step: the host's
//     part of self-play shrinks to reading the records.
//   * TREE_LDS trees use their own record: MzHot (N, first_child, value_sum, prior and q = reward + discount * value(),
//     refreshed by the backup that changes it, so the walk does not divide value_sum / N again at every visit) plus one
//     float of reward per node; sqrt(N) of the walk comes from a table filled with the same IEEE sqrt.  Same operations on
//     the same operands as mz_descend / mz_grow_backup: same bits (tests: fused search == step-by-step == CPython).
constexpr int kMzHidden = 64;   // floats of a hidden state (= kMzH below)
struct __attribute__((aligned(16))) MzHot {
    int32_t N, first_child;
    double value_sum, prior, q;
};
static_assert(sizeof(MzHot) == 32, "MzHot layout");

// N IEEE fp64 divisions x[i] / y[i] with their dependent chains INTERLEAVED: a division is a chain of 11 dependent fp64
// operations (~200 cycles for a wave, however few lanes are active), and the compiler emits one chain after the other.
// The operations are exactly those of its own expansion of `x / y` (v_div_scale x 2, v_rcp, the Newton steps,
// v_div_fmas, v_div_fixup): same bits.
template <int N>
__device__ __forceinline__ void mz_divide(const double (&x)[N], const double (&y)[N], double (&out)[N]) {
    double d[N], ns[N], r[N], e[N], m[N];
    bool flag[N], unused;
#pragma unroll
    for (int i = 0; i < N; ++i) d[i] = __builtin_amdgcn_div_scale(x[i], y[i], false, &unused);
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = __builtin_amdgcn_rcp(d[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) e[i] = __builtin_fma(-d[i], r[i], 1.0);
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = __builtin_fma(r[i], e[i], r[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) e[i] = __builtin_fma(-d[i], r[i], 1.0);
#pragma unroll
    for (int i = 0; i < N; ++i) ns[i] = __builtin_amdgcn_div_scale(x[i], y[i], true, &flag[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = __builtin_fma(r[i], e[i], r[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) m[i] = ns[i] * r[i];
#pragma unroll
    for (int i = 0; i < N; ++i) e[i] = __builtin_fma(-d[i], m[i], ns[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) m[i] = __builtin_amdgcn_div_fmas(e[i], r[i], m[i], flag[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = __builtin_amdgcn_div_fixup(m[i], y[i], x[i]);
}

// the value of lane l ^ 1 / l ^ 2 (DPP quad_perm: one VALU move per dword)
__device__ __forceinline__ int mz_quad_xor1(int x) { return __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true); }
__device__ __forceinline__ int mz_quad_xor2(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true); }
template <int WHICH>
__device__ __forceinline__ double mz_quad_xor(double x) {
    const long long b = __double_as_longlong(x);
    const int lo = (int)b, hi = (int)(b >> 32);
    const int olo = WHICH == 1 ? mz_quad_xor1(lo) : mz_quad_xor2(lo), ohi = WHICH == 1 ? mz_quad_xor1(hi) : mz_quad_xor2(hi);
    return __longlong_as_double(((long long)ohi << 32) | (unsigned int)olo);
}

// the value of lane J of the quad in all four
template <int J>
__device__ __forceinline__ double mz_quad_bcast(double x) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, J * 0x55, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), J * 0x55, 0xF, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Touch a hidden-state row the walk may need next (global_load into a register nobody reads: the line is on its way to
// L2 / L1 while the walk goes on).  `sink` keeps the destination register reserved until the caller has waited for its
// vector memory operations (they return in order, and the compiler's own s_waitcnt counts only get stricter).
__device__ __forceinline__ void mz_touch(const float *p, float &sink) {
    asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(p) : "memory");
}
__device__ __forceinline__ void mz_touch_done(float &sink) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink) : : "memory");
}

// mz_descend on the MzHot trees with the FOUR lanes of a quad walking one tree together: lane `ca` of the quad scores
// ONE child of a level (the two divisions of a score are the walk's long operations; a single wave issues one
// instruction every ~5 cycles, so the walk is written for few instructions: straight-line, selects instead of
// branches), the quad then takes the best (score, action) -- larger action on a tie, as the sequential scan -- and all
// four lanes step down together.  Up to 2 actions lanes 2, 3 mirror lanes 0, 1 (one exchange); 3 or 4 actions take a
// second exchange; more fall back to every lane scanning the children ca, ca + 4, ...
// Every lane returns the same (depth, parent, action, leaf).
// AFIX > 0: the action count is a compile-time constant (2 for CartPole's whole MOVES: the per-level code loses its runtime tests)
template <int AFIX = 0, typename HotP, typename PathP, typename TabP>
__device__ __forceinline__ int mz_descend_hot(const MzDev &E, HotP hot, PathP path, TabP pb_log, TabP sq_tab, double lo, double hi,
                                              int ca, const float *hidden_g, float &sink, int &par_out, int &act_out, int &leaf_out) {
    const int A = AFIX > 0 ? AFIX : E.n_actions;
    int node = 0, depth = 0, last_action = 0, par = 0;
    path[0] = 0;
    int fc = hot[0].first_child, pn = hot[0].N;
    const bool ranged = hi > lo;   // MinMaxStats.normalize
    const double divisor = ranged ? hi - lo : 1.0;
    if (A <= 4) {
        const int my_a = A <= 2 ? (ca & 1) : ca;
        const bool has_child = my_a < A;
        const int a_eff = has_child ? my_a : 0;
        while (fc >= 0 && depth + 1 < E.path_stride) {
            const MzHot ch = hot[fc + a_eff];
            const int idx = pn <= E.n_sims + 1 ? pn : E.n_sims + 1;
            const double pb_c0 = pb_log[idx] + E.pb_c_init;
            const double sq = pn <= E.n_sims + 1 ? sq_tab[idx] : sqrt((double)pn);
            const double num[2] = {sq, ch.q - lo}, den[2] = {(double)(ch.N + 1), divisor};
            double quo[2];
            mz_divide<2>(num, den, quo);
            const double own = pb_c0 * quo[0] * ch.prior + (ch.N > 0 ? (ranged ? quo[1] : ch.q) : 0.0);
            const double score = has_child ? own : -INFINITY;
            // the neighbour's child (action my_a ^ 1) wins with a larger score, or the same score and the larger action
            double ob = mz_quad_xor<1>(score);
            int ofc = mz_quad_xor1(ch.first_child), on = mz_quad_xor1(ch.N);
            bool take = ob > score || (ob == score && (my_a & 1) == 0);
            double best = take ? ob : score;
            int besta = take ? (my_a ^ 1) : my_a, best_fc = take ? ofc : ch.first_child, best_n = take ? on : ch.N;
            if (A > 2) {
                ob = mz_quad_xor<2>(best);
                const int oa = mz_quad_xor2(besta);
                ofc = mz_quad_xor2(best_fc);
                on = mz_quad_xor2(best_n);
                take = ob > best || (ob == best && oa > besta);
                besta = take ? oa : besta;
                best_fc = take ? ofc : best_fc;
                best_n = take ? on : best_n;
            }
            par = node;
            last_action = besta;
            node = fc + besta;
            fc = best_fc;
            pn = best_n;
            depth += 1;
            path[depth] = node;
            // the node stepped into is the parent whose hidden state the gather reads if the walk ends below it: the
            // quad touches the four 64-byte pieces of its row now (~1 us from the Infinity Cache otherwise)
            mz_touch(hidden_g + (long long)node * kMzHidden + 16 * ca, sink);
        }
    } else {
        while (fc >= 0 && depth + 1 < E.path_stride) {
            const int idx = pn <= E.n_sims + 1 ? pn : E.n_sims + 1;
            const double pb_c0 = pb_log[idx] + E.pb_c_init;
            const double sq = pn <= E.n_sims + 1 ? sq_tab[idx] : sqrt((double)pn);
            double best = -INFINITY;
            int besta = -1, best_fc = -1, best_n = 0;
            for (int a = ca; a < A; a += 4) {
                const MzHot ch = hot[fc + a];
                const double num[2] = {sq, ch.q - lo}, den[2] = {(double)(ch.N + 1), divisor};
                double quo[2];
                mz_divide<2>(num, den, quo);
                const double score = pb_c0 * quo[0] * ch.prior + (ch.N > 0 ? (ranged ? quo[1] : ch.q) : 0.0);
                if (score >= best) {  // the later (larger) action wins a tie
                    best = score;
                    besta = a;
                    best_fc = ch.first_child;
                    best_n = ch.N;
                }
            }
            {   // best of the quad
                double ob = mz_quad_xor<1>(best);
                int oa = mz_quad_xor1(besta), ofc = mz_quad_xor1(best_fc), on = mz_quad_xor1(best_n);
                bool take = ob > best || (ob == best && oa > besta);
                best = take ? ob : best;
                besta = take ? oa : besta;
                best_fc = take ? ofc : best_fc;
                best_n = take ? on : best_n;
                ob = mz_quad_xor<2>(best);
                oa = mz_quad_xor2(besta);
                ofc = mz_quad_xor2(best_fc);
                on = mz_quad_xor2(best_n);
                take = ob > best || (ob == best && oa > besta);
                besta = take ? oa : besta;
                best_fc = take ? ofc : best_fc;
                best_n = take ? on : best_n;
            }
            par = node;
            last_action = besta;
            node = fc + besta;
            fc = best_fc;
            pn = best_n;
            depth += 1;
            path[depth] = node;
        }
    }
    par_out = par;
    act_out = last_action;
    leaf_out = node;
    return depth;
}

// expand + backup on the MzHot trees, the four lanes of a quad together: the backup takes four levels of the path at a
// time, lane `ca` the level d - ca (one load of its node, ONE value_sum / N division, one store); only
// v = reward + discount * v chains through the levels, so every lane runs that short chain on the quad's four rewards.
// A level above the root is played on the spare record `spare` (slot index relative to `hot` / `rew`; never read for a
// result).  MinMaxStats takes the quad's extremes (max / min do not depend on the order of the updates).
template <int MAXA, int AFIX = 0, typename HotP, typename RewP, typename PathP>
__device__ __forceinline__ void mz_grow_backup_hot(const MzDev &E, HotP hot, RewP rew, PathP path, int spare, int ca, int depth, int &top,
                                                   double &lo, double &hi, float reward_g, const float *probs, float value_g) {
    const int leaf = path[depth];
    const int A = AFIX > 0 ? AFIX : E.n_actions;
    if (top + A > E.cap) {
        atomicOr(E.err, RZ_FLAG_ARENA_FULL);
    } else {
        rew[leaf] = reward_g;
        hot[leaf].first_child = top;
#pragma unroll
        for (int a = 0; a < MAXA; ++a) {
            if (a < A) {
                hot[top + a] = MzHot{0, -1, 0.0, (double)probs[a], 0.0};
                rew[top + a] = 0.0f;
            }
        }
        top += A;
    }
    double v = (double)value_g;
    for (int d = depth; d >= 0; d -= 4) {
        const bool real = d - ca >= 0;
        const int slot = real ? path[real ? d - ca : 0] : spare;
        const double vs = hot[slot].value_sum, r = (double)rew[slot];
        const int n = real ? hot[slot].N + 1 : 1;
        const double r0 = mz_quad_bcast<0>(r), r1 = mz_quad_bcast<1>(r), r2 = mz_quad_bcast<2>(r), r3 = mz_quad_bcast<3>(r);
        const double v1 = r0 + E.discount * v;
        const double v2 = r1 + E.discount * v1;
        const double v3 = r2 + E.discount * v2;
        const double mine = ca == 0 ? v : ca == 1 ? v1 : ca == 2 ? v2 : v3;
        v = r3 + E.discount * v3;
        const double sum = real ? vs + mine : 0.0;
        const double nv = sum / (double)n;
        hot[slot].value_sum = sum;
        hot[slot].N = real ? n : 0;
        hot[slot].q = r + E.discount * nv;
        double up = real ? nv : -INFINITY, dn = real ? nv : INFINITY;   // MinMaxStats.update over the quad's levels
        double o = mz_quad_xor<1>(up);
        up = o > up ? o : up;
        o = mz_quad_xor<2>(up);
        up = o > up ? o : up;
        o = mz_quad_xor<1>(dn);
        dn = o < dn ? o : dn;
        o = mz_quad_xor<2>(dn);
        dn = o < dn ? o : dn;
        hi = up > hi ? up : hi;
        lo = dn < lo ? dn : lo;
    }
}

constexpr int kMzH = 64, kMzWaves = 4, kMzMaxA = 8, kMzTile = 16, kMzKX = kMzH + kMzMaxA, kMzMaxGpw = 16, kMzObs = 8;
constexpr int kMzRedRows = 4 + kMzMaxA;   // min, max, reward, value, logits
constexpr int kMzHeadRows = 2 + kMzMaxA;  // rew2, val, pol rows
typedef float mz_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 mz_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 mz_f16x4 __attribute__((ext_vector_type(4)));

struct MzModel {        // device pointers, weights k-major: w[k][unit]
    const float *dyn1_w, *dyn1_b, *dyn2_w, *dyn2_b, *rew1_w, *rew1_b, *rew2_w, *rew2_b, *pre1_w, *pre1_b, *pol_w, *pol_b,
        *val_w, *val_b, *rep1_w, *rep1_b, *rep2_w, *rep2_b;   // rep1_w: [kMzObs][64], rows >= obs_dim zero
    // whole MOVES run dyn1 / dyn2 / rew1 / pre1 on the f16 matrix pipe with hi + lo operand pairs (k_mz_search, F16): the powers of
    // two that bring each layer's largest weight into [2^13, 2^14) -- sw of dyn1 (its 64 state columns), dyn2, rew1, pre1 -- and the
    // scale s1 <= 16 of relu(dyn1)'s f16 pieces from a bound on them (states lie in [0, 1]): [5] floats
    const float *f16_scales;
};

struct MzTrace {        // optional per-simulation outputs for the parity tests: [n_sims][n_games] (probs: x n_actions)
    int32_t *parent, *action, *leaf;
    float *reward, *probs, *value;
};

struct MzPlay {         // rz_mz_play_cartpole
    int n_moves, row;                       // row = doubles per packed record = 4 + 4 + A
    double *state;                          // [G][4]
    long long *steps, *episode;             // [G]
    unsigned long long env_seed, noise_seed;
    double noise_frac, inv_temperature;     // inv_temperature <= 0: arg-max of the visit counts
    float alpha;
    // history on the device: every move's record -- obs (4) | action | reward | visits (A) | root value | done -- goes
    // to ring[environment][step % hist]; when an episode ends, its records are copied into `arena` as one contiguous
    // run and (environment, end step, length, first arena row) is appended to `entries`: the host reads finished
    // episodes, not moves
    double *ring;                           // [G][hist][row]
    int hist;
    long long t0;                           // global step index of the first move of this launch
    long long *ep_start;                    // [G]: step index where the running episode of an environment began
    double *arena;                          // [arena_rows][row]
    long long arena_rows;
    long long *counters;                    // [0] arena rows claimed, [1] entries, [2] episodes that did not fit (row -1)
    long long *entries;                     // [max_entries][4]
    long long max_entries;
};

// LDS of k_mz_search in bytes: activations, reductions, head weights, per-game scalars, paths, log table (+ the trees)
__host__ __device__ inline int mz_search_fixed_floats() {
    return kMzKX * kMzTile + (kMzH * kMzTile + 128) + kMzRedRows * kMzWaves * kMzTile + 16 + kMzHeadRows * kMzH + kMzObs * kMzTile + 16 +
           kMzTile * 16 +                  // (+ the environments of whole MOVES: 7 doubles per game, 8 reserved)
           2 * kMzH + kMzTile;             // (+ dyn1's action columns [2 actions][unit] and the games' actions: the f16 layers of whole
                                           // MOVES; every byte counts -- two workgroups of 16 games share a CU's 160 KB with 1.5 KB to spare)
}
__host__ __device__ inline int mz_search_lds_bytes(int gpw, int cap, int path_stride, int n_sims_cfg, bool tree_lds) {
    int bytes = mz_search_fixed_floats() * 4 + gpw * path_stride * 4;
    bytes = (bytes + 15) / 16 * 16;
    bytes += 2 * (((n_sims_cfg + 2) * 8 + 15) / 16 * 16);   // log and sqrt tables
    if (tree_lds) bytes += gpw * (cap + 1) * (int)sizeof(MzHot) + (gpw * (cap + 1) * 4 + 15) / 16 * 16;   // records (+ a spare per game) + rewards
    return bytes;
}

// Development aid (not built by default): -DRZ_MZ_PROFILE accumulates the shader-clock cycles wave 0 of workgroup 0 spends
// in each stage of k_mz_search into mz_prof[] (read with rz_mz_debug_profile).
#ifdef RZ_MZ_PROFILE
__device__ long long mz_prof[16];
#define MZ_TICK(i) do { const long long now_ = clock64(); prof_acc[i] += now_ - prof_t; prof_t = now_; } while (0)
#else
#define MZ_TICK(i)
#endif

// acc += W(16 units of this wave x 4 STEPS) . B([k][game] in LDS) on the matrix pipe
template <int STEPS>
__device__ __forceinline__ mz_f32x4 mz_tile(const float (&a)[STEPS], const float *B, int q, int n, mz_f32x4 acc) {
#pragma unroll
    for (int s = 0; s < STEPS; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], B[(4 * s + q) * kMzTile + n], acc, 0, 0, 0);
    return acc;
}

// sum over the 4 lanes that hold the rows of one game (q = 0 .. 3): every lane ends with the same bits
// the value of lane l ^ 16 / l ^ 32 (gfx950 v_permlane16_swap / v_permlane32_swap: two VALU operations instead of a trip
// through the LDS crossbar)
__device__ __forceinline__ float mz_xor16(float x, int l) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float((l & 16) ? r[0] : r[1]);
}
__device__ __forceinline__ float mz_xor32(float x, int l) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float((l & 32) ? r[0] : r[1]);
}
__device__ __forceinline__ float mz_rowsum(float x, int l) {
    x += mz_xor16(x, l);
    x += mz_xor32(x, l);
    return x;
}

// dot product of the 4 rows a lane holds (16w + 4q .. +3) with a head's weights, summed over the wave's 16 rows
__device__ __forceinline__ float mz_head_partial(const mz_f32x4 &p, const float *head_row, int w, int q, int l) {
    const float4 wv = *reinterpret_cast<const float4 *>(head_row + 16 * w + 4 * q);
    float s = p[0] * wv.x;
    s = fmaf(p[1], wv.y, s);
    s = fmaf(p[2], wv.z, s);
    s = fmaf(p[3], wv.w, s);
    return mz_rowsum(s, l);
}

// MAXA: the action count the per-action loops are unrolled for (register arrays of that size): 8 in general, 2 for whole MOVES --
// CartPole has two actions, and with 8-entry arrays the MOVES variants do not fit the 256 registers of two workgroups per CU
// (200 bytes of scratch per lane, ~20 scratch loads inside every simulation's serial chain)
template <bool TREE_LDS, bool MOVES, int MAXA = (MOVES ? 2 : kMzMaxA)>
__global__ __launch_bounds__(64 * kMzWaves, 2) void k_mz_search(MzDev E, MzModel M, float *hidden, int n_sims, int gpw, MzTrace T, MzPlay P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mz_lds[];
    constexpr int AFIX = MOVES ? 2 : 0;   // whole MOVES are CartPole's: two actions (rz_mz_play_cartpole checks)
    const int A = AFIX > 0 ? AFIX : E.n_actions, KX = kMzH + A;
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6, n = l & 15, q = l >> 4;
    const int g0 = blockIdx.x * gpw;
    float *XS = reinterpret_cast<float *>(mz_lds);   // [72][16]: parent state + one-hot action; later the scaled next state
    float *HP = XS + kMzKX * kMzTile;                // [64][16]: relu(dyn1) (relu(rep1) in the initial inference)
    float *RED = HP + kMzH * kMzTile + 128;          // [12][4 waves][16]: per-wave min / max / reward / value / logit partials
    float *HB = RED + kMzRedRows * kMzWaves * kMzTile;   // [0] rew2 bias, [1] val bias, [2 .. 2 + A) pol biases
    float *HW = HB + 16;                             // [10][64]: rew2 / val / pol weights
    float *OBS = HW + kMzHeadRows * kMzH;            // [8][16]: observations, [k][game] (MOVES)
    int *Gleaf = reinterpret_cast<int *>(OBS + kMzObs * kMzTile);   // [16]
    // MOVES: the environments (x, x_dot, theta, theta_dot, steps, episode, episode start) wait in LDS between a move's first and
    // last stage -- 14 registers that the simulations in between would otherwise carry
    double *ENV = reinterpret_cast<double *>(Gleaf + 16);           // [16][8]
    float *W1A = reinterpret_cast<float *>(ENV + kMzTile * 8);       // [2][64]: dyn1's weights of the one-hot action rows (F16: two actions)
    int *Gact = reinterpret_cast<int *>(W1A + 2 * kMzH);             // [16]: the action of the edge into each game's leaf (F16)
    int32_t *PATH = Gact + kMzTile;                                  // [gpw][path_stride]
    unsigned char *pp = mz_lds + (mz_search_fixed_floats() * 4 + gpw * E.path_stride * 4 + 15) / 16 * 16;
    double *PBL = reinterpret_cast<double *>(pp);    // [n_sims + 2]: the host's log table
    pp += ((E.n_sims + 2) * 8 + 15) / 16 * 16;
    double *SQT = reinterpret_cast<double *>(pp);    // [n_sims + 2]: sqrt(n)
    pp += ((E.n_sims + 2) * 8 + 15) / 16 * 16;
    const int tcap = E.cap + 1;                      // slots per game in LDS: the tree + one spare record (mz_grow_backup_hot)
    MzHot *TREE = reinterpret_cast<MzHot *>(pp);     // [gpw][tcap] (TREE_LDS)
    float *REW = reinterpret_cast<float *>(pp + gpw * tcap * (int)sizeof(MzHot));   // [gpw][tcap] rewards (TREE_LDS)

    // ---- once: weight fragments into registers (A operand of 16x16x4: lane holds W[unit 16w + n][k = 4s + q])
    // (the representation network's fragments -- 18 + 8 registers, used once per MOVE -- are fetched at the start of every
    // move instead: kept for the whole launch they push the MOVES variants past the 256 registers of two workgroups per CU
    // and into scratch, in the middle of the per-simulation chain)
    // F16 (whole MOVES): the four 64 x 64 layers of a simulation on the f16 matrix pipe, v_mfma_f32_16x16x32_f16 with every
    // f32 operand a hi + lo pair of f16 values (three MFMAs per product, f32 accumulation: the network trunk's arithmetic,
    // rz_net.hip) -- 6 MFMAs of 16 cycles per layer instead of 16 .. 18 of 32.  The activations then live in LDS as
    // [game][hi: 64 f16 | lo: 64 f16 | 32 B pad] (a B fragment = one ds_read_b128: lane = 16 (k block) + game), the hidden
    // states in HBM as the same 256 bytes (the gather is a plain copy), dyn1's one-hot action rows are added as a column of
    // f32 weights.  Activation scales: 16 for states (they lie in [0, 1]), s1 for relu(dyn1) (rz_mz_load_model's bound).
    constexpr bool F16 = MOVES;
    constexpr int kRec = 288;   // bytes of a game's activation record
    float a1[F16 ? 1 : kMzKX / 4], a2[F16 ? 1 : kMzH / 4], ar[F16 ? 1 : kMzH / 4], ap[F16 ? 1 : kMzH / 4];
    mz_f16x8 f1[2][2], f2[2][2], fr[2][2], fp[2][2];   // (F16) [K-step of 32][hi | lo]: lane = 16 g + r holds W[k = 32 s + 8 g + j][unit 16 w + r]
    float d1 = 1.0f, d2 = 1.0f, dr = 1.0f, dp = 1.0f, s1 = 1.0f;
    const int unit = 16 * w + n;
    if constexpr (F16) {
        const float sw1 = M.f16_scales[0], sw2 = M.f16_scales[1], swr = M.f16_scales[2], swp = M.f16_scales[3];
        s1 = M.f16_scales[4];
        d1 = 1.0f / (16.0f * sw1);
        d2 = 1.0f / (s1 * sw2);
        dr = 1.0f / (s1 * swr);
        dp = 1.0f / (16.0f * swp);
        auto frag = [&](const float *wk, float sw, mz_f16x8 (&f)[2][2]) {
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = wk[(32 * st + 8 * q + j) * kMzH + unit] * sw;
                    const _Float16 hi = (_Float16)v;
                    f[st][0][j] = hi;
                    f[st][1][j] = (_Float16)(v - (float)hi);
                }
        };
        frag(M.dyn1_w, sw1, f1);
        frag(M.dyn2_w, sw2, f2);
        frag(M.rew1_w, swr, fr);
        frag(M.pre1_w, swp, fp);
        static_assert(!F16 || AFIX == 2, "the f16 layers are built for CartPole's two actions");
        for (int i = tid; i < 2 * kMzH; i += 64 * kMzWaves) W1A[i] = M.dyn1_w[kMzH * kMzH + i];   // rows 64, 65 of [k][unit]
    } else {
#pragma unroll
        for (int s = 0; s < kMzKX / 4; ++s) {
            const int k = 4 * s + q;
            a1[s] = k < KX ? M.dyn1_w[k * kMzH + unit] : 0.0f;
        }
#pragma unroll
        for (int s = 0; s < kMzH / 4; ++s) {
            const int k = 4 * s + q;
            a2[s] = M.dyn2_w[k * kMzH + unit];
            ar[s] = M.rew1_w[k * kMzH + unit];
            ap[s] = M.pre1_w[k * kMzH + unit];
        }
    }
    // (F16) acc += W . X over the 64 state values of the games' records at `rec` (LDS), hi + lo pairs: 6 MFMAs
    auto mfma16 = [&](const mz_f16x8 (&f)[2][2], const unsigned char *rec, mz_f32x4 acc) {
        const unsigned char *p = rec + n * kRec + q * 16;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const mz_f16x8 bh = *reinterpret_cast<const mz_f16x8 *>(p + st * 64), bl = *reinterpret_cast<const mz_f16x8 *>(p + 128 + st * 64);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[st][0], bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[st][0], bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[st][1], bh, acc, 0, 0, 0);
        }
        return acc;
    };
    // (F16) the lane's 4 values (units 16 w + 4 q + i of game n), already scaled, as hi + lo pieces into the games' records
    auto store16 = [&](unsigned char *rec, const float (&z)[4], mz_f16x4 *hi_out = nullptr, mz_f16x4 *lo_out = nullptr) {
        mz_f16x4 hi, lo;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            hi[i] = (_Float16)z[i];
            lo[i] = (_Float16)(z[i] - (float)hi[i]);
        }
        unsigned char *p = rec + n * kRec + (16 * w + 4 * q) * 2;
        *reinterpret_cast<mz_f16x4 *>(p) = hi;
        *reinterpret_cast<mz_f16x4 *>(p + 128) = lo;
        if (hi_out) {
            *hi_out = hi;
            *lo_out = lo;
        }
    };
    unsigned char *XS16 = reinterpret_cast<unsigned char *>(XS), *HP16 = reinterpret_cast<unsigned char *>(HP);
    mz_f32x4 b1, b2, br, bp;   // biases in the C/D layout: rows 16w + 4q + i
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 16 * w + 4 * q + i;
        b1[i] = M.dyn1_b[row];
        b2[i] = M.dyn2_b[row];
        br[i] = M.rew1_b[row];
        bp[i] = M.pre1_b[row];
    }
    if (tid == 0) {
        HB[0] = M.rew2_b[0];
        HB[1] = M.val_b[0];
    }
    if (tid < A) HB[2 + tid] = M.pol_b[tid];
    for (int i = tid; i < kMzHeadRows * kMzH; i += 64 * kMzWaves) {
        const int row = i >> 6, k = i & 63;
        HW[i] = row == 0 ? M.rew2_w[k] : row == 1 ? M.val_w[k] : row - 2 < A ? M.pol_w[(row - 2) * kMzH + k] : 0.0f;
    }
    for (int i = tid; i < kMzKX * kMzTile; i += 64 * kMzWaves) XS[i] = 0.0f;   // (the padding rows 64 + A .. 71 stay zero)
    for (int i = tid; i < kMzObs * kMzTile; i += 64 * kMzWaves) OBS[i] = 0.0f;
    for (int i = tid; i < E.n_sims + 2; i += 64 * kMzWaves) {
        PBL[i] = E.pb_log[i];
        SQT[i] = sqrt((double)i);
    }
    // the trees of this workgroup's games
    // wave 0 walks the trees: the four lanes of quad `ge` own game g0 + ge together -- they score different children in
    // the walk and otherwise run the same tree code on the same values (same LDS words, same bits); `lead` writes to HBM
    const int ge = l >> 2, ca = l & 3;
    const bool mine = w == 0 && ge < gpw && g0 + ge < E.n_games;
    const bool lead = mine && ca == 0;
    const int g = g0 + (ge < gpw ? ge : 0);
    const bool live_n = n < gpw && g0 + n < E.n_games;           // column n of the tiles is a game
    int top = 0, depth = 0;
    double lo = 0.0, hi = 0.0;
    double *env_g = ENV + 8 * ge;
    auto env_load = [&](MzCartPole &c, long long &ep0) {
        c = MzCartPole{env_g[0], env_g[1], env_g[2], env_g[3], __double_as_longlong(env_g[4]), __double_as_longlong(env_g[5])};
        ep0 = __double_as_longlong(env_g[6]);
    };
    auto env_store = [&](const MzCartPole &c, long long ep0) {   // (the four lanes of a quad write the same values)
        env_g[0] = c.x;
        env_g[1] = c.x_dot;
        env_g[2] = c.theta;
        env_g[3] = c.theta_dot;
        env_g[4] = __longlong_as_double(c.steps);
        env_g[5] = __longlong_as_double(c.episode);
        env_g[6] = __longlong_as_double(ep0);
    };
    if (mine) {
        if (MOVES) {
            env_store(MzCartPole{P.state[4 * g], P.state[4 * g + 1], P.state[4 * g + 2], P.state[4 * g + 3], P.steps[g], P.episode[g]},
                      P.ep_start[g]);
        } else {
            top = E.top[g];
            lo = E.vmin[g];
            hi = E.vmax[g];
        }
    }
    if (TREE_LDS && !MOVES) {
        for (int ee = 0; ee < gpw && g0 + ee < E.n_games; ++ee) {
            const MzNode *src = E.nodes + (long long)(g0 + ee) * E.cap;
            const int used = E.top[g0 + ee] < E.cap ? E.top[g0 + ee] : E.cap;
            for (int i = tid; i < used; i += 64 * kMzWaves) {
                const MzNode nd = src[i];
                TREE[ee * tcap + i] = MzHot{nd.N, nd.first_child, nd.value_sum, nd.prior,
                                            nd.N > 0 ? (double)nd.reward + E.discount * node_value(nd) : 0.0};
                REW[ee * tcap + i] = nd.reward;
            }
        }
    }
    MzNode *nodes = E.nodes + (long long)g * E.cap;                 // (!TREE_LDS)
    MzHot *hot = TREE + (ge < gpw ? ge : 0) * tcap;                 // (TREE_LDS)
    float *rew = REW + (ge < gpw ? ge : 0) * tcap;
    int32_t *path = PATH + (ge < gpw ? ge : 0) * E.path_stride;
    __syncthreads();
    if (MOVES && mine) {   // (after the zero fill of OBS)
        OBS[0 * kMzTile + ge] = (float)env_g[0];
        OBS[1 * kMzTile + ge] = (float)env_g[1];
        OBS[2 * kMzTile + ge] = (float)env_g[2];
        OBS[3 * kMzTile + ge] = (float)env_g[3];
    }

    // ---- the pieces the initial inference and a simulation share
    // per-sample min-max scaling of a new state t (network.py scale_hidden) from the per-wave extremes in RED: the scaled
    // rows go to XS (the next layer's input) and to the hidden-state slot Gleaf[game] of the game
    auto scale_and_store = [&](const mz_f32x4 &t) {
        float mn = RED[n], mx = RED[kMzWaves * kMzTile + n];
#pragma unroll
        for (int u = 1; u < kMzWaves; ++u) {
            mn = fminf(mn, RED[u * kMzTile + n]);
            mx = fmaxf(mx, RED[(kMzWaves + u) * kMzTile + n]);
        }
        const float inv = fmaxf(mx - mn, 1e-5f);
        float4 sv;
        if constexpr (F16) {   // one division per game and lane instead of four (1 ulp from x / inv: the pieces carry 1e-7 anyway)
            const float rinv = 1.0f / inv;
            sv.x = (t[0] - mn) * rinv;
            sv.y = (t[1] - mn) * rinv;
            sv.z = (t[2] - mn) * rinv;
            sv.w = (t[3] - mn) * rinv;
        } else {
            sv.x = (t[0] - mn) / inv;
            sv.y = (t[1] - mn) / inv;
            sv.z = (t[2] - mn) / inv;
            sv.w = (t[3] - mn) / inv;
        }
        if constexpr (F16) {
            const float z[4] = {sv.x * 16.0f, sv.y * 16.0f, sv.z * 16.0f, sv.w * 16.0f};
            mz_f16x4 hi, lo;
            store16(XS16, z, &hi, &lo);
            if (live_n) {   // the same 256 bytes as the game's hidden-state slot
                unsigned char *dst = reinterpret_cast<unsigned char *>(hidden + ((long long)(g0 + n) * E.cap + Gleaf[n]) * kMzH) + (16 * w + 4 * q) * 2;
                *reinterpret_cast<mz_f16x4 *>(dst) = hi;
                *reinterpret_cast<mz_f16x4 *>(dst + 128) = lo;
            }
        } else {
            XS[(16 * w + 4 * q + 0) * kMzTile + n] = sv.x;
            XS[(16 * w + 4 * q + 1) * kMzTile + n] = sv.y;
            XS[(16 * w + 4 * q + 2) * kMzTile + n] = sv.z;
            XS[(16 * w + 4 * q + 3) * kMzTile + n] = sv.w;
            if (live_n) *reinterpret_cast<float4 *>(hidden + ((long long)(g0 + n) * E.cap + Gleaf[n]) * kMzH + 16 * w + 4 * q) = sv;
        }
    };
    auto wave_extremes = [&](const mz_f32x4 &t) {
        float mn = fminf(fminf(t[0], t[1]), fminf(t[2], t[3])), mx = fmaxf(fmaxf(t[0], t[1]), fmaxf(t[2], t[3]));
        mn = fminf(mn, mz_xor16(mn, l));
        mx = fmaxf(mx, mz_xor16(mx, l));
        mn = fminf(mn, mz_xor32(mn, l));
        mx = fmaxf(mx, mz_xor32(mx, l));
        if (q == 0) {
            RED[(0 * kMzWaves + w) * kMzTile + n] = mn;
            RED[(1 * kMzWaves + w) * kMzTile + n] = mx;
        }
    };
    // prediction f(s) on the state in XS: p1 = relu(pre1 s), per-wave partial sums of the value and policy heads
    auto predict_partials = [&]() {
        mz_f32x4 p;
        if constexpr (F16) {
            p = mfma16(fp, XS16, mz_f32x4{0.0f, 0.0f, 0.0f, 0.0f});
#pragma unroll
            for (int i = 0; i < 4; ++i) p[i] = fmaxf(fmaf(p[i], dp, bp[i]), 0.0f);
        } else {
            p = mz_tile(ap, XS, q, n, bp);
#pragma unroll
            for (int i = 0; i < 4; ++i) p[i] = fmaxf(p[i], 0.0f);
        }
        const float pv = mz_head_partial(p, HW + 1 * kMzH, w, q, l);
        if (q == 0) RED[(3 * kMzWaves + w) * kMzTile + n] = pv;
#pragma unroll
        for (int a = 0; a < MAXA; ++a) {
            if (a < A) {
                const float pl = mz_head_partial(p, HW + (2 + a) * kMzH, w, q, l);
                if (q == 0) RED[((4 + a) * kMzWaves + w) * kMzTile + n] = pl;
            }
        }
    };
    // (lane that owns a game) value and softmax(policy logits) from the partial sums, waves in order
    auto finish_prediction = [&](float &value, float (&probs)[MAXA]) {
        value = HB[1];
#pragma unroll
        for (int u = 0; u < kMzWaves; ++u) value += RED[(3 * kMzWaves + u) * kMzTile + ge];
        float logit[MAXA], mxl = -INFINITY;
#pragma unroll
        for (int a = 0; a < MAXA; ++a) {
            logit[a] = -INFINITY;
            if (a < A) {
                float x = HB[2 + a];
#pragma unroll
                for (int u = 0; u < kMzWaves; ++u) x += RED[((4 + a) * kMzWaves + u) * kMzTile + ge];
                logit[a] = x;
                mxl = fmaxf(mxl, x);
            }
        }
        float den = 0.0f;
#pragma unroll
        for (int a = 0; a < MAXA; ++a) {
            probs[a] = a < A ? expf(logit[a] - mxl) : 0.0f;
            den += probs[a];
        }
#pragma unroll
        for (int a = 0; a < MAXA; ++a) probs[a] = probs[a] / den;
    };

#ifdef RZ_MZ_PROFILE
    long long prof_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, prof_t = clock64();
#endif
    float touch_sink = 0.0f;   // mz_touch
    const int n_moves = MOVES ? P.n_moves : 1;
    for (int move = 0; move < n_moves; ++move) {
        if (MOVES) {
            // ---- initial inference: s0 = scale(rep2 relu(rep1 obs)) -> hidden slot 0; root priors = softmax(pol(relu(pre1 s0)))
            // the representation network's fragments and biases, from L2 (laundered pointers: the loads stay inside the move)
            float arep1[kMzObs / 4], arep2[kMzH / 4];
            mz_f32x4 brep1, brep2;
            {
                const float *r1w = M.rep1_w, *r2w = M.rep2_w, *r1b = M.rep1_b, *r2b = M.rep2_b;
                asm volatile("" : "+s"(r1w), "+s"(r2w), "+s"(r1b), "+s"(r2b));
#pragma unroll
                for (int s = 0; s < kMzObs / 4; ++s) arep1[s] = r1w[(4 * s + q) * kMzH + unit];
#pragma unroll
                for (int s = 0; s < kMzH / 4; ++s) arep2[s] = r2w[(4 * s + q) * kMzH + unit];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    brep1[i] = r1b[16 * w + 4 * q + i];
                    brep2[i] = r2b[16 * w + 4 * q + i];
                }
            }
            __syncthreads();   // (OBS of this move is written; the last move's reads of RED / XS are over)
            if (tid < kMzTile) Gleaf[tid] = 0;
            {
                mz_f32x4 h = mz_tile(arep1, OBS, q, n, brep1);
#pragma unroll
                for (int i = 0; i < 4; ++i) HP[(16 * w + 4 * q + i) * kMzTile + n] = fmaxf(h[i], 0.0f);
            }
            __syncthreads();
            const mz_f32x4 t0 = mz_tile(arep2, HP, q, n, brep2);
            wave_extremes(t0);
            __syncthreads();
            scale_and_store(t0);
            __syncthreads();
            predict_partials();
            __syncthreads();
            if (w == 0) {
                float value, probs[MAXA];
                finish_prediction(value, probs);
                if (mine) {
                    // expand_node(root) + add_exploration_noise: prior * (1 - frac) + noise * frac (k_mz_init), fresh MinMaxStats
                    // (`gs`: the game's share of the keys, opaque per move -- hoisted out of the move loop the three hashes of it lived
                    // in registers for the whole launch, and three of them in scratch memory)
                    unsigned long long gs = (unsigned long long)g << 24;
                    asm volatile("" : "+v"(gs));
                    const unsigned long long key = mz_splitmix64(mz_splitmix64(mz_splitmix64(P.noise_seed ^ gs) ^
                                                                                (unsigned long long)__double_as_longlong(env_g[5])) ^
                                                                 (unsigned long long)__double_as_longlong(env_g[4]));
                    float gam[MAXA], gsum = 0.0f;
#pragma unroll
                    for (int a = 0; a < MAXA; ++a) {
                        gam[a] = a < A ? mz_gamma(P.alpha, (uint32_t)(key >> 32) + 0x9E3779B9u * (uint32_t)(a + 1) + (uint32_t)key) : 0.0f;
                        gsum += gam[a];
                    }
                    if (TREE_LDS) {
                        hot[0] = MzHot{0, 1, 0.0, 0.0, 0.0};
                        rew[0] = 0.0f;
                    } else {
                        nodes[0] = MzNode{0, 1, 0.0, 0.0, 0.0f, 0};
                    }
#pragma unroll
                    for (int a = 0; a < MAXA; ++a) {
                        if (a < A) {
                            double pr = (double)probs[a];
                            if (P.noise_frac > 0.0) {
                                double nf = P.noise_frac;
                                asm volatile("" : "+v"(nf));   // (1 - nf formed here, not carried through the launch)
                                pr = pr * (1.0 - nf) + ((double)gam[a] / (double)gsum) * nf;
                            }
                            if (TREE_LDS) {
                                hot[1 + a] = MzHot{0, -1, 0.0, pr, 0.0};
                                rew[1 + a] = 0.0f;
                            } else {
                                nodes[1 + a] = MzNode{0, -1, 0.0, pr, 0.0f, 0};
                            }
                        }
                    }
                    top = 1 + A;
                    lo = INFINITY;
                    hi = -INFINITY;
                    depth = 0;
                }
            }
            MZ_TICK(13);
        }
    for (int sim = 0; sim < n_sims; ++sim) {
        // S0 (wave 0): select, one lane per game; then the gather of the parents' hidden states into XS[k][game]
        // (lane -> game l / 4, four 16-byte pieces of its 256-byte row) and the one-hot action rows
        int par = 0, act = 0, lf = 0;
        if (w == 0) {
            if (mine)
                depth = TREE_LDS ? mz_descend_hot<AFIX>(E, hot, path, PBL, SQT, lo, hi, ca, hidden + (long long)g * E.cap * kMzH, touch_sink, par, act, lf)
                                 : mz_descend(E, nodes, path, PBL, lo, hi, par, act, lf);
            MZ_TICK(0);
#ifdef RZ_MZ_PROFILE
            {   // the wave walks as long as its deepest game
                int md = mine ? depth : 0;
                for (int off = 32; off >= 1; off >>= 1) md = max(md, __shfl_xor(md, off));
                prof_acc[15] += md;
            }
#endif
            if (ca == 0) Gleaf[ge] = lf;
            const int ee = ge, pe = par;   // (the quad of a game also gathers its parent's state: 4 x 64 bytes per lane)
            const bool live_e = mine;
            const float4 *src = reinterpret_cast<const float4 *>(hidden + ((long long)(g0 + (live_e ? ee : 0)) * E.cap + pe) * kMzH);
            if constexpr (F16) {   // the slot holds the record's 256 bytes as they are: a copy (16 games x 4 lanes x 4 pieces)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = (l & 3) + 4 * j;
                    const float4 v = live_e ? src[c] : float4{0.0f, 0.0f, 0.0f, 0.0f};
                    *reinterpret_cast<float4 *>(XS16 + ee * kRec + c * 16) = v;
                }
                if (ca == 0) Gact[ge] = mine ? act : 0;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = (l & 3) + 4 * j;
                    const float4 v = live_e ? src[c] : float4{0.0f, 0.0f, 0.0f, 0.0f};
                    XS[(4 * c + 0) * kMzTile + ee] = v.x;
                    XS[(4 * c + 1) * kMzTile + ee] = v.y;
                    XS[(4 * c + 2) * kMzTile + ee] = v.z;
                    XS[(4 * c + 3) * kMzTile + ee] = v.w;
                }
                const int an = __shfl(act, 4 * n);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int a = q + 4 * j;
                    if (a < A) XS[(kMzH + a) * kMzTile + n] = (live_n && an == a) ? 1.0f : 0.0f;
                }
            }
            mz_touch_done(touch_sink);
            MZ_TICK(1);
        }
        __syncthreads();
        MZ_TICK(2);
        // S1: h1 = relu(dyn1 [x, onehot(a)])
        if constexpr (F16) {
            const mz_f32x4 acc = mfma16(f1, XS16, mz_f32x4{0.0f, 0.0f, 0.0f, 0.0f});
            const float4 wa = *reinterpret_cast<const float4 *>(W1A + Gact[n] * kMzH + 16 * w + 4 * q);   // the one-hot action's row of weights
            const float col[4] = {wa.x, wa.y, wa.z, wa.w};
            float z[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) z[i] = fmaxf(fmaf(acc[i], d1, col[i] + b1[i]), 0.0f) * s1;
            store16(HP16, z);
        } else {
            const mz_f32x4 acc = mz_tile(a1, XS, q, n, b1);
#pragma unroll
            for (int i = 0; i < 4; ++i) HP[(16 * w + 4 * q + i) * kMzTile + n] = fmaxf(acc[i], 0.0f);
        }
        MZ_TICK(3);
        __syncthreads();
        MZ_TICK(4);
        // S2: next state (dyn2, before scaling) and the reward head (rew2 relu(rew1 h1)) from the same activations
        mz_f32x4 t = b2;
        {
            mz_f32x4 r = br;
            if constexpr (F16) {
                t = mfma16(f2, HP16, mz_f32x4{0.0f, 0.0f, 0.0f, 0.0f});
                r = mfma16(fr, HP16, mz_f32x4{0.0f, 0.0f, 0.0f, 0.0f});
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    t[i] = fmaf(t[i], d2, b2[i]);
                    r[i] = fmaf(r[i], dr, br[i]);
                }
            } else {
#pragma unroll
                for (int s = 0; s < kMzH / 4; ++s) {
                    const float x = HP[(4 * s + q) * kMzTile + n];
                    t = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[s], x, t, 0, 0, 0);
                    r = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[s], x, r, 0, 0, 0);
                }
            }
            wave_extremes(t);
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = fmaxf(r[i], 0.0f);
            const float pr = mz_head_partial(r, HW, w, q, l);
            if (q == 0) RED[(2 * kMzWaves + w) * kMzTile + n] = pr;
        }
        MZ_TICK(5);
        __syncthreads();
        MZ_TICK(6);
        // S3: per-sample min-max scaling of the new state, stored for the leaf
        scale_and_store(t);
        MZ_TICK(7);
        __syncthreads();
        MZ_TICK(8);
        // S4: prediction on the new state
        predict_partials();
        MZ_TICK(9);
        __syncthreads();
        MZ_TICK(10);
        // S5 (wave 0): finish the heads, expand + backup
        if (w == 0) {
            float reward = HB[0];
#pragma unroll
            for (int u = 0; u < kMzWaves; ++u) reward += RED[(2 * kMzWaves + u) * kMzTile + ge];
            float value, probs[MAXA];
            finish_prediction(value, probs);
            MZ_TICK(11);
            if (mine) {
                if (!MOVES && lead && T.reward != nullptr) {
                    const long long o = (long long)sim * E.n_games + g;
                    T.parent[o] = par;
                    T.action[o] = act;
                    T.leaf[o] = lf;
                    T.reward[o] = reward;
                    T.value[o] = value;
#pragma unroll
                    for (int a = 0; a < MAXA; ++a)
                        if (a < A) T.probs[o * A + a] = probs[a];
                }
                if (TREE_LDS) mz_grow_backup_hot<MAXA, AFIX>(E, hot, rew, path, E.cap, ca, depth, top, lo, hi, reward, probs, value);
                else mz_grow_backup<MAXA>(E, nodes, path, depth, top, lo, hi, reward, probs, value);
            }
            MZ_TICK(12);
        }
        // (no barrier: the next select and gather are wave 0's too; XS -- which the gather rewrites -- was last read
        // before the barrier behind S4, and RED, which wave 0 reads above, is rewritten only after the next S0 barrier)
    }
        if (MOVES && mine) {
            // ---- the move: action ~ visits ^ (1 / T) (select_action), record for the host, environment step
            MzCartPole env;
            long long ep_start;
            env_load(env, ep_start);
            double wgt[MAXA], total = 0.0, best_w = -1.0;
            int visits[MAXA], arg = 0;
#pragma unroll
            for (int a = 0; a < MAXA; ++a) {
                visits[a] = a < A ? (TREE_LDS ? hot[1 + a].N : nodes[1 + a].N) : 0;
                wgt[a] = 0.0;
                if (a < A) {
                    wgt[a] = P.inv_temperature == 1.0 || P.inv_temperature <= 0.0 ? (double)visits[a] : pow((double)visits[a], P.inv_temperature);
                    total += wgt[a];
                    if (wgt[a] > best_w) {
                        best_w = wgt[a];
                        arg = a;
                    }
                }
            }
            int action = arg;
            if (P.inv_temperature > 0.0) {
                unsigned long long gs = (unsigned long long)g << 24;
                asm volatile("" : "+v"(gs));
                const unsigned long long key = mz_splitmix64(mz_splitmix64(mz_splitmix64(P.noise_seed ^ 0xA5A5A5A5ull ^ gs) ^
                                                                            (unsigned long long)env.episode) ^ (unsigned long long)env.steps);
                const double target = ((double)(key >> 11) / 9007199254740992.0) * total;
                double cum = 0.0;
                bool found = false;
                action = A - 1;
#pragma unroll
                for (int a = 0; a < MAXA; ++a) {
                    if (a < A) {
                        cum += wgt[a];
                        if (!found && cum > target) {
                            action = a;
                            found = true;
                        }
                    }
                }
            }
            const double root_sum = TREE_LDS ? hot[0].value_sum : nodes[0].value_sum;
            const int root_n = TREE_LDS ? hot[0].N : nodes[0].N;
            const long long t = P.t0 + move;
            double *ring_g = P.ring + (long long)g * P.hist * P.row;
            double *rec = ring_g + (t % P.hist) * P.row;
            double last[8 + MAXA];
            last[0] = (double)(float)env.x;   // the observation the search started from
            last[1] = (double)(float)env.x_dot;
            last[2] = (double)(float)env.theta;
            last[3] = (double)(float)env.theta_dot;
            last[4] = (double)action;
            last[5] = 1.0;
#pragma unroll
            for (int a = 0; a < MAXA; ++a) last[6 + a] = (double)visits[a];
            const double root_value = root_sum / (double)(root_n > 1 ? root_n : 1);
            const int done = cartpole_step(env, action);
            if (lead) {
#pragma unroll
                for (int c = 0; c < 6 + MAXA; ++c)
                    if (c < 6 + A) rec[c] = last[c];
                rec[6 + A] = root_value;
                rec[7 + A] = done ? 1.0 : 0.0;
            }
            if (done) {
                // the episode's records as one run in the arena: the quad copies the earlier moves from the ring, the
                // lead adds this move's from its registers
                const long long len = t + 1 - ep_start;
                long long off = -1;
                if (lead) {
                    off = (long long)atomicAdd(reinterpret_cast<unsigned long long *>(P.counters), (unsigned long long)len);
                    if (off + len > P.arena_rows) {
                        off = -1;
                        atomicAdd(reinterpret_cast<unsigned long long *>(P.counters + 2), 1ull);
                    }
                    const long long e = (long long)atomicAdd(reinterpret_cast<unsigned long long *>(P.counters + 1), 1ull);
                    if (e < P.max_entries) {
                        P.entries[4 * e + 0] = g;
                        P.entries[4 * e + 1] = t + 1;
                        P.entries[4 * e + 2] = len;
                        P.entries[4 * e + 3] = off;
                    }
                }
                off = __shfl(off, l & ~3);
                if (off >= 0) {
                    for (long long j = ca; j < len - 1; j += 4) {
                        const double *src = ring_g + ((ep_start + j) % P.hist) * P.row;
                        double *dst = P.arena + (off + j) * P.row;
                        for (int c = 0; c < P.row; ++c)
                            dst[c] = __hip_atomic_load(src + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (past the L1: written by other launches / moves)
                    }
                    if (lead) {
                        double *dst = P.arena + (off + len - 1) * P.row;
#pragma unroll
                        for (int c = 0; c < 6 + MAXA; ++c)
                            if (c < 6 + A) dst[c] = last[c];
                        dst[6 + A] = root_value;
                        dst[7 + A] = 1.0;
                    }
                }
                ep_start = t + 1;
                env.episode += 1;
                int g_reset = g;
                asm volatile("" : "+v"(g_reset));   // (as `gs` above: the reset's key is formed when an episode ends)
                cartpole_reset(env, P.env_seed, g_reset);
            }
            OBS[0 * kMzTile + ge] = (float)env.x;
            OBS[1 * kMzTile + ge] = (float)env.x_dot;
            OBS[2 * kMzTile + ge] = (float)env.theta;
            OBS[3 * kMzTile + ge] = (float)env.theta_dot;
            env_store(env, ep_start);
        }
        MZ_TICK(14);
    }
#ifdef RZ_MZ_PROFILE
    if (blockIdx.x == 0 && tid == 0)
        for (int i = 0; i < 16; ++i) mz_prof[i] = prof_acc[i];
#endif
    if (lead) {
        E.top[g] = top;
        E.vmin[g] = lo;
        E.vmax[g] = hi;
        E.depth[g] = depth;
        if (MOVES) {
            P.state[4 * g] = env_g[0];
            P.state[4 * g + 1] = env_g[1];
            P.state[4 * g + 2] = env_g[2];
            P.state[4 * g + 3] = env_g[3];
            P.steps[g] = __double_as_longlong(env_g[4]);
            P.episode[g] = __double_as_longlong(env_g[5]);
            P.ep_start[g] = __double_as_longlong(env_g[6]);
        }
    }
    if (TREE_LDS) {
        __syncthreads();
        if (w == 0 && ca == 0) Gleaf[ge] = top;   // (every wave needs the final tops)
        __syncthreads();
        for (int ee = 0; ee < gpw && g0 + ee < E.n_games; ++ee) {
            MzNode *dst = E.nodes + (long long)(g0 + ee) * E.cap;
            const int used = Gleaf[ee];
            for (int i = tid; i < used; i += 64 * kMzWaves) {
                const MzHot h = TREE[ee * tcap + i];
                dst[i] = MzNode{h.N, h.first_child, h.value_sum, h.prior, REW[ee * tcap + i], 0};
            }
        }
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
    bool model_loaded = false, representation_loaded = false;
    bool f16_ok = false;   // rz_mz_load_model: finite weights and a finite bound on relu(dyn1): whole MOVES may run their layers on the f16 pipe
    float *d_rep = nullptr;      // rz_mz_load_representation
    int n_cus = 256;             // of cfg.device
    int games_per_wg = 0;        // rz_mz_set_search_shape: 0 = chosen from n_games and n_cus
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
    if (hipDeviceGetAttribute(&e->n_cus, hipDeviceAttributeMultiprocessorCount, cfg->device) != hipSuccess || e->n_cus < 1) e->n_cus = 256;
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
    const size_t off_scales = total;
    total += 8;
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
    {   // the f16 layers of whole MOVES (MzModel::f16_scales)
        auto pow2_for = [](float wmax) {
            int ex = 0;
            if (wmax > 0.0f && std::isfinite(wmax)) {
                (void)std::frexp(wmax, &ex);
                ex = 14 - ex;   // wmax * 2^ex in [2^13, 2^14)
            }
            return std::ldexp(1.0f, ex);
        };
        auto wmax_of = [&](int idx, int n_in) {   // torch layout [out][in], the first 64 input columns
            float m = 0.0f;
            for (int o = 0; o < kMzH; ++o)
                for (int k = 0; k < kMzH; ++k) m = std::fmax(m, std::fabs(h_params[idx][(size_t)o * n_in + k]));
            return m;
        };
        host[off_scales + 0] = pow2_for(wmax_of(0, KX));
        host[off_scales + 1] = pow2_for(wmax_of(2, kMzH));
        host[off_scales + 2] = pow2_for(wmax_of(4, kMzH));
        host[off_scales + 3] = pow2_for(wmax_of(8, kMzH));
        double bound = 0.0;   // relu(dyn1 [state in [0, 1], one-hot action]) <= |bias| + the positive state weights + the largest action weight
        for (int o = 0; o < kMzH; ++o) {
            double acc = std::fabs((double)h_params[1][o]), amax = 0.0;
            for (int k = 0; k < kMzH; ++k) acc += std::fmax(0.0, (double)h_params[0][(size_t)o * KX + k]);
            for (int a = 0; a < A; ++a) amax = std::fmax(amax, (double)h_params[0][(size_t)o * KX + kMzH + a]);
            bound = std::fmax(bound, acc + amax);
        }
        float s1 = 16.0f;
        if (!(bound * 16.0 < 60000.0)) {
            int ex = 0;
            (void)std::frexp(60000.0 / (std::isfinite(bound) && bound > 0.0 ? bound : 1e30), &ex);
            s1 = std::ldexp(1.0f, ex - 1);
        }
        host[off_scales + 4] = s1;
        // the f16 pieces of whole MOVES cannot overflow when the weights are finite: hidden states are min-max scaled to [0, 1]
        // (stored x 16), relu(dyn1) is stored x s1 with bound x s1 < 60000, every weight x its power of two lies below 2^14.  Weights
        // WITHOUT finite values give no such bound: the whole-moves route then refuses to run (rz_mz_play_cartpole) instead of
        // filling the trees with inf -- the move-by-move routes (f32-input MFMA, PyTorch-ROCm) remain.
        bool finite = std::isfinite(bound);
        for (int i = 0; i < 14 && finite; ++i)
            for (size_t q = 0; q < sizes[i]; ++q)
                if (!std::isfinite(h_params[i][q])) {
                    finite = false;
                    break;
                }
        e->f16_ok = finite;
    }
    if (hipDeviceSynchronize() != hipSuccess) return mz_fail(RZ_ERR_HIP, "hipDeviceSynchronize failed");
    if (e->d_model == nullptr) {
        if (hipMalloc((void **)&e->d_model, total * sizeof(float)) != hipSuccess) return mz_fail(RZ_ERR_OOM, "hipMalloc failed (muzero model)");
        e->allocs.push_back(e->d_model);
        e->bytes += (long long)(total * sizeof(float));
        const int most = 160 * 1024;
        if (hipFuncSetAttribute((const void *)k_mz_search<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, most) != hipSuccess ||
            hipFuncSetAttribute((const void *)k_mz_search<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, most) != hipSuccess ||
            hipFuncSetAttribute((const void *)k_mz_search<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, most) != hipSuccess ||
            hipFuncSetAttribute((const void *)k_mz_search<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, most) != hipSuccess)
            return mz_fail(RZ_ERR_HIP, "hipFuncSetAttribute(dynamic LDS) failed");
    }
    if (hipMemcpy(e->d_model, host.data(), total * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        return mz_fail(RZ_ERR_HIP, "hipMemcpy(model) failed");
    const float *b = e->d_model;
    const MzModel seen = e->model;   // (the representation layers are loaded by their own call)
    e->model = MzModel{b + off[0], b + off[1], b + off[2], b + off[3], b + off[4], b + off[5], b + off[6], b + off[7],
                       b + off[8], b + off[9], b + off[10], b + off[11], b + off[12], b + off[13],
                       seen.rep1_w, seen.rep1_b, seen.rep2_w, seen.rep2_b, b + off_scales};
    e->model_loaded = true;
    return RZ_OK;
}

int rz_mz_load_representation(rz_muzero *e, const float *const *h_params, int32_t n_params, int32_t obs_dim, int32_t hidden) {
    int rc = mz_ready(e);
    if (rc != RZ_OK) return rc;
    if (h_params == nullptr || n_params != 4) return mz_fail(RZ_ERR_ARG, "expected rep1.weight, rep1.bias, rep2.weight, rep2.bias");
    if (hidden != kMzH || obs_dim < 1 || obs_dim > kMzObs) return mz_fail(RZ_ERR_ARG, "the fused moves are built for hidden size 64 and <= 8 observation features");
    for (int i = 0; i < 4; ++i)
        if (!h_params[i]) return mz_fail(RZ_ERR_ARG, "a parameter pointer is NULL");
    // rep1.w [64][obs_dim] -> k-major [8][64] (rows >= obs_dim zero), rep1.b, rep2.w [64][64] -> k-major, rep2.b
    const size_t off[4] = {0, (size_t)kMzObs * kMzH, (size_t)kMzObs * kMzH + kMzH, (size_t)kMzObs * kMzH + kMzH + (size_t)kMzH * kMzH};
    const size_t total = off[3] + kMzH;
    std::vector<float> host(total, 0.0f);
    for (int o = 0; o < kMzH; ++o) {
        for (int k = 0; k < obs_dim; ++k) host[off[0] + (size_t)k * kMzH + o] = h_params[0][(size_t)o * obs_dim + k];
        for (int k = 0; k < kMzH; ++k) host[off[2] + (size_t)k * kMzH + o] = h_params[2][(size_t)o * kMzH + k];
        host[off[1] + o] = h_params[1][o];
        host[off[3] + o] = h_params[3][o];
    }
    if (hipDeviceSynchronize() != hipSuccess) return mz_fail(RZ_ERR_HIP, "hipDeviceSynchronize failed");
    if (e->d_rep == nullptr) {
        if (hipMalloc((void **)&e->d_rep, total * sizeof(float)) != hipSuccess) return mz_fail(RZ_ERR_OOM, "hipMalloc failed (muzero representation)");
        e->allocs.push_back(e->d_rep);
        e->bytes += (long long)(total * sizeof(float));
    }
    if (hipMemcpy(e->d_rep, host.data(), total * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        return mz_fail(RZ_ERR_HIP, "hipMemcpy(representation) failed");
    e->model.rep1_w = e->d_rep + off[0];
    e->model.rep1_b = e->d_rep + off[1];
    e->model.rep2_w = e->d_rep + off[2];
    e->model.rep2_b = e->d_rep + off[3];
    e->representation_loaded = true;
    return RZ_OK;
}

static int mz_launch_search(rz_muzero *e, float *d_hidden, int32_t n_sims, const MzTrace &T, const MzPlay *play, void *stream) {
    // games per workgroup: a search is a latency chain per game and two workgroups share a CU without slowing each
    // other, so the 16 columns of a tile are filled first (profiles/r02/muzero_search_shape.txt)
    const int gpw = e->games_per_wg > 0 ? e->games_per_wg : kMzMaxGpw;
    // the trees go to LDS when two workgroups still fit on a CU
    const MzDev &D = e->dev;
    const bool tree_lds = mz_search_lds_bytes(gpw, D.cap, D.path_stride, D.n_sims, true) <= 80 * 1024;
    const int lds = mz_search_lds_bytes(gpw, D.cap, D.path_stride, D.n_sims, tree_lds);
    if (lds > 160 * 1024) return mz_fail(RZ_ERR_ARG, "n_sims too large for the fused search (paths do not fit in LDS)");
    const dim3 grid((unsigned)((e->cfg.n_games + gpw - 1) / gpw)), block(64 * kMzWaves);
    const MzPlay none = {};
    hipStream_t st = (hipStream_t)stream;
    if (play != nullptr) {
        if (tree_lds) k_mz_search<true, true><<<grid, block, lds, st>>>(e->dev, e->model, d_hidden, n_sims, gpw, T, *play);
        else k_mz_search<false, true><<<grid, block, lds, st>>>(e->dev, e->model, d_hidden, n_sims, gpw, T, *play);
    } else {
        if (tree_lds) k_mz_search<true, false><<<grid, block, lds, st>>>(e->dev, e->model, d_hidden, n_sims, gpw, T, none);
        else k_mz_search<false, false><<<grid, block, lds, st>>>(e->dev, e->model, d_hidden, n_sims, gpw, T, none);
    }
    return mz_launched("launch of k_mz_search failed");
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
    return mz_launch_search(e, d_hidden, n_sims, T, nullptr, stream);
}

int rz_mz_play_cartpole(rz_muzero *e, float *d_hidden, int32_t n_sims, int32_t n_moves, const rz_mz_cartpole_play *play, void *stream) {
    int rc = mz_ready(e);
    if (rc != RZ_OK) return rc;
    if (!e->model_loaded || !e->representation_loaded) return mz_fail(RZ_ERR_ARG, "rz_mz_load_model and rz_mz_load_representation first");
    if (e->cfg.n_actions != 2) return mz_fail(RZ_ERR_ARG, "CartPole has 2 actions");
    if (!e->f16_ok)
        return mz_fail(RZ_ERR_ARG, "the model has non-finite weights: whole moves run their layers on the f16 matrix pipe and need finite "
                                   "bounds (use the move-by-move search, rz_mz_search / fused_moves=False)");
    if (d_hidden == nullptr || play == nullptr || n_sims < 1 || n_sims > e->cfg.n_sims || n_moves < 1)
        return mz_fail(RZ_ERR_ARG, "NULL pointer or n_sims / n_moves out of range");
    if (!play->d_state || !play->d_steps || !play->d_episode || !play->d_episode_start || !play->d_ring || !play->d_arena ||
        !play->d_counters || !play->d_entries)
        return mz_fail(RZ_ERR_ARG, "NULL environment / history pointer");
    if (play->ring_steps < 500 + n_moves || play->arena_rows < 1 || play->max_entries < (int64_t)e->cfg.n_games * n_moves || play->first_step < 0)
        return mz_fail(RZ_ERR_ARG, "ring_steps must cover an episode (500 steps) + the launch, max_entries one entry per environment and move");
    if (!(play->dirichlet_alpha > 0.0) || play->noise_frac < 0.0 || play->noise_frac > 1.0) return mz_fail(RZ_ERR_ARG, "dirichlet_alpha must be > 0, noise_frac in [0, 1]");
    MzPlay p = {};
    p.n_moves = n_moves;
    p.row = 4 + 4 + e->cfg.n_actions;
    p.state = play->d_state;
    p.steps = reinterpret_cast<long long *>(play->d_steps);
    p.episode = reinterpret_cast<long long *>(play->d_episode);
    p.env_seed = play->env_seed;
    p.noise_seed = play->noise_seed;
    p.noise_frac = play->noise_frac;
    p.inv_temperature = play->temperature > 0.0 ? 1.0 / play->temperature : 0.0;
    p.alpha = (float)play->dirichlet_alpha;
    p.ring = play->d_ring;
    p.hist = play->ring_steps;
    p.t0 = play->first_step;
    p.ep_start = reinterpret_cast<long long *>(play->d_episode_start);
    p.arena = play->d_arena;
    p.arena_rows = play->arena_rows;
    p.counters = reinterpret_cast<long long *>(play->d_counters);
    p.entries = reinterpret_cast<long long *>(play->d_entries);
    p.max_entries = play->max_entries;
    const MzTrace T = {};
    return mz_launch_search(e, d_hidden, n_sims, T, &p, stream);
}

int rz_cartpole_step(double *d_state, int64_t *d_steps, int64_t *d_episode, const int64_t *d_actions, int32_t n_envs, uint64_t seed,
                     float *d_obs, float *d_reward, uint8_t *d_terminated, uint8_t *d_truncated, void *stream) {
    if (!d_state || !d_steps || !d_episode || !d_actions || !d_obs || !d_reward || !d_terminated || !d_truncated || n_envs < 1)
        return mz_fail(RZ_ERR_ARG, "NULL pointer or n_envs < 1");
    k_cartpole_step<<<dim3((unsigned)((n_envs + 127) / 128)), dim3(128), 0, (hipStream_t)stream>>>(
        d_state, reinterpret_cast<long long *>(d_steps), reinterpret_cast<long long *>(d_episode),
        reinterpret_cast<const long long *>(d_actions), n_envs, seed, d_obs, d_reward, d_terminated, d_truncated);
    return mz_launched("launch of k_cartpole_step failed");
}

int rz_mz_set_search_shape(rz_muzero *e, int32_t games_per_workgroup) {
    if (e == nullptr) return mz_fail(RZ_ERR_ARG, "muzero handle is NULL");
    if (games_per_workgroup < 0 || games_per_workgroup > kMzMaxGpw) return mz_fail(RZ_ERR_ARG, "games_per_workgroup must be 0 (auto) .. 16");
    e->games_per_wg = games_per_workgroup;
    return RZ_OK;
}

#ifdef RZ_MZ_PROFILE
int rz_mz_debug_profile(long long *h_out16) {
    return hipDeviceSynchronize() == hipSuccess && hipMemcpyFromSymbol(h_out16, HIP_SYMBOL(mz_prof), 16 * sizeof(long long)) == hipSuccess ? RZ_OK : RZ_ERR_HIP;
}
#endif

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
