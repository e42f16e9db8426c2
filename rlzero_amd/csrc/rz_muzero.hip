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
//   N int32, value_sum f64, reward f64, prior f64, first_child int32 (-1 = not expanded; the A
//   children of a node are the consecutive slots first_child .. first_child + A - 1).
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

struct MzDev {
    int n_games, n_actions, n_sims, cap, path_stride;
    double discount, pb_c_init;
    int32_t *N, *first_child, *top, *path, *depth, *err;
    double *value_sum, *reward, *prior, *vmin, *vmax;
    const double *pb_log;  // [n_sims + 2]: log((n + pb_c_base + 1) / pb_c_base)
};

// Node.value(): value_sum / visit_count, 0 for an unvisited node
__device__ __forceinline__ double node_value(const MzDev &E, long long slot) {
    const int n = E.N[slot];
    return n > 0 ? E.value_sum[slot] / (double)n : 0.0;
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
    E.N[base] = 0;
    E.value_sum[base] = 0.0;
    E.reward[base] = 0.0;
    E.prior[base] = 0.0;
    E.first_child[base] = 1;
    for (int a = 0; a < E.n_actions; ++a) {
        const long long c = base + 1 + a;
        double p = (double)probs[(long long)g * E.n_actions + a];
        if (noise != nullptr) p = p * (1.0 - frac) + noise[(long long)g * E.n_actions + a] * frac;
        E.N[c] = 0;
        E.value_sum[c] = 0.0;
        E.reward[c] = 0.0;
        E.prior[c] = p;
        E.first_child[c] = -1;
    }
    E.top[g] = 1 + E.n_actions;
    E.vmin[g] = INFINITY;   // MinMaxStats(): minimum = +MAX, maximum = -MAX
    E.vmax[g] = -INFINITY;
    E.depth[g] = 0;
}

// select_child down to the first unexpanded node; max() over (score, action): ties go to the LARGER action
__global__ void k_mz_select(MzDev E, int32_t *parent, int32_t *action, int32_t *leaf, const uint8_t *mask) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.n_games) return;
    if (mask != nullptr && !mask[g]) {
        parent[g] = 0;
        action[g] = 0;
        leaf[g] = 0;
        return;
    }
    const long long base = (long long)g * E.cap;
    int32_t *path = E.path + (long long)g * E.path_stride;
    const double lo = E.vmin[g], hi = E.vmax[g];
    int node = 0, depth = 0, last_action = 0, par = 0;
    path[0] = 0;
    while (E.first_child[base + node] >= 0 && depth + 1 < E.path_stride) {
        const int fc = E.first_child[base + node];
        const int pn = E.N[base + node];
        const double pb_c0 = E.pb_log[pn <= E.n_sims + 1 ? pn : E.n_sims + 1] + E.pb_c_init;
        const double sq = sqrt((double)pn);
        double best = -INFINITY;
        int besta = 0;
        for (int a = 0; a < E.n_actions; ++a) {
            const long long c = base + fc + a;
            const int cn = E.N[c];
            const double pb_c = pb_c0 * (sq / (double)(cn + 1));
            const double prior_score = pb_c * E.prior[c];
            double value_score = 0.0;
            if (cn > 0) value_score = normalize(E.reward[c] + E.discount * node_value(E, c), lo, hi);
            const double score = prior_score + value_score;
            if (score >= best) {  // the later (larger) action wins a tie
                best = score;
                besta = a;
            }
        }
        par = node;
        last_action = besta;
        node = fc + besta;
        depth += 1;
        path[depth] = node;
    }
    E.depth[g] = depth;
    parent[g] = par;
    action[g] = last_action;
    leaf[g] = node;
}

// expand_node(leaf, network_output) + backpropagate(search_path, value, discount, min_max_stats)
__global__ void k_mz_expand_backup(MzDev E, const float *reward, const float *probs, const float *value,
                                   const uint8_t *mask) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.n_games) return;
    if (mask != nullptr && !mask[g]) return;
    const long long base = (long long)g * E.cap;
    const int32_t *path = E.path + (long long)g * E.path_stride;
    const int depth = E.depth[g];
    const int leaf = path[depth];
    const int top = E.top[g];
    if (top + E.n_actions > E.cap) {
        atomicOr(E.err, RZ_FLAG_ARENA_FULL);
    } else {
        E.reward[base + leaf] = (double)reward[g];
        E.first_child[base + leaf] = top;
        for (int a = 0; a < E.n_actions; ++a) {
            const long long c = base + top + a;
            E.N[c] = 0;
            E.value_sum[c] = 0.0;
            E.reward[c] = 0.0;
            E.prior[c] = (double)probs[(long long)g * E.n_actions + a];
            E.first_child[c] = -1;
        }
        E.top[g] = top + E.n_actions;
    }
    double v = (double)value[g];
    double lo = E.vmin[g], hi = E.vmax[g];
    for (int d = depth; d >= 0; --d) {
        const long long slot = base + path[d];
        const double sum = E.value_sum[slot] + v;
        const int n = E.N[slot] + 1;
        E.value_sum[slot] = sum;
        E.N[slot] = n;
        const double nv = sum / (double)n;
        hi = nv > hi ? nv : hi;  // MinMaxStats.update
        lo = nv < lo ? nv : lo;
        v = E.reward[slot] + E.discount * v;
    }
    E.vmin[g] = lo;
    E.vmax[g] = hi;
}

// what: 0 = visit counts (int32 [G][A]), 1 = value sums (f64 [G][A]), 2 = rewards, 3 = priors of the root's children
__global__ void k_mz_root_children(MzDev E, int what, void *out) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.n_games) return;
    const long long base = (long long)g * E.cap;
    const int fc = E.first_child[base];
    for (int a = 0; a < E.n_actions; ++a) {
        const long long o = (long long)g * E.n_actions + a;
        const long long c = base + fc + a;
        if (what == 0) ((int32_t *)out)[o] = fc >= 0 ? E.N[c] : 0;
        else if (what == 1) ((double *)out)[o] = fc >= 0 ? E.value_sum[c] : 0.0;
        else if (what == 2) ((double *)out)[o] = fc >= 0 ? E.reward[c] : 0.0;
        else ((double *)out)[o] = fc >= 0 ? E.prior[c] : 0.0;
    }
}

__global__ void k_mz_root_stats(MzDev E, int32_t *n, double *value_sum, double *vmin, double *vmax) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= E.n_games) return;
    const long long base = (long long)g * E.cap;
    if (n) n[g] = E.N[base];
    if (value_sum) value_sum[g] = E.value_sum[base];
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
    MZ_ALLOC(N, slots);
    MZ_ALLOC(first_child, slots);
    MZ_ALLOC(value_sum, slots);
    MZ_ALLOC(reward, slots);
    MZ_ALLOC(prior, slots);
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
            hipMemset(D.err, 0, 4) != hipSuccess || hipMemset(D.first_child, 0xff, (size_t)slots * 4) != hipSuccess ||
            hipMemset(D.top, 0, (size_t)G * 4) != hipSuccess || hipMemset(D.depth, 0, (size_t)G * 4) != hipSuccess ||
            hipMemset(D.N, 0, (size_t)slots * 4) != hipSuccess)
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
