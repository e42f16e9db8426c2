// rz_net.hip -- fused fp32 forward of the AlphaZero policy-value network on MI355X (gfx950).
//
// Replaces PolicyValueNet.forward (rlzero/games/gomoku/policy_value_net.py:34-52) for the
// batch of MCTS leaves: the one dense contraction of the path (42.8 MFLOP per
// 15x15 position, MFMA-bound).  Exact fp32: v_mfma_f32_16x16x4_f32 is a k-ordered fmaf
// chain, no reduced precision anywhere (tolerance vs the reference's CPU output: 1e-4).
//
// Kernel A (k_trunk): ONE workgroup (8 waves) per board.  The board's activations never
// leave the CU: input planes, conv1 output (32 ch) and conv2 output (64 ch) live in LDS as
// halo-padded planes [channel][18 rows][18 cols] (plane stride 336 floats = 16 mod 32 banks);
// conv3's 128 channels stay in the MFMA accumulators and are consumed by the two 1x1 head
// convolutions in registers.  Implicit GEMM per layer with M = output channels (A = weights,
// pre-packed on the host in fragment order, streamed from L2 as 16-byte loads), N = the 16
// columns of one board row (B = one ds_read_b32 per lane from the halo planes), K = (input
// channel group of 4, tap).  Wave w owns output-channel half (w & 1) and rows 4*(w>>1)..+3,
// i.e. TM x 4 accumulator tiles; the 18 distinct (row offset, dx) fragments of a channel
// group are read once and reused by all 9 taps x 4 rows.
// Kernel B (k_heads): the three fully connected layers, log_softmax and tanh for a tile of
// boards per workgroup.

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "rlzero_hip.h"

void rz_set_error(const char *msg);  // rz_engine.hip

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kRowW = 18;     // halo row width (x = -1 .. 16)
constexpr int kRows = 18;     // halo rows (y = -1 .. 16)
constexpr int kPlane = 336;   // 18*18 = 324 padded so that 4 planes apart hit other banks
constexpr int kPlanesIn = 4, kPlanesC1 = 32, kPlanesC2 = 64;
constexpr int kLdsFloats = (kPlanesIn + kPlanesC1 + kPlanesC2) * kPlane;  // 33600 floats = 131.25 KiB
constexpr int kTrunkThreads = 512;

struct NetDev {
    const f32x4 *w1, *w2, *w3;   // packed [tile][cin_step][3][64 lanes] x 4 taps
    const float *b1, *b2, *b3;   // conv biases
    const float *wh;             // [6][128]: act_conv1 (4 rows) then val_conv1 (2 rows)
    const float *bh;             // [6]
    const float *fc_act_t;       // act_fc1.weight transposed: [4S][S]
    const float *fc_act_b;       // [S]
    const float *fc_val1_t;      // val_fc1.weight transposed: [2S][64]
    const float *fc_val1_b;      // [64]
    const float *fc_val2_w;      // [64]
    const float *fc_val2_b;      // [1]
    int B, S;
};

// acc[m][t] += W(tile m) x in(rows 4*rg + t) over all input channels and taps.
template <int CIN, int TM>
__device__ __forceinline__ void conv_accumulate(const float *__restrict__ in,
                                                const f32x4 *__restrict__ wp, int half, int rg,
                                                int lane, f32x4 (&acc)[TM][4]) {
    constexpr int kSteps = CIN / 4;
    const int x = lane & 15, kq = lane >> 4;
    // lane's channel within a group is kq; fragments are in[(4s+kq)][4rg + ro][x + dxi]
    const float *base = in + kq * kPlane + (4 * rg) * kRowW + x;
    const f32x4 *wbase = wp + (size_t)(half * TM) * kSteps * 3 * 64 + lane;
#pragma unroll 1
    for (int s = 0; s < kSteps; ++s) {
        f32x4 a[TM][3];
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int tg = 0; tg < 3; ++tg) a[m][tg] = wbase[((size_t)(m * kSteps + s) * 3 + tg) * 64];
        float b[6][3];
        const float *p = base + (4 * s) * kPlane;
#pragma unroll
        for (int ro = 0; ro < 6; ++ro)
#pragma unroll
            for (int dxi = 0; dxi < 3; ++dxi) b[ro][dxi] = p[ro * kRowW + dxi];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dyi = tap / 3, dxi = tap % 3;
#pragma unroll
            for (int m = 0; m < TM; ++m) {
                const float av = a[m][tap / 4][tap % 4];
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[t + dyi][dxi], acc[m][t], 0, 0, 0);
            }
        }
    }
}

// out[cout][y+1][x+1] = relu(acc + bias[cout]) for the lane's 4 channels of every tile/row.
template <int TM>
__device__ __forceinline__ void store_relu(float *__restrict__ out, const float *__restrict__ bias,
                                           int half, int rg, int lane, int B, const f32x4 (&acc)[TM][4]) {
    const int x = lane & 15, q = lane >> 4;
    if (x >= B) return;
#pragma unroll
    for (int m = 0; m < TM; ++m) {
        const int c0 = (half * TM + m) * 16 + 4 * q;
        const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias + c0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int y = 4 * rg + t;
            if (y >= B) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                out[(c0 + j) * kPlane + (y + 1) * kRowW + (x + 1)] = fmaxf(acc[m][t][j] + bv[j], 0.0f);
        }
    }
}

template <int TM>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[TM][4]) {
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
}

__global__ __launch_bounds__(kTrunkThreads) void k_trunk(NetDev nd, const float *__restrict__ obs,
                                                         float *__restrict__ feat, int n_boards) {
    __shared__ __attribute__((aligned(16))) float lds[kLdsFloats];
    float *in0 = lds;
    float *c1 = in0 + kPlanesIn * kPlane;
    float *c2 = c1 + kPlanesC1 * kPlane;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = wave & 1, rg = wave >> 1;
    const int B = nd.B, S = nd.S;
    const int board = blockIdx.x;
    if (board >= n_boards) return;

    // zero the halo planes (interiors are overwritten below), then stage the observation
    {
        f32x4 *z = reinterpret_cast<f32x4 *>(lds);
        for (int i = tid; i < kLdsFloats / 4; i += kTrunkThreads) z[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    {
        const float *src = obs + (size_t)board * 4 * S;
        for (int i = tid; i < 4 * S; i += kTrunkThreads) {
            const int c = i / S, r = i - c * S, y = r / B, x = r - y * B;
            in0[c * kPlane + (y + 1) * kRowW + (x + 1)] = src[i];
        }
    }
    __syncthreads();
    const bool busy = 4 * rg < B;  // waves whose rows are all outside the board only hit barriers

    {   // conv1: 4 -> 32 (one 16-channel tile per half)
        f32x4 acc[1][4];
        zero_acc<1>(acc);
        if (busy) {
            conv_accumulate<4, 1>(in0, nd.w1, half, rg, lane, acc);
            store_relu<1>(c1, nd.b1, half, rg, lane, B, acc);
        }
    }
    __syncthreads();
    {   // conv2: 32 -> 64
        f32x4 acc[2][4];
        zero_acc<2>(acc);
        if (busy) {
            conv_accumulate<32, 2>(c1, nd.w2, half, rg, lane, acc);
            store_relu<2>(c2, nd.b2, half, rg, lane, B, acc);
        }
    }
    __syncthreads();
    // conv3: 64 -> 128, kept in registers and fed to the two 1x1 head convolutions
    float part[4][6];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int o = 0; o < 6; ++o) part[t][o] = 0.0f;
    if (busy) {
        f32x4 acc[4][4];
        zero_acc<4>(acc);
        conv_accumulate<64, 4>(c2, nd.w3, half, rg, lane, acc);
        const int q = lane >> 4;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int c0 = (half * 4 + m) * 16 + 4 * q;
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(nd.b3 + c0);
            f32x4 wv[6];
#pragma unroll
            for (int o = 0; o < 6; ++o) wv[o] = *reinterpret_cast<const f32x4 *>(nd.wh + o * 128 + c0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float h = fmaxf(acc[m][t][j] + bv[j], 0.0f);
#pragma unroll
                    for (int o = 0; o < 6; ++o) part[t][o] = fmaf(wv[o][j], h, part[t][o]);
                }
        }
    }
    // sum over the 4 channel quarters held by lanes x, x+16, x+32, x+48
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int o = 0; o < 6; ++o) {
            float v = part[t][o];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            part[t][o] = v;
        }
    // c1 is free now: partial[half][o][y][x]
    float *partial = c1;
    if (lane < 16) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int o = 0; o < 6; ++o)
                partial[((half * 6 + o) * 16 + (4 * rg + t)) * 16 + lane] = part[t][o];
    }
    __syncthreads();
    {
        float *dst = feat + (size_t)board * 6 * S;
        for (int i = tid; i < 6 * S; i += kTrunkThreads) {
            const int o = i / S, r = i - o * S, y = r / B, x = r - y * B;
            const float v = partial[((0 * 6 + o) * 16 + y) * 16 + x] + partial[((1 * 6 + o) * 16 + y) * 16 + x] +
                            nd.bh[o];
            dst[i] = fmaxf(v, 0.0f);
        }
    }
}

// ------------------------------------------------------------------ heads (FC layers)
// One workgroup per kHeadBoards boards; thread j < S accumulates logit j of every board of the
// tile (weights transposed to [k][j] so a wave reads contiguous rows and reuses each weight
// kHeadBoards times); threads then do log_softmax through LDS; the value head uses the first
// 64 threads.  feat: [n][6S] (policy features 4S then value features 2S).
constexpr int kHeadBoards = 8;
constexpr int kHeadThreads = 256;

__global__ __launch_bounds__(kHeadThreads) void k_heads(NetDev nd, const float *__restrict__ feat,
                                                        float *__restrict__ logp, float *__restrict__ value,
                                                        int n_boards) {
    __shared__ float sfeat[kHeadBoards][6 * 256];
    __shared__ float sred[kHeadBoards][kHeadThreads];
    __shared__ float shid[kHeadBoards][64];
    const int tid = threadIdx.x;
    const int S = nd.S;
    const int b0 = blockIdx.x * kHeadBoards;
    for (int i = tid; i < kHeadBoards * 6 * S; i += kHeadThreads) {
        const int bb = i / (6 * S), r = i - bb * 6 * S;
        sfeat[bb][r] = (b0 + bb < n_boards) ? feat[(size_t)(b0 + bb) * 6 * S + r] : 0.0f;
    }
    __syncthreads();
    float logit[kHeadBoards];
#pragma unroll
    for (int bb = 0; bb < kHeadBoards; ++bb) logit[bb] = 0.0f;
    if (tid < S) {
        const float *w = nd.fc_act_t + tid;
        for (int k = 0; k < 4 * S; ++k) {
            const float wv = w[(size_t)k * S];
#pragma unroll
            for (int bb = 0; bb < kHeadBoards; ++bb) logit[bb] = fmaf(sfeat[bb][k], wv, logit[bb]);
        }
        const float bias = nd.fc_act_b[tid];
#pragma unroll
        for (int bb = 0; bb < kHeadBoards; ++bb) logit[bb] += bias;
    }
    // log_softmax over j < S per board
#pragma unroll
    for (int bb = 0; bb < kHeadBoards; ++bb) sred[bb][tid] = tid < S ? logit[bb] : -INFINITY;
    __syncthreads();
    for (int off = kHeadThreads / 2; off >= 1; off >>= 1) {
        if (tid < off)
#pragma unroll
            for (int bb = 0; bb < kHeadBoards; ++bb) sred[bb][tid] = fmaxf(sred[bb][tid], sred[bb][tid + off]);
        __syncthreads();
    }
    float mx[kHeadBoards];
#pragma unroll
    for (int bb = 0; bb < kHeadBoards; ++bb) mx[bb] = sred[bb][0];
    __syncthreads();
#pragma unroll
    for (int bb = 0; bb < kHeadBoards; ++bb) sred[bb][tid] = tid < S ? expf(logit[bb] - mx[bb]) : 0.0f;
    __syncthreads();
    for (int off = kHeadThreads / 2; off >= 1; off >>= 1) {
        if (tid < off)
#pragma unroll
            for (int bb = 0; bb < kHeadBoards; ++bb) sred[bb][tid] += sred[bb][tid + off];
        __syncthreads();
    }
    if (tid < S) {
#pragma unroll
        for (int bb = 0; bb < kHeadBoards; ++bb)
            if (b0 + bb < n_boards)
                logp[(size_t)(b0 + bb) * S + tid] = logit[bb] - mx[bb] - logf(sred[bb][0]);
    }
    // value head: fc(2S -> 64) relu, fc(64 -> 1), tanh
    if (tid < 64) {
        float h[kHeadBoards];
#pragma unroll
        for (int bb = 0; bb < kHeadBoards; ++bb) h[bb] = 0.0f;
        const float *w = nd.fc_val1_t + tid;
        for (int k = 0; k < 2 * S; ++k) {
            const float wv = w[(size_t)k * 64];
#pragma unroll
            for (int bb = 0; bb < kHeadBoards; ++bb) h[bb] = fmaf(sfeat[bb][4 * S + k], wv, h[bb]);
        }
        const float bias = nd.fc_val1_b[tid], w2 = nd.fc_val2_w[tid];
#pragma unroll
        for (int bb = 0; bb < kHeadBoards; ++bb) shid[bb][tid] = fmaxf(h[bb] + bias, 0.0f) * w2;
    }
    __syncthreads();
    if (tid < kHeadBoards && b0 + tid < n_boards) {
        float s = 0.0f;
        for (int k = 0; k < 64; ++k) s += shid[tid][k];
        value[b0 + tid] = tanhf(s + nd.fc_val2_b[0]);
    }
}

}  // namespace

struct rz_net {
    int board_size = 0, device = 0;
    bool loaded = false;
    NetDev dev;
    std::vector<void *> allocs;
    float *d_feat = nullptr;
    long long feat_boards = 0;
};

namespace {

int net_fail(int code, const char *msg, const char *detail = "") {
    char buf[480];
    snprintf(buf, sizeof(buf), "%s%s", msg, detail);
    rz_set_error(buf);
    return code;
}

template <typename T>
int net_upload(rz_net *net, const std::vector<T> &host, const T **out) {
    void *p = nullptr;
    if (hipMalloc(&p, host.size() * sizeof(T)) != hipSuccess) return net_fail(RZ_ERR_OOM, "hipMalloc failed (net)");
    net->allocs.push_back(p);
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

}  // namespace

extern "C" {

int rz_net_create(int32_t board_size, int32_t device, rz_net **out) {
    if (out == nullptr) return net_fail(RZ_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (board_size < 1 || board_size > RZ_MAX_BOARD_SIZE) return net_fail(RZ_ERR_ARG, "board_size out of range");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device < 0 || device >= n_dev)
        return net_fail(RZ_ERR_ARG, "bad device ordinal");
    rz_net *net = new (std::nothrow) rz_net();
    if (!net) return net_fail(RZ_ERR_OOM, "host allocation failed");
    net->board_size = board_size;
    net->device = device;
    memset(&net->dev, 0, sizeof(net->dev));
    net->dev.B = board_size;
    net->dev.S = board_size * board_size;
    *out = net;
    return RZ_OK;
}

int rz_net_destroy(rz_net *net) {
    if (!net) return RZ_OK;
    (void)hipSetDevice(net->device);
    (void)hipDeviceSynchronize();
    for (void *p : net->allocs) (void)hipFree(p);
    if (net->d_feat) (void)hipFree(net->d_feat);
    delete net;
    return RZ_OK;
}

int rz_net_load(rz_net *net, const float *const *h_params, int32_t n_params) {
    if (!net || !h_params) return net_fail(RZ_ERR_ARG, "NULL argument");
    if (n_params != 16) return net_fail(RZ_ERR_ARG, "expected the 16 tensors of PolicyValueNet.state_dict()");
    for (int i = 0; i < 16; ++i)
        if (!h_params[i]) return net_fail(RZ_ERR_ARG, "a parameter pointer is NULL");
    if (hipSetDevice(net->device) != hipSuccess) return net_fail(RZ_ERR_HIP, "hipSetDevice failed");
    (void)hipDeviceSynchronize();
    for (void *p : net->allocs) (void)hipFree(p);
    net->allocs.clear();
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
    up_f(h_params[5], 128, &D.b3);
    {
        std::vector<float> wh(6 * 128), bh(6);
        memcpy(wh.data(), h_params[6], 4 * 128 * sizeof(float));
        memcpy(wh.data() + 4 * 128, h_params[10], 2 * 128 * sizeof(float));
        memcpy(bh.data(), h_params[7], 4 * sizeof(float));
        memcpy(bh.data() + 4, h_params[11], 2 * sizeof(float));
        if (rc == RZ_OK) rc = net_upload(net, wh, &D.wh);
        if (rc == RZ_OK) rc = net_upload(net, bh, &D.bh);
    }
    {
        std::vector<float> t((size_t)4 * S * S);
        for (int j = 0; j < S; ++j)
            for (int k = 0; k < 4 * S; ++k) t[(size_t)k * S + j] = h_params[8][(size_t)j * 4 * S + k];
        if (rc == RZ_OK) rc = net_upload(net, t, &D.fc_act_t);
        up_f(h_params[9], S, &D.fc_act_b);
    }
    {
        std::vector<float> t((size_t)2 * S * 64);
        for (int j = 0; j < 64; ++j)
            for (int k = 0; k < 2 * S; ++k) t[(size_t)k * 64 + j] = h_params[12][(size_t)j * 2 * S + k];
        if (rc == RZ_OK) rc = net_upload(net, t, &D.fc_val1_t);
        up_f(h_params[13], 64, &D.fc_val1_b);
    }
    up_f(h_params[14], 64, &D.fc_val2_w);
    up_f(h_params[15], 1, &D.fc_val2_b);
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
    net->d_feat = nullptr;
    net->feat_boards = 0;
    if (hipMalloc((void **)&net->d_feat, (size_t)max_boards * 6 * net->dev.S * sizeof(float)) != hipSuccess)
        return net_fail(RZ_ERR_OOM, "hipMalloc failed (feature buffer)");
    net->feat_boards = max_boards;
    return RZ_OK;
}

int rz_net_trunk(rz_net *net, const float *d_obs, int32_t n_boards, float *d_feat, void *stream) {
    int rc = net_ready(net, n_boards);
    if (rc != RZ_OK) return rc;
    if (!d_obs || !d_feat) return net_fail(RZ_ERR_ARG, "NULL device pointer");
    if (n_boards == 0) return RZ_OK;
    k_trunk<<<dim3((unsigned)n_boards), dim3(kTrunkThreads), 0, (hipStream_t)stream>>>(net->dev, d_obs, d_feat, n_boards);
    if (hipGetLastError() != hipSuccess) return net_fail(RZ_ERR_HIP, "launch of k_trunk failed");
    return RZ_OK;
}

int rz_net_forward(rz_net *net, const float *d_obs, int32_t n_boards, float *d_logp, float *d_value, void *stream) {
    int rc = net_ready(net, n_boards);
    if (rc != RZ_OK) return rc;
    if (!d_obs || !d_logp || !d_value) return net_fail(RZ_ERR_ARG, "NULL device pointer");
    if (n_boards == 0) return RZ_OK;
    if (n_boards > net->feat_boards)
        return net_fail(RZ_ERR_ARG, "batch larger than rz_net_reserve()d (no allocation on the launch path)");
    k_trunk<<<dim3((unsigned)n_boards), dim3(kTrunkThreads), 0, (hipStream_t)stream>>>(net->dev, d_obs, net->d_feat, n_boards);
    const unsigned blocks = (unsigned)((n_boards + kHeadBoards - 1) / kHeadBoards);
    k_heads<<<dim3(blocks), dim3(kHeadThreads), 0, (hipStream_t)stream>>>(net->dev, net->d_feat, d_logp, d_value, n_boards);
    if (hipGetLastError() != hipSuccess) return net_fail(RZ_ERR_HIP, "launch of k_trunk/k_heads failed");
    return RZ_OK;
}

}  // extern "C"
