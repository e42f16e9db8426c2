p='csrc/rz_tree.h'
s=open(p).read()
def rep(a,b,cnt=1):
    global s
    assert s.count(a)==cnt,(s.count(a),a[:70])
    s=s.replace(a,b)
rep("""namespace rzt {
namespace {
""","""namespace rzt {
namespace {
__device__ long long tree_prof[16];
#define TREE_TICK(i) do { __builtin_amdgcn_sched_barrier(0); const long long now_ = __builtin_readcyclecounter(); if (g == 0 && lane == 0) tree_prof[i] += now_ - tp_t; tp_t = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
""")
rep("""    const int gk = VL ? g * E.K + j : g;
    // every load that does not depend on another one is issued before `active` is tested: a kernel of dependent
    // round trips (an inactive game's slots exist, reading them is harmless)
    const int arena = E.cur_arena[g];""","""    const int gk = VL ? g * E.K + j : g;
    long long tp_t = __builtin_readcyclecounter();
    const int arena = E.cur_arena[g];""")
rep("""    int4 lo = R[0], hi = R[1];  // the record of `node`: loaded for the root, broadcast by the scans below
    for (int it = 0; it <= S; ++it) {
        const int k = rec_k(lo);
        if (k == 0) break;  // leaf: never expanded, or a terminal position""","""    int4 lo = R[0], hi = R[1];  // the record of `node`: loaded for the root, broadcast by the scans below
    for (int it = 0; it <= S; ++it) {
        const int k = rec_k(lo);
        if (it == 0) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); TREE_TICK(0); }
        if (k == 0) break;  // leaf: never expanded, or a terminal position""")
rep("""        if (r >= k) {
            flag(E, g, RZ_FLAG_INTERNAL, lane);
            break;
        }
        if (VL && !fresh && lane == 0) {  // virtual loss on an inner node of the path""","""        if (r >= k) {
            flag(E, g, RZ_FLAG_INTERNAL, lane);
            break;
        }
        if (fresh) TREE_TICK(1); else TREE_TICK(2);
        if (VL && !fresh && lane == 0) {  // virtual loss on an inner node of the path""")
rep("""        const Legal L = legal_of<W>(E, occ, lane);
        int action, cell;""","""        const Legal L = legal_of<W>(E, occ, lane);
        TREE_TICK(6);
        int action, cell;""")
rep("""        if (lane == 0) path[depth] = node;
        if (fresh) break;  // a first-visit child has no statistics and no children yet""","""        if (lane == 0) path[depth] = node;
        TREE_TICK(3);
        if (g == 0 && lane == 0) tree_prof[8] += 1;
        if (fresh) break;  // a first-visit child has no statistics and no children yet""")
rep("""    // game_end_winner on the leaf (gomoku_env.py:196-203)
    int term = 0;""","""    TREE_TICK(7);
    // game_end_winner on the leaf (gomoku_env.py:196-203)
    int term = 0;""")
rep("""    if (lane == 0) {
        E.leaf_node[gk] = node;
        E.leaf_depth[gk] = depth;""","""    TREE_TICK(4);
    if (lane == 0) {
        E.leaf_node[gk] = node;
        E.leaf_depth[gk] = depth;""")
rep("""    if (obs != nullptr)
        write_obs(obs + (long long)gk * 4 * S, to_move == 0 ? st[0] : st[1],
                  to_move == 0 ? st[1] : st[0], last, nst, S, lane);
}""","""    if (obs != nullptr)
        write_obs(obs + (long long)gk * 4 * S, to_move == 0 ? st[0] : st[1],
                  to_move == 0 ? st[1] : st[0], last, nst, S, lane);
    TREE_TICK(5);
    if (g == 0 && lane == 0) tree_prof[9] += 1;
}""")
rep("""    const int gk = VL ? g * E.K + j : g;
    const int slot = DEF ? E.pend[g] : 0;""","""    const int gk = VL ? g * E.K + j : g;
    long long tp_t = __builtin_readcyclecounter();
    const int slot = DEF ? E.pend[g] : 0;""")
rep("""    if (!act) return;
    int4 *R = arena_records(E, g, arena);
    float *P = arena_priors(E, g, arena);

    // the reference evaluates terminal leaves too""","""    if (!act) return;
    TREE_TICK(10);
    int4 *R = arena_records(E, g, arena);
    float *P = arena_priors(E, g, arena);

    // the reference evaluates terminal leaves too""")
rep("""    if (DEF && lane == 0) {
        if (new_pb < 0) E.pend_pb[rec] = -1;   // nothing expanded in this step (a terminal leaf, a full arena)
        E.pend[g] = slot + 1;
    }
""","""    if (DEF && lane == 0) {
        if (new_pb < 0) E.pend_pb[rec] = -1;   // nothing expanded in this step (a terminal leaf, a full arena)
        E.pend[g] = slot + 1;
    }
    TREE_TICK(11);
""")
rep("""            *rec_n(R, node) += 1;
            *rec_wsum(R, node) += x;
        }
    }
}""","""            *rec_n(R, node) += 1;
            *rec_wsum(R, node) += x;
        }
    }
    TREE_TICK(12);
}""")
open(p,'w').write(s)
p='csrc/rz_net.hip'
s=open(p).read()
a="""    return hipDeviceSynchronize() == hipSuccess && hipMemcpyFromSymbol(h_out16, HIP_SYMBOL(net_prof), 24 * sizeof(long long)) == hipSuccess ? RZ_OK : RZ_ERR_HIP;"""
b="""    return hipDeviceSynchronize() == hipSuccess && hipMemcpyFromSymbol(h_out16, HIP_SYMBOL(net_prof), 24 * sizeof(long long)) == hipSuccess
           && hipMemcpyFromSymbol(h_out16 + 24, HIP_SYMBOL(rzt::tree_prof), 16 * sizeof(long long)) == hipSuccess ? RZ_OK : RZ_ERR_HIP;"""
assert s.count(a)==1
s=s.replace(a,b)
open(p,'w').write(s)
