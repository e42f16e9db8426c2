/*
 * rlzero_hip.h -- C ABI of the MI355X (gfx950) AlphaZero self-play MCTS engine.
 *
 * This is the drop-in boundary for ONE path of jianzhnie/RLZero: the
 * select -> expand -> evaluate -> backup loop of AlphaZero MCTS plus the Gomoku /
 * TicTacToe board rules it steps through.  The reference is pure Python and has no FFI;
 * each entry point below names the reference function(s) (file:line, relative to the
 * reference repository) whose work it takes over for a whole batch of lock-stepped games.
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *  - plain C, no torch types.  `d_` pointers are DEVICE pointers (e.g. tensor.data_ptr()),
 *    `h_` pointers are HOST pointers.  `stream` is a hipStream_t passed as void*.
 *  - every function returns 0 (RZ_OK) or a negative RZ_ERR_* code; the message is
 *    available from rz_last_error() (thread local).  No exception crosses the ABI.
 *  - an engine is NOT thread safe: one host thread per engine (the reference is
 *    single-threaded).  Launch functions enqueue on `stream` and do
 *    not synchronise, allocate or free: they can be captured in a hipGraph.
 *  - one tree per game, ONE simulation in flight per tree (the reference runs
 *    simulations strictly sequentially, rlzero/mcts/alphazero_mcts.py:82-85); the
 *    parallelism is across games.
 *
 * Boards: two bitboards per game, RZ_BOARD_WORDS u64 words per colour, bit `a` = move
 * `a` = h*B + w (rlzero/games/gomoku/gomoku_env.py:227-234).  Layout of a board array:
 * [n_games][2][RZ_BOARD_WORDS] (colour 0 = player id 0 = first player).
 */
#ifndef RLZERO_HIP_H
#define RLZERO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RZ_ABI_VERSION 26
#define RZ_MAX_BOARD_SIZE 16
#define RZ_BOARD_WORDS 4 /* 4 x 64 bits >= 16*16 cells */
#define RZ_MAX_IN_FLIGHT 16 /* rz_config.sims_in_flight */

enum {
    RZ_OK = 0,
    RZ_ERR_ARG = -1,      /* bad argument / config */
    RZ_ERR_HIP = -2,      /* a HIP runtime call failed */
    RZ_ERR_OOM = -3,      /* device allocation failed */
    RZ_ERR_OVERFLOW = -4, /* a game ran out of arena slots / block-queue entries */
    RZ_ERR_ILLEGAL = -5,  /* an illegal move was submitted (gomoku_env.py:51-53 asserts) */
    RZ_ERR_INTERNAL = -6
};

enum {
    RZ_GAME_GOMOKU = 0,  /* TicTacToe = Gomoku(board_size 3, n_in_row 3), tools/play.py:35; action = cell */
    RZ_GAME_CONNECT4 = 1 /* no implementation in the reference (docs/open-spiel_alphazero.md:58 only names
                            it): build-defined.  board_height x board_width (default 6 x 7), n_in_row
                            (default 4), action = column, stones drop to the lowest empty cell, cell =
                            row*width + column with row 0 at the bottom; same planes / search / API */
};

enum {
    RZ_SCORE_UCT_REF = 0, /* the reference's rule: W/N + c*sqrt(ln(Np)/N), +inf if unvisited
                             (rlzero/mcts/node.py:41-42,75-88); bit-exact, fp64 */
    RZ_SCORE_PUCT = 1     /* opt-in: Q + c*(P*sqrt(Np)/(N+1)) (node.py:105-117 is dead code in the
                             reference and divides by zero at N=0; here Q=0 at N=0); fp64, every
                             child initialised at expansion */
};

enum { RZ_EVAL_V0 = 0, RZ_EVAL_VLIN = 1 }; /* synthetic evaluators, SURVEY.md Appendix B */

/* per-game error bits reported by rz_get_stats */
enum {
    RZ_FLAG_ARENA_FULL = 1,
    RZ_FLAG_BLOCKS_FULL = 2,
    RZ_FLAG_ILLEGAL_MOVE = 4,
    RZ_FLAG_LOGTAB = 8,
    RZ_FLAG_INTERNAL = 16,
    RZ_FLAG_REUSE_DROPPED = 32 /* NOT an error: rz_advance_roots found the kept subtree larger than pool_factor * n_playout
                                  expanded nodes and restarted that game's search from a fresh root (the reference's tree
                                  is unbounded, alphazero_mcts.py:96-103); counted in rz_stats.reuse_dropped */
};

typedef struct rz_engine rz_engine;

typedef struct rz_config {
    int32_t abi_version; /* RZ_ABI_VERSION */
    int32_t game_kind;   /* RZ_GAME_GOMOKU */
    int32_t board_size;  /* B <= RZ_MAX_BOARD_SIZE (GomokuEnv(board_size=), gomoku_env.py:19-31) */
    int32_t n_in_row;    /* GomokuEnv(n_in_row=) */
    int32_t n_games;     /* games searched in lock-step on this GPU */
    int32_t n_playout;   /* simulations per move: sizes the arenas and the ln table
                            (AlphaZeroPlayer(n_playout=), alphazero_mcts.py:112-130) */
    int32_t score_mode;  /* RZ_SCORE_* */
    int32_t add_noise;   /* != 0: priors are mixed 0.75/0.25 with Dirichlet(0.3) noise at every expanded
                            node (node.py:63-69; is_selfplay in alphazero_mcts.py:124-129).  Drawn on the
                            device from a counter-based stream (noise_seed, game, expansion #; rz_set_noise_keys): same
                            distribution as numpy's, not its global stream.  RZ_SCORE_UCT_REF never reads
                            the prior, so the noise cannot change its search. */
    double c_puct;       /* AlphaZeroPlayer(c_puct=) */
    double pool_factor;  /* arena slots per game = pool_factor*n_playout*B*B + B*B + 2;
                            0 -> 2.0 */
    int32_t device;      /* HIP device ordinal */
    int32_t noise_seed;  /* seed of the Dirichlet stream */
    int32_t board_height; /* RZ_GAME_CONNECT4 only (0 -> 6) */
    int32_t board_width;  /* RZ_GAME_CONNECT4 only (0 -> 7) */
    int32_t sims_in_flight; /* 0 / 1 (default): ONE simulation in flight per tree, the reference's sequential search
                               (alphazero_mcts.py:82-85) -- the only mode the parity tests use.  K > 1 (opt-in, NOT
                               the reference's algorithm): K simulations of a tree share one evaluator batch; every node
                               of a selected path carries a virtual loss (N += 1, W -= 1) until its backup.  Leaf
                               arrays of this ABI (d_obs, d_logp, d_value, d_raw, d_hid) then have n_games * K rows,
                               row = game * K + slot.  Device evaluators only. */
    int32_t in_flight_impl; /* sims_in_flight > 1: 0 = the level-synchronous kernel (a workgroup of K waves per game: child
                               records of all slots' nodes staged per level, slots walked in slot order), 1 = its sequential
                               restatement (one wave, one slot after the other); same trees, bit for bit */
} rz_config;

typedef struct rz_stats {
    int32_t error_flags;     /* OR of RZ_FLAG_* over all games since the last clear */
    int32_t first_bad_game;  /* lowest game index with a flag, or -1 */
    int64_t arena_slots;     /* node-record capacity per game per arena */
    int64_t prior_floats;    /* prior-block capacity (floats) per game per arena */
    int64_t max_slots_used;  /* max over games of the current arena top */
    int64_t max_blocks_used; /* max over games of expanded nodes in the current arena */
    int64_t device_bytes;    /* bytes of HBM the engine allocated */
    int64_t n_select_calls;  /* rz_select_step launches so far */
    int64_t reuse_dropped;   /* kept subtrees dropped by rz_advance_roots since the last clear (RZ_FLAG_REUSE_DROPPED) */
} rz_stats;

int rz_abi_version(void);
/* The hash of the sources, headers and flags this library was built from (rlzero_amd/_build.py): the binding refuses a library
 * whose hash differs from the source tree beside it, and build() recompiles on a hash mismatch, not on file times. */
const char *rz_source_hash(void);
const char *rz_last_error(void);

/* Lifetime.  rz_create allocates every buffer up front (nothing is allocated later). */
int rz_create(const rz_config *cfg, rz_engine **out);
int rz_destroy(rz_engine *e);
/* Board rows, columns and size A of the action space (Gomoku: A = cells; Connect4: A = columns).
 * Every per-action array of this ABI (d_logp, d_probs, d_visits, d_w, d_p) is [n_games][A]. */
int rz_geometry(rz_engine *e, int32_t *height, int32_t *width, int32_t *n_actions);

/* ln(n) table for n = 0 .. count-1 (entry 0 unused).  The engine fills it at creation
 * with the host libm's log(), the function CPython's math.log() calls in
 * rlzero/mcts/node.py:84-85; the device never evaluates a logarithm itself.  This entry
 * replaces the table (e.g. with one produced by math.log on another host). */
int rz_upload_log_table(rz_engine *e, const double *h_table, int64_t count);
int rz_log_table_size(rz_engine *e, int64_t *count);

/* Root positions.  Import boards for the games selected by d_mask (NULL = all games):
 * what AlphaZeroPlayer.get_action reads from `game_env` (alphazero_mcts.py:136-146:
 * states, current player, last_move).  reset_trees != 0 also discards those games'
 * trees (AlphaZeroMCTS.update_with_move(-1), alphazero_mcts.py:96-103). */
int rz_set_roots(rz_engine *e, const uint64_t *d_stones, const int32_t *d_to_move,
                 const int32_t *d_last_move, const uint8_t *d_mask, int reset_trees,
                 void *stream);
int rz_get_roots(rz_engine *e, uint64_t *d_stones, int32_t *d_to_move, int32_t *d_last_move,
                 void *stream);
/* Games with active == 0 are skipped by select / expand_backup (finished games). */
int rz_set_active(rz_engine *e, const uint8_t *d_active, void *stream);
/* The Dirichlet noise of game g (rz_config.add_noise; node.py:63-69) is drawn from a counter-based stream keyed (key[g], expansion #).
 * By default key[g] = noise_seed ^ g << 20: a stream per SLOT of this engine.  A host that deals games to slots, lanes and GPUs
 * (rlzero_amd.selfplay) gives every game a key of its own when it starts -- d_keys uint64 [n_games], d_mask uint8 [n_games] or NULL
 * (all) selects the slots; their expansion counters restart at 0 -- so that a game's noise, like its move draws, depends on (seed,
 * game id) and not on where the game is played.  d_keys == NULL restores the default keys.  (The reference draws from numpy's
 * global stream, node.py:65: one stream for everything, in the order the single process happens to expand nodes.) */
int rz_set_noise_keys(rz_engine *e, const uint64_t *d_keys, const uint8_t *d_mask, void *stream);

/* SELECT + STEP: for every active game walk from the root to a leaf
 * (AlphaZeroMCTS._playout select loop, alphazero_mcts.py:48-54; TreeNode.select /
 * uct_value, node.py:32-42,75-88), applying each chosen move to a private copy of the
 * board (the reference's copy.deepcopy + GomokuEnv.step, alphazero_mcts.py:83,
 * gomoku_env.py:49-70), then classify the leaf (GomokuEnv.game_end_winner,
 * gomoku_env.py:196-203 -> has_a_winner :116-170).  If d_obs != NULL also writes the
 * leaf's observation planes float32 [n_games][4][B][B] (GomokuEnv.current_state,
 * gomoku_env.py:95-114) -- the input of the evaluator. */
int rz_select_step(rz_engine *e, float *d_obs, void *stream);
/* sims_in_flight > 1 only: how many of the K slots the NEXT launches back up (rz_expand_backup*, rz_tree_step*) and
 * select (rz_select_step, rz_tree_step*); default K, K.  A search of n simulations is ceil(n / K) steps, the last one
 * with the remainder, so that N(root) grows by exactly n. */
int rz_set_in_flight(rz_engine *e, int32_t k_backup, int32_t k_select);

/* Device pointers to the engine's leaf arrays, valid for its lifetime: stones uint64 [n_games * K][2][4], side to move and
 * last cell int32 [n_games * K] of the leaves of the last rz_select_step / rz_tree_step (row = game * K + slot).  An
 * evaluator that reads positions (rz_net_trunk_leaves) needs no observation planes: pass d_obs = NULL to the select calls. */
int rz_leaf_buffers(rz_engine *e, const uint64_t **d_stones, const int32_t **d_to_move, const int32_t **d_last_cell);

/* Observation planes of the current leaves / of the root positions (current_state). */
int rz_encode_leaf_obs(rz_engine *e, float *d_obs, void *stream);
int rz_encode_root_obs(rz_engine *e, float *d_obs, void *stream);

/* Leaf positions for an evaluator that runs on the host (any policy_value_fn callable,
 * alphazero_mcts.py:28-31,59).  d_terminal: 0 = not ended, 1 = tie, 2 = won. */
int rz_get_leaves(rz_engine *e, uint64_t *d_stones, int32_t *d_to_move, int32_t *d_last_move,
                  int32_t *d_terminal, void *stream);

/* Synthetic evaluators of SURVEY.md Appendix B computed from the leaf bitboards:
 * d_value float32 [n_games]; d_logp (nullable) float32 [n_games][B*B] = log(1/k) on
 * empty cells, -inf elsewhere. */
int rz_eval_synthetic(rz_engine *e, int kind, float *d_logp, float *d_value, void *stream);

/* Random-rollout evaluator of the pure-MCTS opponent: RolloutMCTS._evaluate
 * (rlzero/mcts/rollout_mcts.py:49-74, rollout_policy :96-100) from every leaf, on bitboards.
 * Move of ply p = the floor(u*k)-th legal move, u from splitmix64(seed, game, sim_index, p)
 * (the reference draws k uniforms from numpy's global stream and takes the arg-max: also a
 * uniform choice).  d_value float32 [n_games] follows the reference's perspective rule (:68-72).
 * Pair with rz_expand_backup(e, NULL, d_value): uniform priors (rollout_mcts.py:102-108). */
int rz_eval_rollout(rz_engine *e, uint64_t seed, uint32_t sim_index, int32_t n_limit, float *d_value,
                    void *stream);

/* EXPAND + BACKUP: terminal rule and value (alphazero_mcts.py:60-68), TreeNode.expand
 * (node.py:44-73; priors = exp(d_logp) on the leaf's legal moves as in
 * AlphaZeroAgent.policy_value_fn, alphazero_agent.py:41-45; d_logp == NULL -> uniform),
 * TreeNode.update_recursive(-leaf_value) (node.py:119-144).  d_value: leaf value from
 * the evaluator, [n_games], float32 (the reference converts the fp32 net output to a
 * Python float exactly, alphazero_agent.py:45) or float64 for host evaluators. */
int rz_expand_backup(rz_engine *e, const float *d_logp, const float *d_value, void *stream);
int rz_expand_backup_f64(rz_engine *e, const float *d_logp, const double *d_value, void *stream);
/* same, but d_probs float32 [n_games][B*B] holds the evaluator's probabilities themselves (a host
 * policy_value_fn returns (action, prob) pairs, alphazero_mcts.py:28-31): stored unchanged. */
int rz_expand_backup_probs(rz_engine *e, const float *d_probs, const double *d_value, void *stream);

/* rz_expand_backup of the pending leaves followed by rz_select_step of the next simulation in
 * ONE launch (two consecutive iterations of the reference's loop, alphazero_mcts.py:82-85,
 * share a kernel boundary).  Same arguments as the two calls it replaces. */
int rz_tree_step(rz_engine *e, const float *d_logp, const float *d_value, float *d_obs, void *stream);

/* The evaluator's UN-NORMALISED head outputs, as rz_net_heads_gemm leaves them: policy logits raw [rows][ld] (ld >= A)
 * and the value head's hidden layer hid [rows][64] with its last weights w2 [64], b2 [1].  n_parts == 1: raw / hid are
 * final (bias added, hid ReLU'd).  n_parts == 4 (RZ_NET_HEADS_SPLIT_PARTS): raw / hid hold the four K-quarter partial
 * sums of the FC GEMM, part q at raw + q * raw_part_stride / hid + q * hid_part_stride (floats), and the consumer
 * finishes logit = ((p0 + p1) + p2) + p3 -> fmaf(sum, *act_scale, act_bias[a]), hidden unit = relu(fmaf(sum,
 * *val_scale, val_bias[u])) -- the very operations, in the very order, of the GEMM kernels that reduce the parts
 * themselves, so every route gives the same bits. */
typedef struct rz_raw_heads {
    const float *raw;
    const float *hid;
    const float *w2;
    const float *b2;
    const float *act_scale; /* [1], n_parts == 4 only */
    const float *act_bias;  /* [ld] */
    const float *val_scale; /* [1] */
    const float *val_bias;  /* [64] */
    int64_t raw_part_stride;
    int64_t hid_part_stride;
    int32_t ld;
    int32_t n_parts;
} rz_raw_heads;

/* rz_expand_backup / rz_tree_step fed with those outputs: log_softmax and tanh(hid . w2 + b2) are finished inside the
 * tree kernel (same wave per game), saving the separate rz_net_heads finishing launch.  Pair with
 * rz_net_trunk(.., NULL, ..) + rz_net_heads_gemm.  Bit-identical to the un-fused route. */
int rz_expand_backup_raw(rz_engine *e, const rz_raw_heads *heads, void *stream);
int rz_tree_step_raw(rz_engine *e, const rz_raw_heads *heads, float *d_obs, void *stream);

/* ---------------------------------------------------------------------------------------
 * DEFERRED PRIORS (RZ_SCORE_UCT_REF, one simulation in flight per tree).  The reference's selection rule
 * (node.py:32-42,75-88) never reads TreeNode.prior: what the NEXT simulation of a tree depends on is the leaf VALUE
 * alone (alphazero_mcts.py:59-71).  So the policy half of policy_value_fn (alphazero_agent.py:41-45: act_fc1,
 * log_softmax, exp) and TreeNode.expand's priors with their Dirichlet noise (node.py:44-73) need not sit between two
 * simulations: the tree step of this route finishes the value head, reserves the expanded node's prior block (same
 * offsets as the other routes), backs up and selects; the leaf's policy features wait in a per-step store and ALL
 * priors of a search are written by one batched GEMM + one kernel before anything reads them (tree reuse, read-outs).
 * Every number stored is the one the other routes store, except that the value head's first layer is summed in f32 by
 * the game's own workgroup (k ascending in eight slices) instead of by the FC GEMM: values agree to f32 rounding,
 * not bit for bit.  A step of a lane is TWO launches (trunk, tree step) instead of three.
 *
 * rz_value_head: what the trunk of this route leaves for the tree step (rz_net_trunk_leaves_deferred). */
typedef struct rz_value_head {
    const float *valfeat; /* [rows][ld]: ReLU'd outputs of val_conv1, plane-major (2 x S), zero padded to ld */
    const float *w1t;     /* val_fc1.weight as [groups][64 hidden units][4 inputs] (inputs zero padded to 4 * groups) */
    const float *b1;      /* [64] val_fc1.bias */
    const float *w2;      /* [64] val_fc2.weight */
    const float *b2;      /* [1] */
    int32_t ld;           /* = 4 * groups */
    int32_t groups;       /* 16, 32, 64 or 128: four waves of the game's workgroup x 2 halves x 2, 4, 8 or 16 groups; 4 * groups
                           * >= 2 * S (refused otherwise): the tree step also sizes its bitboard arithmetic by it -- a head of
                           * 16 / 32 groups means a board of at most 64 cells, one 64-bit word per colour */
} rz_value_head;
/* rz_deferred_logits: the policy logits of the stored leaves after rz_net_deferred_gemm: row = slot * rows_per_slot + leaf */
typedef struct rz_deferred_logits {
    const float *raw; /* [n_slots * rows_per_slot][ld], final (scaled, bias added): what k_heads_split leaves */
    int32_t ld;
    int32_t rows_per_slot;
} rz_deferred_logits;
/* Room for `slots` steps between two flushes (per game: a pending-expansion record of 80 bytes per slot).  Needs
 * RZ_SCORE_UCT_REF and sims_in_flight == 1. */
int rz_deferred_reserve(rz_engine *e, int32_t slots);
/* Opt-in device-side launch trace (rlzero_amd/csrc/rz_trace.h; rlzero_amd/trace.py): d_trace = uint64 [2 + 8 * slots * n_games] of
 * THIS engine's lane with [0] = slots (steps of a search) and [1] = n_games filled in, or NULL to detach.  The deferred route's
 * kernels launched AFTERWARDS (a hipGraph keeps what it captured) leave one record per workgroup: start / end on the 100 MHz
 * clock, step, block, CU; a search overwrites the records of the one before it.  A diagnostic: traced instantiations of the
 * kernels run, not the production ones. */
int rz_trace_attach(rz_engine *e, void *d_trace);   /* the evaluator's half: rz_net_trace_attach below */
/* The engine's device view (pointers and geometry the tree code reads: rlzero_amd/csrc/rz_tree.h, struct Dev) for kernels of the
 * SAME build that run tree code outside the engine's own launches -- rz_net_search_resident.  out_bytes must be that build's
 * sizeof(Dev); valid until rz_destroy / the next rz_deferred_reserve. */
int rz_device_view(rz_engine *e, void *out, int64_t out_bytes);
/* int32 [n_games]: the store slot the NEXT leaf of each game goes to (= steps since the last flush); the trunk reads it */
int rz_deferred_slots(rz_engine *e, const int32_t **d_slot_of_game);
/* rz_expand_backup / rz_tree_step of this route (pair with rz_select_step(e, NULL, ..) + rz_net_trunk_leaves_deferred) */
int rz_expand_backup_deferred(rz_engine *e, const rz_value_head *head, void *stream);
int rz_tree_step_deferred(rz_engine *e, const rz_value_head *head, void *stream);
/* Writes the priors of every expansion pending in slots [0, n_slots) -- exp(log_softmax) over the leaf's legal moves mixed
 * with its Dirichlet noise: the operations of rz_expand_backup_raw -- and empties the slots.  Must run before
 * rz_advance_roots / rz_set_roots / the prior read-outs; without pending slots a no-op. */
int rz_deferred_flush(rz_engine *e, const rz_deferred_logits *logits, int32_t n_slots, void *stream);

/* Root statistics after the simulations (AlphaZeroMCTS.simulate, alphazero_mcts.py:88-90):
 * visit count / W of the root child of every action, 0 for illegal or unvisited actions;
 * [n_games][B*B].  rz_root_stats: N and W of the roots themselves, [n_games]. */
int rz_root_visits(rz_engine *e, int32_t *d_visits, void *stream);
int rz_root_wsum(rz_engine *e, double *d_w, void *stream);
int rz_root_priors(rz_engine *e, float *d_p, void *stream);
int rz_root_stats(rz_engine *e, int32_t *d_n, double *d_w, void *stream);

/* Tree reuse: AlphaZeroMCTS.update_with_move (alphazero_mcts.py:96-103).  d_moves[g] >= 0:
 * the subtree of that root child becomes the tree (statistics kept); -1: fresh tree
 * (reset_player, alphazero_mcts.py:132-134); -2: leave the game untouched.  Must be called
 * BEFORE rz_step_games for the same move (it ranks the move on the current root board). */
int rz_advance_roots(rz_engine *e, const int32_t *d_moves, void *stream);

/* GomokuEnv.step + game_end_winner on the root boards (gomoku_env.py:49-70,196-203):
 * d_moves[g] < 0 = no move.  d_winner: player id or -1 (tie / not ended); d_ended 0/1. */
int rz_step_games(rz_engine *e, const int32_t *d_moves, int32_t *d_winner, uint8_t *d_ended,
                  void *stream);

/* ---------------------------------------------------------------------------------------
 * THE MOVE STEP ON THE DEVICE.  What the self-play loop does between two searches -- AlphaZeroMCTS.simulate's read-out of the
 * root visits (alphazero_mcts.py:88-90), AlphaZeroPlayer.get_action's draw (:147-148), update_with_move (:96-103), env.step +
 * game_end_winner (game.py:109-118), reset_player at the end of a game (:128) and the start of the next one -- for every game
 * of the engine, enqueued on the stream with NO host round trip: the host keeps whole moves enqueued ahead and reads what
 * happened from a log, a move or more behind.
 *
 * The draw.  numpy.random.choice(acts, p=probs) of the reference is acts[searchsorted(cumsum(p) / cumsum(p)[-1], u, 'right')]
 * with p = softmax(log(N + 1e-10) / T) (alphazero_mcts.py:10-14,91-92).  The device evaluates the same expression in fp64 with
 * ITS log / exp (numpy's are not reproducible on another machine, let alone a GPU) on the uniform u = the counter-based
 * uniform of (seed, game id, ply) (rlzero_amd/selfplay.py: move_uniform -- integer arithmetic, the same bits) and draws the move
 * ONLY when u lies farther than stall_margin (relative to the total) from both edges of the chosen interval: rounding differences
 * between the two evaluations are ~1e-14, so the host's numpy expression on the logged visit counts -- the arbiter -- picks the
 * same move.  Otherwise the game STALLS: no move, the slot is skipped by the coming searches, and the host, reading the log,
 * decides with numpy and hands the move back (rz_play_resolve).  pi itself is never computed on the device: the host forms it
 * from the logged counts with the reference's expression (and verifies every move the device drew).
 *
 * The log: int32 [ring_steps][n_games][RZ_PLAY_RECORD_WORDS + A]; the record of move step s (counted per engine from
 * rz_play_attach) and slot g is row s % ring_steps.  Words: [0..1] game id (int64), [2] ply before the move, [3] the move (action)
 * or -1, [4] RZ_PLAY_* flags | (winner + 1) << 16, [5] N(root), [6] float bits of the draw's distance to the nearer interval edge
 * (relative), [7] reserved; then per action the visit count of the root child, -1 for an illegal action.  The host must read a
 * row before ring_steps more moves overwrite it.
 *
 * Slots refill themselves: a slot whose game has ended (or that is idle) takes the next entry of a queue of game ids shared by
 * the engines (lanes) of a GPU -- d_queue_ids int64 [..], d_queue_ctl int32 [2] = {head, entries valid}; the device advances
 * head atomically, the host may append ids and then raise the count -- and starts that game: empty board, player 0 to move, a
 * fresh tree, its Dirichlet stream keyed (seed, game id) exactly as rlzero_amd.selfplay keys it. */
#define RZ_PLAY_RECORD_WORDS 8
enum {
    RZ_PLAY_RUNNING = 1,   /* the slot holds a game: the visit counts are valid */
    RZ_PLAY_STALLED = 2,   /* no move drawn (u too close to an interval edge): waiting for rz_play_resolve */
    RZ_PLAY_RESOLVED = 4,  /* the move came from rz_play_resolve */
    RZ_PLAY_ENDED = 8,     /* the game ended with this move; winner + 1 in bits 16.. (0: tie) */
    RZ_PLAY_SEARCHED = 16  /* the slot took part in the search before this move step (n_playout simulations) */
};
typedef struct rz_play_config {
    uint64_t seed;         /* move uniforms keyed (seed, game id, ply), Dirichlet streams keyed (seed, game id) */
    double temperature;    /* T of softmax(log(N + 1e-10) / T) */
    double stall_margin;   /* 0 -> 1e-10 * max(1, 1 / T); a test hook otherwise (0.05 stalls one draw in ten) */
    const int64_t *d_queue_ids;
    int32_t *d_queue_ctl;
    int32_t *d_log;        /* the log ring [ring_steps][n_games][8 + A]: device memory, or (what rlzero_amd passes) pinned host memory that
                            * the device can address -- the kernels only write it (one read-modify-write of a record's flags when its
                            * game ends), so the host reads rows in place behind an event and no copy sits between two moves; host
                            * memory that this engine's device cannot address is refused (hipHostGetDevicePointer) */
    int32_t ring_steps;
    int32_t reserved;
} rz_play_config;
/* Attach (allocates the per-slot state on first use; every slot idle, inactive, with a fresh tree; move step counter 0). */
int rz_play_attach(rz_engine *e, const rz_play_config *cfg);
/* After the search of a move, BEFORE rz_deferred_flush: read the root visits into the log, draw (or stall, or take a resolved
 * move).  One launch. */
int rz_play_draw(rz_engine *e, void *stream);
/* Then, in ONE launch (a wave per slot): update_with_move + env.step + game_end_winner with the moves just drawn, the end of finished
 * games and the refill of idle slots (rz_play_apply without a draw before it only refills: how a run starts).  A
 * rz_deferred_flush between the two calls leaves the restart of the pending-priors counters to this launch. */
int rz_play_apply(rz_engine *e, void *stream);
/* The host's decision for a stalled slot (one tiny launch); taken by the next rz_play_draw. */
int rz_play_resolve(rz_engine *e, int32_t slot, int32_t move, void *stream);
/* Drop every game: all slots idle with fresh trees (a run that stops early). */
int rz_play_stop(rz_engine *e, void *stream);
/* Host copies of the per-slot state for inspection (synchronous): game ids int64 [n_games] (-1: idle), plies int32, states
 * int32 (0 idle, 1 running, 2 stalled), and the move steps done so far. */
int rz_play_state(rz_engine *e, int64_t *h_game_id, int32_t *h_ply, int32_t *h_state, int64_t *h_steps);

/* Synchronises `stream`-independent state: waits for the device, then reports flags. */
int rz_get_stats(rz_engine *e, rz_stats *out);
int rz_clear_errors(rz_engine *e);
/* The cheap form for a per-move check (the single-game API polls after every search): waits for `stream` only and reads the OR of
 * the games' RZ_FLAG_* bits (and, optionally, the count of dropped subtrees) -- 4 + 4 bytes; rz_get_stats names the game when a
 * bit is set. */
int rz_poll_errors(rz_engine *e, int32_t *h_flags, int32_t *h_reuse_dropped, void *stream);

/* Inspection for parity tests: copies game `g`'s current arena to HOST arrays of max_slots
 * entries -- per node record: N, W, slot of the first child record (-1 = none yet), number of
 * visited children (= child records in use), number of children K (0 = not expanded), offset of
 * the node's block of K child priors -- plus the root's own prior and the used-slot count.
 * rz_copy_priors copies the prior arena those offsets index.  Synchronous. */
int rz_copy_arena(rz_engine *e, int32_t game, int64_t max_slots, int32_t *h_n, double *h_w,
                  int32_t *h_first_child, int32_t *h_n_visited, int32_t *h_n_children,
                  int32_t *h_prior_block, float *h_root_prior, int32_t *h_top);
int rz_copy_priors(rz_engine *e, int32_t game, int64_t max_floats, float *h_priors, int32_t *h_count);

/* The scoring arithmetic alone, for bit-exactness tests against CPython:
 * out[i] = w[i]/n[i] + c*sqrt(ln(np[i])/n[i]) with ln from the engine's table. */
int rz_uct_scores(rz_engine *e, const double *d_w, const int32_t *d_n, const int32_t *d_np,
                  double c_puct, double *d_out, int64_t count, void *stream);

/* ---------------------------------------------------------------------------------------
 * Policy-value network forward (the evaluator's dense contraction), hand-written MFMA kernels.
 * Replaces PolicyValueNet.forward (rlzero/games/gomoku/policy_value_net.py:34-52) for a batch
 * of leaf observations: conv3x3 4->32->64->128 (+ReLU), conv1x1 heads, the three FC layers,
 * log_softmax and tanh.  f32 results: the default trunk (RZ_NET_SPLIT_F16) carries every f32 operand of
 * conv1..conv3 as a pair of f16 values on the f16 matrix pipe and accumulates in f32 (error against fp64
 * at the level of the exact-f32 kernel, 1e-7 relative) and the FC GEMM that follows it does the same
 * (rz_net_set_heads_algo); the other algorithms use the f32-input MFMA.
 *
 * rz_net_load takes HOST pointers to the 16 tensors of PolicyValueNet.state_dict() in its
 * order (conv1.weight, conv1.bias, conv2.*, conv3.*, act_conv1.*, act_fc1.*, val_conv1.*,
 * val_fc1.*, val_fc2.*; policy_value_net.py:12-25), fp32, contiguous, torch layout.
 * rz_net_reserve sizes the internal feature buffer (the launch path never allocates).
 * The board is height x width (<= 16 x 16) and the policy head has n_actions outputs (Gomoku:
 * height = width = B, n_actions = B*B; Connect4: 6 x 7, 7).
 * rz_net_forward: d_obs float32 [n][4][H][W] -> d_logp [n][n_actions] (log-probabilities),
 * d_value [n].  rz_net_trunk exposes the first kernel alone: d_feat [n][6][B*B] = ReLU'd
 * outputs of act_conv1 (4 planes) and val_conv1 (2 planes); d_feat == NULL writes the
 * internal buffer that rz_net_heads (the FC layers + log_softmax + tanh) reads, so
 * rz_net_trunk(.., NULL, ..) + rz_net_heads == rz_net_forward (lets a caller time the
 * dominant kernel by itself). */
typedef struct rz_net rz_net;
enum {
    RZ_NET_DIRECT = 0,      /* conv2/conv3 as direct implicit GEMM on the f32-input MFMA: bit-for-bit a k-ordered fp32 fmaf
                               chain (the exact reference arithmetic; also what a net without finite activation bounds runs on) */
    RZ_NET_WINOGRAD_F4 = 1, /* conv2/conv3 as Winograd F(4x4,3x3) on the f32-input MFMA: 4x fewer multiply-adds than
                               DIRECT; fp32 throughout, ~1e-6 relative from DIRECT */
    RZ_NET_SPLIT_F16 = 2    /* default: conv1..conv3 as direct convolutions on the f16 matrix pipe with every f32 operand carried
                               as a hi + lo pair of f16 values (three MFMAs per product, f32 accumulation): as accurate as
                               DIRECT (1e-7 relative against fp64).  The f16 pieces of a layer's activations are stored times a
                               power of two that rz_net_load derives from bounds on the activations (observation planes in
                               [0, 1]), so weights of any scale stay in range on MCTS leaves; only inputs beyond [0, 1]
                               can overflow, which sets RZ_NET_FLAG_F16_RANGE.  Boards of 11 .. 16 rows and columns run
                               k_trunk_rows (v_mfma_f32_16x16x32_f16, one N-tile per board row, waves split the output
                               channels), all others k_trunk_split (32 x 32 x 16 tiles of whole rows, waves split the rows) */
    ,
    RZ_NET_SPLIT_F16_TILES = 3  /* the same arithmetic with k_trunk_split on every board size (the checker of k_trunk_rows:
                               the two agree to f32 accumulation rounding, not bit for bit -- the MFMA shapes sum in
                               different orders) */
    ,
    RZ_NET_SPLIT_F16_FP8 = 4 /* OPT-IN, narrower than the reference's f32 (never the default, never the benchmark's headline):
                               RZ_NET_SPLIT_F16 with the two CROSS terms hi x lo + lo x hi of conv3 (80 % of the trunk's products)
                               on the block-scaled FP8 pipe -- one v_mfma_scale_f32_16x16x128_f8f6f4 per tap (weights e4m3,
                               activations e5m2) instead of four f16 MFMAs, 2 f16-MFMA equivalents per product instead of 3.
                               The cross terms then carry 3 .. 4 bits instead of 11: ~2^-14 per product, between plain f16
                               (2^-11) and the default (2^-22); measured on the logits: DESIGN.md.  Boards of 11 .. 16 rows and
                               columns, position-fed entry points only (rz_net_trunk_leaves*, rz_net_search_resident);
                               rz_net_trunk / rz_net_forward (float planes) return RZ_ERR_ARG while it is selected. */
};
/* rz_net_range_info: h_info8 = {bound on conv1's, conv2's activations and on the head features for inputs in [0, 1];
 * the three activation scales chosen from them; 1.0 if the bounds are finite (0.0: RZ_NET_SPLIT_F16 runs RZ_NET_DIRECT
 * instead for this net); 0} -- valid after rz_net_load. */
int rz_net_range_info(rz_net *net, float *h_info8);
/* rz_net_error_flags: sticky bits set by the kernels since creation / the last call (synchronises the device) */
enum { RZ_NET_FLAG_F16_RANGE = 1 };
int rz_net_error_flags(rz_net *net, uint32_t *h_flags);
int rz_net_set_algo(rz_net *net, int32_t algo);
/* The Winograd trunk runs as persistent workgroups (one per CU: its LDS and registers fill a CU),
 * each looping over its boards.  max_workgroups > 0 caps their number so that the remaining CUs stay
 * free for the latency-bound tree / FC kernels of ANOTHER stream (a second lane of games) running
 * beside the trunk; 0 (default) = one workgroup per CU.  Workgroups are dealt to the 8 XCDs in turn:
 * use a multiple of 8. */
int rz_net_set_max_workgroups(rz_net *net, int32_t max_workgroups);
/* The first FC layers of the two heads (act_fc1, val_fc1; policy_value_net.py:43,48) as one GEMM over the leaf batch.
 * RZ_NET_HEADS_F32: the f32-input MFMA GEMM (32 x 32 output blocks, many small workgroups).  RZ_NET_HEADS_SPLIT_32 /
 * _64: on the f16 matrix pipe with hi + lo f16 operand pairs (k_heads_split, the arithmetic of RZ_NET_SPLIT_F16), 32 /
 * 64 boards per workgroup; needs the f16 feature pieces that the RZ_NET_SPLIT_F16 trunk writes into the internal
 * buffer, and falls back to F32 when the last trunk was another one.
 * RZ_NET_HEADS_AUTO (default): after the RZ_NET_SPLIT_F16 trunk SPLIT_64 when the trunk is capped by
 * rz_net_set_max_workgroups (the GEMM then has only the few CUs the trunk leaves free), otherwise SPLIT_PARTS up to 256
 * boards and SPLIT_32 above -- all give the same bits; F32 after the other trunks.  RZ_NET_HEADS_SPLIT_PARTS: the same arithmetic with NO reduction inside
 * the GEMM -- one single-wave workgroup per (32 boards x 32 outputs x K quarter), no LDS, few registers, so its waves fit on
 * a CU beside a resident trunk workgroup of another lane; the four partial sums are added by the consumer (the tree kernel
 * of the fused route, k_heads_finish otherwise; see rz_raw_heads) in the order the other shapes use: same bits again.
 * Choose before the trunk is launched: into the internal buffer the
 * RZ_NET_SPLIT_F16 trunk writes only what the chosen GEMM reads (F32 chosen later runs SPLIT on the pieces present). */
enum { RZ_NET_HEADS_AUTO = 0, RZ_NET_HEADS_F32 = 1, RZ_NET_HEADS_SPLIT_32 = 2, RZ_NET_HEADS_SPLIT_64 = 3, RZ_NET_HEADS_SPLIT_PARTS = 4,
       RZ_NET_HEADS_IN_TRUNK = 5   /* boards of up to 10 rows on the RZ_NET_SPLIT_F16 trunk: every trunk workgroup runs these layers on
                                      its own board behind the feature stage (no GEMM launch; the bits of the SPLIT shapes); what AUTO
                                      picks for an un-capped batch of at most one board per CU while the FC weights are at most 40 KB (6 x 6, Connect4);
                                      other boards: as AUTO */ };
int rz_net_set_heads_algo(rz_net *net, int32_t heads_algo);
int rz_net_create(int32_t height, int32_t width, int32_t n_actions, int32_t device, rz_net **out);
int rz_net_destroy(rz_net *net);
/* Uploads (and re-packs) the 16 tensors of PolicyValueNet.state_dict().  Later calls reuse the device
 * buffers of the first one, so launches captured in a hipGraph stay valid across weight updates. */
int rz_net_load(rz_net *net, const float *const *h_params, int32_t n_params);
int rz_net_reserve(rz_net *net, int32_t max_boards);
int rz_net_trunk(rz_net *net, const float *d_obs, int32_t n_boards, float *d_feat, void *stream);
/* The RZ_NET_SPLIT_F16 trunk fed with the leaf POSITIONS instead of their float planes: d_stones uint64 [n][2][4]
 * (colour 0 / colour 1 bitboards, bit = cell), d_to_move, d_last_cell int32 [n] -- the engine's own leaf arrays
 * (rz_leaf_buffers).  The kernel builds the four planes of GomokuEnv.current_state (gomoku_env.py:95-114) itself; into
 * the internal feature buffer only (pair with rz_net_heads_gemm / rz_net_heads).  Same bits as rz_net_trunk on the planes
 * rz_select_step would have written. */
int rz_net_trunk_leaves(rz_net *net, const uint64_t *d_stones, const int32_t *d_to_move, const int32_t *d_last_cell,
                        int32_t n_boards, void *stream);
/* Deferred priors (see rz_value_head): room for `slots` steps of `max_boards` leaves in the policy-feature store
 * (slots x ceil(max_boards / 32) tiles of f16 pieces) and for their logits. */
int rz_net_deferred_reserve(rz_net *net, int32_t max_boards, int32_t slots);
/* rz_net_trunk_leaves of the deferred route: board b's policy features go to slot d_slot_of_board[b] of the store, its
 * value features (f32) to the rows of *out; no FC GEMM follows. */
int rz_net_trunk_leaves_deferred(rz_net *net, const uint64_t *d_stones, const int32_t *d_to_move, const int32_t *d_last_cell,
                                 int32_t n_boards, const int32_t *d_slot_of_board, rz_value_head *out, void *stream);
/* act_fc1 (policy_value_net.py:43) over the stored leaves of slots [0, n_slots) as ONE GEMM (k_heads_split's arithmetic) */
int rz_net_deferred_gemm(rz_net *net, int32_t n_boards, int32_t n_slots, rz_deferred_logits *out, void *stream);
int rz_net_trace_attach(rz_net *net, void *d_trace);   /* see rz_trace_attach */
/* RECEPTIVE-FIELD ("delta") LEAF EVALUATION -- PolicyValueNet.forward (policy_value_net.py:34-52) on the leaves of a search WITHOUT
 * recomputing what the root already determines.  The reference's search is near breadth-first (alphazero_mcts.py:42-71 under
 * node.py:32-42: a 15 x 15 / 800 leaf is the root plus one or two stones), and three 3 x 3 convolutions move conv3's output only in
 * the 7 x 7 window around a changed cell.  rz_net_delta_bases evaluates, per game, two pseudo-positions of the ROOT (its stones seen
 * by the side to move / by the other side one stone later; no last-move plane) and keeps conv1's / conv2's outputs and the six head
 * feature planes in a cache (rz_net_delta_reserve: 209 KB per game); rz_net_delta_leaves then computes, per leaf, only the cells
 * inside the windows of the cells where its planes (gomoku_env.py:95-114) differ from the base of its parity, with k_trunk_rows'
 * arithmetic cell by cell, and takes every other cell from the base: THE SAME BITS as rz_net_trunk_leaves_deferred (the store slot
 * d_slot_of_board[b], the value rows of *out).  The cache validates itself: a leaf whose stones do not contain the cached root's, or
 * with more than four changed cells, or a game without bases, is evaluated by the same kernel without a base (four passes over
 * the board's quadrants) -- correct whatever the caller did, only slower; rebuild the bases whenever the roots move.
 * Boards of 11 .. 16 rows and columns, RZ_NET_SPLIT_F16.  d_active (may be NULL): games whose flag is 0 are skipped.  d_feat32 (may be
 * NULL): also the features as f32 [n][6][S] (tests); d_slot_of_board may then be NULL (nothing goes to the store; out may be NULL).
 * without_base != 0: every leaf takes the four-pass route (a checker). */
int rz_net_delta_reserve(rz_net *net, int32_t n_games);
int rz_net_delta_invalidate(rz_net *net, void *stream);
/* rz_net_search_resident on these boards, once the cache holds the engine's games: the resident search with THIS trunk (k_delta_res:
 * 82 KB of LDS, two games per CU, ANY number of games per launch -- a workgroup depends on nothing outside its game, so a grid beyond
 * 2 x CUs runs in rounds, a CU's free half going to the next game as a search ends; the bases of the roots are built by the call itself
 * when select_first != 0).  on = 0: the full-board resident kernel (one game per CU, at most CUs games) as before.  Default: on. */
int rz_net_delta_resident(rz_net *net, int32_t on);
int rz_net_delta_bases(rz_net *net, const uint64_t *d_root_stones, const int32_t *d_root_to_move, int32_t n_games, void *stream);
int rz_net_delta_leaves(rz_net *net, const uint64_t *d_stones, const int32_t *d_to_move, const int32_t *d_last_cell, int32_t n_boards,
                        const int32_t *d_slot_of_board, const uint8_t *d_active, float *d_feat32, int32_t without_base, rz_value_head *out,
                        void *stream);
/* the same on an engine's own arrays: the bases of every game from its ROOT positions (call whenever the roots have moved: after
 * rz_set_roots, rz_step_games, rz_play_apply -- a forgotten call costs time, not correctness), and rz_net_trunk_leaves_deferred's
 * step on its leaves (board b = game b, store slot pend[b], inactive games skipped).  One simulation in flight per tree. */
int rz_net_delta_bases_engine(rz_net *net, rz_engine *engine, void *stream);
int rz_net_delta_step(rz_net *net, rz_engine *engine, rz_value_head *out, void *stream);
/* ... and rz_net_trunk_leaves' step on its leaves: the features into the internal buffer's f16 tiles (policy and value K-steps),
 * rz_net_heads_gemm next -- the three-launch step (the opt-in PUCT rule) with this trunk.  One simulation in flight per tree;
 * RZ_NET_HEADS_F32 is refused (it reads f32 features).  Replaces the same reference lines as rz_net_trunk_leaves
 * (policy_value_net.py:34-46 on gomoku_env.py:95-114's planes). */
int rz_net_delta_trunk_engine(rz_net *net, rz_engine *engine, void *stream);
/* counters since the last reset (synchronises): {leaves evaluated against a base, leaves without one, conv3 tiles of 16 cells, changed
 * cells, conv2 tiles of 16 cells, 0, and -- of workgroup 0 of the LAST resident launch -- its shader-clock cycles >> 8 and its ticks
 * of the constant 100 MHz clock: the clock the search ran at = 256 [6] / (10 ns [7])} */
int rz_net_delta_stats(rz_net *net, uint32_t *h_out8, int32_t reset);
/* RESIDENT SEARCH -- n_sims consecutive simulations of every active game of `engine` (AlphaZeroMCTS.simulate's loop,
 * alphazero_mcts.py:82-85) in ONE launch, one workgroup per game: trunk -> value head -> expand / backup -> next selection without
 * a kernel boundary, the leaf handed from the tree code to the trunk through LDS.  Kernels that hold a whole CU (151 KB of LDS: boards
 * of 8 .. 10 rows, and 11 .. 16 without rz_net_delta_reserve) take at most one game per CU (the single-game API, BASELINE configs[0]
 * and [1]); the two that hold half a CU -- k_delta_res (above; configs[3]) and k_trunk_split on the compact LDS grid (boards of up to 7
 * columns, 69 KB: TicTacToe .. 7x7, Connect4's 6x7 = configs[2]) -- take any number of games, two per
 * CU at a time.  The deferred-priors route's arithmetic and bookkeeping: the first leaf comes from
 * rz_select_step(engine, NULL, ..) before the call (select_first == 0: a search continued in pieces) or is selected by the launch
 * itself (select_first != 0: the same selection by the same code, one launch less); rz_net_deferred_gemm + rz_deferred_flush later;
 * the engine's slots advance by n_sims.  Same trees, values and priors as rz_net_trunk_leaves_deferred + rz_tree_step_deferred, bit for bit.
 * CONTRACT: game g's leaves go to store slots pend[g] .. pend[g] + n_sims - 1 (pend[g] = its steps since the last
 * rz_deferred_flush), so pend[g] + n_sims must not exceed the slots of rz_net_deferred_reserve / rz_deferred_reserve: flush first.
 * n_sims beyond either capacity is refused (RZ_ERR_ARG); a leaf whose slot still lies beyond the store (a caller that did not
 * flush) is not written anywhere and its game is flagged RZ_FLAG_INTERNAL by the tree code -- never an out-of-bounds write.
 * The same holds for rz_net_trunk_leaves_deferred (slot d_slot_of_board[b]). */
int rz_net_search_resident(rz_net *net, rz_engine *engine, int32_t n_sims, int32_t select_first, void *stream);
int rz_net_heads(rz_net *net, int32_t n_boards, float *d_logp, float *d_value, void *stream);
/* only the FC GEMM of the heads on the internal features; returns the device pointers that
 * rz_tree_step_raw / rz_expand_backup_raw consume (valid until the next rz_net_reserve / load) */
int rz_net_heads_gemm(rz_net *net, int32_t n_boards, rz_raw_heads *out, void *stream);
int rz_net_forward(rz_net *net, const float *d_obs, int32_t n_boards, float *d_logp,
                   float *d_value, void *stream);

/* ---------------------------------------------------------------------------------------
 * MuZero search tree (BASELINE.json configs[4]).  The reference only names
 * MuZero (README.md:3, rlzero/algorithms/rl_args.py:21-24); the algorithm is the published
 * pseudocode of arXiv:1911.08265v2 (run_mcts, select_child, ucb_score, expand_node, backpropagate,
 * MinMaxStats), single-player form.  The learned model stays with the caller: per simulation
 *   rz_mz_select         -> (parent slot, action, leaf slot) per game; the caller gathers the parents'
 *                           hidden states, runs dynamics + prediction on the batch, keeps the new
 *                           hidden state of game g at slot leaf[g];
 *   rz_mz_expand_backup  <- reward, policy PROBABILITIES [n_games][n_actions] (softmax evaluated by the
 *                           network head in fp32), value; expands the leaf and backs the value up with
 *                           the discount, updating the game's min-max statistics.
 * rz_mz_init_roots starts a search (expand_node(root, initial inference) + add_exploration_noise with
 * caller-supplied Dirichlet samples, fresh MinMaxStats).  d_mask (optional, [n_games] bytes): games with
 * 0 are left untouched.  Slots: root = 0; the children of a node are consecutive slots. */
typedef struct rz_muzero rz_muzero;
typedef struct rz_mz_config {
    int32_t abi_version; /* RZ_ABI_VERSION */
    int32_t n_games;
    int32_t n_actions;   /* 1..64 */
    int32_t n_sims;      /* simulations per search (sizes the tree: 1 + n_actions * (n_sims + 1) slots) */
    double discount;     /* 0.997 */
    double pb_c_base;    /* 19652 */
    double pb_c_init;    /* 1.25 */
    int32_t device;
    int32_t reserved;
} rz_mz_config;

int rz_mz_create(const rz_mz_config *cfg, rz_muzero **out);
int rz_mz_destroy(rz_muzero *e);
/* log((n + pb_c_base + 1) / pb_c_base) for n = 0 .. n_sims + 1, filled at creation by the host libm;
 * this entry replaces it (e.g. with math.log's values from another host). */
int rz_mz_upload_log_table(rz_muzero *e, const double *h_table, int64_t count);
int rz_mz_init_roots(rz_muzero *e, const float *d_probs, const double *d_noise, double noise_frac,
                     const uint8_t *d_mask, void *stream);
int rz_mz_select(rz_muzero *e, int32_t *d_parent, int32_t *d_action, int32_t *d_leaf, const uint8_t *d_mask,
                 void *stream);
int rz_mz_expand_backup(rz_muzero *e, const float *d_reward, const float *d_probs, const float *d_value,
                        const uint8_t *d_mask, void *stream);
/* what: 0 = visit counts (int32), 1 = value sums, 2 = rewards, 3 = priors (float64) of the root's children,
 * [n_games][n_actions] */
/* The search in ONE launch (k_mz_search): the model of rlzero_amd/muzero/network.py -- dynamics g(s, a) -> (r, s'),
 * prediction f(s) -> (p, v), hidden size 64, <= 8 actions -- evaluated inside the kernel on the matrix pipe, a workgroup
 * keeping <= 16 games, its waves' weight fragments in registers and (when they fit) the games' trees in LDS for all n_sims
 * simulations.  rz_mz_set_search_shape: games per workgroup, 0 (default) = chosen from n_games and the CU count so that
 * every CU holds at least two workgroups when there are games enough (4 / 8 / 16).  rz_mz_load_model takes HOST pointers to 14 fp32 tensors in torch layout
 * ([out][in]): dyn1.weight [64][64 + A], dyn1.bias, dyn2.weight, dyn2.bias, rew1.weight, rew1.bias, rew2.weight [1][64],
 * rew2.bias, pre1.weight, pre1.bias, pol.weight [A][64], pol.bias, val.weight [1][64], val.bias; call again after every
 * optimiser step.  rz_mz_search: d_hidden float32 [n_games][slots_per_game][64] with slot 0 = the root's state from the
 * initial inference (rz_mz_init_roots first); runs n_sims simulations of every game.  The six trace arrays (all or none):
 * per simulation and game what the kernel selected and what its network returned, [n_sims][n_games] (probs:
 * x n_actions) -- the parity tests feed them to the CPython restatement of the pseudocode. */
int rz_mz_load_model(rz_muzero *e, const float *const *h_params, int32_t n_params, int32_t hidden);
int rz_mz_search(rz_muzero *e, float *d_hidden, int32_t n_sims, int32_t *d_trace_parent, int32_t *d_trace_action,
                 int32_t *d_trace_leaf, float *d_trace_reward, float *d_trace_probs, float *d_trace_value, void *stream);
int rz_mz_set_search_shape(rz_muzero *e, int32_t games_per_workgroup);
/* Whole MOVES of CartPole-v1 environments in one launch (k_mz_search with its MOVES stages): per move the initial
 * inference h(o) -> s0, f(s0) -> root priors (+ Dirichlet(alpha) noise, weight noise_frac), n_sims simulations, the action
 * drawn from visits ^ (1 / temperature) (arg-max at temperature <= 0), one record and the environment step with
 * auto-reset.  rz_mz_load_representation: HOST pointers to rep1.weight [64][obs_dim], rep1.bias, rep2.weight [64][64],
 * rep2.bias (torch layout), beside rz_mz_load_model.
 * The environments: d_state float64 [n_games][4] (x, x_dot, theta, theta_dot), d_steps / d_episode int64 [n_games], updated
 * in place (initial states of an episode: the counter-based stream of rlzero_amd/muzero/cartpole.py keyed (env_seed,
 * environment, episode)).  Random draws come from a counter-based stream keyed (noise_seed, environment, episode, step).
 * The history stays on the device: a record is 8 + A float64 -- observation before the move (4) | action | reward | visit
 * counts (A) | root value | done; every move's record goes to d_ring [n_games][ring_steps][8 + A] at step % ring_steps
 * (global step index = first_step + move of the launch; ring_steps >= 500 + n_moves), d_episode_start int64 [n_games]
 * holds the step at which the running episode began.  When an episode ENDS its records are copied, as one contiguous run,
 * to d_arena [arena_rows][8 + A] and (environment, end step, length, first arena row) is appended to d_entries int64
 * [max_entries][4] -- the host reads finished episodes, not moves.  d_counters int64 [4], zeroed by the caller before the
 * launch: [0] arena rows claimed, [1] entries, [2] episodes that did not fit into the arena (their entry has row -1:
 * read them from d_ring). */
typedef struct rz_mz_cartpole_play {
    double *d_state;
    int64_t *d_steps, *d_episode, *d_episode_start;
    uint64_t env_seed, noise_seed;
    double noise_frac, dirichlet_alpha, temperature;
    double *d_ring;
    int32_t ring_steps, reserved;
    int64_t first_step;
    double *d_arena;
    int64_t arena_rows;
    int64_t *d_counters, *d_entries;
    int64_t max_entries;
} rz_mz_cartpole_play;
int rz_mz_load_representation(rz_muzero *e, const float *const *h_params, int32_t n_params, int32_t obs_dim, int32_t hidden);
int rz_mz_play_cartpole(rz_muzero *e, float *d_hidden, int32_t n_sims, int32_t n_moves, const rz_mz_cartpole_play *play, void *stream);
/* One step of n_envs CartPole-v1 environments (gymnasium classic_control/cartpole.py: Euler, tau 0.02, 500-step limit)
 * with auto-reset, in ONE launch: d_obs float32 [n_envs][4] = the observation AFTER the step (after the reset for a
 * finished environment), d_reward float32, d_terminated / d_truncated uint8. */
int rz_cartpole_step(double *d_state, int64_t *d_steps, int64_t *d_episode, const int64_t *d_actions, int32_t n_envs, uint64_t seed,
                     float *d_obs, float *d_reward, uint8_t *d_terminated, uint8_t *d_truncated, void *stream);
int rz_mz_root_children(rz_muzero *e, int32_t what, void *d_out, void *stream);
int rz_mz_root_stats(rz_muzero *e, int32_t *d_n, double *d_value_sum, double *d_vmin, double *d_vmax, void *stream);
int rz_mz_geometry(rz_muzero *e, int32_t *slots_per_game, int64_t *device_bytes);
int rz_mz_error_flags(rz_muzero *e, int32_t *flags);

#ifdef __cplusplus
}
#endif
#endif /* RLZERO_HIP_H */
