/*
 * othellozero_amd.h -- C ABI of libothellozero_amd.so (MI355X / gfx950).
 *
 * Drop-in boundary for the self-play hot path of Galtvam/OthelloZero.  The
 * reference has no FFI layer (its boundary is Python duck typing), so each entry
 * point below cites the reference interface it replaces (file:line under the
 * reference tree); the othellozero_amd Python package binds them with ctypes and re-creates
 * the reference's Python surfaces on top.
 *
 * Conventions: plain pointers and sizes, caller-owned output buffers, int status
 * return (0 = OZ_OK) + oz_last_error() (thread-local string); no exceptions
 * cross the ABI.  Unless a parameter says "device pointer" every pointer is a
 * host pointer.  Boards: two uint64 bitboards, bit = row*8 + col for every board
 * size n in {4,6,8} (n x n corner of an 8x8 grid).  own/opp = channel 0/1 of a
 * mover-canonical state; black/white = absolute colours.  Squares are reported
 * as sq = row*8 + col; NN policy vectors are indexed row*n + col.
 * All objects are internally serialised (a mutex per object): concurrent calls
 * from ThreadWorker-style Python threads (workers.py:33-37,82-90) are safe.
 */
#ifndef OTHELLOZERO_AMD_H
#define OTHELLOZERO_AMD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define OZ_OK 0
#define OZ_ERR_HIP 1       /* a HIP runtime call failed                         */
#define OZ_ERR_ARG 2       /* bad argument                                      */
#define OZ_ERR_CAPACITY 3  /* a per-game node table / record buffer overflowed      */
#define OZ_ERR_KEY 4       /* the reference would raise KeyError here           */
#define OZ_ERR_STATE 5     /* call sequence error                               */

/* Q accumulation regime (SURVEY.md R-FP): what `N*Q + value` promotes to */
#define OZ_QMODE_NEP50 0   /* NumPy >= 2: float32 once a float32 value arrives  */
#define OZ_QMODE_F64 1     /* NumPy 1.18.5 (requirements.txt:19): float64       */

/* per-game status after a select pass */
#define OZ_LEAF_IDLE 0       /* game slot not searching                          */
#define OZ_LEAF_TERMINAL 1   /* simulation ended on a finished board (MCTS/__init__.py:39-40) */
#define OZ_LEAF_EVAL 2       /* first visit: needs NN evaluation (MCTS/__init__.py:44-57)     */
#define OZ_LEAF_WAIT 3       /* free-running driver with a batch cap: the leaf is chosen and waits for a slot of a later batch */

const char* oz_last_error(void);
int oz_version(void);                 /* 200 */
int oz_device_count(void);
int oz_set_device(int device);       /* device used by objects created afterwards on this thread */

/* ------------------------------------------------------------------ rules
 * Batched Othello/__init__.py static board functions, one HIP thread per position. */
/* get_player_valid_actions (:208-214): legal[i] = bit mask of legal squares for the side holding own[i] */
int oz_rules_legal_moves(const uint64_t* own, const uint64_t* opp, int n, int count, uint64_t* legal);
/* flip_board_squares (:237-247) as the side holding own[i] on sq[i] (no legality check) */
int oz_rules_apply_moves(const uint64_t* own, const uint64_t* opp, const uint8_t* sq, int n, int count,
                         uint64_t* own_out, uint64_t* opp_out);
/* has_board_finished (:249-252), get_board_players_points / get_board_winning_player (:254-260; draw -> ch0) */
int oz_rules_status(const uint64_t* ch0, const uint64_t* ch1, int n, int count, uint8_t* finished,
                    int32_t* pts0, int32_t* pts1, int8_t* winner /* +1 ch0, -1 ch1 */);
/* OthelloGame.play (:136-159): flip, switch player, pass / finish logic. player: +1 BLACK, -1 WHITE */
int oz_rules_play(const uint64_t* black, const uint64_t* white, const int8_t* player, const uint8_t* sq, int n,
                  int count, uint64_t* black_out, uint64_t* white_out, int8_t* player_out, uint8_t* finished_out);

/* ------------------------------------------------------------------ network
 * NNetWrapper (Net/NNet.py:22-101) inference side; OthelloNN graph (Net/OthelloNN.py:42-56). */
typedef struct oz_net oz_net;
/* ONN with `channels` conv filters (reference: 512), for boards n x n, batches up to max_batch */
int oz_net_create(oz_net** out, int n, int channels, int max_batch);
/* BaseNN (Net/BaseNN.py:41-56): same trunk on ONE input plane (+1 mover, -1 opponent); 40 weight arrays, conv1 kernel (3,3,1,C) */
int oz_net_create_bnn(oz_net** out, int n, int channels, int max_batch);
/* deterministic integer-hash stand-in for predict() (test nets; formula in oracle/oz_oracle.c orc_stub_predict) */
int oz_net_create_stub(oz_net** out, int n, uint64_t salt, uint64_t keep_mask, int max_batch);
int oz_net_destroy(oz_net* net);
/* weights in keras Model.get_weights() order: 40 arrays for ONN (model.set_weights / get_weights, Net/NNet.py:98-101) */
int oz_net_num_weights(const oz_net* net);
int oz_net_weight_size(const oz_net* net, int index, int64_t* nelem);
int oz_net_set_weight(oz_net* net, int index, const float* data, int64_t nelem);
int oz_net_get_weight(const oz_net* net, int index, float* data, int64_t nelem);
/* a fresh OthelloNN as Keras initialises it (Net/OthelloNN.py:42-56: glorot_uniform kernels, zero biases, identity BatchNormalization) from a
 * deterministic stream keyed by `seed`; follow with oz_net_commit.  Not NumPy's numbers for that seed: read them back with oz_net_get_weight. */
int oz_net_init_random(oz_net* net, uint64_t seed);
/* arithmetic of the 3x3 convolutions and dense layers: 0 = exact fp32 matrix cores (v_mfma_f32_32x32x2_f32);
 * 1 = "f32 via 2 x fp16 split" (fp32-EQUIVALENT, not fp32): x = h1 + h2, products a1*b1 + a1*b2 + a2*b1 on v_mfma_f32_16x16x32_f16 with
 * fp32 accumulation.  An element is carried with an error of max(2^-24 |x|, 2^-25) inside the fp16 range, and oz_net_commit places every
 * tensor: each activation channel and each weight column gets an exact power-of-two scale -- activations from the maxima of |BN output| over
 * a fixed calibration set of positions -- so that its maximum lands in [2^-3, 2^-2): every element then has an absolute error <= 2^-23 of
 * its channel's (column's) calibration maximum, fp32's own relative precision where a dot product's large terms are (the scales are folded
 * into the BN scale / shift and the next layer's weights: the network function is unchanged; oz_net_get_scaling reads the exponents).
 * Guards, sticky, reported as OZ_ERR_STATE by oz_net_check / predict / selfplay_sync: an activation above 65504, or a pixel row whose
 * largest activation is non-zero and 2^15 or more below its channels' calibration maxima.  oz_net_commit also runs the self-check of
 * OZ_NET_OPT_SELF_CHECK.  Needs channels % 256 == 0 and channels <= 2048.
 * 2 = "f32 via 3 x bf16 split" (fp32-class): x = b1 + b2 + b3 with b1 = bf16(x), b2 = bf16(x - b1), b3 = bf16(x - b1 - b2) -- bf16 has fp32's
 * exponent range and 8 significand bits, so the three planes hold every normal fp32 value EXACTLY: no scaling, no calibration, no guards, no
 * refusal path.  A product keeps six of the nine cross terms (a3 b1, a1 b3, a2 b2, a2 b1, a1 b2, a1 b1 on v_mfma_f32_16x16x32_bf16, fp32
 * accumulation, small terms first); the dropped ones are <= 2^-25 of the product each, within fp32's own rounding of it.  conv3, conv4, fc1 and
 * fc2 run this way on k_gemm_b3 (oz_net_b3.h), conv1 + conv2 from the exact-fp32 pattern tables, the heads in fp32; networks with max_batch < 128
 * (the latency path: weight streams and split-K launches, not matrix rate) run precision 0's kernels unchanged.  Needs channels % 256 == 0.
 * Cost 6 MFMAs at 16x the fp32 rate: 2.67x the fp32 matrix roof.
 * Takes effect at the next oz_net_commit. */
int oz_net_set_precision(oz_net* net, int mode);
int oz_net_get_precision(const oz_net* net);
int oz_net_check(oz_net* net);
/* fold BN (epsilon 1e-3, moving statistics) + re-layout for the kernels; call after the last set_weight */
int oz_net_commit(oz_net* net);
/* NNetWrapper.predict (Net/NNet.py:70-87) for `count` canonical boards: pi[count][n*n] float32, v[count] float32 */
int oz_net_predict(oz_net* net, const uint64_t* own, const uint64_t* opp, int count, float* pi, float* v);
/* the same on boards in the reference's layout (Net/NNet.py:80-84): count x (n, n, 2) bytes NHWC, channel 0 = the mover, non-zero = a disc */
int oz_net_predict_boards(oz_net* net, const uint8_t* boards_nhwc, int count, float* pi, float* v);
/* timing hook for bench.py: run the forward `iters` times on `count` resident boards, return avg ms per forward (HIP events) */
int oz_net_time_forward(oz_net* net, int count, int iters, float* ms_avg);
/* HIP-event timing of the dominant launch on the stream it is launched on: the conv3 implicit GEMM when conv1 + conv2
 * run as a table gather-sum (the default), else the conv2 implicit GEMM; oz_net_profiled_layer says which (3 / 2) */
int oz_net_profile(oz_net* net, int enable /* 0 off, 1 the dominant launch only, 2 every kernel of the forward */);
int oz_net_profile_read(oz_net* net, double* conv2_ms_total, int64_t* conv2_launches);
/* per-kernel totals since creation (or the last reset), slots: 0 input (k_lut_ids, or the conv1 kernel) 1 conv2 (table
 * gather-sum or GEMM) 2 conv3 3 conv4 4 fc1 (+ split-K reduce) 5 fc2 6 heads; only slots enabled by the profile mode advance */
#define OZ_NET_KERNELS 7
int oz_net_profile_kernels(oz_net* net, double* ms_total /* [OZ_NET_KERNELS] */, int64_t* launches /* [OZ_NET_KERNELS] */, int reset);
int oz_net_profiled_layer(oz_net* net, int* layer);
/* how conv1 / conv2 are evaluated.  2 (default): both from tables over the 3^9 neighbourhood patterns of the discrete
 * input planes (no GEMM for conv2; tables rebuilt by oz_net_commit); 1 (precision f16x2, max_batch > 32): conv1 from its table
 * inside conv2's operand gather, conv2 an MFMA GEMM (bit-identical to 0); 0: conv1 kernel + conv2 MFMA GEMM; -1: back to the
 * default.  Takes effect at the next forward. */
int oz_net_set_tables(oz_net* net, int mode);
/* Persistent exact-key evaluation cache of this network: (own, opp) -> (pi, v) for up to ~`entries` positions in HBM (rounded up to a
 * power of two of 4-way buckets; n*n + 6 words per entry; 0 frees it).  The generalisation of the reference's per-search
 * `_predict_cache` (othelo_mcts.py:13,82-88: one dict per OthelloMCTS instance) to every self-play engine that is created with
 * oz_selfplay_config.eval_cache = 1 on this network -- across batches, games and refilled slots.  A hit changes no bit of any result
 * (a position's (pi, v) is independent of the batch it is evaluated in); the cache is emptied by oz_net_commit (new weights) and by an
 * oz_net_set_tables that changes the form of conv1 / conv2 (the table form adds conv2's products in another order than the GEMM forms).
 * Engines that share a cached network may run one after the other (a second engine starts warm) but NOT concurrently: their enqueues are
 * serialised by the network's mutex, their streams are not, and an entry one engine replaces could be read half-written by the other. */
int oz_net_set_eval_cache(oz_net* net, int64_t entries);
int oz_net_eval_cache_stats(oz_net* net, int64_t* entries, int64_t* lookups, int64_t* hits, int64_t* inserts);
/* diagnostics switch, per network (default 0): the 3x3 convolutions of precision f16x2 on the one-barrier-per-k-tile main loop instead of
 * the ping-pong loops (4-phase on the 256-row tile, 2-phase on the 192- and 128-row tiles).  Same accumulation order, bit-identical results: the reference form the LDS-DMA race screen
 * (tools/pp_race_check.py, test_pingpong_conv_loop_bit_identical_to_simple_loop) compares the ping-pong schedule against. */
#define OZ_NET_OPT_SIMPLE_LOOP 1
/* precision f16x2, all take effect at the next oz_net_commit: the powers of two the per-channel calibration maxima (ACT, default -2) and the
 * per-column weight maxima (W, default -2) are moved below -- the defaults are the measured optimum of tools/target_probe.py, other values
 * are test hooks and experiments -- and log2 of the low-side guard's row threshold (default -17; <= -100 switches the guard off) */
#define OZ_NET_OPT_ACT_TARGET_LOG2 2
#define OZ_NET_OPT_LOW_GUARD_LOG2 3
/* precision f16x2, default 1: oz_net_commit runs its calibration positions through the f16x2 kernels and through the exact-fp32 kernels and
 * fails with OZ_ERR_STATE when max |d pi| or max |d v| exceeds 8e-6 (a network that amplifies rounding beyond what 22 of fp32's 24 bits hold
 * within 1e-5: a conditioning problem no range guard can see; healthy networks measure <= 4e-6, of which up to 3e-6 is the fp32 kernels' own
 * rounding); 0 = off, 2 = measure only; oz_net_self_check reads what the last commit measured */
#define OZ_NET_OPT_SELF_CHECK 4
#define OZ_NET_OPT_W_TARGET_LOG2 5
/* diagnostics switch, per network (default 0), precision f32: the 3x3 convolutions of large batches never take the 256 x 256 tile (GmBig) and run
 * on the 128 x 128 one (GmStd) like every other layer.  Both tiles add every output element's products in the same order: bit-identical
 * results -- the screen test_f32_big_tile_bit_identical_to_the_standard_tile compares them.  Takes effect at the next forward. */
#define OZ_NET_OPT_F32_STD_TILE 6
/* precision f16x2, medium networks (32 < max_batch < ~1000), default 0: 1 = the k-splits of the 3x3 convolutions are chosen for LATENCY (a cost
 * model of grid rounds x k-tiles per block + reduce slabs) instead of "the fewest slices that fill 192 blocks".  For callers whose batches are
 * usually far smaller than max_batch -- an arena with the library's de-duplication and evaluation cache evaluates ~15 leaves per step, and a
 * launch of one-slice 144-k-tile blocks takes 205 us whatever it holds: bench.py's config5, 31 -> 45 games/s -- at a price for full batches
 * (407 -> 430 us per 512-leaf step).  A per-network constant: a position's (pi, v) does not depend on the size of the call; two networks with
 * different settings agree to rounding.  Takes effect at the next oz_net_commit. */
#define OZ_NET_OPT_LATENCY_SPLITS 7
/* diagnostics switch, per network (default 0), precision f16x2, max_batch > 32: conv3 runs on the row tile of this height (128, 192 or 256)
 * instead of the one the forward picks for the call (fewest grid rounds x tile height).  All three add every output element's products in the
 * same order: bit-identical results -- test_conv3_tiles_bit_identical compares them.  Takes effect at the next forward. */
#define OZ_NET_OPT_CONV3_TILE 8
/* diagnostics switch, per network (default 1), precision f16x2: main loop of the 128 x 256 tile (conv3 of calls of <= 455 leaves, conv4 and fc1 of
 * medium networks): 1 = one phase per k-tile on three LDS stages (round 6), 2 = the 2-phase loop on two stages (round 5).  Bit-identical results. */
#define OZ_NET_OPT_LOW_LOOP_PHASES 9
/* diagnostics switch, per network (default 0), precision bf16x3: the tile of conv3 / conv4 -- 0 = the launcher picks (256 x 256 where the network's
 * capacity fills the chip on it, else 128 x 256), 128 / 256 = that tile.  Bit-identical results: the screen of the two tiles. */
#define OZ_NET_OPT_B3_TILE 10
int oz_net_set_option(oz_net* net, int option, int value);
int oz_net_self_check(oz_net* net, double* max_dpi, double* max_dv, int* positions);
/* precision f16x2: the exponents chosen at the last commit.  which = 0 .. 4: per-channel activation exponents of the conv1, conv2, conv3,
 * conv4 (channels each) and fc1 (1024) outputs; which = 5 .. 9: per-column weight exponents of conv2, conv3, conv4 (channels), fc1 (1024), fc2 (512) */
int oz_net_get_scaling(oz_net* net, int which, int32_t* out, int64_t nelem);
/* launch facts of the last forward: OZ_NET_INFO_CONV3_TILE_ROWS = the row-tile height conv3 ran on.  Precision f16x2: 256 / 192 / 128, chosen per call
 * from the capacity the caller launches with (256 at bench.py's batch cap of 3640 leaves; 128 when the call fits one grid round on that tile, e.g.
 * an arena's ~430-leaf batches), 128 x 128 tiles on the latency path (max_batch <= 32).
 * Precision f32: what oz_gemm_f32_launch really launched -- 256 (GmBig: a 3x3 convolution whose 256 x 256 tiles fill the chip, e.g. every call
 * of a max_batch = 4096 network), 128 (GmStd: smaller networks, or OZ_NET_OPT_F32_STD_TILE), 64 (the weight-stream kernel of layers with at
 * most 64 rows: one-position networks). */
#define OZ_NET_INFO_CONV3_TILE_ROWS 1
/* precision f16x2: the guard bits (1 = an activation above the fp16 range, 4 = a low row) raised ON THE CALIBRATION POSITIONS by the self-check of the
 * last oz_net_commit -- 0 for a healthy network.  With OZ_NET_OPT_SELF_CHECK = 1 such a commit fails; with 2 (measure only) it succeeds and this
 * says what happened.  The device flag is cleared by the commit that reported it. */
#define OZ_NET_INFO_SELF_CHECK_GUARD 2
/* the arithmetic the network's GEMM layers really run in: the precision mode, except that a precision-2 (bf16x3) network of max_batch < 128 reports 0 --
 * its layers are the exact-fp32 latency kernels (see oz_net_set_precision) */
#define OZ_NET_INFO_ARITHMETIC 3
int oz_net_get_info(oz_net* net, int what, int* value);

/* ------------------------------------------------------------------ search
 * OthelloMCTS / MCTS (othelo_mcts.py:9-88, MCTS/__init__.py:19-187): num_games independent
 * instances, one wavefront per instance, tables resident in HBM. */
typedef struct oz_mcts oz_mcts;
/* node_cap: states per instance (each a fixed-stride record: header + one 24-byte edge per legal move, 1024 B on 8x8).
 * (oz_version 200: the `edge_cap` arguments / field of version 100 are gone -- the edges live inside the node records since round 2) */
int oz_mcts_create(oz_mcts** out, int n, int num_games, int node_cap, double c, int q_mode);
int oz_mcts_destroy(oz_mcts* m);
int oz_mcts_reset(oz_mcts* m, int game /* -1 = all */);                 /* fresh OthelloMCTS() */
/* cross-game leaf de-duplication of oz_mcts_simulate's batches (default on; results are identical either way) */
int oz_mcts_set_dedup(oz_mcts* m, int enable);
/* roots of the next simulations: canonical boards (othelo_mcts.py:22-26); active[i]=0 leaves slot i idle */
int oz_mcts_set_roots(oz_mcts* m, const uint64_t* own, const uint64_t* opp, const uint8_t* active);
/* OthelloMCTS.simulate x nsims for every active slot, leaves evaluated in one batch per step by `net` */
int oz_mcts_simulate(oz_mcts* m, oz_net* net, int nsims);
/* the same split for a host-side evaluator (any Python object with .predict):
 *   select -> leaves -> [caller evaluates OZ_LEAF_EVAL slots] -> backup */
int oz_mcts_select(oz_mcts* m);
int oz_mcts_leaves(oz_mcts* m, int32_t* status, uint64_t* own, uint64_t* opp);
int oz_mcts_backup(oz_mcts* m, const float* pi /* [num_games][n*n] */, const float* v /* [num_games] */);
/* value returned by the last simulate() of each slot and its dynamic type (0 int, 1 float32, 2 float64) */
int oz_mcts_last_value(oz_mcts* m, double* value, int32_t* vtype, int32_t* depth);
/* N(state, action) of the current root for every square (MCTS/__init__.py:73-84): counts[g][64];
 * rc[g]: 0 ok, 1 root unknown (all zero), 2 KeyError (root never selected from) */
int oz_mcts_root_counts(oz_mcts* m, int32_t* counts, uint64_t* legal, int32_t* rc);
/* get_policy_action_probabilities (othelo_mcts.py:51-67) of every game's current root: policy[g] = float64 (n, n) row-major.
 * temperature != 0: N ** (1 / T) on the legal squares over their np.sum (NumPy's summation order; or 1 if it is 0); temperature == 0: one-hot of
 * a best square, of the max-count squares in row-major order the one tie_draws[g] % (how many) picks (random.choice in the reference; null: the
 * first).  rc[g] as in oz_mcts_root_counts; rc 2 (KeyError in the reference) leaves a zero row. */
int oz_mcts_policy(oz_mcts* m, double temperature, const uint64_t* tie_draws, double* policy, int32_t* rc);
/* table inspection (parity tests): nodes of slot `game` in expansion order */
int oz_mcts_num_nodes(oz_mcts* m, int32_t* num_nodes /* [num_games] */);
int oz_mcts_dump_node(oz_mcts* m, int game, int index, uint64_t* own, uint64_t* opp, int32_t* Ns, uint64_t* legal,
                      int32_t* N /*64*/, double* Q /*64*/, uint8_t* qtag /*64: 1 = float32-typed*/, double* P /*64*/);
/* counters since creation: [0] simulations [1] node visits [2] expansions [3] terminal hits [4] uniform-prior fallbacks */
int oz_mcts_stats(oz_mcts* m, int64_t* out5);

/* ------------------------------------------------------------------ self-play
 * execute_episode (training.py:26-72) for num_games concurrent games in lock step. */
typedef struct oz_selfplay oz_selfplay;
typedef struct {
    int32_t n;              /* board size 4/6/8 */
    int32_t num_games;      /* concurrent game slots */
    int32_t sims;           /* num_simulations per move (>= 2) */
    int32_t q_mode;         /* OZ_QMODE_* */
    double c;               /* degree_exploration */
    double temperature;     /* policy_temperature: 0 -> max-visit with stream tie-break, else first max of N */
    double e_greedy;        /* coin <= e_greedy -> greedy */
    uint64_t seed;          /* RNG streams keyed (seed, global game id, ply) */
    uint64_t first_game_id; /* global id of slot 0 (multi-GPU sharding: rank*num_games) */
    uint64_t game_id_stride;/* id step when a slot is refilled (world_size*num_games) */
    int32_t refill;         /* 1: a finished slot immediately starts a new game */
    int32_t node_cap;       /* per-game node table capacity (0 = sims*61+64) */
    int32_t reserved0;      /* 0 */
    int32_t record_cap;     /* move records kept for export (0 = num_games*64*4) */
    int32_t dedup;          /* OZ_DEDUP_*: cross-game leaf de-duplication -- a board that several games reach in the same batch is evaluated
                             * once and every one of them reads the same (pi, v) row; changes no record, count or statistic, only leaves_evaluated */
    int32_t batch_cap;      /* free-running driver: leaves per network batch (0 = none), see oz_selfplay_set_batch_cap */
    int32_t eval_cache;     /* 1: leaves whose board is in the network's evaluation cache (oz_net_set_eval_cache) take their (pi, v) from it and
                             * need no batch slot; evaluated leaves are inserted.  0 (default): every leaf is evaluated by the network */
    int32_t reserved;
} oz_selfplay_config;
#define OZ_DEDUP_DEFAULT 0  /* = on */
#define OZ_DEDUP_ON 1
#define OZ_DEDUP_OFF 2

/* one move of one game; 8-fold symmetry expansion (training.py:13-23) happens in oz_examples_expand */
typedef struct {
    uint64_t black, white;  /* absolute board BEFORE the move (per-move snapshot) */
    uint64_t final_black, final_white; /* board at the end of that game (the reference's aliased view, T2) */
    uint64_t game_id;
    uint8_t ply;
    uint8_t action;         /* sq = row*8+col */
    int8_t player;          /* mover: +1 BLACK, -1 WHITE */
    int8_t z;               /* +1 if winner == mover else -1 (draw -> BLACK wins) */
    uint8_t greedy;         /* 1 = greedy branch of the coin */
    uint8_t pad[3];
} oz_record;

typedef struct {
    int64_t simulations, node_visits, expansions, terminal_hits, fallbacks;
    int64_t moves, games_completed, records;
    int32_t live_games, overflow;
    int64_t leaves_evaluated;   /* positions handed to the network: <= expansions, because a board that several games reach in
                                 * the same step is evaluated once (results are identical either way; oz_selfplay_config.dedup = OZ_DEDUP_OFF disables) */
} oz_selfplay_stats;

int oz_selfplay_create(oz_selfplay** out, const oz_selfplay_config* cfg, oz_net* net);
int oz_selfplay_destroy(oz_selfplay* sp);
/* `rounds` move rounds: every live game runs cfg.sims simulations, chooses, records and plays one move.
 * Asynchronous (stream ordered); oz_selfplay_sync waits. */
int oz_selfplay_run(oz_selfplay* sp, int rounds);
/* free-running form of oz_selfplay_run: `steps` network batches; in each of them every live game first plays its move if the
 * simulations of the move are complete, runs the simulations that need no network (finished boards) and contributes the
 * leaf of its next first-visit simulation -- batches stay full instead of ~92 % full, moves are no longer aligned across
 * games, every game's simulations / moves / records are exactly those of oz_selfplay_run (training.py:39-67). */
int oz_selfplay_run_steps(oz_selfplay* sp, int steps);
/* batch cap of the free-running driver (0 = none, the default): a network batch holds at most `cap` leaves; a game whose leaf finds no
 * slot keeps it (OZ_LEAF_WAIT) and offers it again in the next batch -- slots are handed out in game order from a start that rotates by
 * `cap` games per batch, so every game is served.  A game's simulations / moves / records do not change; what changes is the size of the
 * launches: the convolution grids are a whole number of rounds of the chip at the right cap (4096 8x8 games on the 512-filter network:
 * cap 3640 = 1024 conv3 tiles of 256 x 256 = 4.0 rounds of 256 CUs, against 5.5 rounds paid as 6 without it). */
int oz_selfplay_set_batch_cap(oz_selfplay* sp, int cap);
/* cross-game leaf de-duplication on / off from the next batch on (oz_selfplay_config.dedup sets the initial state) */
int oz_selfplay_set_dedup(oz_selfplay* sp, int enable);
int oz_selfplay_sync(oz_selfplay* sp);
/* continuous self-play (cfg.refill): bring a fresh engine to the steady state of a long-running one before measuring it --
 * slot g is advanced (g * P) / num_games plies into its first game, P = n*n - 4, by searched self-play moves at `sims_pre`
 * simulations each (same kernels, same RNG streams, recorded like any move), so that every later move round completes
 * about num_games / P games instead of none for P - 1 rounds and all of them in one.  First driver call only; asynchronous. */
int oz_selfplay_stagger(oz_selfplay* sp, int sims_pre);
/* HIP-event timing of the tree kernels on the launch stream: slots 0 select 1 leaf compaction 2 evaluator (all network
 * launches) 3 expand + backup 4 roots + move; slot 2 is always timed (oz_selfplay_eval_time), the others while enabled */
#define OZ_TREE_KERNELS 5
int oz_selfplay_profile(oz_selfplay* sp, int enable);
int oz_selfplay_profile_read(oz_selfplay* sp, double* ms_total /* [OZ_TREE_KERNELS] */, int64_t* launches /* [OZ_TREE_KERNELS] */, int reset);
int oz_selfplay_get_stats(oz_selfplay* sp, oz_selfplay_stats* out);
/* per-slot view: boards, player to move, finished flag, plies played, global game id */
int oz_selfplay_state(oz_selfplay* sp, uint64_t* black, uint64_t* white, int8_t* player, uint8_t* finished,
                      int32_t* ply, uint64_t* game_id);
/* records of COMPLETED games, in completion order; returns how many were written */
int oz_selfplay_records(oz_selfplay* sp, oz_record* out, int64_t max_records, int64_t* written);
/* same, device to device, for the RCCL all-gather (dst = device pointer, e.g. a torch tensor) */
int oz_selfplay_records_device(oz_selfplay* sp, void* dst_device, int64_t max_records, int64_t* written);
/* root visit counts of the last move round, counts[num_games][64] (parity tests) */
int oz_selfplay_last_counts(oz_selfplay* sp, int32_t* counts);
/* HIP-event time of the evaluator (NN) launches since creation, and their count */
int oz_selfplay_eval_time(oz_selfplay* sp, double* ms_total, int64_t* launches, int64_t* leaves);

/* ------------------------------------------------------------------ exchange step (multi-GPU; SURVEY.md 8(b) gather_examples(comm), 8(e))
 * One process per GPU, games sharded by global id (first_game_id / game_id_stride), weights replicated; the path's ONE collective is
 * the all-gather of the 48-byte move records of completed games -- RCCL over xGMI, bound at run time (librccl.so.1).  Replaces
 * WorkerManager.get_results' list concatenation and the ssh / pickle return path (workers.py:147-159,180-184).
 * Rank 0 makes an id (oz_comm_unique_id), the host hands its 128 bytes to every rank, every rank calls oz_comm_create on ITS device
 * (oz_set_device first), then oz_selfplay_gather_records collectively. */
typedef struct oz_comm oz_comm;
#define OZ_COMM_ID_BYTES 128
int oz_comm_unique_id(uint8_t* id /* [OZ_COMM_ID_BYTES] */);
int oz_comm_create(oz_comm** out, const uint8_t* id, int rank, int world);
int oz_comm_destroy(oz_comm* comm);
/* COLLECTIVE over `comm`: the records [first_record, completed so far) of every rank's engine, concatenated in rank order, into `out`
 * (host buffer of max_records); *written = their number, per_rank[world] (optional) = what each rank contributed.  Room is checked against
 * the pooled count on EVERY rank's behalf before the payload moves: too little room on any rank fails the call on all of them together
 * (OZ_ERR_ARG), never on one rank alone.  out == NULL and max_records == 0 on every rank: the counts only (*written = the pooled number). */
int oz_selfplay_gather_records(oz_selfplay* sp, oz_comm* comm, int64_t first_record, oz_record* out, int64_t max_records, int64_t* written,
                               int64_t* per_rank);

/* ------------------------------------------------------------------ arena
 * duel_between_agents with two NeuralNetworkOthelloAgent (agents.py:44-84): net_a = BLACK, net_b = WHITE,
 * one OthelloMCTS per agent per game, temperature 0, ties broken by the RNG_TIE stream.
 * One of net_a / net_b may be NULL: that colour is played by RandomOthelloAgent (agents.py:20-24; the evaluation games of
 * main.py:163-233), its random.choice drawn from the RNG_TIE stream at that ply. */
typedef struct oz_arena oz_arena;
int oz_arena_create(oz_arena** out, int n, int num_games, int sims, double c, int q_mode, uint64_t seed,
                    uint64_t first_game_id, oz_net* net_a, oz_net* net_b, int node_cap);
int oz_arena_destroy(oz_arena* a);
int oz_arena_run(oz_arena* a);       /* plays all games to the end (synchronous) */
/* the same, stopping after `max_rounds` further rounds (0 = to the end): a round = one searched ply in every live game.  Synchronous; may
 * be called again to play on.  bench.py's config-5 leg times a bounded number of plies at 800 sims with two real networks. */
int oz_arena_run_rounds(oz_arena* a, int max_rounds);
/* search counters of the BLACK / WHITE agent since creation, layout of oz_mcts_stats */
int oz_arena_stats(oz_arena* a, int64_t* black5, int64_t* white5);
/* cross-game leaf de-duplication of both agents' searches (default on; identical results either way) and the positions the two
 * networks have evaluated so far (<= expansions when concurrent games share boards -- arena games start from one opening) */
int oz_arena_set_dedup(oz_arena* a, int enable);
int oz_arena_leaves_evaluated(oz_arena* a, int64_t* black, int64_t* white);
/* the two agents' leaves go through their networks' persistent evaluation caches (oz_net_set_eval_cache; default 0 = every leaf is evaluated):
 * identical moves, boards and results -- the reference's per-search _predict_cache (othelo_mcts.py:13,82-88) across games, plies and steps */
int oz_arena_set_eval_cache(oz_arena* a, int enable);
/* HIP-event timing of the two agents' tree kernels on the launch stream, slots of oz_selfplay_profile (0 select 1 leaf compaction 2 evaluator = all
 * network launches 3 expand + backup 4 move), summed over both searches; the networks' own kernels: oz_net_profile on net_a / net_b */
int oz_arena_profile(oz_arena* a, int enable);
int oz_arena_profile_read(oz_arena* a, double* ms_total /* [OZ_TREE_KERNELS] */, int64_t* launches /* [OZ_TREE_KERNELS] */, int reset);
int oz_arena_results(oz_arena* a, int8_t* winner /* +1 net_a */, int32_t* points, int32_t* n_moves,
                     uint8_t* actions /* [num_games][128] */, int8_t* players /* [num_games][128] */,
                     uint64_t* final_black, uint64_t* final_white);

/* ------------------------------------------------------------------ examples
 * training_example_symmetries (training.py:13-23) + the returned tuple layout of execute_episode (:58-72):
 * for each record 8 examples in the reference's order; boards[count*8][n][n][2] uint8 {0,1},
 * policy_index[count*8] (one-hot position row*n+col), z[count*8].
 * alias_final != 0 reproduces the reference's aliasing quirk (boards show the final position). */
int oz_examples_expand(const oz_record* records, int64_t count, int n, int alias_final, uint8_t* boards,
                       int32_t* policy_index, int8_t* z);
int oz_symmetry_table(int n, int32_t* perm /* [8][n*n] source index of every output cell */);

/* ------------------------------------------------------------------ training step
 * NNetWrapper.train (Net/NNet.py:53-68) = keras Model.fit on Net/OthelloNN.py:42-56 / Net/BaseNN.py:41-57:
 * losses categorical_crossentropy (on the (n, n)-reshaped policy: per-row renormalisation, see oz_train.hip) +
 * mean_squared_error, Adam(lr, clipvalue) in tf.keras' formulation, BatchNormalization in training mode
 * (momentum bn_momentum, eps 1e-3), inverted Dropout with a counter-based mask keyed (seed, step, layer, element).
 * Weights use the same 40-array get_weights() indexing as oz_net_set_weight.  One optimiser step =
 * oz_trainer_forward_backward (gradients of the batch-mean loss into the gradient arena) + oz_trainer_apply; a
 * data-parallel job all-reduces (averages) the arena between the two calls.  `external_grads` may point to a caller-
 * owned device buffer of oz_trainer_arena_size floats (e.g. a torch tensor handed to RCCL); NULL = library-owned. */
typedef struct oz_trainer oz_trainer;
int oz_trainer_arena_size(int n, int channels, int in_channels, int64_t* nelem);
int oz_trainer_create(oz_trainer** out, int n, int channels, int in_channels, int max_batch, float lr, float clipvalue /* <= 0: none */,
                      float dropout, float bn_momentum, uint64_t seed, float* external_grads);
int oz_trainer_destroy(oz_trainer* t);
/* arithmetic of the 3x3 layers' forward and data-gradient GEMMs: 0 = fp32 matrix cores (default), 1 = f16x2 -- every fp32 value as two
 * fp16 planes, 3 fp16 MFMA products per fp32 product with fp32 accumulation (the inference kernels of precision f16x2; tensors
 * are moved into the fp16 range by exact powers of two taken from their own maxima on the device, per step).  Weight gradients,
 * dense layers, BN, losses and Adam stay fp32.  Needs channels % 256 == 0; an activation above 65504 raises OZ_ERR_STATE at the
 * next synchronising call (forward_backward, fit_epoch). */
int oz_trainer_set_precision(oz_trainer* t, int mode);
int oz_trainer_set_weight(oz_trainer* t, int index, const float* data, int64_t nelem);
int oz_trainer_get_weight(oz_trainer* t, int index, float* data, int64_t nelem);
int oz_trainer_get_grad(oz_trainer* t, int index, float* data, int64_t nelem);      /* trainable arrays only */
int oz_trainer_grad_arena(oz_trainer* t, void** device_ptr, int64_t* nelem);
/* boards as bitboards: own = channel 0, opp = channel 1 of the example board (BaseNN: +1 / -1 squares);
 * pi_target [B][n*n], z_target [B]; losses3 = {total, policy, value} batch means.  Synchronous. */
int oz_trainer_forward_backward(oz_trainer* t, const uint64_t* own, const uint64_t* opp, const float* pi_target,
                                const float* z_target, int B, float* losses3);
int oz_trainer_apply(oz_trainer* t);                         /* Adam step + BN moving-statistics commit (stream-ordered) */
/* keras Model.fit as the reference drives it (Net/NNet.py:67) with the examples RESIDENT on the device: one upload per fit
 * (set_dataset: own / opp / pi_target [N][n*n] / z_target [N]), then per epoch ONE call that runs the optimiser steps of the
 * shuffled order (`order`: `count` example indices, batches of `batch`, the last one may be short) back to back on the
 * stream -- per-step batch gather on the device, no host copy or synchronisation per step -- and returns the
 * sample-weighted mean losses of the epoch {total, policy, value}.  Single-process training; a data-parallel job keeps
 * the step-wise calls (its all-reduce sits between backward and apply). */
int oz_trainer_set_dataset(oz_trainer* t, const uint64_t* own, const uint64_t* opp, const float* pi_target, const float* z_target, int64_t N);
int oz_trainer_fit_epoch(oz_trainer* t, const int32_t* order, int64_t count, int batch, float* losses3);
int oz_trainer_outputs(oz_trainer* t, int B, float* p /* [B][n*n] */, float* v /* [B] */);   /* of the last forward pass */
/* post-activation output of block `layer` (0-3 conv, 4-5 dense; [B][pixels][channels]) of the last forward pass -- inspection */
int oz_trainer_get_activation(oz_trainer* t, int layer, int B, float* data, int64_t nelem);
int oz_trainer_sync(oz_trainer* t);
int oz_trainer_step_count(oz_trainer* t, int64_t* step);

/* ------------------------------------------------------------------ diagnostics
 * device arithmetic behind the PUCT / backup formulas (MCTS/__init__.py:68,168-170), for bit-exact
 * comparison with the host: sqrt(a), a/b in float64; a/b and (a*b+a)/b in float32 (no FMA contraction). */
int oz_selftest_arith(const double* a, const double* b, int count, double* sqrt_a, double* div_ab, float* fdiv_ab,
                      float* fchain);
/* what the matrix pipe of the current device sustains right now: a pure-MFMA loop (no LDS, no loads, no barriers, one wave per SIMD) for
 * about target_ms milliseconds.  kind 0 = v_mfma_f32_32x32x2_f32 (precision f32's instruction), 1 = v_mfma_f32_16x16x32_f16 on operands with
 * busy mantissas (precision f16x2's), 2 = v_mfma_f32_16x16x32_bf16 on operands with random 7-bit mantissas (precision bf16x3's planes).  tflops = issued FLOP / HIP-event time; clock_ghz (optional) = the clock at which back-to-back issue
 * gives that rate; ms_measured (optional).  bench.py reports both kinds as `device_calibration`: the number that separates a slow or
 * power-capped box from a regression of the kernels. */
int oz_selftest_mfma_rate(int kind, double target_ms, double* tflops, double* clock_ghz, double* ms_measured);
/* the three-plane split of precision bf16x3 as the device evaluates it (oz_net_b3.h, b3_split): for x[i], planes[3 i .. 3 i + 2] (optional) = (b1, b2, b3)
 * widened to fp32 and sum[i] = (b1 + b2) + b3 in fp32 -- equal to x[i] bit for bit for every finite x with |x| >= 2^-100 (the claim the mode rests on;
 * tests/test_gpu_parity.py::test_bf16x3_split_is_exact).  Host buffers; count <= 2^28. */
int oz_selftest_b3_split(const float* x, int64_t count, float* planes, float* sum);

#ifdef __cplusplus
}
#endif
#endif
