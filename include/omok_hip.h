/*
 * omok_hip.h -- C ABI of libomok_hip.so, the MI355X-native AlphaZero self-play engine for Omok.
 *
 * The reference (reinforcement-learning-kr/alpha_omok) is pure Python and has no FFI; this is
 * the boundary a maintainer binds with ctypes. Each entry point names the
 * reference interface it replaces (paths relative to /root/reference/2_AlphaOmok/).
 *
 * Conventions: every function returns 0 on success, non-zero on failure (ao_last_error() gives
 * the message). Buffers are caller-owned. "dev" pointers are HIP device pointers on the handle's
 * device (e.g. torch.Tensor.data_ptr()); "host" pointers are ordinary memory. A handle is not
 * thread-safe; all its work is queued on one HIP stream (ao_stream()).
 *
 * A handle owns G concurrent games. Game g has: its move list (the reference's node id without
 * the leading 0), its search tree (structure-of-arrays arena in HBM), and its own numpy-legacy
 * MT19937 stream (the reference's process-global np.random, main.py:60, one per game here).
 */
#ifndef OMOK_HIP_H
#define OMOK_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AO_ABI_VERSION 2   /* 2: ao_config.arena_fraction (round 6) */

typedef struct ao_engine ao_engine; /* search engine: G games                       */
typedef struct ao_net ao_net;       /* policy/value ResNet (model.py PVNet) weights  */

typedef struct ao_config {
    int32_t board;      /* board edge: 3, 9 (env_small.py:18) or 15 (env_regular.py), any 3..15 */
    int32_t win_mark;   /* 0 = reference rule: 3 if board == 3 else 5 (agents.py:46); 1..5, or above
                           the board size (no line wins: only the full board ends a game)            */
    int32_t sims;       /* num_mcts (agents.py:43; main.py:27 N_MCTS = 400)                    */
    int32_t inplanes;   /* IN_PLANES = 2*history+1 (main.py:34); 3, 5, 7 or 9                  */
    int32_t games;      /* G concurrent games (1 for a drop-in ZeroAgent)                      */
    int32_t noise;      /* Dirichlet root noise on/off (agents.py:40,49)                       */
    int32_t node_cap;   /* expanded-node capacity of one game's arena; 0 = default: 16*(sims+1), bounded by 40 % of
                           the device's TOTAL memory for all games' arenas and never below 4*(sims+1) -- a deterministic
                           number per device model; -1 = grow into the HBM that is free now (a quarter of it, at most
                           16*(sims+1)); see ao_trim_stats, ao_node_cap                                       */
    int32_t device;     /* HIP device ordinal                                                  */
    double  c_puct;     /* 0 = 5 (agents.py:48)                                                */
    double  alpha;      /* 0 = 10/board^2 (agents.py:47)                                       */
    double  arena_fraction; /* node_cap == 0 only: the share of the device's TOTAL memory the default rule may give the two
                           arenas of all games; 0 = 0.40 (main.py's over-subscribed engines ask for 0.50); (0, 0.90]. The
                           default path keeps its shrink-on-allocation-failure retry and its clamps (ABI version 2)   */
} ao_config;

/* root status reported by ao_set_root / ao_begin_move (agents.py:82-111) */
enum { AO_ROOT_FRESH = 0,       /* id not in the tree: num_mcts+1 simulations                  */
       AO_ROOT_UNEXPANDED = 1,  /* id known (n == 0, no children): num_mcts simulations        */
       AO_ROOT_EXPANDED = 2 };  /* id known and expanded: children re-noised, num_mcts sims    */

const char *ao_version(void);
int         ao_abi_version(void);

/* ---- engine lifetime ---- replaces ZeroAgent.__init__ (agents.py:40-53) */
int  ao_create(const ao_config *cfg, ao_engine **out);
void ao_destroy(ao_engine *e);
const char *ao_last_error(const ao_engine *e);   /* e may be NULL: error of the failed ao_create */
void *ao_stream(ao_engine *e);                   /* hipStream_t all engine work is queued on     */
int  ao_sync(ao_engine *e);                      /* hipStreamSynchronize                          */
/* Queue all further engine work on `stream` (a hipStream_t, e.g. torch's current stream) so an
 * external evaluator running on that stream needs no extra synchronisation. NULL restores the
 * engine's own stream. */
int  ao_set_stream(ao_engine *e, void *stream);

/* ---- per-game RNG ---- replaces np.random.seed / get_state / set_state (main.py:60).
 * ao_seed == np.random.seed(seed) for game g (MT19937 init_genrand, pos 624, no cached gauss). */
int ao_seed(ao_engine *e, int game, uint32_t seed);
int ao_seed_all(ao_engine *e, const uint32_t *host_seeds /*[G]*/);
/* ao_seed for the n listed games with ONE synchronisation (the refill of finished self-play slots: main.py:248 + np.random.seed per episode) */
int ao_seed_games(ao_engine *e, const int32_t *host_games, const uint32_t *host_seeds, int32_t n);
int ao_get_rng_state(ao_engine *e, int game, uint32_t *host_mt /*[624]*/, int32_t *pos,
                     int32_t *has_gauss, double *gauss);
int ao_set_rng_state(ao_engine *e, int game, const uint32_t *host_mt, int32_t pos,
                     int32_t has_gauss, double gauss);

/* ---- game / tree state ---- */
/* ZeroAgent.reset() (agents.py:55-58) for the games with mask[g] != 0 (NULL = all): clears the
 * tree and the move list (root id becomes (0,)). */
int ao_reset(ao_engine *e, const uint8_t *host_mask);
/* The root_id argument of ZeroAgent.get_pi (agents.py:60,82-84): moves = root_id[1:].
 * Re-roots the tree kept from the previous search if the new id extends the previous root id and
 * the node exists (tree reuse); otherwise the game starts a fresh tree. The engine keeps the
 * subtree of the last root only (what main.self_play / eval_main.del_parents ever revisit). */
int ao_set_root(ao_engine *e, int game, const int32_t *host_moves, int32_t n, int32_t *status);
/* The same for every game with mask[g] != 0 (NULL = all) in ONE launch: moves[g * stride .. + n[g]) is game g's
 * root_id[1:]; status[g] receives AO_ROOT_* (entries of unmasked games are left alone). This is the call pattern of
 * eval_main.Evaluator.get_action (eval_main.py:137-151, 243-252) for G concurrent matches: after the opponent's
 * reply every match's id has grown by two plies that this engine did not choose. */
int ao_set_roots(ao_engine *e, const uint8_t *host_mask, const int32_t *host_moves, int32_t stride,
                 const int32_t *host_n, int32_t *host_status);

/* ---- one move decision, stepwise (external evaluator) ---- replaces _init_mcts/_mcts
 * (agents.py:82-132). Protocol per move:
 *     ao_begin_move -> repeat ao_sims_left() times { ao_collect_leaves -> evaluate ->
 *     ao_apply_evals } -> ao_end_move
 * active[g] == 0 leaves game g untouched for this move (NULL = all games active). */
int ao_begin_move(ao_engine *e, const uint8_t *host_active);
int ao_sims_left(ao_engine *e);
/* _selection (agents.py:134-168) for every game with simulations left, + get_state_pt of the
 * leaf (utils.py:139-168). dev_planes_nchw: optional float32 [G][C][B][B] (the layout
 * Agent.model expects, agents.py:175); NULL if only the engine's own network is used. */
int ao_collect_leaves(ao_engine *e, float *dev_planes_nchw);
/* _expansion_evaluation + _backup (agents.py:170-239) with the evaluator's outputs:
 * dev_policy float32 [G][A] (softmax probabilities), dev_value float32 [G]. Rows of games whose
 * leaf was terminal (or that are inactive) are ignored. */
int ao_apply_evals(ao_engine *e, const float *dev_policy, const float *dev_value);
/* Tail of get_pi (agents.py:64-80): visit, policy (post-noise priors of the root children) and
 * pi = visit/visit.sum(), one-hot through utils.argmax_onehot where tau[g] == 0.
 * Outputs are host float64 [G][A]; any may be NULL. host_tau int8 [G] (NULL = all 1). */
int ao_end_move(ao_engine *e, const int8_t *host_tau, double *host_pi, double *host_visit,
                double *host_policy);
/* utils.get_action (utils.py:189-195) on game g's stream + env step (env_small.py:154-176,196)
 * + re-rooting on the chosen child, for all active games. host_action int32 [G],
 * host_win int32 [G] (0 playing, 1 black, 2 white, 3 draw). Games that ended keep their final
 * position until ao_reset. Must follow ao_end_move. */
int ao_play(ao_engine *e, int32_t *host_action, int32_t *host_win);

/* ---- one move decision, fused (native network) ---- ZeroAgent.get_pi with Agent.model = net.
 * Runs begin_move, all simulations (select -> PVNet forward -> expand/backup on one stream) and
 * end_move. */
int ao_search(ao_engine *e, ao_net *net, const uint8_t *host_active, const int8_t *host_tau,
              double *host_pi, double *host_visit, double *host_policy);

/* ---- introspection (tests, get_visit/get_policy, del_parents prints) ---- */
int ao_get_moves(ao_engine *e, int game, int32_t *host_moves /*[A]*/, int32_t *n);
/* statistics of the root's children in stored child order: action, n, w, q, p */
int ao_get_root_children(ao_engine *e, int game, int32_t *host_action, double *host_n,
                         double *host_w, double *host_q, double *host_p, int32_t *count);
int ao_tree_nodes(ao_engine *e, int game, int64_t *expanded, int64_t *dict_entries);
/* HIP-event timing of the per-simulation tree kernel of ao_search (k_expand_select: expansion + backup of one
 * simulation, selection + terminal test + plane encoding of the next -- agents.py:134-239 for every game), recorded
 * on the engine's launch stream. Returns the total and the number of TIMED launches since the previous call and
 * enables / disables the timing (bench.py's roofline_tree); enable = n > 1: every n-th launch carries the event pair. */
int ao_tree_timing(ao_engine *e, int enable, double *ms_total, int64_t *launches);
/* Arena pressure, cumulative since ao_create. A game's arena holds node_cap expanded nodes; the tree kept across
 * moves (main.py:171 -> agents.py:84) grows by up to `sims` nodes per move when the visits keep following the played
 * line. Re-rooting therefore keeps at most node_cap - sims - 1 nodes, breadth first: a child subtree beyond that
 * becomes an unvisited child again (its n / w / q are forgotten -- the one place the engine departs from the
 * reference's never-pruned dict, and only in games that hit the limit). subtrees_dropped / reroots_trimmed count the
 * events; both stay 0 while node_cap is large enough (raise ao_config.node_cap otherwise). */
int ao_trim_stats(ao_engine *e, int64_t *subtrees_dropped, int64_t *reroots_trimmed);
/* the arena capacity this engine runs with (ao_config.node_cap after defaults) and whether it was derived from the free
 * HBM at creation (node_cap = -1): what decides if agents.py's never-pruned dict (agents.py:52) is reproduced in full. */
int ao_node_cap(ao_engine *e, int32_t *node_cap, int32_t *from_free_memory);
/* Host threads this PROCESS uses for the per-move np.random.dirichlet replay (agents.py:97-98,194-195 need glibc log / pow,
 * so the draws of all games are made on the host at the start of a move): one persistent pool per process of
 * min(32, hardware threads / LOCAL_WORLD_SIZE) threads, the caller included (LOCAL_WORLD_SIZE as torchrun exports it: the
 * ranks of a node share the host; AO_HOST_THREADS overrides). Needs no device. */
int ao_host_threads(void);
/* ao_search repeats a move on the fp32-MFMA trunk when the split-fp16 trunk met an activation beyond the fp16 range
 * (ao_net_status): the games of that move get their pre-move streams back and fresh trees at their positions, the
 * caller gets the repeated move's result. Counted here since ao_create: moves repeated, games searched again. The
 * network returns to the mode ao_net_set_mode asked for after the repeated move; from the third such move with the SAME
 * weights (counted per ao_net since its last ao_net_finalize) it stays on the fp32-MFMA trunk -- ao_net_get_mode then says
 * 2 -- until new weights are finalized or ao_net_set_mode is called. */
int ao_fp16_range_events(ao_engine *e, int64_t *moves_repeated, int64_t *games_redone);
/* search-shape counters since the last ao_begin_move, summed over games: PUCT levels traversed,
 * k>1 random tie-breaks, terminal leaves, evaluated leaves */
int ao_search_stats(ao_engine *e, int64_t *levels, int64_t *ties, int64_t *terminal,
                    int64_t *evaluated);

/* Rows of the evaluation batch per simulation in ao_search (new; the reference evaluates one leaf at a time and runs the net on
 * terminal leaves only to throw the result away, agents.py:171-178,216-221 -- SURVEY Q9).
 *   rows == 0 (default): the host packs the active games to the front of the batch once per move; a game whose leaf is terminal
 *     keeps its (stale) row in that simulation's batch.
 *   rows > 0: the tree kernel hands out the rows PER SIMULATION -- terminal leaves take none -- and counts the live rows in a
 *     device word the trunk kernels read (groups without a live row exit at once). `rows` bounds the batch of one simulation; with
 *     MORE active games than rows (over-subscription, e.g. 5120 games on the 4096 rows the resident trunk fills the chip with)
 *     a share of the games sits out every launch in turn and a leaf that still finds the batch full is evaluated one launch
 *     later; ao_search runs until every game has its simulations. Every game's search stays strictly sequential: visits,
 *     priors, actions and RNG streams are bit-identical to rows == 0 for a network whose result does not depend on a board's
 *     neighbours in the batch. */
int ao_set_row_cap(ao_engine *e, int32_t rows);
/* since ao_create, over the ao_search calls that handed out rows per simulation: network launches, live rows they evaluated,
 * rows they were launched for (launches x batch capacity), leaves that found their simulation's batch full */
int ao_row_stats(ao_engine *e, int64_t *launches, int64_t *rows_live, int64_t *rows_launched, int64_t *waits);
/* Test / debugging hook: what the network returned to the listed games during ao_search. Network launch s (counted from 0
 * when this function was called, over all ao_search calls since) writes, for listed game k, a record of A + 3 floats to dev_log[(s * n + k) * (A + 3) ...]: the policy row
 * and the value at the game's row of the evaluation batch, the number of simulations the game has completed, and its leaf
 * status (1 / 2: the leaf waits for exactly this evaluation; 3: terminal leaf, no evaluation -- the reference evaluates and
 * discards; 0 / 4 / 5: nothing of this game was evaluated in this launch). Launches beyond capacity_floats are not recorded;
 * n == 0 switches the log off; dev_log must stay valid until then (or until ao_destroy). ao_eval_log_count: the launches
 * recorded so far. */
int ao_set_eval_log(ao_engine *e, const int32_t *host_games, int32_t n, float *dev_log, int64_t capacity_floats);
int ao_eval_log_count(ao_engine *e);

/* ---- policy/value network ---- replaces model.PVNet(...).forward in eval() mode
 * (model.py:76-104). Parameters are given under their state_dict names.
 * planes: a multiple of 32 in 32 .. 512 (model.py:76-85 takes any width; the Python layer zero-pads other widths to the next
 * multiple). 128 planes (the reference's OUT_PLANES, main.py:35) run on the split-fp16 MFMA kernels; 160 .. 512 planes on the
 * row-chunked fp32-MFMA layer kernels for every batch size, whatever ao_net_set_mode asks for. */
int  ao_net_create(int n_block, int inplanes, int planes, int board, int device, ao_net **out);
void ao_net_destroy(ao_net *n);
const char *ao_net_last_error(const ao_net *n);
/* host float32 data in PyTorch layout (conv OIHW, linear [out][in]); num_batches_tracked is
 * accepted and ignored. */
int  ao_net_set_param(ao_net *n, const char *name, const float *host_data, int64_t numel);
int  ao_net_finalize(ao_net *n); /* fold BN running stats, repack weights for the MFMA kernels */
/* forward for `batch` positions. dev_planes_nchw float32 [batch][C][B][B] -> dev_policy
 * [batch][A], dev_value [batch]; `stream` is a hipStream_t (NULL = default stream). */
int  ao_net_forward(ao_net *n, const float *dev_planes_nchw, int batch, float *dev_policy,
                    float *dev_value, void *stream);
/* Trunk execution: 0 = auto (by batch size), 1 = one kernel per 3x3 conv over groups of 32 boards
 * (medium batches), 2 = group-resident trunk: one workgroup carries 16 boards through every conv
 * layer in a single launch (4096 boards = 256 groups = one per CU), 3 = per-board NHWC with the
 * cells as the MFMA N dimension (latency path for a handful of games), 4 = one launch per layer
 * over (16-board group x row chunk) for the batch sizes in between -- all fp32 MFMA --, 5 = the
 * group-resident trunk with the fp32 contraction carried by fp16 MFMAs: every operand is split in
 * two halves (x = xh + xl) and x*w = xh*wh + xh*wl + xl*wh with fp32 accumulation (the dropped
 * xl*wl term is <= 2^-22 of a product; measured error against an fp64 evaluation equals the fp32
 * path's; activations are clamped to the fp16 range, 65504). Mode 5 needs 128 planes and at least one
 * ResBlock; it runs as one resident launch (boards up to 9x9, >= 3072 boards) or as one launch per conv
 * over (16-board group x row chunk x column tile) (any board up to 15x15, smaller batches), and is
 * what mode 0 picks on such a net for batches of more than ~7800 cells (96 9x9 boards; 5400 cells on wider boards). 6 = mode 5 restricted to
 * the per-layer kernels for EVERY batch size: slower at both ends, but the arithmetic that evaluates a position is then
 * the same whatever else shares the batch, so a game's trajectory under ao_search depends on its own stream and moves
 * only (modes 0 / 5 switch kernel families with the number of active games; all within 1e-4 of model.py:76-104). */
int  ao_net_set_mode(ao_net *n, int mode);
int  ao_net_get_mode(const ao_net *n);
/* Status word of the network, read on `stream` (which is synchronised): AO_NET_FP16_RANGE is set when the
 * split-fp16 trunk (modes 0 / 5 at 128 planes) met an activation beyond the fp16 range and clamped it to 65504 --
 * the outputs of such a forward are finite but not the fp32-equivalent evaluation of model.py:76-104. clear != 0
 * resets the word. ao_search checks it after every move and repeats such a move on the fp32-MFMA trunk (mode 2),
 * see ao_fp16_range_events. */
enum { AO_NET_FP16_RANGE = 1 };
int  ao_net_status(ao_net *n, void *stream, int32_t *flags, int clear);
/* total device time (ms) and count of the TIMED launches of the dominant trunk kernel since the last call
 * (HIP events on the launch stream); used by bench.py's roofline. enable = n > 1: every n-th forward is timed. */
int  ao_net_conv_timing(ao_net *n, int enable, double *ms_total, int64_t *launches);
/* MFMA products per multiply-add of the split-fp16 conv kernels (model.py:6-31,97-104: the 3x3 convs of PVNet). The kernels
 * compute x*w = xh*wh + xh*wl + xl*wh on fp16 halves of both operands (three products, fp32 accumulate). When every conv weight
 * of the loaded network, scaled by its layer's power of two, IS an fp16 number -- ao_net_finalize checks, per export -- wl is zero
 * everywhere and the middle product adds exact zeros: the two-product kernels leave it out (same bits, a third fewer MFMAs).
 * request: 0 = two products whenever the weights allow (default), 3 = always three, -1 = query only. in_force: 2 or 3;
 * weights_fp16: 1 when the loaded weights qualify. A checkpoint gets there by keeping its conv weights on the fp16 grid
 * (tools/train_omok.py --fp16-grid-weights); nothing else about the state_dict changes. */
int  ao_net_products(ao_net *n, int32_t request, int32_t *in_force, int32_t *weights_fp16);
/* name and algorithmic FLOPs per launch (2*MAC, zero padding counted) of the kernel the timing
 * refers to, for a batch of `boards` positions. */
int  ao_net_dominant_kernel(ao_net *n, int boards, char *name, int name_cap, double *flop_per_launch);
/* the same answer without a network object or a device (pure planning logic): which kernel would carry the conv
 * stack of a PVNet(n_block, inplanes, planes, board) (model.py:76-85) for `boards` positions, trunk_mode as in
 * ao_net_set_mode, in_kind 1 = fp32 plane batch (ao_net_forward), 2 = the engine's bit planes (ao_search).
 * bench.py and tests/test_host_side.py use it to tie a committed rocprofv3 summary to the kernel that runs. */
int  ao_net_plan_kernel(int n_block, int inplanes, int planes, int board, int trunk_mode, int boards, int in_kind,
                        char *name, int name_cap, double *flop_per_launch);

/* ---- replay memory ---- replaces rep_memory = deque(maxlen=MEMORY_SIZE) (main.py:55), its
 * rep_memory.extend(utils.augment_dataset(cur_memory, board_size)) (main.py:229-231,
 * utils.py:226-239) and the mini-batch assembly of main.train (main.py:262-292). The ring lives in
 * HBM; entries are in deque order (index 0 = oldest), the oldest are dropped when full. */
typedef struct ao_replay ao_replay;
int  ao_replay_create(int board, int inplanes, int64_t capacity, int device, ao_replay **out);
void ao_replay_destroy(ao_replay *r);
const char *ao_replay_last_error(const ao_replay *r);  /* r may be NULL: failed ao_replay_create */
int64_t ao_replay_size(const ao_replay *r);            /* len(rep_memory)                        */
int64_t ao_replay_capacity(const ao_replay *r);        /* rep_memory.maxlen                      */
int  ao_replay_clear(ao_replay *r);
/* Appends n samples given as host arrays: states float32 [n][C][B][B] (utils.get_state_pt planes,
 * 0/1: exact in float32), pi float64 [n][A], z float32 [n]. augment != 0 appends the eight
 * symmetries of every sample in the reference's order (rot90 k = 0..3, each followed by its
 * left-right flip), computed on the device; augment == 0 appends the samples as they are. */
int  ao_replay_extend(ao_replay *r, const float *states, const double *pi, const float *z, int64_t n,
                      int augment, void *stream);
/* the same for a call whose first `skipped` samples are NOT supplied: main.self_play(n) appends every sample of its n
 * games (main.py:229-231), but a deque(maxlen) keeps the newest entries only -- when the supplied n samples alone fill
 * the memory (n * 8 >= capacity with augment), the earlier ones of the same call only move the ring position. */
int  ao_replay_extend_skip(ao_replay *r, const float *states, const double *pi, const float *z, int64_t n,
                           int augment, int64_t skipped, void *stream);
/* Device-side sample emission: the same append as ao_replay_extend_skip, but the states are BUILT ON THE DEVICE from the
 * episodes' move lists instead of being uploaded -- replaces the per-ply utils.get_state_pt calls of main.py:159-166 /
 * 219-227 (utils.py:139-168). moves int16 [n_episodes][max_len] (action indices in playing order, anything outside
 * [0, board * board) is padding), sample i = the position of episode ep_of[i] after ply_of[i] of its moves (the root the
 * search of that ply started from); pi / z / augment / skipped as above. */
int  ao_replay_extend_moves(ao_replay *r, const int16_t *moves, int64_t n_episodes, int64_t max_len,
                            const int32_t *ep_of, const int32_t *ply_of, const double *pi, const float *z,
                            int64_t n, int augment, int64_t skipped, void *stream);
/* Mini-batch for m deque indices (host int64; e.g. random.sample(range(len), m)): writes float32
 * device buffers states [m][C][B][B], pi [m][A], z [m] -- the tensors main.train feeds the net. */
int  ao_replay_gather(ao_replay *r, const int64_t *idx, int64_t m, float *dev_states, float *dev_pi,
                      float *dev_z, void *stream);
/* Reads entries [first, first+n) back to host float64 arrays (any may be NULL); this is what
 * main.save_dataset pickles. */
int  ao_replay_read(ao_replay *r, int64_t first, int64_t n, double *states, double *pi, double *z);

/* ---- rollout agents ---- replace PUCTAgent.get_pi (agents.py:263-441) and UCTAgent.get_pi
 * (agents.py:443-614): net-free searches with random playouts. One handle owns G independent
 * games, each with its own numpy-legacy MT19937 stream; a search runs entirely in one kernel. */
typedef struct ao_rollout ao_rollout;
typedef struct ao_rollout_config {
    int32_t board;     /* board edge 3..15                                                     */
    int32_t win_mark;  /* 0 = reference rule: 3 if board == 3 else 5 (agents.py:270)           */
    int32_t sims;      /* num_mcts; every get_pi runs num_mcts + 1 simulations (agents.py:318) */
    int32_t games;     /* G                                                                    */
    int32_t mode;      /* 0 = PUCTAgent, 1 = UCTAgent                                          */
    int32_t device;
    double  c_puct;    /* 0 = 5 (agents.py:271); PUCT only                                     */
} ao_rollout_config;
int  ao_rollout_create(const ao_rollout_config *cfg, ao_rollout **out);
void ao_rollout_destroy(ao_rollout *r);
const char *ao_rollout_last_error(const ao_rollout *r); /* r may be NULL: failed create          */
int  ao_rollout_seed(ao_rollout *r, int game, uint32_t seed);            /* np.random.seed(seed) */
int  ao_rollout_get_rng_state(ao_rollout *r, int game, uint32_t *mt624, int32_t *pos,
                              int32_t *has_gauss, double *gauss);
int  ao_rollout_set_rng_state(ao_rollout *r, int game, const uint32_t *mt624, int32_t pos,
                              int32_t has_gauss, double gauss);
/* get_pi(root_id, board, turn, tau) for every active game: moves [G][A] (row g = root_id[1:] of
 * game g, first nmoves[g] entries used; board and turn follow from the id). Host outputs
 * (any may be NULL): pi float64 [G][A] one-hot; stat float64 [G][A] = visit counts of the root's
 * children (PUCT, 0 elsewhere) or their q (UCT, -inf elsewhere); action int32 [G]. */
int  ao_rollout_search(ao_rollout *r, const int32_t *moves, const int32_t *nmoves,
                       const uint8_t *active, double *pi, double *stat, int32_t *action);

/* ---- 3x3 UCT of 1_tictactoe_MCTS ---- replaces the per-move search loop of mcts_vs.py:153-183
 * (MCTS.selection / expansion / simulation / backup, mcts_vs.py:15-131; BASELINE configs[0]).
 * Randomness is Python's `random` module stream (MT19937 words + index, random.getstate()[1]). */
typedef struct ao_ttt ao_ttt;
typedef struct ao_ttt_config {
    int32_t board;     /* state_size (env.py: 3); any 3..15                                    */
    int32_t win_mark;  /* 0 = 3 on a 3x3 board, else 5                                         */
    int32_t sims;      /* num_mcts (mcts_vs.py:144: 1500)                                      */
    int32_t games;     /* G independent searches per call                                      */
    int32_t device;
} ao_ttt_config;
int  ao_ttt_create(const ao_ttt_config *cfg, ao_ttt **out);
void ao_ttt_destroy(ao_ttt *r);
const char *ao_ttt_last_error(const ao_ttt *r);   /* r may be NULL: failed create */
int  ao_ttt_seed(ao_ttt *r, int game, uint32_t seed);                       /* random.seed(seed)     */
int  ao_ttt_get_rng_state(ao_ttt *r, int game, uint32_t *mt624, int32_t *pos);   /* random.getstate() */
int  ao_ttt_set_rng_state(ao_ttt *r, int game, const uint32_t *mt624, int32_t pos);
/* boards int8 [G][A] row-major (+1 = O / first player, -1 = X), turns int32 [G] (0 = O to move).
 * Host outputs (any may be NULL): q float64 [G][A] of the root's children (-inf elsewhere), n float64
 * [G][A] their visit counts, action int32 [G] = max_action (first maximum of q). */
int  ao_ttt_search(ao_ttt *r, const int8_t *boards, const int32_t *turns, const uint8_t *active,
                   double *q, double *n, int32_t *action);

#ifdef __cplusplus
}
#endif
#endif
