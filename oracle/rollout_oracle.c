/*
 * rollout_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY) for the net-free rollout agents.
 *
 * Plain-C restatement of
 *   /root/reference/2_AlphaOmok/agents.py:263-441  PUCTAgent (uniform priors + random playout)
 *   /root/reference/2_AlphaOmok/agents.py:443-614  UCTAgent  (UCB1 + random playout)
 *   /root/reference/2_AlphaOmok/utils.py:8-19      valid_actions (ascending cell order)
 *   /root/reference/2_AlphaOmok/utils.py:208-223   get_reward
 * with the numpy legacy RandomState of omok_oracle.c as the process-global np.random stream.
 *
 * What the reference does, as restated here:
 *   - get_pi -> _init_mcts overwrites tree[root_id] with a fresh, childless node, so every call
 *     is a fresh search of num_mcts + 1 simulations (UCTAgent sets is_real_root = True in
 *     _init_mcts, PUCTAgent always runs num_mcts + 1).
 *   - _selection: while node.n > 0: check_win(node board) -> terminal returns (node, win);
 *     total_n = sum of child n; PUCT score q + 5 * p * sqrt(total_n) / (n + 1), UCT score
 *     q + (inf if n == 0 else sqrt(2 * log(total_n) / n)); every exact-equal maximum collected in
 *     child order, ids[np.random.choice(len(ids))]. After the loop check_win(leaf board).
 *   - _expansion_simulation: win_index == 0 -> children for every empty cell in ascending order
 *     (n = w = q = 0., p = 1 / len(actions)); if the leaf has a parent, random playout from the
 *     leaf: actions[np.random.choice(len(actions))] until check_win != 0, reward =
 *     get_reward(win, leaf_id); the root gets reward 0 and no playout. win_index != 0 -> reward 1.
 *   - _backup from the leaf up to and including the root: n += 1; w += reward * (-1)**count;
 *     q = w / n (Python floats).
 *   - get_pi: PUCT one-hot on the most visited child, UCT one-hot on the child with the largest
 *     q (-inf for non-children), ties by np.random.choice.
 * np.log in the UCT score is libm log here; numpy's AVX512 log differs from glibc for the
 * integers 9170, 19143, 94869, ... (first difference at 9170), so parity with a reference
 * run on an AVX512 host holds for num_mcts < 9170.
 *
 * Parity status: PINNED by tests/golden/gv11_rollout_agents.npz (tools/gen_golden.py gv11).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "omok_oracle.h"

typedef struct rnode {
    int parent;      /* -1 for the root */
    int action;      /* move that leads here */
    int first_kid;   /* index of the first child node, -1 if not expanded */
    int nkids;
    double n, w, q;
} rnode;

typedef struct rtree {
    rnode *nodes;
    int used, cap;
} rtree;

static int rt_new(rtree *t, int parent, int action)
{
    if (t->used == t->cap) {
        t->cap = t->cap ? t->cap * 2 : 1024;
        t->nodes = (rnode *)realloc(t->nodes, sizeof(rnode) * (size_t)t->cap);
    }
    rnode *nd = &t->nodes[t->used];
    nd->parent = parent; nd->action = action; nd->first_kid = -1; nd->nkids = 0;
    nd->n = 0.; nd->w = 0.; nd->q = 0.;
    return t->used++;
}

/* utils.get_turn(id): 0 (black) iff len(id) is odd, i.e. iff an even number of stones is down */
static int turn_of(int nstones) { return (nstones % 2 == 0) ? 0 : 1; }

int oo_rollout_search(int mode, int board, int win_mark, int num_mcts, const int *root_moves, int nroot,
                      oo_rng *rng, double *pi, double *stat, int *action_out)
{
    const int A = board * board;
    if (win_mark <= 0) win_mark = (board == 3) ? 3 : 5;
    int8_t *root_board = (int8_t *)malloc((size_t)A);
    int8_t *b = (int8_t *)malloc((size_t)A);
    int8_t *sim = (int8_t *)malloc((size_t)A);
    double *qu = (double *)malloc(sizeof(double) * (size_t)A);
    int *acts = (int *)malloc(sizeof(int) * (size_t)A);
    oo_get_board(root_moves, nroot, board, root_board);
    rtree t = {0, 0, 0};
    const int root = rt_new(&t, -1, -1);

    for (int s = 0; s < num_mcts + 1; s++) {
        /* ---- _selection ---- */
        memcpy(b, root_board, (size_t)A);
        int node = root, stones = nroot, win_index = 0, decided = 0;
        while (t.nodes[node].n > 0) {
            win_index = oo_check_win(b, board, win_mark);
            if (win_index != 0) { decided = 1; break; }
            rnode *nd = &t.nodes[node];
            double total_n = 0;
            for (int i = 0; i < nd->nkids; i++) total_n += t.nodes[nd->first_kid + i].n;
            double max_value = 0;
            for (int i = 0; i < nd->nkids; i++) {
                const rnode *c = &t.nodes[nd->first_kid + i];
                double u;
                if (mode == 0) {
                    const double p = 1.0 / (double)nd->nkids;   /* prior_prob = 1 / len(actions) */
                    double x = 5.0 * p;                         /* self.c_puct * p               */
                    x = x * sqrt(total_n);
                    u = x / (c->n + 1);
                } else if (c->n == 0) {
                    u = INFINITY;
                } else {
                    double x = 2 * log(total_n);
                    x = x / c->n;
                    u = sqrt(x);
                }
                qu[i] = c->q + u;
                if (i == 0 || qu[i] > max_value) max_value = qu[i];
            }
            int cnt = 0;
            for (int i = 0; i < nd->nkids; i++) if (qu[i] == max_value) cnt++;
            int r = (int)oo_rng_below(rng, cnt);
            int pick = -1;
            for (int i = 0; i < nd->nkids; i++)
                if (qu[i] == max_value) { if (r == 0) { pick = i; break; } r--; }
            node = nd->first_kid + pick;
            b[t.nodes[node].action] = (int8_t)(turn_of(stones) == 0 ? 1 : -1);
            stones++;
        }
        if (!decided) win_index = oo_check_win(b, board, win_mark);

        /* ---- _expansion_simulation ---- */
        double reward;
        if (win_index == 0) {
            int na = 0;
            for (int c = 0; c < A; c++) if (b[c] == 0) acts[na++] = c;   /* utils.valid_actions */
            int first = -1;
            for (int i = 0; i < na; i++) {
                const int k = rt_new(&t, node, acts[i]);
                if (i == 0) first = k;
            }
            t.nodes[node].first_kid = first;
            t.nodes[node].nkids = na;
            if (t.nodes[node].parent >= 0) {
                memcpy(sim, b, (size_t)A);
                int turn_sim = turn_of(stones);          /* tree[leaf]['player'] = get_turn(leaf_id) */
                int w;
                for (;;) {
                    int ns = 0;
                    for (int c = 0; c < A; c++) if (sim[c] == 0) acts[ns++] = c;
                    const int a = acts[oo_rng_below(rng, ns)];
                    sim[a] = (int8_t)(turn_sim == 0 ? 1 : -1);
                    w = oo_check_win(sim, board, win_mark);
                    if (w == 0) turn_sim = 1 - turn_sim;
                    else break;
                }
                /* utils.get_reward(win, leaf_id): from the side that moved into the leaf */
                const int turn = turn_of(stones);
                if (w == 1) reward = (turn == 1) ? 1. : -1.;
                else if (w == 2) reward = (turn == 1) ? -1. : 1.;
                else reward = 0.;
            } else {
                reward = 0.;   /* "root node don't simulation" */
            }
        } else {
            reward = 1.;       /* terminal node (draws included) */
        }

        /* ---- _backup ---- */
        int count = 0;
        for (int k = node; k >= 0; k = t.nodes[k].parent) {
            rnode *nd = &t.nodes[k];
            nd->n += 1;
            nd->w += reward * ((count % 2 == 0) ? 1 : -1);
            nd->q = nd->w / nd->n;
            count++;
        }
    }

    /* ---- tail of get_pi ---- */
    const rnode *rt = &t.nodes[root];
    for (int a = 0; a < A; a++) { pi[a] = 0.; stat[a] = (mode == 0) ? 0. : -INFINITY; }
    for (int i = 0; i < rt->nkids; i++) {
        const rnode *c = &t.nodes[rt->first_kid + i];
        stat[c->action] = (mode == 0) ? c->n : c->q;
    }
    double mx = stat[0];
    for (int a = 1; a < A; a++) if (stat[a] > mx) mx = stat[a];
    int cnt = 0;
    for (int a = 0; a < A; a++) if (stat[a] == mx) cnt++;
    int r = (int)oo_rng_below(rng, cnt);
    int pick = -1;
    for (int a = 0; a < A; a++) if (stat[a] == mx) { if (r == 0) { pick = a; break; } r--; }
    pi[pick] = 1.;
    if (action_out) *action_out = pick;
    const int used = t.used;
    free(t.nodes); free(root_board); free(b); free(sim); free(qu); free(acts);
    return used;
}

/* =================================================================================================
 * 1_tictactoe_MCTS/mcts_vs.py (BASELINE configs[0]): the UCT search its __main__ runs per move
 *   MCTS.selection  mcts_vs.py:15-46    q + u, u = 5 * sqrt(2 * ln(N_parent) / n), an unvisited child
 *                                        uses n = 0.0001; strict '>' from -100: the FIRST maximum wins
 *   MCTS.expansion  mcts_vs.py:48-93    only the root or a node with n > 10 is expanded; all children
 *                                        are created, ONE is picked by random.sample(childs, 1)
 *   MCTS.simulation mcts_vs.py:95-111   random.choice(valid_actions) until check_win != 0
 *   MCTS.backup     mcts_vs.py:113-131  value = 0.8 (draw) / +1 (root player wins) / -1, the SAME value
 *                                        added at every node of the path (no sign alternation)
 *   driver          mcts_vs.py:153-183  num_mcts iterations, action = first arg-max of the root
 *                                        children's q
 * Randomness is Python's `random` module: MT19937 seeded by init_by_array, getrandbits(k) =
 * genrand_uint32() >> (32 - k), _randbelow(n) = rejection on k = n.bit_length() bits.
 * ================================================================================================= */
void oo_pyrandom_seed(oo_rng *r, uint32_t seed)
{
    /* random.seed(int < 2**32) -> init_by_array([seed]) (CPython _randommodule.c) */
    uint32_t *mt = r->mt;
    mt[0] = 19650218U;
    for (int i = 1; i < 624; i++) mt[i] = 1812433253U * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    int i = 1, j = 0;
    for (int k = 624; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525U)) + seed + (uint32_t)j;
        i++; j++;
        if (i >= 624) { mt[0] = mt[623]; i = 1; }
        if (j >= 1) j = 0;
    }
    for (int k = 623; k; k--) {
        mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941U)) - (uint32_t)i;
        i++;
        if (i >= 624) { mt[0] = mt[623]; i = 1; }
    }
    mt[0] = 0x80000000U;
    r->pos = 624;
    r->has_gauss = 0;
    r->gauss = 0.0;
}

int oo_pyrandom_below(oo_rng *r, int n)
{
    /* Random._randbelow_with_getrandbits (random.py): k = n.bit_length() */
    int k = 0;
    while ((n >> k) != 0) k++;
    uint32_t v;
    do { v = oo_rng_next32(r) >> (32 - k); } while ((int)v >= n);
    return (int)v;
}

typedef struct tnode {
    int parent, action, player, first_kid, nkids, n;
    double w, q;
} tnode;

/* utils.check_win of 1_tictactoe_MCTS (rows, columns, diagonals, anti-diagonals; draw last) */
static int ttt_check_win(const int8_t *b, int n, int k)
{
    int marks = 0;
    for (int i = 0; i < n * n; i++) marks += (b[i] != 0);
    for (int row = 0; row < n; row++)
        for (int col = 0; col + k <= n; col++) {
            int s = 0;
            for (int i = 0; i < k; i++) s += b[row * n + col + i];
            if (s == k) return 1;
            if (s == -k) return 2;
        }
    for (int row = 0; row + k <= n; row++)
        for (int col = 0; col < n; col++) {
            int s = 0;
            for (int i = 0; i < k; i++) s += b[(row + i) * n + col];
            if (s == k) return 1;
            if (s == -k) return 2;
        }
    for (int row = 0; row + k <= n; row++)
        for (int col = 0; col + k <= n; col++) {
            int s = 0;
            for (int i = 0; i < k; i++) s += b[(row + i) * n + col + i];
            if (s == k) return 1;
            if (s == -k) return 2;
        }
    for (int row = k - 1; row < n; row++)
        for (int col = 0; col + k <= n; col++) {
            int s = 0;
            for (int i = 0; i < k; i++) s += b[(row - i) * n + col + i];
            if (s == k) return 1;
            if (s == -k) return 2;
        }
    if (marks == n * n) return 3;
    return 0;
}

int oo_ttt_search(int board, int win_mark, int num_mcts, const int8_t *game_board, int turn, oo_rng *rng,
                  double *q_out, double *n_out)
{
    const int A = board * board;
    tnode *t = NULL;
    int used = 0, cap = 0;
#define TT_NEW(par, act, pl) do { if (used == cap) { cap = cap ? cap * 2 : 1024; t = (tnode *)realloc(t, sizeof(tnode) * (size_t)cap); } \
        t[used].parent = (par); t[used].action = (act); t[used].player = (pl); t[used].first_kid = -1; t[used].nkids = 0; \
        t[used].n = 0; t[used].w = 0; t[used].q = 0; used++; } while (0)
    TT_NEW(-1, -1, turn);
    int8_t *b = (int8_t *)malloc((size_t)A);
    int *acts = (int *)malloc(sizeof(int) * (size_t)A);
    int *path = (int *)malloc(sizeof(int) * (size_t)(A + 2));

    for (int it = 0; it < num_mcts; it++) {
        /* selection */
        int node = 0;
        while (t[node].nkids != 0) {
            double max_value = -100;
            const int leaf = node;
            for (int i = 0; i < t[leaf].nkids; i++) {
                const tnode *c = &t[t[leaf].first_kid + i];
                const double total_n = (double)t[c->parent].n;
                double q, u;
                if (c->n == 0) {
                    q = c->w / 0.0001;
                    double x = 2 * log(total_n);
                    x = x / 0.0001;
                    u = 5 * sqrt(x);
                } else {
                    q = c->w / c->n;
                    double x = 2 * log(total_n);
                    x = x / c->n;
                    u = 5 * sqrt(x);
                }
                if (q + u > max_value) { max_value = q + u; node = t[leaf].first_kid + i; }
            }
            if (node == leaf) break;   /* (the reference would spin forever; unreachable for q >= -1) */
        }
        /* state of the leaf: replay the path from the root */
        memcpy(b, game_board, (size_t)A);
        int plen = 0;
        for (int k = node; t[k].parent >= 0; k = t[k].parent) path[plen++] = k;
        for (int i = plen - 1; i >= 0; i--) b[t[path[i]].action] = (int8_t)(t[t[path[i]].parent].player == 0 ? 1 : -1);
        /* expansion */
        int child = node;
        const int is_terminal = ttt_check_win(b, board, win_mark);
        if (is_terminal == 0 && (node == 0 || t[node].n > 10)) {
            int na = 0;
            for (int c = 0; c < A; c++) if (b[c] == 0) acts[na++] = c;
            const int first = used;
            for (int i = 0; i < na; i++) TT_NEW(node, acts[i], t[node].player == 0 ? 1 : 0);
            t[node].first_kid = first;
            t[node].nkids = na;
            child = first + oo_pyrandom_below(rng, na);   /* random.sample(childs, 1)[0] */
            b[t[child].action] = (int8_t)(t[node].player == 0 ? 1 : -1);
        }
        /* simulation */
        int player = t[child].player, win;
        for (;;) {
            win = ttt_check_win(b, board, win_mark);
            if (win != 0) break;
            int na = 0;
            for (int c = 0; c < A; c++) if (b[c] == 0) acts[na++] = c;
            const int a = acts[oo_pyrandom_below(rng, na)];   /* random.choice(actions) */
            if (player == 0) { player = 1; b[a] = 1; } else { player = 0; b[a] = -1; }
        }
        /* backup */
        double value;
        if (win == 3) value = 0.8;
        else if (win - 1 == t[0].player) value = 1;
        else value = -1;
        for (int k = child;; k = t[k].parent) {
            t[k].n += 1;
            t[k].w += value;
            t[k].q = t[k].w / t[k].n;
            if (t[k].parent == 0) { t[0].n += 1; break; }
            if (t[k].parent < 0) break;   /* child is the root itself: the reference raises here */
        }
    }
    for (int a = 0; a < A; a++) { q_out[a] = -INFINITY; n_out[a] = 0; }
    int best = -1;
    double bq = 0;
    for (int i = 0; i < t[0].nkids; i++) {
        const tnode *c = &t[t[0].first_kid + i];
        q_out[c->action] = c->q;
        n_out[c->action] = c->n;
        if (best < 0 || c->q > bq) { best = c->action; bq = c->q; }   /* max(q_list, key=q_list.get): first maximum */
    }
    free(t); free(b); free(acts); free(path);
    return best;
}
