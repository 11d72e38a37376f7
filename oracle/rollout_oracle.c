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
