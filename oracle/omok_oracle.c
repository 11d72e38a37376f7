/*
 * omok_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; see omok_oracle.h).
 *
 * Sequential, literal restatement of the reference's per-game search. Nothing here is shared
 * with the HIP product; every function cites the reference lines (relative to
 * /root/reference/2_AlphaOmok/) or the third-party routine it follows.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; glibc libm supplies log/pow/sqrt, the
 * same libm numpy calls for its legacy distributions).
 */
#include "omok_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================================
 * numpy legacy RandomState  (numpy/random/_mt19937.pyx, src/mt19937/mt19937.c,
 * src/legacy/legacy-distributions.c, _bounded_integers / distributions.c)
 * Call sites in the reference: agents.py:97-98,163,194-195; utils.py:192,202; main.py:60.
 * ====================================================================================== */
#define MT_N 624
#define MT_M 397

void oo_rng_seed(oo_rng *r, uint32_t seed)
{
    /* mt19937_seed(): init_genrand */
    r->mt[0] = seed;
    for (int i = 1; i < MT_N; i++)
        r->mt[i] = 1812433253u * (r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) + (uint32_t)i;
    r->pos = MT_N;
    r->has_gauss = 0;
    r->gauss = 0.0;
}

static void mt_twist(oo_rng *r)
{
    uint32_t *mt = r->mt, y;
    int kk;
    for (kk = 0; kk < MT_N - MT_M; kk++) {
        y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
        mt[kk] = mt[kk + MT_M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    for (; kk < MT_N - 1; kk++) {
        y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
        mt[kk] = mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    y = (mt[MT_N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
    mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    r->pos = 0;
}

uint32_t oo_rng_next32(oo_rng *r)
{
    if (r->pos == MT_N) mt_twist(r);
    uint32_t y = r->mt[r->pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

double oo_rng_double(oo_rng *r)
{
    /* mt19937_next_double */
    int32_t a = (int32_t)(oo_rng_next32(r) >> 5), b = (int32_t)(oo_rng_next32(r) >> 6);
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

int64_t oo_rng_below(oo_rng *r, int64_t k)
{
    /* RandomState.choice(k) -> randint(0, k) -> _rand_int64 -> masked rejection on 32-bit words.
     * rng == 0 consumes nothing. */
    uint64_t rng = (uint64_t)(k - 1);
    if (rng == 0) return 0;
    uint64_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4;
    mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
    uint32_t v;
    do { v = oo_rng_next32(r) & (uint32_t)mask; } while (v > rng);
    return (int64_t)v;
}

static double legacy_gauss(oo_rng *r)
{
    if (r->has_gauss) {
        double t = r->gauss;
        r->has_gauss = 0;
        r->gauss = 0.0;
        return t;
    }
    double f, x1, x2, r2;
    do {
        x1 = 2.0 * oo_rng_double(r) - 1.0;
        x2 = 2.0 * oo_rng_double(r) - 1.0;
        r2 = x1 * x1 + x2 * x2;
    } while (r2 >= 1.0 || r2 == 0.0);
    f = sqrt(-2.0 * log(r2) / r2);
    r->gauss = f * x1;
    r->has_gauss = 1;
    return f * x2;
}

static double legacy_standard_exponential(oo_rng *r)
{
    return -log(1.0 - oo_rng_double(r));
}

static double legacy_standard_gamma(oo_rng *r, double shape)
{
    double b, c, U, V, X, Y;
    if (shape == 1.0) return legacy_standard_exponential(r);
    if (shape == 0.0) return 0.0;
    if (shape < 1.0) {
        for (;;) {
            U = oo_rng_double(r);
            V = legacy_standard_exponential(r);
            if (U <= 1.0 - shape) {
                X = pow(U, 1. / shape);
                if (X <= V) return X;
            } else {
                Y = -log((1 - U) / shape);
                X = pow(1.0 - shape + shape * Y, 1. / shape);
                if (X <= (V + Y)) return X;
            }
        }
    }
    b = shape - 1. / 3.;
    c = 1. / sqrt(9 * b);
    for (;;) {
        do {
            X = legacy_gauss(r);
            V = 1.0 + c * X;
        } while (V <= 0.0);
        V = V * V * V;
        U = oo_rng_double(r);
        if (U < 1.0 - 0.0331 * (X * X) * (X * X)) return (b * V);
        if (log(U) < 0.5 * X * X + b * (1. - V + log(V))) return (b * V);
    }
}

void oo_rng_dirichlet(oo_rng *r, double alpha, int k, double *out)
{
    /* RandomState.dirichlet(alpha*ones(k)): gammas, sequential sum, multiply by 1/acc */
    double acc = 0.0;
    for (int j = 0; j < k; j++) {
        out[j] = legacy_standard_gamma(r, alpha);
        acc = acc + out[j];
    }
    if (k > 0) {
        double invacc = 1 / acc;
        for (int j = 0; j < k; j++) out[j] = out[j] * invacc;
    }
}

int oo_rng_choice_p(oo_rng *r, const double *p, int n)
{
    /* RandomState.choice(n, p=p): cdf = p.cumsum(); cdf /= cdf[-1];
     * idx = cdf.searchsorted(random_sample(), side='right')  (utils.py:192) */
    double *cdf = (double *)malloc(sizeof(double) * (size_t)n);
    double s = 0.0;
    for (int i = 0; i < n; i++) {
        s = (i == 0) ? p[0] : s + p[i];
        cdf[i] = s;
    }
    double last = cdf[n - 1];
    for (int i = 0; i < n; i++) cdf[i] = cdf[i] / last;
    double u = oo_rng_double(r);
    int lo = 0, hi = n; /* first index with cdf[idx] > u */
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
    }
    free(cdf);
    return lo;
}

/* ======================================================================================
 * numpy pairwise sum (numpy/core/src/umath/loops_utils.h.src: DOUBLE_pairwise_sum)
 * Call site: agents.py:189  prior_prob.sum()
 * ====================================================================================== */
double oo_pairwise_sum(const double *a, int n)
{
    if (n < 8) {
        /* numpy starts from -0.0 so that a sum of -0.0 stays -0.0 (numpy>=1.25); all our
         * inputs are >= +0.0, for which 0.0 and -0.0 starts give identical results except the
         * all-empty case, which is not reachable (n is the full vector length >= 9). */
        double res = -0.0;
        for (int i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8], res;
        int i;
        for (i = 0; i < 8; i++) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int n2 = n / 2;
        n2 -= n2 % 8;
        return oo_pairwise_sum(a, n2) + oo_pairwise_sum(a + n2, n - n2);
    }
}

/* ======================================================================================
 * CPython 3.10 set: iteration order of  set(range(A)) - set(id[1:])   (utils.py:22-27)
 * Objects/setobject.c: set_difference, set_add_entry, set_insert_clean, set_table_resize.
 * ====================================================================================== */
#define LINEAR_PROBES 9
#define PERTURB_SHIFT 5

typedef struct { int *tab; int mask; int fill; } pyset;

static int set_find_slot(const int *tab, int mask, int key)
{
    size_t perturb = (size_t)key;
    size_t i = (size_t)key & (size_t)mask;
    for (;;) {
        int probes = (i + LINEAR_PROBES <= (size_t)mask) ? LINEAR_PROBES : 0;
        size_t e = i;
        do {
            if (tab[e] < 0) return (int)e;
            e++;
        } while (probes--);
        perturb >>= PERTURB_SHIFT;
        i = (i * 5 + 1 + perturb) & (size_t)mask;
    }
}

static void set_resize(pyset *s, int minused)
{
    int newsize = 8;
    while (newsize <= minused) newsize <<= 1;
    int *nt = (int *)malloc(sizeof(int) * (size_t)newsize);
    for (int i = 0; i < newsize; i++) nt[i] = -1;
    for (int i = 0; i <= s->mask; i++)
        if (s->tab[i] >= 0) nt[set_find_slot(nt, newsize - 1, s->tab[i])] = s->tab[i];
    free(s->tab);
    s->tab = nt;
    s->mask = newsize - 1;
}

static void set_add(pyset *s, int key)
{
    s->tab[set_find_slot(s->tab, s->mask, key)] = key;
    s->fill++;
    if ((size_t)s->fill * 5 < (size_t)s->mask * 3) return;
    set_resize(s, s->fill > 50000 ? s->fill * 2 : s->fill * 4);
}

int oo_legal_actions(const int *moves, int nmoves, int board, int *out)
{
    int A = board * board;
    char *occ = (char *)calloc((size_t)A, 1);
    int nstones = 0;
    for (int i = 0; i < nmoves; i++)
        if (!occ[moves[i]]) { occ[moves[i]] = 1; nstones++; }
    int cnt = 0;
    if ((A >> 2) > nstones) {
        /* set_copy_and_difference: the copy has table size > A, ints hash to themselves */
        for (int a = 0; a < A; a++) if (!occ[a]) out[cnt++] = a;
    } else {
        pyset s;
        s.tab = (int *)malloc(sizeof(int) * 8);
        for (int i = 0; i < 8; i++) s.tab[i] = -1;
        s.mask = 7; s.fill = 0;
        for (int a = 0; a < A; a++) if (!occ[a]) set_add(&s, a); /* `so` iterates ascending */
        for (int i = 0; i <= s.mask; i++) if (s.tab[i] >= 0) out[cnt++] = s.tab[i];
        free(s.tab);
    }
    free(occ);
    return cnt;
}

/* ======================================================================================
 * utils.py game helpers
 * ====================================================================================== */
void oo_get_board(const int *moves, int nmoves, int board, int8_t *out)
{
    /* utils.py:171-179 */
    memset(out, 0, (size_t)(board * board));
    for (int i = 0; i < nmoves; i++) out[moves[i]] = (i % 2 == 0) ? 1 : -1;
}

int oo_check_win(const int8_t *b, int n, int k)
{
    /* utils.py:30-59: window scan, black tests before white inside a window, draw last */
    int num_mark = 0;
    for (int i = 0; i < n * n; i++) num_mark += (b[i] != 0);
    for (int row = 0; row < n - k + 1; row++) {
        for (int col = 0; col < n - k + 1; col++) {
            int hb = 0, hw = 0, vb = 0, vw = 0;
            for (int i = 0; i < k; i++) {
                int sh = 0, sv = 0;
                for (int j = 0; j < k; j++) {
                    sh += b[(row + i) * n + col + j];
                    sv += b[(row + j) * n + col + i];
                }
                hb |= (sh == k); hw |= (sh == -k);
                vb |= (sv == k); vw |= (sv == -k);
            }
            int d1 = 0, d2 = 0;
            for (int i = 0; i < k; i++) {
                d1 += b[(row + i) * n + col + i];
                d2 += b[(row + k - 1 - i) * n + col + i];
            }
            if (hb || vb) return 1;
            if (d1 == k || d2 == k) return 1;
            if (hw || vw) return 2;
            if (d1 == -k || d2 == -k) return 2;
        }
    }
    if (num_mark == n * n) return 3;
    return 0;
}

void oo_get_state_pt(const int *moves, int nmoves, int board, int C, float *out)
{
    /* utils.py:139-168, literal: deque(maxlen=C) of plane snapshots, then the colour plane */
    int A = board * board;
    float *ring = (float *)calloc((size_t)C * (size_t)A, sizeof(float)); /* C zero planes */
    int head = 0; /* index of the oldest plane */
    float *sb = (float *)calloc((size_t)A, sizeof(float));
    float *sw = (float *)calloc((size_t)A, sizeof(float));
    int color_idx = 1;
#define PUSH(src) do { memcpy(ring + (size_t)head * A, (src), sizeof(float) * (size_t)A); \
                       head = (head + 1) % C; } while (0)
    /* i == 0 : the leading 0 of the id */
    PUSH(sb);
    PUSH(sw);
    for (int i = 1; i <= nmoves; i++) {
        int a = moves[i - 1];
        if (i % 2 == 1) { sb[a] = 1.f; PUSH(sb); color_idx = 0; }
        else            { sw[a] = 1.f; PUSH(sw); color_idx = 1; }
    }
    float *col = (float *)malloc(sizeof(float) * (size_t)A);
    for (int i = 0; i < A; i++) col[i] = (float)color_idx;
    PUSH(col);
#undef PUSH
    for (int c = 0; c < C; c++)
        memcpy(out + (size_t)c * A, ring + (size_t)((head + c) % C) * A, sizeof(float) * (size_t)A);
    free(ring); free(sb); free(sw); free(col);
}

/* ======================================================================================
 * exact-arithmetic stub evaluator (test-only stand-in for Agent.model)
 * ====================================================================================== */
static uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void oo_stub_eval_planes(const float *planes, int board, int C, int mode, float *policy,
                         float *value)
{
    int A = board * board;
    uint64_t colour = planes[(size_t)(C - 1) * A] != 0.f ? 1u : 0u;
    uint64_t h = 0x9E3779B97F4A7C15ull * (1u + colour);
    for (int pl = 0; pl < C - 1; pl++)
        for (int cell = 0; cell < A; cell++)
            if (planes[(size_t)pl * A + cell] != 0.f)
                h += (uint64_t)(cell + 1 + 1000 * pl) * 0xBF58476D1CE4E5B9ull;
    for (int a = 0; a < A; a++) {
        uint64_t z = splitmix64(h + (uint64_t)(a + 1) * 0xD6E8FEB86659FD93ull);
        uint32_t k = (uint32_t)((z >> 40) & 0xFFFFu);
        if (mode == 1 && ((z >> 56) & 0x1Fu) != 0) k >>= 8;
        policy[a] = (float)(k + 1) / 65536.0f;
    }
    uint64_t z = splitmix64(h ^ 0xA5A5A5A5A5A5A5A5ull);
    int m = (int)((z >> 40) & 0xFFFu);
    *value = (mode == 2) ? (float)(m - 2048) / 16384.0f : (float)(m - 2048) / 2048.0f;
}

/* ======================================================================================
 * ZeroAgent (agents.py:39-260)
 * self.tree is a dict keyed by the id tuple; here: a trie over action indices whose nodes
 * carry an in_tree flag (= key present in the dict).
 * Scalar types follow numpy 2.2.6 / NEP 50: n python float; w,q become np.float32
 * once a float32 value has been added, python float (binary64) until then; p float64.
 * ====================================================================================== */
typedef struct onode {
    struct onode **kids; /* [A], lazily allocated */
    int *child;          /* tree[id]['child'] : action indices in stored order */
    int nchild;
    int in_tree;
    double n;
    double w; int w_f32;
    double q; int q_f32;
    double p;
} onode;

struct oo_agent {
    int board, A, num_mcts, inplanes, win_mark, noise;
    double alpha, c_puct;
    onode *trie;          /* node for id (0,) */
    long tree_size;
    int root_moves[512]; int root_n; int has_root;
    int is_real_root;
    oo_rng rng;
    oo_eval_fn eval; void *eval_ctx;
    int stub_mode;
    long num_evals;
    long st_levels, st_ties, st_term;
    int sim_index;
};

static onode *node_new(void)
{
    return (onode *)calloc(1, sizeof(onode));
}

static void node_free(onode *nd, int A)
{
    if (!nd) return;
    if (nd->kids) {
        for (int a = 0; a < A; a++) node_free(nd->kids[a], A);
        free(nd->kids);
    }
    free(nd->child);
    free(nd);
}

static onode *trie_get(oo_agent *ag, const int *moves, int n, int create)
{
    onode *nd = ag->trie;
    if (!nd) {
        if (!create) return NULL;
        nd = ag->trie = node_new();
    }
    for (int i = 0; i < n; i++) {
        if (!nd->kids) {
            if (!create) return NULL;
            nd->kids = (onode **)calloc((size_t)ag->A, sizeof(onode *));
        }
        onode *k = nd->kids[moves[i]];
        if (!k) {
            if (!create) return NULL;
            k = nd->kids[moves[i]] = node_new();
        }
        nd = k;
    }
    return nd;
}

static void stub_adapter(void *ctx, const int *moves, int nmoves, const float *planes, int sim,
                         float *policy, float *value)
{
    (void)moves; (void)nmoves; (void)sim;
    oo_agent *ag = (oo_agent *)ctx;
    oo_stub_eval_planes(planes, ag->board, ag->inplanes, ag->stub_mode, policy, value);
}

oo_agent *oo_agent_create(int board, int num_mcts, int inplanes, int noise)
{
    /* agents.py:40-53 */
    oo_agent *ag = (oo_agent *)calloc(1, sizeof(oo_agent));
    ag->board = board; ag->A = board * board;
    ag->num_mcts = num_mcts; ag->inplanes = inplanes;
    ag->win_mark = (board == 3) ? 3 : 5;
    ag->alpha = 10.0 / (double)(board * board);
    ag->c_puct = 5.0;
    ag->noise = noise;
    ag->is_real_root = 1;
    oo_rng_seed(&ag->rng, 0);
    return ag;
}

void oo_agent_destroy(oo_agent *ag)
{
    if (!ag) return;
    node_free(ag->trie, ag->A);
    free(ag);
}

void oo_agent_set_eval(oo_agent *ag, oo_eval_fn fn, void *ctx) { ag->eval = fn; ag->eval_ctx = ctx; }
/* ZeroAgent.win_mark is a plain attribute (agents.py:41-44 sets 3 or 5 from the board size); utils.check_win takes it as is:
   a mark above the board size leaves the full-board draw as the only end of a game. */
void oo_agent_set_win_mark(oo_agent *ag, int k) { ag->win_mark = k; }
void oo_agent_use_stub(oo_agent *ag, int mode) { ag->stub_mode = mode; ag->eval = stub_adapter; ag->eval_ctx = ag; }
oo_rng *oo_agent_rng(oo_agent *ag) { return &ag->rng; }
long oo_agent_tree_size(oo_agent *ag) { return ag->tree_size; }
long oo_agent_num_evals(oo_agent *ag) { return ag->num_evals; }
void oo_agent_last_stats(oo_agent *ag, long *levels, long *ties, long *term)
{
    if (levels) *levels = ag->st_levels;
    if (ties) *ties = ag->st_ties;
    if (term) *term = ag->st_term;
}

void oo_agent_reset(oo_agent *ag)
{
    /* agents.py:55-58 */
    node_free(ag->trie, ag->A);
    ag->trie = NULL;
    ag->tree_size = 0;
    ag->has_root = 0;
    ag->is_real_root = 1;
}

static void init_mcts(oo_agent *ag, const int *root_moves, int n)
{
    /* agents.py:82-103 */
    memcpy(ag->root_moves, root_moves, sizeof(int) * (size_t)n);
    ag->root_n = n; ag->has_root = 1;
    onode *root = trie_get(ag, root_moves, n, 1);
    if (!root->in_tree) {
        ag->is_real_root = 1;
        root->in_tree = 1;
        root->n = 0.; root->w = 0.; root->q = 0.; root->p = 0.;
        root->w_f32 = root->q_f32 = 0;
        root->nchild = 0;
        ag->tree_size++;
    } else {
        ag->is_real_root = 0;
        if (ag->noise) {
            int k = root->nchild;
            double *nz = (double *)malloc(sizeof(double) * (size_t)(k > 0 ? k : 1));
            oo_rng_dirichlet(&ag->rng, ag->alpha, k, nz);
            for (int i = 0; i < k; i++) {
                onode *c = root->kids[root->child[i]];
                double t1 = 0.75 * c->p;
                double t2 = 0.25 * nz[i];
                c->p = t1 + t2;
            }
            free(nz);
        }
    }
}

/* moves buffer `path` holds the id[1:] of the current node; returns win_index of the leaf */
static int selection(oo_agent *ag, int *path, int *plen, onode **leaf_out)
{
    /* agents.py:134-168 */
    int A = ag->A;
    int8_t *board = (int8_t *)malloc((size_t)A);
    double *qu = (double *)malloc(sizeof(double) * (size_t)A);
    onode *node = trie_get(ag, path, *plen, 0);
    int win_index;
    while (node->n > 0) {
        oo_get_board(path, *plen, ag->board, board);
        win_index = oo_check_win(board, ag->board, ag->win_mark);
        if (win_index != 0) {
            free(board); free(qu);
            *leaf_out = node;
            return win_index;
        }
        double total_n = 0; /* python: int 0 + python floats */
        for (int i = 0; i < node->nchild; i++) total_n += node->kids[node->child[i]]->n;
        double sq = sqrt(total_n);
        double max_value = 0;
        for (int i = 0; i < node->nchild; i++) {
            onode *c = node->kids[node->child[i]];
            /* u = self.c_puct * p * np.sqrt(total_n) / (n + 1)   (left to right, float64) */
            double t = ag->c_puct * c->p;
            t = t * sq;
            double u = t / (c->n + 1);
            double v = c->q + u;
            qu[i] = v;
            if (i == 0 || v > max_value) max_value = v;
        }
        int cnt = 0;
        for (int i = 0; i < node->nchild; i++) if (qu[i] == max_value) cnt++;
        int r = (int)oo_rng_below(&ag->rng, cnt);
        if (cnt > 1) ag->st_ties++;
        int pick = -1;
        for (int i = 0; i < node->nchild; i++)
            if (qu[i] == max_value) { if (r == 0) { pick = i; break; } r--; }
        int a = node->child[pick];
        path[(*plen)++] = a;
        node = node->kids[a];
        ag->st_levels++;
    }
    oo_get_board(path, *plen, ag->board, board);
    win_index = oo_check_win(board, ag->board, ag->win_mark);
    free(board); free(qu);
    *leaf_out = node;
    return win_index;
}

static int is_root(oo_agent *ag, const int *path, int plen)
{
    return plen == ag->root_n && memcmp(path, ag->root_moves, sizeof(int) * (size_t)plen) == 0;
}

/* returns 1 if reward path (terminal), value in *value otherwise */
static int expansion_evaluation(oo_agent *ag, onode *leaf, const int *path, int plen,
                                int win_index, float *value)
{
    /* agents.py:170-221. The net is evaluated even on terminal leaves (result unused). */
    int A = ag->A;
    float *planes = (float *)malloc(sizeof(float) * (size_t)ag->inplanes * (size_t)A);
    float *policy = (float *)malloc(sizeof(float) * (size_t)A);
    oo_get_state_pt(path, plen, ag->board, ag->inplanes, planes);
    float v = 0.f;
    ag->eval(ag->eval_ctx, path, plen, planes, ag->sim_index, policy, &v);
    ag->num_evals++;
    free(planes);
    if (win_index == 0) {
        int *actions = (int *)malloc(sizeof(int) * (size_t)A);
        int L = oo_legal_actions(path, plen, ag->board, actions);
        double *prior = (double *)calloc((size_t)A, sizeof(double));
        for (int i = 0; i < L; i++) prior[actions[i]] = (double)policy[actions[i]];
        double s = oo_pairwise_sum(prior, A);
        for (int a = 0; a < A; a++) prior[a] = prior[a] / s;
        double *nz = NULL;
        int at_root = is_root(ag, path, plen);
        if (ag->noise && at_root) {
            nz = (double *)malloc(sizeof(double) * (size_t)(L > 0 ? L : 1));
            oo_rng_dirichlet(&ag->rng, ag->alpha, L, nz);
        }
        if (!leaf->kids) leaf->kids = (onode **)calloc((size_t)A, sizeof(onode *));
        leaf->child = (int *)realloc(leaf->child, sizeof(int) * (size_t)(leaf->nchild + L + 1));
        for (int i = 0; i < L; i++) {
            int a = actions[i];
            double prior_p = prior[a];
            if (nz) {
                double t1 = 0.75 * prior_p;
                double t2 = 0.25 * nz[i];
                prior_p = t1 + t2;
            }
            onode *c = leaf->kids[a];
            if (!c) c = leaf->kids[a] = node_new();
            if (!c->in_tree) ag->tree_size++;
            c->in_tree = 1;
            c->n = 0.; c->w = 0.; c->q = 0.; c->w_f32 = c->q_f32 = 0;
            c->p = prior_p;
            /* the dict assignment replaces the record: 'child': [] */
            c->nchild = 0;
            leaf->child[leaf->nchild++] = a;
        }
        free(actions); free(prior); free(nz); free(policy);
        *value = v;
        return 0;
    }
    free(policy);
    ag->st_term++;
    return 1;
}

static void backup(oo_agent *ag, int *path, int plen, float value, int reward)
{
    /* agents.py:223-239: from the leaf up to and including the search root */
    int count = 0;
    int len = plen;
    while (len >= ag->root_n) {
        onode *nd = trie_get(ag, path, len, 0);
        nd->n += 1;
        double sgn = (count % 2 == 0) ? 1.0 : -1.0;
        if (!reward) {
            /* np.float32 (-value) * python int  -> float32 ; w (float or float32) + float32 */
            float add = (float)(-value) * (float)sgn;
            float wf = (float)nd->w; /* exact: python-float w holds small integers only */
            nd->w = (double)(float)(wf + add);
            nd->w_f32 = 1;
        } else {
            double add = 1.0 * sgn;
            if (nd->w_f32) nd->w = (double)(float)((float)nd->w + (float)add);
            else nd->w = nd->w + add;
        }
        count++;
        if (nd->w_f32) { nd->q = (double)(float)((float)nd->w / (float)nd->n); nd->q_f32 = 1; }
        else { nd->q = nd->w / nd->n; nd->q_f32 = 0; }
        len--;
    }
}

int oo_agent_get_pi(oo_agent *ag, const int *root_moves, int n, int tau, double *pi,
                    double *visit, double *policy)
{
    /* agents.py:60-80 + 105-132 */
    int A = ag->A;
    init_mcts(ag, root_moves, n);
    int num = ag->is_real_root ? ag->num_mcts + 1 : ag->num_mcts;
    ag->st_levels = ag->st_ties = ag->st_term = 0;
    int path[512];
    for (int i = 0; i < num; i++) {
        ag->sim_index = i;
        memcpy(path, ag->root_moves, sizeof(int) * (size_t)ag->root_n);
        int plen = ag->root_n;
        onode *leaf = NULL;
        int win_index = selection(ag, path, &plen, &leaf);
        float value = 0.f;
        int reward = expansion_evaluation(ag, leaf, path, plen, win_index, &value);
        backup(ag, path, plen, value, reward);
    }
    onode *root = trie_get(ag, ag->root_moves, ag->root_n, 0);
    for (int a = 0; a < A; a++) { visit[a] = 0.; policy[a] = 0.; }
    for (int i = 0; i < root->nchild; i++) {
        int a = root->child[i];
        visit[a] = root->kids[a]->n;
        policy[a] = root->kids[a]->p;
    }
    double s = oo_pairwise_sum(visit, A);
    for (int a = 0; a < A; a++) pi[a] = visit[a] / s;
    if (tau == 0) {
        /* utils.argmax_onehot (utils.py:198-205) */
        double mx = pi[0];
        for (int a = 1; a < A; a++) if (pi[a] > mx) mx = pi[a];
        int cnt = 0;
        for (int a = 0; a < A; a++) if (pi[a] == mx) cnt++;
        int r = (int)oo_rng_below(&ag->rng, cnt);
        int pick = -1;
        for (int a = 0; a < A; a++) if (pi[a] == mx) { if (r == 0) { pick = a; break; } r--; }
        for (int a = 0; a < A; a++) pi[a] = 0.;
        pi[pick] = 1.;
    }
    return 0;
}

int oo_agent_children(oo_agent *ag, const int *moves, int n, double *cn, double *cw, double *cq,
                      double *cp, int *order)
{
    onode *nd = trie_get(ag, moves, n, 0);
    if (!nd || !nd->in_tree) return -1;
    for (int a = 0; a < ag->A; a++) { cn[a] = cw[a] = cq[a] = cp[a] = 0.; }
    for (int i = 0; i < nd->nchild; i++) {
        int a = nd->child[i];
        onode *c = nd->kids[a];
        cn[a] = c->n; cw[a] = c->w; cq[a] = c->q; cp[a] = c->p;
        if (order) order[i] = a;
    }
    return nd->nchild;
}

int oo_self_play_game(oo_agent *ag, uint32_t seed, int tau_thres, int max_plies, int *moves,
                      double *pis, double *visits, int *win_index_out)
{
    /* main.py:136-227 + 248, one episode; np.random.seed(seed) before the episode */
    int A = ag->A;
    oo_rng_seed(&ag->rng, seed);
    int8_t *board = (int8_t *)calloc((size_t)A, 1);
    double *policy = (double *)malloc(sizeof(double) * (size_t)A);
    int win_index = 0, time_steps = 0, turn = 0;
    int root[512]; int rn = 0;
    while (win_index == 0) {
        int tau = (time_steps < tau_thres) ? 1 : 0;
        double *pi = pis + (size_t)time_steps * A;
        oo_agent_get_pi(ag, root, rn, tau, pi, visits + (size_t)time_steps * A, policy);
        int action_index = oo_rng_choice_p(&ag->rng, pi, A); /* utils.get_action */
        root[rn++] = action_index;
        moves[time_steps] = action_index;
        /* env.step (env_small.py:154-176,196): place stone, flip turn, check_win */
        board[action_index] = (turn == 0) ? 1 : -1;
        turn ^= 1;
        win_index = oo_check_win(board, ag->board, ag->win_mark);
        time_steps++;
        if (max_plies && time_steps >= max_plies) break;
    }
    oo_agent_reset(ag);
    free(board); free(policy);
    *win_index_out = win_index;
    return time_steps;
}
