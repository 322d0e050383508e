/*
 * oz_oracle.c -- CPU ORACLE. TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the self-play hot path of Galtvam/OthelloZero
 * (rules, PUCT search with transposition tables, episode and arena drivers).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (othellozero_amd/) never does.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks every function
 * here against fixtures under tests/golden/ that were produced by importing
 * the reference's own Python modules (tests/golden/gen_golden.py).
 *
 * Deliberately array-based (an (n,n,2) byte board walked ray by ray, like the
 * reference) so that it is an independent check of the bit-parallel HIP code.
 * Compile with -ffp-contract=off: the PUCT float semantics below decide
 * discrete outcomes (SURVEY.md section 8, R-FP).
 *
 * Board I/O convention (shared with the HIP library and the fixtures):
 *   two uint64 bitboards, bit index = row*8 + col for every board size n
 *   (the n x n board sits in the top-left corner of an 8x8 grid);
 *   ch0 = reference channel 0 (BLACK, or "mover" for canonical states),
 *   ch1 = reference channel 1 (WHITE / "opponent").
 *   NN action index = row*n + col.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ */
/* shared integer mixers (stub net + RNG streams); mirrored bit for bit in
 * othellozero_amd/csrc/oz_common.h and tests/golden/gen_golden.py        */
/* ------------------------------------------------------------------ */
static inline uint64_t sm64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

ORC_API uint64_t orc_rng(uint64_t seed, uint64_t game, uint64_t move, uint64_t stream) {
    uint64_t a = sm64(seed + 0x632BE59BD9B4E019ULL * game);
    return sm64(a ^ (move * 0x9E3779B97F4A7C15ULL) ^ (stream * 0xD1B54A32D192ED03ULL));
}
enum { RNG_COIN = 0, RNG_EXPLORE = 1, RNG_TIE = 2 };
static inline double rng_unit(uint64_t u) { return (double)(u >> 11) * (1.0 / 9007199254740992.0); }

static inline uint64_t stub_h(uint64_t own, uint64_t opp, uint64_t salt, uint64_t i) {
    return sm64(sm64(own ^ salt) ^ rotl64(opp, 29) ^ ((i + 1) * 0xD6E8FEB86659FD93ULL));
}

/* Deterministic stand-in for NNetWrapper.predict (Net/NNet.py:70-87): a pure
 * integer function of the board, so every implementation reproduces it bit
 * for bit.  pi[a] = w_a / sum(w) as ONE float32 division, v is an exact
 * float32 in [-1, 1).  keep_mask != 0 zeroes most squares so that the
 * "all valid moves were masked" branch (MCTS/__init__.py:52-55) is reached. */
ORC_API void orc_stub_predict(uint64_t own, uint64_t opp, int n, uint64_t salt, uint64_t keep_mask,
                              float* pi /* n*n */, float* v) {
    uint32_t w[64];
    uint32_t sum = 0;
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) {
            uint64_t u = stub_h(own, opp, salt, (uint64_t)(r * 8 + c));
            uint32_t wi = (uint32_t)((u >> 40) & 0xFFFF);
            if (((u >> 8) & keep_mask) != 0) wi = 0;
            w[r * n + c] = wi;
            sum += wi;
        }
    if (sum == 0) { w[0] = 1; sum = 1; }
    for (int a = 0; a < n * n; ++a) pi[a] = (float)w[a] / (float)sum;
    uint64_t u = stub_h(own, opp, salt, 64);
    int32_t m = (int32_t)(u >> 40);
    *v = (float)(m - 8388608) / 8388608.0f;
}

/* ------------------------------------------------------------------ */
/* Rules -- Othello/__init__.py                                        */
/* ------------------------------------------------------------------ */
typedef struct { uint8_t ch[2][8][8]; } obrd;

static void unpack(uint64_t c0, uint64_t c1, obrd* b) {
    for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 8; ++c) {
            b->ch[0][r][c] = (uint8_t)((c0 >> (r * 8 + c)) & 1);
            b->ch[1][r][c] = (uint8_t)((c1 >> (r * 8 + c)) & 1);
        }
}
static void pack(const obrd* b, uint64_t* c0, uint64_t* c1) {
    uint64_t a = 0, d = 0;
    for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 8; ++c) {
            a |= (uint64_t)b->ch[0][r][c] << (r * 8 + c);
            d |= (uint64_t)b->ch[1][r][c] << (r * 8 + c);
        }
    *c0 = a; *c1 = d;
}

/* ALL_DIRECTIONS, Othello/__init__.py:27 */
static const int DIRS[8][2] = {{1, 1}, {1, 0}, {1, -1}, {0, -1}, {-1, -1}, {-1, 0}, {-1, 1}, {0, 1}};

static inline int sq_free(const obrd* b, int r, int c) { return !(b->ch[0][r][c] | b->ch[1][r][c]); }

/* get_action_flip_squares, Othello/__init__.py:216-235.  Marks the flipped
 * squares and returns how many were emitted (duplicates counted, as the
 * generator yields them).  The "flip-through" behaviour is reproduced: an own
 * disc emits the run collected so far but does NOT end the walk. */
static int flip_marks(const obrd* b, int n, int pch, int r, int c, uint8_t marks[8][8]) {
    int emitted = 0;
    if (marks) memset(marks, 0, 64);
    if (!sq_free(b, r, c)) return 0;
    const int och = 1 - pch;
    for (int d = 0; d < 8; ++d) {
        int dr = DIRS[d][0], dc = DIRS[d][1];
        int rr = r + dr, cc = c + dc;
        if (rr < 0 || rr >= n || cc < 0 || cc >= n) continue;
        if (!b->ch[och][rr][cc]) continue;
        int run_r[8], run_c[8], run = 0;
        run_r[run] = rr; run_c[run] = cc; ++run;
        for (rr += dr, cc += dc; rr >= 0 && rr < n && cc >= 0 && cc < n; rr += dr, cc += dc) {
            if (sq_free(b, rr, cc)) break;
            if (b->ch[pch][rr][cc]) {
                for (int i = 0; i < run; ++i) {
                    if (marks) marks[run_r[i]][run_c[i]] = 1;
                    ++emitted;
                }
            } else {
                run_r[run] = rr; run_c[run] = cc; ++run;
            }
        }
    }
    return emitted;
}

/* get_player_valid_actions, Othello/__init__.py:208-214: free squares in
 * np.argwhere (row-major) order whose flip generator yields at least once. */
static int valid_actions(const obrd* b, int n, int pch, uint8_t* out_sq /* r*8+c */) {
    int k = 0;
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c)
            if (sq_free(b, r, c) && flip_marks(b, n, pch, r, c, NULL) > 0) {
                if (out_sq) out_sq[k] = (uint8_t)(r * 8 + c);
                ++k;
            }
    return k;
}
static int has_actions(const obrd* b, int n, int pch) { return valid_actions(b, n, pch, NULL) > 0; }

/* flip_board_squares, Othello/__init__.py:237-247 (no legality check). */
static void flip_board(obrd* b, int n, int pch, int r, int c) {
    uint8_t marks[8][8];
    flip_marks(b, n, pch, r, c, marks);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            if (marks[i][j]) { b->ch[pch][i][j] = 1; b->ch[1 - pch][i][j] = 0; }
    b->ch[pch][r][c] = 1;
    b->ch[1 - pch][r][c] = 0;
}

/* has_board_finished, Othello/__init__.py:249-252 */
static int board_finished(const obrd* b, int n) { return !has_actions(b, n, 0) && !has_actions(b, n, 1); }

static void board_points(const obrd* b, int n, int* p0, int* p1) {
    int a = 0, d = 0;
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) { a += b->ch[0][r][c]; d += b->ch[1][r][c]; }
    *p0 = a; *p1 = d;
}
/* get_board_winning_player, Othello/__init__.py:254-260: max() over a dict
 * ordered BLACK, WHITE keeps the first maximum => a draw counts for BLACK. */
static int board_winner(const obrd* b, int n, int* points) {
    int p0, p1;
    board_points(b, n, &p0, &p1);
    if (p0 >= p1) { if (points) *points = p0; return +1; }
    if (points) *points = p1;
    return -1;
}

/* initial_board, Othello/__init__.py:177-184 */
ORC_API void orc_initial_board(int n, uint64_t* black, uint64_t* white) {
    obrd b; memset(&b, 0, sizeof b);
    int h = n / 2;
    b.ch[1][h - 1][h - 1] = 1; b.ch[0][h - 1][h] = 1;
    b.ch[0][h][h - 1] = 1;     b.ch[1][h][h] = 1;
    pack(&b, black, white);
}

/* --- flat rule entry points used by the tests --- */
ORC_API uint64_t orc_legal_mask(uint64_t c0, uint64_t c1, int n, int pch) {
    obrd b; unpack(c0, c1, &b);
    uint8_t sq[64]; int k = valid_actions(&b, n, pch, sq);
    uint64_t m = 0;
    for (int i = 0; i < k; ++i) m |= 1ULL << sq[i];
    return m;
}
ORC_API uint64_t orc_flip_mask(uint64_t c0, uint64_t c1, int n, int pch, int sq) {
    obrd b; unpack(c0, c1, &b);
    uint8_t marks[8][8];
    flip_marks(&b, n, pch, sq >> 3, sq & 7, marks);
    uint64_t m = 0;
    for (int r = 0; r < n; ++r) for (int c = 0; c < n; ++c) if (marks[r][c]) m |= 1ULL << (r * 8 + c);
    return m;
}
ORC_API void orc_apply_move(uint64_t* c0, uint64_t* c1, int n, int pch, int sq) {
    obrd b; unpack(*c0, *c1, &b);
    flip_board(&b, n, pch, sq >> 3, sq & 7);
    pack(&b, c0, c1);
}
ORC_API int orc_finished(uint64_t c0, uint64_t c1, int n) {
    obrd b; unpack(c0, c1, &b);
    return board_finished(&b, n);
}
ORC_API int orc_winner(uint64_t c0, uint64_t c1, int n, int* points) {
    obrd b; unpack(c0, c1, &b);
    return board_winner(&b, n, points);
}

/* OthelloGame instance, Othello/__init__.py:29-175 */
typedef struct { obrd b; int n; int player; /* +1 BLACK, -1 WHITE */ int finished; int round; } ogame;
static inline int pch_of(int player) { return player == 1 ? 0 : 1; }

static void game_init(ogame* g, int n) {
    memset(g, 0, sizeof *g);
    uint64_t bl, wh; orc_initial_board(n, &bl, &wh);
    unpack(bl, wh, &g->b);
    g->n = n; g->player = 1; g->finished = 0; g->round = 1;
}
/* play, Othello/__init__.py:136-159 */
static void game_play(ogame* g, int r, int c) {
    flip_board(&g->b, g->n, pch_of(g->player), r, c);
    g->round += 1;
    g->player = -g->player;
    if (!has_actions(&g->b, g->n, pch_of(g->player))) {
        if (!has_actions(&g->b, g->n, pch_of(-g->player))) g->finished = 1;
        else g->player = -g->player;
    }
}
ORC_API void orc_game_play(uint64_t* black, uint64_t* white, int n, int* player, int* finished, int sq) {
    ogame g; memset(&g, 0, sizeof g);
    unpack(*black, *white, &g.b); g.n = n; g.player = *player; g.finished = 0; g.round = 1;
    game_play(&g, sq >> 3, sq & 7);
    pack(&g.b, black, white); *player = g.player; *finished = g.finished;
}

/* ------------------------------------------------------------------ */
/* numpy pairwise sum of a contiguous float64 vector of length n <= 128
 * (what np.sum does for the (n,n) arrays at MCTS/__init__.py:49-51 and
 * othelo_mcts.py:67).  Pinned by tests/golden/pairwise_sum.npz.          */
/* ------------------------------------------------------------------ */
ORC_API double orc_pairwise_sum(const double* a, int n) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += a[i];
        return res;
    }
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

/* ------------------------------------------------------------------ */
/* Search -- MCTS/__init__.py + othelo_mcts.py                         */
/* ------------------------------------------------------------------ */
typedef void (*orc_eval_fn)(void* ctx, uint64_t own, uint64_t opp, int n, float* pi, float* v);

enum { VT_INT = 0, VT_F32 = 1, VT_F64 = 2 };          /* dynamic type of a backed-up value */
enum { QTAG_F64 = 0, QTAG_F32 = 1 };                  /* dynamic type of a stored Q        */
enum { QMODE_NEP50 = 0, QMODE_F64 = 1 };              /* NumPy>=2 vs NumPy 1.18 promotion  */
typedef struct { int t; double v; } oval;

typedef struct {
    uint64_t k0, k1;
    int Ns;
    int edges_init;           /* _Nsa[hash] non-empty (MCTS/__init__.py:61-63) */
    int nact; uint8_t act[64];/* cached _state_actions, ascending square */
    int N[64]; double Q[64]; uint8_t qtag[64]; double P[64];   /* indexed by square r*8+c */
} onode;

typedef struct {
    int n; double c; int qmode;
    orc_eval_fn eval; void* ctx;
    /* builtin stub evaluator */
    int use_stub; uint64_t stub_salt, stub_keep;
    onode* nodes; int nnodes, cap;
    int* ht; int htcap;
    long n_visits, n_expand, n_terminal, n_fallback;
    int max_depth;
} omcts;

static void stub_eval_cb(void* ctx, uint64_t own, uint64_t opp, int n, float* pi, float* v) {
    omcts* m = (omcts*)ctx;
    orc_stub_predict(own, opp, n, m->stub_salt, m->stub_keep, pi, v);
}

ORC_API omcts* orc_mcts_new(int n, double c, int qmode, orc_eval_fn eval, void* ctx) {
    omcts* m = (omcts*)calloc(1, sizeof *m);
    m->n = n; m->c = c; m->qmode = qmode; m->eval = eval; m->ctx = ctx;
    m->cap = 1024; m->nodes = (onode*)malloc(sizeof(onode) * (size_t)m->cap);
    m->htcap = 4096; m->ht = (int*)malloc(sizeof(int) * (size_t)m->htcap);
    for (int i = 0; i < m->htcap; ++i) m->ht[i] = -1;
    return m;
}
ORC_API omcts* orc_mcts_new_stub(int n, double c, int qmode, uint64_t salt, uint64_t keep_mask) {
    omcts* m = orc_mcts_new(n, c, qmode, stub_eval_cb, NULL);
    m->ctx = m; m->use_stub = 1; m->stub_salt = salt; m->stub_keep = keep_mask;
    return m;
}
ORC_API void orc_mcts_free(omcts* m) { if (m) { free(m->nodes); free(m->ht); free(m); } }

static inline uint64_t key_hash(uint64_t k0, uint64_t k1) { return sm64(k0 * 0x2545F4914F6CDD1DULL ^ sm64(k1)); }

static int node_find(const omcts* m, uint64_t k0, uint64_t k1) {
    uint64_t h = key_hash(k0, k1);
    for (int i = (int)(h & (uint64_t)(m->htcap - 1));; i = (i + 1) & (m->htcap - 1)) {
        int idx = m->ht[i];
        if (idx < 0) return -1;
        if (m->nodes[idx].k0 == k0 && m->nodes[idx].k1 == k1) return idx;
    }
}
static void ht_insert(omcts* m, int idx) {
    uint64_t h = key_hash(m->nodes[idx].k0, m->nodes[idx].k1);
    int i = (int)(h & (uint64_t)(m->htcap - 1));
    while (m->ht[i] >= 0) i = (i + 1) & (m->htcap - 1);
    m->ht[i] = idx;
}
static int node_new(omcts* m, uint64_t k0, uint64_t k1) {
    if (m->nnodes == m->cap) { m->cap *= 2; m->nodes = (onode*)realloc(m->nodes, sizeof(onode) * (size_t)m->cap); }
    if (2 * (m->nnodes + 1) > m->htcap) {
        m->htcap *= 2; m->ht = (int*)realloc(m->ht, sizeof(int) * (size_t)m->htcap);
        for (int i = 0; i < m->htcap; ++i) m->ht[i] = -1;
        for (int i = 0; i < m->nnodes; ++i) ht_insert(m, i);
    }
    int idx = m->nnodes++;
    onode* nd = &m->nodes[idx];
    memset(nd, 0, sizeof *nd);
    nd->k0 = k0; nd->k1 = k1;
    ht_insert(m, idx);
    return idx;
}

/* Q <- (N*Q + value)/(N+1), MCTS/__init__.py:68, with the operand types the
 * interpreter would see (SURVEY.md R-FP). */
static void q_update(const omcts* m, onode* nd, int sq, oval val) {
    const int N = nd->N[sq];
    const double Q = nd->Q[sq];
    if (m->qmode == QMODE_F64 || (nd->qtag[sq] == QTAG_F64 && val.t != VT_F32)) {
        nd->Q[sq] = ((double)N * Q + val.v) / (double)(N + 1);
        nd->qtag[sq] = QTAG_F64;
    } else {
        float prod = (nd->qtag[sq] == QTAG_F64) ? (float)((double)N * Q) : (float)N * (float)Q;
        float s = prod + (float)val.v;
        float q = s / (float)(N + 1);
        nd->Q[sq] = (double)q;
        nd->qtag[sq] = QTAG_F32;
    }
}

/* MCTS.simulate, MCTS/__init__.py:30-71, with the OthelloMCTS hooks of
 * othelo_mcts.py:28-49,69-88 inlined. */
static oval simulate(omcts* m, uint64_t k0, uint64_t k1, int depth) {
    const int n = m->n;
    m->n_visits++;
    if (depth > m->max_depth) m->max_depth = depth;
    obrd b; unpack(k0, k1, &b);

    if (board_finished(&b, n)) {                       /* :39-40 + othelo_mcts.py:28-35 */
        m->n_terminal++;
        int reward = board_winner(&b, n, NULL);        /* OthelloPlayer.value of the "BLACK"=ch0 view */
        return (oval){VT_INT, (double)(-reward)};
    }

    int idx = node_find(m, k0, k1);
    if (idx < 0) {                                     /* first visit: expand, :44-57 */
        m->n_expand++;
        idx = node_new(m, k0, k1);
        onode* nd = &m->nodes[idx];
        nd->nact = valid_actions(&b, n, 0, nd->act);   /* _get_state_actions via _mask_valid_moves */
        float pi[64]; float v = 0.f;
        m->eval(m->ctx, k0, k1, n, pi, &v);
        double arr[64];                                /* (n,n) float64, action order */
        for (int a = 0; a < n * n; ++a) arr[a] = 0.0;
        for (int i = 0; i < nd->nact; ++i) {
            int sq = nd->act[i], a = (sq >> 3) * n + (sq & 7);
            arr[a] = (double)pi[a] * 1.0;              /* float32 * float64 mask */
        }
        double sum = orc_pairwise_sum(arr, n * n);
        if (sum > 0) {
            for (int i = 0; i < nd->nact; ++i) {
                int sq = nd->act[i], a = (sq >> 3) * n + (sq & 7);
                nd->P[sq] = arr[a] / sum;
            }
        } else {                                       /* "All valid moves were masked" :52-55 */
            m->n_fallback++;
            double cnt = (double)nd->nact;
            for (int i = 0; i < nd->nact; ++i) nd->P[nd->act[i]] = 1.0 / cnt;
        }
        if (m->qmode == QMODE_F64) return (oval){VT_F64, -(double)v};
        return (oval){VT_F32, (double)(-v)};
    }

    /* revisit: select, :58-71 */
    {
        onode* nd = &m->nodes[idx];
        if (!nd->edges_init) {
            nd->edges_init = 1;
            for (int i = 0; i < nd->nact; ++i) { nd->N[nd->act[i]] = 0; nd->Q[nd->act[i]] = 0.0; nd->qtag[nd->act[i]] = QTAG_F64; }
        }
    }
    int best = -1; double best_u = 0.0;
    {
        const onode* nd = &m->nodes[idx];
        for (int i = 0; i < nd->nact; ++i) {           /* max(state_actions, key=U): first maximum */
            int sq = nd->act[i];
            double bound = sqrt((double)nd->Ns) / (double)(1 + nd->N[sq]);     /* :169 */
            double u = nd->Q[sq] + (m->c * nd->P[sq]) * bound;                 /* :170 */
            if (best < 0 || u > best_u) { best = sq; best_u = u; }
        }
    }
    /* get_next_state, othelo_mcts.py:43-49 */
    obrd nb = b;
    flip_board(&nb, n, 0, best >> 3, best & 7);
    uint64_t n0, n1; pack(&nb, &n0, &n1);
    if (has_actions(&nb, n, 1)) { uint64_t t = n0; n0 = n1; n1 = t; }

    oval value = simulate(m, n0, n1, depth + 1);

    onode* nd = &m->nodes[idx];                        /* re-fetch: the array may have moved */
    q_update(m, nd, best, value);
    nd->N[best] += 1;
    nd->Ns += 1;
    return (oval){value.t, -value.v};
}

/* OthelloMCTS.simulate(state, player), othelo_mcts.py:22-26.
 * (black, white) absolute colours; player +1/-1. Returns the value; *vtype
 * receives its dynamic type. */
ORC_API double orc_mcts_simulate(omcts* m, uint64_t black, uint64_t white, int player, int* vtype) {
    oval r = (player == 1) ? simulate(m, black, white, 0) : simulate(m, white, black, 0);
    if (vtype) *vtype = r.t;
    return r.v;
}

ORC_API int orc_mcts_num_nodes(const omcts* m) { return m->nnodes; }
ORC_API void orc_mcts_stats(const omcts* m, long* out4, int* max_depth) {
    out4[0] = m->n_visits; out4[1] = m->n_expand; out4[2] = m->n_terminal; out4[3] = m->n_fallback;
    if (max_depth) *max_depth = m->max_depth;
}
/* node dump in expansion order (== insertion order of the reference's dicts) */
ORC_API void orc_mcts_dump_node(const omcts* m, int i, uint64_t* k0, uint64_t* k1, int* Ns, int* edges_init,
                                uint64_t* legal, int* N /*64*/, double* Q /*64*/, uint8_t* qtag /*64*/, double* P /*64*/) {
    const onode* nd = &m->nodes[i];
    *k0 = nd->k0; *k1 = nd->k1; *Ns = nd->Ns; *edges_init = nd->edges_init;
    uint64_t lm = 0;
    for (int j = 0; j < nd->nact; ++j) lm |= 1ULL << nd->act[j];
    *legal = lm;
    memcpy(N, nd->N, sizeof nd->N); memcpy(Q, nd->Q, sizeof nd->Q);
    memcpy(qtag, nd->qtag, sizeof nd->qtag); memcpy(P, nd->P, sizeof nd->P);
}
ORC_API int orc_mcts_find(const omcts* m, uint64_t k0, uint64_t k1) { return node_find(m, k0, k1); }

/* N(state, action) per legal square of a canonical state, MCTS/__init__.py:73-84,172-175.
 * returns 0 ok, 1 = state unknown (all zero), 2 = KeyError (node never selected from). */
ORC_API int orc_mcts_counts(const omcts* m, uint64_t k0, uint64_t k1, int* counts /*64 by square*/, uint64_t* legal) {
    memset(counts, 0, 64 * sizeof(int));
    obrd b; unpack(k0, k1, &b);
    uint8_t act[64]; int na = valid_actions(&b, m->n, 0, act);
    uint64_t lm = 0; for (int i = 0; i < na; ++i) lm |= 1ULL << act[i];
    if (legal) *legal = lm;
    int idx = node_find(m, k0, k1);
    if (idx < 0) return 1;
    const onode* nd = &m->nodes[idx];
    if (!nd->edges_init) return 2;
    for (int i = 0; i < nd->nact; ++i) counts[nd->act[i]] = nd->N[nd->act[i]];
    return 0;
}

/* get_policy_action_probabilities, othelo_mcts.py:51-67. out: (n,n) float64 in
 * action order.  tie_u: the 64-bit draw used in place of random.choice(bests). */
static int policy_from_counts(int n, const int* counts /*by square*/, uint64_t legal, double T, uint64_t tie_u, double* out) {
    double p[64];
    for (int a = 0; a < n * n; ++a) p[a] = 0.0;
    if (T == 0) {
        for (int sq = 0; sq < 64; ++sq) if (legal >> sq & 1) p[(sq >> 3) * n + (sq & 7)] = (double)counts[sq];
        double mx = p[0]; for (int a = 1; a < n * n; ++a) if (p[a] > mx) mx = p[a];
        int bests[64] = {0}, nb = 0;
        for (int a = 0; a < n * n; ++a) if (p[a] == mx) bests[nb++] = a;
        int pick = bests[tie_u % (uint64_t)nb];
        for (int a = 0; a < n * n; ++a) out[a] = 0.0;
        out[pick] = 1.0;
        return pick;
    }
    for (int sq = 0; sq < 64; ++sq)
        if (legal >> sq & 1) p[(sq >> 3) * n + (sq & 7)] = pow((double)counts[sq], 1.0 / T);
    double s = orc_pairwise_sum(p, n * n);
    if (s == 0) s = 1;                                   /* (np.sum(...) or 1) */
    int arg = 0;
    for (int a = 0; a < n * n; ++a) { out[a] = p[a] / s; if (out[a] > out[arg]) arg = a; }
    return arg;                                          /* first maximum, row-major */
}
/* the same from given counts (tests: the per-move pi of a golden episode, whose counts and boards are in the fixture) */
ORC_API int orc_policy_from_counts(int n, const int* counts, uint64_t legal, double T, uint64_t tie_u, double* out) {
    return policy_from_counts(n, counts, legal, T, tie_u, out);
}
ORC_API int orc_mcts_policy(const omcts* m, uint64_t k0, uint64_t k1, double T, uint64_t tie_u, double* out, int* argmax) {
    int counts[64]; uint64_t legal;
    int rc = orc_mcts_counts(m, k0, k1, counts, &legal);
    if (rc == 2) return 2;
    int a = policy_from_counts(m->n, counts, legal, T, tie_u, out);
    if (argmax) *argmax = a;
    return rc;
}

/* ------------------------------------------------------------------ */
/* Drivers -- training.py:26-72, agents.py:44-84                       */
/* ------------------------------------------------------------------ */
typedef struct {
    int n_moves;
    uint64_t black[64], white[64];   /* absolute board BEFORE each move (per-move snapshot) */
    int8_t player[64];               /* mover +1/-1 */
    uint8_t action[64];              /* square r*8+c */
    int8_t z[64];                    /* +1 if winner == mover else -1 */
    uint8_t greedy[64];              /* 1 = coin <= e_greedy */
    int32_t counts[64][64];          /* root visit counts by square after the sims */
    uint64_t final_black, final_white;
    int winner, points;
    long stats[4];
} orc_episode_out;

/* execute_episode, training.py:26-72.  RNG draws come from orc_rng streams
 * keyed (seed, game_id, ply, purpose) instead of random/np.random. */
/* orc_episode_sched: the same loop with num_simulations = sims_pre on the plies < pre_plies and `sims` afterwards (what a
 * slot of the build's engine plays after oz_selfplay_stagger); orc_episode = pre_plies 0, the reference's own loop. */
ORC_API int orc_episode_sched(omcts* m, int sims, int sims_pre, int pre_plies, double T, double e_greedy, uint64_t seed,
                              uint64_t game_id, int max_moves, orc_episode_out* out);
ORC_API int orc_episode(omcts* m, int sims, double T, double e_greedy, uint64_t seed, uint64_t game_id,
                        int max_moves, orc_episode_out* out) {
    return orc_episode_sched(m, sims, sims, 0, T, e_greedy, seed, game_id, max_moves, out);
}
ORC_API int orc_episode_sched(omcts* m, int sims, int sims_pre, int pre_plies, double T, double e_greedy, uint64_t seed,
                              uint64_t game_id, int max_moves, orc_episode_out* out) {
    const int n = m->n;
    ogame g; game_init(&g, n);
    memset(out, 0, sizeof *out);
    int ply = 0;
    while (!g.finished) {
        if (max_moves >= 0 && ply >= max_moves) break;
        uint64_t bl, wh; pack(&g.b, &bl, &wh);
        const int nsims = ply < pre_plies ? sims_pre : sims;
        for (int s = 0; s < nsims; ++s) orc_mcts_simulate(m, bl, wh, g.player, NULL);
        uint64_t k0 = g.player == 1 ? bl : wh, k1 = g.player == 1 ? wh : bl;
        int counts[64]; uint64_t legal;
        int rc = orc_mcts_counts(m, k0, k1, counts, &legal);
        if (rc == 2) return -2;                          /* KeyError in the reference */
        double pol[64];
        int arg = policy_from_counts(n, counts, legal, T, orc_rng(seed, game_id, (uint64_t)ply, RNG_TIE), pol);
        double coin = rng_unit(orc_rng(seed, game_id, (uint64_t)ply, RNG_COIN));
        int sq;
        if (coin <= e_greedy) {
            sq = (arg / n) * 8 + (arg % n);
            out->greedy[ply] = 1;
        } else {
            uint8_t acts[64]; int na = 0;
            for (int s = 0; s < 64; ++s) if (legal >> s & 1) acts[na++] = (uint8_t)s;
            sq = acts[orc_rng(seed, game_id, (uint64_t)ply, RNG_EXPLORE) % (uint64_t)na];
        }
        out->black[ply] = bl; out->white[ply] = wh; out->player[ply] = (int8_t)g.player; out->action[ply] = (uint8_t)sq;
        memcpy(out->counts[ply], counts, sizeof counts);
        game_play(&g, sq >> 3, sq & 7);
        ++ply;
    }
    out->n_moves = ply;
    pack(&g.b, &out->final_black, &out->final_white);
    out->winner = board_winner(&g.b, n, &out->points);
    for (int i = 0; i < ply; ++i) out->z[i] = (int8_t)(out->winner == out->player[i] ? 1 : -1);
    orc_mcts_stats(m, out->stats, NULL);
    return g.finished ? 0 : 1;
}

typedef struct {
    int n_moves;
    uint8_t action[128]; int8_t player[128];
    uint64_t final_black, final_white;
    int winner, points;
} orc_arena_out;

/* duel_between_agents with two NeuralNetworkOthelloAgent, agents.py:44-84.
 * ma plays BLACK, mb plays WHITE; temperature forced to 0 (agents.py:46).
 * Either may be NULL: that colour is played by RandomOthelloAgent (agents.py:20-24; main.py:163-233 evaluation games). */
/* max_plies < 0: to the end of the game; else stop after max_plies plies (returns 1 if the game is then unfinished: the
 * bounded form bench.py's config-5 leg replays -- winner / points are then those of the unfinished board) */
ORC_API int orc_arena_plies(omcts* ma, omcts* mb, int sims, uint64_t seed, uint64_t game_id, int max_plies, orc_arena_out* out) {
    const int n = ma ? ma->n : mb->n;            /* a NULL search = RandomOthelloAgent on that colour */
    ogame g; game_init(&g, n);
    memset(out, 0, sizeof *out);
    int ply = 0;
    while (!g.finished && (max_plies < 0 || ply < max_plies)) {
        omcts* m = g.player == 1 ? ma : mb;
        uint64_t bl, wh; pack(&g.b, &bl, &wh);
        if (!m) {   /* RandomOthelloAgent.play, agents.py:20-24: random.choice over the valid actions (ascending row-major) */
            uint8_t acts[64];
            int na = valid_actions(&g.b, n, g.player == 1 ? 0 : 1, acts);        /* ascending row-major, Othello:208-214 */
            int sq = acts[orc_rng(seed, game_id, (uint64_t)ply, RNG_TIE) % (uint64_t)na];
            out->action[ply] = (uint8_t)sq; out->player[ply] = (int8_t)g.player;
            game_play(&g, sq >> 3, sq & 7);
            ++ply;
            continue;
        }
        for (int s = 0; s < sims; ++s) orc_mcts_simulate(m, bl, wh, g.player, NULL);
        uint64_t k0 = g.player == 1 ? bl : wh, k1 = g.player == 1 ? wh : bl;
        int counts[64]; uint64_t legal;
        int rc = orc_mcts_counts(m, k0, k1, counts, &legal);
        if (rc == 2) return -2;
        double pol[64];
        policy_from_counts(n, counts, legal, 0.0, orc_rng(seed, game_id, (uint64_t)ply, RNG_TIE), pol);
        /* max(valid_actions, key=pi): first valid action with the largest pi, agents.py:66-68 */
        int best = -1; double bp = 0;
        for (int s = 0; s < 64; ++s) if (legal >> s & 1) {
            double p = pol[(s >> 3) * n + (s & 7)];
            if (best < 0 || p > bp) { best = s; bp = p; }
        }
        out->action[ply] = (uint8_t)best; out->player[ply] = (int8_t)g.player;
        game_play(&g, best >> 3, best & 7);
        ++ply;
    }
    out->n_moves = ply;
    pack(&g.b, &out->final_black, &out->final_white);
    out->winner = board_winner(&g.b, n, &out->points);
    return g.finished ? 0 : 1;
}
ORC_API int orc_arena(omcts* ma, omcts* mb, int sims, uint64_t seed, uint64_t game_id, orc_arena_out* out) {
    return orc_arena_plies(ma, mb, sims, seed, game_id, -1, out);
}

/* training_example_symmetries, training.py:13-23: source index of every output
 * cell for the 8 outputs in the reference's order (rot90 k=1..4, flip True then
 * False).  perm[t][r*n+c] = r'*n+c' such that out_t[r][c] = in[r'][c']. */
ORC_API void orc_symmetry_perms(int n, int32_t* perm /* 8*n*n */) {
    int t = 0;
    for (int k = 1; k <= 4; ++k)
        for (int flip = 1; flip >= 0; --flip, ++t)
            for (int r = 0; r < n; ++r)
                for (int c = 0; c < n; ++c) {
                    int rr = r, cc = flip ? (n - 1 - c) : c;      /* undo fliplr */
                    for (int q = 0; q < (k & 3); ++q) {           /* undo one CCW rot90: out[i][j] = in[j][n-1-i] */
                        int ti = cc, tj = n - 1 - rr; rr = ti; cc = tj;
                    }
                    perm[(t * n + r) * n + c] = rr * n + cc;
                }
}
