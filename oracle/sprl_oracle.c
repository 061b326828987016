/* TEST INFRASTRUCTURE — CPU restatement ("oracle") of the reference self-play path.
 * See sprl_oracle.h for scope, parity status and who may load this.  Every function cites the
 * reference file:line it restates (paths relative to /root/reference/cpp/src).
 *
 * Array boards (int8: -1 empty, 0, 1) and pointer-linked trees are used on purpose: the device
 * engine uses bitboards and flat arenas, so the two implementations share no rule or tree code.
 */
#include "sprl_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../sprl_amd/csrc/sprl_math.h"

/* ------------------------------------------------------------------------------------------ */
/* geometry                                                                                    */
/* ------------------------------------------------------------------------------------------ */

typedef struct { int rows, cols, cells, A, nsym, hist; } geom_t;

static geom_t geom(int game) {
    geom_t g;
    if (game == ORC_GAME_C4) {           /* games/ConnectFourNode.hpp:8-13 */
        g.rows = 6; g.cols = 7; g.cells = 42; g.A = 7; g.nsym = 2; g.hist = 1;
    } else if (game == ORC_GAME_GO7) {   /* games/GoNode.hpp:16-22: 7x7, 8-ply history, 49 + pass */
        g.rows = 7; g.cols = 7; g.cells = 49; g.A = 50; g.nsym = 8; g.hist = 8;
    } else if (game == ORC_GAME_GO9) {   /* the same rules at width 9, komi 7.5 (BASELINE config 4); pinned by tests/golden/g_go9.npz = the reference compiled with only GO_BOARD_WIDTH / GO_KOMI changed */
        g.rows = 9; g.cols = 9; g.cells = 81; g.A = 82; g.nsym = 8; g.hist = 8;
    } else if (game == ORC_GAME_GO19) {  /* width 19 (BASELINE config 5) */
        g.rows = 19; g.cols = 19; g.cells = 361; g.A = 362; g.nsym = 8; g.hist = 8;
    } else {                             /* games/OthelloNode.hpp:8-11 */
        g.rows = 8; g.cols = 8; g.cells = 64; g.A = 65; g.nsym = 8; g.hist = 1;
    }
    return g;
}

int orc_game_cells(int game) { return geom(game).cells; }
int orc_game_actions(int game) { return geom(game).A; }
int orc_game_nsym(int game) { return geom(game).nsym; }
int orc_game_rows(int game) { return geom(game).rows; }
int orc_game_cols(int game) { return geom(game).cols; }
int orc_game_hist(int game) { return geom(game).hist; }

/* ------------------------------------------------------------------------------------------ */
/* math dispatch                                                                               */
/* ------------------------------------------------------------------------------------------ */

static float m_logf(float x, int mode) { return mode == ORC_MATH_LIBM ? logf(x) : sprl_logf(x); }
static float m_powf(float x, float y, int mode) { return mode == ORC_MATH_LIBM ? powf(x, y) : sprl_powf(x, y); }
static float m_expf(float x, int mode) { return mode == ORC_MATH_LIBM ? expf(x) : sprl_expf(x); }

/* ------------------------------------------------------------------------------------------ */
/* RNG — utils/random.hpp:32-111, utils/random.cpp:61-98 + libstdc++ (GCC 11) distributions     */
/* ------------------------------------------------------------------------------------------ */

static uint32_t mix_bits(uint64_t x) {                    /* random.hpp:76-80 */
    uint32_t xor_shifted = (uint32_t)(((x >> 18) ^ x) >> 27);
    uint32_t rot = (uint32_t)(x >> 59);
    return (xor_shifted >> rot) | (xor_shifted << ((0u - rot) & 31u));
}

uint32_t orc_rng_next(orc_rng* r) {                       /* random.hpp:99-103 */
    uint32_t out = mix_bits(r->state);
    r->state = r->state * 6364136223846793005ULL + r->inc;
    return out;
}

void orc_rng_seed(orc_rng* r, uint64_t seed, int stream) { /* random.hpp:92-97 */
    r->state = 0;
    r->inc = ((uint64_t)(int64_t)stream << 1) | 1u;
    orc_rng_next(r);
    r->state += seed;
    orc_rng_next(r);
}

/* std::uniform_int_distribution<int>(a,b) on a 32-bit URBG: Lemire's nearly-divisionless
 * method (bits/uniform_int_dist.h, _S_nd<uint64_t>), called from random.cpp:76-79. */
int orc_uniform_int(orc_rng* r, int a, int b) {
    uint32_t urange = (uint32_t)b - (uint32_t)a;
    uint32_t ret;
    if (urange != 0xFFFFFFFFu) {
        uint32_t range = urange + 1u;
        uint64_t product = (uint64_t)orc_rng_next(r) * (uint64_t)range;
        uint32_t low = (uint32_t)product;
        if (low < range) {
            uint32_t threshold = (0u - range) % range;
            while (low < threshold) {
                product = (uint64_t)orc_rng_next(r) * (uint64_t)range;
                low = (uint32_t)product;
            }
        }
        ret = (uint32_t)(product >> 32);
    } else {
        ret = orc_rng_next(r);
    }
    return (int)(ret + (uint32_t)a);
}

/* std::generate_canonical<float,24> on a 32-bit URBG (bits/random.tcc): one draw, /2^32,
 * clamped below 1.  This is Random::operator() (random.hpp:64-66) and the _Adaptor used by
 * normal_distribution / gamma_distribution. */
float orc_uniform_float(orc_rng* r) {
    float sum = (float)orc_rng_next(r);
    float ret = sum / 4294967296.0f;
    if (ret >= 1.0f) ret = 0x1.fffffep-1f;
    return ret;
}

typedef struct { int saved_available; float saved; } normal_state;

/* std::normal_distribution<float>(0,1)::operator() — Marsaglia polar (bits/random.tcc). */
static float normal_draw(orc_rng* r, normal_state* ns, int mode) {
    float ret;
    if (ns->saved_available) {
        ns->saved_available = 0;
        ret = ns->saved;
    } else {
        float x, y, r2;
        do {
            x = (float)((double)(2.0f * orc_uniform_float(r)) - 1.0);
            y = (float)((double)(2.0f * orc_uniform_float(r)) - 1.0);
            r2 = x * x + y * y;
        } while (r2 > 1.0f || r2 == 0.0f);
        float mult = sqrtf(-2.0f * m_logf(r2, mode) / r2);
        ns->saved = x * mult;
        ns->saved_available = 1;
        ret = y * mult;
    }
    ret = ret * 1.0f + 0.0f;
    return ret;
}

/* std::gamma_distribution<float>(alpha, 1)::operator() — Marsaglia-Tsang (bits/random.tcc). */
static float gamma_draw(orc_rng* r, normal_state* ns, float alpha, int mode) {
    float malpha = alpha < 1.0f ? alpha + 1.0f : alpha;
    float a1 = malpha - 1.0f / 3.0f;
    float a2 = 1.0f / sqrtf(9.0f * a1);
    float u, v, n;
    do {
        do {
            n = normal_draw(r, ns, mode);
            v = 1.0f + a2 * n;
        } while (v <= 0.0f);
        v = v * v * v;
        u = orc_uniform_float(r);
    } while ((double)u > (double)1.0f - 0.0331 * (double)n * (double)n * (double)n * (double)n
             && ((double)m_logf(u, mode) > (0.5 * (double)n * (double)n
                                            + (double)a1 * (1.0 - (double)v + (double)m_logf(v, mode)))));
    if (alpha == malpha) {
        return a1 * v * 1.0f;
    } else {
        do {
            u = orc_uniform_float(r);
        } while (u == 0.0f);
        return m_powf(u, 1.0f / alpha, mode) * a1 * v * 1.0f;
    }
}

void orc_dirichlet(orc_rng* r, float alpha, int k, float* out, int math_mode) { /* random.cpp:61-74 */
    normal_state ns = { 0, 0.0f };
    float sum = 0;
    for (int i = 0; i < k; ++i) {
        out[i] = gamma_draw(r, &ns, alpha, math_mode);
        sum += out[i];
    }
    float norm = 1 / sum;
    for (int i = 0; i < k; ++i) out[i] *= norm;
}

int orc_sample_cdf(orc_rng* r, const float* cdf, int n) {  /* random.cpp:86-98 */
    float e;
    do {
        e = orc_uniform_float(r) * 1.0f + 0.0f;           /* uniform_real_distribution<float>(0,1) */
    } while (e == 0);
    float x = cdf[n - 1] * e;
    int i = 0;
    while (i < n && cdf[i] < x) ++i;                       /* std::lower_bound */
    return i;
}

/* ------------------------------------------------------------------------------------------ */
/* rules                                                                                       */
/* ------------------------------------------------------------------------------------------ */

static const int R_DELTA[8] = { 1, 1, 0, -1, -1, -1, 0, 1 };   /* OthelloNode.cpp:199-200 */
static const int C_DELTA[8] = { 0, 1, 1, 1, 0, -1, -1, -1 };

static int oth_in_bounds(int r, int c) { return 0 <= r && r < 8 && 0 <= c && c < 8; }

static int oth_can_capture(const int8_t* b, int row, int col, int piece) {  /* OthelloNode.cpp:226-252 */
    int opp = 1 - piece;
    for (int i = 0; i < 8; ++i) {
        int nr = row + R_DELTA[i], nc = col + C_DELTA[i];
        int opp_exists = 0;
        while (oth_in_bounds(nr, nc) && b[nr * 8 + nc] == opp) {
            nr += R_DELTA[i];
            nc += C_DELTA[i];
            opp_exists = 1;
        }
        if (opp_exists && oth_in_bounds(nr, nc) && b[nr * 8 + nc] == piece) return 1;
    }
    return 0;
}

static void oth_action_mask(const int8_t* b, int player, float* mask) {      /* OthelloNode.cpp:156-177 */
    for (int a = 0; a < 65; ++a) mask[a] = 0.0f;
    for (int row = 0; row < 8; ++row)
        for (int col = 0; col < 8; ++col) {
            if (b[row * 8 + col] != -1) continue;
            mask[row * 8 + col] = oth_can_capture(b, row, col, player) ? 1.0f : 0.0f;
        }
    int can_pass = 1;
    for (int i = 0; i < 64; ++i)
        if (mask[i] > 0.0f) { can_pass = 0; break; }
    mask[64] = can_pass ? 1.0f : 0.0f;
}

static int oth_is_terminal(const int8_t* b) {                                /* OthelloNode.cpp:179-191 */
    float m[65];
    oth_action_mask(b, 0, m);
    if (m[64] == 0.0f) return 0;
    oth_action_mask(b, 1, m);
    return m[64] > 0.0f;
}

static void oth_step(const int8_t* board, int player, int action, int8_t* nb, float* nmask,
                     int* terminal, int* winner) {                           /* OthelloNode.cpp:34-87 */
    memcpy(nb, board, 64);
    int piece = player;
    if (action != 64) {
        nb[action] = (int8_t)piece;
        int row = action / 8, col = action % 8;
        int opp = 1 - piece;
        int cap[64], ncap = 0;                                               /* captures(): :193-224 */
        for (int i = 0; i < 8; ++i) {
            int nr = row + R_DELTA[i], nc = col + C_DELTA[i];
            while (oth_in_bounds(nr, nc) && nb[nr * 8 + nc] == opp) {
                nr += R_DELTA[i];
                nc += C_DELTA[i];
            }
            if (oth_in_bounds(nr, nc) && nb[nr * 8 + nc] == piece) {
                for (int r = row + R_DELTA[i], c = col + C_DELTA[i]; r != nr || c != nc;
                     r += R_DELTA[i], c += C_DELTA[i])
                    cap[ncap++] = r * 8 + c;
            }
        }
        for (int i = 0; i < ncap; ++i) nb[cap[i]] = (int8_t)piece;
    }
    *winner = -1;
    *terminal = oth_is_terminal(nb);
    if (*terminal) {
        int c0 = 0, c1 = 0;
        for (int i = 0; i < 64; ++i) {
            if (nb[i] == 0) c0++;
            else if (nb[i] == 1) c1++;
        }
        if (c0 > c1) *winner = 0;
        if (c1 > c0) *winner = 1;
    }
    oth_action_mask(nb, 1 - player, nmask);
}

static int c4_check_win(const int8_t* b, int pr, int pc, int piece) {        /* ConnectFourNode.cpp:135-217 */
    static const int DR[4] = { 0, 1, 1, 1 }, DC[4] = { 1, 0, 1, -1 };
    for (int d = 0; d < 4; ++d) {
        int count = 1;
        int r = pr - DR[d], c = pc - DC[d];
        while (r >= 0 && r < 6 && c >= 0 && c < 7 && b[r * 7 + c] == piece) { count++; r -= DR[d]; c -= DC[d]; }
        r = pr + DR[d]; c = pc + DC[d];
        while (r >= 0 && r < 6 && c >= 0 && c < 7 && b[r * 7 + c] == piece) { count++; r += DR[d]; c += DC[d]; }
        if (count >= 4) return 1;
    }
    return 0;
}

static void c4_step(const int8_t* board, int player, const float* mask, int action, int8_t* nb,
                    float* nmask, int* terminal, int* winner) {              /* ConnectFourNode.cpp:23-78 */
    memcpy(nb, board, 42);
    memcpy(nmask, mask, 7 * sizeof(float));
    int col = action;
    int row = 5;
    while (row >= 0 && nb[row * 7 + col] != -1) row--;
    nb[row * 7 + col] = (int8_t)player;
    if (row == 0) nmask[col] = 0.0f;
    *winner = c4_check_win(nb, row, col, player) ? player : -1;
    int filled = 1;
    for (int c = 0; c < 7; ++c)
        if (nb[c] == -1) { filled = 0; break; }
    *terminal = (*winner != -1) || filled;
    if (*terminal)
        for (int c = 0; c < 7; ++c) nmask[c] = 0.0f;
}


/* ------------------------------------------------------------------------------------------ */
/* Go 7x7 — games/GoNode.{hpp,cpp}, utils/DSU.hpp, utils/Zobrist.hpp                            */
/* ------------------------------------------------------------------------------------------ */
/* The reference fixes the board at 7x7 (GoNode.hpp:16); 9x9 is the same code with the width as a parameter
 * (komi 7.5 there, games/GoDesc.md:127-128).  Only the 7x7 instance is pinned against the reference build. */
#define GO_NMAX 361                             /* 19x19: index types widened to int16 */
#define GO_W (s->w)
#define GO_N (s->w * s->w)
#define GO_PASS GO_N
#define GO_MAX_DEPTH (2 * GO_N)                  /* GoNode.hpp:22 */
#define GO_KOMI (s->w == 7 ? 9.0f : 7.5f)        /* GoNode.hpp:20 */

typedef struct {
    int w;                                       /* board width */
    int8_t board[GO_NMAX];
    int16_t dsu[GO_NMAX];                        /* utils/DSU.hpp (path compression omitted: same sets) */
    int16_t libs[GO_NMAX];                       /* valid at group roots */
    uint64_t comp[GO_NMAX];                      /* per-group Zobrist value, valid at group roots */
    uint64_t hash;
    int depth;
    int action;                                  /* action that led here (m_action), 0 at the start node */
    uint64_t hist[2 * GO_NMAX + 2];              /* m_zobristHistorySet: hashes after every placement */
    int nhist;
} go_state;

/* The reference draws its Zobrist table from the process-global RNG at static-init time (Zobrist.hpp:42-47),
 * i.e. it is random per process and only matters through collisions; any fixed table is equivalent. */
static uint64_t go_zobrist(int coord, int piece) {
    uint64_t z = 0x9E3779B97F4A7C15ULL * (uint64_t)(coord + piece * GO_NMAX + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static int go_neighbors(const go_state* s, int c, int* out) {                                   /* GoNode.hpp:117-130 */
    int n = 0, row = c / GO_W, col = c % GO_W;
    if (row > 0) out[n++] = c - GO_W;
    if (col > 0) out[n++] = c - 1;
    if (row < GO_W - 1) out[n++] = c + GO_W;
    if (col < GO_W - 1) out[n++] = c + 1;
    return n;
}
static int go_find(const go_state* s, int x) {
    while (s->dsu[x] != x) x = s->dsu[x];
    return x;
}
static int go_has_hash(const go_state* s, uint64_t h) {
    for (int i = 0; i < s->nhist; ++i)
        if (s->hist[i] == h) return 1;
    return 0;
}
static int go_compute_liberties(const go_state* s, int coord) {              /* GoNode.cpp:10-52 */
    int piece = s->board[coord];
    if (piece == -1) return 0;
    int visited[GO_NMAX] = { 0 }, q[GO_NMAX], qh = 0, qt = 0, libs = 0, nb[4];
    visited[coord] = 1;
    q[qt++] = coord;
    while (qh < qt) {
        int cur = q[qh++];
        int k = go_neighbors(s, cur, nb);
        for (int i = 0; i < k; ++i) {
            int n = nb[i];
            if (s->board[n] == piece) {
                if (!visited[n]) { visited[n] = 1; q[qt++] = n; }
            } else if (s->board[n] == -1) {
                if (!visited[n]) { visited[n] = 1; ++libs; }
            }
        }
    }
    return libs;
}
static void go_clear_component(go_state* s, int coord, int piece) {          /* GoNode.cpp:54-94 */
    s->board[coord] = -1;
    s->dsu[coord] = (int16_t)coord;
    s->libs[go_find(s, coord)] = 0;
    s->comp[go_find(s, coord)] = 0;
    int groups[4], ng = 0, nb[4];
    int k = go_neighbors(s, coord, nb);
    for (int i = 0; i < k; ++i) {
        int n = nb[i];
        if (s->board[n] == -1) continue;
        if (s->board[n] == piece) {
            go_clear_component(s, n, piece);
        } else {
            int grp = go_find(s, n), seen = 0;
            for (int j = 0; j < ng; ++j)
                if (groups[j] == grp) seen = 1;
            if (seen) continue;
            groups[ng++] = grp;
            s->libs[grp]++;
        }
    }
}
static void go_place(go_state* s, int coord, int piece) {                    /* GoNode.cpp:96-176 */
    int nb[4];
    s->board[coord] = (int8_t)piece;
    uint64_t new_comp = go_zobrist(coord, piece);
    int k = go_neighbors(s, coord, nb);
    for (int i = 0; i < k; ++i) {
        int n = nb[i];
        if (s->board[n] == piece) {
            if (go_find(s, n) == go_find(s, coord)) continue;
            new_comp ^= s->comp[go_find(s, n)];
            int rx = go_find(s, n), ry = go_find(s, coord);                  /* DSU::unite(neighbor, coord) */
            if (rx != ry) s->dsu[rx] = (int16_t)ry;
        }
    }
    s->comp[go_find(s, coord)] = new_comp;
    s->libs[go_find(s, coord)] = (int16_t)go_compute_liberties(s, coord);
    uint64_t update = go_zobrist(coord, piece);
    int groups[4], ng = 0;
    for (int i = 0; i < k; ++i) {
        int n = nb[i];
        if (s->board[n] == 1 - piece) {
            int grp = go_find(s, n), seen = 0;
            for (int j = 0; j < ng; ++j)
                if (groups[j] == grp) seen = 1;
            if (seen) continue;
            groups[ng++] = grp;
            s->libs[grp]--;
            if (s->libs[go_find(s, grp)] == 0) {
                update ^= s->comp[go_find(s, grp)];
                go_clear_component(s, grp, 1 - piece);
            }
        }
    }
    s->hash ^= update;
    s->hist[s->nhist++] = s->hash;
}
static int go_legal(const go_state* s, int coord, int piece) {               /* GoNode.cpp:178-228 */
    if (s->board[coord] != -1) return 0;
    uint64_t nh = s->hash ^ go_zobrist(coord, piece);
    int has_libs = 0, groups[4], ng = 0, nb[4];
    int k = go_neighbors(s, coord, nb);
    for (int i = 0; i < k; ++i) {
        int n = nb[i];
        if (s->board[n] == -1) {
            has_libs = 1;
        } else if (s->board[n] == piece) {
            if (s->libs[go_find(s, n)] > 1) has_libs = 1;
        } else {
            if (s->libs[go_find(s, n)] == 1) {
                has_libs = 1;
                int grp = go_find(s, n), seen = 0;
                for (int j = 0; j < ng; ++j)
                    if (groups[j] == grp) seen = 1;
                if (seen) continue;
                groups[ng++] = grp;
                nh ^= s->comp[grp];
            }
        }
    }
    return has_libs && !go_has_hash(s, nh);
}
static void go_territory(const go_state* s, int* terr) {                     /* GoNode.cpp:230-290 */
    int visited[GO_NMAX] = { 0 }, nb[4];
    terr[0] = terr[1] = 0;
    for (int i = 0; i < GO_N; ++i) {
        if (s->board[i] == 0) { terr[0]++; continue; }
        if (s->board[i] == 1) { terr[1]++; continue; }
        if (visited[i]) continue;
        int q[GO_NMAX], qh = 0, qt = 0, count = 0, poss0 = 1, poss1 = 1;
        visited[i] = 1;
        q[qt++] = i;
        while (qh < qt) {
            int cur = q[qh++];
            ++count;
            int k = go_neighbors(s, cur, nb);
            for (int j = 0; j < k; ++j) {
                int n = nb[j];
                if (s->board[n] == 0) poss1 = 0;
                else if (s->board[n] == 1) poss0 = 0;
                else if (!visited[n]) { visited[n] = 1; q[qt++] = n; }
            }
        }
        if (poss0 && !poss1) terr[0] += count;
        if (poss1 && !poss0) terr[1] += count;
    }
}
static void go_start(go_state* s, int width) {                               /* GoNode.cpp:303-317 */
    memset(s, 0, sizeof(*s));
    s->w = width;
    memset(s->board, -1, GO_NMAX);
    for (int i = 0; i < GO_NMAX; ++i) s->dsu[i] = (int16_t)i;
}
static void go_next(const go_state* p, int player, int action, go_state* c, float* mask, int* terminal,
                    int* winner) {                                           /* GoNode.cpp:319-383 */
    const go_state* s = p;
    *c = *p;
    if (action != GO_PASS) go_place(c, action, player);
    c->action = action;
    c->depth = p->depth + 1;
    *terminal = (p->action == GO_PASS && action == GO_PASS) || c->depth >= GO_MAX_DEPTH;
    *winner = -1;
    for (int a = 0; a <= GO_N; ++a) mask[a] = 0.0f;
    if (!*terminal) {
        for (int i = 0; i < GO_N; ++i) mask[i] = (float)go_legal(c, i, 1 - player);
        mask[GO_PASS] = 1.0f;                                                /* :298 */
    } else {
        int terr[2];
        go_territory(c, terr);
        float s0 = (float)terr[0], s1 = (float)terr[1];
        s1 += GO_KOMI;
        if ((double)s0 > (double)s1 + 0.1) *winner = 0;
        else if ((double)s1 > (double)s0 + 0.1) *winner = 1;
    }
}

void orc_start(int game, int8_t* board, int* player, float* mask) {
    geom_t g = geom(game);
    memset(board, -1, (size_t)g.cells);
    *player = 0;
    if (game == ORC_GAME_C4) {                                               /* ConnectFourNode.cpp:13-21 */
        for (int a = 0; a < 7; ++a) mask[a] = 1.0f;
    } else if (game == ORC_GAME_GO7 || game == ORC_GAME_GO9 || game == ORC_GAME_GO19) {               /* GoNode.cpp:306 */
        for (int a = 0; a < g.A; ++a) mask[a] = 1.0f;
    } else {                                                                 /* OthelloNode.cpp:18-32 */
        board[3 * 8 + 3] = 1;
        board[3 * 8 + 4] = 0;
        board[4 * 8 + 3] = 0;
        board[4 * 8 + 4] = 1;
        oth_action_mask(board, 0, mask);
    }
}

void orc_step(int game, const int8_t* board, int player, const float* mask, int action,
              int8_t* board_out, float* mask_out, int* terminal_out, int* winner_out) {
    if (game == ORC_GAME_C4) c4_step(board, player, mask, action, board_out, mask_out, terminal_out, winner_out);
    else oth_step(board, player, action, board_out, mask_out, terminal_out, winner_out);
}

static void rewards_of(int winner, float* rw) {                              /* OthelloNode.cpp:94-100 */
    rw[0] = winner == 0 ? 1.0f : (winner == 1 ? -1.0f : 0.0f);
    rw[1] = winner == 1 ? 1.0f : (winner == 0 ? -1.0f : 0.0f);
}

int orc_playout(int game, uint64_t seed, int stream, int max_plies, int8_t* boards, int8_t* players,
                int16_t* actions, float* masks, int8_t* terminal, float* rewards) {
    /* boards: [max_plies][hist][cells]; for Go the 8-ply history is filled newest first, missing plies = -2 */
    geom_t g = geom(game);
    orc_rng rng;
    orc_rng_seed(&rng, seed, stream);
    int8_t b[ORC_MAX_CELLS], nb[ORC_MAX_CELLS];
    float m[ORC_MAX_A], nm[ORC_MAX_A];
    int player, term = 0, winner = -1, ply = 0;
    go_state* gs = NULL;
    orc_start(game, b, &player, m);
    if (game == ORC_GAME_GO7 || game == ORC_GAME_GO9 || game == ORC_GAME_GO19) {
        gs = (go_state*)malloc(sizeof(go_state) * (size_t)(max_plies + 1));
        go_start(&gs[0], g.cols);
    }
    const size_t stride = (size_t)g.hist * g.cells;
    for (;;) {
        memset(boards + (size_t)ply * stride, -2, stride);
        for (int t = 0; t < g.hist && t <= ply; ++t)
            memcpy(boards + (size_t)ply * stride + (size_t)t * g.cells,
                   gs ? gs[ply - t].board : b, (size_t)g.cells);
        players[ply] = (int8_t)player;
        memcpy(masks + (size_t)ply * g.A, m, (size_t)g.A * sizeof(float));
        terminal[ply] = (int8_t)term;
        rewards_of(winner, rewards + 2 * ply);
        if (term || ply + 1 >= max_plies) {
            actions[ply] = -1;
            free(gs);
            return ply + 1;
        }
        int legal[ORC_MAX_A], nl = 0;
        for (int a = 0; a < g.A; ++a)
            if (m[a] > 0.0f) legal[nl++] = a;
        int a = legal[orc_uniform_int(&rng, 0, nl - 1)];
        actions[ply] = (int16_t)a;
        if (gs) {
            go_next(&gs[ply], player, a, &gs[ply + 1], nm, &term, &winner);
            memcpy(nb, gs[ply + 1].board, (size_t)g.cells);
        } else {
            orc_step(game, b, player, m, a, nb, nm, &term, &winner);
        }
        memcpy(b, nb, (size_t)g.cells);
        memcpy(m, nm, (size_t)g.A * sizeof(float));
        player = 1 - player;
        ++ply;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* symmetries — symmetry/D4GridSymmetrizer.hpp:30-117, symmetry/ConnectFourSymmetrizer.cpp       */
/* ------------------------------------------------------------------------------------------ */

static void d4_map(int sym, int W, int r, int c, int* tr, int* tc) {         /* D4GridSymmetrizer.hpp:108-117 */
    switch (sym) {
    case 0: *tr = r; *tc = c; break;
    case 1: *tr = c; *tc = W - 1 - r; break;
    case 2: *tr = W - 1 - r; *tc = W - 1 - c; break;
    case 3: *tr = W - 1 - c; *tc = r; break;
    case 4: *tr = r; *tc = W - 1 - c; break;
    case 5: *tr = W - 1 - c; *tc = W - 1 - r; break;
    case 6: *tr = W - 1 - r; *tc = c; break;
    default: *tr = c; *tc = r; break;
    }
}

void orc_symmetrize_board(int game, int sym, const int8_t* in, int8_t* out) {
    geom_t g = geom(game);
    if (game == ORC_GAME_C4) {
        for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 7; ++c) out[r * 7 + (sym == 1 ? 6 - c : c)] = in[r * 7 + c];
        return;
    }
    for (int r = 0; r < g.rows; ++r)
        for (int c = 0; c < g.cols; ++c) {
            int tr, tc;
            d4_map(sym, g.cols, r, c, &tr, &tc);
            out[tr * g.cols + tc] = in[r * g.cols + c];
        }
}

void orc_symmetrize_dist(int game, int sym, const float* in, float* out) {
    geom_t g = geom(game);
    if (game == ORC_GAME_C4) {
        for (int c = 0; c < 7; ++c) out[sym == 1 ? 6 - c : c] = in[c];
        return;
    }
    for (int r = 0; r < g.rows; ++r)
        for (int c = 0; c < g.cols; ++c) {
            int tr, tc;
            d4_map(sym, g.cols, r, c, &tr, &tc);
            out[tr * g.cols + tc] = in[r * g.cols + c];
        }
    out[g.cells] = in[g.cells];                                              /* pass fixed: :91-92 */
}

int orc_inverse_symmetry(int game, int sym) {
    static const int INV[8] = { 0, 3, 2, 1, 4, 5, 6, 7 };                    /* D4GridSymmetrizer.hpp:43-46 */
    if (game == ORC_GAME_C4) return sym;
    return INV[sym];
}

/* ------------------------------------------------------------------------------------------ */
/* evaluators                                                                                  */
/* ------------------------------------------------------------------------------------------ */

void orc_encode_planes(int game, int n, const int8_t* boards, const int8_t* players, const int8_t* sizes,
                       float* planes) {
    geom_t g = geom(game);                                                   /* GridNetwork.hpp:72-97 */
    int P = 2 * g.hist + 1;
    for (int b = 0; b < n; ++b) {
        float* out = planes + (size_t)b * P * g.cells;
        int ours = players[b];
        int size = sizes ? sizes[b] : g.hist;
        memset(out, 0, (size_t)P * g.cells * sizeof(float));
        for (int t = 0; t < size; ++t)
            for (int i = 0; i < g.cells; ++i) {
                int8_t v = boards[((size_t)b * g.hist + t) * g.cells + i];
                if (v == ours) out[(2 * t) * g.cells + i] = 1.0f;
                else if (v == 1 - ours) out[(2 * t + 1) * g.cells + i] = 1.0f;
            }
        for (int i = 0; i < g.cells; ++i) out[(2 * g.hist) * g.cells + i] = ours == 0 ? 1.0f : 0.0f;
    }
}

void orc_decode_policy(int A, const float* logits, const float* mask, float* policy, int math_mode) {
    int num_legal = 0;                                                       /* GridNetwork.hpp:104-138 */
    for (int i = 0; i < A; ++i) policy[i] = m_expf(logits[i], math_mode);
    for (int i = 0; i < A; ++i) {
        if (mask[i] == 0.0f) policy[i] = 0.0f;
        else ++num_legal;
    }
    float sum = 0.0f;
    for (int i = 0; i < A; ++i) sum += policy[i];
    if (sum == 0.0f) {
        float uniform = 1.0f / num_legal;
        for (int i = 0; i < A; ++i) policy[i] = mask[i] == 0.0f ? 0.0f : uniform;
    } else {
        float inv = 1.0f / sum;                                              /* GameActionDist.hpp:284-289 */
        for (int i = 0; i < A; ++i) policy[i] = policy[i] * inv;
    }
}

void orc_evaluate(const orc_config* cfg, int n, const int8_t* boards, const int8_t* players, const int8_t* sizes,
                  const float* masks, float* policies, float* values) {
    geom_t g = geom(cfg->game);
    if (cfg->eval_kind == ORC_EVAL_CALLBACK) {
        int P = 2 * g.hist + 1;
        float* planes = (float*)malloc((size_t)n * P * g.cells * sizeof(float));
        float* logits = (float*)malloc((size_t)n * g.A * sizeof(float));
        orc_encode_planes(cfg->game, n, boards, players, sizes, planes);
        cfg->forward(cfg->forward_user, n, planes, logits, values);
        for (int b = 0; b < n; ++b)
            orc_decode_policy(g.A, logits + (size_t)b * g.A, masks + (size_t)b * g.A,
                              policies + (size_t)b * g.A, cfg->math_mode);
        free(planes);
        free(logits);
        return;
    }
    for (int b = 0; b < n; ++b) {                                            /* RandomNetwork.hpp:21-49 */
        const float* mask = masks + (size_t)b * g.A;
        float* pol = policies + (size_t)b * g.A;
        int num_legal = 0;
        for (int i = 0; i < g.A; ++i)
            if (mask[i] > 0.0f) ++num_legal;
        float uniform = 1.0f / num_legal;
        for (int i = 0; i < g.A; ++i) pol[i] = mask[i] > 0.0f ? uniform : 0.0f;
        values[b] = 0.0f;
        if (cfg->eval_kind == ORC_EVAL_HEURISTIC && cfg->game == ORC_GAME_OTHELLO) {
            const int8_t* bd = boards + (size_t)b * g.hist * g.cells;        /* OthelloHeuristic.cpp:28-49 */
            int num_empty = 0;
            for (int i = 0; i < 64; ++i)
                if (bd[i] == -1) ++num_empty;
            float opp_mask[65];
            oth_action_mask(bd, 1 - players[b], opp_mask);
            int num_opp = 0;
            for (int i = 0; i < 64; ++i)
                if (opp_mask[i] > 0.0f) ++num_opp;
            values[b] = (float)(num_legal - num_opp) / num_empty;
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* UCT tree — uct/UCTNode.hpp, uct/UCTTree.hpp                                                 */
/* ------------------------------------------------------------------------------------------ */

typedef struct node {
    struct node* parent;
    struct node* children[ORC_MAX_A];
    int action;
    int8_t board[ORC_MAX_CELLS];
    int player, winner, terminal;
    float mask[ORC_MAX_A];
    int expanded, evaluated;
    float net_policy[ORC_MAX_A];
    float net_value;
    float P[ORC_MAX_A], W[ORC_MAX_A], N[ORC_MAX_A];  /* EdgeStatistics, UCTNode.hpp:45-60 */
    float* own_N;                                    /* UCTNode.hpp:142 */
    float* own_W;                                    /* UCTNode.hpp:147 */
    go_state* go;                                    /* Go only: DSU / liberties / hashes (GoNode members) */
} node;

typedef struct {
    const orc_config* cfg;
    geom_t g;
    orc_rng* rng;
    orc_stats* st;
    float dummy_N[ORC_MAX_A], dummy_W[ORC_MAX_A];    /* UCTTree.hpp:301 */
    node* root;
    node* decision;
    int64_t live;
} tree;

static node* node_new(tree* t, node* parent, int action) {
    node* n = (node*)calloc(1, sizeof(node));
    n->parent = parent;
    n->action = action;
    if (parent) {
        if (parent->go) {
            n->go = (go_state*)malloc(sizeof(go_state));
            go_next(parent->go, parent->player, action, n->go, n->mask, &n->terminal, &n->winner);
            memcpy(n->board, n->go->board, (size_t)t->g.cells);
        } else {
            orc_step(t->cfg->game, parent->board, parent->player, parent->mask, action, n->board, n->mask,
                     &n->terminal, &n->winner);
        }
        n->player = 1 - parent->player;
        n->own_N = &parent->N[action];
        n->own_W = &parent->W[action];
    } else {
        orc_start(t->cfg->game, n->board, &n->player, n->mask);
        if (t->cfg->game == ORC_GAME_GO7 || t->cfg->game == ORC_GAME_GO9 || t->cfg->game == ORC_GAME_GO19) {
            n->go = (go_state*)malloc(sizeof(go_state));
            go_start(n->go, t->g.cols);
        }
        n->winner = -1;
        n->own_N = &t->dummy_N[0];
        n->own_W = &t->dummy_W[0];
    }
    if (t->st) {
        t->st->nodes_created++;
        t->live++;
        if (t->live > t->st->max_live_nodes) t->st->max_live_nodes = t->live;
    }
    return n;
}

static int64_t node_free(node* n, int A) {
    if (!n) return 0;
    int64_t c = 1;
    for (int a = 0; a < A; ++a) c += node_free(n->children[a], A);
    free(n->go);
    free(n);
    return c;
}

static int best_action(tree* t, node* n) {                                   /* UCTNode.hpp:221-251 */
    int ties[ORC_MAX_A], nt = 0;
    float best = -INFINITY;
    float uw = t->cfg->u_weight;
    for (int a = 0; a < t->g.A; ++a) {
        if (n->mask[a] == 0.0f) continue;
        float q = n->W[a] / (1 + n->N[a]);                                   /* :200 */
        float u = n->P[a] * sqrtf(*n->own_N) / (1 + n->N[a]);                /* :210 */
        float value = q + uw * u;                                            /* :236 */
        if (value > best) {
            best = value;
            nt = 0;
            ties[nt++] = a;
        } else if (value == best) {
            ties[nt++] = a;
        }
    }
    return ties[orc_uniform_int(t->rng, 0, nt - 1)];
}

static node* get_add_child(tree* t, node* n, int action) {                   /* UCTNode.hpp:258-284 */
    if (!n->children[action]) {
        n->children[action] = node_new(t, n, action);
        n->W[action] = (t->cfg->init_q == 0 && n->evaluated) ? n->net_value : 0.0f;   /* InitQ::PARENT / ZERO */
    }
    return n->children[action];
}

static void expand(tree* t, node* n, int add_noise) {                        /* UCTNode.hpp:312-348 */
    n->expanded = 1;
    int num_legal = 0;
    for (int a = 0; a < t->g.A; ++a) {
        if (n->mask[a] == 0.0f) continue;
        n->P[a] = n->net_policy[a];
        ++num_legal;
    }
    if (add_noise) {
        float noise[ORC_MAX_A];
        orc_dirichlet(t->rng, t->cfg->dir_alpha, num_legal, noise, t->cfg->math_mode);
        int read = 0;
        float eps = t->cfg->dir_eps;
        for (int a = 0; a < t->g.A; ++a) {
            if (n->mask[a] == 0.0f) continue;
            n->P[a] = (float)((1.0 - (double)eps) * (double)n->P[a] + (double)(eps * noise[read]));
            ++read;
        }
    }
    if (t->st) t->st->expansions++;
}

static node* select_leaf(tree* t) {                                          /* UCTTree.hpp:225-249 */
    node* cur = t->decision;
    while (cur->expanded && !cur->terminal) {
        int a = best_action(t, cur);
        *cur->own_N = *cur->own_N + 1;
        *cur->own_W = *cur->own_W - 1;
        cur = get_add_child(t, cur, a);
        if (t->st) t->st->levels++;
    }
    *cur->own_N = *cur->own_N + 1;
    *cur->own_W = *cur->own_W - 1;
    return cur;
}

static void backup(tree* t, node* n, float value_estimate) {                 /* UCTTree.hpp:261-273 */
    float estimate = -value_estimate * (float)(n->player == 0 ? 1 : -1);
    node* cur = n;
    while (cur != t->decision->parent) {
        *cur->own_W += 1 + estimate * (float)(cur->player == 0 ? 1 : -1);
        cur = cur->parent;
    }
}

static int search_and_get_leaves(tree* t, node** leaves, int* nleaves) {     /* UCTTree.hpp:76-114 */
    int traversals = 0;
    *nleaves = 0;
    while (traversals < t->cfg->max_batch) {
        ++traversals;
        node* leaf = select_leaf(t);
        if (leaf->terminal) {
            float rw[2];
            rewards_of(leaf->winner, rw);
            backup(t, leaf, rw[leaf->player]);
            if (t->st) t->st->terminal_hits++;
            continue;
        } else if (leaf->evaluated) {
            expand(t, leaf, t->cfg->add_noise && leaf == t->decision);
            backup(t, leaf, leaf->net_value);
            if (t->st) t->st->gray_hits++;
            continue;
        } else {
            leaves[(*nleaves)++] = leaf;
        }
        if (*nleaves >= t->cfg->max_queue) break;
    }
    return traversals;
}

/* getGameState: the node's board, and for Go up to 8 boards walking the parents, newest first
 * (GoNode.cpp:385-398); returns the number of valid plies. */
static int node_state(const tree* t, const node* n, int8_t* boards) {
    int size = 0;
    memset(boards, -2, (size_t)t->g.hist * t->g.cells);
    for (const node* cur = n; cur && size < t->g.hist; cur = cur->parent, ++size)
        memcpy(boards + (size_t)size * t->g.cells, cur->board, (size_t)t->g.cells);
    return size;
}

static void symmetrize_state(const tree* t, int sym, int size, const int8_t* in, int8_t* out) {
    memset(out, -2, (size_t)t->g.hist * t->g.cells);
    for (int p = 0; p < size; ++p)                                           /* D4GridSymmetrizer.hpp:63-71 */
        orc_symmetrize_board(t->cfg->game, sym, in + (size_t)p * t->g.cells, out + (size_t)p * t->g.cells);
}

static void evaluate_and_backprop(tree* t, node** leaves, int n) {           /* UCTTree.hpp:124-184 */
    geom_t g = t->g;
    const size_t stride = (size_t)g.hist * g.cells;
    int8_t* boards = (int8_t*)malloc((size_t)n * stride);
    int8_t* sizes = (int8_t*)malloc((size_t)n);
    int8_t* players = (int8_t*)malloc((size_t)n);
    float* masks = (float*)malloc((size_t)n * g.A * sizeof(float));
    float* policies = (float*)malloc((size_t)n * g.A * sizeof(float));
    float* values = (float*)malloc((size_t)n * sizeof(float));
    int syms[64];
    for (int i = 0; i < n; ++i) {
        sizes[i] = (int8_t)node_state(t, leaves[i], boards + (size_t)i * stride);
        players[i] = (int8_t)leaves[i]->player;
        memcpy(masks + (size_t)i * g.A, leaves[i]->mask, (size_t)g.A * sizeof(float));
        syms[i] = 0;
    }
    if (t->cfg->use_sym) {
        for (int i = 0; i < n; ++i) {
            syms[i] = orc_uniform_int(t->rng, 0, g.nsym - 1);
            int8_t tmp[ORC_MAX_HIST * ORC_MAX_CELLS];
            memcpy(tmp, boards + (size_t)i * stride, stride);
            symmetrize_state(t, syms[i], sizes[i], tmp, boards + (size_t)i * stride);
            if (t->cfg->mask_frame == ORC_MASK_SYMMETRISED)
                orc_symmetrize_dist(t->cfg->game, syms[i], leaves[i]->mask, masks + (size_t)i * g.A);
        }
    }
    orc_evaluate(t->cfg, n, boards, players, sizes, masks, policies, values);
    if (t->st) t->st->nn_evals += n;
    for (int i = 0; i < n; ++i) {
        node* leaf = leaves[i];
        float policy[ORC_MAX_A];
        memcpy(policy, policies + (size_t)i * g.A, (size_t)g.A * sizeof(float));
        if (t->cfg->use_sym) {
            float tmp[ORC_MAX_A];
            memset(tmp, 0, sizeof(tmp));
            orc_symmetrize_dist(t->cfg->game, orc_inverse_symmetry(t->cfg->game, syms[i]), policy, tmp);
            memcpy(policy, tmp, (size_t)g.A * sizeof(float));
        }
        if (!leaf->evaluated) {                                              /* addNetworkOutput, UCTNode.hpp:292-301 */
            leaf->evaluated = 1;
            memcpy(leaf->net_policy, policy, (size_t)g.A * sizeof(float));
            leaf->net_value = values[i];
        } else if (t->st) {
            t->st->dup_hits++;
        }
        if (!leaf->expanded) expand(t, leaf, t->cfg->add_noise && leaf == t->decision);
        backup(t, leaf, leaf->net_value);
    }
    free(boards); free(sizes); free(players); free(masks); free(policies); free(values);
}

static void clear_subtree(tree* t, node* n) {                                /* UCTTree.hpp:283-298 */
    if (!n->expanded) return;
    memset(n->P, 0, sizeof(n->P));
    memset(n->W, 0, sizeof(n->W));
    memset(n->N, 0, sizeof(n->N));
    n->expanded = 0;
    for (int a = 0; a < t->g.A; ++a)
        if (n->children[a]) clear_subtree(t, n->children[a]);
}

static void advance_decision(tree* t, int action) {                          /* UCTTree.hpp:197-210 */
    node* d = t->decision;
    for (int a = 0; a < t->g.A; ++a)                                         /* UCTNode.hpp:356-366 */
        if (a != action && d->children[a]) {
            t->live -= node_free(d->children[a], t->g.A);
            d->children[a] = NULL;
        }
    node* child = get_add_child(t, d, action);
    clear_subtree(t, child);
    t->decision = child;
    t->live -= 1;  /* the old decision node no longer belongs to the searchable subtree */
}

static void tree_init(tree* t, const orc_config* cfg, orc_rng* rng, orc_stats* st) {
    memset(t, 0, sizeof(*t));
    t->cfg = cfg;
    t->g = geom(cfg->game);
    t->rng = rng;
    t->st = st;
    t->root = node_new(t, NULL, 0);
    t->decision = t->root;
}

static int search_move(tree* t) {                                            /* SelfPlay.hpp:99-108 */
    node* leaves[64];
    int traversals = 0;
    while (traversals < t->cfg->num_traversals) {
        int nl = 0;
        int trav = search_and_get_leaves(t, leaves, &nl);
        if (nl > 0) evaluate_and_backprop(t, leaves, nl);
        traversals += trav;
    }
    if (t->st) t->st->traversals += traversals;
    return traversals;
}

int orc_search_trace(const orc_config* cfg, int moves, uint64_t seed, int stream,
                     float* stats, int32_t* trav, int16_t* chosen) {
    orc_rng rng;
    orc_rng_seed(&rng, seed, stream);
    tree t;
    tree_init(&t, cfg, &rng, NULL);
    int A = t.g.A, m = 0;
    for (; m < moves && !t.decision->terminal; ++m) {
        trav[m] = search_move(&t);
        int best = 0;
        for (int a = 0; a < A; ++a) {
            stats[(m * 3 + 0) * A + a] = t.decision->N[a];
            stats[(m * 3 + 1) * A + a] = t.decision->W[a];
            stats[(m * 3 + 2) * A + a] = t.decision->P[a];
            if (t.decision->N[a] > t.decision->N[best]) best = a;
        }
        chosen[m] = (int16_t)best;
        advance_decision(&t, best);
    }
    node_free(t.root, A);
    return m;
}

int orc_match(const orc_config* cfg0, const orc_config* cfg1, int num_games, uint64_t seed, int stream_base,
              int8_t* winners, int16_t* actions, int32_t* nplies, int max_plies) {
    for (int g = 0; g < num_games; ++g) {
        orc_rng rng;
        orc_rng_seed(&rng, seed, stream_base + g);
        tree t0, t1;
        tree_init(&t0, cfg0, &rng, NULL);
        tree_init(&t1, cfg1, &rng, NULL);
        tree* agents[2];                                                     /* Evaluate.cpp:126-130 */
        agents[0] = (g % 2 == 0) ? &t0 : &t1;
        agents[1] = (g % 2 == 0) ? &t1 : &t0;
        int A = t0.g.A, ply = 0;
        while (!t0.decision->terminal) {                                     /* play.hpp:34-52 */
            tree* mover = agents[t0.decision->player];
            search_move(mover);                                              /* UCTNetworkAgent.hpp:45-58 */
            int best = 0;                                                    /* std::max_element: first maximum (:88-89) */
            for (int a = 1; a < A; ++a)
                if (mover->decision->N[a] > mover->decision->N[best]) best = a;
            advance_decision(&t0, best);                                     /* act: :101 / opponentAct: :106-108 */
            advance_decision(&t1, best);
            if (ply < max_plies) actions[(size_t)g * max_plies + ply] = (int16_t)best;
            ++ply;
        }
        nplies[g] = ply;
        winners[g] = (int8_t)t0.decision->winner;
        node_free(t0.root, A);
        node_free(t1.root, A);
    }
    return 0;
}

/* One game of self-play — selfplay/SelfPlay.hpp:51-192.  Returns number of samples appended, -1 on overflow. */
static int self_play(const orc_config* cfg, orc_rng* rng, orc_stats* st, int cap, int n0,
                     int8_t* boards, int8_t* players, int8_t* sizes, float* dists, float* outcomes) {
    tree t;
    tree_init(&t, cfg, rng, st);
    geom_t g = t.g;
    int A = g.A;
    int nsym = cfg->use_sym ? g.nsym : 1;
    int n = n0, move_count = 0, resign_winner = -1;
    int movers[1024];
    (void)sizes;
    while (!t.decision->terminal) {
        if (n + nsym > cap) { node_free(t.root, A); return -1; }
        {
            int8_t cur[ORC_MAX_HIST * ORC_MAX_CELLS];
            const size_t stride = (size_t)g.hist * g.cells;
            int size = node_state(&t, t.decision, cur);
            for (int s = 0; s < nsym; ++s) {                                 /* :86-96 */
                symmetrize_state(&t, s, size, cur, boards + (size_t)(n + s) * stride);
                players[n + s] = (int8_t)t.decision->player;
                if (sizes) sizes[n + s] = (int8_t)size;
            }
        }
        search_move(&t);

        float pdf[ORC_MAX_A], cdf[ORC_MAX_A];                                /* :111-125 */
        float sum = 0.0f;
        for (int a = 0; a < A; ++a) sum += t.decision->N[a];
        float inv = 1.0f / sum;
        for (int a = 0; a < A; ++a) pdf[a] = t.decision->N[a] * inv;
        float ex = move_count < cfg->early_cutoff ? cfg->early_exp : cfg->rest_exp;
        for (int a = 0; a < A; ++a) pdf[a] = m_powf(pdf[a], ex, cfg->math_mode);
        sum = 0.0f;
        for (int a = 0; a < A; ++a) sum += pdf[a];
        inv = 1.0f / sum;
        for (int a = 0; a < A; ++a) pdf[a] = pdf[a] * inv;
        cdf[0] = pdf[0];
        for (int a = 1; a < A; ++a) cdf[a] = cdf[a - 1] + pdf[a];
        inv = 1.0f / cdf[A - 1];
        for (int a = 0; a < A; ++a) cdf[a] = cdf[a] * inv;

        for (int s = 0; s < nsym; ++s) {                                     /* :127-137 */
            float* out = dists + (size_t)(n + s) * A;
            memset(out, 0, (size_t)A * sizeof(float));
            orc_symmetrize_dist(cfg->game, s, pdf, out);
        }
        if (cfg->resign_threshold > 0.0f && move_count >= cfg->resign_min_ply) {
            /* extension (BASELINE config 5), off in every parity configuration: mean backed-up value of the
               decision node from the mover's side, sums in index order; the sample of this ply is kept, no move is made */
            float sn = 0.0f, sw = 0.0f;
            for (int a = 0; a < A; ++a) sn += t.decision->N[a];
            for (int a = 0; a < A; ++a) sw += t.decision->W[a];
            float v = sw * (1.0f / sn);
            if (v < -cfg->resign_threshold) {
                movers[move_count] = t.decision->player;
                resign_winner = 1 - t.decision->player;
                ++move_count;
                n += nsym;
                break;
            }
        }
        int action = orc_sample_cdf(rng, cdf, A);                            /* :140 */
        movers[move_count] = t.decision->player;
        advance_decision(&t, action);
        ++move_count;
        n += nsym;
    }
    float rw[2];
    rewards_of(resign_winner >= 0 ? resign_winner : t.decision->winner, rw); /* :151-189 */
    for (int p = 0; p < move_count; ++p)
        for (int s = 0; s < nsym; ++s) outcomes[n0 + p * nsym + s] = rw[movers[p]];
    if (st) {
        st->games++;
        st->plies += move_count;
    }
    node_free(t.root, A);
    return n - n0;
}

int orc_selfplay(const orc_config* cfg, int num_games, uint64_t seed, int stream_base, int per_game_stream,
                 int cap, int8_t* boards, int8_t* players, int8_t* sizes, float* dists, float* outcomes,
                 int32_t* game_offsets, orc_stats* stats) {                  /* SelfPlay.hpp:204-248 */
    orc_rng rng;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (!per_game_stream) orc_rng_seed(&rng, seed, stream_base);
    int n = 0;
    for (int gi = 0; gi < num_games; ++gi) {
        if (per_game_stream) orc_rng_seed(&rng, seed, stream_base + gi);
        game_offsets[gi] = n;
        int k = self_play(cfg, &rng, stats, cap, n, boards, players, sizes, dists, outcomes);
        if (k < 0) return -1;
        n += k;
    }
    game_offsets[num_games] = n;
    return n;
}

/* ------------------------------------------------------------------------------------------ */
/* records — selfplay/GridWorker.hpp:146-196, utils/npy.hpp:430-476                            */
/* ------------------------------------------------------------------------------------------ */

int orc_write_npy_f32(const char* path, const float* data, int ndim, const uint64_t* shape) {
    char dict[256], tuple[128];
    size_t count = 1;
    int p = 0;
    if (ndim == 0) {
        p += snprintf(tuple + p, sizeof(tuple) - (size_t)p, "()");
    } else if (ndim == 1) {
        p += snprintf(tuple + p, sizeof(tuple) - (size_t)p, "(%llu,)", (unsigned long long)shape[0]);
    } else {
        p += snprintf(tuple + p, sizeof(tuple) - (size_t)p, "(");
        for (int i = 0; i < ndim - 1; ++i)
            p += snprintf(tuple + p, sizeof(tuple) - (size_t)p, "%llu, ", (unsigned long long)shape[i]);
        p += snprintf(tuple + p, sizeof(tuple) - (size_t)p, "%llu)", (unsigned long long)shape[ndim - 1]);
    }
    for (int i = 0; i < ndim; ++i) count *= (size_t)shape[i];
    int dl = snprintf(dict, sizeof(dict), "{'descr': '<f4', 'fortran_order': False, 'shape': %s, }", tuple);
    size_t length = 6 + 2 + 2 + (size_t)dl + 1;
    size_t pad = 16 - length % 16;
    FILE* f = fopen(path, "wb");
    if (!f) return -1;
    static const unsigned char magic[8] = { 0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0 };
    fwrite(magic, 1, 8, f);
    uint16_t hl = (uint16_t)((size_t)dl + pad + 1);
    unsigned char le[2] = { (unsigned char)(hl & 0xff), (unsigned char)(hl >> 8) };
    fwrite(le, 1, 2, f);
    fwrite(dict, 1, (size_t)dl, f);
    for (size_t i = 0; i < pad; ++i) fputc(' ', f);
    fputc('\n', f);
    fwrite(data, sizeof(float), count, f);
    fclose(f);
    return 0;
}

int orc_write_records(const orc_config* cfg, const char* path_prefix, int n, const int8_t* boards,
                      const int8_t* players, const int8_t* sizes, const float* dists, const float* outcomes) {
    geom_t g = geom(cfg->game);
    int P = 2 * g.hist + 1;
    float* planes = (float*)malloc((size_t)n * P * g.cells * sizeof(float));
    orc_encode_planes(cfg->game, n, boards, players, sizes, planes);         /* same layout: GridWorker.hpp:146-171 */
    char path[4096];
    uint64_t s4[4] = { (uint64_t)n, (uint64_t)P, (uint64_t)g.rows, (uint64_t)g.cols };
    uint64_t s2[2] = { (uint64_t)n, (uint64_t)g.A };
    uint64_t s1[1] = { (uint64_t)n };
    int rc = 0;
    snprintf(path, sizeof(path), "%s_states.npy", path_prefix);
    rc |= orc_write_npy_f32(path, planes, 4, s4);
    snprintf(path, sizeof(path), "%s_distributions.npy", path_prefix);
    rc |= orc_write_npy_f32(path, dists, 2, s2);
    snprintf(path, sizeof(path), "%s_outcomes.npy", path_prefix);
    rc |= orc_write_npy_f32(path, outcomes, 1, s1);
    free(planes);
    return rc;
}
