// games_wide.h — Go on boards wider than one wavefront (9x9 = 81 points, 19x19 = 361 points), host + device.
//
// The reference compiles Go as 7x7 only (games/GoNode.hpp:16; `Coord` is int8_t); BASELINE configs 4-5
// ask for 9x9 and 19x19, which are parameter extrapolations of the same rules (games/GoNode.cpp).  GoN<WIDTH> keeps
// the rules of Go7 (games.h) on W-word bit sets; an action / point index a lives in strip a / 64, lane a % 64.
// GoN<7> exists so the wide kernel can be checked against the reference-pinned 7x7 oracle as well.
#ifndef SPRL_GAMES_WIDE_H
#define SPRL_GAMES_WIDE_H

#include "bits.h"
#include "games.h"

enum { SPRL_GAME_GO9 = 3, SPRL_GAME_GO19 = 4, SPRL_GAME_GO7W = 5 };

template <int W>
struct PosW {
    Bits<W> p0, p1, legal;
    uint8_t player, pass_legal, terminal;
    int8_t winner;
    uint8_t last_pass;
    uint16_t depth;
};

template <int WIDTH>
struct GoN {
    static constexpr int ID = WIDTH == 9 ? SPRL_GAME_GO9 : (WIDTH == 19 ? SPRL_GAME_GO19 : SPRL_GAME_GO7W);
    static constexpr int ROWS = WIDTH, COLS = WIDTH, CELLS = WIDTH * WIDTH;
    static constexpr int A = CELLS + 1, NA = CELLS;
    static constexpr int WORDS = (CELLS + 63) / 64, STRIPS = (CELLS + 63) / 64;
    static constexpr int HAS_PASS = 1, PASS_EXCLUSIVE = 0;
    static constexpr int HIST = 8, PLANES = 17, NSYM = 8;
    static constexpr int GAME_MAX_DEPTH = 2 * CELLS;             // GoNode.hpp:22
    static constexpr int HIST_CAP = GAME_MAX_DEPTH + 6;
    static constexpr int MAX_DEPTH = HIST_CAP;
    // komi: 9.0 at 7x7 (GoNode.hpp:20); 7.5 for 9x9 and larger (games/GoDesc.md:127-128)
    static constexpr float KOMI = WIDTH == 7 ? 9.0f : 7.5f;
    using BB = Bits<WORDS>;

    // node layout: rows N, W, P (f32 x 64*S), child (u16 x 64*S), header
    static constexpr int ROW_BYTES = STRIPS * 256;
    static constexpr int OFF_W = ROW_BYTES, OFF_P = 2 * ROW_BYTES, OFF_C = 3 * ROW_BYTES, OFF_H = 3 * ROW_BYTES + STRIPS * 128;
    static constexpr int HDR_BYTES = ((3 * WORDS * 8 + 40) + 15) / 16 * 16;
    static constexpr int NODE_BYTES = (OFF_H + HDR_BYTES + 255) / 256 * 256;

    SPRL_B static BB board_mask() {
        BB b = BB::zero();
        for (int i = 0; i < CELLS; ++i) b.w[i >> 6] |= 1ull << (i & 63);
        return b;
    }
    SPRL_B static BB col_mask(int c) {
        BB b = BB::zero();
        for (int r = 0; r < ROWS; ++r) { int i = r * COLS + c; b.w[i >> 6] |= 1ull << (i & 63); }
        return b;
    }
    SPRL_B static BB dilate(const BB& x) {                       // the 4-neighbourhood (GoNode.hpp:117-130)
        const BB not_last = ~col_mask(COLS - 1), not_first = ~col_mask(0);
        return (x.shl(COLS) | x.shr(COLS) | (x & not_last).shl(1) | (x & not_first).shr(1)) & board_mask();
    }
    SPRL_B static BB flood(const BB& seed, const BB& within) {
        BB g = seed & within;
        for (;;) {
            BB n = (g | dilate(g)) & within;
            if (n == g) return g;
            g = n;
        }
    }
    SPRL_B static void start(PosW<WORDS>& s) {                   // GoNode.cpp:303-317
        s.p0 = s.p1 = BB::zero();
        s.player = 0;
        s.legal = board_mask();
        s.pass_legal = 1;
        s.terminal = 0;
        s.winner = -1;
        s.last_pass = 0;
        s.depth = 0;
    }
    // placement + captures (GoNode.cpp:96-176), ending conditions (:359-360) and Tromp-Taylor score (:230-290,367-379)
    SPRL_B static void apply(const PosW<WORDS>& p, int action, PosW<WORDS>& c) {
        BB own = p.player ? p.p1 : p.p0, opp = p.player ? p.p0 : p.p1;
        if (action != CELLS) {
            const BB mv = BB::bit(action);
            own = own | mv;
            BB adj = dilate(mv) & opp;
            while (adj.any()) {
                BB g = flood(BB::bit(adj.lowest()), opp);
                if (!(dilate(g) & ~(own | opp) & board_mask()).any()) opp = opp & ~g;
                adj = adj & ~g;
            }
        }
        c.p0 = p.player ? opp : own;
        c.p1 = p.player ? own : opp;
        c.player = 1 - p.player;
        c.depth = (uint16_t)(p.depth + 1);
        c.last_pass = action == CELLS;
        c.terminal = (p.last_pass && action == CELLS) || c.depth >= GAME_MAX_DEPTH;
        c.winner = -1;
        c.legal = BB::zero();
        c.pass_legal = c.terminal ? 0 : 1;
        if (c.terminal) {
            const BB empty = ~(c.p0 | c.p1) & board_mask();
            const BB e0 = flood(dilate(c.p0) & empty, empty), e1 = flood(dilate(c.p1) & empty, empty);
            float s0 = (float)(c.p0.popc() + (e0 & ~e1).popc());
            float s1 = (float)(c.p1.popc() + (e1 & ~e0).popc());
            s1 += KOMI;
            if ((double)s0 > (double)s1 + 0.1) c.winner = 0;
            else if ((double)s1 > (double)s0 + 0.1) c.winner = 1;
        }
    }
    SPRL_B static int map_cell(int sym, int cell) {              // D4GridSymmetrizer.hpp:108-117
        const int L = WIDTH - 1;
        int r = cell / WIDTH, c = cell % WIDTH, tr, tc;
        switch (sym) {
        case 0: tr = r; tc = c; break;
        case 1: tr = c; tc = L - r; break;
        case 2: tr = L - r; tc = L - c; break;
        case 3: tr = L - c; tc = r; break;
        case 4: tr = r; tc = L - c; break;
        case 5: tr = L - c; tc = L - r; break;
        case 6: tr = L - r; tc = c; break;
        default: tr = c; tc = r; break;
        }
        return tr * WIDTH + tc;
    }
    SPRL_B static int map_action(int sym, int a) { return a == CELLS ? CELLS : map_cell(sym, a); }
    SPRL_B static int inverse_sym(int sym) { return sym == 1 ? 3 : (sym == 3 ? 1 : sym); }
};

#endif  // SPRL_GAMES_WIDE_H
