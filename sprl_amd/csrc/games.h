// games.h — bitboard rules for the games on the self-play path (host + gfx950 device).
//
// Behaviour follows the reference's array-based rules (file:line relative to /root/reference/cpp/src):
//   Othello       games/OthelloNode.cpp:18-32 (start), :34-87 (move), :156-177 (mask), :179-191 (terminal)
//   Connect Four  games/ConnectFourNode.cpp:13-21 (start), :23-78 (move), :135-217 (win test)
// but the representation is MI355X-first: one uint64 per colour, all lanes of the wavefront that owns
// the tree compute the same scalar bit-parallel update (SALU work), no per-cell loops, no masks in
// memory.  Cell index = row * COLS + col = bit index; player ZERO owns `p0`.
#ifndef SPRL_GAMES_H
#define SPRL_GAMES_H

#include <stdint.h>

#if defined(__HIPCC__) && !defined(SPRL_EMU)
#define SPRL_G __host__ __device__ static inline
#else
#define SPRL_G static inline
#endif

enum { SPRL_GAME_OTHELLO = 0, SPRL_GAME_CONNECT_FOUR = 1, SPRL_GAME_GO7 = 2 };

struct Pos {
    uint64_t p0, p1;     // stones of Player::ZERO / Player::ONE
    uint64_t legal;      // legal lane-actions for the side to move (pass excluded)
    uint8_t player;      // side to move
    uint8_t pass_legal;  // Othello: pass is legal iff nothing else is (OthelloNode.cpp:166-174)
    uint8_t terminal;
    int8_t winner;       // -1 none/draw, 0, 1
    uint8_t last_pass;   // Go: the action that led here was a pass (GoNode.cpp:359)
    uint16_t depth;      // Go: plies since the start of the game (GoNode::m_depth)
};

// ---------------------------------------------------------------------------------------------
// Othello 8x8
// ---------------------------------------------------------------------------------------------
struct Othello {
    static constexpr int ID = SPRL_GAME_OTHELLO;
    static constexpr int ROWS = 8, COLS = 8, CELLS = 64;
    static constexpr int A = 65;          // 64 placements + pass (OthelloNode.hpp:10)
    static constexpr int NA = 64;         // lane-mapped actions
    static constexpr int HAS_PASS = 1;
    static constexpr int PASS_EXCLUSIVE = 1;  // pass is legal only when nothing else is
    static constexpr int HIST = 1, PLANES = 3, HIST_CAP = 1;
    static constexpr int NSYM = 8;        // D4 (symmetry/D4GridSymmetrizer.hpp:30-41)
    static constexpr int MAX_DEPTH = 128; // >= longest possible line (60 placements + interleaved passes)

    static constexpr uint64_t NOT_A = 0xfefefefefefefefeull;  // col != 0
    static constexpr uint64_t NOT_H = 0x7f7f7f7f7f7f7f7full;  // col != 7

    // one step in each of the 8 directions; d: 0 S,1 SE,2 E,3 NE,4 N,5 NW,6 W,7 SW (row+ = south)
    template <int D> SPRL_G uint64_t shift(uint64_t x) {
        if (D == 0) return x << 8;
        if (D == 1) return (x << 9) & NOT_A;
        if (D == 2) return (x << 1) & NOT_A;
        if (D == 3) return (x >> 7) & NOT_A;
        if (D == 4) return x >> 8;
        if (D == 5) return (x >> 9) & NOT_H;
        if (D == 6) return (x >> 1) & NOT_H;
        return (x << 7) & NOT_H;
    }
    template <int D> SPRL_G uint64_t legal_dir(uint64_t own, uint64_t opp, uint64_t empty) {
        uint64_t t = shift<D>(own) & opp;
        t |= shift<D>(t) & opp;
        t |= shift<D>(t) & opp;
        t |= shift<D>(t) & opp;
        t |= shift<D>(t) & opp;
        t |= shift<D>(t) & opp;
        return shift<D>(t) & empty;
    }
    // canCapture over every empty square at once (OthelloNode.cpp:226-252)
    SPRL_G uint64_t legal_moves(uint64_t own, uint64_t opp) {
        uint64_t e = ~(own | opp);
        return legal_dir<0>(own, opp, e) | legal_dir<1>(own, opp, e) | legal_dir<2>(own, opp, e) |
               legal_dir<3>(own, opp, e) | legal_dir<4>(own, opp, e) | legal_dir<5>(own, opp, e) |
               legal_dir<6>(own, opp, e) | legal_dir<7>(own, opp, e);
    }
    template <int D> SPRL_G uint64_t flips_dir(uint64_t own, uint64_t opp, uint64_t mv) {
        uint64_t t = shift<D>(mv) & opp;
        t |= shift<D>(t) & opp;
        t |= shift<D>(t) & opp;
        t |= shift<D>(t) & opp;
        t |= shift<D>(t) & opp;
        t |= shift<D>(t) & opp;
        return (shift<D>(t) & own) ? t : 0ull;
    }
    // captures() (OthelloNode.cpp:193-224)
    SPRL_G uint64_t flips(uint64_t own, uint64_t opp, uint64_t mv) {
        return flips_dir<0>(own, opp, mv) | flips_dir<1>(own, opp, mv) | flips_dir<2>(own, opp, mv) |
               flips_dir<3>(own, opp, mv) | flips_dir<4>(own, opp, mv) | flips_dir<5>(own, opp, mv) |
               flips_dir<6>(own, opp, mv) | flips_dir<7>(own, opp, mv);
    }
    SPRL_G void finish(Pos& c) {
        uint64_t own = c.player ? c.p1 : c.p0, opp = c.player ? c.p0 : c.p1;
        c.legal = legal_moves(own, opp);
        c.pass_legal = c.legal == 0;
        // terminal iff both sides can only pass (OthelloNode.cpp:179-191)
        c.terminal = c.pass_legal && legal_moves(opp, own) == 0;
        c.winner = -1;
        if (c.terminal) {
            int c0 = __builtin_popcountll(c.p0), c1 = __builtin_popcountll(c.p1);
            if (c0 > c1) c.winner = 0;
            if (c1 > c0) c.winner = 1;
        }
    }
    SPRL_G void start(Pos& s) {
        s.p0 = (1ull << (3 * 8 + 4)) | (1ull << (4 * 8 + 3));   // OthelloNode.cpp:26-29
        s.p1 = (1ull << (3 * 8 + 3)) | (1ull << (4 * 8 + 4));
        s.player = 0;
        s.last_pass = 0;
        s.depth = 0;
        finish(s);
    }
    SPRL_G void child(const Pos& p, int action, Pos& c) {
        uint64_t own = p.player ? p.p1 : p.p0, opp = p.player ? p.p0 : p.p1;
        if (action != 64) {
            uint64_t mv = 1ull << action;
            uint64_t f = flips(own, opp, mv);
            own |= mv | f;
            opp &= ~f;
        }
        c.p0 = p.player ? opp : own;
        c.p1 = p.player ? own : opp;
        c.player = 1 - p.player;
        c.last_pass = 0;
        c.depth = 0;
        finish(c);
    }
    // D4 cell maps, out[map(r,c)] = in[r,c] (D4GridSymmetrizer.hpp:108-117)
    SPRL_G int map_cell(int sym, int cell) {
        int r = cell >> 3, c = cell & 7, tr, tc;
        switch (sym) {
        case 0: tr = r; tc = c; break;
        case 1: tr = c; tc = 7 - r; break;
        case 2: tr = 7 - r; tc = 7 - c; break;
        case 3: tr = 7 - c; tc = r; break;
        case 4: tr = r; tc = 7 - c; break;
        case 5: tr = 7 - c; tc = 7 - r; break;
        case 6: tr = 7 - r; tc = c; break;
        default: tr = c; tc = r; break;
        }
        return tr * 8 + tc;
    }
    SPRL_G int map_action(int sym, int a) { return a == 64 ? 64 : map_cell(sym, a); }
    SPRL_G int inverse_sym(int sym) { return sym == 1 ? 3 : (sym == 3 ? 1 : sym); }  // :43-46
};

// ---------------------------------------------------------------------------------------------
// Connect Four 6x7 (row 0 = top)
// ---------------------------------------------------------------------------------------------
struct ConnectFour {
    static constexpr int ID = SPRL_GAME_CONNECT_FOUR;
    static constexpr int ROWS = 6, COLS = 7, CELLS = 42;
    static constexpr int A = 7;
    static constexpr int NA = 7;
    static constexpr int HAS_PASS = 0;
    static constexpr int PASS_EXCLUSIVE = 0;
    static constexpr int HIST = 1, PLANES = 3, HIST_CAP = 1;
    static constexpr int NSYM = 2;        // identity + column mirror (ConnectFourSymmetrizer.cpp:5-13)
    static constexpr int MAX_DEPTH = 64;

    SPRL_G uint64_t top_free(uint64_t occ) { return ~occ & 0x7full; }   // row 0 empty -> column playable
    SPRL_G void start(Pos& s) {
        s.p0 = s.p1 = 0;
        s.player = 0;
        s.legal = 0x7f;
        s.pass_legal = 0;
        s.terminal = 0;
        s.winner = -1;
        s.last_pass = 0;
        s.depth = 0;
    }
    SPRL_G int count_dir(uint64_t mine, int r, int c, int dr, int dc) {
        int n = 0;
        r += dr;
        c += dc;
        while (r >= 0 && r < 6 && c >= 0 && c < 7 && ((mine >> (r * 7 + c)) & 1)) {
            ++n;
            r += dr;
            c += dc;
        }
        return n;
    }
    SPRL_G void child(const Pos& p, int action, Pos& c) {
        uint64_t occ = p.p0 | p.p1;
        int col = action, row = 5;
        while (row >= 0 && ((occ >> (row * 7 + col)) & 1)) --row;        // ConnectFourNode.cpp:40-43
        uint64_t mv = 1ull << (row * 7 + col);
        uint64_t mine = (p.player ? p.p1 : p.p0) | mv;
        c.p0 = p.player ? p.p0 : mine;
        c.p1 = p.player ? mine : p.p1;
        c.player = 1 - p.player;
        c.pass_legal = 0;
        c.last_pass = 0;
        c.depth = 0;
        bool win = 1 + count_dir(mine, row, col, 0, -1) + count_dir(mine, row, col, 0, 1) >= 4 ||
                   1 + count_dir(mine, row, col, -1, 0) + count_dir(mine, row, col, 1, 0) >= 4 ||
                   1 + count_dir(mine, row, col, -1, -1) + count_dir(mine, row, col, 1, 1) >= 4 ||
                   1 + count_dir(mine, row, col, -1, 1) + count_dir(mine, row, col, 1, -1) >= 4;
        c.winner = win ? (int8_t)p.player : (int8_t)-1;
        uint64_t free_cols = top_free(c.p0 | c.p1);
        c.terminal = win || free_cols == 0;                              // :57-66
        // the mask is inherited, a column is cleared when it fills, and zeroed on terminal (:50-52,:69-71)
        c.legal = c.terminal ? 0 : free_cols;
    }
    SPRL_G int map_cell(int sym, int cell) {
        int r = cell / 7, c = cell % 7;
        return sym == 1 ? r * 7 + (6 - c) : cell;
    }
    SPRL_G int map_action(int sym, int a) { return sym == 1 ? 6 - a : a; }
    SPRL_G int inverse_sym(int sym) { return sym; }
};

// ---------------------------------------------------------------------------------------------
// Go 7x7 (the size the reference compiles: games/GoNode.hpp:16-22), komi 9.0, positional superko, depth cap 98
// ---------------------------------------------------------------------------------------------
struct Go7 {
    static constexpr int ID = SPRL_GAME_GO7;
    static constexpr int ROWS = 7, COLS = 7, CELLS = 49;
    static constexpr int A = 50;          // 49 placements + pass (GoNode.hpp:18)
    static constexpr int NA = 49;
    static constexpr int HAS_PASS = 1;
    static constexpr int PASS_EXCLUSIVE = 0;  // pass is always legal and competes in the arg-max (GoNode.cpp:298)
    static constexpr int HIST = 8, PLANES = 17;
    static constexpr int GAME_MAX_DEPTH = 98;      // GO_MAX_DEPTH = 2 * 49 (GoNode.hpp:22)
    static constexpr int HIST_CAP = 104;           // ancestor positions kept per game (>= GAME_MAX_DEPTH + 1)
    static constexpr int NSYM = 8;
    static constexpr int MAX_DEPTH = 104;

    static constexpr uint64_t BOARD = (1ull << 49) - 1;
    static constexpr uint64_t COL0 = 0x0040810204081ull;             // bits r*7 + 0
    static constexpr uint64_t COL6 = COL0 << 6;

    SPRL_G uint64_t dilate(uint64_t x) {            // the 4-neighbourhood (GoNode.hpp:117-130)
        return ((x << 7) | (x >> 7) | ((x & ~COL6) << 1) | ((x & ~COL0) >> 1)) & BOARD;
    }
    SPRL_G uint64_t flood(uint64_t seed, uint64_t within) {
        uint64_t g = seed & within;
        for (;;) {
            uint64_t n = (g | dilate(g)) & within;
            if (n == g) return g;
            g = n;
        }
    }
    SPRL_G void start(Pos& s) {                     // GoNode.cpp:303-317
        s.p0 = s.p1 = 0;
        s.player = 0;
        s.legal = BOARD;
        s.pass_legal = 1;
        s.terminal = 0;
        s.winner = -1;
        s.last_pass = 0;
        s.depth = 0;
    }
    // Board after `action` by p.player: place, then remove every adjacent enemy group left without liberties
    // (placePiece phase two, GoNode.cpp:139-169).  The move is legal by construction, so no suicide handling.
    SPRL_G void apply(const Pos& p, int action, Pos& c) {
        uint64_t own = p.player ? p.p1 : p.p0, opp = p.player ? p.p0 : p.p1;
        if (action != 49) {
            const uint64_t mv = 1ull << action;
            own |= mv;
            uint64_t adj = dilate(mv) & opp;
            while (adj) {
                uint64_t g = flood(adj & (0 - adj), opp);
                if ((dilate(g) & ~(own | opp) & BOARD) == 0) opp &= ~g;
                adj &= ~g;
            }
        }
        c.p0 = p.player ? opp : own;
        c.p1 = p.player ? own : opp;
        c.player = 1 - p.player;
        c.depth = (uint16_t)(p.depth + 1);
        c.last_pass = action == 49;
        c.terminal = (p.last_pass && action == 49) || c.depth >= GAME_MAX_DEPTH;    // GoNode.cpp:359-360
        c.winner = -1;
        c.legal = 0;
        c.pass_legal = c.terminal ? 0 : 1;
        if (c.terminal) {                                                          // Tromp-Taylor, GoNode.cpp:230-290,367-379
            const uint64_t empty = ~(c.p0 | c.p1) & BOARD;
            const uint64_t e0 = flood(dilate(c.p0) & empty, empty), e1 = flood(dilate(c.p1) & empty, empty);
            float s0 = (float)(__builtin_popcountll(c.p0) + __builtin_popcountll(e0 & ~e1));
            float s1 = (float)(__builtin_popcountll(c.p1) + __builtin_popcountll(e1 & ~e0));
            s1 += 9.0f;                                                            // GO_KOMI
            if ((double)s0 > (double)s1 + 0.1) c.winner = 0;
            else if ((double)s1 > (double)s0 + 0.1) c.winner = 1;
        }
    }
    SPRL_G int map_cell(int sym, int cell) {
        int r = cell / 7, c = cell % 7, tr, tc;
        switch (sym) {
        case 0: tr = r; tc = c; break;
        case 1: tr = c; tc = 6 - r; break;
        case 2: tr = 6 - r; tc = 6 - c; break;
        case 3: tr = 6 - c; tc = r; break;
        case 4: tr = r; tc = 6 - c; break;
        case 5: tr = 6 - c; tc = 6 - r; break;
        case 6: tr = 6 - r; tc = c; break;
        default: tr = c; tc = r; break;
        }
        return tr * 7 + tc;
    }
    SPRL_G int map_action(int sym, int a) { return a == 49 ? 49 : map_cell(sym, a); }
    SPRL_G int inverse_sym(int sym) { return sym == 1 ? 3 : (sym == 3 ? 1 : sym); }
};

#endif  // SPRL_GAMES_H
