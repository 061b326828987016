// step_kernel_wide.h — the self-play step for boards wider than one wavefront (Go 9x9, 19x19).
//
// Same algorithm, same reference semantics and the same one-wavefront-per-game mapping as step_kernel.h, with rows
// of S = ceil(actions / 64) strips: point / action a lives in strip a / 64, lane a % 64, every row operation loops
// over the strips, and cross-point reads that can leave the strip (a point's neighbours, the inverse symmetry of a
// policy) go through a small LDS exchange instead of a single ds_bpermute.  The single-strip kernel stays the tuned
// path for the benchmark game; this one trades some speed for generality and is checked against the same oracle
// (including at 7x7, where the oracle is pinned to the reference).
#ifndef SPRL_STEP_KERNEL_WIDE_H
#define SPRL_STEP_KERNEL_WIDE_H

#include "dev_rng.h"
#include "engine_types.h"
#include "games_wide.h"
#include "wave.h"

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace sprlw {

template <int W>
struct NodeHdrW {
    Bits<W> p0, p1, legal;
    float value;
    uint32_t exp_epoch;
    float passN, passW, passP;
    uint32_t passChild;
    uint8_t player, flags;
    int8_t winner;
    uint8_t pad0;
    uint16_t action, depth;
};

// Per-wave LDS.  The positions of the line (real game + current descent) are NOT here: they live in HBM
// (EngineParams::hist_boards, one ring per slot) and only their 64-bit signatures sit in LDS.  At 19x19 the position history
// alone was 70 KB per wave (and the per-group stone masks another 18 KB), which left room for ONE wave per CU; now a 19x19 wave
// needs 23 KB and six of them share a CU.
template <class G>
struct WaveLdsW {
    uint32_t path[G::MAX_DEPTH];
    uint64_t hsig[G::HIST_CAP];               // pos_sig of position i of the line: the superko test compares signatures, and
                                              // whole positions (from the ring in HBM) only where a signature matches
    Bits<G::WORDS> leaf_hist[SPRL_MAXQ][G::HIST][2];
    uint32_t leaf_size[SPRL_MAXQ];
    float xf[G::STRIPS * 64];                 // cross-strip exchange: one float / u32 per point
    uint32_t xu[G::STRIPS * 64];
    uint32_t xl[G::STRIPS * 64];              // liberties per group label
    uint32_t xs[G::STRIPS * 64][2];           // label form: signature (lo, hi) of the stones of each opponent group in atari
    // Bloom filter (two hashes, <= 5 % of the bits set) over the signatures of the REAL game's positions [0, ply]: they do not
    // change during a search, and almost no candidate move repeats one of them - a candidate walks the signature list of the
    // real game only on a filter hit and always the few positions of the current descent.  Round 2 scanned all of them for every
    // empty point of every created node (ply + depth reads per point: the largest part of a late-game 9x9 round).
    static constexpr int BLOOM_BITS = G::HIST_CAP <= 128 ? 4096 : (G::HIST_CAP <= 256 ? 8192 : 32768);
    uint32_t bloom[BLOOM_BITS / 32];
    Bits<G::WORDS> xg[G::STRIPS <= 2 ? G::STRIPS * 64 : 1];    // flood form (boards up to two strips): stone mask per group
    uint32_t fcache[SPRL_FCACHE];             // recycled node ids ready for reuse (GameCtl::fcache while the slot runs)
};

// 64-bit signature of a position (xor of rotated words): equal positions have equal signatures, so the positional-superko test
// only has to compare whole bit sets (2 x WORDS words) with the ancestors whose signature matches the candidate's - exact as
// before.  The signature is xor-linear in the stones (every stone contributes one bit, stone_sig), so the signature of "this
// position + a stone at a - the captured groups" follows from 64-bit group signatures without building the position - what the
// reference does with Zobrist hashes (games/GoNode.cpp:186-227), except that a match is then verified on the whole position.
template <int W>
SPRL_DEV uint64_t pos_sig(const Bits<W>& a, const Bits<W>& b) {
    uint64_t s = 0;
    for (int w = 0; w < W; ++w) {
        const int r0 = (11 * w + 1) & 63, r1 = (11 * w + 33) & 63;
        s ^= (a.w[w] << r0) | (a.w[w] >> (64 - r0));
        s ^= (b.w[w] << r1) | (b.w[w] >> (64 - r1));
    }
    return s;
}
// the two Bloom-filter bit positions of a signature (multiplicative hashes: a signature is a folded bit board, positions that
// differ in one stone differ in one bit of it, so bit slices of the signature itself would collide for exactly the related
// positions a game consists of)
template <class G>
SPRL_DEV void bloom_bits(uint64_t sg, uint32_t& b0, uint32_t& b1) {
    constexpr int LOG = WaveLdsW<G>::BLOOM_BITS == 4096 ? 12 : (WaveLdsW<G>::BLOOM_BITS == 8192 ? 13 : 15);
    b0 = (uint32_t)((sg * 0x9E3779B97F4A7C15ull) >> (64 - LOG));
    b1 = (uint32_t)((sg * 0xC2B2AE3D27D4EB4Full) >> (64 - LOG));
}
template <class G>
SPRL_DEV void bloom_add(WaveLdsW<G>* lds, uint64_t sg) {             // any number of lanes, each with its own signature
    uint32_t b0, b1;
    bloom_bits<G>(sg, b0, b1);
    wv::atomic_or_u32(&lds->bloom[b0 >> 5], 1u << (b0 & 31));
    wv::atomic_or_u32(&lds->bloom[b1 >> 5], 1u << (b1 & 31));
}
template <class G>
SPRL_DEV bool bloom_hit(const WaveLdsW<G>* lds, uint64_t sg) {
    uint32_t b0, b1;
    bloom_bits<G>(sg, b0, b1);
    return ((lds->bloom[b0 >> 5] >> (b0 & 31)) & (lds->bloom[b1 >> 5] >> (b1 & 31)) & 1u) != 0;
}
template <class G>
SPRL_DEV void bloom_clear(WaveLdsW<G>* lds) {                         // all lanes
    for (int i = wv::lane(); i < WaveLdsW<G>::BLOOM_BITS / 32; i += 64) lds->bloom[i] = 0u;
    wv::sync();
}

// contribution of one stone at point a of Player `plane` to pos_sig
SPRL_DEV uint64_t stone_sig(int a, int plane) {
    const int w = a >> 6, r = (11 * w + (plane ? 33 : 1)) & 63;
    return 1ull << (((a & 63) + r) & 63);
}
// position `at` of the line: the bit sets go to the slot's ring in HBM, the signature to LDS (one lane calls this)
template <class G>
SPRL_DEV void hist_put(WaveLdsW<G>* lds, Bits<G::WORDS>* ring, int at, const Bits<G::WORDS>& p0, const Bits<G::WORDS>& p1) {
    ring[2 * at] = p0;
    ring[2 * at + 1] = p1;
    lds->hsig[at] = pos_sig<G::WORDS>(p0, p1);
}

template <class G> SPRL_DEV uint8_t* node_at(uint8_t* abase, uint32_t idx) { return abase + (size_t)idx * G::NODE_BYTES; }
template <class G> SPRL_DEV float* rowN(uint8_t* n) { return (float*)n; }
template <class G> SPRL_DEV float* rowW(uint8_t* n) { return (float*)(n + G::OFF_W); }
template <class G> SPRL_DEV float* rowP(uint8_t* n) { return (float*)(n + G::OFF_P); }
template <class G> SPRL_DEV uint16_t* rowC(uint8_t* n) { return (uint16_t*)(n + G::OFF_C); }
template <class G> SPRL_DEV NodeHdrW<G::WORDS>* hdr_of(uint8_t* n) { return (NodeHdrW<G::WORDS>*)(n + G::OFF_H); }

#define WPASS (G::A - 1)
#define WS (G::STRIPS)

struct GameW {
    Pcg32 rng;
    uint8_t* abase;
    void* ring;                  // Bits<WORDS>[HIST_CAP][2]: positions of the line (this slot's part of EngineParams::hist_boards)
    uint32_t arena, root, n_alloc, epoch, root_player, game_id, status;
    uint32_t rstack_n, fc_n;     // node recycling: reclaim stack height, ready recycled ids (as step_kernel.h)
    float rootN, rootW;
    int ply, traversals, n_leaves;
    int legal_form;         // EngineParams::go_legal_form
    uint32_t d_traversals, d_levels, d_expansions, d_nn_evals, d_terminal, d_gray, d_dup, d_created, d_compactions,
        d_games, d_plies, d_recycled, hi_alloc;
};

SPRL_DEV void raise_error(const EngineParams& P, GameW& g, uint32_t code) {
    if (wv::lane() == 0) {
        if (wv::atomic_cas_u32(&P.counters->error, 0u, code) == 0u) P.counters->error_game = g.game_id;
    }
    g.status = ST_ERROR;
}

// ---- node recycling (see step_kernel.h): the old decision node, its edge to the kept child cut, goes on the game's reclaim
// stack; refills pop ids, push their children (all strips of the child row + the pass child) and hand the ids to the allocator
template <class G>
SPRL_DEV uint32_t alloc_node(GameW& g, WaveLdsW<G>* lds) {
    if (g.fc_n > 0) {
        g.d_recycled++;
        return lds->fcache[--g.fc_n];
    }
    const uint32_t c = g.n_alloc++;
    if (g.n_alloc > g.hi_alloc) g.hi_alloc = g.n_alloc;
    return c;
}

template <class G>
SPRL_DEV void reclaim_refill(const EngineParams& P, GameW& g, int slot, WaveLdsW<G>* lds) {
    const int l = wv::lane();
    uint32_t* stack = P.reclaim + (size_t)slot * (size_t)P.node_cap;
    constexpr int K = WS > 2 ? 2 : 4;         // ids per pass: their child rows (WS strips each) are requested together
    while (g.fc_n < SPRL_FCACHE && g.rstack_n > 0) {
        int k = SPRL_FCACHE - (int)g.fc_n;
        if (k > K) k = K;
        if (k > (int)g.rstack_n) k = (int)g.rstack_n;
        const uint32_t base = g.rstack_n - (uint32_t)k;
        const uint32_t mine = l < k ? stack[base + l] : 0u;
        wv::sync();
        g.rstack_n = base;
        uint32_t id[K], ch[K][WS], pc[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            id[j] = wv::bcast_u32(mine, j < k ? j : 0);
            uint8_t* np = node_at<G>(g.abase, id[j]);
#pragma unroll
            for (int st = 0; st < WS; ++st) ch[j][st] = j < k ? (uint32_t)rowC<G>(np)[st * 64 + l] : (uint32_t)SPRL_NONE16;
            pc[j] = j < k ? hdr_of<G>(np)->passChild : (uint32_t)SPRL_NONE16;
        }
        wv::sync();
#pragma unroll
        for (int j = 0; j < K; ++j) {
            if (j >= k) break;
#pragma unroll
            for (int st = 0; st < WS; ++st) {
                const bool has = ch[j][st] != SPRL_NONE16;
                const uint64_t mask = wv::ballot(has);
                if (has) stack[g.rstack_n + (uint32_t)wv::popc64(mask & wv::lt_mask(l))] = ch[j][st];
                g.rstack_n += (uint32_t)wv::popc64(mask);
            }
            if (pc[j] != SPRL_NONE16) {
                if (l == 0) stack[g.rstack_n] = pc[j];
                g.rstack_n += 1;
            }
            if (l == 0) lds->fcache[g.fc_n] = id[j];
            g.fc_n += 1;
        }
        wv::wave_fence();
    }
}

// A node header is wave-uniform, but it comes from a vector load (the arenas are written by this kernel, so the compiler may
// not use the scalar cache) and would live in vector registers with every lane repeating the same bit-set arithmetic.  On the
// boards of up to two words the fields are moved to scalar registers (v_readfirstlane): move generation then runs on the scalar
// ALU and the kernel needs half the vector registers.  (At six words the three bit sets alone are 36 scalar registers per
// header and several headers / positions are live at once: they stay where they are.)
template <class G>
SPRL_DEV NodeHdrW<G::WORDS> load_hdr(uint8_t* n) {
    NodeHdrW<G::WORDS> h = *hdr_of<G>(n);
    wv::sync();
    if constexpr (G::WORDS <= 2) {
        for (int w = 0; w < G::WORDS; ++w) {
            h.p0.w[w] = wv::uni(h.p0.w[w]);
            h.p1.w[w] = wv::uni(h.p1.w[w]);
            h.legal.w[w] = wv::uni(h.legal.w[w]);
        }
        h.value = wv::uni(h.value);
        h.exp_epoch = wv::uni(h.exp_epoch);
        h.passN = wv::uni(h.passN);
        h.passW = wv::uni(h.passW);
        h.passP = wv::uni(h.passP);
        h.passChild = wv::uni(h.passChild);
        h.player = (uint8_t)wv::uni((uint32_t)h.player);
        h.flags = (uint8_t)wv::uni((uint32_t)h.flags);
        h.winner = (int8_t)wv::uni((int)h.winner);
        h.action = (uint16_t)wv::uni((uint32_t)h.action);
        h.depth = (uint16_t)wv::uni((uint32_t)h.depth);
    }
    return h;
}

template <class G>
SPRL_DEV void write_new_node(uint8_t* np, const PosW<G::WORDS>& s, int action) {
    NodeHdrW<G::WORDS>* h = hdr_of<G>(np);
    h->p0 = s.p0;
    h->p1 = s.p1;
    h->legal = s.legal;
    h->value = 0.0f;
    h->exp_epoch = 0;
    h->passN = 0.0f;
    h->passW = 0.0f;
    h->passP = 0.0f;
    h->passChild = SPRL_NONE16;
    h->player = s.player;
    h->flags = (uint8_t)((s.terminal ? F_TERMINAL : 0) | (s.pass_legal ? F_PASS : 0));
    h->winner = s.winner;
    h->action = (uint16_t)action;
    h->depth = s.depth;
    for (int st = 0; st < WS; ++st) rowC<G>(np)[st * 64 + wv::lane()] = SPRL_NONE16;
}

template <class G>
SPRL_DEV PosW<G::WORDS> pos_of(const NodeHdrW<G::WORDS>& h) {
    PosW<G::WORDS> s;
    s.p0 = h.p0;
    s.p1 = h.p1;
    s.legal = h.legal;
    s.player = h.player;
    s.pass_legal = (h.flags & F_PASS) ? 1 : 0;
    s.terminal = (h.flags & F_TERMINAL) ? 1 : 0;
    s.winner = h.winner;
    s.last_pass = h.action == WPASS && h.depth > 0;
    s.depth = h.depth;
    return s;
}

// Legal mask of a Go position (GoNode.cpp:178-228,292-301), point a = strip * 64 + lane: per-point group flood fill,
// liberties, neighbour inspection through the LDS exchange, exact positional-superko compare with the ancestors.
// Small boards (up to two strips): every stone lane flood-fills its own group as a bit set - fewer, wider steps.
template <class G>
SPRL_DEV Bits<G::WORDS> go_legal_mask_flood(const PosW<G::WORDS>& c, WaveLdsW<G>* lds, const Bits<G::WORDS>* ring, int n_hist, int n_game) {
    // positions [0, n_game) of the line are the real game's (Bloom filter), [n_game, n_hist) the current descent's
    using BB = Bits<G::WORDS>;
    const int l = wv::lane();
    const BB own = c.player ? c.p1 : c.p0, opp = c.player ? c.p0 : c.p1;
    const BB board = G::board_mask();
    const BB empty = ~(own | opp) & board;
    BB gm[WS], within[WS];
    for (int st = 0; st < WS; ++st) {
        const int a = st * 64 + l;
        const bool on = a < G::CELLS;
        const bool is_own = on && own.test(a), is_opp = on && opp.test(a);
        within[st] = is_own ? own : (is_opp ? opp : BB::zero());
        gm[st] = on ? (BB::bit(a) & within[st]) : BB::zero();
    }
    for (;;) {
        bool changed = false;
        for (int st = 0; st < WS; ++st) {
            const BB n = (gm[st] | G::dilate(gm[st])) & within[st];
            changed |= n != gm[st];
            gm[st] = n;
        }
        if (wv::ballot(changed) == 0) break;
    }
    wv::sync();
    for (int st = 0; st < WS; ++st) {
        lds->xu[st * 64 + l] = (uint32_t)(G::dilate(gm[st]) & empty).popc();
        lds->xg[st * 64 + l] = gm[st];
    }
    wv::sync();
    BB legal = BB::zero();
    for (int st = 0; st < WS; ++st) {
        const int a = st * 64 + l;
        bool ok = false;
        if (a < G::CELLS && empty.test(a)) {
            const int row = a / G::COLS, col = a % G::COLS;
            bool has_libs = false;
            BB cap = BB::zero();
            for (int d = 0; d < 4; ++d) {
                const bool valid = d == 0 ? row > 0 : d == 1 ? col > 0 : d == 2 ? row < G::ROWS - 1 : col < G::COLS - 1;
                if (!valid) continue;
                const int nb = d == 0 ? a - G::COLS : d == 1 ? a - 1 : d == 2 ? a + G::COLS : a + 1;
                if (empty.test(nb)) has_libs = true;
                else if (own.test(nb)) { if (lds->xu[nb] > 1) has_libs = true; }
                else if (lds->xu[nb] == 1) { has_libs = true; cap = cap | lds->xg[nb]; }
            }
            const BB nown = own | BB::bit(a), nopp = opp & ~cap;
            const BB np0 = c.player ? nopp : nown, np1 = c.player ? nown : nopp;
            bool repeat = false;
            const uint64_t sg = pos_sig<G::WORDS>(np0, np1);
            for (int i = n_game; i < n_hist; ++i)
                if (lds->hsig[i] == sg) repeat |= (ring[2 * i] == np0) && (ring[2 * i + 1] == np1);
            if (has_libs && bloom_hit<G>(lds, sg))
                for (int i = 0; i < n_game; ++i)
                    if (lds->hsig[i] == sg) repeat |= (ring[2 * i] == np0) && (ring[2 * i + 1] == np1);
            ok = has_libs && !repeat;
        }
        const uint64_t m = wv::ballot(ok);
        legal.w[st] = m;
    }
    return legal;
}

// Large boards: per-lane bit sets would need 2 x STRIPS x WORDS registers and one wide dilate per step; groups are
// labelled instead (measured at 19x19: 3.6-4.9x faster than the flood form; at 9x9 the flood form is 25 % faster).
template <class G>
SPRL_DEV Bits<G::WORDS> go_legal_mask(const PosW<G::WORDS>& c, WaveLdsW<G>* lds, const Bits<G::WORDS>* ring, int n_hist, int n_game,
                                      int form) {
    // form: 0 = by board size (flood up to two strips, labels above), 1 = labels, 2 = flood (tests run both forms on the boards
    // of up to two strips; the per-group stone masks of the flood form have no room in LDS on larger boards)
    if constexpr (G::STRIPS <= 2) {
        if (form == 2 || form == 0) return go_legal_mask_flood<G>(c, lds, ring, n_hist, n_game);
    }
    using BB = Bits<G::WORDS>;
    const int l = wv::lane();
    const BB own = c.player ? c.p1 : c.p0, opp = c.player ? c.p0 : c.p1;
    const BB board = G::board_mask();
    const BB empty = ~(own | opp) & board;
    // 1. connected groups: every stone takes the smallest point index of its group as label (min over same-coloured
    //    neighbours + pointer jumping, until no label moves); one u32 per point in LDS, no per-lane bit sets
    uint32_t lab[WS];
    uint8_t colr[WS];                                    // 0 empty / off board, 1 own, 2 opponent
    for (int st = 0; st < WS; ++st) {
        const int a = st * 64 + l;
        const bool on = a < G::CELLS;
        colr[st] = (uint8_t)(on && own.test(a) ? 1 : (on && opp.test(a) ? 2 : 0));
        lab[st] = colr[st] ? (uint32_t)a : 0xFFFFu;
        lds->xl[st * 64 + l] = 0;
        lds->xs[st * 64 + l][0] = 0;
        lds->xs[st * 64 + l][1] = 0;
    }
    for (;;) {
        for (int st = 0; st < WS; ++st) lds->xu[st * 64 + l] = lab[st];
        wv::sync();
        bool changed = false;
        for (int st = 0; st < WS; ++st) {
            if (!colr[st]) continue;
            const int a = st * 64 + l, row = a / G::COLS, col = a % G::COLS;
            const BB& mine = colr[st] == 1 ? own : opp;
            uint32_t m = lab[st];
            if (row > 0 && mine.test(a - G::COLS)) { const uint32_t v = lds->xu[a - G::COLS]; m = v < m ? v : m; }
            if (col > 0 && mine.test(a - 1)) { const uint32_t v = lds->xu[a - 1]; m = v < m ? v : m; }
            if (row < G::ROWS - 1 && mine.test(a + G::COLS)) { const uint32_t v = lds->xu[a + G::COLS]; m = v < m ? v : m; }
            if (col < G::COLS - 1 && mine.test(a + 1)) { const uint32_t v = lds->xu[a + 1]; m = v < m ? v : m; }
            const uint32_t j = lds->xu[m];               // the label's own label: halves the remaining distance
            m = j < m ? j : m;
            changed |= m != lab[st];
            lab[st] = m;
        }
        wv::sync();
        if (wv::ballot(changed) == 0) break;
    }
    // 2. liberties of a group = empty points touching it, each counted once
    for (int st = 0; st < WS; ++st) {
        const int a = st * 64 + l;
        if (a < G::CELLS && empty.test(a)) {
            const int row = a / G::COLS, col = a % G::COLS;
            uint32_t seen[4];
            int ns = 0;
            for (int d = 0; d < 4; ++d) {
                const bool valid = d == 0 ? row > 0 : d == 1 ? col > 0 : d == 2 ? row < G::ROWS - 1 : col < G::COLS - 1;
                if (!valid) continue;
                const int nb = d == 0 ? a - G::COLS : d == 1 ? a - 1 : d == 2 ? a + G::COLS : a + 1;
                if (empty.test(nb)) continue;
                const uint32_t g = lds->xu[nb];
                bool dup = false;
                for (int k = 0; k < ns; ++k) dup |= seen[k] == g;
                if (!dup) {
                    seen[ns++] = g;
                    wv::atomic_add_u32(&lds->xl[g], 1u);
                }
            }
        }
    }
    wv::sync();
    // 3. signatures of the opponent groups in atari (the only ones a move can capture): every stone of such a group xors its
    //    contribution into the group's entry.  Nothing to do while no group is in atari (most of the game).
    bool in_atari[WS];
    uint64_t any_atari = 0;
    for (int st = 0; st < WS; ++st) {
        in_atari[st] = colr[st] == 2 && lds->xl[lab[st]] == 1u;
        any_atari |= wv::ballot(in_atari[st]);
    }
    if (any_atari) {
        for (int st = 0; st < WS; ++st)
            if (in_atari[st]) {
                const uint64_t sg = stone_sig(st * 64 + l, c.player ? 0 : 1);          // the opponent's plane
                wv::atomic_xor_u32(&lds->xs[lab[st]][0], (uint32_t)sg);
                wv::atomic_xor_u32(&lds->xs[lab[st]][1], (uint32_t)(sg >> 32));
            }
        wv::sync();
    }
    // 4. every empty point: liberties after the move, signature of the position it would give, signature lookup in the line.
    //    A point whose signature matches an ancestor's is a SUSPECT; suspects are verified on whole positions below.
    const uint64_t cur_sig = pos_sig<G::WORDS>(c.p0, c.p1);
    BB legal = BB::zero();
    uint64_t suspect[WS];
    for (int st = 0; st < WS; ++st) {
        const int a = st * 64 + l;
        bool ok = false, sus = false;
        if (a < G::CELLS && empty.test(a)) {
            const int row = a / G::COLS, col = a % G::COLS;
            bool has_libs = false;
            uint32_t capg[4];
            int nc = 0;
            uint64_t sg = cur_sig ^ stone_sig(a, c.player);
            for (int d = 0; d < 4; ++d) {
                const bool valid = d == 0 ? row > 0 : d == 1 ? col > 0 : d == 2 ? row < G::ROWS - 1 : col < G::COLS - 1;
                if (!valid) continue;
                const int nb = d == 0 ? a - G::COLS : d == 1 ? a - 1 : d == 2 ? a + G::COLS : a + 1;
                if (empty.test(nb)) has_libs = true;
                else {
                    const uint32_t g = lds->xu[nb], libs = lds->xl[g];
                    if (own.test(nb)) { if (libs > 1) has_libs = true; }
                    else if (libs == 1) {
                        has_libs = true;
                        bool dup = false;
                        for (int k = 0; k < nc; ++k) dup |= capg[k] == g;
                        if (!dup) {                                   // a group touching the point twice is captured once
                            capg[nc++] = g;
                            sg ^= (uint64_t)lds->xs[g][0] | ((uint64_t)lds->xs[g][1] << 32);
                        }
                    }
                }
            }
            if (has_libs) {
                for (int i = n_game; i < n_hist; ++i) sus |= lds->hsig[i] == sg;          // the current descent: a few positions
                if (bloom_hit<G>(lds, sg))                                                  // the real game: only on a filter hit
                    for (int i = 0; i < n_game; ++i) sus |= lds->hsig[i] == sg;
            }
            ok = has_libs;
        }
        legal.w[st] = wv::ballot(ok);
        suspect[st] = wv::ballot(sus);
    }
    // 5. suspects (a true repetition, or a signature collision): build the position the move gives - captured groups from the
    //    labels by wave-wide ballots - and compare it with the ancestors of equal signature, whole bit sets from the ring in HBM.
    //    Wave-uniform and rare (a ko or a retaken position).
    for (int st = 0; st < WS; ++st)
        for (uint64_t m = suspect[st]; m; m &= m - 1) {
            const int bitpos = wv::ctz64(m), a = st * 64 + bitpos;
            const int row = a / G::COLS, col = a % G::COLS;
            BB cap = BB::zero();
            for (int d = 0; d < 4; ++d) {
                const bool valid = d == 0 ? row > 0 : d == 1 ? col > 0 : d == 2 ? row < G::ROWS - 1 : col < G::COLS - 1;
                if (!valid) continue;
                const int nb = d == 0 ? a - G::COLS : d == 1 ? a - 1 : d == 2 ? a + G::COLS : a + 1;
                if (!opp.test(nb)) continue;
                const uint32_t g = lds->xu[nb];
                if (lds->xl[g] != 1u) continue;
                for (int s2 = 0; s2 < WS; ++s2) cap.w[s2] |= wv::ballot(colr[s2] == 2 && lab[s2] == g);
            }
            const BB nown = own | BB::bit(a), nopp = opp & ~cap;
            const BB np0 = c.player ? nopp : nown, np1 = c.player ? nown : nopp;
            const uint64_t sg = pos_sig<G::WORDS>(np0, np1);
            bool repeat = false;
            for (int i = 0; i < n_hist; ++i)
                if (lds->hsig[i] == sg) repeat |= (ring[2 * i] == np0) && (ring[2 * i + 1] == np1);
            if (repeat) legal.w[st] &= ~(1ull << bitpos);
        }
    return legal;
}

template <class G>
SPRL_DEV void make_child(GameW& g, WaveLdsW<G>* lds, const PosW<G::WORDS>& parent, int action, int at,
                         PosW<G::WORDS>& cs) {
    G::apply(parent, action, cs);
    Bits<G::WORDS>* ring = (Bits<G::WORDS>*)g.ring;
    if (wv::lane() == 0) hist_put<G>(lds, ring, at, cs.p0, cs.p1);
    wv::wave_fence();                          // the ring entry is read by other lanes (suspect verification, leaf history)
    if (!cs.terminal) cs.legal = go_legal_mask<G>(cs, lds, ring, at + 1, g.ply + 1 < at + 1 ? g.ply + 1 : at + 1, g.legal_form);
}

// Dirichlet root noise (UCTNode.hpp:330-347, utils/random.cpp:61-74)
template <class G>
SPRL_DEV void mix_root_noise(const EngineParams& P, GameW& g, uint8_t* np, const Bits<G::WORDS>& legal, bool pass_legal,
                             float pass_p) {
    const int l = wv::lane();
    NodeHdrW<G::WORDS>* h = hdr_of<G>(np);
    const int num_legal = legal.popc() + (pass_legal ? 1 : 0);
    bool mine[WS];
    int my_rank[WS];
    float my_noise[WS];
    for (int st = 0; st < WS; ++st) {
        const int a = st * 64 + l;
        mine[st] = a < G::NA && legal.test(a);
        my_rank[st] = a < G::NA ? legal.rank(a) : 0;
        my_noise[st] = 0.0f;
    }
    NormalState ns = { 0.0f, 0 };
    float pass_noise = 0.0f, sum = 0.0f;
    for (int i = 0; i < num_legal; ++i) {
        float s = rng_gamma(g.rng, ns, P.dir_alpha);
        sum += s;
        for (int st = 0; st < WS; ++st)
            if (mine[st] && my_rank[st] == i) my_noise[st] = s;
        if (i == num_legal - 1 && pass_legal) pass_noise = s;
    }
    const float norm = 1.0f / sum;
    const double keep = 1.0 - (double)P.dir_eps;
    for (int st = 0; st < WS; ++st)
        if (mine[st]) {
            float p = rowP<G>(np)[st * 64 + l];
            rowP<G>(np)[st * 64 + l] = (float)(keep * (double)p + (double)(P.dir_eps * (my_noise[st] * norm)));
        }
    if (pass_legal) h->passP = (float)(keep * (double)pass_p + (double)(P.dir_eps * (pass_noise * norm)));
}

template <class G>
SPRL_DEV void expand_node(const EngineParams& P, GameW& g, uint8_t* np, const NodeHdrW<G::WORDS>& hc, bool add_noise) {
    const int l = wv::lane();
    NodeHdrW<G::WORDS>* h = hdr_of<G>(np);
    const bool pass_legal = (hc.flags & F_PASS) != 0;
    for (int st = 0; st < WS; ++st) {
        rowN<G>(np)[st * 64 + l] = 0.0f;
        rowW<G>(np)[st * 64 + l] = 0.0f;
    }
    h->passN = 0.0f;
    h->passW = 0.0f;
    h->exp_epoch = g.epoch;
    if (add_noise) mix_root_noise<G>(P, g, np, hc.legal, pass_legal, hc.passP);
    g.d_expansions++;
}

// backup over a stored path: lane j owns edges j, j + 64, ... (UCTTree.hpp:261-273)
template <class G>
SPRL_DEV void backup_path(GameW& g, const uint32_t* path, int depth, int leaf_player, float value) {
    const int l = wv::lane();
    wv::sync();
    const float est = -value * (leaf_player == 0 ? 1.0f : -1.0f);
    g.rootW += 1.0f + est * (g.root_player == 0 ? 1.0f : -1.0f);
    for (int j = l; j < depth; j += 64) {
        const uint32_t e = path[j];
        const uint32_t node = e >> 10;
        const int a = (int)(e & 0x3ffu);
        const int child_player = (int)(g.root_player ^ ((uint32_t)(j + 1) & 1u));
        const float add = 1.0f + est * (child_player == 0 ? 1.0f : -1.0f);
        uint8_t* np = node_at<G>(g.abase, node);
        float* wp = a == WPASS ? &hdr_of<G>(np)->passW : &rowW<G>(np)[a];
        *wp = *wp + add;
    }
    wv::wave_fence();
}

// evaluator output -> cached policy in the P rows (UCTTree.hpp:136-175, GridNetwork.hpp:104-138, RandomNetwork.hpp)
template <class G>
SPRL_DEV void evaluate_leaf(const EngineParams& P, GameW& g, WaveLdsW<G>* lds, uint8_t* np, NodeHdrW<G::WORDS>& hc,
                            int sym, const float* logits, float nn_value) {
    using BB = Bits<G::WORDS>;
    const int l = wv::lane();
    NodeHdrW<G::WORDS>* h = hdr_of<G>(np);
    const BB legal = hc.legal;
    const bool pass_legal = (hc.flags & F_PASS) != 0;
    BB used = legal;
    if (P.mask_frame == MASK_SYMMETRISED && sym != 0) {
        for (int st = 0; st < WS; ++st) {
            const int a = st * 64 + l;
            const int src = a < G::NA ? G::map_action(G::inverse_sym(sym), a) : 0;
            used.w[st] = wv::ballot(a < G::NA && legal.test(src));
        }
    }
    const int num_legal = used.popc() + (pass_legal ? 1 : 0);
    float pol[WS], pass_pol = 0.0f, value = 0.0f;
    if (P.eval_kind == EVAL_NETWORK) {
        float e[WS];
        for (int st = 0; st < WS; ++st) {
            const int a = st * 64 + l;
            e[st] = (a < G::NA && used.test(a)) ? sprl_expf(logits[a]) : 0.0f;
        }
        const float e_pass = pass_legal ? sprl_expf(logits[G::A - 1]) : 0.0f;
        float sum = 0.0f;
        for (int st = 0; st < WS; ++st)
            for (uint64_t m = used.w[st]; m; m &= m - 1) sum += wv::bcast_f32(e[st], wv::ctz64(m));
        if (pass_legal) sum += e_pass;
        if (sum == 0.0f) {
            const float uniform = 1.0f / (float)num_legal;
            for (int st = 0; st < WS; ++st) pol[st] = (st * 64 + l < G::NA && used.test(st * 64 + l)) ? uniform : 0.0f;
            pass_pol = pass_legal ? uniform : 0.0f;
        } else {
            const float inv = 1.0f / sum;
            for (int st = 0; st < WS; ++st) pol[st] = e[st] * inv;
            pass_pol = e_pass * inv;
        }
        value = nn_value;
    } else {
        const float uniform = 1.0f / (float)num_legal;
        for (int st = 0; st < WS; ++st) pol[st] = (st * 64 + l < G::NA && used.test(st * 64 + l)) ? uniform : 0.0f;
        pass_pol = pass_legal ? uniform : 0.0f;
    }
    // undo the symmetry through LDS: policy_orig[a] = policy_sym[map_s(a)]
    wv::sync();
    for (int st = 0; st < WS; ++st) lds->xf[st * 64 + l] = pol[st];
    wv::sync();
    for (int st = 0; st < WS; ++st) {
        const int a = st * 64 + l;
        const bool legal_here = a < G::NA && legal.test(a);
        const float v = a < G::NA ? lds->xf[G::map_action(sym, a)] : 0.0f;
        rowP<G>(np)[a] = legal_here ? v : 0.0f;
    }
    hc.passP = pass_legal ? pass_pol : 0.0f;
    hc.value = value;
    hc.flags = (uint8_t)(hc.flags | F_EVAL);
    h->passP = hc.passP;
    h->value = hc.value;
    h->flags = hc.flags;
}

template <class G>
SPRL_DEV void encode_leaf(const EngineParams& P, const WaveLdsW<G>* lds, int q, int player, int sym, int nn_slot) {
    const int l = wv::lane();
    float* out = P.nn_in + (size_t)nn_slot * (G::PLANES * G::CELLS);
    const int size = (int)lds->leaf_size[q];
    for (int st = 0; st < WS; ++st) {
        const int a = st * 64 + l;
        if (a >= G::CELLS) continue;
        const int src = G::map_cell(G::inverse_sym(sym), a);
        for (int t = 0; t < G::HIST; ++t) {
            const Bits<G::WORDS>& p0 = lds->leaf_hist[q][t][0];
            const Bits<G::WORDS>& p1 = lds->leaf_hist[q][t][1];
            const bool own = player ? p1.test(src) : p0.test(src), opp = player ? p0.test(src) : p1.test(src);
            out[(2 * t) * G::CELLS + a] = (t < size && own) ? 1.0f : 0.0f;
            out[(2 * t + 1) * G::CELLS + a] = (t < size && opp) ? 1.0f : 0.0f;
        }
        out[(2 * G::HIST) * G::CELLS + a] = player == 0 ? 1.0f : 0.0f;
    }
}

template <class G>
SPRL_DEV void finish_leaves(const EngineParams& P, GameW& g, int slot, GameCtl* ctl, WaveLdsW<G>* lds) {
    const int nn_base = (int)P.leaf_offset[slot];
    for (int q = 0; q < g.n_leaves; ++q) {
        const uint32_t leaf = ctl->leaf_node[q];
        const int depth = (int)ctl->leaf_depth[q];
        const int sym = (int)ctl->leaf_sym[q];
        uint8_t* np = node_at<G>(g.abase, leaf);
        NodeHdrW<G::WORDS> h = load_hdr<G>(np);
        const float* logits = P.nn_logits + (size_t)(nn_base + q) * G::A;
        const float nn_value = P.eval_kind == EVAL_NETWORK ? P.nn_value[nn_base + q] : 0.0f;
        if (!(h.flags & F_EVAL)) evaluate_leaf<G>(P, g, lds, np, h, sym, logits, nn_value);
        else g.d_dup++;
        if (h.exp_epoch != g.epoch) expand_node<G>(P, g, np, h, P.add_noise && leaf == g.root);
        const uint32_t* path = P.paths + ((size_t)slot * SPRL_MAXQ + q) * P.max_depth;
        backup_path<G>(g, path, depth, h.player, h.value);
    }
    g.n_leaves = 0;
}

template <class G>
SPRL_DEV void select_batch(const EngineParams& P, GameW& g, int slot, GameCtl* ctl, WaveLdsW<G>* lds) {
    const int l = wv::lane();
    int trav = 0;
    while (trav < P.max_batch) {
        ++trav;
        uint32_t cur = g.root;
        float nself = g.rootN;
        g.rootN += 1.0f;
        g.rootW -= 1.0f;
        int depth = 0;
        NodeHdrW<G::WORDS> h = load_hdr<G>(node_at<G>(g.abase, cur));       // (its position is entry g.ply of the line already)
        while (h.exp_epoch == g.epoch && !(h.flags & F_TERMINAL)) {
            uint8_t* np = node_at<G>(g.abase, cur);
            float n[WS], w[WS], score[WS];
            uint32_t ch[WS];
            bool lg[WS];
            const float sq = __builtin_sqrtf(nself);
            float local = -__builtin_inff();
            for (int st = 0; st < WS; ++st) {
                const int a = st * 64 + l;
                n[st] = rowN<G>(np)[a];
                w[st] = rowW<G>(np)[a];
                const float p = rowP<G>(np)[a];
                ch[st] = rowC<G>(np)[a];
                lg[st] = a < G::NA && h.legal.test(a);
                const float den = 1.0f + n[st];
                score[st] = w[st] / den + P.u_weight * (p * sq / den);       // UCTNode.hpp:200,210,236
                if (lg[st] && score[st] > local) local = score[st];
            }
            wv::sync();
            const float pden = 1.0f + h.passN;
            const float pscore = h.passW / pden + P.u_weight * (h.passP * sq / pden);
            float best = wv::fmax_all(local);
            best = pscore > best ? pscore : best;
            uint64_t ties[WS];
            int kb = 0;
            for (int st = 0; st < WS; ++st) {
                ties[st] = wv::ballot(lg[st] && score[st] == best);
                kb += wv::popc64(ties[st]);
            }
            const int k = kb + (pscore == best ? 1 : 0);
            int r = rng_uniform_int(g.rng, (uint32_t)k);                     // UCTNode.hpp:250
            int a = WPASS;
            if (r < kb) {
                for (int st = 0; st < WS; ++st) {
                    const int c = wv::popc64(ties[st]);
                    if (r < c) { a = st * 64 + wv::nth_set_bit(ties[st], r); break; }
                    r -= c;
                }
            }
            g.d_levels++;
            if (depth >= P.max_depth) { raise_error(P, g, ERR_MAX_DEPTH); return; }
            if (l == 0) lds->path[depth] = (cur << 10) | (uint32_t)a;
            ++depth;
            NodeHdrW<G::WORDS>* hp = hdr_of<G>(np);
            uint32_t c = SPRL_NONE16;
            float n_a = 0.0f, w_a = 0.0f;
            if (a == WPASS) {
                c = h.passChild;
                n_a = h.passN;
                w_a = h.passW;
            } else {
                const int sa = a >> 6, la = a & 63;
                for (int st = 0; st < WS; ++st)
                    if (st == sa) {
                        c = wv::bcast_u32(ch[st], la);
                        n_a = wv::bcast_f32(n[st], la);
                        w_a = wv::bcast_f32(w[st], la);
                    }
            }
            bool created = false;
            PosW<G::WORDS> cs;
            if (c == SPRL_NONE16) {
                c = alloc_node<G>(g, lds);
                created = true;
                make_child<G>(g, lds, pos_of<G>(h), a, g.ply + depth, cs);
                write_new_node<G>(node_at<G>(g.abase, c), cs, a);
                w_a = P.init_q_zero ? 0.0f : h.value;                        // InitQ::PARENT / ZERO (UCTNode.hpp:258-284)
                g.d_created++;
                if (a == WPASS) hp->passChild = c;
                else if ((a & 63) == l) rowC<G>(np)[a] = (uint16_t)c;
            }
            if (a == WPASS) {
                hp->passN = n_a + 1.0f;
                hp->passW = w_a - 1.0f;
            } else if ((a & 63) == l) {
                rowN<G>(np)[a] = n_a + 1.0f;
                rowW<G>(np)[a] = w_a - 1.0f;
            }
            nself = n_a;
            cur = c;
            if (created) {
                h.p0 = cs.p0; h.p1 = cs.p1; h.legal = cs.legal; h.value = 0.0f; h.exp_epoch = 0;
                h.passChild = SPRL_NONE16; h.player = cs.player; h.winner = cs.winner;
                h.flags = (uint8_t)((cs.terminal ? F_TERMINAL : 0) | (cs.pass_legal ? F_PASS : 0));
                h.action = (uint16_t)a; h.depth = cs.depth;
                break;
            }
            h = load_hdr<G>(node_at<G>(g.abase, cur));
            if (l == 0) hist_put<G>(lds, (Bits<G::WORDS>*)g.ring, g.ply + depth, h.p0, h.p1);
        }
        if (h.flags & F_TERMINAL) {
            const float value = h.winner < 0 ? 0.0f : (h.winner == (int8_t)h.player ? 1.0f : -1.0f);
            backup_path<G>(g, lds->path, depth, h.player, value);
            g.d_terminal++;
            continue;
        } else if (h.flags & F_EVAL) {
            expand_node<G>(P, g, node_at<G>(g.abase, cur), h, P.add_noise && cur == g.root);
            backup_path<G>(g, lds->path, depth, h.player, h.value);
            g.d_gray++;
            continue;
        } else {
            const int q = g.n_leaves++;
            ctl->leaf_node[q] = cur;
            ctl->leaf_depth[q] = (uint32_t)depth;
            ctl->leaf_player[q] = h.player;
            uint32_t* path = P.paths + ((size_t)slot * SPRL_MAXQ + q) * P.max_depth;
            wv::sync();
            for (int j = l; j < depth; j += 64) path[j] = lds->path[j];
            const int last = g.ply + depth;
            const int size = last + 1 < G::HIST ? last + 1 : G::HIST;
            wv::wave_fence();                  // lane 0 wrote the line's positions, lanes 0..7 read them
            if (l < size) {
                const Bits<G::WORDS>* ring = (const Bits<G::WORDS>*)g.ring;
                lds->leaf_hist[q][l][0] = ring[2 * (last - l)];
                lds->leaf_hist[q][l][1] = ring[2 * (last - l) + 1];
            }
            if (l == 0) lds->leaf_size[q] = (uint32_t)size;
        }
        if (g.n_leaves >= P.max_queue) break;
    }
    g.traversals += trav;
    g.d_traversals += (uint32_t)trav;
    for (int q = 0; q < g.n_leaves; ++q) {
        int sym = 0;
        if (P.use_sym) sym = rng_uniform_int(g.rng, (uint32_t)G::NSYM);
        ctl->leaf_sym[q] = (uint32_t)sym;
        if (P.eval_kind == EVAL_NETWORK) {
            wv::sync();
            encode_leaf<G>(P, lds, q, (int)ctl->leaf_player[q], sym, slot * P.max_queue + q);
        }
    }
    g.d_nn_evals += (uint32_t)g.n_leaves;
    if (P.rec_evals && g.n_leaves && wv::lane() == 0) P.rec_evals[g.game_id] += (uint32_t)g.n_leaves;    // (one wave owns the game)
}

// arena compaction: Cheney copy of the live subtree into a spare arena (same protocol as step_kernel.h)
template <class G>
SPRL_DEV void copy_node(const uint8_t* src, uint8_t* dst) {
    struct alignas(16) Chunk { uint32_t x[4]; };
    for (int i = wv::lane(); i < G::NODE_BYTES / 16; i += 64) ((Chunk*)dst)[i] = ((const Chunk*)src)[i];
}

template <class G>
SPRL_DEV bool compact_arena(const EngineParams& P, GameW& g) {
    const int l = wv::lane();
    uint32_t got = 0xFFFFFFFFu;
    if (l == 0) {
        const uint32_t total = (uint32_t)(P.num_slots + P.num_spare);
        for (uint32_t i = 0; i < total; ++i) {
            uint32_t idx = (g.arena + 1 + i) % total;
            if (wv::atomic_cas_u32(&P.arena_used[idx], 0u, 1u) == 0u) { got = idx; break; }
        }
    }
    got = wv::bcast_u32(got, 0);
    if (got == 0xFFFFFFFFu) { raise_error(P, g, ERR_NO_SPARE); return false; }
    wv::agent_acquire();
    uint8_t* from = g.abase;
    uint8_t* to = P.arenas + (size_t)got * (size_t)P.node_cap * G::NODE_BYTES;
    copy_node<G>(node_at<G>(from, g.root), node_at<G>(to, 0));
    wv::wave_fence();
    uint32_t scan = 0, free_ = 1;
    while (scan < free_) {
        uint8_t* np = node_at<G>(to, scan);
        for (int st = 0; st < WS; ++st) {
            uint32_t ch = rowC<G>(np)[st * 64 + l];
            const bool has = ch != SPRL_NONE16;
            const uint64_t mask = wv::ballot(has);
            if (has) rowC<G>(np)[st * 64 + l] = (uint16_t)(free_ + (uint32_t)wv::popc64(mask & wv::lt_mask(l)));
            uint32_t k = 0;
            for (uint64_t m = mask; m; m &= m - 1, ++k) {
                uint32_t src = wv::bcast_u32(ch, wv::ctz64(m));
                copy_node<G>(node_at<G>(from, src), node_at<G>(to, free_ + k));
            }
            free_ += k;
        }
        NodeHdrW<G::WORDS>* h = hdr_of<G>(np);
        const uint32_t pc = h->passChild;
        wv::sync();
        if (pc != SPRL_NONE16) {
            copy_node<G>(node_at<G>(from, pc), node_at<G>(to, free_));
            h->passChild = free_;
            ++free_;
        }
        wv::wave_fence();
        ++scan;
    }
    wv::agent_release();
    if (l == 0) wv::atomic_store_u32(&P.arena_used[g.arena], 0u);
    g.arena = got;
    g.abase = to;
    g.root = 0;
    g.n_alloc = free_;
    g.rstack_n = 0;                           // the garbage stayed behind in the arena that was given back
    g.fc_n = 0;
    g.d_compactions++;
    return true;
}

template <class G>
SPRL_DEV void init_game(const EngineParams& P, GameW& g, int slot, WaveLdsW<G>* lds, uint32_t gid);

template <class G>
SPRL_DEV void start_game(const EngineParams& P, GameW& g, int slot, WaveLdsW<G>* lds) {
    uint32_t gid = 0;
    if (wv::lane() == 0) gid = wv::atomic_add_u32(&P.counters->next_game, 1u);
    gid = wv::bcast_u32(gid, 0);
    init_game<G>(P, g, slot, lds, gid);
}

// game `gid` starts in this slot (self-play: the next game of the global counter; match play: the pair's next game)
template <class G>
SPRL_DEV void init_game(const EngineParams& P, GameW& g, int slot, WaveLdsW<G>* lds, uint32_t gid) {
    if (gid >= (uint32_t)P.num_games) {
        g.status = ST_IDLE;
        return;
    }
    g.status = ST_ACTIVE;
    g.game_id = gid;
    rng_seed(g.rng, P.seed, P.stream_base + (int)gid);
    PosW<G::WORDS> s;
    G::start(s);
    write_new_node<G>(node_at<G>(g.abase, 0), s, 0);
    g.root = 0;
    g.n_alloc = 1;
    g.rstack_n = 0;
    g.fc_n = 0;
    g.epoch = 1;
    g.root_player = 0;
    g.rootN = 0.0f;
    g.rootW = 0.0f;
    g.ply = 0;
    g.traversals = 0;
    g.n_leaves = 0;
    g.d_created++;
    bloom_clear<G>(lds);
    if (wv::lane() == 0) {
        hist_put<G>(lds, (Bits<G::WORDS>*)g.ring, 0, s.p0, s.p1);
        bloom_add<G>(lds, lds->hsig[0]);
    }
    wv::wave_fence();
}

template <class G>
SPRL_DEV void advance_root(const EngineParams& P, GameW& g, int slot, WaveLdsW<G>* lds, int action);

// SelfPlay.hpp:110-148
template <class G>
SPRL_DEV int play_move(const EngineParams& P, GameW& g, int slot, WaveLdsW<G>* lds) {
    const int l = wv::lane();
    uint8_t* np = node_at<G>(g.abase, g.root);
    const NodeHdrW<G::WORDS> h = load_hdr<G>(np);
    float visits[WS], pdf[WS], cdf[WS];
    uint64_t nz[WS];
    for (int st = 0; st < WS; ++st) {
        const int a = st * 64 + l;
        visits[st] = a < G::NA ? rowN<G>(np)[a] : 0.0f;
    }
    wv::sync();
    const float pass_visits = h.passN;
    float sum = 0.0f;
    for (int st = 0; st < WS; ++st) {
        nz[st] = wv::ballot(visits[st] != 0.0f);
        for (uint64_t m = nz[st]; m; m &= m - 1) sum += wv::bcast_f32(visits[st], wv::ctz64(m));
    }
    sum += pass_visits;
    float inv = 1.0f / sum;
    const float ex = g.ply < P.early_cutoff ? P.early_exp : P.rest_exp;
    float pass_pdf = sprl_powf(pass_visits * inv, ex);
    for (int st = 0; st < WS; ++st) pdf[st] = sprl_powf(visits[st] * inv, ex);
    sum = 0.0f;
    for (int st = 0; st < WS; ++st)
        for (uint64_t m = nz[st]; m; m &= m - 1) sum += wv::bcast_f32(pdf[st], wv::ctz64(m));
    sum += pass_pdf;
    inv = 1.0f / sum;
    for (int st = 0; st < WS; ++st) pdf[st] = pdf[st] * inv;
    pass_pdf = pass_pdf * inv;
    float run = 0.0f;
    for (int st = 0; st < WS; ++st) {
        cdf[st] = run;                                   // prefix over all earlier strips
        for (uint64_t m = nz[st]; m; m &= m - 1) {
            const int k = wv::ctz64(m);
            run += wv::bcast_f32(pdf[st], k);
            if (l >= k) cdf[st] = run;
        }
    }
    float last = run + pass_pdf;
    inv = 1.0f / last;
    for (int st = 0; st < WS; ++st) cdf[st] = cdf[st] * inv;
    last = last * inv;
    if (g.ply >= P.max_plies) { raise_error(P, g, ERR_MAX_PLIES); return 0; }
    const size_t rec = (size_t)g.game_id * (size_t)P.max_plies + (size_t)g.ply;
    Bits<G::WORDS>* rb = (Bits<G::WORDS>*)P.rec_boards + rec * 2;
    rb[0] = h.p0;
    rb[1] = h.p1;
    P.rec_movers[rec] = h.player;
    for (int st = 0; st < WS; ++st)
        if (st * 64 + l < G::NA) P.rec_pdf[rec * G::A + st * 64 + l] = pdf[st];
    P.rec_pdf[rec * G::A + (G::A - 1)] = pass_pdf;
    if (P.resign_threshold > 0.0f && g.ply >= P.resign_min_ply) {     // extension (see step_kernel.h: play_move), default off
        float sn = 0.0f, sw = 0.0f;
        for (int st = 0; st < WS; ++st)
            for (uint64_t m = nz[st]; m; m &= m - 1) sn += wv::bcast_f32(visits[st], wv::ctz64(m));
        sn += pass_visits;
        for (int st = 0; st < WS; ++st) {
            const float wrow = st * 64 + l < G::NA ? rowW<G>(np)[st * 64 + l] : 0.0f;
            for (uint64_t m = nz[st]; m; m &= m - 1) sw += wv::bcast_f32(wrow, wv::ctz64(m));
        }
        sw += h.passW;
        const float v = sw * (1.0f / sn);
        if (v < -P.resign_threshold) {
            g.d_plies++;
            return 2 - (int)h.player;
        }
    }
    float e;
    do {
        e = rng_uniform_float(g.rng);
    } while (e == 0.0f);
    const float x = last * e;
    int action = WPASS;
    for (int st = 0; st < WS; ++st) {
        const uint64_t ge = wv::ballot(st * 64 + l < G::NA && !(cdf[st] < x));
        if (action == WPASS && ge) action = st * 64 + wv::ctz64(ge);
    }
    advance_root<G>(P, g, slot, lds, action);
    g.d_plies++;
    return 0;
}

// UCTTree::advanceDecision (uct/UCTTree.hpp:197-210): the decision node moves along `action` (self-play: the sampled move;
// match play: the agent's own move or the opponent's), its edge statistics become the new root's own N / W (Q6), the epoch
// bump turns every active node gray, the old root goes on the reclaim stack
template <class G>
SPRL_DEV void advance_root(const EngineParams& P, GameW& g, int slot, WaveLdsW<G>* lds, int action) {
    const int l = wv::lane();
    uint8_t* np = node_at<G>(g.abase, g.root);
    const NodeHdrW<G::WORDS> h = load_hdr<G>(np);
    const bool expanded = h.exp_epoch == g.epoch;      // rows of a node that is not active hold no valid statistics
    uint32_t c = SPRL_NONE16;
    float n_a = 0.0f, w_a = 0.0f;
    if (action == WPASS) {
        c = h.passChild;
        n_a = h.passN;
        w_a = h.passW;
    } else {
        const int sa = action >> 6, la = action & 63;
        for (int st = 0; st < WS; ++st) {
            const uint32_t cc = wv::bcast_u32((uint32_t)rowC<G>(np)[st * 64 + l], la);
            const float nn = wv::bcast_f32(rowN<G>(np)[st * 64 + l], la);
            const float ww = wv::bcast_f32(rowW<G>(np)[st * 64 + l], la);
            if (st == sa) { c = cc; n_a = nn; w_a = ww; }
        }
    }
    if (!expanded) { n_a = 0.0f; w_a = 0.0f; }
    wv::sync();
    if (c == SPRL_NONE16) {
        PosW<G::WORDS> cs;
        make_child<G>(g, lds, pos_of<G>(h), action, g.ply + 1, cs);
        c = alloc_node<G>(g, lds);
        write_new_node<G>(node_at<G>(g.abase, c), cs, action);
        g.d_created++;
        w_a = (!P.init_q_zero && (h.flags & F_EVAL)) ? h.value : 0.0f;
    } else if (P.recycle) {                   // pruneChildrenExcept: cut the edge to the kept child, the rest is garbage
        if (action == WPASS) hdr_of<G>(np)->passChild = SPRL_NONE16;
        else if ((action & 63) == l) rowC<G>(np)[action] = (uint16_t)SPRL_NONE16;
        wv::wave_fence();
    }
    if (P.recycle) {
        uint32_t* stack = P.reclaim + (size_t)slot * (size_t)P.node_cap;
        if (l == 0) stack[g.rstack_n] = g.root;
        g.rstack_n += 1;
        wv::wave_fence();
    }
    g.root = c;
    g.rootN = n_a;
    g.rootW = w_a;
    g.root_player ^= 1u;
    g.ply += 1;
    {
        const NodeHdrW<G::WORDS> nh = load_hdr<G>(node_at<G>(g.abase, c));
        if (l == 0) {                                                                   // the real game's line grows by one
            hist_put<G>(lds, (Bits<G::WORDS>*)g.ring, g.ply, nh.p0, nh.p1);
            bloom_add<G>(lds, lds->hsig[g.ply]);
        }
        wv::wave_fence();
    }
    g.epoch += 1;
    g.traversals = 0;
}

// the slot's control block -> the wave's working state (and the line's signatures + their filter back into LDS)
template <class G>
SPRL_DEV void game_load(const EngineParams& P, int slot, GameCtl* ctl, WaveLdsW<G>* lds, GameW& g) {
    g.rng.state = ctl->rng_state;
    g.rng.inc = ctl->rng_inc;
    g.game_id = ctl->game_id;
    g.arena = ctl->arena;
    g.root = ctl->root;
    g.n_alloc = ctl->n_alloc;
    g.epoch = ctl->epoch;
    g.rootN = ctl->rootN;
    g.rootW = ctl->rootW;
    g.ply = ctl->ply;
    g.traversals = ctl->traversals;
    g.n_leaves = ctl->n_leaves;
    g.root_player = ctl->root_player;
    g.rstack_n = ctl->rstack_n;
    g.fc_n = ctl->fc_n;
    if (wv::lane() < SPRL_FCACHE) lds->fcache[wv::lane()] = ctl->fcache[wv::lane()];
    g.hi_alloc = g.n_alloc;
    g.d_traversals = g.d_levels = g.d_expansions = g.d_nn_evals = g.d_terminal = g.d_gray = g.d_dup = 0;
    g.d_created = g.d_compactions = g.d_games = g.d_plies = g.d_recycled = 0;
    g.legal_form = P.go_legal_form;
    g.abase = P.arenas + (size_t)g.arena * (size_t)P.node_cap * G::NODE_BYTES;
    g.ring = (void*)((Bits<G::WORDS>*)P.hist_boards + (size_t)slot * G::HIST_CAP * 2);
    wv::sync();
    if (g.status == ST_ACTIVE) {               // the signatures of the real game's positions, from the ring, and their filter
        bloom_clear<G>(lds);
        const Bits<G::WORDS>* gh = (const Bits<G::WORDS>*)g.ring;
        for (int i = wv::lane(); i <= g.ply; i += 64) {
            const uint64_t sg = pos_sig<G::WORDS>(gh[2 * i], gh[2 * i + 1]);
            lds->hsig[i] = sg;
            bloom_add<G>(lds, sg);
        }
        wv::sync();
    }
}

template <class G>
SPRL_DEV void game_store(GameCtl* ctl, const GameW& g, WaveLdsW<G>* lds) {
    ctl->status = g.status;
    ctl->rng_state = g.rng.state;
    ctl->rng_inc = g.rng.inc;
    ctl->game_id = g.game_id;
    ctl->arena = g.arena;
    ctl->root = g.root;
    ctl->n_alloc = g.n_alloc;
    ctl->epoch = g.epoch;
    ctl->rootN = g.rootN;
    ctl->rootW = g.rootW;
    ctl->ply = g.ply;
    ctl->traversals = g.traversals;
    ctl->n_leaves = g.n_leaves;
    ctl->root_player = g.root_player;
    ctl->rstack_n = g.rstack_n;
    ctl->fc_n = g.fc_n;
    wv::sync();
    if (wv::lane() < SPRL_FCACHE) ctl->fcache[wv::lane()] = lds->fcache[wv::lane()];
    if (wv::lane() == 0) {
        GameStats& t = ctl->stats;
        t.traversals += g.d_traversals;
        t.levels += g.d_levels;
        t.expansions += g.d_expansions;
        t.nn_evals += g.d_nn_evals;
        t.terminal_hits += g.d_terminal;
        t.gray_hits += g.d_gray;
        t.dup_hits += g.d_dup;
        t.nodes_created += g.d_created;
        t.compactions += g.d_compactions;
        t.games += g.d_games;
        t.plies += g.d_plies;
        t.nodes_recycled += g.d_recycled;
        if (g.hi_alloc > t.max_alloc) t.max_alloc = g.hi_alloc;
    }
}

// `need` nodes before a search batch: recycled ids, then fresh arena space, then (rarely) a compaction
template <class G>
SPRL_DEV bool ensure_nodes(const EngineParams& P, GameW& g, int slot, WaveLdsW<G>* lds, uint32_t need) {
    if (P.recycle && g.fc_n < need && g.rstack_n > 0) reclaim_refill<G>(P, g, slot, lds);
    if (g.fc_n + ((uint32_t)P.node_cap - g.n_alloc) < need) {
        if (!compact_arena<G>(P, g)) return false;
        if ((uint32_t)P.node_cap - g.n_alloc < need) { raise_error(P, g, ERR_ARENA_FULL); return false; }
    }
    return true;
}

template <class G>
SPRL_DEV void step_game(const EngineParams& P, int slot, WaveLdsW<G>* lds) {
    GameCtl* ctl = P.ctl + slot;
    GameW g;
    g.status = ctl->status;
    if (g.status == ST_IDLE || g.status == ST_ERROR) return;
    game_load<G>(P, slot, ctl, lds, g);
    if (g.status == ST_FRESH) start_game<G>(P, g, slot, lds);

    int round = 0;
    while (g.status == ST_ACTIVE) {
        if (g.n_leaves > 0) {
            if (P.eval_kind == EVAL_NETWORK && round > 0) break;
            finish_leaves<G>(P, g, slot, ctl, lds);
        }
        bool idle = false;
        while (g.traversals >= P.num_traversals) {
            const int resigned = play_move<G>(P, g, slot, lds);
            if (g.status != ST_ACTIVE) break;
            const NodeHdrW<G::WORDS> rh = load_hdr<G>(node_at<G>(g.abase, g.root));
            if (resigned || (rh.flags & F_TERMINAL)) {
                P.rec_nplies[g.game_id] = resigned ? g.ply + 1 : g.ply;
                P.rec_winner[g.game_id] = resigned ? (int8_t)(resigned - 1) : rh.winner;
                g.d_games++;
                if (wv::lane() == 0) wv::atomic_add_u32(&P.counters->games_done, 1u);
                start_game<G>(P, g, slot, lds);
                if (g.status != ST_ACTIVE) { idle = true; break; }
            }
        }
        if (idle || g.status != ST_ACTIVE) break;
        if (!ensure_nodes<G>(P, g, slot, lds, (uint32_t)P.max_batch + 1u)) break;
        select_batch<G>(P, g, slot, ctl, lds);
        ++round;
        if (P.eval_kind != EVAL_NETWORK && round >= P.rounds) break;
    }

    game_store<G>(ctl, g, lds);
    P.leaf_count[slot] = g.status == ST_ACTIVE ? (uint32_t)g.n_leaves : 0u;
    if (g.status == ST_ACTIVE && wv::lane() == 0) wv::atomic_add_u32(&P.counters->active_slots, 1u);
}

// Match play on the wide boards (Evaluate.cpp:104-170, agents/UCTNetworkAgent.hpp, interface/play.hpp:22-60): the same protocol
// as step_kernel.h: step_match - game g is played by TWO trees in slots (pair, pair + n), only the side to move searches, its
// move (the first maximum of the visit counts, UCTNetworkAgent.hpp:88-89) is applied to its own tree and posted with the game's
// RNG state to the partner's mailbox, which applies it as opponentAct in a LATER launch.
template <class G>
SPRL_DEV void step_match(const EngineParams& P0, int slot, WaveLdsW<G>* lds) {
    const int n = P0.num_slots >> 1;
    const int agent = slot >= n ? 1 : 0;
    const int partner = agent ? slot - n : slot + n;
    EngineParams P = P0;
    P.use_sym = P0.m_use_sym[agent];
    P.eval_kind = P0.m_eval_kind[agent];
    P.init_q_zero = P0.m_init_q_zero[agent];
    GameCtl* ctl = P.ctl + slot;
    GameW g;
    g.status = ctl->status;
    if (g.status == ST_IDLE || g.status == ST_ERROR) {
        P.leaf_count[slot] = 0u;
        return;
    }
    game_load<G>(P, slot, ctl, lds, g);
    if (g.status == ST_FRESH) init_game<G>(P, g, slot, lds, (uint32_t)(slot - agent * n));

    int round = 0;
    while (round < P.rounds && g.status == ST_ACTIVE) {
        if (g.n_leaves > 0) finish_leaves<G>(P, g, slot, ctl, lds);
        const NodeHdrW<G::WORDS> rh = load_hdr<G>(node_at<G>(g.abase, g.root));
        if (rh.flags & F_TERMINAL) {                               // play.hpp:34
            if (agent == 0) {
                P.rec_nplies[g.game_id] = g.ply;
                P.rec_winner[g.game_id] = rh.winner;
                if (wv::lane() == 0) wv::atomic_add_u32(&P.counters->games_done, 1u);
            }
            g.d_games++;
            init_game<G>(P, g, slot, lds, g.game_id + (uint32_t)n);   // this pair's next game, if any
            continue;
        }
        if ((int)g.root_player != (agent ^ (int)(g.game_id & 1u))) {
            // two entries per slot, alternating with the pair's game sequence: the side that ended game g may start game g + n
            // and - if it moves first there - post its first move before the partner has picked up the LAST move of game g
            // (with few traversals per move a whole search fits into one launch); that move must not be overwritten
            const Mailbox mb = P.mailbox[2 * slot + (int)((g.game_id / (uint32_t)n) & 1u)];
            wv::sync();
            const uint32_t m_ply1 = (uint32_t)(mb.ply_launch & 0xffffffffull), m_launch = (uint32_t)(mb.ply_launch >> 32);
            if (m_launch != P.launch_seq && m_ply1 == (uint32_t)g.ply + 1u && mb.game == g.game_id) {
                advance_root<G>(P, g, slot, lds, (int)mb.action);          // opponentAct (UCTNetworkAgent.hpp:106-108)
                g.rng.state = mb.rng_state;
                continue;
            }
            break;                                                  // the partner is still thinking
        }
        if (g.traversals >= P.num_traversals) {
            uint8_t* np = node_at<G>(g.abase, g.root);
            const int l = wv::lane();
            float top = -1.0f;
            float visits[WS];
            for (int st = 0; st < WS; ++st) {
                visits[st] = st * 64 + l < G::NA ? rowN<G>(np)[st * 64 + l] : -1.0f;
                const float m = wv::fmax_all(visits[st]);
                top = m > top ? m : top;
            }
            int action = -1;                                        // std::max_element: the FIRST maximum in index order
            for (int st = 0; st < WS; ++st) {
                const uint64_t eq = wv::ballot(visits[st] == top);
                if (action < 0 && eq) action = st * 64 + wv::ctz64(eq);
            }
            if (rh.passN > top) action = WPASS;
            wv::sync();
            if (g.ply < P.max_plies) P.match_actions[(size_t)g.game_id * (size_t)P.max_plies + (size_t)g.ply] = (int16_t)action;
            const uint32_t ply1 = (uint32_t)g.ply + 1u;
            advance_root<G>(P, g, slot, lds, action);                  // act (:101)
            g.d_plies++;
            if (l == 0) {
                Mailbox* mb = P.mailbox + 2 * partner + (int)((g.game_id / (uint32_t)n) & 1u);
                mb->game = g.game_id;
                mb->action = (uint32_t)action;
                mb->rng_state = g.rng.state;
                mb->ply_launch = (uint64_t)ply1 | ((uint64_t)P.launch_seq << 32);
            }
            continue;
        }
        if (!ensure_nodes<G>(P, g, slot, lds, (uint32_t)P.max_batch + 2u)) break;
        select_batch<G>(P, g, slot, ctl, lds);
        ++round;
    }

    game_store<G>(ctl, g, lds);
    P.leaf_count[slot] = (g.status == ST_ACTIVE && P.eval_kind == EVAL_NETWORK) ? (uint32_t)g.n_leaves : 0u;
    if (g.status == ST_ACTIVE && wv::lane() == 0) wv::atomic_add_u32(&P.counters->active_slots, 1u);
}

}  // namespace sprlw

#endif  // SPRL_STEP_KERNEL_WIDE_H
