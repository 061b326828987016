// step_kernel.h — the self-play hot path: one 64-lane wavefront owns one game and its UCT tree.
//
// One call of step_game<G>() advances one game slot by up to `rounds` search rounds.  A round is the
// reference's inner loop body (selfplay/SelfPlay.hpp:100-108):
//     finish   = UCTTree::evaluateAndBackpropLeaves  (uct/UCTTree.hpp:124-184)  for the leaves queued last round
//     move     = visits -> tempered pdf -> sample -> re-root  (SelfPlay.hpp:111-148) once >= numTraversals
//     select   = UCTTree::searchAndGetLeaves          (uct/UCTTree.hpp:76-114) : <= maxBatch traversals,
//                terminal / gray leaves backed up immediately, empty leaves queued (<= maxQueue)
// With the in-kernel evaluators (RandomNetwork / OthelloHeuristic) rounds chain inside one launch; with the
// network evaluator a launch ends after `select` has written the symmetrised input planes of the queued leaves,
// LibTorch-ROCm runs the CNN on the dense batch, and the next launch starts with `finish`.
//
// Lane mapping: lane a <-> action a (Othello: cell a; pass is a scalar edge in the node header).  Per level
// of a descent each lane loads N[a], W[a], P[a], child[a] (4 coalesced rows), the PUCT score is computed per
// lane with IEEE fp32 ops in the reference's order (UCTNode.hpp:200,210,236; no FMA contraction), the arg-max
// is a wave butterfly, the tie set is a ballot, and the tie is broken with the reference's UniformInt draw.
//
// Coherence rule: a tree is touched by exactly one wavefront, whose vector memory operations are issued and
// performed in program order through one L1, so read-after-write inside the wave needs no cache maintenance.
// Most addresses are written by the lane(s) that later read them (row element a by lane a, header fields by
// all lanes with a uniform value); the cross-lane cases (backup scatter, arena compaction) are followed by a
// workgroup-scope fence (= s_waitcnt vmcnt(0)).  Arenas that change owner cross XCDs: agent-scope release/acquire.
#ifndef SPRL_STEP_KERNEL_H
#define SPRL_STEP_KERNEL_H

#include "dev_rng.h"
#include "engine_types.h"
#include "games.h"
#include "wave.h"

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace sprl {

SPRL_DEV uint8_t* node_at(uint8_t* abase, uint32_t idx) { return abase + (size_t)idx * SPRL_NODE_BYTES; }
SPRL_DEV float* rowN(uint8_t* n) { return (float*)n; }
SPRL_DEV float* rowW(uint8_t* n) { return (float*)(n + 256); }
SPRL_DEV float* rowP(uint8_t* n) { return (float*)(n + 512); }
SPRL_DEV uint16_t* rowC(uint8_t* n) { return (uint16_t*)(n + 768); }
SPRL_DEV NodeHdr* hdr_of(uint8_t* n) { return (NodeHdr*)(n + 896); }
SPRL_DEV uint8_t* rowCH(uint8_t* n) { return n + 960; }     // bits 16..23 of the child indices (arenas above 65535 nodes)
// this lane's child index: 16 bits, or 24 with the high-byte row when the arena is larger than 65535 nodes (wave-uniform branch)
SPRL_DEV uint32_t load_child(const EngineParams& P, uint8_t* np) {
    uint32_t c = rowC(np)[wv::lane()];
    if (P.wide_idx) c |= (uint32_t)rowCH(np)[wv::lane()] << 16;
    return c;
}
SPRL_DEV void store_child(const EngineParams& P, uint8_t* np, uint32_t c) {
    rowC(np)[wv::lane()] = (uint16_t)c;
    if (P.wide_idx) rowCH(np)[wv::lane()] = (uint8_t)(c >> 16);
}
// header + this lane's row elements in one batch of loads (the rows of a non-active node are never used)
SPRL_DEV void load_node(const EngineParams& P, uint8_t* np, NodeHdr& h, float& n, float& w, float& p, uint32_t& ch) {
    const int l = wv::lane();
    h = *hdr_of(np);
    n = rowN(np)[l];
    w = rowW(np)[l];
    p = rowP(np)[l];
    ch = load_child(P, np);
    wv::sync();
}
// wave-uniform copy of a node header; every lane has loaded it before any lane may go on to modify it
SPRL_DEV NodeHdr load_hdr(uint8_t* n) {
    NodeHdr h = *hdr_of(n);
    wv::sync();
    return h;
}

#define PASS_A (G::A - 1)      // index of the pass action in games that have one (Othello 64, Go 49)

// per-wavefront LDS: the descent path, and for Go the positions of all ancestors (game start .. current node)
// for the positional-superko test plus the 8-ply histories of the leaves queued for the network
template <class G>
struct WaveLds {
    uint32_t path[G::MAX_DEPTH];
    uint32_t fcache[SPRL_FCACHE];             // recycled node ids ready for reuse (GameCtl::fcache while the slot runs)
    uint64_t hist[G::HIST_CAP][2];
    uint64_t leaf_hist[SPRL_MAXQ][G::HIST][2];
    uint32_t leaf_size[SPRL_MAXQ];
};

struct Game {                 // per-wave working state (wave-uniform values)
    Pcg32 rng;
    uint8_t* abase;
    uint32_t arena, root, n_alloc, epoch, root_player, game_id, status;
    uint32_t rstack_n, fc_n;     // reclaim stack height, ready recycled ids (node recycling)
    float rootN, rootW;
    int ply, traversals, n_leaves;
    // per-launch counter deltas (32-bit: keeps the wave's scalar registers free for the search itself); they are
    // added to the 64-bit totals in GameCtl once, when the slot's state is written back
    uint32_t d_traversals, d_levels, d_expansions, d_nn_evals, d_terminal, d_gray, d_dup, d_created, d_compactions,
        d_games, d_plies, d_recycled, hi_alloc;
#if defined(SPRL_PHASE_TIMERS) && !defined(SPRL_EMU)
    unsigned long long cyc_finish, cyc_move, cyc_select, cyc_create, cyc_backup, cyc_leafio, cyc_noise, cyc_lvl_wait,
        cyc_lvl_pick, cyc_lvl_desc;
#endif
};

SPRL_DEV void raise_error(const EngineParams& P, Game& g, uint32_t code) {
    if (wv::lane() == 0) {
        if (wv::atomic_cas_u32(&P.counters->error, 0u, code) == 0u) P.counters->error_game = g.game_id;
    }
    g.status = ST_ERROR;
}

// ---------------------------------------------------------------------------------------------------
// node creation (UCTNode::getAddChild -> GameNode::getAddChild -> getNextNodeImpl)
// ---------------------------------------------------------------------------------------------------
template <class G>
SPRL_DEV void write_new_node(const EngineParams& P, uint8_t* np, const Pos& s, int action) {
    NodeHdr* h = hdr_of(np);
    h->p0 = s.p0;
    h->p1 = s.p1;
    h->legal = s.legal;
    h->value = 0.0f;
    h->exp_epoch = 0;
    h->passN = 0.0f;
    h->passW = 0.0f;
    h->passP = 0.0f;
    h->passChild = P.none_idx;
    h->player = s.player;
    h->flags = (uint8_t)((s.terminal ? F_TERMINAL : 0) | (s.pass_legal ? F_PASS : 0));
    h->winner = s.winner;
    h->action = (uint16_t)action;
    h->depth = s.depth;
    store_child(P, np, P.none_idx);
}

// ---------------------------------------------------------------------------------------------------
// node recycling.  The reference frees the siblings of the move played (UCTNode::pruneChildrenExcept,
// uct/UCTNode.hpp:356-366; GameNode.hpp:113-122).  Here the old decision node - its edge to the kept child cut - goes on
// the game's reclaim stack; a refill pops a few ids, pushes their children in their place (one coalesced child-row read
// each, all requested together) and leaves the popped ids in a small cache the allocator takes from before it bumps the
// arena.  So a game's arena holds its live subtree plus the garbage not yet reused, not the whole game's nodes.
// ---------------------------------------------------------------------------------------------------
template <class G>
SPRL_DEV uint32_t alloc_node(Game& g, WaveLds<G>* lds) {
    if (g.fc_n > 0) {
        g.d_recycled++;
        return lds->fcache[--g.fc_n];
    }
    const uint32_t c = g.n_alloc++;
    if (g.n_alloc > g.hi_alloc) g.hi_alloc = g.n_alloc;
    return c;
}

template <class G>
SPRL_DEV void reclaim_push(const EngineParams& P, Game& g, int slot, uint32_t node) {
    if (!P.recycle) return;
    uint32_t* stack = P.reclaim + (size_t)slot * (size_t)P.node_cap;
    if (wv::lane() == 0) stack[g.rstack_n] = node;
    g.rstack_n += 1;
    wv::wave_fence();
}

template <class G>
SPRL_DEV_NOINLINE void reclaim_refill(const EngineParams& P, Game& g, int slot, WaveLds<G>* lds) {
    const int l = wv::lane();
    uint32_t* stack = P.reclaim + (size_t)slot * (size_t)P.node_cap;
    while (g.fc_n < SPRL_FCACHE && g.rstack_n > 0) {
        // up to 8 ids per pass: their child rows are requested together (one memory round trip per pass)
        int k = SPRL_FCACHE - (int)g.fc_n;
        if (k > 8) k = 8;
        if (k > (int)g.rstack_n) k = (int)g.rstack_n;
        const uint32_t base = g.rstack_n - (uint32_t)k;
        const uint32_t mine = l < k ? stack[base + l] : 0u;
        wv::sync();
        g.rstack_n = base;
        uint32_t id[8], ch[8], pc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            id[j] = wv::bcast_u32(mine, j < k ? j : 0);
            uint8_t* np = node_at(g.abase, id[j]);
            ch[j] = j < k ? load_child(P, np) : P.none_idx;
            pc[j] = (G::HAS_PASS && j < k) ? hdr_of(np)->passChild : P.none_idx;
        }
        wv::sync();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j >= k) break;
            const bool has = ch[j] != P.none_idx;
            const uint64_t mask = wv::ballot(has);
            if (has) stack[g.rstack_n + (uint32_t)wv::popc64(mask & wv::lt_mask(l))] = ch[j];
            g.rstack_n += (uint32_t)wv::popc64(mask);
            if (G::HAS_PASS && pc[j] != P.none_idx) {
                if (l == 0) stack[g.rstack_n] = pc[j];
                g.rstack_n += 1;
            }
            if (l == 0) lds->fcache[g.fc_n] = id[j];
            g.fc_n += 1;
        }
        wv::wave_fence();
    }
}

template <class G>
SPRL_DEV Pos pos_of(const NodeHdr& h) {
    Pos s;
    s.p0 = h.p0;
    s.p1 = h.p1;
    s.legal = h.legal;
    s.player = h.player;
    s.pass_legal = (h.flags & F_PASS) ? 1 : 0;
    s.terminal = (h.flags & F_TERMINAL) ? 1 : 0;
    s.winner = h.winner;
    s.last_pass = G::HAS_PASS && h.action == PASS_A && h.depth > 0;
    s.depth = h.depth;
    return s;
}

// ---------------------------------------------------------------------------------------------------
// expand (UCTNode::expand, uct/UCTNode.hpp:312-348).  The P row already holds the cached, legal-masked
// network policy; N and W rows are (re)initialised here, which is what EdgeStatistics::reset() left behind.
// ---------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------
// Go: legal mask of a position, lane <-> point (GoNode::checkLegalPlacement / computeActionMask,
// games/GoNode.cpp:178-228,292-301).  Every stone lane flood-fills its own group as a bit mask (all lanes in
// lockstep, wave-uniform trip count), liberties are one popcount, an empty point then looks at its four
// neighbours' (colour, liberties, group mask) through cross-lane reads, builds the position that placing there
// would give (captures = adjacent enemy groups in atari) and compares it with every ancestor position kept in
// LDS — an exact positional-superko test where the reference compares 64-bit Zobrist hashes.
// ---------------------------------------------------------------------------------------------------
template <class G>
SPRL_DEV uint64_t go_legal_mask(const Pos& c, const uint64_t (*hist)[2], int n_hist) {
    const int l = wv::lane();
    const uint64_t own = c.player ? c.p1 : c.p0, opp = c.player ? c.p0 : c.p1;
    const uint64_t empty = ~(own | opp) & G::BOARD;
    const uint64_t bit = l < G::CELLS ? (1ull << l) : 0ull;
    const bool is_own = (own & bit) != 0, is_opp = (opp & bit) != 0, is_empty = (empty & bit) != 0;
    const uint64_t within = is_own ? own : (is_opp ? opp : 0ull);
    uint64_t gm = bit & within;
    for (;;) {
        const uint64_t n = (gm | G::dilate(gm)) & within;
        const bool changed = n != gm;
        gm = n;
        if (wv::ballot(changed) == 0) break;
    }
    const uint32_t libs = (uint32_t)wv::popc64(G::dilate(gm) & empty);
    const int row = l / G::COLS, col = l % G::COLS;
    bool has_libs = false;
    uint64_t cap = 0;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const bool valid = l < G::CELLS && (d == 0 ? row > 0 : d == 1 ? col > 0 : d == 2 ? row < G::ROWS - 1 : col < G::COLS - 1);
        const int nb = valid ? (d == 0 ? l - G::COLS : d == 1 ? l - 1 : d == 2 ? l + G::COLS : l + 1) : l;
        const uint32_t nlibs = wv::shfl_u32(libs, nb);
        const uint32_t glo = wv::shfl_u32((uint32_t)gm, nb), ghi = wv::shfl_u32((uint32_t)(gm >> 32), nb);
        if (valid) {
            const uint64_t nbit = 1ull << nb;
            if (empty & nbit) has_libs = true;
            else if (own & nbit) { if (nlibs > 1) has_libs = true; }
            else if (nlibs == 1) { has_libs = true; cap |= ((uint64_t)ghi << 32) | glo; }
        }
    }
    const uint64_t nown = own | bit, nopp = opp & ~cap;
    const uint64_t np0 = c.player ? nopp : nown, np1 = c.player ? nown : nopp;
    bool repeat = false;
    for (int i = 0; i < n_hist; ++i) repeat |= (hist[i][0] == np0) && (hist[i][1] == np1);
    return wv::ballot(is_empty && has_libs && !repeat);
}

// GameNode::getAddChild -> getNextNodeImpl for every game; `depth` = tree depth of the parent
template <class G>
SPRL_DEV void make_child(const EngineParams& P, Game& g, WaveLds<G>* lds, const Pos& parent, int action, int at,
                         Pos& cs) {
    // `at` = the child's index among the ancestor positions (game ply of the root + tree depth of the child)
    if constexpr (G::ID == SPRL_GAME_GO7) {
        G::apply(parent, action, cs);
        if (wv::lane() == 0) {
            lds->hist[at][0] = cs.p0;
            lds->hist[at][1] = cs.p1;
        }
        wv::sync();
        if (!cs.terminal) cs.legal = go_legal_mask<G>(cs, lds->hist, at + 1);
    } else {
        G::child(parent, action, cs);
    }
}

// Dirichlet root noise (UCTNode.hpp:330-347): rare (once per move) and heavy (serial gamma draws in double
// arithmetic), so it is kept out of line to keep the descent loop's code and register footprint small.
template <class G>
SPRL_DEV_NOINLINE void mix_root_noise(const EngineParams& P, Game& g, uint8_t* np, uint64_t legal, bool pass_legal,
                                      float pass_p) {
    const int l = wv::lane();
    NodeHdr* h = hdr_of(np);
    SPRL_TIC(t_nz);
    const int num_legal = wv::popc64(legal) + (pass_legal ? 1 : 0);
    const bool mine = (legal >> l) & 1ull;
    const int my_rank = wv::popc64(legal & wv::lt_mask(l));
    NormalState ns = { 0.0f, 0 };
    float my_noise = 0.0f, pass_noise = 0.0f, sum = 0.0f;
    for (int i = 0; i < num_legal; ++i) {                 // Random::Dirichlet, utils/random.cpp:61-74
        float s = rng_gamma(g.rng, ns, P.dir_alpha);
        sum += s;
        if (mine && my_rank == i) my_noise = s;
        if (i == num_legal - 1 && pass_legal) pass_noise = s;
    }
    const float norm = 1.0f / sum;
    my_noise *= norm;
    pass_noise *= norm;
    const double keep = 1.0 - (double)P.dir_eps;          // UCTNode.hpp:341-343 (double arithmetic)
    if (mine) {
        float p = rowP(np)[l];
        rowP(np)[l] = (float)(keep * (double)p + (double)(P.dir_eps * my_noise));
    }
    if (pass_legal) h->passP = (float)(keep * (double)pass_p + (double)(P.dir_eps * pass_noise));
    SPRL_TOC(g.cyc_noise, t_nz);
}

template <class G>
SPRL_DEV void expand_node(const EngineParams& P, Game& g, uint8_t* np, const NodeHdr& hc, bool add_noise) {
    const int l = wv::lane();
    NodeHdr* h = hdr_of(np);
    const bool pass_legal = G::HAS_PASS && (hc.flags & F_PASS);
    rowN(np)[l] = 0.0f;
    rowW(np)[l] = 0.0f;
    h->passN = 0.0f;
    h->passW = 0.0f;
    h->exp_epoch = g.epoch;
    if (add_noise) mix_root_noise<G>(P, g, np, hc.legal, pass_legal, hc.passP);
    g.d_expansions++;
}

// ---------------------------------------------------------------------------------------------------
// backup (UCTTree::backup, uct/UCTTree.hpp:261-273) over a stored path of `depth` edges
// ---------------------------------------------------------------------------------------------------
template <class G>
SPRL_DEV void backup_path(Game& g, uint32_t my_entry0, uint32_t my_entry1, int depth, int leaf_player, float value) {
    // Lane j owns path edge j (and j + 64): all W loads go out together, then all stores — two memory round
    // trips per backup instead of one per level.  The edges of one path are distinct addresses.
    const int l = wv::lane();
    SPRL_TIC(t_bk);
    wv::sync();         // (emulator) the descent's own stores to these edges precede the read-modify-write below
    const float est = -value * (leaf_player == 0 ? 1.0f : -1.0f);
    g.rootW += 1.0f + est * (g.root_player == 0 ? 1.0f : -1.0f);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int j = l + 64 * half;
        if (half == 1 && depth <= 64) break;
        const uint32_t e = half == 0 ? my_entry0 : my_entry1;
        if (j < depth) {
            const uint32_t node = e >> 8;
            const int a = (int)(e & 0xffu);
            const int child_player = (int)(g.root_player ^ ((uint32_t)(j + 1) & 1u));
            const float add = 1.0f + est * (child_player == 0 ? 1.0f : -1.0f);
            uint8_t* np = node_at(g.abase, node);
            float* wp = (G::HAS_PASS && a == PASS_A) ? &hdr_of(np)->passW : &rowW(np)[a];
            *wp = *wp + add;
        }
    }
    wv::wave_fence();   // edges were updated by lane j, they are read back by lane `action` / as header fields
    SPRL_TOC(g.cyc_backup, t_bk);
}

// ---------------------------------------------------------------------------------------------------
// evaluators: produce the cached policy (into the P row / passP, masked by legality) and value of a leaf
// (UCTTree.hpp:136-175 + networks/{RandomNetwork,OthelloHeuristic,GridNetwork}).  `sym` is the symmetry the
// state was presented in; the mask is deliberately NOT symmetrised in MASK_REFERENCE mode.
// ---------------------------------------------------------------------------------------------------
template <class G>
SPRL_DEV void evaluate_leaf(const EngineParams& P, Game& g, uint8_t* np, NodeHdr& hc, int sym,
                           float my_logit, float pass_logit, float nn_value) {
    const int l = wv::lane();
    NodeHdr* h = hdr_of(np);
    const uint64_t legal = hc.legal;
    const bool pass_legal = G::HAS_PASS && (hc.flags & F_PASS);
    // mask as the evaluator sees it, indexed in the symmetrised frame
    uint64_t used;
    if (P.mask_frame == MASK_SYMMETRISED && sym != 0) {
        const int src = l < G::NA ? G::map_action(G::inverse_sym(sym), l) : 0;
        used = wv::ballot(l < G::NA && ((legal >> src) & 1ull));
    } else {
        used = legal;
    }
    const bool mine = l < G::NA && ((used >> l) & 1ull);
    const int num_legal = wv::popc64(used) + (pass_legal ? 1 : 0);
    float pol, pass_pol = 0.0f, value = 0.0f;
    if (P.eval_kind == EVAL_NETWORK) {
        float e = mine ? sprl_expf(my_logit) : 0.0f;                        // GridNetwork.hpp:110-121
        float e_pass = pass_legal ? sprl_expf(pass_logit) : 0.0f;
        float sum = 0.0f;                                                   // GameActionDist::sum, index order
        for (uint64_t m = used; m; m &= m - 1) sum += wv::bcast_f32(e, wv::ctz64(m));
        if (pass_legal) sum += e_pass;
        if (sum == 0.0f) {
            float uniform = 1.0f / (float)num_legal;                        // GridNetwork.hpp:124-129
            pol = mine ? uniform : 0.0f;
            pass_pol = pass_legal ? uniform : 0.0f;
        } else {
            float inv = 1.0f / sum;                                         // GameActionDist.hpp:284-289
            pol = e * inv;
            pass_pol = e_pass * inv;
        }
        value = nn_value;
    } else {
        float uniform = 1.0f / (float)num_legal;                            // RandomNetwork.hpp:29-43
        pol = mine ? uniform : 0.0f;
        pass_pol = pass_legal ? uniform : 0.0f;
        if (P.eval_kind == EVAL_HEURISTIC && G::ID == SPRL_GAME_OTHELLO) {  // OthelloHeuristic.cpp:28-49
            uint64_t own = hc.player ? hc.p1 : hc.p0, opp = hc.player ? hc.p0 : hc.p1;
            int num_opp = wv::popc64(Othello::legal_moves(opp, own));
            int num_empty = 64 - wv::popc64(own | opp);
            value = (float)(num_legal - num_opp) / (float)num_empty;
        }
    }
    // undo the symmetry: policy_orig[a] = policy_sym[map_s(a)] (UCTTree.hpp:162-164)
    const int from = l < G::NA ? G::map_action(sym, l) : 0;
    float pol_orig = wv::shfl_f32(pol, from);
    const bool legal_here = l < G::NA && ((legal >> l) & 1ull);
    rowP(np)[l] = legal_here ? pol_orig : 0.0f;                             // UCTNode.hpp:320-327
    hc.passP = pass_legal ? pass_pol : 0.0f;
    hc.value = value;
    hc.flags = (uint8_t)(hc.flags | F_EVAL);
    h->passP = hc.passP;
    h->value = hc.value;
    h->flags = hc.flags;
}

// symmetrised input planes of a queued leaf (GridNetwork.hpp:72-97 after D4GridSymmetrizer.hpp:52-75): plane 2t / 2t+1 =
// stones of the side to move / the opponent t plies ago (t < size, else zero), last plane = colour to move
template <class G>
SPRL_DEV void encode_leaf(const EngineParams& P, const WaveLds<G>* lds, int q, int player, int sym, int nn_slot) {
    const int l = wv::lane();
    if (l < G::CELLS) {
        const int src = G::map_cell(G::inverse_sym(sym), l);      // out[map_s(i)] = in[i]
        float* out = P.nn_in + (size_t)nn_slot * (G::PLANES * G::CELLS);
        const int size = (int)lds->leaf_size[q];
#pragma unroll
        for (int t = 0; t < G::HIST; ++t) {
            const uint64_t p0 = lds->leaf_hist[q][t][0], p1 = lds->leaf_hist[q][t][1];
            const uint64_t own = player ? p1 : p0, opp = player ? p0 : p1;
            out[(2 * t) * G::CELLS + l] = t < size ? (float)((own >> src) & 1ull) : 0.0f;
            out[(2 * t + 1) * G::CELLS + l] = t < size ? (float)((opp >> src) & 1ull) : 0.0f;
        }
        out[(2 * G::HIST) * G::CELLS + l] = player == 0 ? 1.0f : 0.0f;
    }
}

// ---------------------------------------------------------------------------------------------------
// finish: UCTTree::evaluateAndBackpropLeaves for the leaves queued by the previous select
// ---------------------------------------------------------------------------------------------------
template <class G>
SPRL_DEV void finish_leaves(const EngineParams& P, Game& g, int slot, GameCtl* ctl) {
    const int l = wv::lane();
    const int nn_base = (int)P.leaf_offset[slot];     // row of this slot's first leaf in the dense network batch
    for (int q = 0; q < g.n_leaves; ++q) {
        const uint32_t leaf = ctl->leaf_node[q];
        const int depth = (int)ctl->leaf_depth[q];
        const int sym = (int)ctl->leaf_sym[q];
        uint8_t* np = node_at(g.abase, leaf);
        // everything this leaf needs from memory is requested up front: one round trip
        NodeHdr h = *hdr_of(np);
        const uint32_t* path = P.paths + ((size_t)slot * SPRL_MAXQ + q) * P.max_depth;
        const uint32_t e0 = l < depth ? path[l] : 0u;
        const uint32_t e1 = 64 + l < depth ? path[64 + l] : 0u;
        float my_logit = 0.0f, pass_logit = 0.0f, nn_value = 0.0f;
        if (P.eval_kind == EVAL_NETWORK) {
            const float* logits = P.nn_logits + (size_t)(nn_base + q) * G::A;
            if (l < G::NA) my_logit = logits[l];
            if (G::HAS_PASS) pass_logit = logits[G::A - 1];
            nn_value = P.nn_value[nn_base + q];
        }
        wv::sync();
        if (!(h.flags & F_EVAL)) {
            evaluate_leaf<G>(P, g, np, h, sym, my_logit, pass_logit, nn_value);
        } else {
            g.d_dup++;
        }
        if (h.exp_epoch != g.epoch) expand_node<G>(P, g, np, h, P.add_noise && leaf == g.root);
        backup_path<G>(g, e0, e1, depth, h.player, h.value);
    }
    g.n_leaves = 0;
}

// ---------------------------------------------------------------------------------------------------
// select: UCTTree::searchAndGetLeaves
// ---------------------------------------------------------------------------------------------------
template <class G>
SPRL_DEV void select_batch(const EngineParams& P, Game& g, int slot, GameCtl* ctl, WaveLds<G>* lds) {
    uint32_t* lds_path = lds->path;
    const int l = wv::lane();
    int trav = 0;
    while (trav < P.max_batch) {
        ++trav;
        // ---- selectLeaf (UCTTree.hpp:225-249) ----
        uint32_t cur = g.root;
        float nself = g.rootN;
        g.rootN += 1.0f;                      // virtual loss on the decision node itself (Q5/Q6)
        g.rootW -= 1.0f;
        int depth = 0;
        // header and the four rows of a node are fetched together: one memory round trip per level
        NodeHdr h;
        float n, w, p;
        uint32_t ch;
        load_node(P, node_at(g.abase, cur), h, n, w, p, ch);
        if (G::HIST_CAP > 1 && l == 0) {                   // ancestors: entry ply + d = node at tree depth d
            lds->hist[g.ply][0] = h.p0;
            lds->hist[g.ply][1] = h.p1;
        }
        while (h.exp_epoch == g.epoch && !(h.flags & F_TERMINAL)) {
            uint8_t* np = node_at(g.abase, cur);
#if defined(SPRL_PHASE_TIMERS) && !defined(SPRL_EMU)
            const unsigned long long t_l0 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // diagnostic build: make the memory wait visible
            const unsigned long long t_l1 = __builtin_amdgcn_s_memtime();
            g.cyc_lvl_wait += t_l1 - t_l0;
#endif
            int a;
            if (G::HAS_PASS && G::PASS_EXCLUSIVE && h.legal == 0) {   // pass is the only legal action (mask[64] only)
                (void)rng_uniform_int(g.rng, 1u);           // bestAction still draws (UCTNode.hpp:250)
                a = PASS_A;
            } else if (G::HAS_PASS && !G::PASS_EXCLUSIVE) {  // Go: pass (last index) competes with the placements
                const bool lg = l < G::NA && ((h.legal >> l) & 1ull);
                const float sq = __builtin_sqrtf(nself);
                const float den = 1.0f + n;
                const float score = w / den + P.u_weight * (p * sq / den);
                const float pden = 1.0f + h.passN;
                const float pscore = h.passW / pden + P.u_weight * (h.passP * sq / pden);
                float best = wv::fmax_all(lg ? score : -__builtin_inff());
                best = pscore > best ? pscore : best;
                const uint64_t ties = wv::ballot(lg && score == best);
                const int kb = wv::popc64(ties);
                const int k = kb + (pscore == best ? 1 : 0);
                const int r = rng_uniform_int(g.rng, (uint32_t)k);
                a = r < kb ? wv::nth_set_bit(ties, r) : PASS_A;
            } else {
                const bool lg = l < G::NA && ((h.legal >> l) & 1ull);
                const float sq = __builtin_sqrtf(nself);
                const float den = 1.0f + n;
                const float q = w / den;                    // UCTNode.hpp:200
                const float u = p * sq / den;               // UCTNode.hpp:210
                const float score = q + P.u_weight * u;     // UCTNode.hpp:236
                const float best = wv::fmax_all(lg ? score : -__builtin_inff());
                const uint64_t ties = wv::ballot(lg && score == best);
                const int k = wv::popc64(ties);
                const int r = rng_uniform_int(g.rng, (uint32_t)k);
                a = wv::nth_set_bit(ties, r);
            }
            g.d_levels++;
#if defined(SPRL_PHASE_TIMERS) && !defined(SPRL_EMU)
            const unsigned long long t_l2 = __builtin_amdgcn_s_memtime();
            g.cyc_lvl_pick += t_l2 - t_l1;
#endif
            if (depth >= P.max_depth) { raise_error(P, g, ERR_MAX_DEPTH); return; }
            if (l == 0) lds_path[depth] = (cur << 8) | (uint32_t)a;
            ++depth;
            // ---- getAddChild + the child's virtual loss ----
            NodeHdr* hp = hdr_of(np);
            uint32_t c;
            float n_a, w_a;
            if (G::HAS_PASS && a == PASS_A) {
                c = h.passChild;
                n_a = h.passN;
                w_a = h.passW;
            } else {
                c = wv::bcast_u32(ch, a);
                n_a = wv::bcast_f32(n, a);
                w_a = wv::bcast_f32(w, a);
            }
            bool created = false;
            Pos cs;
            if (c == P.none_idx) {
                SPRL_TIC(t_cr);
                c = alloc_node<G>(g, lds);
                created = true;
                make_child<G>(P, g, lds, pos_of<G>(h), a, g.ply + depth, cs);
                write_new_node<G>(P, node_at(g.abase, c), cs, a);
                SPRL_TOC(g.cyc_create, t_cr);
                w_a = P.init_q_zero ? 0.0f : h.value;       // InitQ::PARENT / ZERO (UCTNode.hpp:267-273)
                g.d_created++;
                if (G::HAS_PASS && a == PASS_A) hp->passChild = c;
                else if (l == a) store_child(P, np, c);
            }
            if (G::HAS_PASS && a == PASS_A) {
                hp->passN = n_a + 1.0f;
                hp->passW = w_a - 1.0f;
            } else if (l == a) {
                rowN(np)[l] = n_a + 1.0f;
                rowW(np)[l] = w_a - 1.0f;
            }
            nself = n_a;
            cur = c;
            if (created) {
                // a freshly created node is empty: header known without reloading it
                h.p0 = cs.p0; h.p1 = cs.p1; h.legal = cs.legal; h.value = 0.0f; h.exp_epoch = 0;
                h.passChild = P.none_idx; h.player = cs.player; h.winner = cs.winner;
                h.action = (uint16_t)a; h.depth = cs.depth;
                h.flags = (uint8_t)((cs.terminal ? F_TERMINAL : 0) | (cs.pass_legal ? F_PASS : 0));
                break;
            }
            load_node(P, node_at(g.abase, cur), h, n, w, p, ch);
#if defined(SPRL_PHASE_TIMERS) && !defined(SPRL_EMU)
            g.cyc_lvl_desc += __builtin_amdgcn_s_memtime() - t_l2;
#endif
            if (G::HIST_CAP > 1 && l == 0) {
                lds->hist[g.ply + depth][0] = h.p0;
                lds->hist[g.ply + depth][1] = h.p1;
            }
        }
        // ---- leaf handling (UCTTree.hpp:87-110) ----
        if (h.flags & F_TERMINAL) {
            float value = h.winner < 0 ? 0.0f : (h.winner == (int8_t)h.player ? 1.0f : -1.0f);
            uint32_t e0 = l < depth ? lds_path[l] : 0u;
            uint32_t e1 = 64 + l < depth ? lds_path[64 + l] : 0u;
            backup_path<G>(g, e0, e1, depth, h.player, value);
            g.d_terminal++;
            continue;
        } else if (h.flags & F_EVAL) {
            expand_node<G>(P, g, node_at(g.abase, cur), h, P.add_noise && cur == g.root);
            uint32_t e0 = l < depth ? lds_path[l] : 0u;
            uint32_t e1 = 64 + l < depth ? lds_path[64 + l] : 0u;
            backup_path<G>(g, e0, e1, depth, h.player, h.value);
            g.d_gray++;
            continue;
        } else {
            const int q = g.n_leaves++;
            ctl->leaf_node[q] = cur;
            ctl->leaf_depth[q] = (uint32_t)depth;
            uint32_t* path = P.paths + ((size_t)slot * SPRL_MAXQ + q) * P.max_depth;
            if (l < depth) path[l] = lds_path[l];
            if (64 + l < depth) path[64 + l] = lds_path[64 + l];
            // the leaf's own position and up to HIST-1 ancestors, newest first (GoNode.cpp:385-398)
            const int last = g.ply + depth;
            const int size = G::HIST_CAP > 1 ? (last + 1 < G::HIST ? last + 1 : G::HIST) : 1;
            wv::sync();
            if (l < size) {
                lds->leaf_hist[q][l][0] = G::HIST_CAP > 1 ? lds->hist[last - l][0] : h.p0;
                lds->leaf_hist[q][l][1] = G::HIST_CAP > 1 ? lds->hist[last - l][1] : h.p1;
            }
            if (l == 0) lds->leaf_size[q] = (uint32_t)size;
            ctl->leaf_player[q] = h.player;
        }
        if (g.n_leaves >= P.max_queue) break;
    }
    g.traversals += trav;
    g.d_traversals += (uint32_t)trav;
    // symmetry draws in queue order (UCTTree.hpp:141-149), then the input planes for the network
    SPRL_TIC(t_io);
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
    SPRL_TOC(g.cyc_leafio, t_io);
}

// ---------------------------------------------------------------------------------------------------
// arena compaction (rare): Cheney copy of the subtree under the decision node into a spare arena
// ---------------------------------------------------------------------------------------------------
SPRL_DEV void copy_node(const uint8_t* src, uint8_t* dst) {
    struct alignas(16) Chunk { uint32_t x[4]; };
    ((Chunk*)dst)[wv::lane()] = ((const Chunk*)src)[wv::lane()];
}

template <class G>
SPRL_DEV_NOINLINE bool compact_arena(const EngineParams& P, Game& g) {
    const int l = wv::lane();
    // acquire a spare arena
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
    wv::agent_acquire();                      // the arena's previous owner may have run on another XCD
    uint8_t* from = g.abase;
    uint8_t* to = P.arenas + (size_t)got * (size_t)P.node_cap * SPRL_NODE_BYTES;
    copy_node(node_at(from, g.root), node_at(to, 0));
    wv::wave_fence();
    uint32_t scan = 0, free_ = 1;
    while (scan < free_) {
        uint8_t* np = node_at(to, scan);
        uint32_t ch = load_child(P, np);
        const bool has = ch != P.none_idx;
        const uint64_t mask = wv::ballot(has);
        wv::sync();
        if (has) store_child(P, np, free_ + (uint32_t)wv::popc64(mask & wv::lt_mask(l)));
        uint32_t k = 0;
        for (uint64_t m = mask; m; m &= m - 1, ++k) {
            uint32_t src = wv::bcast_u32(ch, wv::ctz64(m));
            copy_node(node_at(from, src), node_at(to, free_ + k));
        }
        free_ += k;
        if (G::HAS_PASS) {
            NodeHdr* h = hdr_of(np);
            const uint32_t pc = h->passChild;
            wv::sync();
            if (pc != P.none_idx) {
                copy_node(node_at(from, pc), node_at(to, free_));
                h->passChild = free_;
                ++free_;
            }
        }
        wv::wave_fence();
        ++scan;
    }
    wv::agent_release();                      // no dirty line of the old arena may outlive its release
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

// `need` nodes must be available before a search batch (every traversal creates at most one): recycled ids first, then
// the untouched rest of the arena, then - rarely - a compaction into a spare arena
template <class G>
SPRL_DEV bool ensure_nodes(const EngineParams& P, Game& g, int slot, WaveLds<G>* lds, uint32_t need) {
    if (P.recycle && g.fc_n < need && g.rstack_n > 0) reclaim_refill<G>(P, g, slot, lds);
    if (g.fc_n + ((uint32_t)P.node_cap - g.n_alloc) >= need) return true;
    if (!compact_arena<G>(P, g)) return false;
    if ((uint32_t)P.node_cap - g.n_alloc >= need) return true;
    raise_error(P, g, ERR_ARENA_FULL);
    return false;
}

// ---------------------------------------------------------------------------------------------------
// new game / move
// ---------------------------------------------------------------------------------------------------
template <class G>
SPRL_DEV void init_game(const EngineParams& P, Game& g, int slot, WaveLds<G>* lds, uint32_t gid) {
    if (gid >= (uint32_t)P.num_games) {
        g.status = ST_IDLE;
        return;
    }
    g.status = ST_ACTIVE;
    g.game_id = gid;
    rng_seed(g.rng, P.seed, P.stream_base + (int)gid);
    Pos s;
    G::start(s);
    write_new_node<G>(P, node_at(g.abase, P.alloc_base), s, 0);
    g.root = P.alloc_base;
    g.n_alloc = P.alloc_base + 1;
    g.rstack_n = 0;                          // a new game starts on an empty arena: the old tree is dropped whole
    g.fc_n = 0;
    g.epoch = 1;
    g.root_player = 0;
    g.rootN = 0.0f;                          // UCTTree::m_edgeStatistics slot 0 (UCTTree.hpp:301)
    g.rootW = 0.0f;
    g.ply = 0;
    g.traversals = 0;
    g.n_leaves = 0;
    g.d_created++;
    if (G::HIST_CAP > 1) {
        if (wv::lane() == 0) {
            lds->hist[0][0] = s.p0;
            lds->hist[0][1] = s.p1;
        }
        uint64_t* gh = P.hist_boards + (size_t)slot * G::HIST_CAP * 2;
        gh[0] = s.p0;
        gh[1] = s.p1;
    }
}

template <class G>
SPRL_DEV_NOINLINE void start_game(const EngineParams& P, Game& g, int slot, WaveLds<G>* lds) {
    uint32_t gid = 0;
    if (wv::lane() == 0) gid = wv::atomic_add_u32(&P.counters->next_game, 1u);
    gid = wv::bcast_u32(gid, 0);
    init_game<G>(P, g, slot, lds, gid);
}

// advanceDecision (UCTTree.hpp:197-210): O(1) — follow the edge, carry its stats as the new root's own N()/W() (Q6),
// bump the epoch (== clearSubtree), forget the siblings (== pruneChildrenExcept).  Also serves opponentAct
// (UCTNetworkAgent.hpp:106-108), where the decision node may never have been expanded and the child may not exist.
template <class G>
SPRL_DEV void advance_root(const EngineParams& P, Game& g, int slot, WaveLds<G>* lds, int action) {
    const int l = wv::lane();
    uint8_t* np = node_at(g.abase, g.root);
    const NodeHdr h = load_hdr(np);
    const bool expanded = h.exp_epoch == g.epoch;      // rows of a node that is not active hold no valid statistics
    uint32_t c;
    float n_a, w_a;
    if (G::HAS_PASS && action == PASS_A) {
        c = h.passChild;
        n_a = expanded ? h.passN : 0.0f;
        w_a = expanded ? h.passW : 0.0f;
    } else {
        c = wv::bcast_u32(load_child(P, np), action);
        n_a = wv::bcast_f32(rowN(np)[l], action);
        w_a = wv::bcast_f32(rowW(np)[l], action);
        if (!expanded) { n_a = 0.0f; w_a = 0.0f; }
    }
    wv::sync();
    if (c == P.none_idx) {                    // self-play never gets here (a sampled action has visits > 0)
        Pos cs;
        make_child<G>(P, g, lds, pos_of<G>(h), action, g.ply + 1, cs);
        c = alloc_node<G>(g, lds);
        write_new_node<G>(P, node_at(g.abase, c), cs, action);
        g.d_created++;
        w_a = (!P.init_q_zero && (h.flags & F_EVAL)) ? h.value : 0.0f;
    } else if (P.recycle) {                   // pruneChildrenExcept: cut the edge to the kept child, the rest is garbage
        if (G::HAS_PASS && action == PASS_A) hdr_of(np)->passChild = P.none_idx;
        else if (l == action) store_child(P, np, P.none_idx);
        wv::wave_fence();
    }
    if (P.recycle) reclaim_push<G>(P, g, slot, g.root);
    g.root = c;
    g.rootN = n_a;
    g.rootW = w_a;
    g.root_player ^= 1u;
    g.ply += 1;
    if (G::HIST_CAP > 1) {                     // the new decision node's position joins the real-game history
        const NodeHdr nh = load_hdr(node_at(g.abase, c));
        if (l == 0) {
            lds->hist[g.ply][0] = nh.p0;
            lds->hist[g.ply][1] = nh.p1;
        }
        uint64_t* gh = P.hist_boards + ((size_t)slot * G::HIST_CAP + (size_t)g.ply) * 2;
        gh[0] = nh.p0;
        gh[1] = nh.p1;
    }
    g.epoch += 1;
    g.traversals = 0;
}

// SelfPlay.hpp:110-148: visit pdf, temperature, CDF sample, record, re-root
template <class G>
SPRL_DEV_NOINLINE int play_move(const EngineParams& P, Game& g, int slot, WaveLds<G>* lds) {
    const int l = wv::lane();
    uint8_t* np = node_at(g.abase, g.root);
    const NodeHdr hcopy = load_hdr(np);
    const NodeHdr* h = &hcopy;
    const float visits = l < G::NA ? rowN(np)[l] : 0.0f;
    const float pass_visits = G::HAS_PASS ? h->passN : 0.0f;
    // the root row only holds visits on legal actions; everything else is exactly 0 and adds nothing
    const uint64_t nz = wv::ballot(visits != 0.0f);
    float sum = 0.0f;
    for (uint64_t m = nz; m; m &= m - 1) sum += wv::bcast_f32(visits, wv::ctz64(m));
    if (G::HAS_PASS) sum += pass_visits;
    float inv = 1.0f / sum;
    float pdf = visits * inv, pass_pdf = pass_visits * inv;
    const float ex = g.ply < P.early_cutoff ? P.early_exp : P.rest_exp;
    pdf = sprl_powf(pdf, ex);
    pass_pdf = sprl_powf(pass_pdf, ex);
    sum = 0.0f;
    for (uint64_t m = nz; m; m &= m - 1) sum += wv::bcast_f32(pdf, wv::ctz64(m));
    if (G::HAS_PASS) sum += pass_pdf;
    inv = 1.0f / sum;
    pdf = pdf * inv;
    pass_pdf = pass_pdf * inv;
    // cumsum in index order; lane a ends with the prefix over indices <= a
    float run = 0.0f, cdf = 0.0f;
    for (uint64_t m = nz; m; m &= m - 1) {
        int k = wv::ctz64(m);
        run += wv::bcast_f32(pdf, k);
        if (l >= k) cdf = run;
    }
    float last = G::HAS_PASS ? run + pass_pdf : wv::bcast_f32(cdf, G::A - 1);
    inv = 1.0f / last;
    cdf = cdf * inv;
    last = last * inv;
    // record (compact): board, mover, tempered pdf
    if (g.ply >= P.max_plies) { raise_error(P, g, ERR_MAX_PLIES); return 0; }
    const size_t rec = (size_t)g.game_id * (size_t)P.max_plies + (size_t)g.ply;
    P.rec_boards[rec * 2 + 0] = h->p0;
    P.rec_boards[rec * 2 + 1] = h->p1;
    P.rec_movers[rec] = h->player;
    if (l < G::NA) P.rec_pdf[rec * G::A + l] = pdf;
    if (G::HAS_PASS) P.rec_pdf[rec * G::A + (G::A - 1)] = pass_pdf;
    if (P.resign_threshold > 0.0f && g.ply >= P.resign_min_ply) {
        // extension, off in every parity configuration: the side to move resigns when the mean backed-up value of its
        // decision node, sum W / sum N over the edges (index order, like the oracle), is below -threshold.  The ply's
        // sample stays, no move is made; the caller ends the game for the opponent.
        const float wrow = l < G::NA ? rowW(np)[l] : 0.0f;
        float sn = 0.0f, sw = 0.0f;
        for (uint64_t m = nz; m; m &= m - 1) sn += wv::bcast_f32(visits, wv::ctz64(m));
        if (G::HAS_PASS) sn += pass_visits;
        for (uint64_t m = nz; m; m &= m - 1) sw += wv::bcast_f32(wrow, wv::ctz64(m));
        if (G::HAS_PASS) sw += h->passW;
        const float v = sw * (1.0f / sn);
        if (v < -P.resign_threshold) {
            g.d_plies++;
            return 2 - (int)h->player;              // winner + 1 = (1 - mover) + 1
        }
    }
    // Random::SampleCDF (utils/random.cpp:86-98)
    float e;
    do {
        e = rng_uniform_float(g.rng);
    } while (e == 0.0f);
    const float x = last * e;
    const uint64_t ge = wv::ballot(l < G::NA && !(cdf < x));
    const int action = ge ? wv::ctz64(ge) : G::A - 1;
    advance_root<G>(P, g, slot, lds, action);
    g.d_plies++;
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// one game slot, up to P.rounds rounds
// ---------------------------------------------------------------------------------------------------
template <class G>
SPRL_DEV void game_load(const EngineParams& P, int slot, GameCtl* ctl, WaveLds<G>* lds, Game& g) {
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
    g.d_recycled = 0;
    g.d_traversals = g.d_levels = g.d_expansions = g.d_nn_evals = g.d_terminal = g.d_gray = g.d_dup = 0;
    g.d_created = g.d_compactions = g.d_games = g.d_plies = 0;
#if defined(SPRL_PHASE_TIMERS) && !defined(SPRL_EMU)
    g.cyc_finish = g.cyc_move = g.cyc_select = g.cyc_create = g.cyc_backup = g.cyc_leafio = g.cyc_noise = 0;
    g.cyc_lvl_wait = g.cyc_lvl_pick = g.cyc_lvl_desc = 0;
#endif
    g.abase = P.arenas + (size_t)g.arena * (size_t)P.node_cap * SPRL_NODE_BYTES;
    wv::sync();

    if (G::HIST_CAP > 1 && g.status == ST_ACTIVE) {          // real-game positions 0..ply back into LDS
        const uint64_t* gh = P.hist_boards + (size_t)slot * G::HIST_CAP * 2;
        for (int i = wv::lane(); i <= g.ply; i += 64) {
            lds->hist[i][0] = gh[2 * i];
            lds->hist[i][1] = gh[2 * i + 1];
        }
        wv::sync();
    }
}

template <class G>
SPRL_DEV void game_store(GameCtl* ctl, Game& g, WaveLds<G>* lds, unsigned long long t_all) {
    (void)t_all;
    ctl->rstack_n = g.rstack_n;
    ctl->fc_n = g.fc_n;
    wv::sync();
    if (wv::lane() < SPRL_FCACHE) ctl->fcache[wv::lane()] = lds->fcache[wv::lane()];
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
    if (wv::lane() == 0) {                       // one lane: these are read-modify-writes of wave-uniform locations
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
#if defined(SPRL_PHASE_TIMERS) && !defined(SPRL_EMU)
        const unsigned long long dt = __builtin_amdgcn_s_memtime() - t_all;
        t.cyc_total += dt;
        if (dt > t.cyc_max) t.cyc_max = dt;
        t.cyc_finish += g.cyc_finish; t.cyc_move += g.cyc_move; t.cyc_select += g.cyc_select;
        t.cyc_create += g.cyc_create; t.cyc_backup += g.cyc_backup; t.cyc_leafio += g.cyc_leafio;
        t.cyc_noise += g.cyc_noise;
        t.cyc_lvl_wait += g.cyc_lvl_wait; t.cyc_lvl_pick += g.cyc_lvl_pick; t.cyc_lvl_desc += g.cyc_lvl_desc;
#endif
    }
}

template <class G>
SPRL_DEV void step_game(const EngineParams& P, int slot, WaveLds<G>* lds) {
    GameCtl* ctl = P.ctl + slot;
    Game g;
    g.status = ctl->status;
    if (g.status == ST_IDLE || g.status == ST_ERROR) return;
    game_load<G>(P, slot, ctl, lds, g);
#if defined(SPRL_PHASE_TIMERS) && !defined(SPRL_EMU)
    const unsigned long long t_all = __builtin_amdgcn_s_memtime();
#else
    const unsigned long long t_all = 0;
#endif
    if (g.status == ST_FRESH) start_game<G>(P, g, slot, lds);

    for (int round = 0; round < P.rounds && g.status == ST_ACTIVE; ++round) {
        { SPRL_TIC(t_f); if (g.n_leaves > 0) finish_leaves<G>(P, g, slot, ctl); SPRL_TOC(g.cyc_finish, t_f); }
        // while (traversals < numTraversals) ... ; then the move; then the next ply's search begins
        bool idle = false;
        while (g.traversals >= P.num_traversals) {
            int resigned;
            { SPRL_TIC(t_m); resigned = play_move<G>(P, g, slot, lds); SPRL_TOC(g.cyc_move, t_m); }
            if (g.status != ST_ACTIVE) break;
            const NodeHdr rh = load_hdr(node_at(g.abase, g.root));
            if (resigned || (rh.flags & F_TERMINAL)) {            // SelfPlay.hpp:85,151 (or the resign extension)
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
        { SPRL_TIC(t_s); select_batch<G>(P, g, slot, ctl, lds); SPRL_TOC(g.cyc_select, t_s); }
    }

    game_store<G>(ctl, g, lds, t_all);
    P.leaf_count[slot] = g.status == ST_ACTIVE ? (uint32_t)g.n_leaves : 0u;
    if (g.status == ST_ACTIVE && wv::lane() == 0) wv::atomic_add_u32(&P.counters->active_slots, 1u);
}


// ---------------------------------------------------------------------------------------------------
// match play (Evaluate.cpp:104-170, agents/UCTNetworkAgent.hpp, interface/play.hpp:22-60)
//
// Game g of a match is played by TWO trees, one per agent, in slots (pair, pair + n): agent k owns colour
// k ^ (g & 1) (Evaluate.cpp:126-130).  Only the side to move searches; its move is the first maximum of the visit
// counts (UCTNetworkAgent.hpp:88-89), applied to its own tree (`act`) and posted — together with the game's RNG
// state, which the reference keeps in one global stream — to the partner's mailbox.  The partner applies it as
// `opponentAct` in a LATER launch: kernel boundaries are the only cross-wave ordering the hand-over relies on.
// ---------------------------------------------------------------------------------------------------
template <class G>
SPRL_DEV void step_match(const EngineParams& P0, int slot, WaveLds<G>* lds) {
    const int n = P0.num_slots >> 1;
    const int agent = slot >= n ? 1 : 0;
    const int partner = agent ? slot - n : slot + n;
    EngineParams P = P0;
    P.use_sym = P0.m_use_sym[agent];
    P.eval_kind = P0.m_eval_kind[agent];
    P.init_q_zero = P0.m_init_q_zero[agent];
    GameCtl* ctl = P.ctl + slot;
    Game g;
    g.status = ctl->status;
    if (g.status == ST_IDLE || g.status == ST_ERROR) {
        P.leaf_count[slot] = 0u;
        return;
    }
    game_load<G>(P, slot, ctl, lds, g);
    if (g.status == ST_FRESH) init_game<G>(P, g, slot, lds, (uint32_t)(slot - agent * n));

    int round = 0;
    while (round < P.rounds && g.status == ST_ACTIVE) {
        if (g.n_leaves > 0) finish_leaves<G>(P, g, slot, ctl);
        const NodeHdr rh = load_hdr(node_at(g.abase, g.root));
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
            uint8_t* np = node_at(g.abase, g.root);
            const int l = wv::lane();
            const float visits = l < G::NA ? rowN(np)[l] : -1.0f;
            const float top = wv::fmax_all(visits);
            int action = __builtin_ctzll(wv::ballot(visits == top));
            if (G::HAS_PASS && rh.passN > top) action = PASS_A;        // std::max_element: the first maximum
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

    game_store<G>(ctl, g, lds, 0ull);
    P.leaf_count[slot] = (g.status == ST_ACTIVE && P.eval_kind == EVAL_NETWORK) ? (uint32_t)g.n_leaves : 0u;
    if (g.status == ST_ACTIVE && wv::lane() == 0) wv::atomic_add_u32(&P.counters->active_slots, 1u);
}

}  // namespace sprl

#endif  // SPRL_STEP_KERNEL_H
