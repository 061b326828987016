// records_kernel.h — finished-game records handled ON THE DEVICE, straight from the engine's record buffers:
//   pack    the compact wire format of the multi-GPU gather (sprl_amd/distributed.py: head, offsets, winners, two bit sets
//           of W words per ply, movers, pdfs; 16-byte aligned sections) — ranks hand RCCL a device tensor, no host bounce;
//   expand  the reference worker's training samples (states float32[N][2H+1][R][C], distributions float32[N][A],
//           outcomes float32[N], N = plies x nsym; selfplay/SelfPlay.hpp:86-92,127-133,151-189 and the plane encoding of
//           selfplay/GridWorker.hpp:146-171) written directly into a caller-provided device buffer (the trainer's window).
// One 64-lane wavefront per game (rec_one_game<G>(.., game, lane)); the CPU emulator build calls the same functions lane by
// lane.  Every output element is written by exactly one lane, so no zero-fill pass is needed.
#ifndef SPRL_RECORDS_KERNEL_H
#define SPRL_RECORDS_KERNEL_H

#include "engine_types.h"
#include "games.h"
#include "games_wide.h"

#if defined(__HIPCC__) && !defined(SPRL_EMU)
#define SPRL_R __host__ __device__ inline
#else
#define SPRL_R inline
#endif

struct RecPacked {            // section pointers of one packed shard in device memory (byte offsets: sprl_amd/distributed.py)
    int64_t* head;            // int64[12]
    int32_t* offsets;         // [games + 1]
    int8_t* winners;          // [games]
    uint64_t* stones0;        // [plies][W]
    uint64_t* stones1;
    uint8_t* movers;          // [plies]
    float* pdfs;              // [plies][A]
};

struct RecExpanded {
    float* states;            // [N][2H+1][R][C]
    float* dists;             // [N][A]
    float* outcomes;          // [N]
    int32_t nsym;             // symmetric copies per ply (1 when the engine runs without symmetrisation)
};

template <class G>
constexpr int rec_words() { return (G::CELLS + 63) / 64; }

SPRL_R int rec_bit(const uint64_t* w, int cell) { return (int)((w[cell >> 6] >> (cell & 63)) & 1ull); }

// the compact record of ply p of game g in the engine's buffers: W words of Player::ZERO stones, then W words of Player::ONE
template <class G>
SPRL_R const uint64_t* rec_board(const EngineParams& P, int g, int p) {
    return P.rec_boards + ((size_t)g * (size_t)P.max_plies + (size_t)p) * 2 * rec_words<G>();
}

template <class G>
SPRL_R void rec_pack_game(const EngineParams& P, const RecPacked& o, int g, int lane, int use_sym) {
    constexpr int W = rec_words<G>();
    const int off = P.rec_offsets[g], n = P.rec_nplies[g];
    if (lane == 0) {
        o.offsets[g] = off;
        o.winners[g] = P.rec_winner[g];
        if (g == P.num_games - 1) o.offsets[P.num_games] = off + n;
        if (g == 0) {
            const int64_t h[12] = { P.num_games, P.rec_offsets[P.num_games], G::A, G::CELLS, G::ID, G::NSYM, use_sym, G::ROWS, G::COLS, W,
                                    G::HIST, 0 };
            for (int i = 0; i < 12; ++i) o.head[i] = h[i];
        }
    }
    for (int p = 0; p < n; ++p) {
        const size_t src = (size_t)g * (size_t)P.max_plies + (size_t)p, dst = (size_t)off + (size_t)p;
        const uint64_t* b = rec_board<G>(P, g, p);
        if (lane < W) o.stones0[dst * W + lane] = b[lane];
        else if (lane < 2 * W) o.stones1[dst * W + (lane - W)] = b[lane];
        if (lane == 63) o.movers[dst] = P.rec_movers[src];
        for (int a = lane; a < G::A; a += 64) o.pdfs[dst * G::A + a] = P.rec_pdf[src * G::A + a];
    }
}

template <class G>
SPRL_R void rec_expand_game(const EngineParams& P, const RecExpanded& o, int g, int lane) {
    constexpr int W = rec_words<G>(), CELLS = G::CELLS, A = G::A, H = G::HIST, PL = 2 * G::HIST + 1;
    const int off = P.rec_offsets[g], n = P.rec_nplies[g], ns = o.nsym;
    const int w = P.rec_winner[g];
    for (int p = 0; p < n; ++p) {
        const size_t src = (size_t)g * (size_t)P.max_plies + (size_t)p;
        const int mover = P.rec_movers[src];
        const float reward = w < 0 ? 0.0f : (w == mover ? 1.0f : -1.0f);          // OthelloNode.cpp:94-100
        for (int s = 0; s < ns; ++s) {
            const size_t smp = ((size_t)off + (size_t)p) * (size_t)ns + (size_t)s;
            float* st = o.states + smp * (size_t)PL * (size_t)CELLS;
            for (int c = lane; c < CELLS; c += 64) {
                const int tc = G::map_cell(s, c);                                    // out[map_s(c)] = in[c]
                for (int t = 0; t < H; ++t) {                                        // GridWorker.hpp:152-166
                    float own = 0.0f, opp = 0.0f;
                    if (p - t >= 0) {
                        const uint64_t* b = rec_board<G>(P, g, p - t);
                        const int z = rec_bit(b, c), one = rec_bit(b + W, c);
                        own = (float)(mover == 0 ? z : one);
                        opp = (float)(mover == 0 ? one : z);
                    }
                    st[(size_t)(2 * t) * CELLS + tc] = own;
                    st[(size_t)(2 * t + 1) * CELLS + tc] = opp;
                }
                st[(size_t)(2 * H) * CELLS + c] = mover == 0 ? 1.0f : 0.0f;         // colour plane
            }
            float* di = o.dists + smp * (size_t)A;
            for (int a = lane; a < A; a += 64) di[G::map_action(s, a)] = P.rec_pdf[src * A + a];
            if (lane == 0) o.outcomes[smp] = reward;
        }
    }
}

#endif  // SPRL_RECORDS_KERNEL_H
