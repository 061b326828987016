// TEST INFRASTRUCTURE — not part of the product.
//
// C-ABI harness around the *unmodified* reference sources, compiled from where
// they lie under /root/reference (see oracle/Makefile; outputs go to
// oracle/_ref/ only, which is git-ignored).  No reference source is copied: this
// file only #includes reference headers and calls their public interfaces
// (selfPlay / runIteration / runWorker / UCTTree / GameNode / ISymmetrizer /
// INetwork / Random) so that
//   (1) golden vectors under tests/golden/ can be generated
//       (tests/golden/gen_golden.py), and
//   (2) the CPU restatement in oracle/sprl_oracle.c can be pinned bit-for-bit, and
//   (3) bench.py can time the reference CPU path as `cpu_baseline.kind=reference`.
//
// Determinism: the reference seeds its process-global RNG from random_device
// (constants.hpp: SEED = 0, utils/random.cpp:38-47).  The harness re-seeds that
// global object in place (placement-new on GetRandom()) — the reference code
// itself is untouched.

#include "games/OthelloNode.hpp"
#include "games/ConnectFourNode.hpp"
#include "games/GoNode.hpp"
#include "networks/RandomNetwork.hpp"
#include "networks/OthelloHeuristic.hpp"
#include "symmetry/D4GridSymmetrizer.hpp"
#include "symmetry/ConnectFourSymmetrizer.hpp"
#include "selfplay/SelfPlay.hpp"
#include "uct/UCTTree.hpp"
#include "agents/UCTNetworkAgent.hpp"
#include "evaluate/play.hpp"
#include "utils/random.hpp"
#include "utils/npy.hpp"

#ifdef REF_WITH_TORCH
#include "networks/GridNetwork.hpp"
#include "selfplay/GridWorker.hpp"
#endif

#include <cstdint>
#include <cstring>
#include <new>
#include <sstream>
#include <vector>

using namespace SPRL;

namespace {

void reseed(uint64_t seed, int stream) {
    Random& r = GetRandom();
    r.~Random();
    new (&r) Random(seed, stream);
}

struct OthTraits {
    using Node = OthelloNode;
    static constexpr int CELLS = OTH_BOARD_SIZE;
    static constexpr int HIST = OTH_HISTORY_SIZE;
    static constexpr int A = OTH_ACTION_SIZE;
    using State = GridState<CELLS, HIST>;
    using Sym = D4GridSymmetrizer<OTH_BOARD_WIDTH, OTH_HISTORY_SIZE>;
};

struct GoTraits {                      // the reference compiles Go as 7x7 (games/GoNode.hpp:16-22)
    using Node = GoNode;
    static constexpr int CELLS = GO_BOARD_SIZE;
    static constexpr int HIST = GO_HISTORY_SIZE;
    static constexpr int A = GO_ACTION_SIZE;
    using State = GridState<CELLS, HIST>;
    using Sym = D4GridSymmetrizer<GO_BOARD_WIDTH, GO_HISTORY_SIZE>;
};

struct C4Traits {
    using Node = ConnectFourNode;
    static constexpr int CELLS = C4_BOARD_SIZE;
    static constexpr int HIST = C4_HISTORY_SIZE;
    static constexpr int A = C4_ACTION_SIZE;
    using State = GridState<CELLS, HIST>;
    using Sym = ConnectFourSymmetrizer;
};

// A do-nothing NeuralNetwork type for runWorker's template parameter so that the
// worker loop can run with the built-in initial evaluator only.
template <typename T>
struct NullNet : public INetwork<typename T::State, T::A> {
    explicit NullNet(std::string) {}
    std::vector<std::pair<GameActionDist<T::A>, Value>> evaluate(
        const std::vector<typename T::State>&, const std::vector<GameActionDist<T::A>>&) override {
        return {};
    }
    int getNumEvals() override { return 0; }
};

template <typename T>
INetwork<typename T::State, T::A>* makeEvaluator(int kind) {
    // 0 = RandomNetwork, 1 = OthelloHeuristic (Othello only)
    if constexpr (std::is_same_v<T, OthTraits>) {
        if (kind == 1) return new OthelloHeuristic();
    }
    return new RandomNetwork<typename T::State, T::A>();
}

// boards: [HIST][CELLS]; plies t >= size() are not defined by the reference and are written as -2
template <typename T>
void dumpState(const typename T::State& s, int8_t* board, int8_t* player, int8_t* size = nullptr) {
    const auto& h = s.getHistory();
    for (int t = 0; t < T::HIST; ++t)
        for (int i = 0; i < T::CELLS; ++i)
            board[t * T::CELLS + i] = t < s.size() ? static_cast<int8_t>(h[t][i]) : (int8_t)-2;
    *player = static_cast<int8_t>(s.getPlayer());
    if (size) *size = (int8_t)s.size();
}

// Random legal play-out through the GameNode API (G1).
template <typename T>
int playout(uint64_t seed, int stream, int maxPlies,
            int8_t* boards, int8_t* players, int16_t* actions,
            float* masks, int8_t* terminal, float* rewards) {
    reseed(seed, stream);
    typename T::Node root;
    GameNode<typename T::Node, typename T::State, T::A>* node = &root;
    int ply = 0;
    while (true) {
        dumpState<T>(node->getGameState(), boards + ply * T::HIST * T::CELLS, players + ply);
        const auto& m = node->getActionMask();
        for (int a = 0; a < T::A; ++a) masks[ply * T::A + a] = m[a];
        terminal[ply] = node->isTerminal();
        auto rw = node->getRewards();
        rewards[2 * ply] = rw[0];
        rewards[2 * ply + 1] = rw[1];
        if (node->isTerminal() || ply + 1 >= maxPlies) {
            actions[ply] = -1;
            return ply + 1;
        }
        std::vector<int> legal;
        for (int a = 0; a < T::A; ++a) if (m[a] > 0.0f) legal.push_back(a);
        int a = legal[GetRandom().UniformInt(0, (int)legal.size() - 1)];
        actions[ply] = (int16_t)a;
        node = node->getAddChild((ActionIdx)a);
        ++ply;
    }
}

template <typename T>
int selfplayGames(int evalKind, void* netOverride, int numGames, int numTraversals, int maxBatch, int maxQueue,
                  float eps, float alpha, int useSym, int addNoise,
                  uint64_t seed, int streamBase, int perGameStream, int cap,
                  int8_t* boards, int8_t* players, float* dists, float* outcomes, int32_t* gameOffsets,
                  int8_t* sizes = nullptr) {
    typename T::Sym sym;
    INetwork<typename T::State, T::A>* net =
        netOverride ? static_cast<INetwork<typename T::State, T::A>*>(netOverride) : makeEvaluator<T>(evalKind);
    if (!perGameStream) reseed(seed, streamBase);
    int n = 0;
    for (int g = 0; g < numGames; ++g) {
        if (perGameStream) reseed(seed, streamBase + g);
        std::unique_ptr<GameNode<typename T::Node, typename T::State, T::A>> root = std::make_unique<typename T::Node>();
        auto [states, distributions, outs] = selfPlay<typename T::Node, typename T::State, T::A>(
            std::move(root), net, numTraversals, maxBatch, maxQueue, eps, alpha, InitQ::PARENT,
            useSym ? &sym : nullptr, addNoise != 0);
        gameOffsets[g] = n;
        for (size_t i = 0; i < states.size(); ++i) {
            if (n >= cap) return -1;
            dumpState<T>(states[i], boards + (size_t)n * T::HIST * T::CELLS, players + n, sizes ? sizes + n : nullptr);
            for (int a = 0; a < T::A; ++a) dists[(size_t)n * T::A + a] = distributions[i][a];
            outcomes[n] = outs[i];
            ++n;
        }
    }
    gameOffsets[numGames] = n;
    if (!netOverride) delete net;
    return n;
}

// Search trace (G4): `moves` decisions from the start position; after each search
// dump the decision node's outgoing N/W/P, then advance by the first arg-max-N action.
template <typename T>
int searchTrace(int evalKind, int moves, int numTraversals, int maxBatch, int maxQueue,
                float eps, float alpha, int useSym, int addNoise, uint64_t seed, int stream,
                float* stats /*[moves][3][A]*/, int32_t* trav /*[moves]*/, int16_t* chosen /*[moves]*/) {
    typename T::Sym sym;
    auto* net = makeEvaluator<T>(evalKind);
    reseed(seed, stream);
    std::unique_ptr<GameNode<typename T::Node, typename T::State, T::A>> root = std::make_unique<typename T::Node>();
    UCTTree<typename T::Node, typename T::State, T::A> tree { std::move(root), eps, alpha, InitQ::PARENT,
                                                            useSym ? &sym : nullptr, addNoise != 0 };
    int m = 0;
    for (; m < moves && !tree.getDecisionNode()->isTerminal(); ++m) {
        int traversals = 0;
        while (traversals < numTraversals) {
            auto [leaves, t] = tree.searchAndGetLeaves(maxBatch, maxQueue, net, U_WEIGHT);
            if (leaves.size() > 0) tree.evaluateAndBackpropLeaves(leaves, net);
            traversals += t;
        }
        trav[m] = traversals;
        const auto* es = tree.getDecisionNode()->getEdgeStatistics();
        int best = 0;
        for (int a = 0; a < T::A; ++a) {
            stats[(m * 3 + 0) * T::A + a] = es->m_numVisits[a];
            stats[(m * 3 + 1) * T::A + a] = es->m_totalValues[a];
            stats[(m * 3 + 2) * T::A + a] = es->m_childPriors[a];
            if (es->m_numVisits[a] > es->m_numVisits[best]) best = a;
        }
        chosen[m] = (int16_t)best;
        tree.advanceDecision((ActionIdx)best);
    }
    delete net;
    return m;
}

// Net-vs-net matches exactly as Evaluate.cpp:88-154 sets them up (trees with eps 0.25 / alpha 0.1 / noise on,
// InitQ per agent, optional symmetrizer, UCTNetworkAgent, colours alternating with the game index), driven by the
// reference's playGame (evaluate/play.hpp:24-70).  Each game is replayed once more with the same seed through the
// same agent calls to record the action sequence; the two runs must agree on the winner.
template <typename T>
int playMatches(int kind0, int kind1, int numGames, int numTraversals, int maxBatch, int maxQueue,
                int useSym0, int parentQ0, int useSym1, int parentQ1, uint64_t seed, int streamBase,
                int8_t* winners, int16_t* actions, int32_t* nplies, int maxPlies) {
    typename T::Sym sym;
    auto* net0 = makeEvaluator<T>(kind0);
    auto* net1 = makeEvaluator<T>(kind1);
    using Tree = UCTTree<typename T::Node, typename T::State, T::A>;
    using Agent = UCTNetworkAgent<typename T::Node, typename T::State, T::A>;
    for (int t = 0; t < numGames; ++t) {
        Player winnerRef = Player::NONE;
        for (int pass = 0; pass < 2; ++pass) {
            reseed(seed, streamBase + t);
            Tree tree0 { std::make_unique<typename T::Node>(), 0.25, 0.1, parentQ0 ? InitQ::PARENT : InitQ::ZERO,
                         useSym0 ? &sym : nullptr, true };
            Tree tree1 { std::make_unique<typename T::Node>(), 0.25, 0.1, parentQ1 ? InitQ::PARENT : InitQ::ZERO,
                         useSym1 ? &sym : nullptr, true };
            Agent a0 { net0, &tree0, numTraversals, maxBatch, maxQueue };
            Agent a1 { net1, &tree1, numTraversals, maxBatch, maxQueue };
            std::array<IAgent<typename T::Node, typename T::State, T::A>*, 2> agents;
            if (t % 2 == 0) agents = { &a0, &a1 };
            else agents = { &a1, &a0 };
            typename T::Node rootNode {};
            if (pass == 0) {
                winnerRef = playGame<typename T::Node, typename T::State, T::A>(&rootNode, agents, false);
            } else {
                GameNode<typename T::Node, typename T::State, T::A>* cur = &rootNode;
                int ply = 0;
                while (!cur->isTerminal()) {
                    int pi = static_cast<int>(cur->getPlayer());
                    ActionIdx a = agents[pi]->act(cur, false);
                    agents[1 - pi]->opponentAct(a);
                    if (ply < maxPlies) actions[t * maxPlies + ply] = a;
                    ++ply;
                    cur = cur->getAddChild(a);
                }
                if (cur->getWinner() != winnerRef) return -1;
                nplies[t] = ply;
                winners[t] = static_cast<int8_t>(cur->getWinner());
            }
        }
    }
    delete net0;
    delete net1;
    return 0;
}

template <typename T>
void symmetrize(const int8_t* board, int player, const float* dist, int nsym,
                int8_t* boardsOut, float* distsOut, int8_t* inverseOut) {
    typename T::Sym sym;
    std::array<GridBoard<T::CELLS>, T::HIST> hist;
    for (int i = 0; i < T::CELLS; ++i) hist[0][i] = static_cast<Piece>(board[i]);
    typename T::State st { std::move(hist), T::HIST, static_cast<Player>(player) };
    GameActionDist<T::A> d;
    for (int a = 0; a < T::A; ++a) d[a] = dist[a];
    std::vector<SymmetryIdx> all;
    for (int s = 0; s < nsym; ++s) all.push_back((SymmetryIdx)s);
    auto ss = sym.symmetrizeState(st, all);
    auto dd = sym.symmetrizeActionDist(d, all);
    for (int s = 0; s < nsym; ++s) {
        int8_t p;
        dumpState<T>(ss[s], boardsOut + s * T::CELLS, &p);
        for (int a = 0; a < T::A; ++a) distsOut[s * T::A + a] = dd[s][a];
        inverseOut[s] = sym.inverseSymmetry((SymmetryIdx)s);
    }
}

} // namespace

extern "C" {

void ref_seed(uint64_t seed, int stream) { reseed(seed, stream); }

// G6: raw engine words come out of UniformUint64(0, 2^32-1) (range == engine range
// → libstdc++ returns the raw draw).
void ref_rng_raw(int n, uint32_t* out) {
    for (int i = 0; i < n; ++i) out[i] = (uint32_t)GetRandom().UniformUint64(0, 0xFFFFFFFFull);
}
int ref_uniform_int(int a, int b) { return GetRandom().UniformInt(a, b); }
float ref_uniform_float() { return GetRandom()(); }
void ref_dirichlet(float alpha, int k, float* out) {
    std::vector<float> v(k);
    GetRandom().Dirichlet(alpha, v);
    for (int i = 0; i < k; ++i) out[i] = v[i];
}
int ref_sample_cdf(const float* cdf, int n) {
    return GetRandom().SampleCDF(std::vector<float>(cdf, cdf + n));
}
uint64_t ref_rng_state() { return GetRandom().state(); }

int ref_othello_playout(uint64_t seed, int stream, int maxPlies, int8_t* boards, int8_t* players,
                        int16_t* actions, float* masks, int8_t* terminal, float* rewards) {
    return playout<OthTraits>(seed, stream, maxPlies, boards, players, actions, masks, terminal, rewards);
}
int ref_c4_playout(uint64_t seed, int stream, int maxPlies, int8_t* boards, int8_t* players,
                   int16_t* actions, float* masks, int8_t* terminal, float* rewards) {
    return playout<C4Traits>(seed, stream, maxPlies, boards, players, actions, masks, terminal, rewards);
}

// Known-answer of cpp/tests/test_c4.cpp:11-25 executed on the reference code itself.
int ref_c4_known_answer(float* rewardsOut) {
    ConnectFourNode root;
    GameNode<ConnectFourNode, ConnectFourNode::State, C4_ACTION_SIZE>* cur = &root;
    std::vector<ActionIdx> actions { 3, 3, 4, 4, 2, 3, 1 };
    int ok = 1;
    for (ActionIdx a : actions) {
        auto* nxt = cur->getAddChild(a);
        ok &= !cur->isTerminal();
        ok &= cur->getWinner() == Player::NONE;
        ok &= nxt->getParent() == cur;
        ok &= nxt->getPlayer() == otherPlayer(cur->getPlayer());
        cur = nxt;
    }
    ok &= cur->isTerminal();
    auto rw = cur->getRewards();
    rewardsOut[0] = rw[0];
    rewardsOut[1] = rw[1];
    return ok;
}

// One Othello transition from an arbitrary position (public OthelloNode ctor).
void ref_othello_step(const int8_t* board, int player, int action,
                      int8_t* boardOut, float* maskOut, int8_t* terminalOut, float* rewardsOut) {
    OthelloNode::Board b;
    for (int i = 0; i < OTH_BOARD_SIZE; ++i) b[i] = static_cast<Piece>(board[i]);
    GameActionDist<OTH_ACTION_SIZE> allOnes;
    allOnes.fill(1.0f);
    OthelloNode n(nullptr, 0, std::move(allOnes), static_cast<Player>(player), Player::NONE, false, std::move(b));
    GameNode<OthelloNode, OthTraits::State, OTH_ACTION_SIZE>* c = n.getAddChild((ActionIdx)action);
    int8_t p;
    dumpState<OthTraits>(c->getGameState(), boardOut, &p);
    for (int a = 0; a < OTH_ACTION_SIZE; ++a) maskOut[a] = c->getActionMask()[a];
    *terminalOut = c->isTerminal();
    auto rw = c->getRewards();
    rewardsOut[0] = rw[0];
    rewardsOut[1] = rw[1];
}

void ref_othello_symmetrize(const int8_t* board, int player, const float* dist,
                            int8_t* boardsOut, float* distsOut, int8_t* inverseOut) {
    symmetrize<OthTraits>(board, player, dist, 8, boardsOut, distsOut, inverseOut);
}
void ref_c4_symmetrize(const int8_t* board, int player, const float* dist,
                       int8_t* boardsOut, float* distsOut, int8_t* inverseOut) {
    symmetrize<C4Traits>(board, player, dist, 2, boardsOut, distsOut, inverseOut);
}

// OthelloHeuristic / RandomNetwork on explicit (board, player, mask) batches (G7-lite).
void ref_othello_evaluate(int kind, int n, const int8_t* boards, const int8_t* players, const float* masks,
                          float* policies, float* values) {
    auto* net = makeEvaluator<OthTraits>(kind);
    std::vector<OthTraits::State> states;
    std::vector<GameActionDist<OTH_ACTION_SIZE>> ms;
    for (int b = 0; b < n; ++b) {
        std::array<GridBoard<OTH_BOARD_SIZE>, 1> hist;
        for (int i = 0; i < OTH_BOARD_SIZE; ++i) hist[0][i] = static_cast<Piece>(boards[b * OTH_BOARD_SIZE + i]);
        states.emplace_back(std::move(hist), 1, static_cast<Player>(players[b]));
        GameActionDist<OTH_ACTION_SIZE> m;
        for (int a = 0; a < OTH_ACTION_SIZE; ++a) m[a] = masks[b * OTH_ACTION_SIZE + a];
        ms.push_back(m);
    }
    auto out = net->evaluate(states, ms);
    for (int b = 0; b < n; ++b) {
        for (int a = 0; a < OTH_ACTION_SIZE; ++a) policies[b * OTH_ACTION_SIZE + a] = out[b].first[a];
        values[b] = out[b].second;
    }
    delete net;
}

int ref_othello_selfplay(int evalKind, int numGames, int numTraversals, int maxBatch, int maxQueue,
                         float eps, float alpha, int useSym, int addNoise,
                         uint64_t seed, int streamBase, int perGameStream, int cap,
                         int8_t* boards, int8_t* players, float* dists, float* outcomes, int32_t* gameOffsets) {
    return selfplayGames<OthTraits>(evalKind, nullptr, numGames, numTraversals, maxBatch, maxQueue, eps, alpha, useSym,
                                    addNoise, seed, streamBase, perGameStream, cap, boards, players, dists, outcomes,
                                    gameOffsets);
}
int ref_c4_selfplay(int evalKind, int numGames, int numTraversals, int maxBatch, int maxQueue,
                    float eps, float alpha, int useSym, int addNoise,
                    uint64_t seed, int streamBase, int perGameStream, int cap,
                    int8_t* boards, int8_t* players, float* dists, float* outcomes, int32_t* gameOffsets) {
    return selfplayGames<C4Traits>(evalKind, nullptr, numGames, numTraversals, maxBatch, maxQueue, eps, alpha, useSym,
                                   addNoise, seed, streamBase, perGameStream, cap, boards, players, dists, outcomes,
                                   gameOffsets);
}

int ref_go_playout(uint64_t seed, int stream, int maxPlies, int8_t* boards, int8_t* players,
                   int16_t* actions, float* masks, int8_t* terminal, float* rewards) {
    return playout<GoTraits>(seed, stream, maxPlies, boards, players, actions, masks, terminal, rewards);
}
int ref_go_selfplay(int evalKind, int numGames, int numTraversals, int maxBatch, int maxQueue,
                    float eps, float alpha, int useSym, int addNoise,
                    uint64_t seed, int streamBase, int perGameStream, int cap,
                    int8_t* boards, int8_t* players, float* dists, float* outcomes, int32_t* gameOffsets,
                    int8_t* sizes) {
    return selfplayGames<GoTraits>(evalKind, nullptr, numGames, numTraversals, maxBatch, maxQueue, eps, alpha, useSym,
                                   addNoise, seed, streamBase, perGameStream, cap, boards, players, dists, outcomes,
                                   gameOffsets, sizes);
}
int ref_go_search_trace(int evalKind, int moves, int numTraversals, int maxBatch, int maxQueue,
                        float eps, float alpha, int useSym, int addNoise, uint64_t seed, int stream,
                        float* stats, int32_t* trav, int16_t* chosen) {
    return searchTrace<GoTraits>(evalKind, moves, numTraversals, maxBatch, maxQueue, eps, alpha, useSym, addNoise,
                                 seed, stream, stats, trav, chosen);
}
int ref_othello_match(int kind0, int kind1, int numGames, int numTraversals, int maxBatch, int maxQueue,
                      int useSym0, int parentQ0, int useSym1, int parentQ1, uint64_t seed, int streamBase,
                      int8_t* winners, int16_t* actions, int32_t* nplies, int maxPlies) {
    return playMatches<OthTraits>(kind0, kind1, numGames, numTraversals, maxBatch, maxQueue, useSym0, parentQ0, useSym1,
                                  parentQ1, seed, streamBase, winners, actions, nplies, maxPlies);
}
int ref_c4_match(int kind0, int kind1, int numGames, int numTraversals, int maxBatch, int maxQueue,
                 int useSym0, int parentQ0, int useSym1, int parentQ1, uint64_t seed, int streamBase,
                 int8_t* winners, int16_t* actions, int32_t* nplies, int maxPlies) {
    return playMatches<C4Traits>(kind0, kind1, numGames, numTraversals, maxBatch, maxQueue, useSym0, parentQ0, useSym1,
                                 parentQ1, seed, streamBase, winners, actions, nplies, maxPlies);
}
int ref_go_board_width() { return GO_BOARD_WIDTH; }
float ref_go_komi() { return GO_KOMI; }

int ref_othello_search_trace(int evalKind, int moves, int numTraversals, int maxBatch, int maxQueue,
                             float eps, float alpha, int useSym, int addNoise, uint64_t seed, int stream,
                             float* stats, int32_t* trav, int16_t* chosen) {
    return searchTrace<OthTraits>(evalKind, moves, numTraversals, maxBatch, maxQueue, eps, alpha, useSym, addNoise,
                                  seed, stream, stats, trav, chosen);
}
int ref_c4_search_trace(int evalKind, int moves, int numTraversals, int maxBatch, int maxQueue,
                        float eps, float alpha, int useSym, int addNoise, uint64_t seed, int stream,
                        float* stats, int32_t* trav, int16_t* chosen) {
    return searchTrace<C4Traits>(evalKind, moves, numTraversals, maxBatch, maxQueue, eps, alpha, useSym, addNoise,
                                 seed, stream, stats, trav, chosen);
}

// The reference's vendored .npy writer on caller data (header-format golden).
int ref_write_npy_f32(const char* path, const float* data, int ndim, const uint64_t* shape) {
    npy::npy_data_ptr<float> d {};
    d.data_ptr = data;
    for (int i = 0; i < ndim; ++i) d.shape.push_back((unsigned long)shape[i]);
    try {
        npy::write_npy(path, d);
    } catch (...) {
        return -1;
    }
    return 0;
}

#ifdef REF_WITH_TORCH

// The reference's GridNetwork::evaluate (LibTorch CPU) on explicit Othello batches (G7).
int ref_torch_othello_evaluate(const char* modelPath, int n, const int8_t* boards, const int8_t* players,
                               const float* masks, float* policies, float* values) {
    GridNetwork<8, 8, 1, 65> net { modelPath };
    std::vector<OthTraits::State> states;
    std::vector<GameActionDist<OTH_ACTION_SIZE>> ms;
    for (int b = 0; b < n; ++b) {
        std::array<GridBoard<OTH_BOARD_SIZE>, 1> hist;
        for (int i = 0; i < OTH_BOARD_SIZE; ++i) hist[0][i] = static_cast<Piece>(boards[b * OTH_BOARD_SIZE + i]);
        states.emplace_back(std::move(hist), 1, static_cast<Player>(players[b]));
        GameActionDist<OTH_ACTION_SIZE> m;
        for (int a = 0; a < OTH_ACTION_SIZE; ++a) m[a] = masks[b * OTH_ACTION_SIZE + a];
        ms.push_back(m);
    }
    auto out = net.evaluate(states, ms);
    for (int b = 0; b < n; ++b) {
        for (int a = 0; a < OTH_ACTION_SIZE; ++a) policies[b * OTH_ACTION_SIZE + a] = out[b].first[a];
        values[b] = out[b].second;
    }
    return 0;
}

// selfPlay with the reference GridNetwork on LibTorch-CPU: the CPU baseline proper.
int ref_torch_othello_selfplay(const char* modelPath, int numGames, int numTraversals, int maxBatch, int maxQueue,
                               float eps, float alpha, uint64_t seed, int streamBase, int perGameStream, int cap,
                               int8_t* boards, int8_t* players, float* dists, float* outcomes, int32_t* gameOffsets,
                               int64_t* numEvals) {
    at::set_num_threads(1);
    GridNetwork<8, 8, 1, 65> net { modelPath };
    int n = selfplayGames<OthTraits>(0, &net, numGames, numTraversals, maxBatch, maxQueue, eps, alpha, 1, 1, seed,
                                     streamBase, perGameStream, cap, boards, players, dists, outcomes, gameOffsets);
    if (numEvals) *numEvals = net.getNumEvals();
    return n;
}
int ref_torch_c4_selfplay(const char* modelPath, int numGames, int numTraversals, int maxBatch, int maxQueue,
                          float eps, float alpha, uint64_t seed, int streamBase, int perGameStream, int cap,
                          int8_t* boards, int8_t* players, float* dists, float* outcomes, int32_t* gameOffsets,
                          int64_t* numEvals) {
    at::set_num_threads(1);
    GridNetwork<6, 7, 1, 7> net { modelPath };
    int n = selfplayGames<C4Traits>(0, &net, numGames, numTraversals, maxBatch, maxQueue, eps, alpha, 1, 1, seed,
                                    streamBase, perGameStream, cap, boards, players, dists, outcomes, gameOffsets);
    if (numEvals) *numEvals = net.getNumEvals();
    return n;
}

// The reference worker loop (selfplay/GridWorker.hpp:84-198) for one iteration with its
// built-in initial evaluator; writes the three .npy files under saveDir (G5 byte streams).
void ref_othello_run_worker(const char* runName, const char* saveDir, int evalKind, int games, int traversals,
                            int maxBatch, int maxQueue, float eps, float alpha, uint64_t seed, int stream) {
    auto* net = makeEvaluator<OthTraits>(evalKind);
    OthTraits::Sym sym;
    reseed(seed, stream);
    runWorker<NullNet<OthTraits>, OthelloNode, 8, 8, 1, 65>(runName, saveDir, net, &sym, 1, games, traversals,
                                                            maxBatch, maxQueue, games, traversals, maxBatch, maxQueue,
                                                            eps, alpha);
    delete net;
}
void ref_c4_run_worker(const char* runName, const char* saveDir, int games, int traversals,
                       int maxBatch, int maxQueue, float eps, float alpha, uint64_t seed, int stream) {
    auto* net = makeEvaluator<C4Traits>(0);
    C4Traits::Sym sym;
    reseed(seed, stream);
    runWorker<NullNet<C4Traits>, ConnectFourNode, 6, 7, 1, 7>(runName, saveDir, net, &sym, 1, games, traversals,
                                                              maxBatch, maxQueue, games, traversals, maxBatch,
                                                              maxQueue, eps, alpha);
    delete net;
}
#endif

} // extern "C"
