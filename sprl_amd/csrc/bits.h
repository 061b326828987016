// bits.h — fixed-width bit sets for board games (host + gfx950 device).
// Boards up to 64 cells use one uint64 (Othello, Connect Four, Go 7x7); wider boards (Go 9x9 = 81 cells,
// 19x19 = 361 cells) use W words.  Bits<1> is a plain uint64 with conversions, so single-word code paths keep
// their native 64-bit arithmetic.
#ifndef SPRL_BITS_H
#define SPRL_BITS_H

#include <stdint.h>

#if defined(__HIPCC__) && !defined(SPRL_EMU)
#define SPRL_B __host__ __device__ inline
#else
#define SPRL_B inline
#endif

template <int W>
struct Bits {
    uint64_t w[W];

    SPRL_B static Bits zero() {
        Bits b;
        for (int i = 0; i < W; ++i) b.w[i] = 0;
        return b;
    }
    SPRL_B static Bits bit(int i) {
        Bits b = zero();
        b.w[i >> 6] = 1ull << (i & 63);
        return b;
    }
    SPRL_B bool test(int i) const { return (w[i >> 6] >> (i & 63)) & 1ull; }
    SPRL_B bool any() const {
        uint64_t o = 0;
        for (int i = 0; i < W; ++i) o |= w[i];
        return o != 0;
    }
    SPRL_B int popc() const {
        int n = 0;
        for (int i = 0; i < W; ++i) n += __builtin_popcountll(w[i]);
        return n;
    }
    // number of set bits strictly below position i
    SPRL_B int rank(int i) const {
        int n = 0;
        for (int k = 0; k < W; ++k) {
            if (k < (i >> 6)) n += __builtin_popcountll(w[k]);
            else if (k == (i >> 6)) n += __builtin_popcountll(w[k] & ((1ull << (i & 63)) - 1ull));
        }
        return n;
    }
    SPRL_B int lowest() const {                // index of the lowest set bit (any() must hold)
        for (int i = 0; i < W; ++i)
            if (w[i]) return i * 64 + __builtin_ctzll(w[i]);
        return -1;
    }
    SPRL_B Bits shl(int k) const {             // 0 < k < 64
        Bits r;
        for (int i = W - 1; i >= 0; --i) r.w[i] = (w[i] << k) | (i > 0 ? (w[i - 1] >> (64 - k)) : 0ull);
        return r;
    }
    SPRL_B Bits shr(int k) const {
        Bits r;
        for (int i = 0; i < W; ++i) r.w[i] = (w[i] >> k) | (i + 1 < W ? (w[i + 1] << (64 - k)) : 0ull);
        return r;
    }
    SPRL_B Bits operator|(const Bits& o) const { Bits r; for (int i = 0; i < W; ++i) r.w[i] = w[i] | o.w[i]; return r; }
    SPRL_B Bits operator&(const Bits& o) const { Bits r; for (int i = 0; i < W; ++i) r.w[i] = w[i] & o.w[i]; return r; }
    SPRL_B Bits operator~() const { Bits r; for (int i = 0; i < W; ++i) r.w[i] = ~w[i]; return r; }
    SPRL_B bool operator==(const Bits& o) const {
        uint64_t d = 0;
        for (int i = 0; i < W; ++i) d |= w[i] ^ o.w[i];
        return d == 0;
    }
    SPRL_B bool operator!=(const Bits& o) const { return !(*this == o); }
};

#endif  // SPRL_BITS_H
