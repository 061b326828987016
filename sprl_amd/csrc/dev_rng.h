// dev_rng.h — per-game random streams on the device.
//
// The reference has ONE process-global PCG32 (utils/random.cpp:29-32) feeding libstdc++ distributions
// (utils/random.cpp:61-98).  On the GPU every game owns its own PCG32 stream — Random(seed, stream =
// stream_base + global game index) in the reference's own constructor semantics (utils/random.hpp:92-103) —
// and the draw order inside a game is the reference's, so a game is reproducible independent of how many
// games run concurrently and can be replayed bit-for-bit by the CPU oracle.
//
// The distribution algorithms are restated from libstdc++ (GCC 11) so the integer draws match the
// reference exactly: uniform_int_distribution = Lemire's method on a 32-bit URBG; generate_canonical<float,24>
// = one draw / 2^32 clamped below 1; normal_distribution = Marsaglia polar with one saved deviate;
// gamma_distribution = Marsaglia-Tsang.  log/pow come from sprl_math.h (deterministic, <= 1 ulp from libm).
// All lanes of the wave execute these with identical (wave-uniform) operands.
#ifndef SPRL_DEV_RNG_H
#define SPRL_DEV_RNG_H

#include "sprl_math.h"
#include "wave.h"

struct Pcg32 {
    uint64_t state, inc;
};

SPRL_DEV uint32_t rng_next(Pcg32& r) {
    uint64_t x = r.state;
    uint32_t xs = (uint32_t)(((x >> 18) ^ x) >> 27);
    uint32_t rot = (uint32_t)(x >> 59);
    r.state = x * 6364136223846793005ull + r.inc;
    return (xs >> rot) | (xs << ((0u - rot) & 31u));
}

SPRL_DEV void rng_seed(Pcg32& r, uint64_t seed, int stream) {
    r.state = 0;
    r.inc = ((uint64_t)(int64_t)stream << 1) | 1u;
    rng_next(r);
    r.state += seed;
    rng_next(r);
}

// UniformInt(0, k-1), k >= 1 (utils/random.cpp:76-79)
SPRL_DEV int rng_uniform_int(Pcg32& r, uint32_t k) {
    uint64_t product = (uint64_t)rng_next(r) * (uint64_t)k;
    uint32_t low = (uint32_t)product;
    if (low < k) {
        uint32_t threshold = (0u - k) % k;
        while (low < threshold) {
            product = (uint64_t)rng_next(r) * (uint64_t)k;
            low = (uint32_t)product;
        }
    }
    return (int)(product >> 32);
}

// Random::operator() / generate_canonical<float,24> (utils/random.hpp:64-66)
SPRL_DEV float rng_uniform_float(Pcg32& r) {
    float ret = (float)rng_next(r) / 4294967296.0f;
    return ret >= 1.0f ? 0x1.fffffep-1f : ret;
}

struct NormalState {
    float saved;
    int available;
};

SPRL_DEV float rng_normal(Pcg32& r, NormalState& ns) {
    if (ns.available) {
        ns.available = 0;
        return ns.saved;
    }
    float x, y, r2;
    do {
        x = 2.0f * rng_uniform_float(r) - 1.0f;
        y = 2.0f * rng_uniform_float(r) - 1.0f;
        r2 = x * x + y * y;
    } while (r2 > 1.0f || r2 == 0.0f);
    float mult = __builtin_sqrtf(-2.0f * sprl_logf(r2) / r2);
    ns.saved = x * mult;
    ns.available = 1;
    return y * mult;
}

SPRL_DEV float rng_gamma(Pcg32& r, NormalState& ns, float alpha) {
    float malpha = alpha < 1.0f ? alpha + 1.0f : alpha;
    float a1 = malpha - 1.0f / 3.0f;
    float a2 = 1.0f / __builtin_sqrtf(9.0f * a1);
    float u, v, n;
    for (;;) {
        do {
            n = rng_normal(r, ns);
            v = 1.0f + a2 * n;
        } while (v <= 0.0f);
        v = v * v * v;
        u = rng_uniform_float(r);
        double nd = (double)n;
        if (!((double)u > 1.0 - 0.0331 * nd * nd * nd * nd)) break;
        if (!((double)sprl_logf(u) > 0.5 * nd * nd + (double)a1 * (1.0 - (double)v + (double)sprl_logf(v)))) break;
    }
    if (alpha == malpha) return a1 * v;
    do {
        u = rng_uniform_float(r);
    } while (u == 0.0f);
    return sprl_powf(u, 1.0f / alpha) * a1 * v;
}

#endif  // SPRL_DEV_RNG_H
