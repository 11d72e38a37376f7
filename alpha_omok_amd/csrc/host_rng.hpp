// host_rng.hpp -- numpy-legacy RandomState pieces the engine needs ON THE HOST.
//
// The per-game random stream is an MT19937 state that lives in HBM (tie-break randint and the
// action sample run in the tree kernels). The one consumer that cannot run on the device
// bit-exactly is np.random.dirichlet (agents.py:97-98,194-195): numpy's legacy gamma sampler
// calls glibc log()/pow(), so the draw is done here, on a host copy of the game's state, at the
// start of each move -- the Dirichlet is always the first consumer of the stream in a move.
//
// Follows numpy 2.2.6: random/src/mt19937/mt19937.c, random/src/legacy/legacy-distributions.c.
#pragma once
#include <cmath>
#include <cstdint>

namespace ao {

struct HostMT {
    uint32_t* mt;   // [624], caller-owned (a row of the host mirror of the device state)
    int32_t* pos;
    int32_t* has_gauss;
    double* gauss;

    static void seed(uint32_t* mt, uint32_t s) {
        mt[0] = s;
        for (uint32_t i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + i;
    }

    void regenerate() {
        constexpr uint32_t UP = 0x80000000u, LO = 0x7fffffffu, MAT = 0x9908b0dfu;
        for (int k = 0; k < 624; ++k) {
            uint32_t y = (mt[k] & UP) | (mt[(k + 1) % 624] & LO);
            mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? MAT : 0u);
        }
        *pos = 0;
    }

    uint32_t next32() {
        if (*pos >= 624) regenerate();
        uint32_t y = mt[(*pos)++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }

    double next_double() {
        const int32_t a = static_cast<int32_t>(next32() >> 5);
        const int32_t b = static_cast<int32_t>(next32() >> 6);
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }

    double gauss_polar() {
        if (*has_gauss) {
            const double t = *gauss;
            *has_gauss = 0;
            *gauss = 0.0;
            return t;
        }
        double x1, x2, r2;
        do {
            x1 = 2.0 * next_double() - 1.0;
            x2 = 2.0 * next_double() - 1.0;
            r2 = x1 * x1 + x2 * x2;
        } while (r2 >= 1.0 || r2 == 0.0);
        const double f = std::sqrt(-2.0 * std::log(r2) / r2);
        *gauss = f * x1;
        *has_gauss = 1;
        return f * x2;
    }

    double std_exponential() { return -std::log(1.0 - next_double()); }

    double std_gamma(double shape) {
        if (shape == 1.0) return std_exponential();
        if (shape == 0.0) return 0.0;
        if (shape < 1.0) {
            for (;;) {
                const double U = next_double();
                const double V = std_exponential();
                if (U <= 1.0 - shape) {
                    const double X = std::pow(U, 1. / shape);
                    if (X <= V) return X;
                } else {
                    const double Y = -std::log((1 - U) / shape);
                    const double X = std::pow(1.0 - shape + shape * Y, 1. / shape);
                    if (X <= (V + Y)) return X;
                }
            }
        }
        const double b = shape - 1. / 3.;
        const double c = 1. / std::sqrt(9 * b);
        for (;;) {
            double X, V;
            do {
                X = gauss_polar();
                V = 1.0 + c * X;
            } while (V <= 0.0);
            V = V * V * V;
            const double U = next_double();
            if (U < 1.0 - 0.0331 * (X * X) * (X * X)) return b * V;
            if (std::log(U) < 0.5 * X * X + b * (1. - V + std::log(V))) return b * V;
        }
    }

    // np.random.dirichlet(alpha * ones(k)) -> out[0..k)
    void dirichlet(double alpha, int k, double* out) {
        double acc = 0.0;
        for (int j = 0; j < k; ++j) {
            out[j] = std_gamma(alpha);
            acc += out[j];
        }
        if (k > 0) {
            const double inv = 1 / acc;
            for (int j = 0; j < k; ++j) out[j] *= inv;
        }
    }
};

}  // namespace ao
