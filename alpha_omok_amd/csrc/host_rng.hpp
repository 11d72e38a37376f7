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
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

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

// The host threads that replay the Dirichlet draws of all games at the start of a move (ao_begin_move): ONE persistent
// pool per process, created on first use, instead of a std::thread spawn per move.
//
// Per-rank budget. With one process per GPU, eight ranks share the host: the pool takes
// hardware_concurrency / LOCAL_WORLD_SIZE threads (torchrun exports LOCAL_WORLD_SIZE; 1 without a launcher), at most 32,
// at least 1; AO_HOST_THREADS overrides. The calling thread works too, so `threads()` - 1 workers are parked on a
// condition variable between moves. Work is handed out in chunks through one atomic counter.
class HostPool {
  public:
    static HostPool& get() {
        static HostPool* pool = new HostPool();   // never destroyed: no join at interpreter exit, the workers end with the process
        pool->after_fork();
        return *pool;
    }

    static unsigned budget() {
        if (const char* v = std::getenv("AO_HOST_THREADS")) {
            const long n = std::strtol(v, nullptr, 10);
            if (n >= 1) return static_cast<unsigned>(std::min<long>(n, 256));
        }
        long ranks = 1;
        if (const char* v = std::getenv("LOCAL_WORLD_SIZE")) ranks = std::max<long>(1, std::strtol(v, nullptr, 10));
        const long hw = std::max<long>(1, static_cast<long>(std::thread::hardware_concurrency()));
        return static_cast<unsigned>(std::min<long>(32, std::max<long>(1, hw / ranks)));
    }

    unsigned threads() const { return nthreads_; }

    // fn(i0, i1) over [0, n) in chunks of `chunk`; returns when every chunk is done. Not re-entrant (one caller at a time:
    // an engine handle is single-threaded, and engines of one process take turns on the mutex).
    void run(int n, int chunk, const std::function<void(int, int)>& fn) {
        if (n <= 0) return;
        if (nthreads_ <= 1 || n <= chunk) {
            fn(0, n);
            return;
        }
        std::lock_guard<std::mutex> serial(run_mu_);
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn;
            n_ = n;
            chunk_ = chunk;
            next_.store(0, std::memory_order_relaxed);
            busy_ = static_cast<int>(workers_.size());
            ++generation_;
        }
        cv_.notify_all();
        drain();
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [&] { return busy_ == 0; });
        fn_ = nullptr;
    }

  private:
    HostPool() { start(); }

    void start() {
        pid_ = getpid();
        nthreads_ = budget();
        workers_.clear();
        for (unsigned t = 1; t < nthreads_; ++t) workers_.emplace_back([this] { worker(); });
        for (auto& w : workers_) w.detach();
    }

    void after_fork() {
        // a forked child inherits the object but none of its threads: start over there (the old vector of detached
        // handles holds nothing to join)
        if (pid_ != getpid()) {
            new (&mu_) std::mutex();
            new (&run_mu_) std::mutex();
            new (&cv_) std::condition_variable();
            new (&done_cv_) std::condition_variable();
            generation_ = 0;
            busy_ = 0;
            start();
        }
    }

    void drain() {
        for (;;) {
            const int i0 = next_.fetch_add(chunk_, std::memory_order_relaxed);
            if (i0 >= n_) break;
            (*fn_)(i0, std::min(n_, i0 + chunk_));
        }
    }

    void worker() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return generation_ != seen; });
                seen = generation_;
            }
            drain();
            std::lock_guard<std::mutex> lk(mu_);
            if (--busy_ == 0) done_cv_.notify_one();
        }
    }

    std::mutex mu_, run_mu_;
    std::condition_variable cv_, done_cv_;
    std::vector<std::thread> workers_;
    const std::function<void(int, int)>* fn_ = nullptr;
    std::atomic<int> next_{0};
    int n_ = 0, chunk_ = 1, busy_ = 0;
    uint64_t generation_ = 0;
    unsigned nthreads_ = 1;
    pid_t pid_ = 0;
};

}  // namespace ao
