// Profile mode only: kernel intervals of every engine / model of the process on one clock, so that the time during which at
// least one launch of a kind was executing ("busy time") can be told apart from the sum of the launch durations when several
// engines (bench.py --populations) run the same kernel on different HIP streams at the same time.
//
// Event times are built from chains of short hipEventElapsedTime differences (the API returns float milliseconds: a direct
// difference to an event recorded a minute earlier would be quantised to several microseconds).
#pragma once
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <mutex>
#include <utility>
#include <vector>

namespace busy {

struct Log {
    std::mutex mu;
    std::vector<std::pair<double, double>> iv;     // [start, end) in ms since base()
    void add(const std::vector<std::pair<double, double>>& more) {
        std::lock_guard<std::mutex> g(mu);
        iv.insert(iv.end(), more.begin(), more.end());
    }
    void add(const std::pair<double, double>& one) {
        std::lock_guard<std::mutex> g(mu);
        iv.push_back(one);
    }
    void reset() {
        std::lock_guard<std::mutex> g(mu);
        iv.clear();
    }
    // length of the union of the intervals; sum_ms (optional) = sum of their lengths
    double union_ms(double* sum_ms) {
        std::lock_guard<std::mutex> g(mu);
        std::sort(iv.begin(), iv.end());
        double busy = 0.0, sum = 0.0, hi = -1e300;
        for (const auto& p : iv) {
            sum += p.second - p.first;
            if (p.first >= hi) busy += p.second - p.first;
            else if (p.second > hi) busy += p.second - hi;
            if (p.second > hi) hi = p.second;
        }
        if (sum_ms) *sum_ms = sum;
        return busy;
    }
};

// the process-wide zero of the clock: recorded once, never destroyed
inline hipEvent_t base() {
    static hipEvent_t ev = nullptr;
    static std::once_flag once;
    std::call_once(once, [] {
        if (hipEventCreate(&ev) != hipSuccess) { ev = nullptr; return; }
        (void)hipEventRecord(ev, nullptr);
        (void)hipEventSynchronize(ev);
    });
    return ev;
}

// Events are reused: creating and destroying two of them around every launch costs host time on the enqueue path (measured:
// together with the rest of the profile mode 4 % of the bench's games/s).  ONE pool per process, keyed by device and guarded
// by a mutex (an uncontended lock is tens of nanoseconds; hipEventCreate is microseconds): engines are driven from short-lived
// host threads (bench.py / sprl_worker --populations start fresh threads per step), so a per-thread pool would be empty on
// every new thread and would leak what it held when the thread ends.  The pool is emptied when the last owner (Chain) of the
// process goes away.
struct Pool {
    std::mutex mu;
    std::vector<std::pair<int, std::vector<hipEvent_t>>> by_dev;   // a handful of devices at most: linear search
    int owners = 0;
    std::vector<hipEvent_t>& of(int dev) {
        for (auto& p : by_dev)
            if (p.first == dev) return p.second;
        by_dev.emplace_back(dev, std::vector<hipEvent_t>());
        return by_dev.back().second;
    }
    void drain() {                                  // caller holds mu
        for (auto& p : by_dev) {
            for (hipEvent_t e : p.second) (void)hipEventDestroy(e);
            p.second.clear();
        }
    }
};
inline Pool& pool() {
    static Pool* p = new Pool();                    // never destroyed: other static destructors may still return events
    return *p;
}
inline int current_device() {
    int d = 0;
    (void)hipGetDevice(&d);
    return d;
}
// an event of the CURRENT device (the caller has bound its engine's / model's device)
inline hipEvent_t get_event() {
    Pool& P = pool();
    const int dev = current_device();
    {
        std::lock_guard<std::mutex> g(P.mu);
        auto& v = P.of(dev);
        if (!v.empty()) {
            hipEvent_t e = v.back();
            v.pop_back();
            return e;
        }
    }
    hipEvent_t e = nullptr;
    return hipEventCreate(&e) == hipSuccess ? e : nullptr;
}
// returns an event obtained with get_event on the current device
inline void put_event(hipEvent_t e) {
    if (!e) return;
    Pool& P = pool();
    const int dev = current_device();
    {
        std::lock_guard<std::mutex> g(P.mu);
        auto& v = P.of(dev);
        if (P.owners > 0 && v.size() < 65536) {
            v.push_back(e);
            return;
        }
    }
    (void)hipEventDestroy(e);
}

// One per owner (engine or model): turns its event pairs into intervals on the process clock.
struct Chain {
    hipEvent_t ref = nullptr;     // an already resolved event of this owner (or base()); owned unless it is base()
    double ref_ms = 0.0;
    Chain() {
        std::lock_guard<std::mutex> g(pool().mu);
        ++pool().owners;
    }
    Chain(const Chain&) = delete;
    Chain& operator=(const Chain&) = delete;
    ~Chain() {
        release();
        std::lock_guard<std::mutex> g(pool().mu);
        if (--pool().owners == 0) pool().drain();
    }
    // both events have completed; takes ownership of `start` (kept as the next reference), the caller destroys `end`
    std::pair<double, double> resolve(hipEvent_t start, hipEvent_t end, double* dur_ms) {
        if (!ref) { ref = base(); ref_ms = 0.0; }
        float a = 0.0f, d = 0.0f;
        if (ref) (void)hipEventElapsedTime(&a, ref, start);
        (void)hipEventElapsedTime(&d, start, end);
        const double s = ref_ms + (double)a;
        if (ref && ref != base()) put_event(ref);
        ref = start;
        ref_ms = s;
        if (dur_ms) *dur_ms = (double)d;
        return { s, s + (double)d };
    }
    void release() {
        if (ref && ref != base()) put_event(ref);
        ref = nullptr;
    }
};

}  // namespace busy
