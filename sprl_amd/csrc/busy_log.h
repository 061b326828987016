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
// together with the rest of the profile mode 4 % of the bench's games/s).  One pool per host thread, no locking.
inline std::vector<hipEvent_t>& pool() {
    static thread_local std::vector<hipEvent_t> p;
    return p;
}
inline hipEvent_t get_event() {
    auto& p = pool();
    if (!p.empty()) {
        hipEvent_t e = p.back();
        p.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    return hipEventCreate(&e) == hipSuccess ? e : nullptr;
}
inline void put_event(hipEvent_t e) {
    if (!e) return;
    auto& p = pool();
    if (p.size() < 65536) p.push_back(e);
    else (void)hipEventDestroy(e);
}

// One per owner (engine or model): turns its event pairs into intervals on the process clock.
struct Chain {
    hipEvent_t ref = nullptr;     // an already resolved event of this owner (or base()); owned unless it is base()
    double ref_ms = 0.0;
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
