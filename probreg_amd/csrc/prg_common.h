// Shared host-side helpers for libprobreg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include <chrono>
#include <string>

#include "../../include/probreg_hip.h"

namespace prg {

void set_error(const char* fmt, ...);

#define PRG_HIP(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            prg::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,    \
                           __LINE__);                                                          \
            return PRG_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)

#define PRG_REQUIRE(cond, status, ...)                                                         \
    do {                                                                                       \
        if (!(cond)) {                                                                         \
            prg::set_error(__VA_ARGS__);                                                       \
            return (status);                                                                   \
        }                                                                                      \
    } while (0)

#define PRG_TRY(expr)                                                                          \
    do {                                                                                       \
        int _s = (expr);                                                                       \
        if (_s != PRG_OK) return _s;                                                           \
    } while (0)

static inline int64_t round_up(int64_t v, int64_t q) { return (v + q - 1) / q * q; }
static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Sentinel coordinates for padding points: source pads sit at +1e18, target pads at -1e18, so
// every pair that involves a pad has a finite, astronomically large squared distance and
// contributes exp2(-huge) == 0 to every sum - inner loops need no bounds checks.
constexpr float kSrcPad = 1.0e18f;
constexpr float kTgtPad = -1.0e18f;

// RAII device selection for API entry points.
struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};


// Wait for a device-written sequence number in a mapped host mailbox without draining the queue: a short spin (the
// answer normally arrives within microseconds of the kernel that writes it), bounded by a time budget - a stream that
// is blocked behind a cross-stream event or a collective must not keep a host core at 100 % - after which the wait
// becomes a plain hipStreamSynchronize.  Returns false if the stream has drained and the number never came.
static inline bool wait_mailbox(volatile const unsigned* seq_ptr, unsigned seq, hipStream_t st, hipError_t* err) {
    *err = hipSuccess;
    (void)hipStreamQuery(st);  // (makes sure everything enqueued so far has been handed to the device)
    const auto t0 = std::chrono::steady_clock::now();
    for (uint64_t spins = 0; *seq_ptr != seq; ++spins) {
        if ((spins & 0xFFFull) == 0xFFFull) {
            const bool late = std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5);
            if (late || hipStreamQuery(st) != hipErrorNotReady) {
                // the stream has drained (or failed), or the budget is spent: block instead of spinning
                if (*seq_ptr == seq) break;
                *err = hipStreamSynchronize(st);
                if (*err != hipSuccess || *seq_ptr != seq) return false;
                break;
            }
        }
        __builtin_ia32_pause();
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return true;
}

}  // namespace prg
