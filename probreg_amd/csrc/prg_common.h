// Shared host-side helpers for libprobreg_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
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

}  // namespace prg
