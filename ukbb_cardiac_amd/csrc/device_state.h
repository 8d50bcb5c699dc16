// Per-device launch bookkeeping.  Header-only and free of HIP so that it can be unit-tested on the CPU
// (tests/test_device_state.py compiles it with g++).
//
// One process may hold engine handles on several GPUs (include/ukbb_fcn.h: one handle per device).  Two things the
// launch helpers cache are properties of a DEVICE, not of the process:
//   * hipFuncSetAttribute(MaxDynamicSharedMemorySize) -- kernels that use more than 64 KB of LDS must be granted it on
//     every device they are launched on;
//   * the compute-unit count that sizes the persistent grids.
// r02 guarded both with function-local `static bool` / `static const int`: correct for the first device a process used,
// silently wrong for the second.  Here they are small arrays indexed by the device ordinal.
#pragma once
#include <atomic>

namespace ukbb {

constexpr int MAX_DEVICES = 64;                 // ordinals beyond it are never cached (the action runs every time)

// "Run an action once per device."  One instance per kernel instantiation (function-local static).
// Lock-free: two threads racing on the same device may both run the action (it is idempotent), none skips it.
struct OncePerDevice {
    std::atomic<unsigned char> done[MAX_DEVICES];

    // f() -> 0 on success; a failing action is retried by the next call.  Returns f()'s status (0 when already done).
    template <class F>
    int run(int device, F &&f) {
        const bool cached = device >= 0 && device < MAX_DEVICES;
        if (cached && done[device].load(std::memory_order_acquire)) return 0;
        const int rc = f();
        if (rc == 0 && cached) done[device].store(1, std::memory_order_release);
        return rc;
    }
    bool is_done(int device) const { return device >= 0 && device < MAX_DEVICES && done[device].load(std::memory_order_acquire); }
};

// A per-device integer property, queried on first use (0 = not yet known; properties cached here are positive).
struct PerDeviceInt {
    std::atomic<int> v[MAX_DEVICES];

    template <class F>
    int get(int device, F &&query /* () -> int, <= 0 on failure */, int fallback) {
        const bool cached = device >= 0 && device < MAX_DEVICES;
        if (cached) {
            const int have = v[device].load(std::memory_order_acquire);
            if (have > 0) return have;
        }
        const int got = query();
        if (got <= 0) return fallback;
        if (cached) v[device].store(got, std::memory_order_release);
        return got;
    }
};

}  // namespace ukbb
