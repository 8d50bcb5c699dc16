// CPU unit test of ukbb_cardiac_amd/csrc/device_state.h (built and run by tests/test_device_state.py).
#include <cassert>
#include <cstdio>
#include <thread>
#include <vector>

#include "../../ukbb_cardiac_amd/csrc/device_state.h"

using namespace ukbb;

static OncePerDevice g_once;          // zero-initialised static storage, like the function-local statics of the launch helpers
static PerDeviceInt g_cu;

int main() {
    int calls[MAX_DEVICES + 2] = {0};
    // the action runs once per device, not once per process
    for (int round = 0; round < 3; ++round)
        for (int d = 0; d < 8; ++d) assert(g_once.run(d, [&] { ++calls[d]; return 0; }) == 0);
    for (int d = 0; d < 8; ++d) assert(calls[d] == 1 && g_once.is_done(d));
    assert(!g_once.is_done(8));
    // a failing action reports its status and is retried by the next launch
    int fails = 0;
    assert(g_once.run(9, [&] { ++fails; return 719; }) == 719);
    assert(!g_once.is_done(9));
    assert(g_once.run(9, [&] { ++fails; return 0; }) == 0 && fails == 2 && g_once.is_done(9));
    // ordinals outside the table (or an unknown current device, -1) are never cached: the action runs every time
    int outside = 0;
    for (int k = 0; k < 3; ++k) { g_once.run(MAX_DEVICES, [&] { ++outside; return 0; }); g_once.run(-1, [&] { ++outside; return 0; }); }
    assert(outside == 6);
    // per-device integer: queried once per device, distinct values per device, fallback on failure without caching it
    int queries = 0;
    for (int round = 0; round < 2; ++round)
        for (int d = 0; d < 4; ++d) assert(g_cu.get(d, [&] { ++queries; return 100 + d; }, 256) == 100 + d);
    assert(queries == 4);
    assert(g_cu.get(5, [&] { return 0; }, 256) == 256);
    assert(g_cu.get(5, [&] { return 304; }, 256) == 304);
    assert(g_cu.get(-1, [&] { return 0; }, 256) == 256);
    // threads racing on one device: nobody skips the action before it has completed once
    static OncePerDevice race;
    std::atomic<int> ran{0};
    std::vector<std::thread> th;
    for (int t = 0; t < 8; ++t)
        th.emplace_back([&] { for (int k = 0; k < 1000; ++k) { race.run(3, [&] { ++ran; return 0; }); assert(race.is_done(3)); } });
    for (auto &t : th) t.join();
    assert(ran.load() >= 1 && ran.load() <= 8);
    std::puts("device_state ok");
    return 0;
}
