// cclqr_wscache.h -- bookkeeping of the per-thread device workspace cache of the host-pointer entry points (capi.hip), with the
// allocator passed in so that tests/emu/ws_cache_test.cpp can drive it on the CPU (no HIP in this header).
#pragma once
#include <stddef.h>
#include <utility>
#include <vector>

namespace cclqr {

struct WsCache {
    std::vector<std::pair<void*, size_t>> blocks;
    size_t used = 0;
    int device = -1;      // the device the cached blocks were allocated on
};
// The bookkeeping of the cache with the allocator passed in, so that a CPU test can drive it (tests/emu/ws_cache_test.cpp): blocks
// are handed out by call position; a block that is too small is replaced; when the calling thread has switched GPU since the blocks
// were allocated (cclqr_set_device), EVERY cached block is released on its own device first -- a block of another device handed to a
// kernel is a memory fault without peer access and silent cross-GPU traffic with it.
template <class Alloc, class Free, class SetDev>
inline int ws_get_on(WsCache& w, int cur_dev, void** p, size_t bytes, Alloc alloc, Free release, SetDev set_dev) {
    if (bytes == 0) bytes = 8;
    if (w.device != cur_dev) {
        if (w.used != 0) return -1;                   // a switch in the middle of an entry point: refuse rather than mix devices
        bool any = false;
        for (auto& b : w.blocks) any = any || b.first;
        if (any) {
            if (w.device >= 0) set_dev(w.device);
            for (auto& b : w.blocks) if (b.first) { release(b.first); b.first = nullptr; b.second = 0; }
            set_dev(cur_dev);
        }
        w.blocks.clear();
        w.device = cur_dev;
    }
    if (w.used == w.blocks.size()) w.blocks.push_back({nullptr, 0});
    auto& b = w.blocks[w.used];
    if (b.second < bytes) {
        if (b.first) { int e = release(b.first); b.first = nullptr; b.second = 0; if (e != 0) return e; }
        int e = alloc(&b.first, bytes);
        if (e != 0) { b.first = nullptr; return e; }
        b.second = bytes;
    }
    *p = b.first;
    w.used++;
    return 0;
}

}  // namespace cclqr
