// CPU test of the workspace-cache bookkeeping of capi.hip (cclqr_wscache.h): blocks by call position, regrow, and the release of every
// cached block ON ITS OWN DEVICE when the calling thread has switched GPU.  Fake allocator: records (device, size) per block.
#include <cstdio>
#include <cstdlib>
#include <map>
#include "../../constrainedcontrol.jl_amd/csrc/cclqr_wscache.h"
using namespace cclqr;
static int g_cur = 0;                                  // the fake runtime's current device
static std::map<void*, std::pair<int, size_t>> g_live; // block -> (device it was allocated on, bytes)
static int g_wrong_device_frees = 0;
static int fake_alloc(void** p, size_t n) { *p = malloc(n ? n : 1); g_live[*p] = {g_cur, n}; return 0; }
static int fake_free(void* p) { if (g_live.at(p).first != g_cur) g_wrong_device_frees++; g_live.erase(p); free(p); return 0; }
static void fake_set(int d) { g_cur = d; }
#define CHECK(c) do { if (!(c)) { printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)
int main() {
    WsCache w;
    void *a = nullptr, *b = nullptr, *c = nullptr;
    // entry point 1 on device 0: two blocks
    CHECK(ws_get_on(w, 0, &a, 100, fake_alloc, fake_free, fake_set) == 0);
    CHECK(ws_get_on(w, 0, &b, 200, fake_alloc, fake_free, fake_set) == 0);
    CHECK(a != b && g_live.size() == 2 && w.device == 0);
    w.used = 0;                                           // ~WsScope
    // entry point 2 on device 0: same positions are reused, a larger request replaces the block
    void* a2 = nullptr;
    CHECK(ws_get_on(w, 0, &a2, 50, fake_alloc, fake_free, fake_set) == 0 && a2 == a);
    CHECK(ws_get_on(w, 0, &c, 4000, fake_alloc, fake_free, fake_set) == 0 && g_live.at(c).second == 4000 && g_live.size() == 2);
    // switching device in the middle of an entry point is refused
    void* x = nullptr;
    CHECK(ws_get_on(w, 1, &x, 8, fake_alloc, fake_free, fake_set) == -1);
    w.used = 0;
    // the thread moved to device 1 (cclqr_set_device): the old blocks are freed on device 0, the new one lives on device 1,
    // and the runtime is left on device 1
    g_cur = 1;
    CHECK(ws_get_on(w, 1, &x, 64, fake_alloc, fake_free, fake_set) == 0);
    CHECK(g_wrong_device_frees == 0 && g_live.size() == 1 && g_live.at(x).first == 1 && g_cur == 1 && w.device == 1);
    w.used = 0;
    // and back
    g_cur = 0;
    CHECK(ws_get_on(w, 0, &a, 16, fake_alloc, fake_free, fake_set) == 0);
    CHECK(g_wrong_device_frees == 0 && g_live.size() == 1 && g_live.at(a).first == 0 && g_cur == 0);
    for (auto& blk : w.blocks) if (blk.first) fake_free(blk.first);      // cclqr_release_workspaces
    CHECK(g_live.empty() && g_wrong_device_frees == 0);
    printf("WS_CACHE_OK\n");
    return 0;
}
