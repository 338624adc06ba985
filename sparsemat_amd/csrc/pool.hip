// pool.hip -- device memory of libsparsemat_hip.so goes through a small caching layer.
//
// Why: on MI355X / ROCm 7.2 a hipMalloc of a few GB takes 0.02-0.2 ms most of the time and 1.2-1.4 s every few calls, when
// the runtime has handed freed memory back to the driver in between (tools/dev/alloc_probe.py, profiles/r02_alloc_probe.log:
// every third transposition of BASELINE C2 took 1.2 s instead of 19 ms).  Every create / assemble / transpose / prod call and
// every lazily built plan allocates, so the library keeps what it frees:
//   * blocks of 1 MiB and more are rounded up to 2 MiB and, when freed, kept per device (at most SMH_POOL_MAX_BYTES, default
//     16 GiB per device; 0 switches the layer off); a later request is served by the smallest kept block that is large enough
//     and at most a quarter larger; smaller allocations go to the runtime directly (its own sub-allocator is fast for those);
//   * freeing keeps hipFree's meaning for the caller: the owning device is synchronised first, so nothing in flight can
//     still use a block when it is handed out again (the code base frees scratch right after enqueueing its last reader);
//   * an allocation the runtime refuses is retried after everything kept on that device has been returned to it;
//   * smh_pool_trim() returns everything kept to the runtime, smh_pool_stats() reports kept and live bytes.
// Every .hip file of the library reaches this through the hipMalloc / hipFree macros at the end of internal.hpp.
#define SMH_POOL_IMPL
#include "internal.hpp"

#include <cstdlib>
#include <map>
#include <mutex>
#include <unordered_map>

namespace smh {
namespace {

constexpr size_t kPoolMinBytes = 1ull << 20;
constexpr size_t kPoolGranule = 2ull << 20;
constexpr int kPoolDevices = 64;

struct Live { size_t bytes; int device; };

struct Pool {
    std::mutex mu;
    std::unordered_map<void *, Live> live;              // pooled blocks in use
    std::multimap<size_t, void *> kept[kPoolDevices];   // freed blocks, by size
    size_t kept_bytes[kPoolDevices] = {};
    size_t live_bytes = 0;
    size_t cap = 16ull << 30;
    Pool() {
        if (const char *e = getenv("SMH_POOL_MAX_BYTES")) cap = (size_t)strtoull(e, nullptr, 10);
    }
};

thread_local long long t_net_bytes = 0;

Pool &pool() {
    static Pool *p = new Pool();  // (never destroyed: frees may still arrive from static destructors of the host)
    return *p;
}

// returns everything kept on `device` to the runtime (lock held)
void trim_locked(Pool &P, int device) {
    for (auto &kv : P.kept[device]) (void)hipFree(kv.second);
    P.kept[device].clear();
    P.kept_bytes[device] = 0;
}

}  // namespace

hipError_t pool_malloc(void **out, size_t bytes) {
    if (!out) return hipErrorInvalidValue;
    Pool &P = pool();
    if (P.cap == 0 || bytes < kPoolMinBytes) return hipMalloc(out, bytes);
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    if (device < 0 || device >= kPoolDevices) return hipMalloc(out, bytes);
    const size_t want = (bytes + kPoolGranule - 1) / kPoolGranule * kPoolGranule;
    std::lock_guard<std::mutex> g(P.mu);
    auto it = P.kept[device].lower_bound(want);
    if (it != P.kept[device].end() && it->first <= want + want / 4) {
        *out = it->second;
        P.live.emplace(it->second, Live{it->first, device});
        P.live_bytes += it->first;
        t_net_bytes += (long long)it->first;
        P.kept_bytes[device] -= it->first;
        P.kept[device].erase(it);
        return hipSuccess;
    }
    e = hipMalloc(out, want);
    if (e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation) {
        (void)hipGetLastError();
        trim_locked(P, device);
        e = hipMalloc(out, want);
    }
    if (e != hipSuccess) return e;
    P.live.emplace(*out, Live{want, device});
    P.live_bytes += want;
    t_net_bytes += (long long)want;
    return hipSuccess;
}

hipError_t pool_free(void *p) {
    if (!p) return hipSuccess;
    Pool &P = pool();
    Live blk{0, 0};
    {
        std::lock_guard<std::mutex> g(P.mu);
        auto it = P.live.find(p);
        if (it == P.live.end()) {
            // not one of the pooled blocks (small, foreign, or allocated with the layer off): the runtime's own free
        } else {
            blk = it->second;
            P.live.erase(it);
            P.live_bytes -= blk.bytes;
            t_net_bytes -= (long long)blk.bytes;
        }
    }
    if (blk.bytes == 0) return hipFree(p);
    // hipFree's contract: nothing on the device still uses the block afterwards
    int cur = 0;
    hipError_t e = hipGetDevice(&cur);
    if (e == hipSuccess && cur != blk.device) e = hipSetDevice(blk.device);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (cur != blk.device) (void)hipSetDevice(cur);
    if (e != hipSuccess) { (void)hipFree(p); return e; }
    std::lock_guard<std::mutex> g(P.mu);
    if (P.kept_bytes[blk.device] + blk.bytes > P.cap) return hipFree(p);
    P.kept[blk.device].emplace(blk.bytes, p);
    P.kept_bytes[blk.device] += blk.bytes;
    return hipSuccess;
}

long long pool_thread_net_bytes() { return t_net_bytes; }

}  // namespace smh

extern "C" {

int smh_pool_trim(void) {
    smh::Pool &P = smh::pool();
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) { (void)hipGetLastError(); return SMH_OK; }  // no device: nothing was ever kept
    std::lock_guard<std::mutex> g(P.mu);
    for (int d = 0; d < smh::kPoolDevices; ++d) {
        if (P.kept[d].empty()) continue;
        (void)hipSetDevice(d);
        smh::trim_locked(P, d);
    }
    (void)hipSetDevice(cur);
    return SMH_OK;
}

int smh_pool_stats(size_t *kept_bytes_out, size_t *live_bytes_out) {
    smh::Pool &P = smh::pool();
    std::lock_guard<std::mutex> g(P.mu);
    size_t kept = 0;
    for (int d = 0; d < smh::kPoolDevices; ++d) kept += P.kept_bytes[d];
    if (kept_bytes_out) *kept_bytes_out = kept;
    if (live_bytes_out) *live_bytes_out = P.live_bytes;
    return SMH_OK;
}

}  // extern "C"
