// mock_rccl.cpp -- TEST INFRASTRUCTURE, not part of the product.
//
// RCCL refuses two ranks on one GPU ("Duplicate GPU detected"), and this build has one GPU.  The library's one-process-
// per-GPU code (csrc/par.hip with a communicator from smh_comm_create: the ranks' plan table, the window send / receive
// pairing, the in-place all-gather with its ragged tail, the CG folds) had therefore only ever run with ONE rank.  This file
// is a stand-in for the handful of RCCL entry points par.hip calls, LD_PRELOADed into the rank processes of
// tests/test_par_mock_ranks_gpu.py, so that 2..4 ranks SHARING the one device run exactly that code: same call sequence,
// same buffers and offsets, same group semantics -- with the bytes moved through POSIX shared memory instead of xGMI.
// It checks what real RCCL would deadlock or corrupt on: every receive must meet a send of the same size from that peer in
// the same group, collectives must arrive in the same order with the same counts on every rank, and a rank that waits 60 s
// for its peers fails instead of hanging.  It says nothing about RCCL's own behaviour or speed.
//
// Semantics: every call is executed at ncclGroupEnd (or at once outside a group): the streams of the queued operations are
// synchronised, outgoing bytes are published to this rank's mailbox, all ranks meet, incoming bytes are copied to the device,
// all ranks meet again.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

constexpr int kMaxRanks = 8;
constexpr size_t kMailbox = 24u << 20;  // bytes a rank can publish per round
constexpr int kMaxMsgs = 256;

struct Msg {  // one published item of a round
    int kind;  // 1 all-gather chunk, 2 broadcast payload, 3 p2p send, 4 all-reduce value
    int peer;  // p2p: destination
    uint64_t bytes, offset;
    uint64_t count;  // elements (checked against the receiver's expectation)
};

struct RankBox {
    std::atomic<uint32_t> n_msgs;
    Msg msgs[kMaxMsgs];
};

struct Shared {
    std::atomic<uint32_t> arrived;
    std::atomic<uint32_t> generation;
    std::atomic<uint32_t> failed;
    RankBox box[kMaxRanks];
    // followed by kMaxRanks mailboxes of kMailbox bytes
};

struct Op {
    int kind;  // 1 all-gather, 2 broadcast, 3 send, 4 recv, 5 all-reduce(max, f64)
    const void *send;
    void *recv;
    size_t count, elem;
    int peer;  // root / destination / source
    hipStream_t stream;
};

}  // namespace

struct ncclComm {
    int n = 0, rank = 0;
    Shared *sh = nullptr;
    char *mail = nullptr;
    size_t map_bytes = 0;
    char name[96] = {0};
};

namespace {

thread_local int g_depth = 0;
thread_local std::vector<std::pair<ncclComm *, Op>> g_queue;
ncclComm *g_only = nullptr;  // the process's communicator (the library creates one per rank process)

size_t elem_size(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        default: return 8;
    }
}

bool barrier(ncclComm *c) {
    Shared *s = c->sh;
    const uint32_t gen = s->generation.load();
    if (s->arrived.fetch_add(1) + 1 == (uint32_t)c->n) {
        s->arrived.store(0);
        s->generation.fetch_add(1);
        return s->failed.load() == 0;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (s->generation.load() == gen) {
        if (s->failed.load()) return false;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) {
            fprintf(stderr, "[mock rccl] rank %d: peers did not arrive within 60 s (a collective was not called by every rank?)\n", c->rank);
            s->failed.store(1);
            return false;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
    return s->failed.load() == 0;
}

ncclResult_t fail(ncclComm *c, const char *what) {
    fprintf(stderr, "[mock rccl] rank %d: %s\n", c ? c->rank : -1, what);
    if (c && c->sh) c->sh->failed.store(1);
    return ncclInvalidUsage;
}

ncclResult_t run_round(ncclComm *c, std::vector<Op> &ops) {
    for (const Op &o : ops)
        if (hipStreamSynchronize(o.stream) != hipSuccess) return fail(c, "stream synchronisation failed");
    RankBox &mine = c->sh->box[c->rank];
    char *my_mail = c->mail + (size_t)c->rank * kMailbox;
    uint64_t used = 0;
    uint32_t n_msgs = 0;
    auto publish = [&](int kind, int peer, const void *dev, size_t count, size_t elem) -> bool {
        const uint64_t bytes = (uint64_t)count * elem;
        if (n_msgs >= (uint32_t)kMaxMsgs || used + bytes > kMailbox) return false;
        if (bytes && hipMemcpy(my_mail + used, dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) return false;
        mine.msgs[n_msgs] = Msg{kind, peer, bytes, used, (uint64_t)count};
        used += (bytes + 15) & ~uint64_t(15);
        ++n_msgs;
        return true;
    };
    for (const Op &o : ops) {
        bool ok = true;
        if (o.kind == 1) ok = publish(1, -1, o.send, o.count, o.elem);
        else if (o.kind == 2) { if (o.peer == c->rank) ok = publish(2, -1, o.send, o.count, o.elem); }
        else if (o.kind == 3) ok = publish(3, o.peer, o.send, o.count, o.elem);
        else if (o.kind == 5) ok = publish(4, -1, o.send, o.count, o.elem);
        if (!ok) return fail(c, "mailbox too small for this round (test sizes only) or a device copy failed");
    }
    mine.n_msgs.store(n_msgs);
    if (!barrier(c)) return ncclSystemError;
    // consume: the k-th collective of this round meets the k-th collective message of every rank; the k-th receive from peer p
    // meets p's k-th send addressed to this rank
    std::vector<uint32_t> p2p_seen(c->n, 0);
    auto nth = [&](int r, int kind, int to, uint32_t k) -> const Msg * {
        const RankBox &b = c->sh->box[r];
        uint32_t seen = 0;
        for (uint32_t i = 0; i < b.n_msgs.load(); ++i) {
            const Msg &m = b.msgs[i];
            const bool match = to >= 0 ? (m.kind == 3 && m.peer == to) : (m.kind != 3);
            if (!match) continue;
            if (seen++ == k) return (to >= 0 || m.kind == kind) ? &m : nullptr;
        }
        return nullptr;
    };
    uint32_t coll_index = 0;
    for (const Op &o : ops) {
        if (o.kind == 1) {
            for (int r = 0; r < c->n; ++r) {
                const Msg *m = nth(r, 1, -1, coll_index);
                if (!m || m->count != o.count) return fail(c, "all-gather: a rank called another collective or another count here");
                if (m->bytes && hipMemcpy((char *)o.recv + (size_t)r * o.count * o.elem, c->mail + (size_t)r * kMailbox + m->offset, m->bytes, hipMemcpyHostToDevice) != hipSuccess)
                    return fail(c, "device copy failed");
            }
            ++coll_index;
        } else if (o.kind == 2) {
            // the root's k-th collective message; the other ranks published nothing for a broadcast, so count per rank
            uint32_t k_root = 0;
            for (const Op &q : ops) {
                if (&q == &o) break;
                if (q.kind == 1 || q.kind == 5 || (q.kind == 2 && q.peer == o.peer)) ++k_root;  // what the ROOT published before this op
            }
            const Msg *m = nth(o.peer, 2, -1, k_root);
            if (!m || m->count != o.count) return fail(c, "broadcast: the root did not publish a matching payload");
            if (o.peer != c->rank || o.recv != o.send)
                if (m->bytes && hipMemcpy(o.recv, c->mail + (size_t)o.peer * kMailbox + m->offset, m->bytes, hipMemcpyHostToDevice) != hipSuccess)
                    return fail(c, "device copy failed");
        } else if (o.kind == 4) {
            const Msg *m = nth(o.peer, 3, c->rank, p2p_seen[o.peer]++);
            if (!m) return fail(c, "receive without a matching send in the peer's group (real RCCL would hang here)");
            if (m->count != o.count) return fail(c, "receive and send disagree on the count (real RCCL would corrupt or hang)");
            if (m->bytes && hipMemcpy(o.recv, c->mail + (size_t)o.peer * kMailbox + m->offset, m->bytes, hipMemcpyHostToDevice) != hipSuccess)
                return fail(c, "device copy failed");
        } else if (o.kind == 5) {
            double best = 0;
            for (int r = 0; r < c->n; ++r) {
                const Msg *m = nth(r, 4, -1, coll_index);
                if (!m || m->count != 1 || m->bytes != 8) return fail(c, "all-reduce: only one f64 with max is mocked, on every rank");
                double v;
                memcpy(&v, c->mail + (size_t)r * kMailbox + m->offset, 8);
                best = (r == 0 || v > best) ? v : best;
            }
            if (hipMemcpy(o.recv, &best, 8, hipMemcpyHostToDevice) != hipSuccess) return fail(c, "device copy failed");
            ++coll_index;
        }
    }
    // every send of mine must have been received: the peers check their side; here: nobody was sent something it did not ask for
    for (int r = 0; r < c->n; ++r) {
        uint32_t to_me = 0;
        const RankBox &b = c->sh->box[r];
        for (uint32_t i = 0; i < b.n_msgs.load(); ++i) to_me += b.msgs[i].kind == 3 && b.msgs[i].peer == c->rank;
        if (to_me != p2p_seen[r]) return fail(c, "a peer sent a message this rank did not receive in the same group (real RCCL would hang)");
    }
    if (!barrier(c)) return ncclSystemError;
    return ncclSuccess;
}

ncclResult_t submit(ncclComm *c, const Op &o) {
    if (!c || !c->sh) return ncclInvalidArgument;
    if (g_depth > 0) {
        g_queue.emplace_back(c, o);
        return ncclSuccess;
    }
    std::vector<Op> one{o};
    return run_round(c, one);
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "smh_mock_%d_%lld", (int)getpid(),
             (long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int n, ncclUniqueId id, int rank) {
    if (n < 1 || n > kMaxRanks || rank < 0 || rank >= n) return ncclInvalidArgument;
    ncclComm *c = new ncclComm();
    c->n = n;
    c->rank = rank;
    id.internal[sizeof id.internal - 1] = 0;
    snprintf(c->name, sizeof c->name, "/%.80s", id.internal);
    for (char *p = c->name + 1; *p; ++p)
        if (*p == '/') *p = '_';
    c->map_bytes = sizeof(Shared) + (size_t)kMaxRanks * kMailbox;
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) { delete c; return ncclSystemError; }
    void *m = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { delete c; return ncclSystemError; }
    c->sh = static_cast<Shared *>(m);  // (a fresh segment is zero-filled: counters start at 0)
    c->mail = static_cast<char *>(m) + sizeof(Shared);
    if (!barrier(c)) { delete c; return ncclSystemError; }  // like the real call: collective
    g_only = c;
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t *, int, const int *) { return ncclInvalidUsage; }  // (one process, many devices: not what this mock is for)

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return ncclSuccess;
    if (g_only == c) g_only = nullptr;
    if (c->sh) {
        (void)barrier(c);
        munmap(c->sh, c->map_bytes);
        if (c->rank == 0) shm_unlink(c->name);
    }
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t c, int *count) { *count = c->n; return ncclSuccess; }
ncclResult_t ncclCommCuDevice(const ncclComm_t c, int *device) { (void)c; return hipGetDevice(device) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError; }
ncclResult_t ncclGetVersion(int *version) { *version = 0; return ncclSuccess; }  // (0: "this is the stand-in")

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error (mock)" : "mock rccl: see stderr"; }

ncclResult_t ncclGroupStart() {
    ++g_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    // One communicator per process in this mock's use.  A group without operations still meets the peers' round: every rank
    // of the library calls the same groups and some of them are empty (a block nobody exchanges with) -- real RCCL does
    // nothing for an empty group and pairs the others' sends and receives without a global meeting; the mock's rounds are global.
    ncclComm *c = g_queue.empty() ? g_only : g_queue.front().first;
    std::vector<Op> ops;
    for (auto &q : g_queue) ops.push_back(q.second);
    g_queue.clear();
    if (!c) return ncclSuccess;
    return run_round(c, ops);
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t s) {
    return submit(c, Op{1, send, recv, count, elem_size(t), -1, s});
}

ncclResult_t ncclBroadcast(const void *send, void *recv, size_t count, ncclDataType_t t, int root, ncclComm_t c, hipStream_t s) {
    return submit(c, Op{2, send, recv, count, elem_size(t), root, s});
}

ncclResult_t ncclSend(const void *send, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) {
    return submit(c, Op{3, send, nullptr, count, elem_size(t), peer, s});
}

ncclResult_t ncclRecv(void *recv, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t s) {
    return submit(c, Op{4, nullptr, recv, count, elem_size(t), peer, s});
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t s) {
    if (count != 1 || t != ncclDouble || op != ncclMax) return ncclInvalidUsage;
    return submit(c, Op{5, send, recv, count, 8, -1, s});
}

}  // extern "C"
