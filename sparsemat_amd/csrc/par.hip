// par.hip -- SparseMatPar<SparseMatCRS<T,u32>> behind the C ABI: row blocks on the GPUs of one node, the exchange of
// the dense vector INSIDE the library (RCCL over xGMI, or direct peer reads), device-resident CG.
//
// Reference (sparsemat_par.rs:12-35, 86-107): n_blocks sub-matrices of R = max_n_rows / n_blocks local rows each, block b
// owning the global rows [b R, (b+1) R) with local row ids and GLOBAL column ids; `mvp` is the serial trait default
// walking iter_row(row) -> sub_matrices[block].iter_row(local).  Its commented-out mvp_par (:37-68) is the design this
// file completes: every block multiplies against the shared rhs, the results go to offset b * R, the pieces are
// gathered.  get_block_and_row_id clamps the block id to n_blocks (:32) -- one past the last block -- so a row beyond
// n_blocks R panics; here, as SURVEY 8b prescribes, the LAST block takes the remainder (clamp to n_blocks - 1).
//
// Device formulation.  Block b is an ordinary smh_crs on its device (all kernel families apply) with a stream of its
// own.  A distributed vector (smh_par_vec) is one FULL-LENGTH buffer per local block; block b owns [b R, (b+1) R) of it.
// y = A x: every block writes its slice of y, then ONE exchange makes y usable as the next x:
//   ALLGATHER  in-place ncclAllGather at recvbuff + b R (+ ncclBroadcast of a ragged last block's tail)
//   WINDOW     block q receives exactly [lo_q, hi_q] \ own slice (its column interval, a create-time statistic):
//              grouped ncclSend / ncclRecv on slices of the vector itself -- a banded matrix moves a halo
// Backends: RCCL (ncclCommInitAll over the blocks' devices in one process; ncclCommInitRank with one process per GPU,
// smh_comm_*), or PEER for a one-process handle: one pull kernel per block reads the peers' slices straight through
// peer access (all xGMI links of the destination at once; hipMemcpyPeerAsync where peer access is unavailable).  PEER is
// also what runs when several blocks share a device, which is how the whole path -- partition, plans, exchanges, folds,
// solver -- is exercised on a one-GPU box.
// CG (linearsolver.rs:27-61): x, r, p, Ap distributed by rows; per iteration one WINDOW exchange of p, the local SpMV,
// and two cross-block folds whose operands stay on the devices: each block reduces its rows to one value, the values
// meet (PEER: slots in pinned host memory every device maps, ordered by events; RCCL: a 1-element ncclAllGather) and
// every block folds the same n_blocks values with the same fixed tree -> identical alpha / beta / stop decision on all
// blocks, no host round trip, bitwise reproducible.  The host polls the stop flag every check_every iterations; the
// gated kernels of later iterations are no-ops (cg.hip), so x is exactly the x of the iteration that converged.
#include "internal.hpp"

#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace smh;

namespace smh {
// cg.hip
size_t cg_scalars_bytes(int dtype);
void cg_read_scalars(int dtype, const void *host_copy, int *converged, uint64_t *iters, double *rr);
int cg_fold(int dtype, const void *partials, uint32_t count, void *out, hipStream_t s);
int cg_par_init(int dtype, void *sc, double tol, size_t iter_max, hipStream_t s);
int cg_par_set_rr(int dtype, void *sc, const void *vals, uint32_t nb, hipStream_t s);
int cg_par_update(int dtype, const void *sc_in, void *sc_out, const void *pap_vals, uint32_t nb, void *r, const void *ap, size_t n, void *partials,
                  uint32_t *count_out, hipStream_t s);
int cg_par_p(int dtype, const void *sc_in, void *sc_out, const void *rr_vals, uint32_t nb, void *p, const void *r, void *x, size_t n, hipStream_t s);
}  // namespace smh

#define SMH_NCCL(call)                                                                                    \
    do {                                                                                                  \
        ncclResult_t r__ = (call);                                                                        \
        if (r__ != ncclSuccess)                                                                           \
            return ::smh::fail(SMH_ERR_COMM, "%s failed: %s (%s:%d)", #call, ncclGetErrorString(r__), __FILE__, __LINE__); \
    } while (0)

struct smh_comm {
    ncclComm_t comm = nullptr;
    int n_ranks = 1, rank = 0, device = 0;
    hipStream_t s = nullptr;  // for the host-facing helpers (barrier, max)
    void *d_scratch = nullptr;
};

namespace {

struct ParBlock {
    size_t index = 0;             // global block id
    int device = 0;
    smh_crs *m = nullptr;
    bool owns_m = true;
    size_t r0 = 0, r1 = 0;        // global rows [r0, r1)
    hipStream_t s = nullptr;
    hipStream_t sx = nullptr;       // the exchange's stream when it runs beside the interior rows' product
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;  // s -> sx (what the exchange needs is written), sx -> s (the exchange is done)
    // INTERIOR rows [in0, in1) (local): no other block references them and they reference no other block's columns, so their
    // product needs nothing from the exchange and gives nothing to it (in1 <= in0: none)
    size_t in0 = 0, in1 = 0;
    hipEvent_t ev_slice = nullptr;  // own slice of the vector being exchanged is written
    hipEvent_t ev_done = nullptr;   // this block's pulls from its peers are complete
    hipEvent_t ev_red[2] = {nullptr, nullptr};  // its value of fold slot 0 / 1 is written
    hipEvent_t ev_all[2] = {nullptr, nullptr};  // (block 0) every block's value of the slot is written
    ncclComm_t comm = nullptr;    // RCCL backend: this block's rank of the communicator
    // staging of the host-vector API (smh_par_spmv)
    void *d_x = nullptr, *d_y = nullptr;
    // CG state (first solve)
    void *d_r = nullptr, *d_ap = nullptr, *d_partials = nullptr, *d_sc = nullptr;
    void *d_sc2 = nullptr;  // the scalars are double-buffered: the alpha / update launch reads d_sc and writes d_sc2, the beta / p launch back (cg.hip)
    void *d_dotp = nullptr;  // the SpMV's p.Ap partials (one per 256-row tile), folded by launch_fold2 through d_partials
    size_t dotp_cap = 0;
    void *d_redv = nullptr;       // RCCL backend: 2 x n_blocks values (fold slots)
};

constexpr int kMaxPull = 32;  // segments per pull-kernel launch
struct PullArgs {
    const uint32_t *src[kMaxPull];
    uint64_t w0[kMaxPull], w1[kMaxPull];  // 32-bit words [w0, w1) of the full-length buffers
    int n;
};

// dst[w] = src_k[w] for every word w of every segment k: the peers' slices are read where they lie (peer access over
// xGMI; plain device memory when the "peer" block shares the device).  16-byte accesses where both sides allow.
__global__ void __launch_bounds__(kBlock) k_peer_pull(uint32_t *__restrict__ dst, PullArgs a) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * blockDim.x;
    for (int k = 0; k < a.n; ++k) {
        const uint32_t *__restrict__ src = a.src[k];
        const uint64_t w0 = a.w0[k], w1 = a.w1[k];
        // words [w0, head_end) one by one, [head_end, body_end) as 16-byte pieces, [body_end, w1) one by one.  A word has
        // the same index in both buffers, so the pieces are aligned in both iff the base pointers are equally misaligned.
        uint64_t head_end = w1, body_end = w1;
        const uintptr_t da = reinterpret_cast<uintptr_t>(dst), sa = reinterpret_cast<uintptr_t>(src);
        if (((da ^ sa) & 15u) == 0) {
            const uint64_t mis = (da >> 2) & 3u;  // words by which word 0 is past a 16-byte boundary
            const uint64_t a0 = ((w0 + mis + 3) & ~uint64_t(3)) - mis;
            const uint64_t up = (w1 + mis) & ~uint64_t(3);
            if (up >= mis && a0 < up - mis) { head_end = a0; body_end = up - mis; }
        }
        for (uint64_t w = w0 + tid; w < head_end; w += nthreads) dst[w] = src[w];
        const uint64_t pieces = (body_end - head_end) / 4;
        for (uint64_t q = tid; q < pieces; q += nthreads)
            reinterpret_cast<u32x4 *>(dst + head_end)[q] = reinterpret_cast<const u32x4 *>(src + head_end)[q];
        for (uint64_t w = body_end + tid; w < w1; w += nthreads) dst[w] = src[w];
    }
}

// flag[t] = 1 when a row of the 256-row tile t references a column outside [c0, c1) (the block's own slice of the vector)
__global__ void __launch_bounds__(kBlock) k_par_remote_tiles(const uint32_t *__restrict__ off, const uint32_t *__restrict__ col, uint64_t n_rows,
                                                             uint32_t c0, uint32_t c1, uint8_t *__restrict__ flag) {
    const uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    bool remote = false;
    if (r < n_rows)
        for (uint64_t k = off[r], e = off[r + 1]; k < e; ++k) {
            const uint32_t c = col[k];
            remote |= c < c0 || c >= c1;
        }
    const bool any = __syncthreads_or(remote);
    if (threadIdx.x == 0) flag[blockIdx.x] = any;
}

}  // namespace

// One ISSUING HOST THREAD per local block (one-process handles with several blocks).  The solver and the product step are
// sequences of PHASES -- "every block does X on its streams" -- separated by the points where one block's stream must wait for an
// event another block has recorded (the record has to be issued, on the host, before the wait is).  Round 3 issued every phase from
// the caller's thread: ~30 runtime calls per block and CG iteration, 0.84-0.92 ms per iteration with 8 blocks on one device against
// 0.27 ms of kernels.  Now block k's calls are issued by thread k (block 0's by the caller), all at once, and the phases meet at a
// host barrier; the streams, events, kernels and their order per block are unchanged, so every result is bit for bit the same.
// (An earlier attempt to get rid of the calls -- ONE hipGraph captured across all blocks' streams -- died inside the runtime during
// capture from 4 blocks on; this needs nothing of the graph machinery.)
struct ParPool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<uint64_t> gen{0};       // phases handed out so far
    std::atomic<uint32_t> pending{0};   // workers still inside the current phase
    std::atomic<uint32_t> sleepers{0};  // workers blocked on cv (a phase start wakes them)
    std::atomic<bool> stop{false};
    const std::function<int(size_t)> *fn = nullptr;
    std::vector<int> rc;
    std::vector<std::string> msg;
};

struct smh_par {
    int dtype = SMH_F32;
    size_t n_rows = 0, n_cols = 0, rows_per_block = 0, n_blocks = 0;
    std::vector<ParBlock> b;        // LOCAL blocks: all of them (one process), or this rank's (one process per GPU)
    smh_comm *rank_comm = nullptr;  // one process per GPU (borrowed)
    int backend = SMH_PAR_BACKEND_PEER;  // resolved
    bool comms_ready = false;
    bool overlap = true;            // window exchanges run beside the interior rows' product (smh_par_set_overlap; SMH_PAR_OVERLAP=0)
    // column intervals of ALL blocks: block q references [lo[q], hi[q]] when needs[q]
    std::vector<uint32_t> lo, hi;
    std::vector<uint8_t> needs;
    std::vector<size_t> split;      // empty: the reference's partition (R rows per block); else n_blocks + 1 row boundaries
    std::vector<uint8_t> peer_ok;   // [dst local * n_local + src local]: dst's device can read src's memory directly
    void *h_red = nullptr;          // PEER backend: 2 x n_blocks fold slots in pinned host memory (all devices map it)
    void *h_sc = nullptr;           // pinned copy of block 0's CG scalars
    smh_par_vec *cg_p = nullptr;    // the search direction of the solver (full-length per block)
    smh_par_vec *io_b = nullptr, *io_x = nullptr;  // staging of the host-vector solve
    ParPool *pool = nullptr;        // issuing threads (lazy; blocks 1 .. n_local - 1)
    int use_threads = -1;           // -1 automatic (= off unless SMH_PAR_THREADS=1: threads_wanted() says why), 0 never, 1 always
};

struct smh_par_vec {
    smh_par *p = nullptr;
    size_t n = 0;
    std::vector<void *> d;  // per local block, n entries on its device
};

namespace {

int use(const ParBlock &blk) {
    // (called before every runtime call of a block; a thread that already is on the block's device -- always, when the blocks share
    // one -- skips the runtime's device switch)
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur == blk.device) return SMH_OK;
    SMH_HIP(hipSetDevice(blk.device));
    return SMH_OK;
}

struct DeviceGuard {
    int prev = 0;
    DeviceGuard() { (void)hipGetDevice(&prev); }
    ~DeviceGuard() { (void)hipSetDevice(prev); }
};

int sync_all(smh_par *p) {
    for (ParBlock &blk : p->b) {
        SMH_TRY(use(blk));
        SMH_HIP(hipStreamSynchronize(blk.s));
        if (blk.sx) SMH_HIP(hipStreamSynchronize(blk.sx));
    }
    return SMH_OK;
}

bool threads_wanted(const smh_par *p) {
    if (p->b.size() < 2) return false;
    if (p->use_threads >= 0) return p->use_threads != 0;
    // automatic = OFF: measured with 8 blocks on one device (the only multi-block set-up this build has had), the runtime serialises
    // its callers -- 8 threads issue an iteration of the solver in 0.37 ms where one thread takes 0.47 ms, and the interleaving they
    // produce costs the device more than that saves (0.59 against 0.475 ms per iteration, profiles/r04_par_cg_threads.log).  On a
    // node, a device per block, the runtime's per-device queues may make it pay: SMH_PAR_THREADS=1 / smh_par_set_threads(p, 1)
    static const bool on = getenv("SMH_PAR_THREADS") && atoi(getenv("SMH_PAR_THREADS")) != 0;  // tuning knob
    return on;
}

void pool_worker(smh_par *p, size_t k) {
    ParPool &P = *p->pool;
    (void)hipSetDevice(p->b[k].device);
    uint64_t seen = 0;
    for (;;) {
        // wait for the next phase: spin while the solver is running (phases follow each other within microseconds), sleep otherwise
        unsigned spins = 0;
        while (P.gen.load(std::memory_order_acquire) == seen && !P.stop.load(std::memory_order_acquire)) {
            if (++spins < 200000u) {
                __builtin_ia32_pause();
            } else {
                std::unique_lock<std::mutex> lk(P.mu);
                P.sleepers.fetch_add(1, std::memory_order_acq_rel);
                P.cv.wait(lk, [&] { return P.gen.load(std::memory_order_acquire) != seen || P.stop.load(std::memory_order_acquire); });
                P.sleepers.fetch_sub(1, std::memory_order_acq_rel);
                spins = 0;
            }
        }
        if (P.stop.load(std::memory_order_acquire)) return;
        seen = P.gen.load(std::memory_order_acquire);
        const int rc = (*P.fn)(k);
        P.rc[k] = rc;
        if (rc != SMH_OK) P.msg[k] = smh_last_error();  // (the message is thread-local: hand it to the caller)
        P.pending.fetch_sub(1, std::memory_order_acq_rel);
    }
}

void pool_destroy(smh_par *p) {
    if (!p->pool) return;
    {
        std::lock_guard<std::mutex> lk(p->pool->mu);
        p->pool->stop.store(true, std::memory_order_release);
    }
    p->pool->cv.notify_all();
    for (std::thread &t : p->pool->th) t.join();
    delete p->pool;
    p->pool = nullptr;
}

// fn(k) for every local block k -- concurrently, one thread per block, when the handle has its issuing threads; returns when all
// have returned (a HOST barrier: what fn issued is on the streams, not necessarily executed), with the first failure
int par_for(smh_par *p, const std::function<int(size_t)> &fn) {
    const size_t nl = p->b.size();
    if (!threads_wanted(p)) {
        for (size_t k = 0; k < nl; ++k) SMH_TRY(fn(k));
        return SMH_OK;
    }
    if (!p->pool) {
        p->pool = new (std::nothrow) ParPool();
        if (!p->pool) return fail(SMH_ERR_OOM, "host allocation failed");
        p->pool->rc.assign(nl, SMH_OK);
        p->pool->msg.assign(nl, std::string());
        for (size_t k = 1; k < nl; ++k) p->pool->th.emplace_back(pool_worker, p, k);
    }
    ParPool &P = *p->pool;
    P.fn = &fn;
    P.pending.store((uint32_t)(nl - 1), std::memory_order_release);
    {
        // (the generation changes under the lock a sleeper checks it under: no wake-up is lost)
        std::lock_guard<std::mutex> lk(P.mu);
        P.gen.fetch_add(1, std::memory_order_acq_rel);
    }
    if (P.sleepers.load(std::memory_order_acquire)) P.cv.notify_all();
    const int rc0 = fn(0);  // block 0 is the caller's
    while (P.pending.load(std::memory_order_acquire)) __builtin_ia32_pause();
    if (rc0 != SMH_OK) return rc0;
    for (size_t k = 1; k < nl; ++k)
        if (P.rc[k] != SMH_OK) return fail(P.rc[k], "%s", P.msg[k].c_str());
    return SMH_OK;
}

// Which rows a block owns.  split == NULL: the reference's arithmetic -- R = n_rows / n_blocks (sparsemat_par.rs:21), block k owns
// [k R, (k + 1) R), the last one also the remainder.  split != NULL (n_blocks + 1 ascending rows, split[0] = 0, split[n_blocks] =
// n_rows): block k owns [split[k], split[k + 1]) -- SURVEY 8e's "nnz-balanced split points ... for skewed matrices".
struct Part {
    size_t n_blocks, n_rows;
    const size_t *split;
};

void block_rows(const Part &pt, size_t k, size_t *r0, size_t *r1) {
    if (pt.split) { *r0 = pt.split[k]; *r1 = pt.split[k + 1]; return; }
    const size_t rpb = pt.n_rows / pt.n_blocks;  // sparsemat_par.rs:21
    *r0 = k * rpb;
    *r1 = k + 1 == pt.n_blocks ? pt.n_rows : (k + 1) * rpb;  // the last block takes the remainder
}

// the part of block src's slice that block q's columns reference (empty: *a == *e == 0)
void recv_range(const Part &pt, const uint8_t *needs, const uint32_t *lo, const uint32_t *hi, size_t q, size_t src, size_t *a, size_t *e) {
    *a = *e = 0;
    if (q == src || !needs[q]) return;
    size_t s0, s1;
    block_rows(pt, src, &s0, &s1);
    const size_t x0 = lo[q] > s0 ? lo[q] : s0, x1 = (size_t)hi[q] + 1 < s1 ? (size_t)hi[q] + 1 : s1;
    if (x0 < x1) { *a = x0; *e = x1; }
}

void plan_summary(const Part &pt, const uint8_t *needs, const uint32_t *lo, const uint32_t *hi, int *auto_mode, size_t *max_recv) {
    size_t worst = 0;
    for (size_t q = 0; q < pt.n_blocks; ++q) {
        size_t got = 0;
        for (size_t src = 0; src < pt.n_blocks; ++src) {
            size_t a, e;
            recv_range(pt, needs, lo, hi, q, src, &a, &e);
            got += e - a;
        }
        worst = got > worst ? got : worst;
    }
    if (max_recv) *max_recv = worst;
    if (auto_mode) *auto_mode = pt.n_blocks <= 1 ? SMH_EXCHANGE_NONE : (worst * 2 < pt.n_rows ? SMH_EXCHANGE_WINDOW : SMH_EXCHANGE_ALLGATHER);
}

Part part_of(const smh_par *p) { return Part{p->n_blocks, p->n_rows, p->split.empty() ? nullptr : p->split.data()}; }

// One block has nobody to exchange with.  SMH_PAR_EXCHANGE_SINGLE=1 (test knob) still sends a lone RANK through the RCCL
// calls -- an in-place all-gather of one rank, an empty send/receive group, the 1-element gathers of the folds -- so that
// a one-GPU box executes the very call sequence the ranks of a node run.
bool lone_block_skips(const smh_par *p) {
    if (p->n_blocks > 1) return false;
    static const bool forced = getenv("SMH_PAR_EXCHANGE_SINGLE") && atoi(getenv("SMH_PAR_EXCHANGE_SINGLE")) != 0;
    return !(forced && p->rank_comm);
}

int resolve_mode(const smh_par *p, int mode, int *out) {
    if (mode < SMH_EXCHANGE_NONE || mode > SMH_EXCHANGE_AUTO) return fail(SMH_ERR_INVALID, "unknown exchange mode %d", mode);
    if (lone_block_skips(p)) { *out = SMH_EXCHANGE_NONE; return SMH_OK; }
    if (p->n_blocks <= 1) { *out = mode == SMH_EXCHANGE_AUTO ? SMH_EXCHANGE_ALLGATHER : mode; return SMH_OK; }
    if (mode == SMH_EXCHANGE_AUTO) {
        int m = SMH_EXCHANGE_ALLGATHER;
        // a window is addressed by column AND owned by row: only meaningful when the vector is both (square matrix)
        if (p->n_rows == p->n_cols) plan_summary(part_of(p), p->needs.data(), p->lo.data(), p->hi.data(), &m, nullptr);
        *out = m;
        return SMH_OK;
    }
    if (mode == SMH_EXCHANGE_WINDOW && p->n_rows != p->n_cols)
        return fail(SMH_ERR_INVALID, "a window exchange needs a square matrix (%zu x %zu)", p->n_rows, p->n_cols);
    *out = mode;
    return SMH_OK;
}

ncclDataType_t nccl_type(int dtype) { return dtype == SMH_F64 ? ncclDouble : ncclFloat; }

ncclComm_t comm_of(const smh_par *p, const ParBlock &blk) { return p->rank_comm ? p->rank_comm->comm : blk.comm; }

// RCCL backend of a one-process handle: one communicator rank per block, rank id = block id
int ensure_comms(smh_par *p) {
    if (p->comms_ready || p->rank_comm || p->n_blocks <= 1) return SMH_OK;
    std::vector<int> devs(p->b.size());
    for (size_t k = 0; k < p->b.size(); ++k) devs[k] = p->b[k].device;
    for (size_t i = 0; i < devs.size(); ++i)
        for (size_t j = i + 1; j < devs.size(); ++j)
            if (devs[i] == devs[j])
                return fail(SMH_ERR_INVALID, "RCCL backend: blocks %zu and %zu share device %d (one device per block needed; use the PEER backend)", i, j, devs[i]);
    std::vector<ncclComm_t> comms(p->b.size(), nullptr);
    SMH_NCCL(ncclCommInitAll(comms.data(), (int)p->b.size(), devs.data()));
    for (size_t k = 0; k < p->b.size(); ++k) p->b[k].comm = comms[k];
    p->comms_ready = true;
    return SMH_OK;
}

// ---- the exchange -------------------------------------------------------------------------------------------------
// side: on the blocks' exchange streams (sx) instead of their main ones -- the caller forks and joins them
hipStream_t xs(const ParBlock &blk, bool side) { return side ? blk.sx : blk.s; }

int exchange_rccl(smh_par *p, smh_par_vec *v, int mode, bool side) {
    SMH_TRY(ensure_comms(p));
    const size_t vs = dtype_size(p->dtype), nb = p->n_blocks, R = p->rows_per_block;
    const ncclDataType_t dt = nccl_type(p->dtype);
    if (mode == SMH_EXCHANGE_ALLGATHER && !p->split.empty()) {
        // blocks of unequal size (a split table): every slice is broadcast from its owner, all of them in ONE group
        SMH_NCCL(ncclGroupStart());
        for (size_t k = 0; k < p->b.size(); ++k) {
            ParBlock &blk = p->b[k];
            SMH_TRY(use(blk));
            for (size_t j = 0; j < nb; ++j) {
                const size_t a = p->split[j], e = p->split[j + 1];
                if (e > a) SMH_NCCL(ncclBroadcast((char *)v->d[k] + a * vs, (char *)v->d[k] + a * vs, e - a, dt, (int)j, comm_of(p, blk), xs(blk, side)));
            }
        }
        SMH_NCCL(ncclGroupEnd());
        return SMH_OK;
    }
    if (mode == SMH_EXCHANGE_ALLGATHER) {
        // in place: block b's slice already sits at b R of every gathered vector (count R from every rank) ...
        SMH_NCCL(ncclGroupStart());
        for (size_t k = 0; k < p->b.size(); ++k) {
            ParBlock &blk = p->b[k];
            SMH_TRY(use(blk));
            char *buf = (char *)v->d[k];
            SMH_NCCL(ncclAllGather(buf + blk.index * R * vs, buf, R, dt, comm_of(p, blk), xs(blk, side)));
        }
        SMH_NCCL(ncclGroupEnd());
        // ... and the last block's remainder rows [n_blocks R, n_rows) follow as one broadcast from their owner
        const size_t rem = p->n_rows - nb * R;
        if (rem) {
            SMH_NCCL(ncclGroupStart());
            for (size_t k = 0; k < p->b.size(); ++k) {
                ParBlock &blk = p->b[k];
                SMH_TRY(use(blk));
                char *tail = (char *)v->d[k] + nb * R * vs;
                SMH_NCCL(ncclBroadcast(tail, tail, rem, dt, (int)(nb - 1), comm_of(p, blk), xs(blk, side)));
            }
            SMH_NCCL(ncclGroupEnd());
        }
        return SMH_OK;
    }
    // WINDOW: every piece is a contiguous slice of the vector itself -- sends read the block's own slice, receives land
    // in the peers' slices (disjoint from it): ONE grouped launch of point-to-point operations, no staging, no packing
    SMH_NCCL(ncclGroupStart());
    for (size_t k = 0; k < p->b.size(); ++k) {
        ParBlock &blk = p->b[k];
        SMH_TRY(use(blk));
        char *buf = (char *)v->d[k];
        for (size_t q = 0; q < nb; ++q) {
            if (q == blk.index) continue;
            size_t a, e;
            recv_range(part_of(p), p->needs.data(), p->lo.data(), p->hi.data(), blk.index, q, &a, &e);  // what I need of q's slice
            if (a < e) SMH_NCCL(ncclRecv(buf + a * vs, e - a, dt, (int)q, comm_of(p, blk), xs(blk, side)));
            recv_range(part_of(p), p->needs.data(), p->lo.data(), p->hi.data(), q, blk.index, &a, &e);  // what q needs of mine
            if (a < e) SMH_NCCL(ncclSend(buf + a * vs, e - a, dt, (int)q, comm_of(p, blk), xs(blk, side)));
        }
    }
    SMH_NCCL(ncclGroupEnd());
    return SMH_OK;
}

// The PEER exchange in three per-block steps; between two of them every block's calls of the step before must have been ISSUED
// (a wait names an event another block records).  pulled: [q * n_local + src] = q read from src (filled by step 2, read by step 3).
// 1. block k's slice is complete once what its stream holds so far has run
int peer_mark(smh_par *p, size_t k, bool side) {
    ParBlock &blk = p->b[k];
    SMH_TRY(use(blk));
    SMH_HIP(hipEventRecord(blk.ev_slice, xs(blk, side)));
    return SMH_OK;
}
// 2. block qi pulls what it needs from the owners' buffers
int peer_pull(smh_par *p, smh_par_vec *v, int mode, size_t qi, bool side, uint8_t *pulled) {
    const size_t vs = dtype_size(p->dtype), nl = p->b.size();
    const size_t wpe = vs / 4;  // 32-bit words per entry
    ParBlock &q = p->b[qi];
    SMH_TRY(use(q));
    PullArgs args;
    args.n = 0;
    uint64_t words = 0;
    auto flush = [&]() -> int {
        if (args.n == 0) return SMH_OK;
        uint64_t blocks = (words / 4 + kBlock - 1) / kBlock;
        blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
        hipLaunchKernelGGL(k_peer_pull, dim3((unsigned)blocks), dim3(kBlock), 0, xs(q, side), (uint32_t *)v->d[qi], args);
        SMH_HIP(hipGetLastError());
        args.n = 0;
        words = 0;
        return SMH_OK;
    };
    for (size_t si = 0; si < nl; ++si) {
        pulled[qi * nl + si] = 0;
        if (si == qi) continue;
        ParBlock &src = p->b[si];
        size_t a = src.r0, e = src.r1;
        if (mode == SMH_EXCHANGE_WINDOW)
            recv_range(part_of(p), p->needs.data(), p->lo.data(), p->hi.data(), q.index, src.index, &a, &e);
        if (a >= e) continue;
        pulled[qi * nl + si] = 1;
        SMH_HIP(hipStreamWaitEvent(xs(q, side), src.ev_slice, 0));
        if (p->peer_ok[qi * nl + si]) {
            args.src[args.n] = (const uint32_t *)v->d[si];
            args.w0[args.n] = (uint64_t)a * wpe;
            args.w1[args.n] = (uint64_t)e * wpe;
            words += (uint64_t)(e - a) * wpe;
            if (++args.n == kMaxPull) SMH_TRY(flush());
        } else {  // no direct access between the two devices: the runtime stages the copy
            SMH_HIP(hipMemcpyPeerAsync((char *)v->d[qi] + a * vs, q.device, (const char *)v->d[si] + a * vs, src.device, (e - a) * vs, xs(q, side)));
        }
    }
    SMH_TRY(flush());
    SMH_HIP(hipEventRecord(q.ev_done, xs(q, side)));
    return SMH_OK;
}
// 3. block si does not overwrite its slice while a peer may still be reading it
int peer_guard(smh_par *p, size_t si, bool side, const uint8_t *pulled) {
    const size_t nl = p->b.size();
    ParBlock &src = p->b[si];
    SMH_TRY(use(src));
    for (size_t qi = 0; qi < nl; ++qi)
        if (pulled[qi * nl + si]) SMH_HIP(hipStreamWaitEvent(xs(src, side), p->b[qi].ev_done, 0));
    return SMH_OK;
}

int exchange_peer(smh_par *p, smh_par_vec *v, int mode, bool side) {
    const size_t nl = p->b.size();
    std::vector<uint8_t> pulled(nl * nl, 0);
    SMH_TRY(par_for(p, [&](size_t k) { return peer_mark(p, k, side); }));
    SMH_TRY(par_for(p, [&](size_t k) { return peer_pull(p, v, mode, k, side, pulled.data()); }));
    return par_for(p, [&](size_t k) { return peer_guard(p, k, side, pulled.data()); });
}

int exchange(smh_par *p, smh_par_vec *v, int mode, bool side = false) {
    int m = SMH_EXCHANGE_NONE;
    SMH_TRY(resolve_mode(p, mode, &m));
    if (m == SMH_EXCHANGE_NONE) return SMH_OK;
    if (v->n != p->n_rows)
        return fail(SMH_ERR_DIM_MISMATCH, "exchange: the vector has %zu entries, the partition owns %zu rows", v->n, p->n_rows);
    return p->backend == SMH_PAR_BACKEND_RCCL ? exchange_rccl(p, v, m, side) : exchange_peer(p, v, m, side);
}

// ---- the exchange beside the product ("overlap the gather with the next block of rows") ----------------------
// The blocks are independent (sparsemat_par.rs:54-64: every block multiplies on its own, results at b R), so WHICH of a block's
// rows are multiplied first is free of semantics.  A WINDOW exchange moves only what blocks reference of each other; a block's
// interior rows [in0, in1) neither feed it nor need it.  So the exchange runs on a second stream per block while the interior
// rows are multiplied: fork() after what the exchange reads is written, join() before what it writes is read.
int fork_one(ParBlock &blk) {
    SMH_TRY(use(blk));
    SMH_HIP(hipEventRecord(blk.ev_fork, blk.s));
    SMH_HIP(hipStreamWaitEvent(blk.sx, blk.ev_fork, 0));
    return SMH_OK;
}
int join_one(ParBlock &blk) {
    SMH_TRY(use(blk));
    SMH_HIP(hipEventRecord(blk.ev_join, blk.sx));
    SMH_HIP(hipStreamWaitEvent(blk.s, blk.ev_join, 0));
    return SMH_OK;
}
int fork_side(smh_par *p) {
    for (ParBlock &blk : p->b) SMH_TRY(fork_one(blk));
    return SMH_OK;
}
int join_side(smh_par *p) {
    for (ParBlock &blk : p->b) SMH_TRY(join_one(blk));
    return SMH_OK;
}

// the interior of a block as ITS kernel for `variant` can launch it: [*a, *e) (local rows; *a == *e: the product is not split)
int interior_for(ParBlock &blk, int variant, size_t *a, size_t *e) {
    *a = *e = 0;
    if (blk.in1 <= blk.in0) return SMH_OK;
    size_t gran = 0;
    SMH_TRY(spmv_rows_granularity(blk.m, variant, &gran));
    if (gran == 0) return SMH_OK;
    const size_t n_loc = blk.r1 - blk.r0;
    const size_t lo = (blk.in0 + gran - 1) / gran * gran, hi = blk.in1 == n_loc ? n_loc : blk.in1 / gran * gran;
    if (lo < hi && (lo > 0 || hi < n_loc)) { *a = lo; *e = hi; }
    return SMH_OK;
}

// does a window exchange run beside the products for this call?  (every local block decides the same way: the mode is global)
bool overlapped(const smh_par *p, int resolved_mode) {
    static const bool off = getenv("SMH_PAR_OVERLAP") && atoi(getenv("SMH_PAR_OVERLAP")) == 0;  // tuning knob
    return p->overlap && !off && resolved_mode == SMH_EXCHANGE_WINDOW && !lone_block_skips(p);
}

// a block's interior: the largest run of 256-row tiles that holds no row another block references and no row that references
// another block's columns (square matrices: the window exchange's precondition)
int find_interiors(smh_par *p) {
    const size_t nb = p->n_blocks;
    for (ParBlock &blk : p->b) {
        blk.in0 = blk.in1 = 0;
        const size_t n_loc = blk.r1 - blk.r0;
        if (p->n_rows != p->n_cols || n_loc == 0 || nb <= 1) continue;
        SMH_TRY(use(blk));
        const size_t n_tiles = (n_loc + kBlock - 1) / kBlock;
        std::vector<uint8_t> dirty(n_tiles, 0);
        if (smh_crs_nnz(blk.m)) {
            uint8_t *d_flag = nullptr;
            SMH_HIP(hipMalloc((void **)&d_flag, n_tiles));
            hipLaunchKernelGGL(k_par_remote_tiles, dim3((unsigned)n_tiles), dim3(kBlock), 0, blk.s, blk.m->d_off, blk.m->d_col, (uint64_t)n_loc,
                               (uint32_t)blk.r0, (uint32_t)blk.r1, d_flag);
            hipError_t e = hipGetLastError();
            if (e == hipSuccess) e = hipMemcpyAsync(dirty.data(), d_flag, n_tiles, hipMemcpyDeviceToHost, blk.s);
            if (e == hipSuccess) e = hipStreamSynchronize(blk.s);
            (void)hipFree(d_flag);
            SMH_HIP(e);
        }
        for (size_t q = 0; q < nb; ++q) {  // what block q references of my slice
            size_t a, e;
            recv_range(part_of(p), p->needs.data(), p->lo.data(), p->hi.data(), q, blk.index, &a, &e);
            if (a < e)
                for (size_t t = (a - blk.r0) / kBlock; t <= (e - 1 - blk.r0) / kBlock; ++t) dirty[t] = 1;
        }
        size_t best0 = 0, best1 = 0, run0 = 0;
        for (size_t t = 0; t <= n_tiles; ++t) {
            if (t == n_tiles || dirty[t]) {
                if (t - run0 > best1 - best0) { best0 = run0; best1 = t; }
                run0 = t + 1;
            }
        }
        blk.in0 = best0 * kBlock;
        blk.in1 = best1 * kBlock < n_loc ? best1 * kBlock : n_loc;
        if (blk.in1 <= blk.in0) blk.in0 = blk.in1 = 0;
    }
    return SMH_OK;
}

// ---- cross-block folds of device-resident scalars ---------------------------------------------------------------------
// slot s of block blk: where ITS value goes / where all n_blocks values are read by its fold kernel
void *red_all(const smh_par *p, const ParBlock &blk, int slot) {
    const size_t vs = dtype_size(p->dtype);
    char *base = p->backend == SMH_PAR_BACKEND_RCCL ? (char *)blk.d_redv : (char *)p->h_red;
    return base + (size_t)slot * p->n_blocks * vs;
}
void *red_mine(const smh_par *p, const ParBlock &blk, int slot) { return (char *)red_all(p, blk, slot) + blk.index * dtype_size(p->dtype); }

// PEER backend, the three steps of a fold's meeting (each needs the step before ISSUED by every block it names):
int red_mark(smh_par *p, size_t k, int slot) {  // block k's value of the slot is written once its stream gets here
    ParBlock &blk = p->b[k];
    SMH_TRY(use(blk));
    SMH_HIP(hipEventRecord(blk.ev_red[slot], blk.s));
    return SMH_OK;
}
int red_hub(smh_par *p, int slot) {  // block 0's stream has seen them all
    ParBlock &hub = p->b[0];
    SMH_TRY(use(hub));
    for (size_t k = 1; k < p->b.size(); ++k) SMH_HIP(hipStreamWaitEvent(hub.s, p->b[k].ev_red[slot], 0));
    if (p->b.size() > 1) SMH_HIP(hipEventRecord(hub.ev_all[slot], hub.s));
    return SMH_OK;
}
int red_wait(smh_par *p, size_t k, int slot) {  // ... and block k's waits for that
    if (k == 0) return SMH_OK;
    SMH_TRY(use(p->b[k]));
    SMH_HIP(hipStreamWaitEvent(p->b[k].s, p->b[0].ev_all[slot], 0));
    return SMH_OK;
}

// after every local block wrote red_mine(slot): make all n_blocks values visible to every block's stream
int combine(smh_par *p, int slot) {
    if (lone_block_skips(p)) return SMH_OK;
    if (p->backend == SMH_PAR_BACKEND_RCCL) {
        SMH_TRY(ensure_comms(p));
        SMH_NCCL(ncclGroupStart());
        for (ParBlock &blk : p->b) {
            SMH_TRY(use(blk));
            SMH_NCCL(ncclAllGather(red_mine(p, blk, slot), red_all(p, blk, slot), 1, nccl_type(p->dtype), comm_of(p, blk), blk.s));
        }
        SMH_NCCL(ncclGroupEnd());
        return SMH_OK;
    }
    // every block's value is written -> every block may read all of them.  Through ONE meeting point (block 0's stream waits for
    // the others' records, the others wait for its record after that): 2 (n - 1) waits instead of n (n - 1)
    for (size_t k = 0; k < p->b.size(); ++k) SMH_TRY(red_mark(p, k, slot));
    SMH_TRY(red_hub(p, slot));
    for (size_t k = 0; k < p->b.size(); ++k) SMH_TRY(red_wait(p, k, slot));
    return SMH_OK;
}

int ensure_cg_state(smh_par *p) {
    const size_t vs = dtype_size(p->dtype);
    for (ParBlock &blk : p->b) {
        if (blk.d_sc) continue;
        SMH_TRY(use(blk));
        const size_t n_loc = blk.r1 - blk.r0;
        SMH_HIP(hipMalloc(&blk.d_r, (n_loc ? n_loc : 1) * vs));
        SMH_HIP(hipMalloc(&blk.d_ap, (n_loc ? n_loc : 1) * vs));
        SMH_HIP(hipMalloc(&blk.d_partials, ((size_t)kReducePartials + 8) * vs));
        blk.dotp_cap = (n_loc + 255) / 256 + 8;
        SMH_HIP(hipMalloc(&blk.d_dotp, blk.dotp_cap * vs));
        SMH_HIP(hipMalloc(&blk.d_redv, 2 * p->n_blocks * vs));
        SMH_HIP(hipMemset(blk.d_redv, 0, 2 * p->n_blocks * vs));
        SMH_HIP(hipMalloc(&blk.d_sc, cg_scalars_bytes(p->dtype)));
        SMH_HIP(hipMalloc(&blk.d_sc2, cg_scalars_bytes(p->dtype)));
    }
    if (!p->h_red) {
        SMH_HIP(hipHostMalloc(&p->h_red, 2 * p->n_blocks * sizeof(double), hipHostMallocPortable | hipHostMallocMapped));
        memset(p->h_red, 0, 2 * p->n_blocks * sizeof(double));
    }
    if (!p->h_sc) SMH_HIP(hipHostMalloc(&p->h_sc, cg_scalars_bytes(p->dtype), hipHostMallocDefault));
    return SMH_OK;
}

int vec_create(smh_par *p, size_t n, smh_par_vec **out) {
    smh_par_vec *v = new (std::nothrow) smh_par_vec();
    if (!v) return fail(SMH_ERR_OOM, "host allocation failed");
    v->p = p;
    v->n = n;
    v->d.assign(p->b.size(), nullptr);
    const size_t vs = dtype_size(p->dtype);
    for (size_t k = 0; k < p->b.size(); ++k) {
        int rc = use(p->b[k]);
        if (rc == SMH_OK) {
            hipError_t e = hipMalloc(&v->d[k], (n ? n : 1) * vs);
            if (e == hipSuccess) e = hipMemsetAsync(v->d[k], 0, (n ? n : 1) * vs, p->b[k].s);
            if (e != hipSuccess) rc = hip_fail(e, "hipMalloc(par vec)", __FILE__, __LINE__);
        }
        if (rc != SMH_OK) {
            for (size_t j = 0; j <= k; ++j) { (void)hipSetDevice(p->b[j].device); (void)hipFree(v->d[j]); }
            delete v;
            return rc;
        }
    }
    *out = v;
    return SMH_OK;
}

void vec_destroy(smh_par_vec *v) {
    if (!v) return;
    for (size_t k = 0; k < v->d.size(); ++k) {
        (void)hipSetDevice(v->p->b[k].device);
        (void)hipStreamSynchronize(v->p->b[k].s);
        (void)hipFree(v->d[k]);
    }
    (void)hipGetLastError();
    delete v;
}

int check_vec(const smh_par *p, const smh_par_vec *v, const char *what) {
    if (!v) return fail(SMH_ERR_INVALID, "NULL %s", what);
    if (v->p != p) return fail(SMH_ERR_INVALID, "%s belongs to another partition", what);
    return SMH_OK;
}

// streams, events, peer access, the table of column intervals, the backend: everything after the blocks exist
int finish_par(smh_par *p) {
    const size_t nl = p->b.size();
    for (ParBlock &blk : p->b) {
        SMH_TRY(use(blk));
        SMH_HIP(hipStreamCreateWithFlags(&blk.s, hipStreamNonBlocking));
        SMH_HIP(hipStreamCreateWithFlags(&blk.sx, hipStreamNonBlocking));
        SMH_HIP(hipEventCreateWithFlags(&blk.ev_fork, hipEventDisableTiming));
        SMH_HIP(hipEventCreateWithFlags(&blk.ev_join, hipEventDisableTiming));
        SMH_HIP(hipEventCreateWithFlags(&blk.ev_slice, hipEventDisableTiming));
        SMH_HIP(hipEventCreateWithFlags(&blk.ev_done, hipEventDisableTiming));
        SMH_HIP(hipEventCreateWithFlags(&blk.ev_red[0], hipEventDisableTiming));
        SMH_HIP(hipEventCreateWithFlags(&blk.ev_red[1], hipEventDisableTiming));
        SMH_HIP(hipEventCreateWithFlags(&blk.ev_all[0], hipEventDisableTiming));
        SMH_HIP(hipEventCreateWithFlags(&blk.ev_all[1], hipEventDisableTiming));
    }
    // direct device-to-device reads where the hardware offers them (xGMI)
    p->peer_ok.assign(nl * nl, 0);
    for (size_t a = 0; a < nl; ++a)
        for (size_t c = 0; c < nl; ++c) {
            if (p->b[a].device == p->b[c].device) { p->peer_ok[a * nl + c] = 1; continue; }
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, p->b[a].device, p->b[c].device) == hipSuccess && can) {
                (void)hipSetDevice(p->b[a].device);
                const hipError_t e = hipDeviceEnablePeerAccess(p->b[c].device, 0);
                if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) p->peer_ok[a * nl + c] = 1;
            }
            (void)hipGetLastError();
        }
    bool distinct = true;
    for (size_t a = 0; a < nl; ++a)
        for (size_t c = a + 1; c < nl; ++c)
            if (p->b[a].device == p->b[c].device) distinct = false;
    if (p->rank_comm) {
        p->backend = SMH_PAR_BACKEND_RCCL;
    } else {
        p->backend = distinct && nl > 1 ? SMH_PAR_BACKEND_RCCL : SMH_PAR_BACKEND_PEER;
        if (const char *e = getenv("SMH_PAR_BACKEND")) {
            if (!strcmp(e, "peer")) p->backend = SMH_PAR_BACKEND_PEER;
            else if (!strcmp(e, "rccl") && distinct) p->backend = SMH_PAR_BACKEND_RCCL;
        }
    }
    return SMH_OK;
}

int own_interval(ParBlock &blk, size_t n_cols, uint8_t *needs, uint32_t *lo, uint32_t *hi) {
    *needs = smh_crs_nnz(blk.m) != 0;
    SMH_TRY(smh_crs_col_range(blk.m, lo, hi));
    if (*needs && (size_t)*hi >= n_cols)
        return fail(SMH_ERR_INDEX_RANGE, "block %zu: column %u out of range for %zu columns", blk.index, *hi, n_cols);
    return SMH_OK;
}

}  // namespace

extern "C" {

// ---- one rank of an RCCL communicator (one process per GPU) ----------------------------------------------------------
int smh_comm_unique_id(void *id_out) {
    if (!id_out) return fail(SMH_ERR_INVALID, "NULL id buffer");
    static_assert(sizeof(ncclUniqueId) == SMH_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    SMH_NCCL(ncclGetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return SMH_OK;
}

int smh_comm_create(const void *id, int n_ranks, int rank, smh_comm **out) {
    if (!id || !out) return fail(SMH_ERR_INVALID, "NULL argument");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(SMH_ERR_INVALID, "rank %d of %d", rank, n_ranks);
    SMH_TRY(require_device());
    smh_comm *c = new (std::nothrow) smh_comm();
    if (!c) return fail(SMH_ERR_OOM, "host allocation failed");
    c->n_ranks = n_ranks;
    c->rank = rank;
    c->device = current_device();
    auto go = [&]() -> int {
        ncclUniqueId uid;
        memcpy(&uid, id, sizeof uid);
        SMH_NCCL(ncclCommInitRank(&c->comm, n_ranks, uid, rank));
        SMH_HIP(hipStreamCreateWithFlags(&c->s, hipStreamNonBlocking));
        SMH_HIP(hipMalloc(&c->d_scratch, 64));
        return SMH_OK;
    };
    const int rc = go();
    if (rc != SMH_OK) {
        char keep[512];
        strncpy(keep, smh_last_error(), sizeof keep);
        keep[sizeof keep - 1] = 0;
        smh_comm_destroy(c);
        return fail(rc, "%s", keep);
    }
    *out = c;
    return SMH_OK;
}

int smh_comm_destroy(smh_comm *c) {
    if (!c) return SMH_OK;
    DeviceGuard g;
    (void)hipSetDevice(c->device);
    if (c->s) { (void)hipStreamSynchronize(c->s); (void)hipStreamDestroy(c->s); }
    (void)hipFree(c->d_scratch);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    (void)hipGetLastError();
    delete c;
    return SMH_OK;
}

int smh_comm_size(const smh_comm *c) { return c ? c->n_ranks : 0; }
int smh_comm_rank(const smh_comm *c) { return c ? c->rank : -1; }

int smh_comm_ranks_seen(const smh_comm *c, int *count_out, int *device_out) {
    if (!c || !c->comm) return fail(SMH_ERR_INVALID, "NULL communicator");
    if (count_out) SMH_NCCL(ncclCommCount(c->comm, count_out));
    if (device_out) SMH_NCCL(ncclCommCuDevice(c->comm, device_out));
    return SMH_OK;
}

int smh_rccl_version(int *version_out) {
    if (!version_out) return fail(SMH_ERR_INVALID, "NULL argument");
    SMH_NCCL(ncclGetVersion(version_out));
    return SMH_OK;
}

int smh_comm_max_f64(smh_comm *c, double *value_inout) {
    if (!c || !value_inout) return fail(SMH_ERR_INVALID, "NULL argument");
    DeviceGuard g;
    SMH_HIP(hipSetDevice(c->device));
    SMH_HIP(hipMemcpyAsync(c->d_scratch, value_inout, sizeof(double), hipMemcpyHostToDevice, c->s));
    SMH_NCCL(ncclAllReduce(c->d_scratch, c->d_scratch, 1, ncclDouble, ncclMax, c->comm, c->s));
    SMH_HIP(hipMemcpyAsync(value_inout, c->d_scratch, sizeof(double), hipMemcpyDeviceToHost, c->s));
    SMH_HIP(hipStreamSynchronize(c->s));
    return SMH_OK;
}

int smh_comm_barrier(smh_comm *c) {
    if (!c) return fail(SMH_ERR_INVALID, "NULL communicator");
    DeviceGuard g;
    SMH_HIP(hipSetDevice(c->device));
    SMH_HIP(hipDeviceSynchronize());  // this rank's device work first: the barrier then orders the ranks' host timelines
    double one = 1.0;
    return smh_comm_max_f64(c, &one);
}

// ---- construction ----------------------------------------------------------------------------------------------------
static int par_fail_cleanup(smh_par *p, int rc, int prev_device) {
    char keep[512];
    strncpy(keep, smh_last_error(), sizeof keep);
    keep[sizeof keep - 1] = 0;
    smh_par_destroy(p);
    (void)hipSetDevice(prev_device);
    return fail(rc, "%s", keep);
}

// nnz-balanced boundaries: block k starts at the first row whose entries begin at or beyond k nnz / n_blocks (every block keeps at
// least one row)
static void split_by_nnz(size_t n_blocks, size_t n_rows, const uint32_t *off, std::vector<size_t> &split) {
    split.assign(n_blocks + 1, 0);
    const uint64_t nnz = (uint64_t)off[n_rows] - off[0];
    for (size_t k = 1; k < n_blocks; ++k) {
        const uint64_t want = (uint64_t)off[0] + nnz * k / n_blocks;
        size_t lo = 0, hi = n_rows;  // first row r with off[r] >= want
        while (lo < hi) {
            const size_t mid = (lo + hi) / 2;
            if (off[mid] < want) lo = mid + 1; else hi = mid;
        }
        size_t r = lo;
        if (r < split[k - 1] + 1) r = split[k - 1] + 1;
        if (r > n_rows - (n_blocks - k)) r = n_rows - (n_blocks - k);
        split[k] = r;
    }
    split[n_blocks] = n_rows;
}

static int check_split(size_t n_blocks, size_t n_rows, const size_t *split) {
    if (split[0] != 0 || split[n_blocks] != n_rows) return fail(SMH_ERR_INVALID, "split table: must run from 0 to n_rows (%zu)", n_rows);
    for (size_t k = 0; k < n_blocks; ++k)
        if (split[k + 1] < split[k]) return fail(SMH_ERR_INVALID, "split table: block %zu ends before it begins", k);
    return SMH_OK;
}

int smh_par_create(smh_dtype dtype, size_t n_blocks, const int *device_ids, size_t n_rows, size_t n_cols,
                   const uint32_t *offset_rows, const uint32_t *columns, const void *values, int validate, smh_par **out) {
    return smh_par_create_split(dtype, n_blocks, device_ids, n_rows, n_cols, offset_rows, columns, values, validate, SMH_SPLIT_ROWS, out);
}

int smh_par_create_split(smh_dtype dtype, size_t n_blocks, const int *device_ids, size_t n_rows, size_t n_cols,
                         const uint32_t *offset_rows, const uint32_t *columns, const void *values, int validate, int split_mode, smh_par **out) {
    if (!out) return fail(SMH_ERR_INVALID, "NULL out pointer");
    if (split_mode != SMH_SPLIT_ROWS && split_mode != SMH_SPLIT_NNZ) return fail(SMH_ERR_INVALID, "unknown split mode %d", split_mode);
    if (dtype != SMH_F32 && dtype != SMH_F64) return fail(SMH_ERR_INVALID, "unknown dtype %d", (int)dtype);
    if (n_blocks == 0) return fail(SMH_ERR_INVALID, "SparseMatPar needs at least one block");
    if (!offset_rows) return fail(SMH_ERR_INVALID, "NULL offset_rows");
    const size_t rpb = n_rows / n_blocks;  // sparsemat_par.rs:21
    if (rpb == 0) return fail(SMH_ERR_INVALID, "fewer rows (%zu) than blocks (%zu): rows per block would be 0 (sparsemat_par.rs:21,32)", n_rows, n_blocks);
    int n_dev = 0;
    SMH_TRY(smh_device_count(&n_dev));
    if (n_dev == 0) return fail(SMH_ERR_NO_DEVICE, "no HIP device visible: libsparsemat_hip has no CPU fallback");
    int prev = 0;
    (void)hipGetDevice(&prev);
    smh_par *p = new (std::nothrow) smh_par();
    if (!p) return fail(SMH_ERR_OOM, "host allocation failed");
    p->dtype = dtype; p->n_rows = n_rows; p->n_cols = n_cols; p->rows_per_block = rpb; p->n_blocks = n_blocks;
    p->b.resize(n_blocks);
    p->lo.assign(n_blocks, 0); p->hi.assign(n_blocks, 0); p->needs.assign(n_blocks, 0);
    if (split_mode == SMH_SPLIT_NNZ) split_by_nnz(n_blocks, n_rows, offset_rows, p->split);
    const size_t vs = dtype_size(dtype);
    auto go = [&]() -> int {
        std::vector<uint32_t> off;
        for (size_t k = 0; k < n_blocks; ++k) {
            ParBlock &blk = p->b[k];
            blk.index = k;
            blk.device = device_ids ? device_ids[k] : (int)(k % (size_t)n_dev);
            if (blk.device < 0 || blk.device >= n_dev) return fail(SMH_ERR_INVALID, "block %zu: device %d of %d", k, blk.device, n_dev);
            block_rows(part_of(p), k, &blk.r0, &blk.r1);
            SMH_TRY(use(blk));
            const size_t rows = blk.r1 - blk.r0;
            const uint32_t base = offset_rows[blk.r0];
            if (offset_rows[blk.r1] < base) return fail(SMH_ERR_INVALID, "offset_rows is not monotone");
            const size_t nnz = offset_rows[blk.r1] - base;
            off.resize(rows + 1);
            for (size_t i = 0; i <= rows; ++i) off[i] = offset_rows[blk.r0 + i] - base;  // local offsets, global columns
            SMH_TRY(smh_crs_create(dtype, rows, n_cols, nnz, off.data(), columns ? columns + base : nullptr,
                                   values ? (const char *)values + (size_t)base * vs : nullptr, validate, &blk.m));
            SMH_TRY(own_interval(blk, n_cols, &p->needs[k], &p->lo[k], &p->hi[k]));
        }
        SMH_TRY(finish_par(p));
        return find_interiors(p);
    };
    const int rc = go();
    if (rc != SMH_OK) return par_fail_cleanup(p, rc, prev);
    (void)hipSetDevice(prev);
    *out = p;
    return SMH_OK;
}

int smh_par_adopt(size_t n_blocks, smh_crs *const *blocks, size_t n_rows, smh_par **out) {
    return smh_par_adopt_split(n_blocks, blocks, n_rows, nullptr, out);
}

int smh_par_adopt_split(size_t n_blocks, smh_crs *const *blocks, size_t n_rows, const size_t *split_rows, smh_par **out) {
    if (!out || !blocks) return fail(SMH_ERR_INVALID, "NULL argument");
    if (split_rows && n_blocks) SMH_TRY(check_split(n_blocks, n_rows, split_rows));
    if (n_blocks == 0) return fail(SMH_ERR_INVALID, "SparseMatPar needs at least one block");
    const size_t rpb = n_rows / n_blocks;
    if (rpb == 0) return fail(SMH_ERR_INVALID, "fewer rows (%zu) than blocks (%zu): rows per block would be 0 (sparsemat_par.rs:21,32)", n_rows, n_blocks);
    for (size_t k = 0; k < n_blocks; ++k)
        if (!blocks[k]) return fail(SMH_ERR_INVALID, "block %zu is NULL", k);
    int prev = 0;
    (void)hipGetDevice(&prev);
    smh_par *p = new (std::nothrow) smh_par();
    if (!p) return fail(SMH_ERR_OOM, "host allocation failed");
    p->dtype = blocks[0]->dtype; p->n_rows = n_rows; p->n_cols = blocks[0]->n_cols; p->rows_per_block = rpb; p->n_blocks = n_blocks;
    p->b.resize(n_blocks);
    p->lo.assign(n_blocks, 0); p->hi.assign(n_blocks, 0); p->needs.assign(n_blocks, 0);
    if (split_rows) p->split.assign(split_rows, split_rows + n_blocks + 1);
    auto go = [&]() -> int {
        for (size_t k = 0; k < n_blocks; ++k) {
            ParBlock &blk = p->b[k];
            blk.index = k;
            blk.m = blocks[k];
            blk.owns_m = false;
            blk.device = blk.m->device;
            block_rows(part_of(p), k, &blk.r0, &blk.r1);
            if (blk.m->dtype != p->dtype || blk.m->n_cols != p->n_cols)
                return fail(SMH_ERR_INVALID, "block %zu: dtype / n_cols differ from block 0", k);
            if (blk.m->n_rows != blk.r1 - blk.r0)
                return fail(SMH_ERR_DIM_MISMATCH, "block %zu has %zu rows, the partition of %zu rows into %zu blocks gives it %zu", k,
                            blk.m->n_rows, n_rows, n_blocks, blk.r1 - blk.r0);
            SMH_TRY(use(blk));
            SMH_TRY(own_interval(blk, p->n_cols, &p->needs[k], &p->lo[k], &p->hi[k]));
        }
        SMH_TRY(finish_par(p));
        return find_interiors(p);
    };
    const int rc = go();
    if (rc != SMH_OK) return par_fail_cleanup(p, rc, prev);
    (void)hipSetDevice(prev);
    *out = p;
    return SMH_OK;
}

int smh_par_create_rank(smh_comm *comm, size_t n_rows, smh_crs *block, smh_par **out) {
    return smh_par_create_rank_split(comm, n_rows, block, (size_t)-1, out);
}

// row_begin == (size_t)-1: the reference's partition (this rank's block must hold rows [rank R, (rank + 1) R), the last rank also
// the remainder); else this rank's block holds the rows [row_begin, row_begin + its row count) and the ranks' ranges, all-gathered
// here, must tile [0, n_rows) in rank order (every rank passes a row_begin, or none does)
int smh_par_create_rank_split(smh_comm *comm, size_t n_rows, smh_crs *block, size_t row_begin, smh_par **out) {
    if (!out || !comm || !block) return fail(SMH_ERR_INVALID, "NULL argument");
    const bool own_split = row_begin != (size_t)-1;
    const size_t n_blocks = (size_t)comm->n_ranks, k = (size_t)comm->rank;
    const size_t rpb = n_rows / n_blocks;
    if (rpb == 0) return fail(SMH_ERR_INVALID, "fewer rows (%zu) than blocks (%zu): rows per block would be 0 (sparsemat_par.rs:21,32)", n_rows, n_blocks);
    if (block->device != comm->device) return fail(SMH_ERR_INVALID, "the block lives on device %d, the communicator rank on device %d", block->device, comm->device);
    int prev = 0;
    (void)hipGetDevice(&prev);
    smh_par *p = new (std::nothrow) smh_par();
    if (!p) return fail(SMH_ERR_OOM, "host allocation failed");
    p->dtype = block->dtype; p->n_rows = n_rows; p->n_cols = block->n_cols; p->rows_per_block = rpb; p->n_blocks = n_blocks;
    p->rank_comm = comm;
    p->b.resize(1);
    p->lo.assign(n_blocks, 0); p->hi.assign(n_blocks, 0); p->needs.assign(n_blocks, 0);
    auto go = [&]() -> int {
        ParBlock &blk = p->b[0];
        blk.index = k;
        blk.m = block;
        blk.owns_m = false;
        blk.device = block->device;
        if (own_split) { blk.r0 = row_begin; blk.r1 = row_begin + block->n_rows; }
        else block_rows(part_of(p), k, &blk.r0, &blk.r1);
        // Everything a rank can check ALONE happens before the gather, but a rank that fails it must not return while its peers
        // block inside the collective: it publishes the failure with its table row, every rank reads every row and all of them fail
        // (or succeed) together, with the same verdict.
        auto local = [&]() -> int {
            SMH_TRY(use(blk));
            SMH_TRY(finish_par(p));  // (streams and events first: independent of what the checks say)
            if (block->n_rows != blk.r1 - blk.r0 || blk.r1 > n_rows)
                return fail(SMH_ERR_DIM_MISMATCH, "rank %zu holds %zu rows, the partition of %zu rows into %zu blocks gives it %zu", k, block->n_rows,
                            n_rows, n_blocks, blk.r1 - blk.r0);
            return own_interval(blk, p->n_cols, &p->needs[k], &p->lo[k], &p->hi[k]);
        };
        const int lrc = local();
        const std::string lmsg = lrc == SMH_OK ? std::string() : std::string(smh_last_error());
        hipStream_t gs = blk.s ? blk.s : comm->s;  // (the communicator's own stream when this block's could not be made)
        // the ranks publish their column intervals and row ranges (the exchange plan is global): 9 u32 per rank, all-gathered in place
        constexpr size_t W = 9;  // needs, lo, hi, row_begin (low, high word), has its own split, row count (low, high word), local status
        std::vector<uint32_t> table(W * n_blocks, 0);
        table[W * k] = p->needs[k]; table[W * k + 1] = p->lo[k]; table[W * k + 2] = p->hi[k];
        table[W * k + 3] = (uint32_t)blk.r0; table[W * k + 4] = (uint32_t)((uint64_t)blk.r0 >> 32); table[W * k + 5] = own_split;
        table[W * k + 6] = (uint32_t)block->n_rows; table[W * k + 7] = (uint32_t)((uint64_t)block->n_rows >> 32); table[W * k + 8] = (uint32_t)lrc;
        uint32_t *d_table = nullptr;
        SMH_HIP(hipMalloc((void **)&d_table, W * n_blocks * sizeof(uint32_t)));
        auto gather = [&]() -> int {
            SMH_HIP(hipMemcpyAsync(d_table, table.data(), W * n_blocks * sizeof(uint32_t), hipMemcpyHostToDevice, gs));
            SMH_NCCL(ncclAllGather(d_table + W * k, d_table, W, ncclUint32, comm->comm, gs));
            SMH_HIP(hipMemcpyAsync(table.data(), d_table, W * n_blocks * sizeof(uint32_t), hipMemcpyDeviceToHost, gs));
            SMH_HIP(hipStreamSynchronize(gs));
            return SMH_OK;
        };
        const int grc = gather();
        (void)hipFree(d_table);
        SMH_TRY(grc);
        // from here on every rank holds the same table and runs the same checks in the same order
        if (lrc != SMH_OK) return fail(lrc, "%s", lmsg.c_str());
        for (size_t q = 0; q < n_blocks; ++q)
            if (table[W * q + 8] != 0) return fail((int)table[W * q + 8], "rank %zu failed its local checks of the partition (status %u); no rank keeps a handle", q, table[W * q + 8]);
        for (size_t q = 0; q < n_blocks; ++q) { p->needs[q] = (uint8_t)table[W * q]; p->lo[q] = table[W * q + 1]; p->hi[q] = table[W * q + 2]; }
        size_t first_with = n_blocks, first_without = n_blocks;
        for (size_t q = n_blocks; q-- > 0;) (table[W * q + 5] ? first_with : first_without) = q;
        if (first_with < n_blocks && first_without < n_blocks)
            return fail(SMH_ERR_INVALID, "rank %zu passes a row_begin, rank %zu does not (all or none)", first_with, first_without);
        if (own_split) {
            p->split.assign(n_blocks + 1, n_rows);
            for (size_t q = 0; q < n_blocks; ++q) p->split[q] = (size_t)((uint64_t)table[W * q + 3] | (uint64_t)table[W * q + 4] << 32);
            // the ranges must tile [0, n_rows) in rank order: no gap, no overlap -- the WHOLE table, identically on every rank
            for (size_t q = 0; q < n_blocks; ++q) {
                const size_t rows_q = (size_t)((uint64_t)table[W * q + 6] | (uint64_t)table[W * q + 7] << 32);
                if (p->split[q] + rows_q != p->split[q + 1])
                    return fail(SMH_ERR_INVALID, "rank %zu: its rows [%zu, %zu) end at %zu, %s begins at %zu (the blocks must tile the rows in rank order)", q,
                                p->split[q], p->split[q] + rows_q, p->split[q] + rows_q, q + 1 < n_blocks ? "the next rank" : "the end of the matrix", p->split[q + 1]);
            }
            SMH_TRY(check_split(n_blocks, n_rows, p->split.data()));
        }
        return find_interiors(p);
    };
    const int rc = go();
    if (rc != SMH_OK) return par_fail_cleanup(p, rc, prev);
    (void)hipSetDevice(prev);
    *out = p;
    return SMH_OK;
}

int smh_par_destroy(smh_par *p) {
    if (!p) return SMH_OK;
    pool_destroy(p);
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (ParBlock &blk : p->b) {
        (void)hipSetDevice(blk.device);
        if (blk.s) (void)hipStreamSynchronize(blk.s);
    }
    vec_destroy(p->cg_p); vec_destroy(p->io_b); vec_destroy(p->io_x);
    for (ParBlock &blk : p->b) {
        (void)hipSetDevice(blk.device);
        if (blk.comm) (void)ncclCommDestroy(blk.comm);
        if (blk.sx) { (void)hipStreamSynchronize(blk.sx); (void)hipStreamDestroy(blk.sx); }
        if (blk.s) (void)hipStreamDestroy(blk.s);
        if (blk.ev_fork) (void)hipEventDestroy(blk.ev_fork);
        if (blk.ev_join) (void)hipEventDestroy(blk.ev_join);
        if (blk.ev_slice) (void)hipEventDestroy(blk.ev_slice);
        if (blk.ev_done) (void)hipEventDestroy(blk.ev_done);
        if (blk.ev_red[0]) (void)hipEventDestroy(blk.ev_red[0]);
        if (blk.ev_red[1]) (void)hipEventDestroy(blk.ev_red[1]);
        if (blk.ev_all[0]) (void)hipEventDestroy(blk.ev_all[0]);
        if (blk.ev_all[1]) (void)hipEventDestroy(blk.ev_all[1]);
        if (blk.owns_m) (void)smh_crs_destroy(blk.m);
        (void)hipFree(blk.d_x); (void)hipFree(blk.d_y); (void)hipFree(blk.d_r); (void)hipFree(blk.d_ap);
        (void)hipFree(blk.d_partials); (void)hipFree(blk.d_dotp); (void)hipFree(blk.d_sc); (void)hipFree(blk.d_sc2); (void)hipFree(blk.d_redv);
    }
    if (p->h_red) (void)hipHostFree(p->h_red);
    if (p->h_sc) (void)hipHostFree(p->h_sc);
    (void)hipGetLastError();
    (void)hipSetDevice(prev);
    delete p;
    return SMH_OK;
}

size_t smh_par_n_blocks(const smh_par *p) { return p ? p->n_blocks : 0; }
size_t smh_par_n_local_blocks(const smh_par *p) { return p ? p->b.size() : 0; }
size_t smh_par_n_rows(const smh_par *p) { return p ? p->n_rows : 0; }
size_t smh_par_n_cols(const smh_par *p) { return p ? p->n_cols : 0; }
size_t smh_par_rows_per_block(const smh_par *p) { return p ? p->rows_per_block : 0; }

size_t smh_par_nnz(const smh_par *p) {  // sparsemat_par.rs:117-123
    size_t n = 0;
    if (p) for (const ParBlock &blk : p->b) n += smh_crs_nnz(blk.m);
    return n;
}

int smh_par_block(const smh_par *p, size_t block, smh_crs **crs_out, size_t *row_begin, size_t *row_end, int *device) {
    if (!p || block >= p->b.size()) return fail(SMH_ERR_INVALID, "no such block");
    const ParBlock &blk = p->b[block];
    if (crs_out) *crs_out = blk.m;
    if (row_begin) *row_begin = blk.r0;
    if (row_end) *row_end = blk.r1;
    if (device) *device = blk.device;
    return SMH_OK;
}

int smh_par_block_stream(const smh_par *p, size_t block, void **stream_out) {
    if (!p || !stream_out || block >= p->b.size()) return fail(SMH_ERR_INVALID, "no such block");
    *stream_out = p->b[block].s;
    return SMH_OK;
}

int smh_par_get_block_and_row_id(const smh_par *p, size_t row, size_t *block_out, size_t *row_out) {
    if (!p || !block_out || !row_out) return fail(SMH_ERR_INVALID, "NULL argument");
    if (!p->split.empty()) {  // a split table: the last block whose first row is <= row
        size_t lo = 0, hi = p->n_blocks;
        while (hi - lo > 1) {
            const size_t mid = (lo + hi) / 2;
            if (p->split[mid] <= row) lo = mid; else hi = mid;
        }
        *block_out = lo;
        *row_out = row - p->split[lo];
        return SMH_OK;
    }
    size_t k = row / p->rows_per_block;  // sparsemat_par.rs:32, clamped to the last block instead of one past it
    if (k > p->n_blocks - 1) k = p->n_blocks - 1;
    *block_out = k;
    *row_out = row - k * p->rows_per_block;
    return SMH_OK;
}

int smh_par_split(const smh_par *p, size_t *rows_out) {
    if (!p || !rows_out) return fail(SMH_ERR_INVALID, "NULL argument");
    const Part pt = part_of(p);
    for (size_t k = 0; k < p->n_blocks; ++k) {
        size_t r0, r1;
        block_rows(pt, k, &r0, &r1);
        rows_out[k] = r0;
        rows_out[k + 1] = r1;
    }
    return SMH_OK;
}

int smh_par_scale(smh_par *p, double a) {  // sparsemat_par.rs:135-139
    if (!p) return fail(SMH_ERR_INVALID, "NULL handle");
    DeviceGuard g;
    for (ParBlock &blk : p->b) {
        SMH_TRY(use(blk));
        SMH_HIP(hipStreamSynchronize(blk.s));
        SMH_TRY(smh_crs_scale(blk.m, a));
    }
    return SMH_OK;
}

int smh_par_set_backend(smh_par *p, int backend) {
    if (!p) return fail(SMH_ERR_INVALID, "NULL handle");
    if (p->rank_comm) {
        if (backend == SMH_PAR_BACKEND_PEER) return fail(SMH_ERR_INVALID, "one process per GPU: the exchange is RCCL");
        return SMH_OK;
    }
    bool distinct = true;
    for (size_t a = 0; a < p->b.size(); ++a)
        for (size_t c = a + 1; c < p->b.size(); ++c)
            if (p->b[a].device == p->b[c].device) distinct = false;
    if (backend == SMH_PAR_BACKEND_AUTO) backend = distinct && p->b.size() > 1 ? SMH_PAR_BACKEND_RCCL : SMH_PAR_BACKEND_PEER;
    if (backend != SMH_PAR_BACKEND_PEER && backend != SMH_PAR_BACKEND_RCCL) return fail(SMH_ERR_INVALID, "unknown backend %d", backend);
    if (backend == SMH_PAR_BACKEND_RCCL && !distinct && p->b.size() > 1)
        return fail(SMH_ERR_INVALID, "RCCL backend: several blocks share a device (one device per block needed)");
    DeviceGuard g;
    SMH_TRY(sync_all(p));
    p->backend = backend;
    return SMH_OK;
}

int smh_par_backend(const smh_par *p) { return p ? p->backend : SMH_PAR_BACKEND_AUTO; }

int smh_par_exchange_mode(const smh_par *p, int mode, int *resolved_out, size_t *max_recv_out) {
    if (!p) return fail(SMH_ERR_INVALID, "NULL handle");
    int m = SMH_EXCHANGE_NONE;
    SMH_TRY(resolve_mode(p, mode, &m));
    if (resolved_out) *resolved_out = m;
    if (max_recv_out) plan_summary(part_of(p), p->needs.data(), p->lo.data(), p->hi.data(), nullptr, max_recv_out);
    return SMH_OK;
}

int smh_par_plan(size_t n_blocks, size_t n_rows, const uint8_t *needs, const uint32_t *lo, const uint32_t *hi, size_t block,
                 size_t *recv_begin, size_t *recv_end, size_t *send_begin, size_t *send_end, int *auto_mode_out, size_t *max_recv_out) {
    return smh_par_plan_split(n_blocks, n_rows, nullptr, needs, lo, hi, block, recv_begin, recv_end, send_begin, send_end, auto_mode_out, max_recv_out);
}

int smh_par_plan_split(size_t n_blocks, size_t n_rows, const size_t *split_rows, const uint8_t *needs, const uint32_t *lo, const uint32_t *hi,
                       size_t block, size_t *recv_begin, size_t *recv_end, size_t *send_begin, size_t *send_end, int *auto_mode_out,
                       size_t *max_recv_out) {
    if (n_blocks == 0 || (!split_rows && n_rows / n_blocks == 0)) return fail(SMH_ERR_INVALID, "rows per block would be 0 (sparsemat_par.rs:21,32)");
    if (!needs || !lo || !hi || block >= n_blocks) return fail(SMH_ERR_INVALID, "bad plan arguments");
    if (split_rows) SMH_TRY(check_split(n_blocks, n_rows, split_rows));
    const Part pt{n_blocks, n_rows, split_rows};
    for (size_t q = 0; q < n_blocks; ++q) {
        size_t a, e;
        recv_range(pt, needs, lo, hi, block, q, &a, &e);
        if (recv_begin) recv_begin[q] = a;
        if (recv_end) recv_end[q] = e;
        recv_range(pt, needs, lo, hi, q, block, &a, &e);
        if (send_begin) send_begin[q] = a;
        if (send_end) send_end[q] = e;
    }
    plan_summary(pt, needs, lo, hi, auto_mode_out, max_recv_out);
    return SMH_OK;
}

// ---- distributed vectors -----------------------------------------------------------------------------------------------
int smh_par_vec_create(smh_par *p, size_t n, smh_par_vec **out) {
    if (!p || !out) return fail(SMH_ERR_INVALID, "NULL argument");
    DeviceGuard g;
    return vec_create(p, n, out);
}

int smh_par_vec_destroy(smh_par_vec *v) {
    DeviceGuard g;
    vec_destroy(v);
    return SMH_OK;
}

size_t smh_par_vec_dim(const smh_par_vec *v) { return v ? v->n : 0; }

int smh_par_vec_upload(smh_par_vec *v, const void *host) {
    if (!v || (v->n && !host)) return fail(SMH_ERR_INVALID, "NULL argument");
    DeviceGuard g;
    smh_par *p = v->p;
    const size_t vs = dtype_size(p->dtype);
    for (size_t k = 0; k < p->b.size(); ++k) {
        SMH_TRY(use(p->b[k]));
        if (v->n) SMH_HIP(hipMemcpyAsync(v->d[k], host, v->n * vs, hipMemcpyHostToDevice, p->b[k].s));
    }
    return sync_all(p);  // the host buffer is borrowed for the call only
}

int smh_par_vec_download(const smh_par_vec *v, void *host) {
    if (!v || !host) return fail(SMH_ERR_INVALID, "NULL argument");
    smh_par *p = v->p;
    if (v->n != p->n_rows) return fail(SMH_ERR_DIM_MISMATCH, "the vector has %zu entries, the partition owns %zu rows", v->n, p->n_rows);
    DeviceGuard g;
    const size_t vs = dtype_size(p->dtype);
    for (size_t k = 0; k < p->b.size(); ++k) {
        const ParBlock &blk = p->b[k];
        SMH_TRY(use(blk));
        if (blk.r1 > blk.r0)
            SMH_HIP(hipMemcpyAsync((char *)host + blk.r0 * vs, (const char *)v->d[k] + blk.r0 * vs, (blk.r1 - blk.r0) * vs, hipMemcpyDeviceToHost, blk.s));
    }
    return sync_all(p);
}

int smh_par_vec_download_block(const smh_par_vec *v, size_t local_block, void *host) {
    if (!v || !host || local_block >= v->d.size()) return fail(SMH_ERR_INVALID, "bad argument");
    DeviceGuard g;
    const ParBlock &blk = v->p->b[local_block];
    SMH_TRY(use(blk));
    if (v->n) SMH_HIP(hipMemcpyAsync(host, v->d[local_block], v->n * dtype_size(v->p->dtype), hipMemcpyDeviceToHost, blk.s));
    SMH_HIP(hipStreamSynchronize(blk.s));
    return SMH_OK;
}

int smh_par_vec_ptr(const smh_par_vec *v, size_t local_block, void **dev_ptr_out) {
    if (!v || !dev_ptr_out || local_block >= v->d.size()) return fail(SMH_ERR_INVALID, "bad argument");
    *dev_ptr_out = v->d[local_block];
    return SMH_OK;
}

// ---- y = A x, device resident ---------------------------------------------------------------------------------------------
int smh_par_spmv_dev(smh_par *p, const smh_par_vec *x, smh_par_vec *y, int variant, int mode) {
    if (!p) return fail(SMH_ERR_INVALID, "NULL handle");
    SMH_TRY(check_vec(p, x, "x"));
    SMH_TRY(check_vec(p, y, "y"));
    if (x == y) return fail(SMH_ERR_INVALID, "x and y must be different vectors");
    if (y->n != p->n_rows) return fail(SMH_ERR_DIM_MISMATCH, "Dimension mismatch");
    DeviceGuard g;
    const size_t vs = dtype_size(p->dtype);
    int m = SMH_EXCHANGE_NONE;
    SMH_TRY(resolve_mode(p, mode, &m));
    if (!overlapped(p, m)) {
        SMH_TRY(par_for(p, [&](size_t k) -> int {
            ParBlock &blk = p->b[k];
            SMH_TRY(use(blk));
            return smh_crs_spmv_dev(blk.m, x->d[k], x->n, (char *)y->d[k] + blk.r0 * vs, variant, blk.s);  // results at b R (:64)
        }));
        return exchange(p, y, mode);
    }
    // the rows other blocks reference and their exchange on the side streams, the interior rows on the main ones -- BOTH from the start:
    // the boundary rows are one or two launches of a few workgroups (13 us each on an otherwise idle chip: 27 us of a 355 us step when
    // they ran ahead of everything, profiles/r04_par_boundary_on_side_stream.log); every block's calls from its own issuing thread, the
    // phases meeting where a wait names another block's event (see ParPool)
    const size_t nl = p->b.size();
    std::vector<size_t> ia(nl, 0), ie(nl, 0);
    std::vector<uint8_t> pulled(nl * nl, 0);
    const bool peer = p->backend != SMH_PAR_BACKEND_RCCL;
    auto interior = [&](size_t k) -> int {
        ParBlock &blk = p->b[k];
        if (ie[k] <= ia[k]) return SMH_OK;
        SMH_TRY(use(blk));
        return spmv_enqueue_rows(blk.m, x->d[k], x->n, (char *)y->d[k] + blk.r0 * vs, variant, blk.s, ia[k], ie[k]);
    };
    SMH_TRY(par_for(p, [&](size_t k) -> int {
        ParBlock &blk = p->b[k];
        SMH_TRY(use(blk));
        SMH_TRY(interior_for(blk, variant, &ia[k], &ie[k]));
        char *yk = (char *)y->d[k] + blk.r0 * vs;
        if (ie[k] > ia[k]) {
            SMH_TRY(fork_one(blk));  // (the side stream after everything the main one holds: x is written, y's readers are through)
            SMH_TRY(spmv_enqueue_rows_short(blk.m, x->d[k], x->n, yk, variant, blk.sx, 0, ia[k]));
            SMH_TRY(spmv_enqueue_rows_short(blk.m, x->d[k], x->n, yk, variant, blk.sx, ie[k], blk.r1 - blk.r0));
        } else {
            SMH_TRY(smh_crs_spmv_dev(blk.m, x->d[k], x->n, yk, variant, blk.s));
            SMH_TRY(fork_one(blk));
        }
        return peer ? peer_mark(p, k, true) : SMH_OK;
    }));
    if (!peer) {
        SMH_TRY(exchange(p, y, mode, true));  // (RCCL: one group for all local blocks, issued here)
        SMH_TRY(par_for(p, interior));
        return join_side(p);
    }
    if (y->n != p->n_rows) return fail(SMH_ERR_DIM_MISMATCH, "exchange: the vector has %zu entries, the partition owns %zu rows", y->n, p->n_rows);
    SMH_TRY(par_for(p, [&](size_t k) -> int {
        SMH_TRY(peer_pull(p, y, m, k, true, pulled.data()));
        return interior(k);
    }));
    return par_for(p, [&](size_t k) -> int {
        SMH_TRY(peer_guard(p, k, true, pulled.data()));
        return join_one(p->b[k]);
    });
}

int smh_par_set_threads(smh_par *p, int mode) {
    if (!p) return fail(SMH_ERR_INVALID, "NULL handle");
    if (mode < -1 || mode > 1) return fail(SMH_ERR_INVALID, "mode must be -1 (automatic), 0 (one issuing thread) or 1 (a thread per local block)");
    DeviceGuard g;
    SMH_TRY(sync_all(p));
    p->use_threads = mode;
    if (!threads_wanted(p)) pool_destroy(p);
    return SMH_OK;
}

int smh_par_set_overlap(smh_par *p, int on) {
    if (!p) return fail(SMH_ERR_INVALID, "NULL handle");
    DeviceGuard g;
    SMH_TRY(sync_all(p));
    p->overlap = on != 0;
    return SMH_OK;
}

int smh_par_interior(const smh_par *p, size_t local_block, int variant, size_t *row_begin, size_t *row_end) {
    if (!p || local_block >= p->b.size() || !row_begin || !row_end) return fail(SMH_ERR_INVALID, "bad argument");
    DeviceGuard g;
    ParBlock &blk = const_cast<smh_par *>(p)->b[local_block];
    SMH_TRY(use(blk));
    return interior_for(blk, variant, row_begin, row_end);
}

int smh_par_exchange(smh_par *p, smh_par_vec *v, int mode) {
    if (!p) return fail(SMH_ERR_INVALID, "NULL handle");
    SMH_TRY(check_vec(p, v, "vector"));
    DeviceGuard g;
    return exchange(p, v, mode);
}

int smh_par_synchronize(smh_par *p) {
    if (!p) return fail(SMH_ERR_INVALID, "NULL handle");
    DeviceGuard g;
    return sync_all(p);
}

// y[0..n_rows) = A x on host vectors: every block gets the part of x its columns reference, all blocks run concurrently
int smh_par_spmv(smh_par *p, const void *x_host, size_t x_len, void *y_host, int variant) {
    if (!p) return fail(SMH_ERR_INVALID, "NULL handle");
    if (p->rank_comm) return fail(SMH_ERR_INVALID, "smh_par_spmv takes whole host vectors: one-process handles only (use smh_par_spmv_dev)");
    if (!y_host || (x_len && !x_host)) return fail(SMH_ERR_INVALID, "NULL host vector");
    const size_t vs = dtype_size(p->dtype);
    DeviceGuard g;
    auto go = [&]() -> int {
        for (ParBlock &blk : p->b) {
            SMH_TRY(use(blk));
            if (!blk.d_x) {
                SMH_HIP(hipMalloc(&blk.d_x, (p->n_cols ? p->n_cols : 1) * vs));
                SMH_HIP(hipMalloc(&blk.d_y, (blk.r1 > blk.r0 ? blk.r1 - blk.r0 : 1) * vs));
            }
            if (p->needs[blk.index]) {
                const uint32_t lo = p->lo[blk.index], hi = p->hi[blk.index];
                if ((size_t)hi >= x_len)  // rhs.get(j): densevec.rs:41
                    return fail(SMH_ERR_INDEX_RANGE, "index out of bounds: the len is %zu but the index is %u", x_len, hi);
                SMH_HIP(hipMemcpyAsync((char *)blk.d_x + (size_t)lo * vs, (const char *)x_host + (size_t)lo * vs, ((size_t)hi - lo + 1) * vs,
                                       hipMemcpyHostToDevice, blk.s));
            }
            SMH_TRY(smh_crs_spmv_dev(blk.m, blk.d_x, x_len < p->n_cols ? x_len : p->n_cols, blk.d_y, variant, blk.s));
            if (blk.r1 > blk.r0)
                SMH_HIP(hipMemcpyAsync((char *)y_host + blk.r0 * vs, blk.d_y, (blk.r1 - blk.r0) * vs, hipMemcpyDeviceToHost, blk.s));
        }
        return sync_all(p);
    };
    const int rc = go();
    if (rc != SMH_OK) (void)sync_all(p);
    return rc;
}

// ---- ConjugateGradient::solve (linearsolver.rs:27-61) on the partitioned matrix --------------------------------------------
int smh_par_cg_solve_vec(smh_par *p, const smh_par_vec *b, smh_par_vec *x, double tol, size_t iter_max, int variant,
                         size_t check_every, size_t *iters_out, double *rr_out) {
    if (!p) return fail(SMH_ERR_INVALID, "NULL handle");
    SMH_TRY(check_vec(p, b, "b"));
    SMH_TRY(check_vec(p, x, "x"));
    if (b == x) return fail(SMH_ERR_INVALID, "b and x must be different vectors");
    if (p->n_rows != p->n_cols) return fail(SMH_ERR_NOT_SQUARE, "Matrix is not symmetric");                        // :30-32
    if (p->n_rows != b->n || p->n_rows != x->n) return fail(SMH_ERR_DIM_MISMATCH, "Matrix and vector size mismatch");  // :33-36
    if (check_every == 0) check_every = 8;
    const size_t vs = dtype_size(p->dtype), n = p->n_rows;
    const int dt = p->dtype;
    const uint32_t nb = (uint32_t)p->n_blocks;
    DeviceGuard g;
    size_t iters = 0;
    double rr = 0.0;
    auto go = [&]() -> int {
        SMH_TRY(ensure_cg_state(p));
        if (!p->cg_p) SMH_TRY(vec_create(p, n, &p->cg_p));
        smh_par_vec *pv = p->cg_p;
        int mode = SMH_EXCHANGE_NONE;
        SMH_TRY(resolve_mode(p, SMH_EXCHANGE_AUTO, &mode));
        // r = b - A x (:38); p = r.clone() (:39); rr = r.r (:40)
        SMH_TRY(exchange(p, x, mode));  // x on every block's column interval
        for (size_t k = 0; k < p->b.size(); ++k) {
            ParBlock &blk = p->b[k];
            SMH_TRY(use(blk));
            const size_t n_loc = blk.r1 - blk.r0;
            SMH_TRY(smh_crs_spmv_dev(blk.m, x->d[k], n, blk.d_r, variant, blk.s));
            SMH_TRY(launch_ew(dt, Ew::RSubInto, blk.d_r, (const char *)b->d[k] + blk.r0 * vs, n_loc, 0.0, nullptr, blk.s));
            SMH_HIP(hipMemcpyAsync((char *)pv->d[k] + blk.r0 * vs, blk.d_r, n_loc * vs, hipMemcpyDeviceToDevice, blk.s));
            SMH_TRY(cg_par_init(dt, blk.d_sc, tol, iter_max, blk.s));
            SMH_TRY(launch_dot(dt, blk.d_r, blk.d_r, n_loc, blk.d_partials, red_mine(p, blk, 1), blk.s));
        }
        SMH_TRY(combine(p, 1));
        for (ParBlock &blk : p->b) {
            SMH_TRY(use(blk));
            SMH_TRY(cg_par_set_rr(dt, blk.d_sc, red_all(p, blk, 1), nb, blk.s));
        }
        // ---- one iteration (:41-60) as phases: within a phase every block's calls are issued by its own host thread (par_for);
        // between phases all of them have been issued -- what a wait on another block's event needs.  Streams, events, kernels and
        // their order per block are those of the one-thread form, whatever the number of threads.
        const bool peer = p->backend != SMH_PAR_BACKEND_RCCL;
        const bool lone = lone_block_skips(p);
        const size_t nl = p->b.size();
        std::vector<uint8_t> pulled(nl * nl, 0);
        // the product of block k's rows [part 0: its interior, on the main stream | part 1: the rest -- on the SIDE stream, behind the
        // exchange they wait for and beside the interior rows -- or, unsplit, all of them | part 2, after the join: the block's p.Ap]
        auto products = [&](size_t k, int part, bool side) -> int {
            ParBlock &blk = p->b[k];
            SMH_TRY(use(blk));
            const size_t n_loc = blk.r1 - blk.r0;
            size_t ia = 0, ie = 0;
            if (side) SMH_TRY(interior_for(blk, variant, &ia, &ie));
            const bool split = ie > ia;
            if (part == 0 && !split) return SMH_OK;
            // :43 and :45 -- with the CSR-stream kernel p.Ap rides the product's epilogue (one partial per tile, lhs = this
            // block's slice of p) as in the single-matrix solver: no second pass over p and Ap
            const size_t n_dot = spmv_fused_dot_partials(blk.m, n, variant, true);
            const bool fused = n_dot && n_dot <= blk.dotp_cap;
            void *dotp = fused ? blk.d_dotp : nullptr;
            const void *lhs = fused ? (const char *)pv->d[k] + blk.r0 * vs : nullptr;
            if (part == 0) return spmv_enqueue_rows(blk.m, pv->d[k], n, blk.d_ap, variant, blk.s, ia, ie, dotp, lhs);
            if (part == 1 && split) {
                if (!dotp) {
                    SMH_TRY(spmv_enqueue_rows_short(blk.m, pv->d[k], n, blk.d_ap, variant, blk.sx, 0, ia));
                    return spmv_enqueue_rows_short(blk.m, pv->d[k], n, blk.d_ap, variant, blk.sx, ie, n_loc);
                }
                SMH_TRY(spmv_enqueue_rows(blk.m, pv->d[k], n, blk.d_ap, variant, blk.sx, 0, ia, dotp, lhs));
                return spmv_enqueue_rows(blk.m, pv->d[k], n, blk.d_ap, variant, blk.sx, ie, n_loc, dotp, lhs);
            }
            if (part == 1) return SMH_OK;  // (unsplit: everything in part 2, after the exchange has been joined)
            if (!split) {
                if (fused) SMH_TRY(spmv_enqueue(blk.m, pv->d[k], n, blk.d_ap, variant, blk.s, blk.d_dotp, lhs));
                else SMH_TRY(smh_crs_spmv_dev(blk.m, pv->d[k], n, blk.d_ap, variant, blk.s));
            }
            if (fused) return launch_fold2(dt, blk.d_dotp, n_dot, blk.d_partials, red_mine(p, blk, 0), blk.s);
            return launch_dot(dt, (const char *)pv->d[k] + blk.r0 * vs, blk.d_ap, n_loc, blk.d_partials, red_mine(p, blk, 0), blk.s);
        };
        auto update_xr = [&](size_t k) -> int {  // alpha, then x / r and this block's share of r.r
            ParBlock &blk = p->b[k];
            SMH_TRY(use(blk));
            uint32_t cnt = 0;
            SMH_TRY(cg_par_update(dt, blk.d_sc, blk.d_sc2, red_all(p, blk, 0), nb, blk.d_r, blk.d_ap, blk.r1 - blk.r0, blk.d_partials, &cnt, blk.s));  // :45, :49-51
            return cg_fold(dt, blk.d_partials, cnt, red_mine(p, blk, 1), blk.s);
        };
        auto update_p = [&](size_t k) -> int {  // beta (and the stop test before it), then p
            ParBlock &blk = p->b[k];
            SMH_TRY(use(blk));
            return cg_par_p(dt, blk.d_sc2, blk.d_sc, red_all(p, blk, 1), nb, (char *)pv->d[k] + blk.r0 * vs, blk.d_r, (char *)x->d[k] + blk.r0 * vs,
                            blk.r1 - blk.r0, blk.s);  // :52-56, :47, :58-59
        };
        auto iteration = [&]() -> int {
            // the entries of p a block references and another owns -- beside the product of the interior rows, which need none
            // of them, when the exchange is a window (the rows that do wait for it at the join)
            const bool side = overlapped(p, mode);
            const bool xchg = mode != SMH_EXCHANGE_NONE;
            if (peer) {
                if (xchg) SMH_TRY(par_for(p, [&](size_t k) -> int {
                    if (side) SMH_TRY(fork_one(p->b[k]));
                    return peer_mark(p, k, side);
                }));
                SMH_TRY(par_for(p, [&](size_t k) -> int {
                    if (xchg) SMH_TRY(peer_pull(p, pv, mode, k, side, pulled.data()));
                    return side ? products(k, 0, side) : SMH_OK;
                }));
                SMH_TRY(par_for(p, [&](size_t k) -> int {
                    if (side) SMH_TRY(products(k, 1, side));  // (behind this block's pulls on its side stream)
                    if (xchg) SMH_TRY(peer_guard(p, k, side, pulled.data()));
                    if (side) SMH_TRY(join_one(p->b[k]));
                    SMH_TRY(products(k, 2, side));
                    return lone ? SMH_OK : red_mark(p, k, 0);
                }));
                if (!lone) SMH_TRY(red_hub(p, 0));
                SMH_TRY(par_for(p, [&](size_t k) -> int {
                    if (!lone) SMH_TRY(red_wait(p, k, 0));
                    SMH_TRY(update_xr(k));
                    return lone ? SMH_OK : red_mark(p, k, 1);
                }));
                if (!lone) SMH_TRY(red_hub(p, 1));
                return par_for(p, [&](size_t k) -> int {
                    if (!lone) SMH_TRY(red_wait(p, k, 1));
                    return update_p(k);
                });
            }
            // RCCL: the collectives of all local blocks are ONE group issued by the caller (a process per GPU has one block anyway)
            if (side) SMH_TRY(fork_side(p));
            SMH_TRY(exchange(p, pv, mode, side));
            if (side) {
                SMH_TRY(par_for(p, [&](size_t k) -> int {
                    SMH_TRY(products(k, 1, side));
                    return products(k, 0, side);
                }));
                SMH_TRY(join_side(p));
            }
            SMH_TRY(par_for(p, [&](size_t k) { return products(k, 2, side); }));
            SMH_TRY(combine(p, 0));
            SMH_TRY(par_for(p, update_xr));
            SMH_TRY(combine(p, 1));
            return par_for(p, update_p);
        };
        size_t launched = 0;
        int converged = 0;
        auto poll = [&]() -> int {
            ParBlock &b0 = p->b[0];
            SMH_TRY(use(b0));
            SMH_HIP(hipMemcpyAsync(p->h_sc, b0.d_sc, cg_scalars_bytes(dt), hipMemcpyDeviceToHost, b0.s));
            // (block 0's stream alone: its scalars are every block's scalars -- all fold the same values -- and the other blocks'
            // streams keep their queues full across the poll; everything is drained once, when the solve returns)
            SMH_HIP(hipStreamSynchronize(b0.s));
            uint64_t it64 = 0;
            cg_read_scalars(dt, p->h_sc, &converged, &it64, &rr);
            iters = (size_t)it64;
            return SMH_OK;
        };
        while (launched < iter_max) {
            const size_t batch = iter_max - launched < check_every ? iter_max - launched : check_every;
            static const bool trace = getenv("SMH_PAR_TRACE") && atoi(getenv("SMH_PAR_TRACE")) != 0;  // development aid
            const auto t_a = std::chrono::steady_clock::now();
            for (size_t i = 0; i < batch; ++i) SMH_TRY(iteration());
            const auto t_b = std::chrono::steady_clock::now();
            launched += batch;
            SMH_TRY(poll());
            if (trace) {
                const auto t_c = std::chrono::steady_clock::now();
                fprintf(stderr, "[par cg] %zu iterations: issued in %.3f ms (%.3f per iteration), drained %.3f ms later\n", batch,
                        std::chrono::duration<double, std::milli>(t_b - t_a).count(), std::chrono::duration<double, std::milli>(t_b - t_a).count() / (double)batch,
                        std::chrono::duration<double, std::milli>(t_c - t_b).count());
            }
            if (converged) break;
        }
        if (iter_max == 0) SMH_TRY(poll());
        return sync_all(p);
    };
    const int rc = go();
    if (rc != SMH_OK) {
        char keep[512];
        strncpy(keep, smh_last_error(), sizeof keep);
        keep[sizeof keep - 1] = 0;
        (void)sync_all(p);
        return fail(rc, "%s", keep);
    }
    if (iters_out) *iters_out = iters;
    if (rr_out) *rr_out = rr;
    return SMH_OK;
}

int smh_par_cg_solve(smh_par *p, const void *b_host, size_t b_len, void *x_host_inout, size_t x_len, double tol, size_t iter_max,
                     int variant, size_t *iters_out, double *rr_out) {
    if (!p) return fail(SMH_ERR_INVALID, "NULL handle");
    if (p->n_rows != p->n_cols) return fail(SMH_ERR_NOT_SQUARE, "Matrix is not symmetric");                    // :30-32
    if (p->n_rows != b_len || p->n_rows != x_len) return fail(SMH_ERR_DIM_MISMATCH, "Matrix and vector size mismatch");  // :33-36
    if (!b_host || !x_host_inout) return fail(SMH_ERR_INVALID, "NULL host vector");
    DeviceGuard g;
    if (!p->io_b) SMH_TRY(vec_create(p, p->n_rows, &p->io_b));
    if (!p->io_x) SMH_TRY(vec_create(p, p->n_rows, &p->io_x));
    SMH_TRY(smh_par_vec_upload(p->io_b, b_host));
    SMH_TRY(smh_par_vec_upload(p->io_x, x_host_inout));
    SMH_TRY(smh_par_cg_solve_vec(p, p->io_b, p->io_x, tol, iter_max, variant, 0, iters_out, rr_out));
    return smh_par_vec_download(p->io_x, x_host_inout);
}

}  // extern "C"
