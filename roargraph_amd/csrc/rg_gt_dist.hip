// rg_gt_dist.hip -- multi-rank ground truth behind the C ABI (BASELINE configs[2]).
//
// Replaces: the external `compute_groundtruth` run over the whole base (README.md:62-75) whose ids the build consumes
// (LoadLearnBaseKNN, src/index_bipartite.cpp:2622-2642).
//
// One rank per GPU.  A rank owns a contiguous ROW SHARD of the base, resident in its HBM.  Queries are STREAMED in
// batches of Qb (65,536 by default): every rank scores the batch against its shard (K2), the per-shard K-lists are
// exchanged all-to-all so that each rank receives the lists of the 1/world of the batch it owns, K3 merges them, and the
// merged rows go back to the host.  Two streams per rank: K2 of batch b+1 runs on the compute stream while the exchange,
// merge and download of batch b run on the communication stream (double-buffered by batch parity).  Host memory is
// O(Qb): two pinned query buffers and two pinned result buffers per rank, whatever nq and nb are.
//
// Transports behind one interface (rg_comm):
//   RCCL    ncclSend/ncclRecv grouped per batch (point-to-point xGMI links: every pair of GPUs has its own link, so the
//           all-to-all is one hop, no ring).  librccl is dlopen'ed on first use -- a process that already holds an RCCL
//           (PyTorch bundles one) keeps exactly one copy -- either one communicator per process (rg_comm_init_rank,
//           the torch.distributed / MPI form) or all ranks in this process (rg_comm_init_local, the CLI's --devices).
//   local   ranks are threads of this process and move the lists with peer-to-peer hipMemcpyAsync ordered by events;
//           used when several ranks share one device (tests on a one-GPU box) or RCCL cannot be loaded.
#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // declarations only; the library is dlopen'ed
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "rg.h"
#include "rg_internal.h"

using rg::set_error;

#define RG_HIP(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return set_error(RG_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));        \
    } while (0)

namespace rg {

// ---------------------------------------------------------------------------------------------------- RCCL, dlopen'ed
struct RcclApi {
    void *h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
};

static RcclApi *rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        if (getenv("RG_NO_RCCL")) { api.err = "RG_NO_RCCL is set"; return; }
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.h) break;
        }
        if (!api.h) { api.err = std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "?"); return; }
#define RG_SYM(field, sym) api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.h, sym)); if (!api.field) { api.err = std::string("librccl lacks ") + sym; api.h = nullptr; return; }
        RG_SYM(GetUniqueId, "ncclGetUniqueId") RG_SYM(CommInitRank, "ncclCommInitRank") RG_SYM(CommInitAll, "ncclCommInitAll")
        RG_SYM(CommDestroy, "ncclCommDestroy") RG_SYM(GroupStart, "ncclGroupStart") RG_SYM(GroupEnd, "ncclGroupEnd")
        RG_SYM(Send, "ncclSend") RG_SYM(Recv, "ncclRecv") RG_SYM(GetErrorString, "ncclGetErrorString")
#undef RG_SYM
    });
    return api.h ? &api : nullptr;
}

#define RG_NCCL(expr)                                                                                              \
    do {                                                                                                           \
        ncclResult_t r_ = (expr);                                                                                  \
        if (r_ != ncclSuccess) return set_error(RG_ERR_DEVICE, std::string(#expr) + ": " + rg::rccl_api()->GetErrorString(r_)); \
    } while (0)

// ------------------------------------------------------------------------------------------------- local transport
// ranks = threads of one process.  Per batch parity every rank publishes its K2 output buffers and an event; peers copy
// the rows they own straight out of them (peer-to-peer over xGMI when the ranks sit on different devices).
struct LocalGroup {
    int world = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long gen = 0;
    bool failed = false;
    int refs = 0;
    struct Slot { const uint32_t *ids = nullptr; const float *vals = nullptr; hipEvent_t ready = nullptr, read_done = nullptr; };
    std::vector<Slot> slot[2];
    bool barrier() {   // false: a rank failed, give up
        std::unique_lock<std::mutex> lk(mu);
        if (failed) return false;
        const unsigned long g = gen;
        if (++arrived == world) { arrived = 0; ++gen; cv.notify_all(); return true; }
        cv.wait(lk, [&] { return gen != g || failed; });
        return !failed;
    }
    void fail() { std::lock_guard<std::mutex> lk(mu); failed = true; cv.notify_all(); }
    // buffers of a rank that gave up: peers may still have copies out of them queued on their own streams, so they are
    // released with the group (after every rank has left and drained), not by the failing rank
    std::vector<std::shared_ptr<void>> parked;
    void park(std::shared_ptr<void> p) { std::lock_guard<std::mutex> lk(mu); parked.push_back(std::move(p)); }
};

}  // namespace rg

struct rg_comm {
    int rank = 0, world = 1, device = 0;
    ncclComm_t nccl = nullptr;
    rg::LocalGroup *local = nullptr;
};

namespace rg {

static std::vector<std::pair<uint32_t, uint32_t>> ranges_of(uint32_t n, int world) {   // balanced contiguous ranges
    std::vector<std::pair<uint32_t, uint32_t>> r((size_t)world);
    const uint32_t per = n / (uint32_t)world, extra = n % (uint32_t)world;
    for (uint32_t i = 0; i < (uint32_t)world; ++i) {
        const uint32_t lo = i * per + std::min(i, extra);
        r[i] = {lo, lo + per + (i < extra ? 1u : 0u)};
    }
    return r;
}

// The exchange of one query batch, as rank `rank` sees it: for every peer j, the rows of its own K-lists that j owns go to j
// (send_row0 / send_rows: rows of the batch, i.e. element offset send_row0 * K in d_ids / d_vals), and j's lists of the rows THIS
// rank owns arrive in slot j of the receive buffer (recv_slot_row0 = j * per: element offset j * per * K in d_rids / d_rvals).
// One function for the RCCL calls below and for rg_gt_exchange_plan (the schedule as data: tests drive it for eight ranks on a
// box without a GPU).
struct XferOp { uint32_t peer, send_row0, send_rows, recv_slot_row0, recv_rows; };
static std::vector<XferOp> exchange_ops(int world, int rank, uint32_t nqb, uint32_t per) {
    const auto own = ranges_of(nqb, world);
    const uint32_t n_own = own[(size_t)rank].second - own[(size_t)rank].first;
    std::vector<XferOp> ops;
    for (int j = 0; j < world; ++j) {
        const uint32_t jl = own[(size_t)j].first, jn = own[(size_t)j].second - jl;
        ops.push_back({(uint32_t)j, jl, jn, (uint32_t)j * per, n_own});
    }
    return ops;
}

// where a rank's query batches come from and where its merged rows go (memory arrays or files)
struct GtIo {
    std::function<rg_status(uint32_t q0, uint32_t n, float *dst, uint32_t dst_stride)> fill;      // rows -> pinned, zero padded
    std::function<rg_status(uint32_t row0, uint32_t n, const uint32_t *ids, const float *vals)> emit;
};

struct RankBufs {   // everything a rank allocates, released on every exit path
    int device = 0;
    hipStream_t s_comp = nullptr, s_comm = nullptr;
    hipEvent_t ev_k2[2] = {nullptr, nullptr}, ev_x[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr}, ev_q[2] = {nullptr, nullptr},
               ev_rd[2] = {nullptr, nullptr};
    float *h_q[2] = {nullptr, nullptr}, *d_q[2] = {nullptr, nullptr};
    uint32_t *d_ids[2] = {nullptr, nullptr}, *d_rids[2] = {nullptr, nullptr}, *d_oids[2] = {nullptr, nullptr}, *h_oids[2] = {nullptr, nullptr};
    float *d_vals[2] = {nullptr, nullptr}, *d_rvals[2] = {nullptr, nullptr}, *d_ovals[2] = {nullptr, nullptr}, *h_ovals[2] = {nullptr, nullptr};
    GtWorkspace ws;   // K2's scratch, this rank's own
    ~RankBufs() {
        (void)hipSetDevice(device);
        if (s_comp) (void)hipStreamSynchronize(s_comp);
        if (s_comm) (void)hipStreamSynchronize(s_comm);
        for (int p = 0; p < 2; ++p) {
            for (hipEvent_t e : {ev_k2[p], ev_x[p], ev_out[p], ev_q[p], ev_rd[p]}) if (e) (void)hipEventDestroy(e);
            for (void *h : {(void *)h_q[p], (void *)h_oids[p], (void *)h_ovals[p]}) if (h) (void)hipHostFree(h);
            for (void *d : {(void *)d_q[p], (void *)d_ids[p], (void *)d_rids[p], (void *)d_oids[p], (void *)d_vals[p], (void *)d_rvals[p], (void *)d_ovals[p]})
                if (d) (void)hipFree(d);
        }
        gt_workspace_free(&ws);
        if (s_comp) (void)hipStreamDestroy(s_comp);
        if (s_comm) (void)hipStreamDestroy(s_comm);
    }
};

// one rank's whole job.  d_base: its shard (nb_shard rows at bstride floats, already normalised for cosine).
static rg_status gt_rank_body(rg_comm *cm, const float *d_base, uint32_t nb_shard, uint32_t bstride, uint32_t id_base, uint32_t nq,
                              uint32_t dim, int metric, uint32_t K, uint32_t batch, const GtIo &io, RankBufs &B) {
    const int world = cm->world, rank = cm->rank;
    if (K == 0 || K > nb_shard) return set_error(RG_ERR_ARG, "K must be in [1, rows in the shard]");
    if ((uint64_t)world * K > 1024) return set_error(RG_ERR_ARG, "world * K larger than 1024 is not supported by the merge");
    if (nq == 0) return RG_OK;
    const uint32_t ad = aligned_dim(dim);
    const uint32_t Qb = std::min<uint32_t>(nq, batch ? batch : 65536u);
    const uint32_t per = (Qb + (uint32_t)world - 1) / (uint32_t)world;   // longest owned range of a batch
    const int m = metric == RG_METRIC_COSINE ? RG_METRIC_IP : metric;
    RG_HIP(hipSetDevice(cm->device));
    B.device = cm->device;
    RG_HIP(hipStreamCreateWithFlags(&B.s_comp, hipStreamNonBlocking));
    RG_HIP(hipStreamCreateWithFlags(&B.s_comm, hipStreamNonBlocking));
    for (int p = 0; p < 2; ++p) {
        for (hipEvent_t *e : {&B.ev_k2[p], &B.ev_x[p], &B.ev_out[p], &B.ev_q[p], &B.ev_rd[p]}) RG_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
        RG_HIP(hipHostMalloc(&B.h_q[p], (size_t)Qb * ad * 4));
        RG_HIP(hipMalloc(&B.d_q[p], (size_t)Qb * ad * 4));
        RG_HIP(hipMalloc(&B.d_ids[p], (size_t)Qb * K * 4));
        RG_HIP(hipMalloc(&B.d_vals[p], (size_t)Qb * K * 4));
        RG_HIP(hipMalloc(&B.d_rids[p], (size_t)world * per * K * 4));
        RG_HIP(hipMalloc(&B.d_rvals[p], (size_t)world * per * K * 4));
        RG_HIP(hipMemset(B.d_rids[p], 0, (size_t)world * per * K * 4));
        RG_HIP(hipMemset(B.d_rvals[p], 0, (size_t)world * per * K * 4));
        RG_HIP(hipMalloc(&B.d_oids[p], (size_t)per * K * 4));
        RG_HIP(hipMalloc(&B.d_ovals[p], (size_t)per * K * 4));
        RG_HIP(hipHostMalloc(&B.h_oids[p], (size_t)per * K * 4));
        RG_HIP(hipHostMalloc(&B.h_ovals[p], (size_t)per * K * 4));
    }
    RG_HIP(hipDeviceSynchronize());
    LocalGroup *lg = cm->local;
    RcclApi *nc = cm->nccl ? rccl_api() : nullptr;
    const uint32_t nbatch = (nq + Qb - 1) / Qb;
    struct Done { uint32_t row0 = 0, n = 0; bool live = false; } done[2];
    auto collect = [&](int p) -> rg_status {   // merged rows of the batch that used parity p -> the caller
        if (!done[p].live) return RG_OK;
        RG_HIP(hipEventSynchronize(B.ev_out[p]));
        done[p].live = false;
        return done[p].n ? io.emit(done[p].row0, done[p].n, B.h_oids[p], B.h_ovals[p]) : RG_OK;
    };
    for (uint32_t b = 0; b < nbatch; ++b) {
        const int p = (int)(b & 1u);
        const uint32_t q0 = b * Qb, nqb = std::min(Qb, nq - q0);
        rg_status st = collect(p);          // batch b-2 is finished with this parity's host buffers (and its query upload)
        if (st != RG_OK) return st;
        st = io.fill(q0, nqb, B.h_q[p], ad);
        if (st != RG_OK) return st;
        if (metric == RG_METRIC_COSINE) rg_normalize_rows(B.h_q[p], nqb, ad, dim);
        const auto own = ranges_of(nqb, world);
        const uint32_t lo = own[(size_t)rank].first, n_own = own[(size_t)rank].second - lo;
        // ---- compute stream: upload + K2 over the shard
        if (b >= 2) {
            RG_HIP(hipStreamWaitEvent(B.s_comp, B.ev_x[p], 0));       // exchange of batch b-2 no longer reads d_ids/d_vals[p]
            if (lg)                                                    // nor do the peers (their copies of batch b-2)
                for (int j = 0; j < world; ++j) RG_HIP(hipStreamWaitEvent(B.s_comp, lg->slot[p][(size_t)j].read_done, 0));
        }
        RG_HIP(hipMemcpyAsync(B.d_q[p], B.h_q[p], (size_t)nqb * ad * 4, hipMemcpyHostToDevice, B.s_comp));
        RG_HIP(hipEventRecord(B.ev_q[p], B.s_comp));
        st = gt_shard_ws(d_base, nb_shard, bstride, B.d_q[p], nqb, ad, ad, m, K, id_base, B.d_ids[p], B.d_vals[p], cm->device, B.s_comp, &B.ws);
        if (st != RG_OK) return st;
        RG_HIP(hipEventRecord(B.ev_k2[p], B.s_comp));
        // ---- communication stream: all-to-all of the K-lists (rank j receives the rows of the range it owns), K3, download
        RG_HIP(hipStreamWaitEvent(B.s_comm, B.ev_k2[p], 0));
        if (world == 1 && !nc) {
            RG_HIP(hipMemcpyAsync(B.d_oids[p], B.d_ids[p], (size_t)nqb * K * 4, hipMemcpyDeviceToDevice, B.s_comm));
            RG_HIP(hipMemcpyAsync(B.d_ovals[p], B.d_vals[p], (size_t)nqb * K * 4, hipMemcpyDeviceToDevice, B.s_comm));
            RG_HIP(hipEventRecord(B.ev_x[p], B.s_comm));
        } else {
            if (nc) {
                RG_NCCL(nc->GroupStart());
                for (const XferOp &op : exchange_ops(world, rank, nqb, per)) {
                    const int j = (int)op.peer;
                    if (op.send_rows) {
                        RG_NCCL(nc->Send(B.d_ids[p] + (size_t)op.send_row0 * K, (size_t)op.send_rows * K, ncclUint32, j, cm->nccl, B.s_comm));
                        RG_NCCL(nc->Send(B.d_vals[p] + (size_t)op.send_row0 * K, (size_t)op.send_rows * K, ncclFloat32, j, cm->nccl, B.s_comm));
                    }
                    if (op.recv_rows) {
                        RG_NCCL(nc->Recv(B.d_rids[p] + (size_t)op.recv_slot_row0 * K, (size_t)op.recv_rows * K, ncclUint32, j, cm->nccl, B.s_comm));
                        RG_NCCL(nc->Recv(B.d_rvals[p] + (size_t)op.recv_slot_row0 * K, (size_t)op.recv_rows * K, ncclFloat32, j, cm->nccl, B.s_comm));
                    }
                }
                RG_NCCL(nc->GroupEnd());
            } else {
                LocalGroup::Slot &mine = lg->slot[p][(size_t)rank];
                mine.ids = B.d_ids[p]; mine.vals = B.d_vals[p]; mine.ready = B.ev_k2[p]; mine.read_done = B.ev_rd[p];
                if (!lg->barrier()) return set_error(RG_ERR_DEVICE, "a peer rank failed");
                for (int j = 0; j < world && n_own; ++j) {
                    const LocalGroup::Slot &sj = lg->slot[p][(size_t)j];
                    RG_HIP(hipStreamWaitEvent(B.s_comm, sj.ready, 0));
                    RG_HIP(hipMemcpyAsync(B.d_rids[p] + (size_t)j * per * K, sj.ids + (size_t)lo * K, (size_t)n_own * K * 4, hipMemcpyDefault, B.s_comm));
                    RG_HIP(hipMemcpyAsync(B.d_rvals[p] + (size_t)j * per * K, sj.vals + (size_t)lo * K, (size_t)n_own * K * 4, hipMemcpyDefault, B.s_comm));
                }
                RG_HIP(hipEventRecord(B.ev_rd[p], B.s_comm));
                if (!lg->barrier()) return set_error(RG_ERR_DEVICE, "a peer rank failed");   // every read_done of this batch is recorded
            }
            RG_HIP(hipEventRecord(B.ev_x[p], B.s_comm));
            if (n_own) {
                st = rg_gt_merge_dev(B.d_rids[p], B.d_rvals[p], (uint32_t)world, per, K, m, B.d_oids[p], B.d_ovals[p], cm->device, B.s_comm);
                if (st != RG_OK) return st;
            }
        }
        if (n_own) {
            RG_HIP(hipMemcpyAsync(B.h_oids[p], B.d_oids[p], (size_t)n_own * K * 4, hipMemcpyDeviceToHost, B.s_comm));
            RG_HIP(hipMemcpyAsync(B.h_ovals[p], B.d_ovals[p], (size_t)n_own * K * 4, hipMemcpyDeviceToHost, B.s_comm));
        }
        RG_HIP(hipEventRecord(B.ev_out[p], B.s_comm));
        done[p].row0 = q0 + lo; done[p].n = n_own; done[p].live = true;
    }
    for (int p : {(int)(nbatch & 1u), (int)((nbatch + 1) & 1u)}) {   // older batch first
        rg_status st = collect(p);
        if (st != RG_OK) return st;
    }
    RG_HIP(hipStreamSynchronize(B.s_comp));
    RG_HIP(hipStreamSynchronize(B.s_comm));
    // in-process transport: peers copy straight out of this rank's K2 buffers, and their last copies may still be queued
    // when this rank is done -- nobody releases anything before every rank has drained its own streams
    if (lg && !lg->barrier()) return set_error(RG_ERR_DEVICE, "a peer rank failed");
    return RG_OK;
}

// Every exit of a rank that is not RG_OK -- a precondition that fails on this rank only (K > rows of ITS shard), an
// allocation, a fill / emit callback, a kernel -- must release the peers of the in-process transport, which would
// otherwise wait in LocalGroup::barrier() for ever; and the buffers they may still be copying from must outlive them.
static rg_status gt_rank_run(rg_comm *cm, const float *d_base, uint32_t nb_shard, uint32_t bstride, uint32_t id_base, uint32_t nq,
                             uint32_t dim, int metric, uint32_t K, uint32_t batch, const GtIo &io) {
    std::shared_ptr<RankBufs> B = std::make_shared<RankBufs>();
    const rg_status st = gt_rank_body(cm, d_base, nb_shard, bstride, id_base, nq, dim, metric, K, batch, io, *B);
    if (st != RG_OK && cm->local) {
        const std::string msg = rg_last_error();
        cm->local->fail();
        (void)hipSetDevice(cm->device);
        if (B->s_comp) (void)hipStreamSynchronize(B->s_comp);     // nothing of this rank's is queued any more
        if (B->s_comm) (void)hipStreamSynchronize(B->s_comm);
        cm->local->park(B);
        return set_error(st, msg);
    }
    return st;
}

// rows [row0, row0+n) of a host matrix (or file) -> device at the aligned stride, zero padded, cosine rows normalised;
// staged through two pinned chunks, so host memory stays O(chunk)
static rg_status upload_rows(const std::function<rg_status(uint32_t, uint32_t, float *, uint32_t)> &fill, uint32_t row0, uint32_t n,
                             uint32_t dim, bool cosine, float *d_dst, hipStream_t s) {
    const uint32_t ad = aligned_dim(dim), chunk = std::max<uint32_t>(1u, std::min<uint32_t>(n, (64u << 20) / (ad * 4u)));
    struct Pin { float *p[2] = {nullptr, nullptr}; hipEvent_t e[2] = {nullptr, nullptr};
                 ~Pin() { for (int i = 0; i < 2; ++i) { if (e[i]) (void)hipEventDestroy(e[i]); if (p[i]) (void)hipHostFree(p[i]); } } } P;
    for (int i = 0; i < 2; ++i) { RG_HIP(hipHostMalloc(&P.p[i], (size_t)chunk * ad * 4)); RG_HIP(hipEventCreateWithFlags(&P.e[i], hipEventDisableTiming)); }
    uint32_t it = 0;
    for (uint32_t r = 0; r < n; r += chunk, ++it) {
        const int b = (int)(it & 1u);
        const uint32_t m = std::min(chunk, n - r);
        if (it >= 2) RG_HIP(hipEventSynchronize(P.e[b]));
        rg_status st = fill(row0 + r, m, P.p[b], ad);
        if (st != RG_OK) return st;
        if (cosine) rg_normalize_rows(P.p[b], m, ad, dim);
        RG_HIP(hipMemcpyAsync(d_dst + (size_t)r * ad, P.p[b], (size_t)m * ad * 4, hipMemcpyHostToDevice, s));
        RG_HIP(hipEventRecord(P.e[b], s));
    }
    RG_HIP(hipStreamSynchronize(s));
    return RG_OK;
}

static rg_status comm_init_local(const int *devices, int nranks, std::vector<rg_comm *> &out) {
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0)
        return set_error(RG_ERR_DEVICE, "no HIP device visible: the gfx950 path cannot run (there is no CPU fallback)");
    bool distinct = true;
    for (int i = 0; i < nranks; ++i) {
        if (devices[i] < 0 || devices[i] >= visible) return set_error(RG_ERR_ARG, "device index out of range");
        for (int j = 0; j < i; ++j) distinct = distinct && devices[i] != devices[j];
    }
    out.assign((size_t)nranks, nullptr);
    for (int i = 0; i < nranks; ++i) { out[(size_t)i] = new rg_comm(); out[(size_t)i]->rank = i; out[(size_t)i]->world = nranks; out[(size_t)i]->device = devices[i]; }
    // (RG_GT_FORCE_RCCL: a single rank still goes through RCCL -- send/recv to itself -- so that a one-GPU box can exercise it)
    RcclApi *nc = (distinct && (nranks > 1 || getenv("RG_GT_FORCE_RCCL"))) ? rccl_api() : nullptr;
    if (nc) {
        std::vector<ncclComm_t> cs((size_t)nranks);
        if (nc->CommInitAll(cs.data(), nranks, devices) == ncclSuccess) {
            for (int i = 0; i < nranks; ++i) out[(size_t)i]->nccl = cs[(size_t)i];
            return RG_OK;
        }
    }
    if (nranks > 1) {   // in-process transport
        LocalGroup *lg = new LocalGroup();
        lg->world = nranks; lg->refs = nranks;
        lg->slot[0].resize((size_t)nranks); lg->slot[1].resize((size_t)nranks);
        for (int i = 0; i < nranks; ++i) {
            out[(size_t)i]->local = lg;
            for (int j = 0; j < i; ++j)
                if (devices[i] != devices[j]) {   // peer access for the direct copies (ignored if already on / unsupported)
                    (void)hipSetDevice(devices[i]); (void)hipDeviceEnablePeerAccess(devices[j], 0);
                    (void)hipSetDevice(devices[j]); (void)hipDeviceEnablePeerAccess(devices[i], 0);
                    (void)hipGetLastError();
                }
        }
    }
    return RG_OK;
}

static void comm_destroy(rg_comm *c) {
    if (!c) return;
    if (c->nccl && rccl_api()) { (void)hipSetDevice(c->device); (void)rccl_api()->CommDestroy(c->nccl); }
    if (c->local) {
        bool last;
        { std::lock_guard<std::mutex> lk(c->local->mu); last = --c->local->refs == 0; }
        if (last) delete c->local;
    }
    delete c;
}

// the whole job inside this process: one thread per rank (device list entry)
static rg_status run_local(const std::vector<int> &devs, uint32_t nb, uint32_t nq, uint32_t dim, int metric, uint32_t K, uint32_t batch,
                           const std::function<rg_status(uint32_t, uint32_t, float *, uint32_t)> &fill_base, const GtIo &io) {
    const int nd = (int)devs.size();
    std::vector<rg_comm *> comms;
    rg_status st = comm_init_local(devs.data(), nd, comms);
    if (st != RG_OK) return st;
    const auto shards = ranges_of(nb, nd);
    const uint32_t ad = aligned_dim(dim);
    std::vector<rg_status> rs((size_t)nd, RG_OK);
    std::vector<std::string> msgs((size_t)nd);
    auto work = [&](int r) {
        rg_comm *cm = comms[(size_t)r];
        const uint32_t lo = shards[(size_t)r].first, n = shards[(size_t)r].second - lo;
        float *d_base = nullptr;
        rg_status s = RG_OK;
        hipStream_t up = nullptr;
        if (hipSetDevice(cm->device) != hipSuccess || hipMalloc(&d_base, std::max<size_t>((size_t)n * ad * 4, 16)) != hipSuccess ||
            hipStreamCreateWithFlags(&up, hipStreamNonBlocking) != hipSuccess)
            s = set_error(RG_ERR_DEVICE, "cannot allocate the base shard");
        if (s == RG_OK) s = upload_rows(fill_base, lo, n, dim, metric == RG_METRIC_COSINE, d_base, up);
        if (s == RG_OK) s = gt_rank_run(cm, d_base, n, ad, lo, nq, dim, metric, K, batch, io);
        if (s != RG_OK) { msgs[(size_t)r] = rg_last_error(); if (cm->local) cm->local->fail(); }
        if (up) (void)hipStreamDestroy(up);
        if (d_base) (void)hipFree(d_base);
        rs[(size_t)r] = s;
    };
    if (nd == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int r = 0; r < nd; ++r) th.emplace_back(work, r);
        for (auto &t : th) t.join();
    }
    for (rg_comm *c : comms) comm_destroy(c);
    for (int r = 0; r < nd; ++r)
        if (rs[(size_t)r] != RG_OK) return set_error(rs[(size_t)r], msgs[(size_t)r]);
    return RG_OK;
}

static uint32_t batch_knob() {
    const char *e = getenv("RG_GT_BATCH");   // test hook: small batches exercise the double-buffered pipeline
    return e ? (uint32_t)std::max(1, atoi(e)) : 0u;
}

static std::vector<int> device_list(const int *devices, int ndev, uint32_t nb, uint32_t K) {
    std::vector<int> devs;
    if (devices && ndev > 0) devs.assign(devices, devices + ndev);
    else devs.push_back(0);
    while (devs.size() > 1 && (nb / devs.size() < K || devs.size() * (size_t)K > 1024)) devs.pop_back();
    return devs;
}

}  // namespace rg

extern "C" {

rg_status rg_comm_unique_id(void *id128) {
    if (!id128) return set_error(RG_ERR_ARG, "null argument");
    rg::RcclApi *nc = rg::rccl_api();
    if (!nc) return set_error(RG_ERR_DEVICE, "RCCL is not available");
    ncclUniqueId id;
    RG_NCCL(nc->GetUniqueId(&id));
    std::memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return RG_OK;
}

rg_status rg_gt_exchange_plan(int world, int rank, uint32_t nq, uint32_t batch, uint32_t *out, uint64_t cap_words, uint64_t *n_words) {
    if (world < 1 || rank < 0 || rank >= world || !n_words) return rg::set_error(RG_ERR_ARG, "rg_gt_exchange_plan: bad world / rank / null argument");
    const uint32_t Qb = std::min<uint32_t>(std::max<uint32_t>(nq, 1u), batch ? batch : 65536u);
    const uint32_t per = (Qb + (uint32_t)world - 1) / (uint32_t)world;
    const uint32_t nbatch = nq ? (nq + Qb - 1) / Qb : 0;
    uint64_t n = 0;
    for (uint32_t b = 0; b < nbatch; ++b) {
        const uint32_t q0 = b * Qb, nqb = std::min(Qb, nq - q0);
        const auto own = rg::ranges_of(nqb, world);
        for (const rg::XferOp &op : rg::exchange_ops(world, rank, nqb, per)) {
            const uint32_t rec[10] = {b, b & 1u, q0, nqb, op.peer, op.send_row0, op.send_rows, op.recv_slot_row0, op.recv_rows, q0 + own[(size_t)rank].first};
            if (out && n + 10 <= cap_words) memcpy(out + n, rec, sizeof rec);
            n += 10;
        }
    }
    *n_words = n;
    if (out && n > cap_words) return rg::set_error(RG_ERR_ARG, "rg_gt_exchange_plan: output buffer too small");
    return RG_OK;
}

rg_status rg_comm_init_rank(const void *id128, int rank, int world, int device, rg_comm **out) {
    if (!out || world < 1 || rank < 0 || rank >= world) return set_error(RG_ERR_ARG, "bad rank / world");
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0)
        return set_error(RG_ERR_DEVICE, "no HIP device visible: the gfx950 path cannot run (there is no CPU fallback)");
    if (device < 0 || device >= visible) return set_error(RG_ERR_ARG, "device index out of range");
    rg_comm *c = new rg_comm();
    c->rank = rank; c->world = world; c->device = device;
    if (world > 1) {
        rg::RcclApi *nc = rg::rccl_api();
        if (!nc || !id128) { delete c; return set_error(RG_ERR_DEVICE, "RCCL is not available"); }
        ncclUniqueId id;
        std::memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
        if (hipSetDevice(device) != hipSuccess) { delete c; return set_error(RG_ERR_DEVICE, "cannot select the device"); }
        ncclResult_t r = nc->CommInitRank(&c->nccl, world, id, rank);
        if (r != ncclSuccess) { delete c; return set_error(RG_ERR_DEVICE, std::string("ncclCommInitRank: ") + nc->GetErrorString(r)); }
    }
    *out = c;
    return RG_OK;
}

rg_status rg_comm_init_local(const int *devices, int nranks, rg_comm **out) {
    if (!devices || nranks < 1 || !out) return set_error(RG_ERR_ARG, "null argument");
    std::vector<rg_comm *> v;
    rg_status st = rg::comm_init_local(devices, nranks, v);
    if (st != RG_OK) return st;
    for (int i = 0; i < nranks; ++i) out[i] = v[(size_t)i];
    return RG_OK;
}

int rg_comm_uses_rccl(const rg_comm *comm) { return comm && comm->nccl ? 1 : 0; }

void rg_comm_destroy(rg_comm *comm) { rg::comm_destroy(comm); }

rg_status rg_groundtruth_rank(rg_comm *comm, const float *d_base_shard, uint32_t nb_shard, uint32_t bstride, uint32_t id_base,
                              const float *queries, uint32_t nq, uint32_t qstride, uint32_t dim, int metric, uint32_t K, uint32_t batch,
                              uint32_t *out_ids, float *out_dists) {
    if (!comm || !d_base_shard || !queries || !out_ids || !out_dists) return set_error(RG_ERR_ARG, "null argument");
    if (metric != RG_METRIC_L2 && metric != RG_METRIC_IP && metric != RG_METRIC_COSINE) return set_error(RG_ERR_ARG, "Unknown distance type");
    if (qstride < dim) return set_error(RG_ERR_ARG, "query stride smaller than the dimension");
    rg::GtIo io;
    io.fill = [&](uint32_t q0, uint32_t n, float *dst, uint32_t ds) -> rg_status {
        for (size_t i = 0; i < n; ++i) {
            std::memcpy(dst + i * ds, queries + ((size_t)q0 + i) * qstride, (size_t)dim * 4);
            if (ds > dim) std::memset(dst + i * ds + dim, 0, (size_t)(ds - dim) * 4);
        }
        return RG_OK;
    };
    io.emit = [&](uint32_t row0, uint32_t n, const uint32_t *ids, const float *vals) -> rg_status {
        std::memcpy(out_ids + (size_t)row0 * K, ids, (size_t)n * K * 4);
        std::memcpy(out_dists + (size_t)row0 * K, vals, (size_t)n * K * 4);
        return RG_OK;
    };
    return rg::gt_rank_run(comm, d_base_shard, nb_shard, bstride, id_base, nq, dim, metric, K, batch, io);
}

rg_status rg_groundtruth_mem(const float *base, uint32_t nb, uint32_t bstride, const float *queries, uint32_t nq,
                             uint32_t qstride, uint32_t dim, int metric, uint32_t K, uint32_t *out_ids,
                             float *out_dists, const int *devices, int ndev) {
    if (!base || !queries || !out_ids || !out_dists) return set_error(RG_ERR_ARG, "null argument");
    if (K == 0 || K > nb) return set_error(RG_ERR_ARG, "K must be in [1, number of base rows]");
    if (bstride < dim || qstride < dim) return set_error(RG_ERR_ARG, "stride smaller than the dimension");
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0)
        return set_error(RG_ERR_DEVICE, "no HIP device visible: the gfx950 path cannot run (there is no CPU fallback)");
    auto rows_from = [dim](const float *src, uint32_t stride) {
        return [src, stride, dim](uint32_t r0, uint32_t n, float *dst, uint32_t ds) -> rg_status {
            for (size_t i = 0; i < n; ++i) {
                std::memcpy(dst + i * ds, src + ((size_t)r0 + i) * stride, (size_t)dim * 4);
                if (ds > dim) std::memset(dst + i * ds + dim, 0, (size_t)(ds - dim) * 4);
            }
            return RG_OK;
        };
    };
    rg::GtIo io;
    io.fill = rows_from(queries, qstride);
    io.emit = [&](uint32_t row0, uint32_t n, const uint32_t *ids, const float *vals) -> rg_status {
        std::memcpy(out_ids + (size_t)row0 * K, ids, (size_t)n * K * 4);
        std::memcpy(out_dists + (size_t)row0 * K, vals, (size_t)n * K * 4);
        return RG_OK;
    };
    return rg::run_local(rg::device_list(devices, ndev, nb, K), nb, nq, dim, metric, K, rg::batch_knob(), rows_from(base, bstride), io);
}

/* file form (the CLI twin's body): base shards and query batches are read straight from the .fbin files and the result
 * rows are written straight into the gt file, so host memory stays O(batch) however large the files are */
rg_status rg_groundtruth(const char *base_fbin, const char *query_fbin, const char *gt_out, int metric, uint32_t K,
                         const int *devices, int ndev) {
    if (!base_fbin || !query_fbin || !gt_out) return set_error(RG_ERR_ARG, "null argument");
    uint32_t nb = 0, bd = 0, nq = 0, qd = 0;
    rg_status st = rg_fbin_meta(base_fbin, &nb, &bd);
    if (st != RG_OK) return st;
    st = rg_fbin_meta(query_fbin, &nq, &qd);
    if (st != RG_OK) return st;
    if (bd != qd) return set_error(RG_ERR_ARG, "base and query dimension mismatch");
    if (K == 0 || K > nb) return set_error(RG_ERR_ARG, "K must be in [1, number of base rows]");
    struct Fd { int fd = -1; ~Fd() { if (fd >= 0) close(fd); } } fb, fq, fo;
    fb.fd = open(base_fbin, O_RDONLY);
    fq.fd = open(query_fbin, O_RDONLY);
    if (fb.fd < 0 || fq.fd < 0) return set_error(RG_ERR_IO, "cannot open the input files");
    fo.fd = open(gt_out, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fo.fd < 0) return set_error(RG_ERR_IO, std::string("cannot open ") + gt_out);
    const uint32_t hdr[2] = {nq, K};
    if (pwrite(fo.fd, hdr, 8, 0) != 8 || ftruncate(fo.fd, 8 + (off_t)nq * K * 8) != 0) return set_error(RG_ERR_IO, "cannot size the output file");
    const uint32_t dim = bd;
    auto rows_from = [dim](int fd) {
        return [fd, dim](uint32_t r0, uint32_t n, float *dst, uint32_t ds) -> rg_status {
            if (ds == dim) {
                const size_t bytes = (size_t)n * dim * 4;
                size_t got = 0;
                while (got < bytes) {
                    const ssize_t k = pread(fd, reinterpret_cast<char *>(dst) + got, bytes - got, 8 + (off_t)r0 * dim * 4 + (off_t)got);
                    if (k <= 0) return set_error(RG_ERR_IO, "short read");
                    got += (size_t)k;
                }
                return RG_OK;
            }
            for (size_t i = 0; i < n; ++i) {
                if (pread(fd, dst + i * ds, (size_t)dim * 4, 8 + ((off_t)r0 + (off_t)i) * dim * 4) != (ssize_t)((size_t)dim * 4))
                    return set_error(RG_ERR_IO, "short read");
                std::memset(dst + i * ds + dim, 0, (size_t)(ds - dim) * 4);
            }
            return RG_OK;
        };
    };
    rg::GtIo io;
    io.fill = rows_from(fq.fd);
    const int ofd = fo.fd;
    io.emit = [ofd, nq, K](uint32_t row0, uint32_t n, const uint32_t *ids, const float *vals) -> rg_status {
        const size_t bytes = (size_t)n * K * 4;   // layout: all id rows, then all distance rows (util.h:139-147)
        if (pwrite(ofd, ids, bytes, 8 + (off_t)row0 * K * 4) != (ssize_t)bytes ||
            pwrite(ofd, vals, bytes, 8 + (off_t)nq * K * 4 + (off_t)row0 * K * 4) != (ssize_t)bytes)
            return set_error(RG_ERR_IO, "short write");
        return RG_OK;
    };
    return rg::run_local(rg::device_list(devices, ndev, nb, K), nb, nq, dim, metric, K, rg::batch_knob(), rows_from(fb.fd), io);
}

}  // extern "C"
