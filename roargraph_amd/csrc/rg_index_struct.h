// rg_index_struct.h -- the opaque rg_index of include/rg.h (shared by rg_search.hip and the GPU-assisted build)
//
// The index proper (base, adjacency, knobs) is immutable after open.  Everything a search launch writes lives in a
// SearchCtx: one per stream with batches in flight, handed out under the index mutex, so that host threads searching one
// index on distinct streams never share mutable state -- the reference calls SearchRoarGraph from many OpenMP threads
// against one index (tests/test_search_roargraph.cpp:203-209; the only shared mutable state there is the visited-list
// pool behind its mutex, visited_list_pool.h:47-65).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "rg.h"

namespace rg {

// one rg_search_dev call in flight: what rg_search_wait needs to finish it (small; pooled per context)
struct Batch {
    unsigned long long *d_stat = nullptr;  // [0] min over failing queries of (query << 32 | queue size), ~0 = none;
                                           // [1], [2] K4 totals: evaluations performed / distinct nodes
    unsigned long long *h_stat = nullptr;  // pinned copy of d_stat[0..2]; [3] = number of overflowed id logs
    uint32_t *d_ovf = nullptr;             // [0] overflow count, [1] K4 work counter, [2..] queries whose log overflowed
    uint32_t ovf_cap = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    const float *q = nullptr;
    uint32_t nq = 0, qstride = 0, k = 0, L = 0;
    uint32_t *ids = nullptr;
    float *dists = nullptr;
    uint32_t *cmps = nullptr, *hops = nullptr;
    bool counted = false;   // filter + id log + K4 ran: logs that overflowed are recounted in rg_search_wait
    bool timed = false;     // adaptive default: ev0/ev1 bracket the batch
    bool is_trial = false;
    bool cold = false;      // a buffer was (re)allocated while the batch was enqueued: its event pair spans the allocation
    int mode = 2;           // the exact form the batch ran in (0 = HBM words, 2 = filter + log + K4)
};

// per-stream launch state: work-queue head, visited words, id logs, host-form staging; grow-only
struct SearchCtx {
    hipStream_t key = nullptr;   // the caller's stream this context currently serves
    bool keyed = false;          // key is valid (batches pending on it, or reserved)
    bool reserved = false;       // held by a host-form call (rg_search) on its private stream
    hipStream_t own = nullptr;   // private stream for host-form calls
    uint32_t *d_counter = nullptr;
    unsigned long long *d_scratch_stat = nullptr;   // status sink of launches nobody waits for (build mode, recounts)
    uint32_t *d_visited = nullptr, *d_epoch = nullptr;
    uint32_t slots = 0, vwords = 0;
    // byte form of the exact visited set (look-ahead kernel form): one epoch byte per node and slot, its own epochs
    uint32_t *d_vtags = nullptr, *d_epoch8 = nullptr;
    uint32_t tslots = 0, twords = 0;
    uint32_t *d_qlog = nullptr, *d_qlog_n = nullptr;
    uint32_t qlog_nq = 0, logcap = 0, qlog_chunk = 0;
    uint32_t log_holds = 0;      // queries whose logs the last filter+log batch left in d_qlog (0: searched in sub-batches)
    rg_status deferred = RG_OK;  // "not enough results" of batches that were collected before the caller waited
    std::string deferred_msg;
    uint32_t allocs = 0;         // bumped by every (re)allocation of the buffers above
    std::vector<Batch *> pending, spare;
    // host-form staging (rg_search): queries up, results down
    float *d_q = nullptr, *d_dist = nullptr;
    uint32_t *d_ids = nullptr, *d_ch = nullptr;
    size_t q_cap = 0, res_cap = 0, ch_cap = 0;
    float *d_front = nullptr, *cur_front = nullptr;   // shared-frontier scores of the batch being enqueued ([nq][front_stride])
    size_t front_cap = 0;
    uint32_t front_stride = 0;
    void *h_pin = nullptr;       // pinned staging of the host form: queries up, then ids | dists | cmps | hops down
    size_t h_cap = 0;
};

}  // namespace rg

struct rg_index {
    int device = 0;
    int metric = RG_METRIC_IP;
    uint32_t nd = 0, dim = 0, stride = 0, ep = 0;
    float *d_base = nullptr;
    bool own_base = false;
    bool base_copied = false;    // own_base because a caller-owned device base was copied into a balanced buffer (rg_index_open_dev)
    // graph
    uint64_t *d_offsets = nullptr;
    uint32_t *d_nbrs = nullptr;
    uint32_t *d_ell = nullptr;
    uint32_t ell_stride = 0;
    uint64_t n_edges = 0;
    uint32_t max_deg = 0;
    // knobs (rg_index_set; read by every launch, written only between searches)
    int waves_per_cu = 0;   // 0 = auto
    int rows_per_pass = 0;  // 4*R (R = staging ring depth); 0 = auto: 8 on graphs of average out-degree >= 28, else 4
    int force_csr = 0;
    int diag = 0;
    // 0 = exact visited words in HBM; 1 = LDS exact-match filter only (cmps = evaluations performed);
    // 2 = LDS filter + id log + exact distinct count (K4): everything bit-exact incl. cmps (default)
    int visited_mode = 2;
    int log_budget_kb = 16 << 20;  // HBM budget of the id logs per context (KiB, default 16 GiB = 32768 queries): larger batches are searched in sub-batches
    bool gather_roll = true;   // register-staged gather: streamed (set j re-loaded as soon as it is scored) instead of batch by batch
    int visited_bytes = -1;   // look-ahead form: -1 / 1 = one epoch byte per node (marks are plain stores), 0 = the epoch-tagged words
    int visited_budget_kb = 24 << 20;  // HBM budget of the exact visited words per context (KiB, default 24 GiB): caps the slots = the grid of a mode-0 launch
    int visited_uncached = 0;        // knob: exact visited words in 1 = uncached (MTYPE_UC), 2 = fine-grained device memory
    int log_cap_knob = 0;       // 0 = auto; tests force small logs to exercise the exact fallback
    int count_table_log2 = 15;  // K4 LDS table: at most 2^15 words = 128 KiB
    bool count_table_auto = true;   // sized per launch from L_pq (knob "count_table_log2" <= 0) or fixed by the knob
    bool fast_bf16 = false;      // opt-in non-parity mode: traverse a bf16 copy of the base, exact re-rank of the beam
    int multi_expand = 0;        // opt-in non-parity mode: the two closest unexpanded entries are expanded per iteration
    uint16_t *d_base_bf = nullptr;
    uint32_t stride_bf = 0;
    // split rows (d = 200 with ELL adjacency): the first 192 elements of every row at a 768-B stride, the
    // 8-element tails once per edge in adjacency order (+ one slot for the entry point), first edge of every node
    float *d_main = nullptr, *d_etail = nullptr;
    uint32_t *d_tail_off = nullptr;
    uint32_t main_dim = 0, tail_dim = 0;
    bool split_rows = true;      // knob: use them (when they exist)
    bool exact_filter = true;    // mode 0: the LDS filter screens the exact HBM words (hits skip the atomics)
    bool ell_tagged = false;     // ELL neighbour words carry min(255, in-degree) in their top byte (nd <= 2^24)
    int filter_min_indeg = 2;    // knob: the LDS visited filter keeps entries only for nodes of at least this in-degree (a node of in-degree 1 is
                                 // met once per query at most: remembering it is wasted; larger thresholds gain about 1 % on the bench index)
    bool shared_frontier = false;   // opt-in (knob): the first hop of a batch is scored once for all its queries (f-4, third mode; exact)
    uint32_t *d_front_ids = nullptr;   // [front_n]: the entry point, then its neighbours in adjacency order
    uint32_t front_n = 0;
    bool log_early = true;       // knob: the id-log store of a hop leaves right behind the row loads (rg_search_kernel.h: expand)
    int count_in_k1 = -1;        // knob: beams up to this wide count their distinct ids inside K1 (-1 = 40, 0 = never: K4 counts)
    bool adaptive = true;        // knob: 0 = the default visited mode never leaves (or tries to leave) its filter + log + K4 form for the exact tags
    int lset_tags = 1;           // knob: 1 = where the exact LDS set alone does not pay but still holds 0.6 x the visits, its overflow goes to the exact byte tags (VIS = 3 + tags); 2 = wherever it fits (tests); 0 = never
    int front_set = -1;          // knob: look-ahead byte-tag form with an exact set in front of the screen: -1 = 85 % of the region where that holds 0.4 x a query's visits, 0 = never, N = N % always
    bool hub_levels = false;     // ELL neighbour words carry the hub level of the neighbour in their top nibble (computed at open; RG_HUB_BITS=0: no)
    int hub_bits = -1;           // knob: hub bitmap of the visited region: -1 = the look-ahead tag form takes one (size by "hub_pct"), 0 = never,
                                 // m = 2^m bits whatever the region's size suggests (tests, experiments)
    int hub_pct = -1;            // knob: largest share of the visited region the bitmap may take, percent (-1 = 90; 60 at L_pq <= 420)
    uint32_t hub_m_last = 0;     // statistics: log2 of the bitmap of the last search launch (0 = none)
    uint64_t n_plain_allocs = 0; // large buffers of this index that fell back to a plain allocation (one memory class: the slower placement)
    int lset_bytes = 0;          // knob (tests): cap of the exact LDS set's region in bytes (0 = what the launch has)
    int lset = -1;               // knob "lset" (round 4): default visited mode, narrow beams: the exact visited set in LDS (K1 VIS = 3: no id
                                 // log, no K4, no de-duplicating inserts).  -1 = wherever a query's visits fit the LDS a launch can give it,
                                 // 0 = never, N = beams up to N wide whatever the estimate says
    int gather_form = -1;        // register-staged K1: 0 = 16-byte loads + LDS bounce, otherwise compute-layout loads where instantiated
    int lookahead = -1;          // mode 0, knob "lookahead": -1 = automatic (by beam width), 0 = returning atomics, 1 = look-ahead form, 2 = look-ahead
                                 // form without the early guess (same results in every form; rg_search.hip: launch_k1)
    bool adj_dups = false;       // some adjacency list names a node twice (found at open): the look-ahead form is not used
    bool query_in_lds = false;   // K1: force the generic (query staged in LDS) instantiation for d = 200 / 512
    bool count_full_ids = false; // K4: force the full-id bucket form (the half-word form is used when the remainder fits)
    int filter_log2 = 0;    // VIS=1: log2 of the LDS filter's 16-bit entries; 0 = automatic (per launch)
    int filter_fill = 1;    // automatic size only: the filter also takes the LDS the resident queries leave unused
    int num_cu = 256;
    size_t lds_per_cu = 160 * 1024;
    // ---- mutable state, all behind `mu`
    std::mutex mu;
    std::vector<rg::SearchCtx *> ctxs;
    // adaptive default visited mode: beam widths from exact_from_L on use the exact HBM words; decided by a timed trial
    uint32_t exact_from_L = 0xffffffffu;
    uint32_t trial_L = 0, filter_ok_upto = 0;
    float filter_per_q = 0.0f;
    // exact LDS set: nodes a query visits at a beam width (mean of the last counted batch): what the set is sized by
    std::map<uint32_t, float> evals_at;
    // counters (rg_index_stat)
    uint64_t n_batches_lset = 0, n_batches_filter_log = 0, n_batches_exact_hbm = 0, n_batches_filter_only = 0, n_lset_left = 0, n_recounted = 0;
};

namespace rg {
struct uint2_pod { uint32_t x, y; };   // (distance bits, id) pairs of the build-mode expansion list
// K1 in build mode (SearchProjectionGraphInternal, src/index_bipartite.cpp:1279-1350): query i is base row node0+i, the
// node itself is never scored, and the output is the sequence of expanded (popped) nodes instead of the top-k.
rg_status build_search_dev(rg_index *ix, uint32_t node0, uint32_t n, uint32_t L, uint2_pod *d_exp, uint32_t exp_cap,
                           uint32_t *d_nexp, void *stream);
// an index over a caller-owned device base whose ELL adjacency (fixed stride) is overwritten between batches
rg_status build_index_create(const float *d_base, uint32_t nd, uint32_t dim, uint32_t stride, uint32_t ep, int metric,
                             int device, uint32_t ell_stride, rg_index **out);
rg_status build_index_set_ell(rg_index *ix, const uint32_t *h_ell, void *stream);
rg_status build_index_update_rows(rg_index *ix, const uint32_t *d_rows, const uint32_t *d_idx, uint32_t n, void *stream);
// PruneProjectionBaseSearchCandidates (:1846-1940) of n expansion lists on the GPU (rg_build_prune.hip): d_have[i] = length +
// ids of node i's projection list (row stride hs words), d_out[i] = length + pruned ids (row stride M + 1; length
// 0xffffffff = left to the host).  Bit-identical to Builder::prune_search.
bool build_prune_supported(const rg_index *ix, uint32_t M, uint32_t exp_cap);
rg_status build_prune_dev(rg_index *ix, uint32_t node0, uint32_t n, uint32_t M, const uint2_pod *d_exp, uint32_t exp_cap,
                          const uint32_t *d_nexp, const uint32_t *d_have, uint32_t hs, uint32_t *d_out, void *stream);
// PruneBiSearchBaseGetBase (:1612-1694) of n training queries' knn rows on the GPU (rg_build_prune.hip): d_out[i] = length + pruned
// ids (row stride M + 1; length 0xffffffff = left to the host).  Bit-identical to Builder::prune_get_base.
bool build_prune_knn_supported(uint32_t dim, uint32_t M, uint32_t ncol, size_t lds_per_cu);
rg_status build_prune_knn_dev(const float *d_base, uint32_t dim, uint32_t stride, int metric, int device, int num_cu, size_t lds_per_cu,
                              const uint32_t *d_knn, uint32_t n, uint32_t kdim, uint32_t ncol, uint32_t M, uint2_pod *d_exp, uint32_t cap,
                              uint32_t *d_pivots, uint32_t *d_out, void *stream);
}  // namespace rg
