// rg_index_struct.h -- the opaque rg_index of include/rg.h (shared by rg_search.hip and the GPU-assisted build)
#pragma once
#include <cstddef>
#include <cstdint>

#include "rg.h"

struct rg_index {
    int device = 0;
    int metric = RG_METRIC_IP;
    uint32_t nd = 0, dim = 0, stride = 0, ep = 0;
    float *d_base = nullptr;
    bool own_base = false;
    // graph
    uint64_t *d_offsets = nullptr;
    uint32_t *d_nbrs = nullptr;
    uint32_t *d_ell = nullptr;
    uint32_t ell_stride = 0;
    uint64_t n_edges = 0;
    uint32_t max_deg = 0;
    // search scratch (lazily sized)
    uint32_t *d_visited = nullptr;
    uint32_t *d_epoch = nullptr;
    uint32_t slots = 0, vwords = 0;
    uint32_t *d_counter = nullptr;
    unsigned long long *d_status = nullptr;
    unsigned long long *h_status = nullptr;  // pinned
    // knobs
    int waves_per_cu = 0;   // 0 = auto
    int rows_per_pass = 0;  // 4*R (R = staging ring depth); 0 = auto: 8 on graphs of average out-degree >= 28, else 4
    int force_csr = 0;
    int diag = 0;
    // 0 = exact visited words in HBM; 1 = LDS exact-match filter only (cmps = evaluations performed);
    // 2 = LDS filter + id log + exact distinct count (K4): everything bit-exact incl. cmps (default)
    int visited_mode = 2;
    uint32_t *d_qlog = nullptr, *d_qlog_n = nullptr, *d_ovf = nullptr;
    size_t qlog_cap_total = 0;
    uint32_t qlog_nq = 0, logcap = 0, qlog_chunk = 0, ovf_nq = 0;
    int log_budget_kb = 16 << 20;  // HBM budget of the id logs (KiB, default 16 GiB = 32768 queries): larger batches are searched in sub-batches
    int log_cap_knob = 0;       // 0 = auto; tests force small logs to exercise the exact fallback
    int count_table_log2 = 15;  // K4 LDS table: 2^15 words = 128 KiB
    bool fast_bf16 = false;      // opt-in non-parity mode: traverse a bf16 copy of the base, exact re-rank of the beam
    bool bf_launch = false;      // transient: the launch being prepared uses the bf16 copy
    uint16_t *d_base_bf = nullptr;
    uint32_t stride_bf = 0;
    bool exact_filter = true;    // mode 0: the LDS filter screens the exact HBM words (hits skip the atomics)
    bool query_in_lds = false;   // K1: force the generic (query staged in LDS) instantiation for d = 200 / 512
    bool count_full_ids = false; // K4: force the full-id bucket form (the half-word form is used when the remainder fits)
    struct Pending { bool active = false; const float *q = nullptr; uint32_t nq = 0, qstride = 0, k = 0, L = 0; uint32_t *ids = nullptr; float *dists = nullptr; uint32_t *cmps = nullptr, *hops = nullptr; } pending;
    int filter_log2 = 0;    // VIS=1: log2 of the LDS filter's 16-bit entries; 0 = automatic (launch_k1)
    uint32_t exact_from_L = 0xffffffffu;   // adaptive default mode: beam widths from here on use the exact HBM words
    struct Tune {                          // timed trial behind that decision (search_dev / search_wait)
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        bool timed = false, is_trial = false;
        int mode = 2;
        uint32_t L = 0, nq = 0, trial_L = 0, filter_ok_upto = 0;
        float filter_per_q = 0.0f;
    } tune;
    int filter_auto = 9;    // the automatic choice of the launch being prepared
    int num_cu = 256;
    size_t lds_per_cu = 160 * 1024;
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
}  // namespace rg
