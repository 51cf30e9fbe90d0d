/*
 * rg.h -- C ABI of the MI355X-native RoarGraph hot path (librg_hip.so).
 *
 * Plain pointers and sizes only; no C++ or torch types cross this boundary.
 * Every entry point names the reference interface it replaces (file:line into
 * matchyc/RoarGraph @ 2024_10_08).  INTEGRATION.md shows the binding a
 * maintainer of the reference would add.
 *
 * Conventions kept from the reference:
 *   - smaller score = closer.  Inner product returns -dot (distance.h:223),
 *     L2 returns the squared distance (distance.h:87), cosine = IP on
 *     L2-normalised vectors (index_bipartite.cpp:2679-2684).
 *   - metric codes are efanna2e::Metric values (distance.h:15).
 *   - vectors are row-major fp32 with a row stride of ceil(dim/8)*8 floats,
 *     zero padded (util.h:37-75, 191-199).
 *   - results of rg_search are bit-identical to SearchRoarGraph's for the
 *     same .index / base / query / L_pq.
 *
 * Error handling: no exception crosses the ABI.  Functions return RG_OK or a
 * negative rg_status; rg_last_error() returns the message (thread local), with
 * the reference's wording where it has one ("Data file size wrong!",
 * "not enough results: N, expected: K").
 *
 * Pointer suffixes: plain = host memory, d_ = device (HBM) memory on the
 * index's GPU.  `stream` is a hipStream_t passed as void* (NULL = default).
 */
#ifndef RG_H
#define RG_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int rg_status;
enum {
    RG_OK = 0,
    RG_ERR_IO = -1,         /* cannot open / short read */
    RG_ERR_FORMAT = -2,     /* "Data file size wrong!" and friends */
    RG_ERR_ARG = -3,        /* bad argument (k > L_pq, dim mismatch, ...) */
    RG_ERR_DEVICE = -4,     /* HIP error, or no gfx950 device */
    RG_ERR_NOT_ENOUGH = -5, /* "not enough results" (index_bipartite.cpp:2408-2412) */
    RG_ERR_OOM = -6
};
enum { RG_METRIC_L2 = 0, RG_METRIC_IP = 1, RG_METRIC_COSINE = 4 }; /* distance.h:15 */

typedef struct rg_index rg_index;

const char *rg_last_error(void);
const char *rg_version(void);
/* number of visible HIP devices (0 on a CPU-only host; never fails) */
int rg_device_count(void);

/* ------------------------------------------------------------------ formats
 * Host-side readers/writers with the reference's validation rules. Buffers
 * returned through ** are malloc'ed by the library; release with rg_free. */
void rg_free(void *p);
/* load_meta<float>, util.h:106-127 */
rg_status rg_fbin_meta(const char *path, uint32_t *npts, uint32_t *dim);
/* load_data<float> + data_align, util.h:179-211, 37-75: rows at stride ceil(dim/8)*8, zero padded */
rg_status rg_fbin_load(const char *path, uint32_t *npts, uint32_t *dim, uint32_t *stride, float **data);
rg_status rg_fbin_save(const char *path, const float *data, uint32_t npts, uint32_t dim, uint32_t stride);
/* load_gt_meta / load_gt_data_with_dist, util.h:84-105, 129-155 */
rg_status rg_gt_meta(const char *path, uint32_t *npts, uint32_t *k);
rg_status rg_gt_load(const char *path, uint32_t *npts, uint32_t *k, uint32_t **ids, float **dists);
/* writer of the same layout (what compute_groundtruth emits, README.md:70-74) */
rg_status rg_gt_save(const char *path, const uint32_t *ids, const float *dists, uint32_t npts, uint32_t k);
/* LoadLearnBaseKNN, index_bipartite.cpp:2622-2642: header + ids only */
rg_status rg_knn_ids_load(const char *path, uint32_t *npts, uint32_t *k, uint32_t **ids);
/* LoadProjectionGraph / SaveProjectionGraph, index_bipartite.cpp:2097-2117 / 2606-2619; CSR in memory */
rg_status rg_graph_load(const char *path, uint32_t *nd, uint32_t *ep, uint64_t **offsets, uint32_t **nbrs);
rg_status rg_graph_save(const char *path, uint32_t nd, uint32_t ep, const uint64_t *offsets, const uint32_t *nbrs);
/* ComputeRecall, tests/test_search_roargraph.cpp:23-36 */
float rg_recall(uint32_t nq, uint32_t k, uint32_t gt_dim, const uint32_t *res, const uint32_t *gt);
/* normalize<float>, util.h:214-225 (cosine) */
void rg_normalize_rows(float *data, size_t n, size_t stride, uint32_t dim);

/* ---------------------------------------------------------------- lifecycle
 * Replaces: IndexBipartite(dim, n, metric, nullptr) (index_bipartite.h:27)
 *           -> LoadSearchNeededData(base, "")       (index_bipartite.h:62-64, .cpp:2664-2695)
 *           -> LoadProjectionGraph(file)            (index_bipartite.h:105, .cpp:2097-2117)
 *           -> InitVisitedListPool(T)               (index_bipartite.h:133)
 * The index is immutable after open. */
rg_status rg_index_open(const char *base_fbin, const char *index_path, int metric, int device, rg_index **out);
/* SURVEY 8(b)'s lifecycle signature: one REPLICA per listed device (the search shards by queries, the index is replicated:
 * SURVEY 8(e)); the two files are read once, every replica is uploaded from memory.  out[0 .. ndev) receives the replicas
 * (all null when the call fails); they are searched together with rg_search_sharded and closed one by one.  An rg_index
 * itself stays a single-device object: every device-form entry point takes one stream, which belongs to one device. */
rg_status rg_index_open_multi(const char *base_fbin, const char *index_path, int metric, const int *devices, int ndev, rg_index **out);
/* same from host memory (copied to HBM); offsets has nd+1 entries */
rg_status rg_index_open_mem(const float *base, uint32_t nd, uint32_t dim, uint32_t stride, const uint64_t *offsets,
                            const uint32_t *nbrs, uint32_t ep, int metric, int device, rg_index **out);
/* same from buffers already resident in HBM.  d_base is BORROWED (caller keeps it alive and unchanged);
 * the graph is converted to the library's own layout, d_offsets/d_nbrs may be freed after the call.
 * The searches do not necessarily read d_base itself: for d = 200 the index keeps its own split copy of the rows (7.7 + 4.7 GB at
 * 10M rows), and a base of 2 GiB or more that would be read directly (d = 512) is copied into a buffer balanced over the memory
 * classes of the device when twice its size + 8 GiB is free (RG_COPY_BASE=0 in the environment: never) -- so a later change of
 * d_base is NOT seen by the searches, and freeing it does not free the rows.  rg_index_stat "base_copied" says which it was. */
rg_status rg_index_open_dev(const float *d_base, uint32_t nd, uint32_t dim, uint32_t stride,
                            const uint64_t *d_offsets, const uint32_t *d_nbrs, uint32_t ep, int metric, int device,
                            rg_index **out);
/* Empty destructor of the reference (index_bipartite.cpp:40); here the index's device buffers are released.  RETENTION: its large buffers
 * (>= 2 GiB, built from 1-GiB granules spread over the memory classes) stay MAPPED in the allocator's cache -- up to min(64 GiB, a quarter
 * of the device) per device, RG_MEM_CACHE_GIB -- so that the same index opened again maps nothing anew; rg_mem_release(device) hands that
 * memory back to the device (another allocator in the process, e.g. torch's, cannot see it), and so does any allocation of this library
 * that the device refuses. */
void rg_index_close(rg_index *idx);
rg_status rg_index_info(const rg_index *idx, uint32_t *nd, uint32_t *dim, uint32_t *stride, uint32_t *ep,
                        float *avg_degree, uint32_t *max_degree, int *device);
/* tuning knobs: "waves_per_cu", "rows_per_pass", "filter_log2", "log_cap", "log_budget_kb", "count_table_log2",
 * "count_full_ids", "query_in_lds", "exact_filter", "split_rows", "lookahead", "gather_form", "visited_budget_kb" never change results (0 = automatic where a knob has an
 * automatic choice: "rows_per_pass", "filter_log2", "waves_per_cu").
 * "split_rows" (default 1; d = 200 with the default adjacency layout): searches read a split copy of the base made at
 * open -- the first 192 elements of every row at a 768-byte stride (six whole 128-byte lines instead of the seven an
 * 800-byte row spans) and the 8-element tails once per edge in adjacency order, so one hop's tails come from
 * ceil(degree/4) lines; costs nd * 768 + (edges + 1) * 32 bytes of HBM (environment RG_SPLIT_ROWS=0: never built).
 * "rows_per_pass" = 4 * (passes of 4 rows a query keeps in flight; 16 / 32 use register staging at d = 200).
 * "visited" selects how the visited set is kept:
 *   2 (default) ids, dists, hops AND cmps bit-exact.  Narrow beams (what a query visits fits the LDS): the exact visited set in
 *               LDS, nothing else (knob "lset").  Otherwise LDS exact-match filter + per-query id log + exact distinct count.
 *               Adaptive: once a batch shows the filter re-scoring > 4 % extra nodes at some L_pq (long searches on
 *               indexes with locality), the next batch of that L_pq is a timed trial of mode 0 and the faster of the two
 *               exact forms is kept from that L_pq on -- the same bits either way
 *   1           LDS filter only: ids, dists, hops bit-exact; cmps = evaluations performed (>= the reference's)
 *   0           exact visited words in HBM (the reference's tag array, visited_list_pool.h): everything bit-exact
 * "fast_bf16" = 1 is an OPT-IN mode that is NOT parity with the reference; default 0.  The traversal
 * scores a bf16 copy of the base (made on first use, nd * round_up(dim,128) * 2 bytes of HBM; d = 200 / 512 with the
 * default adjacency layout only, otherwise the knob has no effect), then every beam entry is re-scored with the exact
 * fp32 routine and the k best by exact (distance, id) are returned: out_dists are exact for the returned ids, the ids
 * can differ from the reference's (recall is reported separately by bench.py), cmps = evaluations performed.
 * It combines with "visited": 0 = exact HBM words (no repeated evaluations), 1 / 2 = LDS filter only.
 * Round-3 knobs, none of which changes a result: "lookahead" (-1 automatic / 0 / 1 / 2: form of the exact visited words, see
 * rg_search_kernel.h VIS = 2), "gather_form" (0 = 16-byte loads + LDS bounce, otherwise compute-layout loads where
 * instantiated), "filter_min_indeg" (the LDS filter keeps entries only for neighbours of at least this in-degree; default 2; the adjacency words carry
 * min(15, in-degree), so values above 15 mean 15),
 * "count_in_k1" (beams up to this wide count their distinct ids inside the search kernel; default 40, 0 = never),
 * "log_early", "visited_budget_kb" (cap of the exact visited words per stream; default 24 GiB), "visited_uncached",
 * "visited_bytes" (look-ahead form: one epoch byte per node instead of the epoch-tagged words; default on), "filter_fill"
 * (the LDS visited filter takes the LDS the resident queries leave: any slot count; default on), "gather_roll" (streamed
 * row gather; default on).
 * "shared_frontier" = 1 (opt-in, EXACT: every output stays bit-identical; SURVEY 8 f-4, third mode): every query of a batch
 * starts at the entry point, so the first expansion scores the same rows for all of them -- they are scored once for the
 * batch (rg_front_score_kernel, the exact routine) and the first hop of every query reads the scores.
 * "multi_expand" = 1 is the second OPT-IN mode that is NOT parity (speculative multi-expansion): every
 * iteration pops the TWO closest unexpanded entries and expands both in one adjacency / visited / gather phase, whether or
 * not the second would have been the reference's next pop; twice the fresh neighbours per latency chain, a slightly
 * different visiting order (recall is reported beside it by bench.py).
 * Round-4 knobs, none of which changes a result: "lset" (-1 automatic / 0 never / N = beams up to N wide: the exact visited
 * set in LDS of the default mode at narrow beams, rg_search_kernel.h VIS = 3), "lset_bytes" (tests: cap of that set's LDS
 * region), "lset_tags" (1, default: where that set alone no longer pays but still holds 0.6 x a query's visits, the nodes it has
 * no room for go to the exact byte tags in HBM; 2: wherever it fits; 0: never), "front_set" (look-ahead byte-tag form: -1, default =
 * an exact set in front of the tags where it holds 0.4 x a query's visits, 0 never, N = N % of the LDS region always), "adaptive" (0: the
 * default mode keeps its filter + log form at every width -- bench.py's row-reuse statistics need the logs), "hub_bits" (round 5:
 * the look-ahead byte-tag form keeps an exact bitmap of the launch's hubs -- the nodes of highest in-degree per hashed position -- at the
 * front of its LDS region: -1, default = sized by "hub_pct" (largest share of the region, percent; default 90, 60 at L_pq <= 420), 0 = never, m = 2^m bits). */
rg_status rg_index_set(rg_index *idx, const char *name, int value);
/* Counters of the search path since the index was opened (diagnostics: which form the batches ran in).  Names:
 * "batches_lset" / "batches_filter_log" / "batches_exact_hbm" / "batches_filter_only" (batches enqueued per form),
 * "lset_left" (queries that outgrew their exact LDS set), "recounted" (queries whose cmps the host recounted), "hub_levels" (1: the
 * adjacency carries hub levels), "hub_m_last" (log2 of the hub bitmap of the last search launch, 0 = none), "placement_balanced"
 * (1: every large buffer this index allocated so far -- rows, adjacency, visited tags, id logs -- is spread over the memory
 * classes; 0: at least one fell back to a plain allocation, the slower placement), "plain_allocs" (how many), "base_copied" (1: the
 * searches read the index's own copy of a caller-owned device base, 2: its split copy for d = 200, 0: the caller's buffer). */
rg_status rg_index_stat(const rg_index *idx, const char *name, uint64_t *value);
/* Where the large buffers of the indexes on `device` live (diagnostics; no counterpart in the reference).  The library
 * builds every buffer of 2 GiB and more from 1-GiB granules taken round robin over the memory classes of the device
 * (csrc/rg_mem.hip: K1's random row reads and visited tests run 4 - 12 % faster when rows and tags are spread over the
 * classes than when a plain allocation puts them into one).  buffers = balanced buffers made so far, plain = large
 * requests that fell back to a plain allocation, classes = memory classes found, granules_per_class[4] = granules of the
 * live buffers per class.  RG_BALANCED_ALLOC=0 in the environment turns the balancing off. */
rg_status rg_mem_stats(int device, uint64_t *buffers, uint64_t *plain, uint32_t *classes, uint64_t *granules_per_class);
/* More of the same (round 5), vals[0 .. nvals): balanced buffers handed out, plain fallbacks, classes found, probe launches, wall time
 * spent classifying granules (microseconds), bytes of virtual address space reserved so far (round 6: one ARENA per process, 4 TiB of
 * addresses and no memory, from which every mapping's address is carved once -- the runtime recycles the addresses of freed buffers
 * into new reservations, and first touches of mappings made there faulted: csrc/rg_mem.hip), requests served from the cache of freed buffers, bytes cached, bytes live,
 * classes a buffer is spread over.  A freed balanced buffer stays mapped and cached (RG_MEM_CACHE_GIB, default 64 GiB per device) and
 * serves the next request of its size -- an index opened again reserves no new address space and runs no probe;
 * rg_mem_release hands the cache (and the pool's spare granules) back to the device. */
rg_status rg_mem_stats_ex(int device, uint64_t *vals, int nvals);
rg_status rg_mem_release(int device);
/* Post-mortem of a GPU memory fault (round 6; no counterpart in the reference).  The runtime answers a GPU page fault with one line
 * on stderr -- "Memory access fault by GPU ... on address 0x..." -- and abort().  rg_mem_fault_report(path) (or RG_FAULT_REPORT=path in
 * the environment at load time) installs SIGABRT / SIGSEGV / SIGBUS handlers that write, before the process dies, the library's journal
 * of address-space events (every range mapped, cached, handed out again, unmapped: the last 4096), its live and cached buffers and
 * /proc/self/maps to `path`, then pass the signal on to the handler that was installed before; benchlib/fault.py names
 * the buffer a fault address belongs to from that file.  rg_mem_journal_dump writes the same report now (tests). */
rg_status rg_mem_fault_report(const char *path);
rg_status rg_mem_journal_dump(const char *path);
/* Diagnostics (round 6): the allocator's walk in a loop -- per_round granules created, mapped at fresh addresses, zeroed, probed, dropped,
 * `rounds` times; the step both GPU faults on record sit in (scripts/r06/walk_stress.py).  *granules = how many were mapped and touched. */
rg_status rg_mem_walk_stress(int device, uint32_t per_round, uint32_t rounds, uint64_t *granules);
/* The device adjacency of an index as the search kernel reads it (diagnostics, tests; no counterpart in the reference):
 * [npts][*stride] words, word 0 of a row = its degree, then the neighbours: id in the low 24 bits and -- on indexes of up to 2^24
 * nodes -- min(15, in-degree of the neighbour) in bits 24..27 and its hub level in bits 28..31 (knob "hub_bits": at 2^m bits the
 * hub bitmap of a launch gives position p to the node of highest in-degree among those with (id * 0x9E3779B1) >> (32 - m) == p;
 * level = the smallest such m, 8 .. 22, minus 8; 15 = never).  *nwords = npts * stride; host_out = NULL only reports the sizes.
 * RG_ERR_ARG when the index keeps a CSR adjacency instead. */
rg_status rg_index_debug_ell(const rg_index *idx, uint32_t *host_out, uint64_t *nwords, uint32_t *stride);

/* ----------------------------------------------------------------- operator
 * Replaces: float Distance::compare(const float *a, const float *b, unsigned length) (distance.h:18;
 * DistanceInnerProduct distance.h:108-225, DistanceL2 distance.h:39-89), batched:
 *   out[i] == compare(base + ids[i]*stride, query, stride)   bit for bit. */
rg_status rg_score_batch(rg_index *idx, const float *query, const uint32_t *ids, uint32_t n, float *out);
rg_status rg_score_batch_dev(rg_index *idx, const float *d_query, const uint32_t *d_ids, uint32_t n, float *d_out,
                             void *stream);

/* ------------------------------------------------------------------- search
 * Replaces: the OpenMP loop over IndexBipartite::SearchRoarGraph(query, k, qid, params{L_pq}, indices, res_dists)
 * (tests/test_search_roargraph.cpp:203-209; index_bipartite.h:100-101, .cpp:2311-2420).
 *   queries  nq rows, row stride qstride floats (>= the index stride's dim; only dim values are read)
 *   out_ids  nq*k, out_dists nq*k (first k queue entries, (distance,id) order), out_cmps/out_hops nq
 *            (the pair SearchRoarGraph returns; either may be NULL)
 * Errors: RG_ERR_ARG if k > L_pq (the CLI's check, test_search_roargraph.cpp:192-195);
 *         RG_ERR_NOT_ENOUGH "not enough results: N, expected: K" for the first failing query. */
rg_status rg_search(rg_index *idx, const float *queries, uint32_t nq, uint32_t qstride, uint32_t k, uint32_t L_pq,
                    uint32_t *out_ids, float *out_dists, uint32_t *out_cmps, uint32_t *out_hops);
/* Query-sharded form over index REPLICAS, one per device (the index is replicated -- 10 GB for t2i-10M --
 * and the queries are split, no data-path collective).  Replica r searches the r-th contiguous slice of the batch on
 * its own device and stream, all slices run concurrently, results land in the caller's host arrays in query order and
 * are identical to rg_search on a single replica.  The one-process-per-GPU form of the same thing is
 * roargraph_amd/dist.py (torch.distributed); this is the single-process form a C++ host links. */
rg_status rg_search_sharded(rg_index *const *replicas, int nreplicas, const float *queries, uint32_t nq, uint32_t qstride,
                            uint32_t k, uint32_t L_pq, uint32_t *out_ids, float *out_dists, uint32_t *out_cmps,
                            uint32_t *out_hops);
/* device-resident form; enqueues on `stream` and returns without synchronising.
 * rg_search_wait() synchronises that stream and reports the deferred RG_ERR_NOT_ENOUGH, if any. */
rg_status rg_search_dev(rg_index *idx, const float *d_queries, uint32_t nq, uint32_t qstride, uint32_t k,
                        uint32_t L_pq, uint32_t *d_ids, float *d_dists, uint32_t *d_cmps, uint32_t *d_hops,
                        void *stream);
rg_status rg_search_wait(rg_index *idx, void *stream);
/* Optional: allocate now what batches of up to nq queries at beam widths up to L_pq will need on `stream` (the id logs of
 * the default visited mode; the visited tags of the exact form where a launch will use them: visited = 0, or a beam width
 * from which the adaptive default already chose the exact set), so that the first search does not pay for it.  Plays
 * the part of InitVisitedListPool(num_threads) (index_bipartite.h:133; tests/test_search_roargraph.cpp:173), which the
 * reference calls before its timed loop.  Never required: searches allocate on first use. */
rg_status rg_search_prepare(rg_index *idx, void *stream, uint32_t nq, uint32_t L_pq);
/* Measurement aid (bench.py's roofline block; no counterpart in the reference): over the id logs the last batch on
 * `stream` left behind -- it must have run in the default visited mode in one piece and have been waited for --
 * the number of distance evaluations the launch performed (re-scored rows included: each is a row read) and the number
 * of DISTINCT base rows among them.  distinct / evaluations is the share of a launch's row reads that are first touches,
 * i.e. that no cache can have served from an earlier read of the same launch.  d_row_counts (optional, device; the
 * caller provides nd zero-initialised counters, one per base row -- the call cannot check the size) receives how often each
 * row was read: the popularity distribution.  The kernels run on `stream`; the call returns when they have finished. */
rg_status rg_search_reuse_stats(rg_index *idx, void *stream, uint64_t *evaluations, uint64_t *distinct_rows,
                                uint32_t *d_row_counts);

/* ------------------------------------------------------------- ground truth
 * Replaces: the external `compute_groundtruth --data_type float --dist_fn {l2,mips,cosine} --base_file F
 * --query_file F --gt_file F --K n` (README.md:62-75; thirdparty/DiskANN is an empty submodule in the
 * reference tree).  Output rows are sorted best first; dists are +inner product for mips
 * (tests/test_search_bipartite.cpp:46-48) and squared L2 for l2.
 *
 * rg_gt_shard_dev: one GPU's part. Scores all nq queries against base rows [0, nb) of this shard and
 * writes the shard-local top-K (ids offset by id_base) sorted best first.  K <= 896.
 * rg_gt_merge_dev: merges nlists sorted K-lists per query (layout [list][nq][K]) into one; nlists*K <= 1024. */
rg_status rg_gt_shard_dev(const float *d_base, uint32_t nb, uint32_t bstride, const float *d_queries, uint32_t nq,
                          uint32_t qstride, uint32_t dim, int metric, uint32_t K, uint32_t id_base, uint32_t *d_ids,
                          float *d_dists, int device, void *stream);
rg_status rg_gt_merge_dev(const uint32_t *d_ids_in, const float *d_dists_in, uint32_t nlists, uint32_t nq, uint32_t K,
                          int metric, uint32_t *d_ids, float *d_dists, int device, void *stream);
/* Multi-rank form (one rank per GPU; BASELINE configs[2]: base sharded over 8 GPUs + exchange of the per-shard top-K over
 * xGMI).  A rank owns a contiguous row shard of the base, resident in its HBM; all ranks pass the same host query matrix.
 * Queries are streamed in batches of `batch` (0 = 65,536): K2 over the shard, all-to-all of the per-shard K-lists on a
 * second stream (RCCL ncclSend/ncclRecv, grouped) so that each rank receives the lists of the 1/world of the batch it
 * owns, K3 merge, rows downloaded -- overlapped with the next batch's K2.  Host memory is O(batch).  Rank r writes the
 * rows it owns (the r-th balanced contiguous slice of every batch) into out_ids / out_dists; the other rows are left
 * alone, so ranks that are threads of one process can share the output arrays.
 * rg_comm: RCCL communicator (librccl is dlopen'ed on first use) --
 *   rg_comm_unique_id + rg_comm_init_rank  one process per GPU (the id travels by whatever the launcher offers:
 *                                          torch.distributed broadcast, MPI, a file)
 *   rg_comm_init_local                     all ranks in this process, one per entry of `devices` (out: nranks handles);
 *                                          ranks that share a device, or a process without RCCL, move the lists with
 *                                          peer-to-peer copies instead (rg_comm_uses_rccl tells which) */
typedef struct rg_comm rg_comm;
rg_status rg_comm_unique_id(void *id128);
rg_status rg_comm_init_rank(const void *id128, int rank, int world, int device, rg_comm **out);
rg_status rg_comm_init_local(const int *devices, int nranks, rg_comm **out);
int rg_comm_uses_rccl(const rg_comm *comm);
void rg_comm_destroy(rg_comm *comm);
/* The exchange schedule of rg_groundtruth_rank as data (no GPU needed; the RCCL calls are issued from this very table):
 * for rank `rank` of `world`, `nq` queries in batches of `batch` (0 = 65,536), ten words per (batch, peer):
 *   batch, parity of its double buffer, q0, queries of the batch, peer, send_row0, send_rows (rows of the batch whose K-lists go
 *   to the peer: elements [send_row0 * K, (send_row0 + send_rows) * K) of this rank's lists), recv_slot_row0, recv_rows (where the
 *   peer's lists of the rows this rank owns arrive in its receive buffer, in rows), first global query row this rank owns.
 * out = NULL only reports *n_words.  tests/test_dist_gloo.py plays the table for eight ranks against a pure-Python model. */
rg_status rg_gt_exchange_plan(int world, int rank, uint32_t nq, uint32_t batch, uint32_t *out, uint64_t cap_words, uint64_t *n_words);
rg_status rg_groundtruth_rank(rg_comm *comm, const float *d_base_shard, uint32_t nb_shard, uint32_t bstride, uint32_t id_base,
                              const float *queries, uint32_t nq, uint32_t qstride, uint32_t dim, int metric, uint32_t K,
                              uint32_t batch, uint32_t *out_ids, float *out_dists);
/* host-memory convenience: whole job on `ndev` GPUs of this process (base sharded by rows, one thread per GPU running the
 * multi-rank form above; shards are uploaded through pinned chunks, no full-size staging copy) */
rg_status rg_groundtruth_mem(const float *base, uint32_t nb, uint32_t bstride, const float *queries, uint32_t nq,
                             uint32_t qstride, uint32_t dim, int metric, uint32_t K, uint32_t *out_ids,
                             float *out_dists, const int *devices, int ndev);
/* file form, the CLI twin's body: shards and query batches are read straight from the files and result rows written
 * straight into the gt file -- host memory O(batch) for a 10M x 10M job */
rg_status rg_groundtruth(const char *base_fbin, const char *query_fbin, const char *gt_out, int metric, uint32_t K,
                         const int *devices, int ndev);

/* --------------------------------------------------------- graph construction
 * Replaces: IndexBipartite::BuildRoarGraph(n_sq, nullptr, n_bp, base, params{M_sq, M_pjbp, L_pjpq, num_threads})
 * after LoadLearnBaseKNN (tests/test_build_roargraph.cpp:117-136; src/index_bipartite.cpp:143-233, 1043-1277).
 * CPU code, as in the reference.  knn_ids = the train-query ground truth ids (nq x knn_k,
 * best first); the result is the projection graph in CSR form (release with rg_free) and its entry point.
 * One thread gives the reference's T=1 construction.  More threads (round 3) give ONE result for any thread count -- the
 * reference's own multi-threaded build depends on scheduling: phases 1 and 2 come out exactly as at one thread (every
 * list replays its own reverse edges in the one-thread order), phase 3 runs in the batches of rg_build_schedule(nb, 0):
 * the nodes of a batch search the graph as it stood when the batch began and are linked in node order. */
rg_status rg_build_roargraph(const float *base, uint32_t nb, uint32_t dim, uint32_t stride, const uint32_t *knn_ids,
                             uint32_t nq, uint32_t knn_k, int metric, uint32_t M_sq, uint32_t M_pjbp, uint32_t L_pjpq,
                             uint32_t num_threads, uint32_t *out_ep, uint64_t **out_offsets, uint32_t **out_nbrs);
/* CalculateProjectionep (src/index_bipartite.cpp:2004-2041): the entry point, i.e. the base row nearest (squared L2) to the
 * centroid, with the reference's arithmetic -- float sums over the rows in index order per dimension, j-order distance
 * sums, the first of equal distances.  The device form gives the same bits (it keeps the serial order per dimension and
 * gets its speed from owning 32 dimensions per workgroup); rg_build_roargraph_gpu uses it. */
rg_status rg_projection_ep(const float *base, uint32_t nd, uint32_t dim, uint32_t stride, uint32_t *out_ep);
rg_status rg_projection_ep_dev(const float *d_base, uint32_t nd, uint32_t dim, uint32_t stride, int device, uint32_t *out_ep);
/* Same construction with phase 3 -- the n beam searches of the connectivity enhancement (index_bipartite.cpp:1192-1220,
 * 1279-1350), 85-93 % of the build time -- on the GPU (K1 in build mode), `batch` nodes at a time (0 = auto), together with
 * the occlusion pruning of phases 1 and 3; reverse-edge insertion stays on the host threads.  Nodes of a batch search the
 * graph as it stood when the batch started and are linked in node order: the result is the same for any num_threads, equals
 * rg_build_roargraph's at more than one thread when batch = 0, and equals the one-thread construction when batch = 1
 * (tests/test_gpu_cli.py, tests/test_build.py compare all of them with the oracle's restatement byte for byte).
 * Needs dim % 8 == 0 and stride % 4 == 0. */
rg_status rg_build_roargraph_gpu(const float *base, uint32_t nb, uint32_t dim, uint32_t stride, const uint32_t *knn_ids,
                                 uint32_t nq, uint32_t knn_k, int metric, uint32_t M_sq, uint32_t M_pjbp,
                                 uint32_t L_pjpq, uint32_t num_threads, int device, uint32_t batch, uint32_t *out_ep,
                                 uint64_t **out_offsets, uint32_t **out_nbrs);
/* the batches phase 3 runs in for `nb` nodes and this `batch` argument (0 = the builders' own choice: the first 2,048
 * nodes one by one, then batches of at most a quarter of what is linked): writes up to `cap` batch sizes, *count = how
 * many there are.  For checkers that restate the construction (oracle/rg_oracle_build.c takes the same list). */
rg_status rg_build_schedule(uint32_t nb, uint32_t batch, uint32_t *sizes, uint32_t cap, uint32_t *count);
/* Tests (no counterpart in the reference): ONE call of one occlusion-pruning rule of the construction, so that the rules can be held
 * against goldens made with the reference's own Distance / Neighbor objects (tests/golden/prune_*.npz, oracle/ref_driver.cpp `prune`).
 * kind 0 = PruneBiSearchBaseGetBase (index_bipartite.cpp:1612-1694), 1 = PruneProjectionReverseCandidates (:1526-1610), 2 =
 * PruneProjectionInternalReverseCandidates (:1434-1524), 3 = PruneProjectionBaseSearchCandidates (:1846-1940, have = the pivot's
 * projection list).  ids / dists [np] = the candidate pool (kinds 1, 2: the list, dists unused); out has room for max(M, np) ids.
 * use_gpu 0 = the builder's host routine, 1 = the pruning kernel of the GPU-assisted build (kinds 0 and 3). */
rg_status rg_build_prune_debug(const float *base, uint32_t nb, uint32_t dim, uint32_t stride, int metric, uint32_t M, int kind, uint32_t pivot,
                               const uint32_t *ids, const float *dists, uint32_t np, const uint32_t *have, uint32_t nhave, uint32_t *out,
                               uint32_t *nout, int use_gpu, int device);

#ifdef __cplusplus
}
#endif
#endif
