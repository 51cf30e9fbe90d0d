// ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.  Builds to oracle/_ref/rg_ref.
//
// A command-line driver around the reference's OWN header-only code, included
// from where it lies under /root/reference (never copied into this repo):
//   efanna2e/distance.h       DistanceInnerProduct / DistanceL2      (a1, a2)
//   efanna2e/neighbor.h       Neighbor, NeighborPriorityQueue        (a5)
//   visited_list_pool.h       VisitedList, VisitedListPool           (a6)
//   efanna2e/util.h           load_meta, load_data, data_align, load_gt_*  (a7)
// compiled with the reference's Release flags (CMakeLists.txt:24,28).
//
// What is NOT the reference here: src/index_bipartite.cpp cannot be compiled
// in this image (it includes boost/dynamic_bitset.hpp, boost/container/set.hpp
// and tsl/robin_set.h, none of which exist here, and stand-ins are not
// allowed).  So the per-query loop of SearchRoarGraph (:2311-2420) and the
// 15-line .index reader (:2097-2117) are restated below, driving the genuine
// queue / visited-pool / distance objects.  Everything arithmetic or
// order-sensitive on the path therefore runs the reference's own code.
//
// Sub-commands write raw little-endian binaries that scripts/make_golden.py
// turns into tests/golden fixtures and tests/test_oracle_vs_ref.py compares
// with oracle/rg_oracle.c.
#include <omp.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "efanna2e/distance.h"
#include "efanna2e/neighbor.h"
#include "efanna2e/util.h"
#include "visited_list_pool.h"

using efanna2e::Neighbor;
using efanna2e::NeighborPriorityQueue;

static std::vector<char> slurp(const char *p) {
    std::ifstream in(p, std::ios::binary);
    if (!in) throw std::runtime_error(std::string("cannot open ") + p);
    return std::vector<char>((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
}
static void spit(const char *p, const void *d, size_t n) {
    std::ofstream out(p, std::ios::binary);
    out.write((const char *)d, n);
}

// metric -> kernel, as src/index.cpp:8-26 chooses it
static efanna2e::Distance *make_distance(const std::string &m) {
    if (m == "l2") return new efanna2e::DistanceL2();
    return new efanna2e::DistanceInnerProduct();
}

// dist <l2|ip> in out : in = u32 n, u32 d, a[n][d], b[n][d]; out = f32[n]
static int cmd_dist(int argc, char **argv) {
    if (argc < 5) return 2;
    auto buf = slurp(argv[3]);
    uint32_t n, d;
    memcpy(&n, buf.data(), 4);
    memcpy(&d, buf.data() + 4, 4);
    const float *a = (const float *)(buf.data() + 8), *b = a + (size_t)n * d;
    std::unique_ptr<efanna2e::Distance> dist(make_distance(argv[2]));
    std::vector<float> out(n);
    for (uint32_t i = 0; i < n; ++i) out[i] = dist->compare(a + (size_t)i * d, b + (size_t)i * d, d);
    spit(argv[4], out.data(), out.size() * 4);
    return 0;
}

// queue in out : in = u32 cap, u32 nops, u8 op[nops], u32 id[nops], f32 dist[nops] (op 0 = insert, 1 = pop)
// out = u32 size, u32 cur, u32 npop, ids[size], dists[size], u8 flags[size], u32 pop_ids[npop]
static int cmd_queue(int argc, char **argv) {
    if (argc < 4) return 2;
    auto buf = slurp(argv[2]);
    uint32_t cap, nops;
    memcpy(&cap, buf.data(), 4);
    memcpy(&nops, buf.data() + 4, 4);
    const uint8_t *op = (const uint8_t *)buf.data() + 8;
    std::vector<uint32_t> ids(nops);
    std::vector<float> ds(nops);
    memcpy(ids.data(), op + nops, (size_t)nops * 4);
    memcpy(ds.data(), op + nops + (size_t)nops * 4, (size_t)nops * 4);
    NeighborPriorityQueue q(cap);
    std::vector<uint32_t> pops;
    for (uint32_t i = 0; i < nops; ++i) {
        if (op[i] == 0) q.insert(Neighbor(ids[i], ds[i], false));
        else if (q.has_unexpanded_node()) pops.push_back(q.closest_unexpanded().id);
    }
    // _cur is private: it equals the index of the first entry whose flag is clear
    uint32_t size = (uint32_t)q.size(), cur = size, npop = (uint32_t)pops.size();
    for (uint32_t i = 0; i < size; ++i) if (!q[i].flag) { cur = i; break; }
    std::ofstream out(argv[3], std::ios::binary);
    out.write((char *)&size, 4); out.write((char *)&cur, 4); out.write((char *)&npop, 4);
    for (uint32_t i = 0; i < size; ++i) { uint32_t v = q[i].id; out.write((char *)&v, 4); }
    for (uint32_t i = 0; i < size; ++i) { float v = q[i].distance; out.write((char *)&v, 4); }
    for (uint32_t i = 0; i < size; ++i) { uint8_t v = q[i].flag; out.write((char *)&v, 1); }
    out.write((char *)pops.data(), (size_t)npop * 4);
    return 0;
}

struct Graph {
    uint32_t ep = 0;
    std::vector<std::vector<uint32_t>> adj;
};
static Graph read_index(const char *path) {  // layout of index_bipartite.cpp:2097-2117
    std::ifstream in(path, std::ios::binary);
    if (!in) throw std::runtime_error("cannot open index");
    Graph g;
    uint32_t n = 0;
    in.read((char *)&g.ep, 4);
    in.read((char *)&n, 4);
    g.adj.resize(n);
    for (auto &row : g.adj) {
        uint32_t deg = 0;
        in.read((char *)&deg, 4);
        row.resize(deg);
        in.read((char *)row.data(), (std::streamsize)deg * 4);
    }
    return g;
}

// search base.fbin graph.index query.fbin <l2|ip|cosine> k L T out.bin [repeat] [prefetch]
// out = u32 nq, u32 k, ids[nq][k], dists[nq][k], cmps[nq], hops[nq]; prints "QPS <v> threads <T> ms <ms> prefetch <0|1>"
// prefetch (default 1): the loop issues the software prefetches SearchRoarGraph issues -- prefetch_vector of the entry
// point's row (:2324, util.h:77-80, with the reference's own byte count: `dimension_` BYTES of the row) and, on every
// neighbour iteration, _MM_HINT_T0 of the NEXT neighbour's visited tag and of the first line of its base row
// (:2374-2375).  The reference reads cur_nbrs[j + 1] one past the end of the list on the last iteration; here the last
// iteration prefetches nothing (a prefetch changes no result, only the time).  prefetch = 0 is the loop without them
// (what the round-2 baseline timed): both rates are reported by bench.py.
static int cmd_search(int argc, char **argv) {
    if (argc < 10) return 2;
    const char *base_f = argv[2], *index_f = argv[3], *query_f = argv[4];
    std::string metric = argv[5];
    uint32_t k = (uint32_t)atoi(argv[6]), L = (uint32_t)atoi(argv[7]);
    int T = atoi(argv[8]);
    int repeat = argc > 10 ? atoi(argv[10]) : 1;
    const bool prefetch = argc > 11 ? atoi(argv[11]) != 0 : true;
    // the loader sequence of tests/test_search_roargraph.cpp:119-132 and LoadVectorData (:2664-2695)
    uint32_t nb, bd, nq, qd;
    efanna2e::load_meta<float>(base_f, nb, bd);
    float *base = nullptr;
    efanna2e::load_data<float>(base_f, nb, bd, base);
    if (metric == "cosine") for (size_t i = 0; i < nb; ++i) efanna2e::normalize<float>(base + i * (uint64_t)bd, (uint64_t)bd);
    base = efanna2e::data_align(base, nb, bd);
    efanna2e::load_meta<float>(query_f, nq, qd);
    float *query = nullptr;
    efanna2e::load_data<float>(query_f, nq, qd, query);
    query = efanna2e::data_align(query, nq, qd);
    if (metric == "cosine") for (uint32_t i = 0; i < nq; ++i) efanna2e::normalize<float>(query + (size_t)i * qd, qd);
    Graph g = read_index(index_f);
    std::unique_ptr<efanna2e::Distance> dist(make_distance(metric == "l2" ? "l2" : "ip"));
    VisitedListPool pool(T, (int)nb);  // InitVisitedListPool, index_bipartite.h:133
    size_t dim = qd;

    std::vector<uint32_t> ids((size_t)nq * k), cmps(nq), hops(nq);
    std::vector<float> dists((size_t)nq * k);
    std::string err;
    omp_set_num_threads(T);
    double best_ms = 1e30;
    for (int rep = 0; rep < repeat; ++rep) {
        auto t0 = std::chrono::high_resolution_clock::now();
#pragma omp parallel for schedule(dynamic, 1)
        for (uint32_t qi = 0; qi < nq; ++qi) {
            const float *q = query + (size_t)qi * dim;
            NeighborPriorityQueue beam(L);
            VisitedList *vl = pool.getFreeVisitedList();
            vl_type *seen = vl->mass;
            const vl_type stamp = vl->curV;
            // entry point goes into the beam unmarked
            if (prefetch) efanna2e::prefetch_vector((const char *)(base + (size_t)g.ep * dim), dim);   // :2324
            beam.insert(Neighbor(g.ep, dist->compare(base + (size_t)g.ep * dim, q, (unsigned)dim), false));
            uint32_t ncmp = 0, nhop = 0;
            while (beam.has_unexpanded_node()) {
                const unsigned node = beam.closest_unexpanded().id;
                const uint32_t *nbrs = g.adj[node].data();
                const size_t deg = g.adj[node].size();
                ++nhop;
                for (size_t j = 0; j < deg; ++j) {
                    const uint32_t nb_id = nbrs[j];
                    if (prefetch && j + 1 < deg) {                                                      // :2374-2375
                        _mm_prefetch((const char *)(seen + nbrs[j + 1]), _MM_HINT_T0);
                        _mm_prefetch((const char *)(base + (size_t)nbrs[j + 1] * dim), _MM_HINT_T0);
                    }
                    if (seen[nb_id] == stamp) continue;
                    seen[nb_id] = stamp;
                    const float dd = dist->compare(base + (size_t)nb_id * dim, q, (unsigned)dim);
                    ++ncmp;
                    beam.insert(Neighbor(nb_id, dd, false));
                }
            }
            pool.releaseVisitedList(vl);
            cmps[qi] = ncmp;
            hops[qi] = nhop;
            if (beam.size() < k) {
#pragma omp critical
                {
                    std::stringstream ss;
                    ss << "not enough results: " << beam.size() << ", expected: " << k;
                    err = ss.str();
                }
                continue;
            }
            for (uint32_t i = 0; i < k; ++i) {
                ids[(size_t)qi * k + i] = beam[i].id;
                dists[(size_t)qi * k + i] = beam[i].distance;
            }
        }
        auto t1 = std::chrono::high_resolution_clock::now();
        double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        if (ms < best_ms) best_ms = ms;
    }
    if (!err.empty()) { std::cerr << "EXC: " << err << std::endl; return 3; }
    std::ofstream out(argv[9], std::ios::binary);
    out.write((char *)&nq, 4); out.write((char *)&k, 4);
    out.write((char *)ids.data(), ids.size() * 4);
    out.write((char *)dists.data(), dists.size() * 4);
    out.write((char *)cmps.data(), cmps.size() * 4);
    out.write((char *)hops.data(), hops.size() * 4);
    std::cout << "QPS " << (nq / (best_ms / 1000.0)) << " threads " << T << " ms " << best_ms << " prefetch " << (prefetch ? 1 : 0) << std::endl;
    return 0;
}

// meta <fbin|gt> file : prints "OK npts dim" or "EXC: <what>"
static int cmd_meta(int argc, char **argv) {
    if (argc < 4) return 2;
    uint32_t n = 0, d = 0;
    if (std::string(argv[2]) == "gt") efanna2e::load_gt_meta<uint32_t>(argv[3], n, d);
    else efanna2e::load_meta<float>(argv[3], n, d);
    std::cout << "OK " << n << " " << d << std::endl;
    return 0;
}

// gtload file out : genuine load_gt_meta + load_gt_data_with_dist; out = ids then dists
static int cmd_gtload(int argc, char **argv) {
    if (argc < 4) return 2;
    uint32_t n = 0, k = 0;
    uint32_t *ids = nullptr;
    float *ds = nullptr;
    efanna2e::load_gt_meta<uint32_t>(argv[2], n, k);
    efanna2e::load_gt_data_with_dist<uint32_t, float>(argv[2], n, k, ids, ds);
    std::ofstream out(argv[3], std::ios::binary);
    out.write((char *)ids, (size_t)n * k * 4);
    out.write((char *)ds, (size_t)n * k * 4);
    std::cout << "OK " << n << " " << k << std::endl;
    return 0;
}

// fbinload file out : genuine load_meta + load_data + data_align; out = u32 n, u32 aligned_dim, rows
static int cmd_fbinload(int argc, char **argv) {
    if (argc < 4) return 2;
    uint32_t n = 0, d = 0;
    float *data = nullptr;
    efanna2e::load_meta<float>(argv[2], n, d);
    efanna2e::load_data<float>(argv[2], n, d, data);
    data = efanna2e::data_align(data, n, d);
    std::ofstream out(argv[3], std::ios::binary);
    out.write((char *)&n, 4); out.write((char *)&d, 4);
    out.write((char *)data, (size_t)n * d * 4);
    std::cout << "OK " << n << " " << d << std::endl;
    return 0;
}

// ep base.fbin : the loops of IndexBipartite::CalculateProjectionep (src/index_bipartite.cpp:2004-2041) -- the TU itself
// cannot be compiled here (see the header), so its four plain loops are restated and compiled with the reference's flags
// (-Ofast: the compiler is free to vectorise the j loop, as it is in the reference's build); prints "EP <row>"
static int cmd_ep(int argc, char **argv) {
    if (argc < 3) return 2;
    uint32_t n = 0, d = 0;
    float *data = nullptr;
    efanna2e::load_meta<float>(argv[2], n, d);
    efanna2e::load_data<float>(argv[2], n, d, data);
    data = efanna2e::data_align(data, n, d);
    const size_t nd_ = n, dimension_ = d;
    float *center = new float[dimension_]();
    for (size_t i = 0; i < nd_; ++i)
        for (size_t dd = 0; dd < dimension_; ++dd) center[dd] += data[i * dimension_ + dd];
    for (size_t dd = 0; dd < dimension_; ++dd) center[dd] /= (float)nd_;
    float *distances = new float[nd_]();
#pragma omp parallel for
    for (size_t i = 0; i < nd_; ++i) {
        const float *cur_data = data + i * dimension_;
        float diff = 0;
        for (size_t j = 0; j < dimension_; ++j) diff += ((center[j] - cur_data[j]) * (center[j] - cur_data[j]));
        distances[i] = diff;
    }
    uint32_t closest = 0;
    for (size_t i = 1; i < nd_; ++i)
        if (distances[i] < distances[closest]) closest = static_cast<uint32_t>(i);
    std::cout << "EP " << closest << std::endl;
    delete[] center;
    delete[] distances;
    return 0;
}

// prune base.fbin <l2|ip> M calls.bin out.bin : the four occlusion-pruning rules of the construction, one call after the other.
// src/index_bipartite.cpp cannot be compiled here (see the header), so the rules are restated around the reference's genuine
// objects -- Distance::compare for every distance, Neighbor with ITS operator< under std::sort and ITS operator== under std::find:
//   kind 0  PruneBiSearchBaseGetBase                     (:1612-1694)  pool = (id, distance) pairs in search order, pivot = tgt_base
//   kind 1  PruneProjectionReverseCandidates             (:1526-1610)  pool = ids of the list (distances are computed), pivot = src_node
//   kind 2  PruneProjectionInternalReverseCandidates     (:1434-1524)  the same with the queue starting as list.size() value-initialised
//                                                                      Neighbors (:1438) and no top-up
//   kind 3  PruneProjectionBaseSearchCandidates          (:1846-1940)  pool = (id, distance) pairs, pivot = qid, have = projection_graph_[qid]
// calls.bin = u32 ncalls, then per call: u32 kind, pivot, np, nhave; u32 ids[np]; f32 dists[np]; u32 have[nhave]
// out.bin   = per call: u32 n; u32 ids[n]
namespace prune_rules {
struct Ctx {
    const efanna2e::Distance *dist;
    const float *data;
    size_t dim;
    uint32_t M;
    float d(uint32_t a, uint32_t b) const { return dist->compare(data + dim * a, data + dim * b, (unsigned)dim); }
};
// one occlusion sweep over q[start + 1 ..): a candidate joins unless it is in the result, or some member of the result is
// closer to it than it is to the pivot; `self` never joins; refuse_dup = the extra std::find of the second sweeps
static void sweep(const Ctx &c, std::vector<Neighbor> &q, uint32_t &start, uint32_t self, std::vector<uint32_t> &result, bool refuse_dup) {
    while (result.size() < c.M && (++start) < q.size()) {
        Neighbor &p = q[start];
        bool occlude = false;
        for (size_t t = 0; t < result.size() && !occlude; ++t) {
            if (p.id == result[t]) { occlude = true; break; }
            const float djk = c.d(p.id, result[t]);
            if (refuse_dup ? (1.0 * djk < p.distance) : (djk < p.distance)) occlude = true;
        }
        if (occlude || p.id == self) continue;
        if (refuse_dup && std::find(result.begin(), result.end(), p.id) != result.end()) continue;
        result.push_back(p.id);
    }
}
static std::vector<uint32_t> get_base(const Ctx &c, std::vector<Neighbor> &search_pool, uint32_t tgt) {
    std::vector<Neighbor> base_pool;
    std::vector<uint32_t> seen;
    for (auto &b : search_pool) {
        if (std::find(seen.begin(), seen.end(), b.id) != seen.end() || b.id == tgt) continue;
        base_pool.push_back(b);
        seen.push_back(b.id);
    }
    std::sort(base_pool.begin(), base_pool.end());
    std::vector<uint32_t> result;
    uint32_t start = 0;
    result.push_back(base_pool[start].id);
    sweep(c, base_pool, start, tgt, result, false);
    start = 0;                                   // the second sweep walks the search pool as it came, and skips members first
    while (result.size() < c.M && (++start) < search_pool.size()) {
        Neighbor &p = search_pool[start];
        if (std::find(result.begin(), result.end(), p.id) != result.end()) continue;
        bool occlude = false;
        for (size_t t = 0; t < result.size() && !occlude; ++t) {
            if (p.id == result[t]) { occlude = true; break; }
            if (1.0 * c.d(p.id, result[t]) < p.distance) occlude = true;
        }
        if (!occlude && p.id != tgt && std::find(result.begin(), result.end(), p.id) == result.end()) result.push_back(p.id);
    }
    for (size_t i = 1; i < base_pool.size() && result.size() < c.M; ++i)
        if (std::find(result.begin(), result.end(), base_pool[i].id) == result.end() && base_pool[i].id != tgt) result.push_back(base_pool[i].id);
    return result;
}
static std::vector<uint32_t> reverse(const Ctx &c, uint32_t src, const std::vector<uint32_t> &list, bool phantoms) {
    std::vector<Neighbor> q(phantoms ? list.size() : 0);          // (:1438: value-initialised entries, id 0 / distance 0)
    for (uint32_t id : list) {
        const Neighbor nn(id, c.d(src, id), false);
        if (std::find(q.begin(), q.end(), nn) == q.end()) q.push_back(nn);
    }
    std::sort(q.begin(), q.end());
    std::vector<uint32_t> result;
    uint32_t start = 0;
    if (q[start].id == src) ++start;
    result.push_back(q[start].id);
    sweep(c, q, start, src, result, false);
    start = 0;
    sweep(c, q, start, src, result, true);
    if (!phantoms)
        for (size_t i = 0; i < list.size() && result.size() < c.M; ++i)
            if (std::find(result.begin(), result.end(), list[i]) == result.end()) result.push_back(list[i]);
    return result;
}
static std::vector<uint32_t> search(const Ctx &c, std::vector<Neighbor> &pool, uint32_t qid, const std::vector<uint32_t> &have) {
    std::sort(pool.begin(), pool.end());
    std::vector<uint32_t> result;
    uint32_t start = 0;
    if (pool[start].id == qid) ++start;
    while (std::find(have.begin(), have.end(), pool[start].id) != have.end()) ++start;
    result.push_back(pool[start].id);
    sweep(c, pool, start, qid, result, false);
    start = 0;
    sweep(c, pool, start, qid, result, true);
    return result;
}
}  // namespace prune_rules

static int cmd_prune(int argc, char **argv) {
    if (argc < 7) return 2;
    uint32_t n = 0, d = 0;
    float *data = nullptr;
    efanna2e::load_meta<float>(argv[2], n, d);
    efanna2e::load_data<float>(argv[2], n, d, data);
    data = efanna2e::data_align(data, n, d);
    std::unique_ptr<efanna2e::Distance> dist(make_distance(argv[3]));
    prune_rules::Ctx c{dist.get(), data, d, (uint32_t)atoi(argv[4])};
    auto buf = slurp(argv[5]);
    const uint32_t *w = (const uint32_t *)buf.data();
    const uint32_t ncalls = *w++;
    std::ofstream out(argv[6], std::ios::binary);
    for (uint32_t i = 0; i < ncalls; ++i) {
        const uint32_t kind = w[0], pivot = w[1], np = w[2], nhave = w[3];
        const uint32_t *ids = w + 4;
        const float *ds = (const float *)(ids + np);
        const uint32_t *have = (const uint32_t *)(ds + np);
        w = have + nhave;
        std::vector<uint32_t> res;
        if (kind == 1 || kind == 2) res = prune_rules::reverse(c, pivot, std::vector<uint32_t>(ids, ids + np), kind == 2);
        else {
            std::vector<Neighbor> pool;
            for (uint32_t j = 0; j < np; ++j) pool.push_back(Neighbor(ids[j], ds[j], false));
            res = kind == 0 ? prune_rules::get_base(c, pool, pivot) : prune_rules::search(c, pool, pivot, std::vector<uint32_t>(have, have + nhave));
        }
        const uint32_t m = (uint32_t)res.size();
        out.write((const char *)&m, 4);
        out.write((const char *)res.data(), (std::streamsize)m * 4);
    }
    std::cout << "OK " << ncalls << std::endl;
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 2) { std::cerr << "usage: rg_ref <dist|queue|search|meta|gtload|fbinload|ep|prune> ..." << std::endl; return 2; }
    std::string c = argv[1];
    try {
        if (c == "dist") return cmd_dist(argc, argv);
        if (c == "queue") return cmd_queue(argc, argv);
        if (c == "search") return cmd_search(argc, argv);
        if (c == "meta") return cmd_meta(argc, argv);
        if (c == "gtload") return cmd_gtload(argc, argv);
        if (c == "fbinload") return cmd_fbinload(argc, argv);
        if (c == "ep") return cmd_ep(argc, argv);
        if (c == "prune") return cmd_prune(argc, argv);
    } catch (const std::exception &e) {
        std::cout << "EXC: " << e.what() << std::endl;
        return 3;
    }
    return 2;
}
