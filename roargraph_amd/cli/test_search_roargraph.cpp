// test_search_roargraph -- drop-in twin of the reference's search driver (tests/test_search_roargraph.cpp:64-250)
// running the search on an MI355X through the C ABI (include/rg.h).  Same flags (including the required-but-ignored
// --data_type), same stdout table and CSV row: L_pq, QPS, avg_visited, mean_latency, recall@k, avg_hops.
//
// Differences, all additive:
//   --device N          GPU to use (default 0); -T/--num_threads is accepted and ignored (the OpenMP loop of
//                       test_search_roargraph.cpp:203-209 runs inside the kernel, one wave per in-flight query)
//   timing              QPS uses a nanosecond clock around the device-resident batch (queries already in HBM); the
//                       reference's integer-millisecond clock (:210-213) cannot resolve a 10k-query batch on a GPU
//   warm-up             min(100, q_pts) queries, as :198-200
//   --devices 0 1 ...   one index replica per listed GPU, the query batch split between them (host buffers in the timed
//                       region: rg_search_sharded); --device is ignored when this is given
//   --fast_bf16 1       opt-in non-parity mode of the library (default 0 = the reference's results bit for bit)
//   --steady 1          a seventh stdout column QPS_steady: a later pass over the same queries, after the device path has
//                       settled on one of its two exact visited forms for this beam width (default 0 = the reference's
//                       six columns, test_search_roargraph.cpp:190, and one timed pass per L_pq)
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "argparse_lite.h"
#include "rg.h"

#define CK(x)                                                                  \
    do {                                                                       \
        rg_status s_ = (x);                                                    \
        if (s_ != RG_OK) { std::cerr << rg_last_error() << std::endl; return -1; } \
    } while (0)
#define HK(x)                                                                                        \
    do {                                                                                             \
        hipError_t e_ = (x);                                                                         \
        if (e_ != hipSuccess) { std::cerr << #x << ": " << hipGetErrorString(e_) << std::endl; return -1; } \
    } while (0)

int main(int argc, char **argv) {
    Args a;
    a.add("data_type", true, "data type <int8/uint8/float>");
    a.add("dist", true, "distance function <l2/ip>");
    a.add("base_data_path", true, "Input data file in bin format");
    a.add("query_path", true, "Query file in bin format");
    a.add("gt_path", true, "Groundtruth file in bin format");
    a.add("projection_index_save_path", true, "Path prefix for saving projetion index file components");
    a.add("L_pq", true, "Priority queue length for searching");
    a.add("k", false, "k nearest neighbors", "1");
    a.add("evaluation_save_path", false, "Path prefix for saving evaluation results", "");
    a.add("num_threads", false, "accepted for compatibility (ignored on the GPU path)", "0", "T");
    a.add("device", false, "HIP device index", "0");
    a.add("devices", false, "several HIP devices: one index replica each, queries sharded (rg_search_sharded)", "");
    a.add("fast_bf16", false, "1 = opt-in NON-parity mode: bf16 traversal + exact fp32 re-rank (rg.h)", "0");
    a.add("steady", false, "1 = extra stdout column QPS_steady (two more passes per L_pq); default: the reference's six columns", "0");
    if (!a.parse(argc, argv)) return -1;
    if (a.help()) { a.usage(std::cout); return 0; }

    const std::string dist = a.str("dist");
    int metric = RG_METRIC_IP;
    if (dist == "l2") { metric = RG_METRIC_L2; std::cout << "Using l2 as distance metric" << std::endl; }
    else if (dist == "ip") { metric = RG_METRIC_IP; std::cout << "Using inner product as distance metric" << std::endl; }
    else if (dist == "cosine") { metric = RG_METRIC_COSINE; std::cout << "Using cosine as distance metric" << std::endl; }
    else { std::cout << "Unknown distance type: " << dist << std::endl; return -1; }

    uint32_t base_num = 0, base_dim = 0;
    CK(rg_fbin_meta(a.str("base_data_path").c_str(), &base_num, &base_dim));
    uint32_t q_pts = 0, q_dim = 0, q_stride = 0;
    float *query = nullptr;
    CK(rg_fbin_load(a.str("query_path").c_str(), &q_pts, &q_dim, &q_stride, &query));
    uint32_t gt_pts = 0, gt_dim = 0, *gt_ids = nullptr;
    float *gt_dists = nullptr;
    CK(rg_gt_load(a.str("gt_path").c_str(), &gt_pts, &gt_dim, &gt_ids, &gt_dists));
    {
        std::ifstream probe(a.str("projection_index_save_path"));
        if (!probe.good()) { std::cout << "projection index file does not exist." << std::endl; return -1; }
    }
    std::vector<int> devices;
    for (const std::string &d : a.list("devices"))
        if (!d.empty()) devices.push_back((int)std::strtol(d.c_str(), nullptr, 10));
    const int device = devices.empty() ? (int)a.u("device") : devices[0];
    std::cout << "Load graph index: " << a.str("projection_index_save_path") << std::endl;
    // one replica per device (--devices) or the one index of --device: the files are read once (rg_index_open_multi)
    std::vector<int> devs = devices.empty() ? std::vector<int>{device} : devices;
    std::vector<rg_index *> replicas(devs.size(), nullptr);
    CK(rg_index_open_multi(a.str("base_data_path").c_str(), a.str("projection_index_save_path").c_str(), metric, devs.data(), (int)devs.size(),
                           replicas.data()));
    rg_index *index = replicas[0];
    if (replicas.size() > 1) std::cout << "Index replicated on " << replicas.size() << " devices, queries sharded" << std::endl;
    uint32_t nd, dim, stride, ep, maxdeg;
    float avgdeg;
    CK(rg_index_info(index, &nd, &dim, &stride, &ep, &avgdeg, &maxdeg, nullptr));
    if (a.u("fast_bf16")) {
        for (rg_index *rep : replicas) CK(rg_index_set(rep, "fast_bf16", 1));
        std::cout << "fast_bf16: traversal on a bf16 copy of the base, exact re-rank (results are NOT the reference's)" << std::endl;
    }
    std::cout << "Projection graph, ep: " << ep << std::endl;
    std::cout << "Projection graph, avg_degree: " << avgdeg << std::endl;
    if (q_stride != dim) { std::cerr << "base and query dimension mismatch" << std::endl; return -1; }
    if (metric == RG_METRIC_COSINE) {
        std::cout << "Normalizing query data" << std::endl;
        rg_normalize_rows(query, q_pts, q_stride, q_stride);
    }
    const uint32_t k = (uint32_t)a.u("k");
    std::cout << "k: " << k << std::endl;

    HK(hipSetDevice(device));
    float *d_q = nullptr, *d_dist = nullptr;
    uint32_t *d_ids = nullptr, *d_cmps = nullptr, *d_hops = nullptr;
    HK(hipMalloc(&d_q, (size_t)q_pts * q_stride * 4));
    HK(hipMalloc(&d_ids, (size_t)q_pts * k * 4));
    HK(hipMalloc(&d_dist, (size_t)q_pts * k * 4));
    HK(hipMalloc(&d_cmps, (size_t)q_pts * 4));
    HK(hipMalloc(&d_hops, (size_t)q_pts * 4));
    HK(hipMemcpy(d_q, query, (size_t)q_pts * q_stride * 4, hipMemcpyHostToDevice));
    std::vector<uint32_t> res((size_t)q_pts * k), cmps(q_pts), hops(q_pts);

    std::ofstream evaluation_out;
    if (!a.str("evaluation_save_path").empty()) evaluation_out.open(a.str("evaluation_save_path"), std::ios::out);
    std::cout << "Using thread: " << a.u("num_threads") << " (GPU path: one wave per in-flight query)" << std::endl;
    // InitVisitedListPool (test_search_roargraph.cpp:173): what the searches below will need is allocated before the loop,
    // as the reference allocates its visited lists before its own
    uint32_t L_max = k;
    for (const std::string &ls : a.list("L_pq")) L_max = std::max<uint32_t>(L_max, (uint32_t)std::strtoul(ls.c_str(), nullptr, 10));
    if (replicas.size() == 1) CK(rg_search_prepare(index, nullptr, q_pts, L_max));
    // Columns: the reference's six (:190; QPS = its protocol: 100 warm-up queries, then ONE timed pass over the query file,
    // :198-213).  --steady 1 adds QPS_steady = a later pass over the same queries, after the device path has settled on one
    // of its two exact visited forms for this beam width (same results either way; RG_TRACE_ADAPTIVE=1 shows the decisions)
    const bool steady = a.u("steady") != 0;
    std::cout << "L_pq" << "\t\tQPS" << "\t\t\tavg_visited" << "\tmean_latency" << "\trecall@" << k << "\tavg_hops";
    if (steady) std::cout << "\tQPS_steady";
    std::cout << std::endl;
    for (const std::string &ls : a.list("L_pq")) {
        const uint32_t L_pq = (uint32_t)std::strtoul(ls.c_str(), nullptr, 10);
        if (k > L_pq) { std::cout << "L_pq must greater or equal than k" << std::endl; return 1; }
        const uint32_t warm = q_pts < 100 ? q_pts : 100;
        double ms = 0.0, ms_steady = 0.0;
        if (replicas.size() > 1) {
            std::vector<float> dist_h((size_t)q_pts * k);
            CK(rg_search_sharded(replicas.data(), (int)replicas.size(), query, warm, q_stride, k, L_pq, res.data(), dist_h.data(),
                                 cmps.data(), hops.data()));
            auto t0 = std::chrono::high_resolution_clock::now();
            CK(rg_search_sharded(replicas.data(), (int)replicas.size(), query, q_pts, q_stride, k, L_pq, res.data(), dist_h.data(),
                                 cmps.data(), hops.data()));
            auto t1 = std::chrono::high_resolution_clock::now();
            ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
            if (steady) {
                CK(rg_search_sharded(replicas.data(), (int)replicas.size(), query, q_pts, q_stride, k, L_pq, res.data(), dist_h.data(),
                                     cmps.data(), hops.data()));
                auto t2 = std::chrono::high_resolution_clock::now();
                CK(rg_search_sharded(replicas.data(), (int)replicas.size(), query, q_pts, q_stride, k, L_pq, res.data(), dist_h.data(),
                                     cmps.data(), hops.data()));
                auto t3 = std::chrono::high_resolution_clock::now();
                ms_steady = std::chrono::duration<double, std::milli>(t3 - t2).count();
            }
        } else {
            CK(rg_search_dev(index, d_q, warm, q_stride, k, L_pq, d_ids, d_dist, d_cmps, d_hops, nullptr));   // :198-201
            CK(rg_search_wait(index, nullptr));
            auto t0 = std::chrono::high_resolution_clock::now();                                               // :203-210
            CK(rg_search_dev(index, d_q, q_pts, q_stride, k, L_pq, d_ids, d_dist, d_cmps, d_hops, nullptr));
            CK(rg_search_wait(index, nullptr));
            auto t1 = std::chrono::high_resolution_clock::now();
            ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
            HK(hipMemcpy(res.data(), d_ids, res.size() * 4, hipMemcpyDeviceToHost));
            HK(hipMemcpy(cmps.data(), d_cmps, cmps.size() * 4, hipMemcpyDeviceToHost));
            HK(hipMemcpy(hops.data(), d_hops, hops.size() * 4, hipMemcpyDeviceToHost));
            if (steady) {
                // steady state: one more untimed pass (where the adaptive default may try its other exact form), then a timed one
                CK(rg_search_dev(index, d_q, q_pts, q_stride, k, L_pq, d_ids, d_dist, d_cmps, d_hops, nullptr));
                CK(rg_search_wait(index, nullptr));
                auto t2 = std::chrono::high_resolution_clock::now();
                CK(rg_search_dev(index, d_q, q_pts, q_stride, k, L_pq, d_ids, d_dist, d_cmps, d_hops, nullptr));
                CK(rg_search_wait(index, nullptr));
                auto t3 = std::chrono::high_resolution_clock::now();
                ms_steady = std::chrono::duration<double, std::milli>(t3 - t2).count();
            }
        }
        const float qps = (float)q_pts / ((float)ms / 1000.0f);
        const float recall = rg_recall(q_pts, k, gt_dim, res.data(), gt_ids);
        float avg_cmps = 0.0f, avg_hops = 0.0f;
        for (uint32_t i = 0; i < q_pts; ++i) { avg_cmps += cmps[i]; avg_hops += hops[i]; }
        avg_cmps /= q_pts;
        avg_hops /= (float)q_pts;
        std::cout << L_pq << "\t\t" << qps << "\t\t" << avg_cmps << "\t\t" << ((float)ms / q_pts) << "\t\t" << recall
                  << "\t\t" << avg_hops;
        if (steady) std::cout << "\t\t" << ((float)q_pts / ((float)ms_steady / 1000.0f));
        std::cout << std::endl;
        if (evaluation_out.is_open())
            evaluation_out << L_pq << "," << qps << "," << avg_cmps << "," << ((float)ms / q_pts) << "," << recall << ","
                           << avg_hops << std::endl;
    }
    if (evaluation_out.is_open()) evaluation_out.close();
    for (rg_index *rep : replicas) rg_index_close(rep);
    rg_free(query); rg_free(gt_ids); rg_free(gt_dists);
    (void)hipFree(d_q); (void)hipFree(d_ids); (void)hipFree(d_dist); (void)hipFree(d_cmps); (void)hipFree(d_hops);
    return 0;
}
