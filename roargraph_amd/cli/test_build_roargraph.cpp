// test_build_roargraph -- drop-in twin of the reference's build driver (tests/test_build_roargraph.cpp:22-139):
// same flags, reads the base .fbin and the train-query ground truth (LoadLearnBaseKNN layout), writes the .index.
// Graph construction is CPU code in the reference and here (rg_build_roargraph); the GPU enters upstream
// (compute_groundtruth) and downstream (test_search_roargraph).
#include <chrono>
#include <iostream>
#include <string>
#include <thread>

#include "argparse_lite.h"
#include "rg.h"

#define CK(x)                                                                      \
    do {                                                                           \
        rg_status s_ = (x);                                                        \
        if (s_ != RG_OK) { std::cerr << rg_last_error() << std::endl; return -1; } \
    } while (0)

int main(int argc, char **argv) {
    Args a;
    const std::string ncpu = std::to_string(std::max(1u, std::thread::hardware_concurrency()));
    a.add("data_type", true, "data type <int8/uint8/float>");
    a.add("dist", true, "distance function <l2/ip>");
    a.add("base_data_path", true, "Input data file in bin format");
    a.add("sampled_query_data_path", true, "Sampled query file in bin format");
    a.add("projection_index_save_path", true, "Path prefix for saving projetion index file components");
    a.add("M_sq", false, "Number of neighbors for sampled query points to build the bipartite graph", "32");
    a.add("M_pjbp", false, "Number of neighbors for projection graph", "32");
    a.add("L_pjpq", false, "Priority queue length for projection graph searching", "32");
    a.add("num_threads", false, "Number of threads used for building index", ncpu, "T");
    a.add("learn_base_nn_path", true, "Path of learn-base NN file");
    a.add("device", false, "HIP device for the phase-3 beam searches (-1 = all on the CPU, the reference's path)", "-1");
    if (!a.parse(argc, argv)) return -1;
    if (a.help()) { a.usage(std::cout); return 0; }
    std::cout << "sampled query: " << a.str("sampled_query_data_path") << std::endl;
    uint32_t base_num, base_dim, sq_num, sq_dim;
    CK(rg_fbin_meta(a.str("base_data_path").c_str(), &base_num, &base_dim));
    CK(rg_fbin_meta(a.str("sampled_query_data_path").c_str(), &sq_num, &sq_dim));
    int metric;
    const std::string dist = a.str("dist");
    if (dist == "l2") { metric = RG_METRIC_L2; std::cout << "Using l2 as distance metric" << std::endl; }
    else if (dist == "ip") { metric = RG_METRIC_IP; std::cout << "Using inner product as distance metric" << std::endl; }
    else if (dist == "cosine") { metric = RG_METRIC_COSINE; std::cout << "Using cosine as distance metric" << std::endl; }
    else { std::cout << "Unknown distance type: " << dist << std::endl; return -1; }
    uint32_t n = 0, d = 0, stride = 0;
    float *base = nullptr;
    CK(rg_fbin_load(a.str("base_data_path").c_str(), &n, &d, &stride, &base));
    std::cout << "Index save path: " << a.str("projection_index_save_path") << std::endl;
    uint32_t knn_n = 0, knn_k = 0, *knn = nullptr;
    CK(rg_knn_ids_load(a.str("learn_base_nn_path").c_str(), &knn_n, &knn_k, &knn));
    std::cout << "learn base knn npts: " << knn_n << ", k_dim: " << knn_k << std::endl;
    auto s = std::chrono::high_resolution_clock::now();
    uint32_t ep = 0, *nbrs = nullptr;
    uint64_t *off = nullptr;
    // the reference passes the UNALIGNED dimension as the row length (test_build_roargraph.cpp:117) and scores it through
    // AVX-512 4-wide and masked tails (distance.h:206-219); here rows are zero padded and scored at the padded stride.
    // For dim % 8 == 0 (every BASELINE dimension) that is the same arithmetic; for other dimensions the tail is summed in
    // a different association, distances can differ in the last bit and the built graph is not claimed to match.
    const int device = std::atoi(a.str("device").c_str());
    if (device < 0)
        CK(rg_build_roargraph(base, n, stride, stride, knn, knn_n, knn_k, metric, (uint32_t)a.u("M_sq"), (uint32_t)a.u("M_pjbp"),
                              (uint32_t)a.u("L_pjpq"), (uint32_t)a.u("num_threads"), &ep, &off, &nbrs));
    else
        CK(rg_build_roargraph_gpu(base, n, stride, stride, knn, knn_n, knn_k, metric, (uint32_t)a.u("M_sq"),
                                  (uint32_t)a.u("M_pjbp"), (uint32_t)a.u("L_pjpq"), (uint32_t)a.u("num_threads"), device, 0,
                                  &ep, &off, &nbrs));
    const double secs = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - s).count();
    std::cout << "projection ep: " << ep << std::endl;
    std::cout << "Projection degree avg: " << (double)off[n] / n << std::endl;
    std::cout << "indexing time: " << secs << "\n";
    CK(rg_graph_save(a.str("projection_index_save_path").c_str(), n, ep, off, nbrs));
    std::cout << "Save index to " << a.str("projection_index_save_path") << std::endl;
    rg_free(base); rg_free(knn); rg_free(off); rg_free(nbrs);
    return 0;
}
