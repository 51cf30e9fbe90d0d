// compute_groundtruth -- twin of the external tool RoarGraph's pipeline calls (README.md:62-75):
//   ./compute_groundtruth --data_type float --dist_fn mips --base_file B.fbin --query_file Q.fbin --gt_file gt.bin --K 100
// Exact top-K by brute force on MI355X (fp32-input MFMA + fused top-K), base rows sharded over --devices.
// Output: u32 npts, u32 K, ids[npts][K], dists[npts][K] (ids block then dists block; +inner product for mips).
#include <chrono>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "argparse_lite.h"
#include "rg.h"

int main(int argc, char **argv) {
    Args a;
    a.add("data_type", true, "data type <float> (int8/uint8 are not supported)");
    a.add("dist_fn", true, "distance function <l2/mips/cosine>");
    a.add("base_file", true, "File containing the base vectors in binary format");
    a.add("query_file", true, "File containing the query vectors in binary format");
    a.add("gt_file", true, "File name for the writing ground truth in binary format");
    a.add("K", true, "Number of ground truth nearest neighbors to compute");
    a.add("devices", false, "comma separated HIP device indices", "0");
    if (!a.parse(argc, argv)) return -1;
    if (a.help()) { a.usage(std::cout); return 0; }
    if (a.str("data_type") != "float") { std::cout << "Unsupported type. float, int8 and uint8 types are supported." << std::endl; return -1; }
    int metric;
    const std::string fn = a.str("dist_fn");
    if (fn == "l2") metric = RG_METRIC_L2;
    else if (fn == "mips") metric = RG_METRIC_IP;
    else if (fn == "cosine") metric = RG_METRIC_COSINE;
    else { std::cerr << "Unsupported distance function. Use l2/mips/cosine." << std::endl; return -1; }
    std::vector<int> devs;
    std::stringstream ss(a.str("devices"));
    for (std::string t; std::getline(ss, t, ',');) devs.push_back(std::atoi(t.c_str()));
    uint32_t nb = 0, nq = 0, d = 0;
    if (rg_fbin_meta(a.str("base_file").c_str(), &nb, &d) != RG_OK || rg_fbin_meta(a.str("query_file").c_str(), &nq, &d) != RG_OK) {
        std::cerr << rg_last_error() << std::endl;
        return -1;
    }
    auto t0 = std::chrono::high_resolution_clock::now();
    if (rg_groundtruth(a.str("base_file").c_str(), a.str("query_file").c_str(), a.str("gt_file").c_str(), metric,
                       (uint32_t)a.u("K"), devs.data(), (int)devs.size()) != RG_OK) {
        std::cerr << rg_last_error() << std::endl;
        return -1;
    }
    const double s = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - t0).count();
    std::cout << "Finished writing truthset to " << a.str("gt_file") << " : " << nq << " x " << a.u("K") << " from " << nb
              << " base points in " << s << " s (" << (double)nq * nb / s << " distances/s incl. file I/O)" << std::endl;
    return 0;
}
