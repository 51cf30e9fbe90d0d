// K1 instantiations: metric = IP, adjacency = CSR (see rg_search_kernel.h)
#include "rg_search_kernel.h"

namespace rg {
rg_status launch_search_ip_csr(const SearchParams &P, const K1Launch &c, hipStream_t s) {
    return launch_search_family<false, false>(P, c, s);
}
}  // namespace rg
