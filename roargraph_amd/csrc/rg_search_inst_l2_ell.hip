// K1 instantiations: metric = L2, adjacency = ELL (see rg_search_kernel.h)
#include "rg_search_kernel.h"

namespace rg {
rg_status launch_search_l2_ell(const SearchParams &P, const K1Launch &c, hipStream_t s) {
    return launch_search_family<true, true>(P, c, s);
}
}  // namespace rg
