// rg_internal.h -- shared between the host-only and the HIP translation units of librg_hip.so
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

#include "rg.h"

namespace rg {
extern thread_local std::string g_last_error;
rg_status set_error(rg_status code, const std::string &msg);
// row stride rule of data_align (include/efanna2e/util.h:37-75): pad to a multiple of 8 floats
inline uint32_t aligned_dim(uint32_t d) { return (d + 7u) / 8u * 8u; }
// caller-owned scratch of the ground-truth shard kernel (rg_gt.hip); grow-only, one per rank thread
struct GtWorkspace { void *p[10] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; size_t cap[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; };
void gt_workspace_free(GtWorkspace *ws);
rg_status gt_shard_ws(const float *d_base, uint32_t nb, uint32_t bstride, const float *d_queries, uint32_t nq, uint32_t qstride,
                      uint32_t dim, int metric, uint32_t K, uint32_t id_base, uint32_t *d_ids, float *d_dists, int device,
                      void *stream, GtWorkspace *ws);
}  // namespace rg
