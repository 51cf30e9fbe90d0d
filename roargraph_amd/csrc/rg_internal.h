// rg_internal.h -- shared between the host-only and the HIP translation units of librg_hip.so
#pragma once
#include <cstdint>
#include <string>

#include "rg.h"

namespace rg {
extern thread_local std::string g_last_error;
rg_status set_error(rg_status code, const std::string &msg);
// row stride rule of data_align (include/efanna2e/util.h:37-75): pad to a multiple of 8 floats
inline uint32_t aligned_dim(uint32_t d) { return (d + 7u) / 8u * 8u; }
}  // namespace rg
