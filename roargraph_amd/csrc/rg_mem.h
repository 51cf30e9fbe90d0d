// rg_mem.h -- device memory for the large, randomly read buffers of an index, balanced over the memory classes of the
// device (rg_mem.hip has the measurement and the method).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>

#include "rg.h"

namespace rg {
// bytes of device memory on `device` (the current device must be `device`).  2 GiB and more: 1-GiB granules of measured
// class, taken round robin over the classes; less -- or wherever balancing is not possible -- a plain hipMalloc.
// RG_ERR_OOM when not even that succeeds.  Contents are undefined.
rg_status dev_alloc(int device, size_t bytes, void **out);
template <typename T>
inline rg_status dev_alloc_t(int device, size_t n, T **out) {
    void *p = nullptr;
    rg_status st = dev_alloc(device, n * sizeof(T), &p);
    *out = static_cast<T *>(p);
    return st;
}
// releases what dev_alloc returned (or any hipMalloc'ed pointer); null is fine
void dev_free(void *p);
// hands the pool's spare granules back to the device (freed balanced buffers stay cached, mapped, for the next request of
// their size: rg_mem_release hands those back too)
void dev_trim(int device);
// hipMalloc that, refused, hands the allocator's cache and spare granules back to the (current) device and tries once more
hipError_t dev_malloc_retry(void **out, size_t bytes);
// did the calling thread's last dev_alloc of 2 GiB or more fall back to a plain allocation (one memory class)?
bool dev_last_plain();
// host (pageable) -> device through two pinned 64-MiB chunks (round 5): gigabytes of pageable memory handed to hipMemcpy are
// pinned page by page by the runtime for the time of the copy; two bench runs of the round died of a GPU memory access fault a few
// pages into such a region while the host was under memory pressure.  The library's own large uploads go through memory the
// driver owns instead.  Synchronous; falls back to a plain hipMemcpy when the chunks cannot be had.
rg_status upload_staged(void *d_dst, const void *h_src, size_t bytes);
}  // namespace rg
