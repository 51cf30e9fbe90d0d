// rg_formats.cpp -- host side of the drop-in surface: the reference's file formats
//  with its validation rules and messages.  Pure C++, no HIP.
//
//   .fbin   load_meta / load_data / data_align       include/efanna2e/util.h:106-127, 179-211, 37-75
//   gt      load_gt_meta / load_gt_data_with_dist    include/efanna2e/util.h:84-105, 129-155
//   knn ids LoadLearnBaseKNN                         src/index_bipartite.cpp:2622-2642
//   .index  Load/SaveProjectionGraph                 src/index_bipartite.cpp:2097-2117, 2606-2619
//   recall  ComputeRecall                            tests/test_search_roargraph.cpp:23-36
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rg.h"
#include "rg_internal.h"

namespace rg {
thread_local std::string g_last_error;
rg_status set_error(rg_status code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

namespace {
struct File {
    FILE *f = nullptr;
    explicit File(const char *path, const char *mode) { f = std::fopen(path, mode); }
    ~File() { if (f) std::fclose(f); }
    bool ok() const { return f != nullptr; }
    uint64_t size() {
        long cur = std::ftell(f);
        std::fseek(f, 0, SEEK_END);
        long n = std::ftell(f);
        std::fseek(f, cur, SEEK_SET);
        return (uint64_t)n;
    }
    bool read(void *dst, size_t bytes) { return std::fread(dst, 1, bytes, f) == bytes; }
    bool write(const void *src, size_t bytes) { return std::fwrite(src, 1, bytes, f) == bytes; }
};

// header {npts, width} + the reference's "points contained" rule; mult = 1 for .fbin, 2 for gt files
rg_status read_header(const char *path, uint32_t mult, uint32_t *npts, uint32_t *width) {
    File in(path, "rb");
    if (!in.ok()) return set_error(RG_ERR_IO, "open file error");
    uint32_t h[2] = {0, 0};
    if (!in.read(h, 8) || h[1] == 0) return set_error(RG_ERR_FORMAT, "Data file size wrong!");
    uint64_t bytes = in.size();
    uint32_t contained = (uint32_t)((bytes - 8) / h[1] / 4);
    if ((uint32_t)(h[0] * mult) != contained)
        return set_error(RG_ERR_FORMAT, "Data file size wrong!");
    *npts = h[0];
    *width = h[1];
    return RG_OK;
}
}  // namespace
}  // namespace rg

using rg::set_error;

extern "C" {

const char *rg_last_error(void) { return rg::g_last_error.c_str(); }
const char *rg_version(void) { return "roargraph_amd 0.1 (gfx950)"; }
void rg_free(void *p) { std::free(p); }

rg_status rg_fbin_meta(const char *path, uint32_t *npts, uint32_t *dim) { return rg::read_header(path, 1, npts, dim); }

rg_status rg_fbin_load(const char *path, uint32_t *npts, uint32_t *dim, uint32_t *stride, float **data) {
    rg_status st = rg::read_header(path, 1, npts, dim);
    if (st != RG_OK) return st;
    rg::File in(path, "rb");
    if (!in.ok()) return set_error(RG_ERR_IO, "open file error");
    std::fseek(in.f, 8, SEEK_SET);
    const size_t n = *npts, d = *dim, sd = rg::aligned_dim(*dim);
    float *buf = (float *)std::malloc(std::max<size_t>(n * sd * sizeof(float), 64));
    if (!buf) return set_error(RG_ERR_OOM, "out of host memory");
    if (sd == d) {
        if (!in.read(buf, n * d * 4)) { std::free(buf); return set_error(RG_ERR_FORMAT, "Data file size wrong!"); }
    } else {
        for (size_t i = 0; i < n; ++i) {
            if (!in.read(buf + i * sd, d * 4)) { std::free(buf); return set_error(RG_ERR_FORMAT, "Data file size wrong!"); }
            std::memset(buf + i * sd + d, 0, (sd - d) * 4);
        }
    }
    *stride = (uint32_t)sd;
    *data = buf;
    return RG_OK;
}

rg_status rg_fbin_save(const char *path, const float *data, uint32_t npts, uint32_t dim, uint32_t stride) {
    rg::File out(path, "wb");
    if (!out.ok()) return set_error(RG_ERR_IO, "cannot open file");
    uint32_t h[2] = {npts, dim};
    out.write(h, 8);
    for (size_t i = 0; i < npts; ++i)
        if (!out.write(data + i * (size_t)stride, (size_t)dim * 4)) return set_error(RG_ERR_IO, "short write");
    return RG_OK;
}

rg_status rg_gt_meta(const char *path, uint32_t *npts, uint32_t *k) { return rg::read_header(path, 2, npts, k); }

rg_status rg_gt_load(const char *path, uint32_t *npts, uint32_t *k, uint32_t **ids, float **dists) {
    rg_status st = rg::read_header(path, 2, npts, k);
    if (st != RG_OK) return st;
    rg::File in(path, "rb");
    if (!in.ok()) return set_error(RG_ERR_IO, "open file error");
    std::fseek(in.f, 8, SEEK_SET);
    const size_t cells = (size_t)*npts * *k;
    uint32_t *a = (uint32_t *)std::malloc(cells * 4 + 4);
    float *b = (float *)std::malloc(cells * 4 + 4);
    if (!a || !b) { std::free(a); std::free(b); return set_error(RG_ERR_OOM, "out of host memory"); }
    if (!in.read(a, cells * 4) || !in.read(b, cells * 4)) {
        std::free(a); std::free(b);
        return set_error(RG_ERR_FORMAT, "Data file size wrong!");
    }
    *ids = a;
    *dists = b;
    return RG_OK;
}

rg_status rg_gt_save(const char *path, const uint32_t *ids, const float *dists, uint32_t npts, uint32_t k) {
    rg::File out(path, "wb");
    if (!out.ok()) return set_error(RG_ERR_IO, "cannot open file");
    uint32_t h[2] = {npts, k};
    bool ok = out.write(h, 8) && out.write(ids, (size_t)npts * k * 4);
    if (ok && dists) ok = out.write(dists, (size_t)npts * k * 4);
    return ok ? RG_OK : set_error(RG_ERR_IO, "short write");
}

rg_status rg_knn_ids_load(const char *path, uint32_t *npts, uint32_t *k, uint32_t **ids) {
    rg::File in(path, "rb");
    if (!in.ok()) return set_error(RG_ERR_IO, std::string("Could not open file ") + path);
    uint32_t h[2];
    if (!in.read(h, 8)) return set_error(RG_ERR_FORMAT, "learn base knn file error");
    const size_t cells = (size_t)h[0] * h[1];
    uint32_t *a = (uint32_t *)std::malloc(cells * 4 + 4);
    if (!a) return set_error(RG_ERR_OOM, "out of host memory");
    if (!in.read(a, cells * 4)) { std::free(a); return set_error(RG_ERR_FORMAT, "learn base knn file error"); }
    *npts = h[0];
    *k = h[1];
    *ids = a;
    return RG_OK;
}

rg_status rg_graph_load(const char *path, uint32_t *nd, uint32_t *ep, uint64_t **offsets, uint32_t **nbrs) {
    rg::File in(path, "rb");
    if (!in.ok()) return set_error(RG_ERR_IO, "projection index file does not exist.");
    const uint64_t bytes = in.size();
    if (bytes < 8) return set_error(RG_ERR_FORMAT, "index file truncated");
    // one read, then a single pass that splits degree words from neighbour ids
    std::vector<uint32_t> raw(bytes / 4);
    if (!in.read(raw.data(), raw.size() * 4)) return set_error(RG_ERR_IO, "index file truncated");
    const uint32_t n = raw[1];
    uint64_t *off = (uint64_t *)std::malloc(((size_t)n + 1) * 8);
    uint32_t *nb = (uint32_t *)std::malloc(std::max<size_t>(raw.size() * 4, 4));
    if (!off || !nb) { std::free(off); std::free(nb); return set_error(RG_ERR_OOM, "out of host memory"); }
    size_t pos = 2, edges = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (pos >= raw.size() || pos + 1 + raw[pos] > raw.size()) {
            std::free(off); std::free(nb);
            return set_error(RG_ERR_FORMAT, "index file truncated");
        }
        const uint32_t deg = raw[pos];
        off[i] = edges;
        std::memcpy(nb + edges, raw.data() + pos + 1, (size_t)deg * 4);
        edges += deg;
        pos += 1 + (size_t)deg;
    }
    off[n] = edges;
    *nd = n;
    *ep = raw[0];
    *offsets = off;
    *nbrs = nb;
    return RG_OK;
}

rg_status rg_graph_save(const char *path, uint32_t nd, uint32_t ep, const uint64_t *offsets, const uint32_t *nbrs) {
    rg::File out(path, "wb");
    if (!out.ok()) return set_error(RG_ERR_IO, "cannot open file");
    uint32_t h[2] = {ep, nd};
    bool ok = out.write(h, 8);
    for (uint32_t i = 0; ok && i < nd; ++i) {
        const uint32_t deg = (uint32_t)(offsets[i + 1] - offsets[i]);
        ok = out.write(&deg, 4) && out.write(nbrs + offsets[i], (size_t)deg * 4);
    }
    return ok ? RG_OK : set_error(RG_ERR_IO, "short write");
}

float rg_recall(uint32_t nq, uint32_t k, uint32_t gt_dim, const uint32_t *res, const uint32_t *gt) {
    uint64_t hit = 0;
    for (uint32_t q = 0; q < nq; ++q) {
        const uint32_t *r = res + (size_t)q * k, *g = gt + (size_t)q * gt_dim;
        for (uint32_t a = 0; a < k; ++a) {
            bool found = false;
            for (uint32_t b = 0; b < k && !found; ++b) found = r[b] == g[a];
            hit += found;
        }
    }
    return (float)(uint32_t)hit / (float)(k * nq);
}

void rg_normalize_rows(float *data, size_t n, size_t stride, uint32_t dim) {
    for (size_t i = 0; i < n; ++i) {
        float *row = data + i * stride;
        float ss = 0.0f;
        for (uint32_t j = 0; j < dim; ++j) ss += row[j] * row[j];
        const float norm = std::sqrt(ss);
        for (uint32_t j = 0; j < dim; ++j) row[j] = row[j] / norm;
    }
}

}  // extern "C"
