/*
 * rg_oracle_build.c -- TEST INFRASTRUCTURE ONLY (part of oracle/librg_oracle.so).
 *
 * A plain-C restatement of the reference's graph construction at ONE thread -- IndexBipartite::BuildRoarGraph ->
 * CalculateProjectionep -> LinkProjection with its pruning rules -- written from src/index_bipartite.cpp alone, sweep by
 * sweep, without the shortcuts the product's builder takes (roargraph_amd/csrc/rg_build.cpp skips the iterations of the
 * second sweeps that cannot change a list).  It exists so that the product's build has a checker that is not the
 * product: tests/test_build.py compares the two byte for byte, and both with the md5 of the index the survey's probe
 * build of the REFERENCE wrote.
 *
 * Parity status: UNPINNED.  src/index_bipartite.cpp cannot be compiled in this image (Boost / tsl headers absent,
 * stand-ins not allowed), so this file is a second restatement, not a pin.  Where the reference would read past the
 * end of a vector (an empty candidate pool, a pool whose every entry is already a neighbour) the list comes out
 * empty here.
 *
 * All file:line citations are into /root/reference/src/index_bipartite.cpp.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "rg_oracle.h"

typedef struct { uint32_t *v; uint32_t n, cap; } list_t;
typedef struct { uint32_t id; float dist; } nb_t;      /* efanna2e::Neighbor without the flag (neighbor.h:21-33) */

static void list_push(list_t *l, uint32_t x) {
    if (l->n == l->cap) { l->cap = l->cap ? 2 * l->cap : 8; l->v = (uint32_t *)realloc(l->v, (size_t)l->cap * 4); }
    l->v[l->n++] = x;
}
static int list_has(const list_t *l, uint32_t x) {
    for (uint32_t i = 0; i < l->n; ++i) if (l->v[i] == x) return 1;
    return 0;
}
static void list_assign(list_t *dst, const list_t *src) {
    dst->n = 0;
    for (uint32_t i = 0; i < src->n; ++i) list_push(dst, src->v[i]);
}
/* Neighbor::operator< (neighbor.h:29-31): distance, then id */
static int nb_cmp(const void *a, const void *b) {
    const nb_t *x = (const nb_t *)a, *y = (const nb_t *)b;
    if (x->dist < y->dist) return -1;
    if (x->dist > y->dist) return 1;
    return x->id < y->id ? -1 : x->id > y->id ? 1 : 0;
}

typedef struct {
    const float *base;
    size_t stride;
    unsigned d;
    int metric;          /* RGO_L2 or RGO_IP (cosine: normalised rows, IP kernel -- index.cpp:8-26) */
    uint32_t nd, M, L, Nq, ep;
    list_t *proj, *supply;
} B;

static float dist(const B *b, uint32_t x, uint32_t y) {   /* distance_->compare(data_bp_ + dim * x, data_bp_ + dim * y, dim) */
    return rgo_compare(b->metric, b->base + (size_t)x * b->stride, b->base + (size_t)y * b->stride, b->d);
}

/* the occlusion sweep all four rules share: walk q[(*start)+1 ..) while the result is short; p is dropped when it is in the
 * result already or some chosen r is closer to p than p is to the pivot (djk < p.distance); `self` is never chosen;
 * `check_dup`: the second sweeps also refuse an id the result already holds (a no-op given the id test above) */
static void sweep(const B *b, const nb_t *q, uint32_t nq, uint32_t *start, uint32_t self, list_t *result, int check_dup) {
    while (result->n < b->M && ++(*start) < nq) {
        const nb_t *p = &q[*start];
        int occlude = 0;
        for (uint32_t i = 0; i < result->n; ++i) {
            if (p->id == result->v[i]) { occlude = 1; break; }
            if (dist(b, p->id, result->v[i]) < p->dist) { occlude = 1; break; }
        }
        if (!occlude && p->id != self && !(check_dup && list_has(result, p->id))) list_push(result, p->id);
    }
}

/* PruneBiSearchBaseGetBase, :1612-1694 */
static void prune_get_base(const B *b, const nb_t *pool, uint32_t np, uint32_t tgt, list_t *out) {
    nb_t *bp = (nb_t *)malloc(((size_t)np + 1) * sizeof(nb_t));
    uint32_t nbp = 0;
    for (uint32_t i = 0; i < np; ++i) {                    /* :1621-1629: first occurrence of every id, the target left out */
        int seen = 0;
        for (uint32_t j = 0; j < nbp; ++j) if (bp[j].id == pool[i].id) { seen = 1; break; }
        if (seen || pool[i].id == tgt) continue;
        bp[nbp++] = pool[i];
    }
    list_t result = {0};
    if (nbp) {
        qsort(bp, nbp, sizeof(nb_t), nb_cmp);               /* :1631 */
        uint32_t start = 0;
        list_push(&result, bp[0].id);                       /* :1635 */
        sweep(b, bp, nbp, &start, tgt, &result, 0);         /* :1637-1655 */
        start = 0;                                          /* :1657-1681: the second sweep walks the UNSORTED search pool */
        while (result.n < b->M && ++start < np) {
            const nb_t *p = &pool[start];
            if (list_has(&result, p->id)) continue;
            int occlude = 0;
            for (uint32_t t = 0; t < result.n; ++t) {
                if (p->id == result.v[t]) { occlude = 1; break; }
                if (dist(b, p->id, result.v[t]) < p->dist) { occlude = 1; break; }
            }
            if (!occlude && p->id != tgt && !list_has(&result, p->id)) list_push(&result, p->id);
        }
        for (uint32_t i = 1; i < nbp && result.n < b->M; ++i)   /* :1683-1689 top-up in sorted order */
            if (!list_has(&result, bp[i].id) && bp[i].id != tgt) list_push(&result, bp[i].id);
    }
    list_assign(out, &result);
    free(result.v);
    free(bp);
}

/* PruneProjectionReverseCandidates (:1526-1610, phantoms = 0) and PruneProjectionInternalReverseCandidates (:1434-1524,
 * phantoms = 1: the queue starts with list->n value-initialised Neighbors, id 0 / distance 0 -- so a genuine neighbour 0 is
 * never queued, and node 0 can enter a list it was never in; no top-up sweep) */
static void prune_reverse(const B *b, uint32_t src, list_t *list, int phantoms) {
    nb_t *q = (nb_t *)malloc(((size_t)2 * list->n + 1) * sizeof(nb_t));
    uint32_t nq = 0;
    if (phantoms) for (uint32_t i = 0; i < list->n; ++i) { q[nq].id = 0; q[nq].dist = 0.0f; ++nq; }   /* :1438 */
    for (uint32_t i = 0; i < list->n; ++i) {
        const float dd = dist(b, src, list->v[i]);
        int seen = 0;                                        /* std::find with Neighbor::operator== (id only) */
        for (uint32_t j = 0; j < nq; ++j) if (q[j].id == list->v[i]) { seen = 1; break; }
        if (!seen) { q[nq].id = list->v[i]; q[nq].dist = dd; ++nq; }
    }
    list_t result = {0};
    if (nq) {
        qsort(q, nq, sizeof(nb_t), nb_cmp);
        uint32_t start = 0;
        if (q[start].id == src) ++start;
        if (start < nq) {
            list_push(&result, q[start].id);
            sweep(b, q, nq, &start, src, &result, 0);       /* first sweep */
            start = 0;
            sweep(b, q, nq, &start, src, &result, 1);       /* second sweep, from q[1] */
            if (!phantoms)                                    /* :1594-1598: top-up in the original list order */
                for (uint32_t i = 0; i < list->n && result.n < b->M; ++i)
                    if (!list_has(&result, list->v[i])) list_push(&result, list->v[i]);
        }
    }
    list_assign(list, &result);
    free(result.v);
    free(q);
}

/* ProjectionAddReverse (:1391-1432: graph = proj, limit M, plain rule) / SupplyAddReverse (:1352-1389: graph = supply,
 * limit 2M, phantom rule) */
static void add_reverse(B *b, list_t *g, uint32_t src, uint32_t limit, int phantoms) {
    for (uint32_t i = 0; i < g[src].n; ++i) {               /* the live list: pruning `des` never rewrites g[src] */
        const uint32_t des = g[src].v[i];
        list_t *dn = &g[des];
        if (list_has(dn, src)) continue;
        if (dn->n < limit) { list_push(dn, src); continue; }
        list_t copy = {0};
        list_assign(&copy, dn);
        list_push(&copy, src);
        prune_reverse(b, des, &copy, phantoms);
        list_assign(dn, &copy);
        free(copy.v);
    }
}

/* PruneProjectionBaseSearchCandidates, :1846-1940 */
static void prune_search(const B *b, nb_t *pool, uint32_t np, uint32_t node, list_t *out) {
    list_t result = {0};
    if (np) {
        qsort(pool, np, sizeof(nb_t), nb_cmp);              /* :1853 */
        uint32_t start = 0;
        if (pool[start].id == node) ++start;                /* :1858-1860 */
        const list_t *have = &b->proj[node];
        while (start < np && list_has(have, pool[start].id)) ++start;   /* :1862-1864 */
        if (start < np) {
            list_push(&result, pool[start].id);             /* :1865 */
            sweep(b, pool, np, &start, node, &result, 0);   /* :1867-1892 */
            start = 0;
            sweep(b, pool, np, &start, node, &result, 1);   /* :1893-1924 */
        }
    }
    list_assign(out, &result);
    free(result.v);
}

/* SearchProjectionGraphInternal, :1279-1350: beam search over supply_nbrs_ from the entry point with the node's own row as
 * the query; the node itself is never scored; full_retset = the nodes in the order they were popped */
static uint32_t search_internal(const B *b, uint32_t node, uint8_t *visited, uint32_t *touched, nb_t *full) {
    rgo_queue q;
    rgo_queue_init(&q, b->L);
    const float *query = b->base + (size_t)node * b->stride;
    uint32_t nt = 0, nfull = 0;
    rgo_queue_insert(&q, b->ep, rgo_compare(b->metric, b->base + (size_t)b->ep * b->stride, query, b->d));
    visited[b->ep] = 1; touched[nt++] = b->ep;
    while (rgo_queue_has_unexpanded(&q)) {
        const rgo_nb cur = rgo_queue_pop(&q);
        full[nfull].id = cur.id; full[nfull].dist = cur.dist; ++nfull;
        const list_t *l = &b->supply[cur.id];
        for (uint32_t i = 0; i < l->n; ++i) {
            const uint32_t nbr = l->v[i];
            if (visited[nbr] || nbr == node) continue;
            visited[nbr] = 1; touched[nt++] = nbr;
            rgo_queue_insert(&q, nbr, rgo_compare(b->metric, b->base + (size_t)nbr * b->stride, query, b->d));
        }
    }
    for (uint32_t i = 0; i < nt; ++i) visited[touched[i]] = 0;   /* (the reference allocates a fresh bitset per node, :1195) */
    rgo_queue_free(&q);
    return nfull;
}

/* pool of a node's current list, first occurrence of every id, scored against the node (:1112-1123, :1229-1239) */
static uint32_t scored_unique(const B *b, const list_t *l, uint32_t node, nb_t *out) {
    uint32_t n = 0;
    for (uint32_t j = 0; j < l->n; ++j) {
        int seen = 0;
        for (uint32_t k = 0; k < n; ++k) if (out[k].id == l->v[j]) { seen = 1; break; }
        if (seen) continue;
        out[n].id = l->v[j]; out[n].dist = dist(b, l->v[j], node); ++n;
    }
    return n;
}

/* `sched` (may be NULL): phase 3 as a sequence of batches, sched[i] = number of consecutive nodes in batch i (their sum must be
 * nb).  The nodes of a batch all search the supply graph AS IT STOOD WHEN THE BATCH BEGAN and are then linked one after the
 * other in node order -- the order in which a batched (GPU-assisted) builder sees the graph; the reference's own OpenMP loop
 * (:1192, schedule(dynamic)) lets every thread search a graph that is some other threads' links behind in the same way, only
 * without saying which.  A batch of one node is the reference's one-thread sequence; NULL means all batches are of one. */
int rgo_build_roargraph_sched(const float *base_in, size_t stride, uint32_t nb, unsigned d, int metric, const uint32_t *knn, uint32_t nq,
                              uint32_t knn_k, uint32_t M_sq, uint32_t M_pjbp, uint32_t L_pjpq, const uint32_t *sched, uint32_t nsched,
                              uint32_t *out_ep, uint64_t **out_off, uint32_t **out_nbrs) {
    B b;
    memset(&b, 0, sizeof b);
    float *normed = NULL;
    if (metric == RGO_COSINE) {                              /* BuildRoarGraph normalises the base in place, :176-182 */
        normed = (float *)malloc((size_t)nb * stride * sizeof(float));
        memcpy(normed, base_in, (size_t)nb * stride * sizeof(float));
        rgo_normalize_rows(normed, nb, stride, d);
        base_in = normed;
    }
    b.base = base_in; b.stride = stride; b.d = d; b.metric = metric == RGO_L2 ? RGO_L2 : RGO_IP;
    b.nd = nb; b.M = M_pjbp; b.L = L_pjpq; b.Nq = M_sq;
    b.proj = (list_t *)calloc(nb, sizeof(list_t));
    b.supply = (list_t *)calloc(nb, sizeof(list_t));
    b.ep = rgo_projection_ep(base_in, stride, nb, d);        /* CalculateProjectionep, :203 / :2004-2041 */
    const uint32_t pool_cap = (knn_k > 4 * M_pjbp ? knn_k : 4 * M_pjbp) + L_pjpq + 8;
    nb_t *pool = (nb_t *)malloc((size_t)pool_cap * sizeof(nb_t));
    /* ---- phase 1, :1059-1097: every training query links its nearest base point to its other near neighbours */
    for (uint32_t sq = 0; sq < nq; ++sq) {
        const uint32_t n = knn_k < M_sq ? knn_k : M_sq;      /* nn_base.resize(Nq), :1063-1066 */
        if (n == 0) continue;
        const uint32_t *nn = knn + (size_t)sq * knn_k;
        const uint32_t tgt = nn[0];
        uint32_t np = 0;
        for (uint32_t i = 0; i < n; ++i) {
            if (nn[i] == tgt) continue;
            pool[np].id = nn[i]; pool[np].dist = dist(&b, nn[i], tgt); ++np;
        }
        prune_get_base(&b, pool, np, tgt, &b.proj[tgt]);     /* projection_graph_[cur_tgt] = pruned_list, :1087-1090 */
        add_reverse(&b, b.proj, tgt, b.M, 0);                /* :1091 */
    }
    /* ---- phase 2, :1100-1137: reverse edges of every node, then lists above M are pruned again */
    for (uint32_t node = 0; node < nb; ++node) add_reverse(&b, b.proj, node, b.M, 0);
    for (uint32_t node = 0; node < nb; ++node) {
        if (b.proj[node].n <= b.M) continue;
        uint32_t np = scored_unique(&b, &b.proj[node], node, pool), w = 0;
        for (uint32_t j = 0; j < np; ++j) if (pool[j].id != node) pool[w++] = pool[j];   /* :1124-1129 */
        prune_get_base(&b, pool, w, node, &b.proj[node]);
    }
    for (uint32_t i = 0; i < nb; ++i) list_assign(&b.supply[i], &b.proj[i]);   /* :1183-1188 */
    /* ---- phase 3, :1192-1220: connectivity enhancement -- every node searches the supply graph with its own row */
    uint8_t *visited = (uint8_t *)calloc(nb, 1);
    uint32_t *touched = (uint32_t *)malloc((size_t)nb * 4);
    nb_t *full = (nb_t *)malloc(((size_t)nb + 1) * sizeof(nb_t));
    for (uint32_t node = 0, bi = 0; node < nb; ++bi) {
        const uint32_t n = (sched && bi < nsched && sched[bi]) ? sched[bi] : 1;
        if (n == 1) {                                             /* the reference's one-thread sequence */
            uint32_t nf = search_internal(&b, node, visited, touched, full), w = 0;
            for (uint32_t j = 0; j < nf; ++j) if (full[j].id != node) full[w++] = full[j];   /* :1203-1208 */
            prune_search(&b, full, w, node, &b.supply[node]);     /* supply_nbrs_[node] = pruned_list, :1209-1214 */
            add_reverse(&b, b.supply, node, 2 * b.M, 1);          /* SupplyAddReverse, :1215 */
            ++node;
            continue;
        }
        /* a batch: searches over the frozen graph first (their pruned lists kept aside), then the links in node order */
        const uint32_t hi = node + n < nb ? node + n : nb;
        list_t *kept = (list_t *)calloc(hi - node, sizeof(list_t));
        for (uint32_t x = node; x < hi; ++x) {
            uint32_t nf = search_internal(&b, x, visited, touched, full), w = 0;
            for (uint32_t j = 0; j < nf; ++j) if (full[j].id != x) full[w++] = full[j];
            prune_search(&b, full, w, x, &kept[x - node]);
        }
        for (uint32_t x = node; x < hi; ++x) {
            list_assign(&b.supply[x], &kept[x - node]);
            add_reverse(&b, b.supply, x, 2 * b.M, 1);
            free(kept[x - node].v);
        }
        free(kept);
        node = hi;
    }
    /* ---- phase 4, :1224-1249: supply lists above M are pruned with the search rule */
    for (uint32_t node = 0; node < nb; ++node) {
        if (b.supply[node].n <= b.M) continue;
        const uint32_t np = scored_unique(&b, &b.supply[node], node, full);
        prune_search(&b, full, np, node, &b.supply[node]);
    }
    /* ---- phase 5, :1252-1270: the supply list joins the projection list, at most 2M new ids that it does not hold yet */
    for (uint32_t i = 0; i < nb; ++i) {
        list_t ok = {0};
        for (uint32_t j = 0; j < b.supply[i].n; ++j) {
            if (ok.n >= 2 * b.M) break;
            if (!list_has(&b.proj[i], b.supply[i].v[j])) list_push(&ok, b.supply[i].v[j]);
        }
        for (uint32_t j = 0; j < ok.n; ++j) list_push(&b.proj[i], ok.v[j]);
        free(ok.v);
    }
    /* ---- SaveProjectionGraph's content (:2606-2619) as CSR */
    uint64_t *off = (uint64_t *)malloc(((size_t)nb + 1) * 8), ne = 0;
    for (uint32_t i = 0; i < nb; ++i) { off[i] = ne; ne += b.proj[i].n; }
    off[nb] = ne;
    uint32_t *nbrs = (uint32_t *)malloc((ne ? ne : 1) * 4);
    for (uint32_t i = 0; i < nb; ++i) memcpy(nbrs + off[i], b.proj[i].v, (size_t)b.proj[i].n * 4);
    *out_ep = b.ep; *out_off = off; *out_nbrs = nbrs;
    for (uint32_t i = 0; i < nb; ++i) { free(b.proj[i].v); free(b.supply[i].v); }
    free(b.proj); free(b.supply); free(pool); free(visited); free(touched); free(full); free(normed);
    return 0;
}

int rgo_build_roargraph(const float *base_in, size_t stride, uint32_t nb, unsigned d, int metric, const uint32_t *knn, uint32_t nq,
                        uint32_t knn_k, uint32_t M_sq, uint32_t M_pjbp, uint32_t L_pjpq, uint32_t *out_ep, uint64_t **out_off,
                        uint32_t **out_nbrs) {
    return rgo_build_roargraph_sched(base_in, stride, nb, d, metric, knn, nq, knn_k, M_sq, M_pjbp, L_pjpq, NULL, 0, out_ep, out_off, out_nbrs);
}


/* One call of one pruning rule (tests: the rules against the goldens `rg_ref prune` made with the reference's own Distance / Neighbor
 * objects -- tests/golden/prune_*.npz).  kind 0 = PruneBiSearchBaseGetBase, 1 = PruneProjectionReverseCandidates, 2 =
 * PruneProjectionInternalReverseCandidates, 3 = PruneProjectionBaseSearchCandidates (`have` = projection_graph_[pivot]).
 * out has room for max(M, np) ids; returns 0. */
int rgo_prune(const float *base, size_t stride, uint32_t nb, unsigned d, int metric, uint32_t M, int kind, uint32_t pivot, const uint32_t *ids,
              const float *dists, uint32_t np, const uint32_t *have, uint32_t nhave, uint32_t *out, uint32_t *nout) {
    B b;
    memset(&b, 0, sizeof b);
    b.base = base; b.stride = stride; b.d = d; b.metric = metric; b.nd = nb; b.M = M;
    list_t res = {0};
    if (kind == 1 || kind == 2) {
        for (uint32_t i = 0; i < np; ++i) list_push(&res, ids[i]);
        prune_reverse(&b, pivot, &res, kind == 2);
    } else {
        nb_t *pool = (nb_t *)malloc(((size_t)np + 1) * sizeof(nb_t));
        for (uint32_t i = 0; i < np; ++i) { pool[i].id = ids[i]; pool[i].dist = dists[i]; }
        if (kind == 0) prune_get_base(&b, pool, np, pivot, &res);
        else {
            list_t hv = {0};
            for (uint32_t i = 0; i < nhave; ++i) list_push(&hv, have[i]);
            b.proj = (list_t *)calloc((size_t)pivot + 1, sizeof(list_t));
            b.proj[pivot] = hv;
            prune_search(&b, pool, np, pivot, &res);
            free(hv.v);
            free(b.proj);
        }
        free(pool);
    }
    for (uint32_t i = 0; i < res.n; ++i) out[i] = res.v[i];
    *nout = res.n;
    free(res.v);
    return 0;
}
