#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the reference's own code.

Runs ONLY in the build container (needs /root/reference to build oracle/_ref/rg_ref, the driver around the
reference's headers: distance.h, neighbor.h, visited_list_pool.h, util.h).  The outputs are data -- seeded inputs and
the values the reference code returned for them -- and are committed so that the GPU box (which has no reference tree)
can check both the oracle and the HIP path against them.

  G1 dist_<metric>_<d>.npz    a[n,d], b[n,d], expect_bits[n]            DistanceInnerProduct/DistanceL2::compare
  G2 queue_traces.npz         op/id/dist traces + final state            NeighborPriorityQueue insert / closest_unexpanded
  G3 search_<name>.npz        base, queries, graph (CSR), ep and per (L,k): ids, dist bits, cmps, hops
                              (genuine queue + visited pool + distance, loop restated -- see oracle/ref_driver.cpp)
  G4 formats.npz              bytes of small .fbin / gt files (good and truncated) + what util.h's loaders said

  G6 prune_<name>.npz         calls of the four occlusion-pruning rules of the construction (index_bipartite.cpp:1434-1694, 1846-1940) over the
                              base of search_<name>.npz: pools with the reference's own distance bits, and the lists `rg_ref prune` returned
                              (genuine Distance::compare, Neighbor::operator< under std::sort, operator== under std::find; rules restated)

  G5 cli_table.json           the stdout header / row / CSV row FORMATS of tests/test_search_roargraph.cpp (:190, :231-236):
                              the string literals of its stream statements evaluated, variables left as {name}

usage: python scripts/make_golden.py [g1 g2 g3 g3c g4 g5 g6]   (default: all)
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from roargraph_amd import io, synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def g1():
    rng = np.random.default_rng(20240501)
    for metric in ("ip", "l2"):
        for d, n in ((8, 128), (16, 128), (24, 128), (40, 128), (200, 128), (512, 96), (136, 64)):
            a = rng.standard_normal((n, d)).astype(np.float32)
            b = (0.3 + 0.5 * rng.standard_normal((n, d))).astype(np.float32)
            # a few exact-cancellation / zero rows as edge cases
            a[0] = 0.0
            b[1] = a[1]
            exp = po.ref_dist(metric, a, b)
            np.savez_compressed(os.path.join(OUT, "dist_%s_%d.npz" % (metric, d)), a=a, b=b,
                                expect_bits=exp.view(np.uint32))


def g2():
    rng = np.random.default_rng(7)
    traces = {}
    for t, (cap, nops, nid) in enumerate(((1, 40, 8), (4, 200, 20), (10, 600, 60), (50, 3000, 4000),
                                           (100, 6000, 100000), (500, 8000, 100000))):
        ops = (rng.random(nops) < 0.25).astype(np.uint8)
        ids = rng.integers(0, nid, nops).astype(np.uint32)
        # coarse distances so that ties (same distance, different id) and re-inserted ids are common
        ds = (rng.integers(0, 64, nops) / 8.0).astype(np.float32)
        r = po.ref_queue(cap, ops, ids, ds)
        traces.update({"t%d_cap" % t: cap, "t%d_ops" % t: ops, "t%d_ids" % t: ids, "t%d_dists" % t: ds,
                       "t%d_size" % t: r["size"], "t%d_cur" % t: r["cur"], "t%d_out_ids" % t: r["ids"],
                       "t%d_out_dists" % t: r["dists"].view(np.uint32), "t%d_out_flags" % t: r["flags"],
                       "t%d_pops" % t: r["pops"]})
    traces["ntraces"] = 6
    np.savez_compressed(os.path.join(OUT, "queue_traces.npz"), **traces)


def g3():
    sets = (("ip200", "ip", 2000, 200, 48), ("l2_512", "l2", 1000, 512, 32), ("ip24", "ip", 600, 24, 32))
    with tempfile.TemporaryDirectory() as td:
        for name, metric, nb, d, nq in sets:
            base, q = synth.make_synth(1234, nb, nq, d)
            tq = synth.make_synth(4321, nb, 500, d)[1]
            lists, ep = synth.knn_graph(base, metric, M=10, train_queries=tq)
            # format corner cases the file layout allows: an empty list, a list > 64 long, a duplicated id, a self loop
            lists[5] = np.zeros(0, np.uint32)
            lists[7] = np.arange(100, 100 + 150, dtype=np.uint32) % nb
            lists[9] = np.concatenate([lists[9], lists[9][:3], [9]]).astype(np.uint32)
            off, nbrs = io.lists_to_csr(lists)
            bf, qf, gf = (os.path.join(td, x) for x in ("b.fbin", "q.fbin", "g.index"))
            io.write_fbin(bf, base)
            io.write_fbin(qf, q)
            io.write_index(gf, off, nbrs, ep)
            out = dict(base=base, queries=q, offsets=off, nbrs=nbrs, ep=ep, metric=metric)
            cfgs = []
            for L, k in ((10, 10), (50, 10), (100, 100), (500, 10), (1, 1), (64, 10), (65, 65)):
                ids, ds, cmps, hops, _ = po.ref_search(bf, gf, qf, metric, k, L, threads=2)
                tag = "L%d_k%d" % (L, k)
                cfgs.append(tag)
                out.update({tag + "_ids": ids, tag + "_dist_bits": ds.view(np.uint32), tag + "_cmps": cmps,
                            tag + "_hops": hops})
            out["configs"] = np.array(cfgs)
            np.savez_compressed(os.path.join(OUT, "search_%s.npz" % name), **out)


def g3_cosine():
    """G3c search_cos200.npz: `rg_ref search ... cosine` -- the reference's normalize<float> (util.h:214-225) over base
    rows (LoadVectorData, index_bipartite.cpp:2679-2684) and queries (test_search_roargraph.cpp:167-172), then the IP
    kernel.  The fixture holds the RAW rows; the expectations are what the reference build of this container returned.
    normalize compiles (GCC 11.4, -Ofast) to a 16-lane sum of squares, vrsqrtss + one Newton step, and a multiply: the
    vrsqrtss estimate differs between CPU vendors, so these bits are those of THIS machine's reference build and the
    tests compare with the float tolerance of the north star, not bit for bit (tests/test_gpu_golden.py)."""
    nb, d, nq = 1500, 200, 40
    base, q = synth.make_synth(777, nb, nq, d)
    base = (base * np.linspace(0.5, 3.0, nb, dtype=np.float32)[:, None]).astype(np.float32)   # norms that matter
    nbase = base / np.linalg.norm(base.astype(np.float64), axis=1)[:, None].astype(np.float32)
    tq = synth.make_synth(778, nb, 400, d)[1]
    lists, ep = synth.knn_graph(nbase.astype(np.float32), "ip", M=10, train_queries=tq)
    off, nbrs = io.lists_to_csr(lists)
    with tempfile.TemporaryDirectory() as td:
        bf, qf, gf = (os.path.join(td, x) for x in ("b.fbin", "q.fbin", "g.index"))
        io.write_fbin(bf, base)
        io.write_fbin(qf, q)
        io.write_index(gf, off, nbrs, ep)
        out = dict(base=base, queries=q, offsets=off, nbrs=nbrs, ep=ep, metric="cosine")
        cfgs = []
        for L, k in ((10, 10), (50, 10), (100, 100), (300, 10)):
            ids, ds, cmps, hops, _ = po.ref_search(bf, gf, qf, "cosine", k, L, threads=2)
            tag = "L%d_k%d" % (L, k)
            cfgs.append(tag)
            out.update({tag + "_ids": ids, tag + "_dist_bits": ds.view(np.uint32), tag + "_cmps": cmps, tag + "_hops": hops})
        out["configs"] = np.array(cfgs)
        np.savez_compressed(os.path.join(OUT, "search_cos200.npz"), **out)


def g4():
    out = {}
    rng = np.random.default_rng(3)
    with tempfile.TemporaryDirectory() as td:
        def ask(kind, raw):
            p = os.path.join(td, "f.bin")
            open(p, "wb").write(raw)
            r = po.ref_run("meta", kind, p, check=False)
            lines = [l for l in r.stdout.splitlines() if l.startswith(("OK", "EXC"))]
            return lines[-1] if lines else "EXIT %d" % r.returncode

        data = rng.standard_normal((37, 24)).astype(np.float32)
        good = np.array([37, 24], np.uint32).tobytes() + data.tobytes()
        cases = {"fbin_good": good, "fbin_short_row": good[:-96], "fbin_short_bytes": good[:-5],
                 "fbin_extra_bytes": good + b"\0" * 40, "fbin_extra_row": good + b"\0" * 96,
                 "fbin_wrong_count": np.array([36, 24], np.uint32).tobytes() + data.tobytes()}
        for k, raw in cases.items():
            out[k] = np.frombuffer(raw, np.uint8)
            out[k + "_says"] = ask("fbin", raw)
        ids = rng.integers(0, 1000, (11, 5)).astype(np.uint32)
        ds = rng.standard_normal((11, 5)).astype(np.float32)
        ggood = np.array([11, 5], np.uint32).tobytes() + ids.tobytes() + ds.tobytes()
        gcases = {"gt_good": ggood, "gt_ids_only": ggood[: 8 + 220], "gt_short": ggood[:-20],
                  "gt_extra": ggood + b"\0" * 19}
        for k, raw in gcases.items():
            out[k] = np.frombuffer(raw, np.uint8)
            out[k + "_says"] = ask("gt", raw)
        # loaded content of the good files through the genuine loaders
        p = os.path.join(td, "g.bin")
        open(p, "wb").write(ggood)
        po.ref_run("gtload", p, os.path.join(td, "o.bin"))
        raw = np.fromfile(os.path.join(td, "o.bin"), np.uint32)
        out["gt_good_ids"] = raw[:55].reshape(11, 5)
        out["gt_good_dist_bits"] = raw[55:].reshape(11, 5)
        open(p, "wb").write(good)
        po.ref_run("fbinload", p, os.path.join(td, "o.bin"))
        raw = np.fromfile(os.path.join(td, "o.bin"), np.uint32)
        out["fbin_good_loaded_shape"] = raw[:2]
        out["fbin_good_loaded_bits"] = raw[2:].reshape(int(raw[0]), int(raw[1]))
    np.savez_compressed(os.path.join(OUT, "formats.npz"), **out)


def g5():
    """What the reference's search CLI prints: its three stream statements (header :190, table row :231-232, CSV row
    :233-236) evaluated -- string literals decoded, every variable left as a {placeholder}.  Output data, not source."""
    import json
    import re
    src = open("/root/reference/tests/test_search_roargraph.cpp").read()

    def stream_format(stmt):
        parts = [x.strip() for x in stmt.split("<<")]
        out = []
        for x in parts[1:]:
            if x in ("std::endl", "std::endl;"):
                break
            m = re.fullmatch(r'"((?:[^"\\]|\\.)*)"', x)
            if m:
                out.append(m.group(1).encode().decode("unicode_escape"))
            else:
                out.append("{%s}" % re.sub(r"[^A-Za-z0-9_/ ()]", "", x).strip())
        return "".join(out)

    def statement(anchor, start=0):
        i = src.index(anchor, start)
        j = src.index("std::endl;", i)
        return " ".join(src[i:j + len("std::endl;")].split()), j
    head, j = statement('std::cout << "L_pq"')
    row, j = statement("std::cout << L_pq", j)
    csv, j = statement("evaluation_out << L_pq", j)
    out = {"source": "tests/test_search_roargraph.cpp:190,231-236", "header": stream_format(head), "row": stream_format(row),
           "csv_row": stream_format(csv)}
    json.dump(out, open(os.path.join(OUT, "cli_table.json"), "w"), indent=1)


def prune_calls(base, metric, rng, M):
    """Seeded calls of the four rules with pools shaped like the construction's: near neighbours of the pivot (so that occlusion
    happens), a few far rows, ties, repeated ids, the pivot itself, node 0, lists longer and shorter than M."""
    nb = base.shape[0]
    sc = (base @ base.T) if metric == "ip" else -((base[:, None, :2] - base[None, :, :2]) ** 2).sum(-1)      # a cheap notion of "near"
    if metric != "ip":
        g = base @ base.T
        n2 = (base * base).sum(1)
        sc = -(n2[:, None] + n2[None, :] - 2 * g)
    calls = []

    def dists_to(pivot, ids):
        return po.ref_dist(metric, base[np.asarray(ids, np.int64)], np.repeat(base[pivot][None], len(ids), 0))
    for _ in range(40):      # kind 0: a training query's knn row scored against its nearest base point (:1074-1082)
        qv = rng.standard_normal(base.shape[1]).astype(np.float32) * 0.5 + 0.3
        s_q = base @ qv if metric == "ip" else -((base - qv) ** 2).sum(1)
        row = np.argsort(-s_q)[: int(rng.integers(5, 101))].astype(np.uint32)
        pivot = int(row[0])
        if rng.random() < 0.25:      # repeated ids / the pivot again (the rule keeps first occurrences; host only in the product)
            row = np.concatenate([row, row[rng.integers(0, row.size, 3)], [pivot]]).astype(np.uint32)
        calls.append(("get_base", pivot, row, dists_to(pivot, row), None))
    for kind in ("reverse", "reverse_phantoms"):      # a list that grew beyond its limit by one reverse edge (:1391-1432, :1352-1389)
        for _ in range(40):
            src = int(rng.integers(0, nb))
            near = np.argsort(-sc[src])[: 3 * M]
            n = int(rng.integers(2, 2 * M + 2))
            lst = rng.choice(near, size=min(n, near.size), replace=False).astype(np.uint32)
            extra = rng.integers(0, nb, int(rng.integers(0, 4))).astype(np.uint32)
            lst = np.concatenate([lst, extra])
            if rng.random() < 0.3:
                lst = np.concatenate([lst, [0]]).astype(np.uint32)          # node 0 meets the phantoms of :1438
            if rng.random() < 0.2:
                lst = np.concatenate([lst, lst[:2]]).astype(np.uint32)      # repeated ids (std::find drops them)
            if rng.random() < 0.2:
                lst = np.concatenate([[src], lst]).astype(np.uint32)        # the pivot itself in its list
            calls.append((kind, src, rng.permutation(lst).astype(np.uint32), None, None))
    for _ in range(40):      # kind 3: the expansion list of a node's own search (:1192-1220) against its projection list
        node = int(rng.integers(0, nb))
        order = np.argsort(-sc[node])
        order = order[order != node]
        np_ = int(rng.integers(3, 400))
        pool = np.concatenate([order[: np_ // 2], rng.choice(order[np_ // 2: 8 * np_], size=np_ - np_ // 2, replace=False)]).astype(np.uint32)
        pool = rng.permutation(pool).astype(np.uint32)
        ds = dists_to(node, pool)
        srt = pool[np.lexsort((pool, ds))]
        nh = int(rng.integers(0, min(M, srt.size - 2) + 1))
        have = np.concatenate([srt[: nh // 2], rng.integers(0, nb, nh - nh // 2)]).astype(np.uint32)     # the nearest ones are neighbours already
        calls.append(("search", node, pool, ds, have))      # as LinkProjection calls the rule: the node erased from the pool (:1203-1208)
        if rng.random() < 0.3:      # the bare rule with the node still in its pool (its own `start++`, :1858-1860; the product's GPU kernel folds the erase in)
            pool2 = rng.permutation(np.concatenate([pool, [node]])).astype(np.uint32)
            calls.append(("search", node, pool2, dists_to(node, pool2), have))
    return calls


def g6():
    with tempfile.TemporaryDirectory() as td:
        for name, seed in (("ip200", 61), ("l2_512", 62)):
            z = np.load(os.path.join(OUT, "search_%s.npz" % name))
            base, metric = z["base"], str(z["metric"])
            bf = os.path.join(td, "b.fbin")
            io.write_fbin(bf, base)
            rng = np.random.default_rng(seed)
            rec = {"kind": [], "pivot": [], "M": [], "pool_off": [0], "pool_ids": [], "pool_dist_bits": [], "have_off": [0], "have": [], "out_off": [0], "out": []}
            for M in (8, 35):
                calls = prune_calls(base, metric, rng, M)
                res = po.ref_prune(bf, metric, M, calls)
                for (kind, pivot, ids, ds, have), r in zip(calls, res):
                    rec["kind"].append(po.PRUNE_KINDS[kind]); rec["pivot"].append(pivot); rec["M"].append(M)
                    rec["pool_ids"].append(np.asarray(ids, np.uint32))
                    rec["pool_dist_bits"].append((np.asarray(ds, np.float32) if ds is not None else np.zeros(len(ids), np.float32)).view(np.uint32))
                    rec["have"].append(np.asarray(have if have is not None else [], np.uint32))
                    rec["out"].append(np.asarray(r, np.uint32))
                    rec["pool_off"].append(rec["pool_off"][-1] + len(ids)); rec["have_off"].append(rec["have_off"][-1] + rec["have"][-1].size)
                    rec["out_off"].append(rec["out_off"][-1] + len(r))
            np.savez_compressed(os.path.join(OUT, "prune_%s.npz" % name), base_of="search_%s.npz" % name, metric=metric,
                                kind=np.array(rec["kind"], np.uint32), pivot=np.array(rec["pivot"], np.uint32), M=np.array(rec["M"], np.uint32),
                                pool_off=np.array(rec["pool_off"], np.uint64), pool_ids=np.concatenate(rec["pool_ids"]),
                                pool_dist_bits=np.concatenate(rec["pool_dist_bits"]), have_off=np.array(rec["have_off"], np.uint64),
                                have=np.concatenate(rec["have"]), out_off=np.array(rec["out_off"], np.uint64), out=np.concatenate(rec["out"]))


if __name__ == "__main__":
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    if not po.have_ref():
        sys.exit("oracle/_ref/rg_ref is not available (needs /root/reference and an AVX-512 host)")
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1:]
    for name, fn in (("g1", g1), ("g2", g2), ("g3", g3), ("g3c", g3_cosine), ("g4", g4), ("g5", g5), ("g6", g6)):
        if not only or name in only:
            fn()
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("golden fixtures written to", OUT, "(%.2f MB)" % (tot / 1e6))
