"""The four occlusion-pruning rules of the construction (src/index_bipartite.cpp:1434-1694, 1846-1940) against goldens made with the
reference's OWN Distance::compare, Neighbor::operator< (under std::sort) and operator== (under std::find): tests/golden/prune_*.npz,
scripts/make_golden.py g6, oracle/ref_driver.cpp `prune` (the TU itself cannot be compiled in this image, so the rules are restated
around the genuine objects -- the standing `rg_ref search` has for SearchRoarGraph).  330-odd calls per base: knn rows, lists that grew by a
reverse edge (with and without the phantom entries of :1438), expansion lists against a projection list; repeated ids, the pivot in its
own pool, node 0, ties.  Checked here: the oracle's rules (CPU), the product's host rules (CPU: the builder is host code), the product's
pruning kernel (-m gpu), and -- where /root/reference is present -- the goldens regenerated live.  What stays UNPINNED: the order in
which LinkProjection's phases apply the rules."""
import os

import numpy as np
import pytest

from oracle import pyoracle as po

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KIND_NAMES = {v: k for k, v in po.PRUNE_KINDS.items()}


def calls_of(name):
    z = np.load(os.path.join(GOLD, "prune_%s.npz" % name))
    base = np.load(os.path.join(GOLD, str(z["base_of"])))["base"]
    po_, ho, oo = (z[k].astype(np.int64) for k in ("pool_off", "have_off", "out_off"))
    out = []
    for i in range(z["kind"].size):
        out.append(dict(kind=int(z["kind"][i]), pivot=int(z["pivot"][i]), M=int(z["M"][i]), ids=z["pool_ids"][po_[i]:po_[i + 1]],
                        dists=z["pool_dist_bits"][po_[i]:po_[i + 1]].view(np.float32), have=z["have"][ho[i]:ho[i + 1]], want=z["out"][oo[i]:oo[i + 1]]))
    return base, str(z["metric"]), out


@pytest.mark.parametrize("name", ["ip200", "l2_512"])
def test_oracle_rules_equal_the_goldens(name):
    base, metric, calls = calls_of(name)
    assert len(calls) >= 320 and {c["kind"] for c in calls} == {0, 1, 2, 3}
    for i, c in enumerate(calls):
        got = po.prune(base, metric, c["M"], KIND_NAMES[c["kind"]], c["pivot"], c["ids"], c["dists"], c["have"])
        assert got.tolist() == c["want"].tolist(), (name, i, KIND_NAMES[c["kind"]])


@pytest.mark.parametrize("name", ["ip200", "l2_512"])
def test_product_host_rules_equal_the_goldens(name):
    """Builder::prune_get_base / prune_reverse / prune_search of roargraph_amd/csrc/rg_build.cpp (they skip the iterations of the second
    sweeps that cannot change a list: the lists must be the reference's all the same)."""
    from roargraph_amd import build
    base, metric, calls = calls_of(name)
    for i, c in enumerate(calls):
        got = build.prune_debug(base, metric, c["M"], c["kind"], c["pivot"], c["ids"], c["dists"], c["have"])
        assert got.tolist() == c["want"].tolist(), (name, i, KIND_NAMES[c["kind"]])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ip200", "l2_512"])
def test_product_gpu_pruning_kernel_equals_the_goldens(name):
    """rg_prune_search_kernel as the GPU-assisted build launches it: kind 3 from the expansion list, kind 0 from the knn row (distances
    computed on the device).  Calls the kernel leaves to the host by design (a pool that names an id twice) are the host rule's."""
    from roargraph_amd import build
    from roargraph_amd._lib import RgError
    base, metric, calls = calls_of(name)
    ran = left = 0
    for i, c in enumerate(calls):
        if c["kind"] not in (0, 3):
            continue
        dup = len(set(c["ids"].tolist())) != c["ids"].size
        if c["kind"] == 0 and (dup or c["ids"][0] != c["pivot"]):
            continue
        if c["kind"] == 3 and c["pivot"] in c["ids"].tolist():
            continue      # the bare rule with the node in its pool: LinkProjection erases the node first (:1203-1208), and the kernel folds that erase in
        try:
            got = build.prune_debug(base, metric, c["M"], c["kind"], c["pivot"], c["ids"], c["dists"], c["have"], use_gpu=True)
        except RgError as e:
            assert "left this list to the host" in str(e), (name, i, str(e))
            left += 1
            continue
        ran += 1
        assert got.tolist() == c["want"].tolist(), (name, i, KIND_NAMES[c["kind"]])
    assert ran >= 120 and left <= 10, (ran, left)


@pytest.mark.skipif(not po.have_ref() or not os.path.isdir("/root/reference"), reason="needs oracle/_ref/rg_ref and the reference tree")
def test_goldens_regenerate_bit_for_bit(tmp_path):
    """the committed lists are what the reference-header driver returns today for the committed pools"""
    from roargraph_amd import io
    for name in ("ip200", "l2_512"):
        base, metric, calls = calls_of(name)
        bf = str(tmp_path / "b.fbin")
        io.write_fbin(bf, base)
        for M in (8, 35):
            sub = [c for c in calls if c["M"] == M]
            res = po.ref_prune(bf, metric, M, [(KIND_NAMES[c["kind"]], c["pivot"], c["ids"], c["dists"], c["have"]) for c in sub])
            for c, r in zip(sub, res):
                assert r.tolist() == c["want"].tolist()
            # the pool distances themselves are the reference's compare() of (row, pivot row)
            for c in sub[:40]:
                if c["kind"] in (0, 3):
                    want = po.ref_dist(metric, base[c["ids"].astype(np.int64)], np.repeat(base[c["pivot"]][None], c["ids"].size, 0))
                    assert (want.view(np.uint32) == c["dists"].view(np.uint32)).all()
