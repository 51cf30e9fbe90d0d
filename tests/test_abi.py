"""CPU: the C-ABI library loads, exports every symbol include/rg.h declares, its host-side format logic matches the
goldens / oracle, and compute entry points FAIL LOUDLY without a GPU (no CPU fallback exists)."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

from helpers import bits

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def rg():
    if not os.path.exists(os.path.join(ROOT, "roargraph_amd", "librg_hip.so")):
        import __graft_entry__
        __graft_entry__.build()
    from roargraph_amd import index
    return index


def header_symbols():
    src = open(os.path.join(ROOT, "include", "rg.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rg_[a-z0-9_]+)\s*\(", src)))


def test_exports_every_declared_symbol(rg):
    from roargraph_amd._lib import SYMBOLS, lib
    names = header_symbols()
    assert len(names) >= 25
    assert sorted(SYMBOLS) == names, "roargraph_amd/_lib.py SYMBOLS out of sync with include/rg.h"
    L = lib()
    for n in names:
        assert hasattr(L, n), "librg_hip.so does not export %s" % n
    assert b"gfx950" in L.rg_version()


def test_header_cites_reference_lines():
    src = open(os.path.join(ROOT, "include", "rg.h")).read()
    for needle in ("distance.h:18", "index_bipartite.h:100-101", "util.h:106-127", "index_bipartite.cpp:2097-2117",
                   "README.md:62-75", "test_search_roargraph.cpp:23-36"):
        assert needle in src


def test_format_rules_match_reference_goldens(rg, tmp_path):
    from roargraph_amd._lib import RgError
    z = np.load(os.path.join(GOLD, "formats.npz"))
    p = str(tmp_path / "f.bin")
    for key in z.files:
        if not key.endswith("_says"):
            continue
        case = key[:-5]
        open(p, "wb").write(z[case].tobytes())
        said = str(z[key])
        fn = rg.gt_meta if case.startswith("gt_") else rg.fbin_meta
        if said.startswith("OK"):
            assert list(fn(p)) == [int(x) for x in said.split()[1:]], case
        else:
            with pytest.raises(RgError, match="Data file size wrong!"):
                fn(p)
    open(p, "wb").write(z["gt_good"].tobytes())
    ids, ds = rg.gt_load(p)
    assert (ids == z["gt_good_ids"]).all() and (bits(ds) == z["gt_good_dist_bits"]).all()
    assert (rg.knn_ids_load(p) == ids).all()
    open(p, "wb").write(z["fbin_good"].tobytes())
    arr, d = rg.fbin_load(p)
    assert d == 24 and (bits(arr) == z["fbin_good_loaded_bits"]).all()
    with pytest.raises(RgError, match="open file error"):
        rg.fbin_meta(str(tmp_path / "missing.fbin"))


def test_fbin_padding_to_multiple_of_8(rg, oracle, tmp_path):
    from roargraph_amd import io
    data = np.random.default_rng(0).standard_normal((9, 13)).astype(np.float32)
    p = str(tmp_path / "odd.fbin")
    io.write_fbin(p, data)
    arr, d = rg.fbin_load(p)
    assert d == 13 and arr.shape == (9, 16)
    assert (arr[:, :13] == data).all() and (arr[:, 13:] == 0).all()
    o, od = oracle.fbin_load(p)
    assert od == 13 and (bits(o) == bits(arr)).all()


def test_graph_and_gt_roundtrip(rg, oracle, tmp_path):
    from roargraph_amd import io
    rng = np.random.default_rng(1)
    lists = [rng.integers(0, 40, rng.integers(0, 9)).astype(np.uint32) for _ in range(40)]
    lists[3] = np.zeros(0, np.uint32)
    off, nbrs = io.lists_to_csr(lists)
    p = str(tmp_path / "g.index")
    rg.graph_save(p, off, nbrs, 17)
    for loader in (rg.graph_load, oracle.index_load, io.read_index):
        o2, n2, ep = loader(p)
        assert ep == 17 and (np.asarray(o2) == off).all() and (np.asarray(n2) == nbrs).all()
    q = str(tmp_path / "g2.index")
    oracle.index_save(q, off, nbrs, 17)
    assert open(p, "rb").read() == open(q, "rb").read()
    io.write_index(q, off, nbrs, 17)
    assert open(p, "rb").read() == open(q, "rb").read()
    # truncated index
    open(q, "wb").write(open(p, "rb").read()[:-6])
    from roargraph_amd._lib import RgError
    with pytest.raises(RgError, match="truncated"):
        rg.graph_load(q)
    ids = rng.integers(0, 99, (6, 4)).astype(np.uint32)
    ds = rng.standard_normal((6, 4)).astype(np.float32)
    g = str(tmp_path / "gt.bin")
    rg.gt_save(g, ids, ds)
    assert oracle.gt_meta(g) == (6, 4)
    i2, d2 = oracle.gt_load(g)
    assert (i2 == ids).all() and (bits(d2) == bits(ds)).all()


def test_recall_matches_oracle(rg, oracle):
    rng = np.random.default_rng(2)
    res = rng.integers(0, 50, (30, 10)).astype(np.uint32)
    gt = rng.integers(0, 50, (30, 25)).astype(np.uint32)
    assert rg.recall(res, gt, 10) == oracle.recall(res, gt, 10)
    # a result matrix wider than k (top-100 searches scored at recall@10): the first k ids of every row count
    wide = np.concatenate([res, rng.integers(0, 50, (30, 90)).astype(np.uint32)], axis=1)
    assert rg.recall(wide, gt, 10) == oracle.recall(res, gt, 10)


@pytest.mark.parametrize("nd,d,stride", [(1, 8, 8), (300, 200, 200), (4097, 200, 208), (700, 24, 24)])
def test_projection_ep_host_loop_matches_oracle(rg, oracle, nd, d, stride):
    """rg_projection_ep (host loop, no device needed) == the oracle's CalculateProjectionep, ties included (every row
    twice: the first index wins, index_bipartite.cpp:2031-2035)."""
    from roargraph_amd._lib import check, lib
    rng = np.random.default_rng(nd + d)
    base = np.zeros((nd, stride), np.float32)
    base[:, :d] = (rng.standard_normal((nd, d)) * 3 + 50.0).astype(np.float32)
    if nd > 10:
        base[nd // 2:] = base[: nd - nd // 2]
    got = C.c_uint32()
    check(lib().rg_projection_ep(base.ctypes.data_as(C.c_void_p), C.c_uint32(nd), C.c_uint32(d), C.c_uint32(stride), C.byref(got)))
    assert got.value == oracle.projection_ep(base, dim=d)


def test_normalize_matches_oracle(rg, oracle):
    from roargraph_amd._lib import lib
    a = np.random.default_rng(3).standard_normal((20, 24)).astype(np.float32)
    b = a.copy()
    lib().rg_normalize_rows(a.ctypes.data_as(C.c_void_p), C.c_size_t(20), C.c_size_t(24), C.c_uint32(24))
    oracle.lib().rgo_normalize_rows(b.ctypes.data_as(C.c_void_p), C.c_size_t(20), C.c_size_t(24), C.c_uint(24))
    assert (bits(a) == bits(b)).all()


def test_compute_fails_loudly_without_gpu(rg):
    """On a CPU-only host every compute entry point must return RG_ERR_DEVICE -- never silently compute elsewhere."""
    from roargraph_amd._lib import RG_ERR_DEVICE, RgError, lib
    if lib().rg_device_count() > 0:
        pytest.skip("a GPU is visible here")
    base = np.zeros((10, 8), np.float32)
    with pytest.raises(RgError) as e:
        rg.IndexBipartite.from_arrays(base, np.zeros(11, np.uint64), np.zeros(0, np.uint32), 0)
    assert e.value.code == RG_ERR_DEVICE and "no CPU fallback" in str(e.value)
    from roargraph_amd import groundtruth
    with pytest.raises(RgError) as e:
        groundtruth.compute_groundtruth(base, base, "ip", 2)
    assert e.value.code == RG_ERR_DEVICE


def test_product_never_touches_the_oracle():
    """The shipped path (roargraph_amd/, bench hot path aside) must not import, link or name anything under oracle/."""
    pkg = os.path.join(ROOT, "roargraph_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "pyoracle" not in txt and "rg_oracle" not in txt and "librg_oracle" not in txt, f
    import subprocess
    so = os.path.join(pkg, "librg_hip.so")
    if os.path.exists(so):
        out = subprocess.run(["ldd", so], capture_output=True, text=True).stdout
        assert "oracle" not in out


def test_cli_table_golden_is_current_and_the_twin_prints_it():
    """tests/golden/cli_table.json holds the reference CLI's stdout header / row / CSV row formats (stream statements of
    tests/test_search_roargraph.cpp:190,231-236 evaluated by scripts/make_golden.py g5).  Where the reference tree is
    present (this container) the golden is regenerated and must be unchanged; everywhere, the twin's source must print the
    header's pieces in the reference's order with nothing added by default (the seventh column sits behind --steady)."""
    import json
    import re
    golden = os.path.join(ROOT, "tests", "golden", "cli_table.json")
    fmt = json.load(open(golden))
    if os.path.exists("/root/reference/tests/test_search_roargraph.cpp"):
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import importlib
        mg = importlib.import_module("make_golden")
        import tempfile
        keep = mg.OUT
        with tempfile.TemporaryDirectory() as td:
            mg.OUT = td
            try:
                mg.g5()
            finally:
                mg.OUT = keep
            assert json.load(open(os.path.join(td, "cli_table.json"))) == fmt, "golden is stale: run scripts/make_golden.py g5"
    assert fmt["header"].count("\t") == 8 and fmt["csv_row"].count(",") == 5
    src = open(os.path.join(ROOT, "roargraph_amd", "cli", "test_search_roargraph.cpp")).read()
    stmt = src[src.index('std::cout << "L_pq"'):]
    stmt = stmt[:stmt.index(";")]
    lits = "".join(x.encode().decode("unicode_escape") if x is not None else "{k}"
                   for x in (m.group(1) if m.group(1) is not None else None
                             for m in re.finditer(r'<<\s*(?:"((?:[^"\\]|\\.)*)"|k\b)', stmt)))
    assert lits == fmt["header"], (lits, fmt["header"])


def test_bench_child_that_dies_fails_the_bench(tmp_path, monkeypatch, capsys):
    """bench.py's one-GPU runs work in a child process.  The parent's rule (round 6: NO second attempt -- a bench that re-runs itself after
    a GPU fault hides a memory bug of the product), on stand-in children: a child killed by a signal ends the bench with a non-zero status,
    no line on stdout, and the post-mortem of the library's fault report on stderr; an error the child reports itself keeps its status;
    a child that prints its line passes it on with `bench_attempts: 1`."""
    import json
    import sys
    import textwrap
    sys.path.insert(0, ROOT)
    import bench

    def run(body):
        fake = tmp_path / "fake_bench.py"
        fake.write_text(textwrap.dedent(body))
        monkeypatch.setattr(bench, "__file__", str(fake))
        monkeypatch.setattr(bench, "ROOT", str(tmp_path))
        monkeypatch.setattr(sys, "argv", [str(fake)])
        monkeypatch.delenv("RG_BENCH_CHILD", raising=False)
        monkeypatch.delenv("RG_FAULT_REPORT", raising=False)
        with pytest.raises(SystemExit) as e:
            bench.supervise()
        out = capsys.readouterr()
        return e.value.code, out.out, out.err

    # a child that dies the way the HIP runtime kills a process after a GPU page fault: the message, the library's report, abort()
    code, out, err = run("""
        import os, sys
        sys.path.insert(0, %r)
        from roargraph_amd._lib import lib
        lib()                                      # RG_FAULT_REPORT (set by the parent) installs the report writer
        a = id(sys)                                # an address of this process: the post-mortem finds it in /proc/self/maps
        print("Memory access fault by GPU node-2 (Agent handle: 0x1) on address 0x%%x. Reason: Unknown." %% a, file=sys.stderr, flush=True)
        os.abort()
    """ % ROOT)
    assert code != 0 and out == "", (code, out, err)
    assert err.count("ended with status") == 1 and "no second attempt" in err and "fault address 0x" in err and "/proc/self/maps:" in err, err
    assert (tmp_path / "bench_fault_report.txt").read_text().startswith("rg_mem fault report v1 (SIGABRT)")
    code, out, err = run("""
        import sys
        print("bench.py needs an MI355X", file=sys.stderr); sys.exit(3)
    """)
    assert code == 3 and out == "" and "post-mortem" not in err
    code, out, err = run("""
        print('{"metric": "x", "value": 2}')
    """)
    assert code == 0 and json.loads(out.strip()) == {"metric": "x", "value": 2, "bench_attempts": 1}
