"""CPU: the post-mortem of a GPU memory fault (csrc/rg_mem.hip journal + signal handler, benchlib/fault.py)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REPORT = """rg_mem fault report v1 (SIGABRT)
now_us 9000000
journal 6 events (the last 4096 are kept): t_us kind va bytes device aux
J 1000 g 0x7000000000 1073741824 0 0
J 2000 u 0x7000000000 1073741824 0 0
J 2500 B 0x7100000000 3221225472 0 0
J 3000 C 0x7100000000 3221225472 0 0
J 4000 H 0x7100000000 3221225472 0 0
J 8000000 F 0x7100000000 3221225472 0 1
LIVE 0 0x7300000000 2147483648
CACHED 0 0x7400000000 2147483648
MAPS
7100000000-71c0000000 ---p 00000000 00:00 0
7300000000-7380000000 rw-s 00000000 00:06 123                    /dev/dri/renderD128
END
"""


def test_attribution_of_a_fault_address():
    sys.path.insert(0, ROOT)
    from benchlib import fault
    err = "Memory access fault by GPU node-2 (Agent handle: 0x5c2ce1ad5490) on address 0x710004c000. Reason: Unknown.\n"
    assert fault.fault_addresses(err) == [0x710004c000]
    r = fault.attribute(REPORT, 0x710004c000)
    assert r["state"].startswith("UNMAPPED by the library 1.000 s before the report"), r
    assert [h["kind"] for h in r["history"]] == ["B", "C", "H", "F"] and r["range"][2] == 0x4c000
    assert "---p" in r["maps_line"]
    assert fault.attribute(REPORT, 0x7300000010)["state"] == "LIVE balanced buffer"
    assert fault.attribute(REPORT, 0x7400000010)["state"].startswith("CACHED")
    assert fault.attribute(REPORT, 0x7000000010)["state"].startswith("UNMAPPED")
    r = fault.attribute(REPORT, 0x10)
    assert r["state"].startswith("not a range") and r["maps_line"].startswith("no mapping")
    assert any("stale pointer" in l for l in fault.describe(REPORT, err))


def test_the_library_writes_its_report_when_the_process_aborts(tmp_path):
    """RG_FAULT_REPORT=<path>: SIGABRT (what the HIP runtime raises after a GPU page fault) leaves the journal and /proc/self/maps
    behind, and the process still dies of the signal (faulthandler, installed later, is passed through)."""
    rep = tmp_path / "report.txt"
    code = ("import sys, os; sys.path.insert(0, %r)\n"
            "import faulthandler; faulthandler.enable()\n"
            "from roargraph_amd._lib import lib; lib()\n"
            "os.abort()\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RG_FAULT_REPORT=str(rep)), capture_output=True, text=True, timeout=120)
    assert p.returncode in (-6, 134), (p.returncode, p.stderr[-500:])
    text = rep.read_text()
    assert text.startswith("rg_mem fault report v1 (SIGABRT)") and "MAPS" in text and text.rstrip().endswith("END")
    from benchlib import fault
    assert len(fault.parse(text)["maps"]) > 10
