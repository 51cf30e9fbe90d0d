"""Post-mortem of a `Memory access fault by GPU`: names what a fault address belonged to.

The HIP runtime answers a GPU page fault with one line on stderr and abort().  With RG_FAULT_REPORT=<path> (or
rg_mem_fault_report) librg_hip.so writes, in its SIGABRT handler, its journal of address-space events, its live / cached
buffers and /proc/self/maps to <path> (csrc/rg_mem.hip).  `attribute(report_text, address)` reads that file.
"""
import re

KINDS = {"g": "granule mapped at its pool address", "u": "granule unmapped at its pool address", "B": "balanced buffer mapped",
         "C": "buffer freed into the cache (still mapped)", "H": "buffer handed out again from the cache", "F": "buffer unmapped",
         "P": "plain hipMalloc of a large request", "f": "hipFree of a pointer the pools do not know", "A": "arena of address space reserved (no memory)"}

FAULT_RE = re.compile(r"Memory access fault by GPU.*?on address (0x[0-9a-fA-F]+)")


def fault_addresses(stderr_text):
    """The addresses the runtime named on stderr (usually one)."""
    return [int(m.group(1), 16) for m in FAULT_RE.finditer(stderr_text or "")]


def parse(report_text):
    out = {"why": None, "now_us": None, "journal": [], "live": [], "cached": [], "reps": [], "spare": [], "maps": []}
    in_maps = False
    for line in report_text.splitlines():
        if in_maps:
            if line == "END":
                break
            m = re.match(r"([0-9a-f]+)-([0-9a-f]+)\s+(\S+)\s+\S+\s+\S+\s+\S+\s*(.*)", line)
            if m:
                out["maps"].append((int(m.group(1), 16), int(m.group(2), 16), m.group(3), m.group(4)))
            continue
        f = line.split()
        if not f:
            continue
        if f[0] == "rg_mem" and "(" in line:
            out["why"] = line[line.index("(") + 1:line.rindex(")")]
        elif f[0] == "now_us":
            out["now_us"] = int(f[1])
        elif f[0] == "J" and len(f) >= 7:
            out["journal"].append({"t_us": int(f[1]), "kind": f[2], "va": int(f[3], 16), "bytes": int(f[4]), "device": int(f[5]), "aux": int(f[6])})
        elif f[0] in ("LIVE", "CACHED", "REP", "SPARE") and len(f) >= 4:
            out[{"LIVE": "live", "CACHED": "cached", "REP": "reps", "SPARE": "spare"}[f[0]]].append((int(f[2], 16), int(f[3])))
        elif f[0] == "MAPS":
            in_maps = True
    return out


def attribute(report_text, address):
    """What `address` was when the process died: a dict with `state` (live / cached / pool / unmapped-by-the-library / not-ours),
    the journal events of the range that holds it (oldest first) and the /proc/self/maps line that covers it."""
    rep = parse(report_text)
    res = {"address": hex(address), "state": "not a range of the library's allocator", "range": None, "history": [], "maps_line": None}
    for name, label in (("live", "LIVE balanced buffer"), ("cached", "CACHED (freed, still mapped) balanced buffer"),
                        ("reps", "pool: class representative granule"), ("spare", "pool: spare granule")):
        for va, nbytes in rep[name]:
            if va <= address < va + nbytes:
                res["state"] = label
                res["range"] = (hex(va), nbytes, address - va)
    for e in rep["journal"]:
        if e["kind"] == "A":
            continue
        span = e["bytes"] if e["bytes"] else 1
        if e["va"] <= address < e["va"] + span:
            res["history"].append({"t_us": e["t_us"], "kind": e["kind"], "what": KINDS.get(e["kind"], "?"), "va": hex(e["va"]), "bytes": e["bytes"],
                                   "offset": address - e["va"], "aux": e["aux"]})
    if res["range"] is None and res["history"]:
        last = res["history"][-1]
        if last["kind"] in ("F", "u"):
            res["state"] = "UNMAPPED by the library %.3f s before the report (%s): a stale pointer" % (
                ((rep["now_us"] or last["t_us"]) - last["t_us"]) / 1e6, last["what"])
        else:
            res["state"] = "a range of the library (last event: %s)" % last["what"]
        res["range"] = (last["va"], last["bytes"], last["offset"])
    for lo, hi, perms, what in rep["maps"]:
        if lo <= address < hi:
            res["maps_line"] = "%x-%x %s %s (offset %d)" % (lo, hi, perms, what, address - lo)
    if res["maps_line"] is None:
        res["maps_line"] = "no mapping of the process covers the address"
    return res


def describe(report_text, stderr_text):
    """Human-readable lines for every fault address found on stderr."""
    lines = []
    for a in fault_addresses(stderr_text):
        r = attribute(report_text, a)
        lines.append("fault address %s: %s" % (r["address"], r["state"]))
        if r["range"]:
            lines.append("  range %s, %d bytes, offset %d" % r["range"])
        for h in r["history"][-6:]:
            lines.append("  t=%.3f s %s %s" % (h["t_us"] / 1e6, h["kind"], h["what"]))
        lines.append("  /proc/self/maps: %s" % r["maps_line"])
    return lines
