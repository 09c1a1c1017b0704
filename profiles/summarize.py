#!/usr/bin/env python3
"""Turn rocprofv3 output directories into the summaries committed here.

    python profiles/summarize.py --trace DIR --fetch DIR --write DIR --tag r01_v4 \
        --cmd "python3 bench.py --steps 3 --warmup 1"

* DIR/**/…_kernel_stats.csv (from `rocprofv3 --kernel-trace --stats`) is copied,
  restricted to our kernels (trx::*), to profiles/<tag>_kernel_stats.csv;
* the FETCH_SIZE / WRITE_SIZE passes (two separate `--pmc` runs, as
  /opt/skills/guides/MI355X_MICROARCH.md prescribes) become
  profiles/<tag>_pmc_traffic.json: per kernel, KB summed over the XCD instances the
  CSV reports, averaged over launches, and HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024
  (the guide's gfx950 correction: FETCH_SIZE reports half the bytes of wide coalesced
  reads -- 128-B requests tallied at 64 B -- while WRITE_SIZE is exact).

Launch averages are given twice: over ALL launches of a kernel (the same population
the kernel trace's AverageNs and bench.py's roofline.avg_launch_ms use -- chunks that the
device-side gate turns into no-ops included) and over ACTIVE launches only (traffic > 1 MB).
"""
import argparse
import collections
import csv
import glob
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
HERE_OUT = HERE


def find(d, suffix, prefix=""):
    hits = sorted(glob.glob(os.path.join(d, "**", prefix + "*" + suffix), recursive=True))
    if not hits:
        raise SystemExit("no *%s under %s" % (suffix, d))
    return hits[0]


def short(name):
    name = name.split("(")[0]
    return name.replace("void ", "").strip()


def counter_by_kernel(d, counter, prefix=""):
    per_dispatch = collections.OrderedDict()
    with open(find(d, "counter_collection.csv", prefix)) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            key = (row["Dispatch_Id"], short(row["Kernel_Name"]))
            per_dispatch[key] = per_dispatch.get(key, 0.0) + float(row["Counter_Value"])
    out = collections.OrderedDict()
    for (_, k), v in per_dispatch.items():
        out.setdefault(k, []).append(v)
    return out


def kernel_source_hash():
    """bench.kernel_source_hash(): bench.py quotes a traffic figure only from a summary whose
    fingerprint matches the kernel sources of the tree it runs from."""
    import hashlib
    h = hashlib.sha1()
    for f in ("trx_walk.hip.h", "trx_kernels.hip.h", "trx_rows.hip.h", "trx_tail.hip.h", "trx_lanes.hip.h"):
        h.update(open(os.path.join(HERE, "..", "transit_amd", "csrc", "hip", f), "rb").read())
    return h.hexdigest()[:12]


def workload_key(src):
    """What bench.py calls this workload (its own n_wn x n_layers x n_lines / layers needed: the geometry and the CIA tables change the last), from the bench line of the
    traced run: bench.py quotes counters only from a summary of the workload it is running."""
    line = os.path.join(src, "bench_line_under_trace.json")
    try:
        c = json.loads(open(line).read().strip().splitlines()[-1])["config"]
        return "%d x %d x %d / %d" % (c["n_wn"], c["n_layers"], c["n_lines"], c["layers_needed"])
    except Exception:
        return None


def trace_calls(d):
    """{kernel: launches} of the TRACE run (bench.py at its default steps: the step plan's own mix of
    kernels, which is what bench.py weights per-launch averages with)."""
    try:
        out = {}
        with open(find(d, "kernel_stats.csv", "trace")) as f:
            for row in csv.DictReader(f):
                if "trx::" in row["Name"]:
                    out[short(row["Name"])] = int(row["Calls"])
        return out
    except (SystemExit, Exception):
        return {}


def counter_pass(d, tag, prefix, names, suffix, what, key):
    """One PMC pass (its own rocprofv3 run) -> profiles/<tag>_<suffix>.json: per kernel and counter the
    average per launch (summed over the XCD instances the CSV reports) and the number of launches."""
    out = collections.OrderedDict()
    for c in names:
        try:
            for k, v in counter_by_kernel(d, c, prefix).items():
                if "trx::" in k:
                    out.setdefault(k, collections.OrderedDict())[c + "_avg_launch"] = sum(v) / len(v)
                    out[k]["launches"] = len(v)
        except SystemExit:
            return
    calls = trace_calls(d)
    for k in out:
        out[k]["trace_calls"] = calls.get(k, 0)
    dst = os.path.join(HERE_OUT, tag + "_" + suffix + ".json")
    json.dump({"source": "rocprofv3 --kernel-trace --pmc " + " ".join(names) + " (own pass)", "what": what,
               "workload_key": key, "kernel_sources": kernel_source_hash(), "kernels": out},
              open(dst, "w"), indent=1)
    print("wrote", dst)


def sq_mix(d, tag, key):
    """SQ instruction counts per launch and kernel (the walk is issue-bound, not HBM-bound)."""
    counter_pass(d, tag, "sq", ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_WAVES"), "sq_mix",
                 "instructions issued per launch", key)
    counter_pass(d, tag, "wait", ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                                  "SQ_BUSY_CYCLES", "SQ_INSTS_VMEM_RD", "GRBM_GUI_ACTIVE"), "sq_wait",
                 "wave-cycle split (SQ_* cycle counters in units of four clocks; GRBM_GUI_ACTIVE summed over the 8 XCDs)", key)
    counter_pass(d, tag, "tcp", ("TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"), "tcp",
                 "L1 tag look-ups, L1 -> L2 read requests, LDS-array cycles", key)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--from", dest="src", help="directory written by profiles/collect.sh (trace/fetch/write/sq CSVs)")
    ap.add_argument("--tag", required=True)
    ap.add_argument("--dest", default=None, help="where the summaries go (default: this directory); collect.sh summarises on the GPU box into gpurun_out/ and drops the raw counter CSVs, which run to tens of MB per pass")
    ap.add_argument("--cmd", default="python3 bench.py --steps 3 --warmup 1")
    ap.add_argument("--workload", default="CH4-demo shape, 1e6 lines, 2501 wn, 100 layers (bench.py defaults)")
    a = ap.parse_args()
    global HERE_OUT
    HERE_OUT = a.dest or HERE
    os.makedirs(HERE_OUT, exist_ok=True)
    if a.src:
        a.trace = a.fetch = a.write = a.src

    if a.trace:
        src = find(a.trace, "kernel_stats.csv", "trace" if a.src else "")
        dst = os.path.join(HERE_OUT, a.tag + "_kernel_stats.csv")
        with open(src) as f, open(dst, "w", newline="") as g:
            r = csv.reader(f)
            w = csv.writer(g, quoting=csv.QUOTE_ALL)
            w.writerow(next(r))
            for row in r:
                if "trx::" in row[0]:
                    row[0] = short(row[0])
                    w.writerow(row)
        print("wrote", dst)

    if a.fetch and a.write:
        fe = counter_by_kernel(a.fetch, "FETCH_SIZE", "fetch" if a.src else "")
        wr = counter_by_kernel(a.write, "WRITE_SIZE", "write" if a.src else "")
        kernels = collections.OrderedDict()
        for k in fe:
            if "trx::" not in k or k not in wr:
                continue
            f, w = fe[k], wr[k]
            fa = [x for x in f if x > 1024.0] or [0.0]
            wa = [x for x in w if x > 1024.0] or [0.0]
            favg, wavg = sum(f) / len(f), sum(w) / len(w)
            kernels[k] = {
                "launches_FETCH_pass": len(f), "launches_WRITE_pass": len(w),
                "FETCH_SIZE_KB_avg_launch": favg, "WRITE_SIZE_KB_avg_launch": wavg,
                "hbm_bytes_per_launch": (2.0 * favg + wavg) * 1024.0,
                "hbm_bytes_per_launch_uncorrected": (favg + wavg) * 1024.0,
                "launches_active_FETCH_pass": len(fa) if fa != [0.0] else 0,
                "FETCH_SIZE_KB_avg_active_launch": sum(fa) / len(fa),
                "WRITE_SIZE_KB_avg_active_launch": sum(wa) / len(wa),
                "hbm_bytes_per_active_launch": (2.0 * sum(fa) / len(fa) + sum(wa) / len(wa)) * 1024.0,
            }
        if a.src:
            calls = trace_calls(a.src)
            for k in kernels:
                kernels[k]["trace_calls"] = calls.get(k, 0)
        doc = {
            "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- " + a.cmd,
            "workload": a.workload, "workload_key": workload_key(a.src) if a.src else None,
            "kernel_sources": kernel_source_hash(),
            "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md (HBM section: on gfx950 "
                          "FETCH_SIZE tallies 128-B requests at 64 B).  The guide calibrates that factor for wide coalesced "
                          "16-B-per-lane streams only; the walk reads 32-byte scalar records and 4-byte gathers, for which it "
                          "is uncalibrated -- hence the uncorrected figure next to it (the truth lies between), and "
                          "Infinity-Cache hits are counted by these counters, not excluded.",
            "kernels": kernels,
        }
        dst = os.path.join(HERE_OUT, a.tag + "_pmc_traffic.json")
        json.dump(doc, open(dst, "w"), indent=1)
        print("wrote", dst)
    if a.src:
        sq_mix(a.src, a.tag, workload_key(a.src))
        line = os.path.join(a.src, "bench_line_under_trace.json")
        if os.path.exists(line):
            import shutil
            shutil.copy(line, os.path.join(HERE_OUT, a.tag + "_bench_line_under_trace.json"))


if __name__ == "__main__":
    main()
