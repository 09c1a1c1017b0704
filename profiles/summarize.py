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


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))
    if not hits:
        raise SystemExit("no *%s under %s" % (suffix, d))
    return hits[0]


def short(name):
    name = name.split("(")[0]
    return name.replace("void ", "").strip()


def counter_by_kernel(d, counter):
    per_dispatch = collections.OrderedDict()
    with open(find(d, "_counter_collection.csv")) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            key = (row["Dispatch_Id"], short(row["Kernel_Name"]))
            per_dispatch[key] = per_dispatch.get(key, 0.0) + float(row["Counter_Value"])
    out = collections.OrderedDict()
    for (_, k), v in per_dispatch.items():
        out.setdefault(k, []).append(v)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--tag", required=True)
    ap.add_argument("--cmd", default="python3 bench.py --steps 3 --warmup 1")
    ap.add_argument("--workload", default="CH4-demo shape, 1e6 lines, 2501 wn, 100 layers, chunk 12")
    a = ap.parse_args()

    if a.trace:
        src = find(a.trace, "_kernel_stats.csv")
        dst = os.path.join(HERE, a.tag + "_kernel_stats.csv")
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
        fe, wr = counter_by_kernel(a.fetch, "FETCH_SIZE"), counter_by_kernel(a.write, "WRITE_SIZE")
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
                "launches_active_FETCH_pass": len(fa) if fa != [0.0] else 0,
                "FETCH_SIZE_KB_avg_active_launch": sum(fa) / len(fa),
                "WRITE_SIZE_KB_avg_active_launch": sum(wa) / len(wa),
                "hbm_bytes_per_active_launch": (2.0 * sum(fa) / len(fa) + sum(wa) / len(wa)) * 1024.0,
            }
        doc = {
            "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- " + a.cmd,
            "workload": a.workload,
            "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md (HBM section)",
            "kernels": kernels,
        }
        dst = os.path.join(HERE, a.tag + "_pmc_traffic.json")
        json.dump(doc, open(dst, "w"), indent=1)
        print("wrote", dst)


if __name__ == "__main__":
    main()
