#!/usr/bin/env python3
"""Headline benchmark: the CH4-demo-shaped emission spectrum (BASELINE.json
configs[1]): 2500-5000 cm-1 at 1 cm-1 (2501 points, wnosamp 2160), 100 layers,
10^6 synthetic CH4 lines, 60x60 Voigt grid, H2-H2 CIA, eclipse geometry with
five angles -- one "step" = one full spectrum (trx_run: CIA + line sweep +
optical depth + intensities + flux) with the line list, grids and Voigt table
already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  metric = wavenumber-points x layers / s where
"layers" is the number of layers the reference's lazy sweep needs for this
input (deepest toomuch crossing), as in BASELINE.md.
"""
from __future__ import annotations

import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_VALU_TFLOPS = 78.6        # MI355X vector fp64 (spec; half the guide's 157.3 TFLOPS vector fp32)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--lines", type=int, default=1_000_000)
    ap.add_argument("--layers", type=int, default=100)
    ap.add_argument("--wnlow", type=float, default=2500.0)
    ap.add_argument("--wnhigh", type=float, default=5000.0)
    ap.add_argument("--wnosamp", type=int, default=2160)
    ap.add_argument("--wndelt", type=float, default=1.0)
    ap.add_argument("--layer-chunk", type=int, default=0)
    ap.add_argument("--solution", choices=("eclipse", "transit"), default="eclipse")
    ap.add_argument("--ncia", type=int, default=1)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="N>1: weak = every GPU gets its own CH4-demo-sized slice (band and line list grow with N); "
                         "strong = the one CH4-demo run split N ways")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse", action="store_true",
                    help="N>1 on FEWER GPUs (ranks share devices): gloo instead of RCCL -- the engine's "
                         "exchanges go over the host transport and the gather through host memory. "
                         "Exercises the multi-rank code path; its timing means nothing")
    ap.add_argument("--comm-single", action="store_true",
                    help="N=1 only: run through a 1-rank RCCL communicator (the code path every rank of an N>1 job takes)")
    ap.add_argument("--cpu-lines", type=int, default=0, help="lines of the CPU-baseline sample (0 = full workload)")
    return ap.parse_args()


def measured_traffic(kernel):
    """HBM bytes per launch (all launches, like avg_launch_ms) of `kernel` from the committed rocprofv3 PMC
    passes (profiles/*pmc_traffic*.json; FETCH_SIZE/WRITE_SIZE collected in separate
    --pmc runs and corrected as MI355X_MICROARCH.md prescribes).  None when no
    profile of the current kernels is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        k = d["kernels"]["trx::" + kernel]
        return float(k.get("hbm_bytes_per_launch", k.get("hbm_bytes_per_active_launch")))
    except Exception:
        return None


def make_workload(args, tag, nlines, verb=2, wnhigh=None, unique=True):
    from transit_amd import synth
    d = os.path.join(tempfile.gettempdir(), "transit_bench_%s_%d_%d_%s" % (
        tag, nlines, args.layers, os.getpid() if unique else os.environ.get("MASTER_PORT", "0")))
    synth.make_case(d, nlines=nlines, wnlow=args.wnlow, wnhigh=wnhigh or args.wnhigh, wndelt=args.wndelt,
                    wnosamp=args.wnosamp,
                    nlayers=args.layers, solution=args.solution, toomuch=10.0, ethresh=1e-50, nwidth=20.0,
                    raygrid="0 20 40 60 80", ncia=args.ncia, seed=1234, extra={"verb": verb})
    return d


def cpu_baseline(args, gpu_spectrum_full):
    """Time the CPU path on this box's host cores (one thread: the reference is
    single-threaded).  Prefers the real reference binary (oracle/_ref/transit,
    built from the reference sources by oracle/Makefile); falls back to our C
    restatement.  Test infrastructure, used here only as the baseline/checker."""
    nlines = args.cpu_lines or args.lines
    d = make_workload(args, "cpu", nlines, verb=4)     # verb 4 prints the reference's stage timers
    ref = os.path.join(ROOT, "oracle", "_ref", "transit")
    out = {"cores": 1, "unit": "wavenumber-points*layers/s"}
    if os.path.exists(ref) and os.access(ref, os.X_OK):
        t0 = time.time()
        p = subprocess.run([ref, "-c", "case.cfg"], cwd=d, capture_output=True, text=True)
        wall = time.time() - t0
        if p.returncode == 0:
            stage = {}
            for m in re.finditer(r"Check point: \d+ - (\d+) ([^:]+):\s+dt = ([0-9.]+) sec", p.stdout):
                stage.setdefault(m.group(2).strip(), 0.0)
                stage[m.group(2).strip()] += float(m.group(3))
            t_path = sum(v for k, v in stage.items() if k in ("makeipsample", "interpcs", "idxrefrac", "extwn",
                                                              "tau eclipse", "emergent intensity", "flux"))
            t_table = stage.get("opacity", 0.0)
            tm = np.loadtxt(os.path.join(d, "toomuch.dat"), comments="#", skiprows=2)
            layers = int(tm[:, 3].max()) + 1
            spec = np.loadtxt(os.path.join(d, "spectrum.dat"), comments="#")[:, 1]
            out.update(kind="reference", value=len(spec) * layers / t_path, seconds=t_path,
                       seconds_voigt_table=t_table, seconds_wall=wall, layers=layers,
                       sample="full workload: %d lines x %d layers, spectrum path = reference stages "
                              "interpcs..flux (its own timers), Voigt table %.2f s excluded" % (nlines, args.layers, t_table))
            if gpu_spectrum_full is not None and nlines == args.lines and len(spec) == len(gpu_spectrum_full):
                out["gpu_vs_reference_max_rel"] = float(np.max(np.abs(gpu_spectrum_full / spec - 1)))
            return out
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    from transit_amd.host import Problem
    nl = min(nlines, 200_000)
    d = make_workload(args, "cpuport", nl)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    eng = ol.OracleEngine(P.static)
    t0 = time.time()
    r = eng.run(P.atm, P.opts, debug=True)
    t = time.time() - t0
    layers = int(r["last"].max()) + 1
    out.update(kind="port", value=P.nwn * layers / t, seconds=t, layers=layers,
               sample="%d of %d lines x %d layers (CPU cost is linear in lines)" % (nl, args.lines, args.layers))
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse:
            local = local % max(torch.cuda.device_count(), 1)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    dev = torch.device("cuda", local)
    cdev = torch.device("cpu") if (world > 1 and args.rehearse) else dev       # where collectives run

    from transit_amd.host import Problem
    from transit_amd.engine import Engine
    from transit_amd.shard import shard_bounds
    from transit_amd import dist as tdist

    # N=1: the CH4-demo run.  N>1, weak: the band and the line list grow N-fold at the
    # same resolution and line density, so every GPU owns one CH4-demo-sized slice of one
    # spectrum (2501 bins, ~1e6 lines + halo).  N>1, strong: the N=1 run split N ways.
    grow = world if args.scaling == "weak" else 1
    wnhigh = args.wnlow + grow * (args.wnhigh - args.wnlow)
    if world > 1:       # one copy of the input files, written by rank 0
        if rank == 0:
            d = make_workload(args, "w%d" % world, grow * args.lines, wnhigh=wnhigh, unique=False)
        dist.barrier()
        if rank != 0:
            d = os.path.join(tempfile.gettempdir(), "transit_bench_w%d_%d_%d_%s" % (
                world, grow * args.lines, args.layers, os.environ.get("MASTER_PORT", "0")))
    else:
        d = make_workload(args, "r0", args.lines)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    nwn, nlayer = P.nwn, P.nlayer
    lo, hi = shard_bounds(nwn, world, rank)
    P.set_shard(lo, hi)
    st = P.static
    st.device = local
    comm = None
    if (world > 1 and not args.rehearse) or args.comm_single:   # RCCL communicator of the handle: its one gather
        comm = tdist.create_comm(world, rank, local)
        st.comm, st.nranks, st.rank = comm, world, rank
    t0 = time.time()
    eng = Engine(st)
    t_create = time.time() - t0
    opts = P.opts
    opts.layer_chunk = args.layer_chunk
    opts.profile = 0

    mpad = tdist.padded_len(nwn, world)
    spec_local = torch.zeros(mpad, dtype=torch.float64, device=dev)      # slice + padding to equal counts
    gathered = torch.zeros(mpad * world, dtype=torch.float64, device=cdev) if world > 1 else None

    def step():
        eng.run_device(P.atm, opts, spec_local.data_ptr())
        if world > 1:       # the single exchange of the path
            if comm is not None:      # trx_gather: ncclAllGather on the handle's stream (C ABI)
                eng.gather(spec_local.data_ptr(), gathered.data_ptr(), mpad)
            else:                     # rehearsal on a shared GPU: gloo through host memory
                dist.all_gather_into_tensor(gathered, spec_local.cpu())

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_step = 1e3 * elapsed / args.steps

    # one profiled run for the per-kernel event timings and the counters
    opts.profile = 1
    r = eng.run(P.atm, opts, debug=("last",))
    opts.profile = 0
    stats = eng.stats()
    layers_needed = int(r["last"].max()) + 1
    if world > 1:
        t = torch.tensor([layers_needed], dtype=torch.int64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        layers_needed = int(t.item())
        full = tdist.gather_spectrum(spec_local if cdev == dev else spec_local.cpu(), nwn, world, rank,
                                     out=gathered).cpu().numpy()
    else:
        full = spec_local[: hi - lo].cpu().numpy()

    if rank == 0:
        L, R, nang = stats["nlines_inrange"], stats["layers_swept"], int(opts.nangles)
        launches = max(int(stats["sweep_launches"]), 1)
        kern = {"k_group_sweep": stats["ms_k_sweep"], "k_sticky_index": stats["ms_k_sticky"],
                "k_accumulate": stats["ms_k_accum"]}
        dom = max(kern, key=kern.get)
        # Two byte counts per kernel:
        #  * ref  = the REFERENCE's data flow: the line list (26 B/line) is scanned once per layer
        #           and per pass -- 52 B per line-layer for the strength kernel, which does both
        #           passes' scans; 4 B per accumulated bin + 8 B per stored e for the accumulation;
        #  * kmin = what THIS data flow has to move (SURVEY's B_min idea): the strength kernel reads
        #           a line ONCE per launch (27 B) for all the launch's layers and writes 9 B per
        #           (group, layer); the accumulation reads those 9 B + the group's fine-grid index
        #           (4 B) per (group, layer), the table entries and writes e.
        # Fusing a launch's layers makes ref/launch-time exceed the HBM peak (it is not HBM traffic),
        # so roofline.achieved uses kmin -- it cannot exceed 1 -- and ref is reported beside it.
        G = stats["ngroups"]
        ref = {"k_group_sweep": 52.0 * L * R, "k_sticky_index": 0.0,
               "k_accumulate": 4.0 * stats["sum_bins"] + 8.0 * R * (hi - lo)}
        kmin = {"k_group_sweep": 27.0 * L * launches + 9.0 * G * R, "k_sticky_index": 0.0,
                "k_accumulate": 13.0 * G * R + 4.0 * stats["sum_bins"] + 8.0 * R * (hi - lo)}
        alg = kmin
        ach = alg[dom] / launches / (kern[dom] / launches * 1e-3) / 1e9 if kern[dom] > 0 else 0.0
        ref_gbs = ref[dom] / (kern[dom] * 1e-3) / 1e9 if kern[dom] > 0 else 0.0
        # rank 0's share (its lines, bins and slice); x world for the job when N>1
        b_alg_run = world * (52.0 * L * R + 4.0 * stats["sum_bins"] + 24.0 * R * (hi - lo) + 8.0 * (hi - lo) * (1 + nang))
        out = {
            "metric": "wavenumber-points*layers/sec (CH4 2-4um emission spectrum)",
            "value": nwn * layers_needed / (ms_step * 1e-3),
            "unit": "wavenumber-points*layers/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "CH4-demo-shaped emission (BASELINE configs[1]%s): %g-%g cm-1 @%g cm-1, wnosamp %d, "
                                   "%s, H2-H2%s CIA" % (
                                       ", band and line list x%d: one demo-sized slice per GPU" % grow if grow > 1 else "",
                                       args.wnlow, wnhigh, args.wndelt, args.wnosamp,
                                       "eclipse, 5 angles" if args.solution == "eclipse" else "transit (slant paths)",
                                       " + H2-He" if args.ncia > 1 else ""),
                       "n_wn": nwn, "n_layers": nlayer, "layers_needed": layers_needed, "layers_swept": R,
                       "n_lines": int(P.static.nlines), "n_groups": stats["ngroups"], "n_kmax_candidates": stats["ncandidates"], "sum_bins": stats["sum_bins"],
                       "voigt_grid": "%dx%d" % (st.ndop, st.nlor), "table_floats": stats["table_floats"],
                       "parallelism": "wn-shard x%d" % world,
                       "layer_chunk": args.layer_chunk or "auto: the previous run's depth in equal steps of <= 32 "
                                                          "layers (12 per step on a handle's first run)",
                       "steps_per_run": launches,
                       "depth_hint": "step plan ends at the previous run's deepest layer (retrieval-loop reuse; "
                                     "warm-up runs provide it); a run that needs more goes on from there",
                       "ms_create_total": 1e3 * t_create, "ms_create_voigt_table_kernels": stats["ms_create_table"],
                       "ms_kernels": {k: round(v, 4) for k, v in kern.items()},
                       "ms_tau": stats["ms_tau"], "ms_run_device": stats["ms_run_total"],
                       "ms_host_cia": stats["ms_cia"], "ms_host_total_profiled_run": stats["ms_host_total"],
                       "b_min_run_bytes": world * (27.0 * L * launches + 22.0 * G * R + 4.0 * stats["sum_bins"]
                                                   + 24.0 * R * (hi - lo) + 8.0 * (hi - lo) * (1 + nang)),
                       "b_alg_run_bytes": b_alg_run,
                       "whole_run_alg_GBs": b_alg_run / (ms_step * 1e-3) / 1e9,
                       "line_layer_bins_per_s": world * stats["sum_bins"] / (ms_step * 1e-3)},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS,
                         "traffic": measured_traffic(dom) if (world == 1 and args.lines == 1_000_000) else None,
                         "note": "achieved = bytes this data flow must move per launch (lines once per launch + "
                                 "9 B per group-layer out) / measured launch time; `traffic` = the PMC-measured HBM "
                                 "bytes per launch.  The reference's own flow (52 B per line-layer) is in "
                                 "ref_flow_GBs: above the HBM peak because a launch covers ~27 layers per line read. "
                                 "This kernel is bound by the fp64 vector pipe (two exp per line-layer; 110 VALU "
                                 "instructions per line-layer by SQ_INSTS_VALU, ~32 of them the exps), see "
                                 "valu_issue_frac_est",
                         "alg_bytes_per_launch": alg[dom] / launches, "avg_launch_ms": kern[dom] / launches,
                         "launches": launches, "ref_flow_GBs": ref_gbs,
                         "ref_flow_bytes_per_launch": ref[dom] / launches},
        }
        if dom == "k_group_sweep" and kern[dom] > 0:
            # 110 wave instructions per line-layer (SQ_INSTS_VALU x 64 / line-layers), 4 cycles each
            # on a 16-lane SIMD, 1024 SIMDs at 2.4 GHz
            out["roofline"]["valu_issue_frac_est"] = (110.0 * L * R / 64.0) / (kern[dom] * 1e-3) / (1024 * 2.4e9 / 4.0)
        if dom == "k_accumulate" and ach > HBM_PEAK_GBS:
            # wide-profile regime (fine grids): neighbouring lines re-read the same profile rows, which
            # therefore come out of L1/L2 -- the 4 B per accumulated bin never reach HBM and an HBM
            # fraction would exceed 1.  The kernel is then bound by the fp64 vector pipe
            # (cvt + mul + add per term; mul/add = 2 flop); peak = MI355X vector fp64 spec.
            tf = 2.0 * stats["sum_bins"] / (kern[dom] * 1e-3) / 1e12
            out["roofline"].update(bound="fp64-valu", achieved=tf, peak=FP64_VALU_TFLOPS, unit="TFLOP/s",
                                   frac=tf / FP64_VALU_TFLOPS, alg_table_GBs_from_cache=ach,
                                   note="table rows served from L1/L2 (reused by neighbouring lines); "
                                        "achieved = 2 flop per accumulated bin / launch time")
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args, full)
            except Exception as e:          # the baseline must never sink the measurement
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    eng.close()
    if comm is not None:
        tdist.destroy_comm(comm)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
