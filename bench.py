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
import datetime
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

# A handle keeps three queues busy at once (walks; the earlier steps' combines and optical depths;
# the CIA kernels).  The HIP runtime maps streams onto 4 hardware queues per device by default, and
# an RCCL communicator in the process takes some of them: two of the handle's streams then share a
# queue and the CIA kernels run AFTER the walks instead of under them (+0.11 ms per spectrum,
# measured with a one-rank communicator).  Must be set before the runtime is loaded.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

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
    ap.add_argument("--species", type=int, choices=(1, 3), default=1,
                    help="line databases: 1 = CH4 (two isotopes), 3 = H2O + CH4 + CO (six isotopes, --lines in total: BASELINE configs[2])")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="N>1 headline: strong = the ONE CH4 2-4 um demo run split N ways -- BASELINE.json's metric "
                         "(wall-clock of that run at 1/2/4/8 GPUs; 2501 rays are too few to fill 8 GPUs: DESIGN.md "
                         "section 5); weak = every GPU gets its own demo-sized slice of a band and line list grown "
                         "N-fold.  The other mode rides along as config.<mode>")
    ap.add_argument("--no-extras", action="store_true",
                    help="N>1: skip the secondary measurements (weak scaling; BASELINE configs[4] split N ways)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse", action="store_true",
                    help="N>1 on FEWER GPUs (ranks share devices): gloo instead of RCCL, the gather goes through host "
                         "memory.  Exercises the multi-rank code path; its timing means nothing")
    ap.add_argument("--comm-single", action="store_true",
                    help="N=1 only: run through a 1-rank RCCL communicator (the code path every rank of an N>1 job takes)")
    ap.add_argument("--cpu-lines", type=int, default=0, help="lines of the CPU-baseline sample (0 = full workload)")
    return ap.parse_args()


def kernel_source_hash():
    """Fingerprint of the kernel sources a committed PMC profile must match to be quoted."""
    import hashlib
    h = hashlib.sha1()
    for f in ("trx_walk.hip.h", "trx_kernels.hip.h", "trx_rows.hip.h", "trx_tail.hip.h", "trx_lanes.hip.h"):
        h.update(open(os.path.join(ROOT, "transit_amd", "csrc", "hip", f), "rb").read())
    return h.hexdigest()[:12]


def production_launches(kernels, kernel, launch_key, family=True):
    """{name: launches} of the PRODUCTION instantiations of `kernel` in a profile summary that belong to
    the bench's step plan.  Names carry their template arguments -- trx::k_line_walk<NB, PROF, LPL>,
    trx::k_line_walk_lanes<NB, D>, trx::k_accumulate_rows<COUNT, M> -- and a boolean is the counting switch: those
    with it on (instrumented variants, one launch per bench run) are left out by PARSING the arguments.
    The plan's kernels are the ones every timed run launches; an instantiation that only the first,
    unhinted run of a handle meets (a frame size on the way down) has a handful of launches in the
    pass and is left out too (< half of the most-launched one)."""
    prod = {}
    for name, k in kernels.items():
        m = re.match(r"(?:void )?trx::(\w+)(?:<([^>]*)>)?", name)
        if not m or (m.group(1) != kernel and not (family and kernel == "k_line_walk" and m.group(1) in ("k_line_walk_packed", "k_line_walk_lanes"))):
            continue
        targs = [a.strip() for a in (m.group(2) or "").split(",") if a.strip()]
        if "true" in targs:            # PROF / COUNT
            continue
        # weights: the launches of the TRACE run (bench.py at its default steps -- the step plan's own mix
        # of kernels) where the summary has them; a PMC pass is three steps behind two warm-ups, whose first,
        # unhinted run launches other instantiations
        prod[name] = float(k.get("trace_calls") or k[launch_key])
    if not prod:
        return {}
    top = max(prod.values())
    return {n: v for n, v in prod.items() if v >= 0.5 * top}


DEMO_KEY = "2501 x 100 x 1000000 / 81"        # summaries without a workload_key (round 2) were taken on the default workload


def measured_traffic(kernel, key=DEMO_KEY):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary
    (profiles/*pmc_traffic*.json; FETCH_SIZE / WRITE_SIZE from separate --pmc passes, corrected as
    MI355X_MICROARCH.md prescribes) -- quoted ONLY when that summary was collected from the kernel
    sources of this tree (its `kernel_sources` fingerprint); otherwise None: a stale number is
    worse than none.  Returns ((corrected, uncorrected) bytes per launch, file) or (None, None)."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")), reverse=True):
        try:
            d = json.load(open(f))
            if d.get("kernel_sources") != kernel_source_hash() or (d.get("workload_key") or DEMO_KEY) != key:
                continue
            w = production_launches(d["kernels"], kernel, "launches_FETCH_pass")
            tot = raw = n = 0.0
            for name, nl in w.items():
                k = d["kernels"][name]
                tot += float(k["hbm_bytes_per_launch"]) * nl
                raw += float(k.get("hbm_bytes_per_launch_uncorrected", 0.0)) * nl
                n += nl
            if n:
                return (tot / n, raw / n), os.path.relpath(f, ROOT)
        except Exception:
            continue
    return None, None


def measured_valu(kernel, key=DEMO_KEY, family=True):
    """SQ_INSTS_VALU per launch of `kernel` (weighted over the production instantiations of the step
    plan) from the committed SQ pass that matches this tree's kernel sources, or None."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic*.json")), reverse=True):
        try:
            t = json.load(open(f))
            if t.get("kernel_sources") != kernel_source_hash() or (t.get("workload_key") or DEMO_KEY) != key:
                continue
            d = json.load(open(f.replace("pmc_traffic", "sq_mix")))
            w = production_launches(d["kernels"], kernel, "launches", family)
            tot = n = 0.0
            for name, nl in w.items():
                tot += float(d["kernels"][name]["SQ_INSTS_VALU_avg_launch"]) * nl
                n += nl
            if n:
                return tot / n
        except Exception:
            continue
    return None


def make_workload(args, tag, nlines, verb=2, wnhigh=None, unique=True):
    from transit_amd import synth
    d = os.path.join(tempfile.gettempdir(), "transit_bench_%s_%d_%d_%s" % (
        tag, nlines, args.layers, os.getpid() if unique else os.environ.get("MASTER_PORT", "0")))
    if getattr(args, "species", 1) == 3:
        d += "_3sp"
    if not unique and os.path.exists(os.path.join(d, "case.cfg")):
        return d
    dbs = synth.three_species_dbs(nlines // 3, args.wnlow, wnhigh or args.wnhigh) if getattr(args, "species", 1) == 3 else None
    synth.make_case(d, nlines=nlines, dbs=dbs, wnlow=args.wnlow, wnhigh=wnhigh or args.wnhigh, wndelt=args.wndelt,
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
            out["all_cores"] = cpu_all_cores(args, d, layers, spec)
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


def cpu_all_cores(args, workdir, layers_needed, ref_spectrum):
    """What all host cores can do: our C restatement with its one parallel axis switched on --
    the layers of the extinction sweep, which are independent (OpenMP; the reference itself is
    single-threaded, and sweeps layers lazily).  The sweep takes every layer (the lazy order is
    what makes the reference serial), so `value` counts the layers the spectrum needed, as above."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as ol
        from transit_amd.host import Problem
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        try:                                                 # a container's CPU quota, where there is one
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            if q != "max":
                cores = max(1, min(cores, int(float(q) / float(per) + 0.5)))
        except Exception:
            pass
        cores = min(cores, 16)                               # one GPU's share of the host on this pool
        os.environ.setdefault("OMP_NUM_THREADS", str(cores))
        P = Problem.from_cfg(os.path.join(workdir, "case.cfg"))
        eng = ol.OracleEngine(P.static)                      # (includes its Voigt table: not timed)
        opts = P.opts
        opts.eager = 1
        t0 = time.time()
        r = eng.run(P.atm, opts)
        t = time.time() - t0
        opts.eager = 0
        eng.close()
        ok = float(np.max(np.abs(r["spectrum"] / ref_spectrum - 1))) if len(ref_spectrum) == len(r["spectrum"]) else None
        return {"kind": "port", "cores": int(os.environ["OMP_NUM_THREADS"]), "seconds": t,
                "value": P.nwn * layers_needed / t, "unit": "wavenumber-points*layers/s",
                "sample": "full workload, all %d layers swept in parallel (OpenMP over layers), then the serial "
                          "optical-depth / emission loops" % P.nlayer,
                "vs_reference_max_rel": ok}
    except Exception as e:                                   # a baseline, never a reason to lose the bench line
        return {"kind": "port", "error": "%s: %s" % (type(e).__name__, e)}


def baseline_config(args, grow=1):
    """Which of BASELINE.json's configs the flags describe ("custom" when none): the label of config.workload."""
    near = lambda a, b: abs(a - b) <= 1e-6 * max(abs(a), abs(b), 1.0)
    demo_grid = near(args.wnlow, 2500.0) and near(args.wnhigh, 5000.0) and near(args.wndelt, 1.0) and args.wnosamp == 2160
    if grow == 1 and demo_grid and args.layers == 100 and args.lines == 1_000_000 and args.species == 1:
        if args.solution == "eclipse" and args.ncia == 1:
            return "BASELINE configs[1]: CH4-demo-shaped emission"
        if args.solution == "transit" and args.ncia == 2:
            return "BASELINE configs[3]: transmission (slant paths), H2-H2 + H2-He CIA"
    wide = near(args.wnlow, 333.33) and near(args.wnhigh, 10000.0)
    if grow == 1 and wide and near(args.wndelt, 1.0) and args.wnosamp == 2160 and args.layers == 200 and args.lines == 3_000_000 \
            and args.species == 3 and args.solution == "eclipse":
        return "BASELINE configs[2]: H2O + CH4 + CO, 1-30 um at 1 cm-1, 200 layers"
    if grow == 1 and wide and near(args.wndelt, 0.0009667) and args.wnosamp == 1 and args.layers == 150 and args.lines == 10_000_000 \
            and args.solution == "eclipse":
        return "BASELINE configs[4]: retrieval scale, 1e7 wavenumbers x 150 layers x 1e7 lines"
    return "custom (no BASELINE config has these flags)"


def frac_or_none(x):
    """A fraction of a roof, or None where the byte count it comes from was never moved (> 1)."""
    return x if x <= 1.0 else None


def time_steps(step, fence, warmup, steps, reduce_max=None):
    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    el = time.perf_counter() - t0
    if reduce_max:
        el = reduce_max(el)
    return 1e3 * el / steps


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
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=600))
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=datetime.timedelta(seconds=600))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    dev = torch.device("cuda", local)
    cdev = torch.device("cpu") if (world > 1 and args.rehearse) else dev       # where torch collectives run

    from transit_amd.host import Problem
    from transit_amd.engine import Engine
    from transit_amd.shard import balanced_bounds, bin_costs
    from transit_amd import dist as tdist

    def reduce_max(x):
        t = torch.tensor([x], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    comm = tdist.create_comm(world, rank, local) if ((world > 1 and not args.rehearse) or args.comm_single) else None

    def shared_case(tag, nlines, wnhigh=None, **over):
        """Input files of one workload: written by rank 0, read by every rank."""
        a2 = argparse.Namespace(**{**vars(args), **over})
        if world == 1:
            return make_workload(a2, tag, nlines, wnhigh=wnhigh)
        if rank == 0:
            d = make_workload(a2, "%s_w%d" % (tag, world), nlines, wnhigh=wnhigh, unique=False)
        dist.barrier()
        if rank != 0:
            d = os.path.join(tempfile.gettempdir(), "transit_bench_%s_w%d_%d_%d_%s" % (
                tag, world, nlines, a2.layers, os.environ.get("MASTER_PORT", "0")))
        return d

    def sharded_run(d, steps, warmup, keep=False):
        """One workload on `world` ranks: bins split by cost, every rank runs its shard on its own
        handle, trx_gather collects the slices.  Returns a dict of measurements (rank 0: complete)."""
        P = Problem.from_cfg(os.path.join(d, "case.cfg"))
        nwn = P.nwn
        bounds = balanced_bounds(bin_costs(P.static, P.nlayer), world)
        lo, hi = bounds[rank]
        P.set_shard(lo, hi)
        st = P.static
        st.device = local
        if comm is not None:
            st.comm, st.nranks, st.rank = comm, world, rank
        t0 = time.time()
        eng = Engine(st)
        t_first_create = time.time() - t0
        opts = P.opts
        opts.layer_chunk = args.layer_chunk
        opts.profile = 0
        mpad = max(h - l for l, h in bounds)
        spec_local = torch.zeros(mpad, dtype=torch.float64, device=dev)      # slice + padding to equal counts
        gathered = torch.zeros(mpad * world, dtype=torch.float64, device=dev if comm is not None else cdev) if world > 1 else None

        host_spec = np.zeros(hi - lo)

        def step_host():          # SURVEY 8(d)'s t_run: trx_run, H2D of the atmosphere and D2H of the spectrum included
            eng.run_into(P.atm, opts, host_spec)

        def step():
            eng.run_device(P.atm, opts, spec_local.data_ptr())
            if world > 1:       # the single exchange of the path
                if comm is not None:      # trx_gather: ncclAllGather on the handle's stream (C ABI)
                    eng.gather(spec_local.data_ptr(), gathered.data_ptr(), mpad)
                else:                     # rehearsal on a shared GPU: gloo through host memory
                    dist.all_gather_into_tensor(gathered, spec_local.cpu())

        # N = 1: the headline is trx_run with the spectrum handed back to the HOST (8(d)); N > 1: the spectrum stays in
        # device memory for the gather (trx_run_device + trx_gather).  The other form rides along (ms_device_spectrum).
        timed = step_host if world == 1 else step
        ms_step = time_steps(timed, fence, warmup, steps, reduce_max if world > 1 else None)
        # rider: the same step timed over at least 200 more steps right behind the timed region (no further
        # warm-up).  The headline keeps the driver's K and W; with K = 20 its 7 ms sit on a device whose
        # clocks are still on their way up (5 warm-up steps are 2 ms of GPU work): 0.362 ms per step
        # against 0.345 over 200 and 0.341 over 2000 steps, same box, same build (round 4, gpurun_out/r4_base_*).
        steady_steps = max(200, steps) if steps < 2000 else 0
        ms_steady = time_steps(timed, fence, 0, steady_steps, reduce_max if world > 1 else None) if steady_steps else ms_step
        ms_device_spectrum = time_steps(step, fence, 0, steady_steps or steps) if world == 1 else None
        # the same steps once more with the production kernels bracketed by HIP events on the
        # streams they are launched on (profile 1): every launch counts, sum / launches is what
        # a kernel trace reports as the average
        opts.profile = 1
        ev = {"ms_k_sweep": 0.0, "ms_k_walk": 0.0, "ms_k_accum": 0.0, "ms_tau": 0.0, "ms_run_total": 0.0, "sweep_launches": 0, "runs": 0,
              "ms_k_walk_form": [0.0, 0.0, 0.0], "ms_walk_span": 0.0}
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.run_device(P.atm, opts, spec_local.data_ptr())
            s1 = eng.stats()
            for k in ("ms_k_sweep", "ms_k_walk", "ms_k_accum", "ms_tau", "ms_run_total", "sweep_launches", "ms_walk_span"):
                ev[k] += s1[k]
            for fi in range(3):
                ev["ms_k_walk_form"][fi] += s1["ms_k_walk_form"][fi]
            ev["runs"] += 1
        ev["ms_per_step_with_events"] = 1e3 * (time.perf_counter() - t0) / max(steps, 1)
        opts.profile = 2                 # one counted run (instrumented kernel variants): bins, evaluated / skipped groups
        r = eng.run(P.atm, opts, debug=("last",))
        opts.profile = 0
        stats = eng.stats()
        stats["events"] = ev
        layers_needed = int(r["last"].max()) + 1
        if world > 1:
            layers_needed = int(reduce_max(layers_needed))
            g = gathered.cpu().numpy()
            full = np.concatenate([g[k * mpad: k * mpad + (h - l)] for k, (l, h) in enumerate(bounds)])
        else:
            full = spec_local[: hi - lo].cpu().numpy()
            if not np.array_equal(full, host_spec):
                raise SystemExit("bench: trx_run and trx_run_device returned different spectra")
        out = dict(P=P, eng=eng, stats=stats, ms_step=ms_step, ms_steady=ms_steady, steady_steps=steady_steps or steps,
                   ms_device_spectrum=ms_device_spectrum,
                   layers_needed=layers_needed, full=full, nwn=nwn,
                   bounds=bounds, t_first_create=t_first_create, opts=opts, dir=d, static=st)
        if not keep:
            eng.close()
            out["eng"] = None
        return out

    # ---- headline: the CH4-demo run.  N>1: that ONE run split N ways (BASELINE.json's metric: strong
    # scaling) unless --scaling weak (one demo-sized slice per GPU, the band and the line list grown
    # N-fold); either way the other mode is measured too and reported under config.
    grow = world if (args.scaling == "weak" and world > 1) else 1
    wnhigh = args.wnlow + grow * (args.wnhigh - args.wnlow)
    d = shared_case("demo", grow * args.lines, wnhigh=wnhigh)
    M = sharded_run(d, args.steps, args.warmup, keep=True)
    P, eng, stats, ms_step, layers_needed, full, nwn = (M[k] for k in ("P", "eng", "stats", "ms_step", "layers_needed", "full", "nwn"))
    opts, st = M["opts"], M["static"]
    lo, hi = M["bounds"][rank]
    nlayer = P.nlayer

    extras = {"ms_create_first_in_process": 1e3 * M["t_first_create"]}
    if world > 1:
        # what every rank paid, so that a SCALE line explains itself: the event-timed kernels of ITS shard (walks, combine /
        # accumulate, optical depth), the device time of a whole run and the host's time in the call, per spectrum
        ev_r = M["stats"]["events"]; nr_ = max(ev_r["runs"], 1)
        mine = {"rank": rank, "bins": int(hi - lo), "ms_k_walk": ev_r["ms_k_walk"] / nr_, "ms_k_accum": ev_r["ms_k_accum"] / nr_,
                "ms_tau": ev_r["ms_tau"] / nr_, "ms_run_device": ev_r["ms_run_total"] / nr_,
                "ms_per_step_with_events": ev_r["ms_per_step_with_events"], "walk_steps": int(M["stats"]["walk_steps"]),
                "layers_swept": int(M["stats"]["layers_swept"])}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        extras["per_rank"] = per_rank
        if args.rehearse and rank == 0:
            # the multi-rank path's contract, checked where ranks can be rehearsed: the stitched spectrum of the N-way job
            # IS the one-GPU spectrum, bit for bit (eclipse geometry: every sum of a bin has one order, whatever the shard)
            P1 = Problem.from_cfg(os.path.join(d, "case.cfg"))
            st1 = P1.static
            st1.device = local
            e1 = Engine(st1)
            whole = e1.run(P1.atm, P1.opts)["spectrum"]
            e1.close()
            same = bool(np.array_equal(whole, full))
            extras["rehearse_stitched_equals_one_gpu_bitwise"] = same
            if not same and args.solution == "eclipse":
                raise SystemExit("bench --rehearse: the stitched spectrum of %d ranks differs from the one-GPU spectrum (max rel %g)"
                                 % (world, float(np.max(np.abs(full / whole - 1.0)))))
    if world == 1:
        # a handle made in a warm process (the first create of a process also pays for loading the
        # code object and the runtime's lazy initialisation), its first -- unhinted -- spectrum
        t0 = time.time(); e2 = Engine(st); extras["ms_create_total"] = 1e3 * (time.time() - t0)
        t0 = time.time(); e2.run(P.atm, opts); extras["ms_first_spectrum_cold"] = 1e3 * (time.time() - t0)
        e2.close()
        if not args.no_extras:
            # several handles computing spectra at the same time (retrieval chains sharing a GPU): one
            # spectrum's launch gaps and latency chains are another one's room.  Not the headline:
            # BASELINE's metric is one spectrum at a time.
            import threading
            engs = [Engine(st) for _ in range(3)]
            def _loop(e, n):
                for _ in range(n):
                    e.run(P.atm, opts)
            for e in engs:
                _loop(e, 5)
            n3 = max(20, args.steps // 2)
            th = [threading.Thread(target=_loop, args=(e, n3)) for e in engs]
            t0 = time.time()
            for t in th: t.start()
            for t in th: t.join()
            extras["ms_per_spectrum_3_handles_at_once"] = 1e3 * (time.time() - t0) / (3 * n3)
            for e in engs:
                e.close()
            # the same through the C ABI's own batch call: 8 atmospheres per trx_run_batch, dealt to 3 handles
            # by the library's host threads (no Python between the spectra)
            from transit_amd.engine import Batch
            bt = Batch(st, ways=3)
            atms8 = [P.atm] * 8
            for _ in range(3):
                bt.run(atms8, opts)
            nb8 = max(8, args.steps // 4)
            t0 = time.time()
            for _ in range(nb8):
                bt.run(atms8, opts)
            extras["ms_per_spectrum_batch8"] = 1e3 * (time.time() - t0) / (8 * nb8)
            extras["batch_ways"] = 3
            bt.close()
        exe = os.path.join(ROOT, "transit_amd", "lib", "transit_hip")
        if os.path.exists(exe):          # the one-shot command, fresh process, same input files
            t0 = time.time()
            pr = subprocess.run([exe, "-c", "case.cfg", "--outspec", "cli_spectrum.dat"], cwd=M["dir"], capture_output=True, text=True)
            extras["ms_cli_wall"] = 1e3 * (time.time() - t0) if pr.returncode == 0 else None
    elif not args.no_extras:
        # secondary: the other scaling mode, and BASELINE configs[4] (retrieval scale) split N ways --
        # the one configuration large enough for 8 GPUs to show their worth.  They ride along with
        # the headline and must never cost it: each runs only while the time budget lasts (rank 0
        # decides, every rank follows), and a failure is recorded instead of raised.
        budget_s = float(os.environ.get("TRANSIT_BENCH_EXTRAS_BUDGET_S", "240"))
        t_extras = time.time()

        def agreed(ok):
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=cdev)
            dist.broadcast(t, src=0)
            return bool(t.item())

        def extra(name, need_s, fn):
            if not agreed(time.time() - t_extras + need_s <= budget_s):
                extras[name] = {"skipped": "time budget of the secondary measurements (%.0f s) used up" % budget_s}
                return
            try:
                extras[name] = fn()
            except Exception as e:          # (a rank that fails alone leaves the others to the process group's timeout)
                extras[name] = {"error": "%s: %s" % (type(e).__name__, e)}

        def other_mode():
            other = "weak" if args.scaling == "strong" else "strong"
            g2 = world if other == "weak" else 1
            d2 = shared_case("alt", g2 * args.lines, wnhigh=args.wnlow + g2 * (args.wnhigh - args.wnlow))
            M2 = sharded_run(d2, max(5, args.steps // 10), 3)
            return {"mode": other, "ms_per_step": M2["ms_step"], "value": M2["nwn"] * M2["layers_needed"] / (M2["ms_step"] * 1e-3),
                    "n_wn": M2["nwn"], "n_lines": g2 * args.lines}

        def c5_strong():
            d5 = shared_case("c5", 10_000_000, wnhigh=10000.0, wnlow=333.33, wndelt=0.0009667, wnosamp=1, layers=150)
            M5 = sharded_run(d5, 3, 2)
            return {"ms_per_step": M5["ms_step"], "value": M5["nwn"] * M5["layers_needed"] / (M5["ms_step"] * 1e-3),
                    "n_wn": M5["nwn"], "n_layers": 150, "n_lines": 10_000_000,
                    "workload": "BASELINE configs[4]: 333.33-10000 cm-1 @0.0009667 cm-1, 150 layers, 1e7 lines"}

        extra("weak" if args.scaling == "strong" else "strong", 30, other_mode)
        extra("c5_strong", 120, c5_strong)

    if rank == 0:
        L, R, nang = stats["nlines_inrange"], stats["layers_swept"], int(opts.nangles)
        launches = max(int(stats["sweep_launches"]), 1)
        wsteps = int(stats["walk_steps"]); ssteps = launches - wsteps           # walk steps / two-kernel steps of a run
        walked = wsteps == launches
        mixed = 0 < wsteps < launches            # upper layers walked, deep ones in the two-kernel form
        ev = stats["events"]
        runs = max(ev["runs"], 1)
        nbins = hi - lo
        G = stats["ngroups"]
        Rw = int(stats["walk_layers"]); Rs = R - Rw                               # layers swept by walk steps / by two-kernel steps
        bins_w = stats["sum_bins_walk"]; bins_s = stats["sum_bins"] - bins_w
        # Per run, from the event-timed repeat of the timed region (production kernels), each kernel family with
        # ITS launches and ITS bytes.  The walk reads one 32-byte record per line and step, 4 bytes per accumulated
        # bin (table) and writes its partial records (8 B per layer of the step); the combine reads those and
        # writes e.  Two-kernel form: 27 B per line and step + 9 B per group and layer written (strength, Doppler
        # index), then 13 B per group and layer read back + 4 B per accumulated bin + e.
        acc_name = "k_walk_combine" if walked else ("k_accumulate" if wsteps == 0 else "k_walk_combine + k_accumulate")
        kern, alg, nlaunch = {}, {}, {}
        # the walk family, form by form (trx_stats.walk_form_*): k_line_walk reads a 32-byte record per line and step,
        # k_line_walk_lanes also the line's 8-byte base point (40 B); both 4 B per accumulated bin and their partial records
        FORMS = ("k_line_walk", "k_line_walk_lanes", "k_line_walk_packed")
        walk_forms = {}
        for fi, fname in enumerate(FORMS):
            fsteps = int(stats["walk_form_steps"][fi])
            if not fsteps:
                continue
            fms = ev["ms_k_walk_form"][fi] / runs
            fbytes = (40.0 if fi == 1 else 32.0) * L * fsteps + 4.0 * stats["walk_form_bins"][fi] + 8.0 * stats["walk_form_record_lanes"][fi]
            walk_forms[fname] = {"steps_per_run": fsteps, "layers": int(stats["walk_form_layers"][fi]), "ms_per_run": fms,
                                 "avg_launch_ms": fms / fsteps, "alg_bytes_per_launch": fbytes / fsteps,
                                 "achieved_GBs": fbytes / (fms * 1e-3) / 1e9 if fms > 0 else None,
                                 "frac": fbytes / (fms * 1e-3) / 1e9 / HBM_PEAK_GBS if fms > 0 else None}
        # the walks of a hinted two-step run share the device on two queues: the family's time is then the span from the
        # first walk's start to the last one's end, less than the sum of the kernels' own (overlapping) durations
        walk_span = ev["ms_walk_span"] / runs
        walks_overlap = wsteps >= 2 and 0 < walk_span < 0.98 * ev["ms_k_walk"] / runs
        if wsteps:
            kern["k_line_walk"] = walk_span if walks_overlap else ev["ms_k_walk"] / runs; nlaunch["k_line_walk"] = wsteps
            alg["k_line_walk"] = sum(v["alg_bytes_per_launch"] * v["steps_per_run"] for v in walk_forms.values())
        if ssteps:
            kern["k_group_sweep"] = ev["ms_k_sweep"] / runs; nlaunch["k_group_sweep"] = ssteps
            alg["k_group_sweep"] = 27.0 * L * ssteps + 9.0 * G * Rs
        tail_run = walked and launches <= 2 and ev["ms_k_accum"] == 0          # the combines are part of k_ray_tail (timed as ms_tau)
        kern[acc_name] = ev["ms_k_accum"] / runs; nlaunch[acc_name] = launches
        alg[acc_name] = (8.0 * stats["walk_record_lanes"] + 8.0 * Rw * nbins) + (13.0 * G * Rs + 4.0 * bins_s + 8.0 * Rs * nbins)
        dom = max(kern, key=kern.get)
        line_k = "k_line_walk" if wsteps else "k_group_sweep"
        dlaunch = max(nlaunch[dom], 1)
        # (the walk family under its members' names: achieved / avg_launch_ms are over all its launches, `kernels` has each form)
        dom_label = " + ".join(walk_forms) if (dom == "k_line_walk" and walk_forms) else dom
        ach = alg[dom] / (kern[dom] * 1e-3) / 1e9 if kern[dom] > 0 else 0.0
        # whole run against SURVEY 8(d): B_alg follows the reference's flow (two scans of the 26-byte
        # line record per layer), B_min is the layer-fused minimum (lines once, 4 B per bin, the
        # per-(wn, layer) arrays, the outputs).  Rank 0's share x world for the job.
        b_alg_run = world * (52.0 * L * R + 4.0 * stats["sum_bins"] + 24.0 * R * nbins + 8.0 * nbins * (1 + nang))
        b_min_run = world * (52.0 * L + 4.0 * stats["sum_bins"] + 24.0 * R * nbins + 8.0 * nbins * (1 + nang))
        # the roof this kernel is actually near: vector-instruction issue.  A wave64 instruction holds
        # its SIMD's 16 lanes for 4 clocks; 256 CUs x 4 SIMDs at the 2.4 GHz peak clock.
        wkey = "%d x %d x %d / %d" % (nwn, nlayer, int(st.nlines), layers_needed)
        prof_k = "k_accumulate_rows" if ("k_accumulate" in dom and args.wnosamp == 1) else dom.split(" ")[0]     # (the kernel's name in a trace)
        valu = measured_valu(prof_k, wkey) if world == 1 else None
        if world == 1:
            for fname, v in walk_forms.items():      # each form's own instruction count, where a profile of this tree holds it
                fv = measured_valu(fname, wkey, family=False)
                v["valu_issue_frac"] = (fv * 4.0 / (v["avg_launch_ms"] * 1e-3 * 1024 * 2.4e9)) if (fv and v["avg_launch_ms"] > 0) else None
        valu_frac = (valu * 4.0 / (kern[dom] / dlaunch * 1e-3 * 1024 * 2.4e9)) if (valu and kern[dom] > 0) else None
        tr_pair, traffic_file = measured_traffic(prof_k, wkey) if world == 1 else (None, None)
        traffic, traffic_raw = tr_pair if tr_pair else (None, None)
        out = {
            "metric": "wavenumber-points*layers/sec (CH4 2-4um emission spectrum)",
            "value": nwn * layers_needed / (ms_step * 1e-3),
            "unit": "wavenumber-points*layers/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "ms_per_step_steady": M["ms_steady"], "steady_steps": M["steady_steps"],
            "timed_call": "trx_run: atmosphere from host memory, spectrum back to host memory (SURVEY 8(d) t_run)" if world == 1
                          else "trx_run_device + trx_gather: the slices stay in device memory for the one ncclAllGather",
            "ms_per_step_device_spectrum": M["ms_device_spectrum"],
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s%s: %g-%g cm-1 @%g cm-1, wnosamp %d, %s, "
                                   "%s, H2-H2%s CIA" % (
                                       baseline_config(args, grow),
                                       " -- BASELINE configs[1] with band and line list x%d: one demo-sized slice per GPU" % grow if grow > 1 else "",
                                       args.wnlow, wnhigh, args.wndelt, args.wnosamp,
                                       "H2O + CH4 + CO lines" if args.species == 3 else "CH4 lines",
                                       "eclipse, 5 angles" if args.solution == "eclipse" else "transit (slant paths)",
                                       " + H2-He" if args.ncia > 1 else ""),
                       "n_wn": nwn, "n_layers": nlayer, "layers_needed": layers_needed, "layers_swept": R,
                       "n_lines": int(st.nlines), "n_groups": stats["ngroups"], "n_kmax_candidates": stats["ncandidates"],
                       "sum_bins": stats["sum_bins"],
                       "voigt_grid": "%dx%d" % (st.ndop, st.nlor), "table_floats": stats["table_floats"],
                       "parallelism": "wn-shard x%d, bins split by cost %s, one trx_gather per spectrum" % (
                           world, [h - l for l, h in M["bounds"]]) if world > 1 else "one GPU",
                       "layer_chunk": args.layer_chunk or "auto: the previous run's depth in equal steps (<= 64 layers where "
                                                          "the line walk applies, one lane per layer)",
                       "steps_per_run": launches, "walk_steps_per_run": wsteps,
                       "line_kernel": " + ".join(["%s (%d steps, %d layers)" % (n, v["steps_per_run"], v["layers"]) for n, v in walk_forms.items()] +
                                                 (["k_group_sweep (%d steps, %d layers)" % (ssteps, Rs)] if ssteps else [])),
                       "depth_hint": "step plan ends at the previous run's deepest layer (retrieval-loop reuse; "
                                     "warm-up runs provide it); a run that needs more goes on from there",
                       "ms_create_voigt_table_kernels": stats["ms_create_table"],
                       "ms_kernels": dict({k: round(v, 4) for k, v in kern.items() if not (tail_run and k == acc_name)},
                                          **({"k_ray_tail": round(ev["ms_tau"] / runs, 4)} if tail_run else {})),
                       "ms_walks_sum_of_own_durations": round(ev["ms_k_walk"] / runs, 4) if walks_overlap else None,
                       "ms_kernels_source": "HIP events on the kernels' own streams over a repeat of the timed region "
                                            "(%d runs, %.4f ms per step with the events' packets in): the same plan, queues and kernels "
                                            "as the timed region%s" % (runs, ev["ms_per_step_with_events"],
                                            " -- two walks next to each other on two queues (k_line_walk = first start to last end), "
                                            "then ONE kernel, k_ray_tail (combines, optical depths, spectrum)" if walks_overlap else ""),
                       "after_the_walks": "k_ray_tail (hinted plans of one or two walk steps; profiles/*_kernel_stats.csv)"
                                          if walked and launches <= 2 else "step kernels (combine, optical depth, spectrum)",
                       "ms_tau": ev["ms_tau"] / runs, "ms_run_device": ev["ms_run_total"] / runs,
                       "ms_host_cia": stats["ms_cia"], "ms_host_total_profiled_run": stats["ms_host_total"],
                       "b_alg_run_bytes": b_alg_run, "b_min_run_bytes": b_min_run,
                       "whole_run_alg_GBs": b_alg_run / (ms_step * 1e-3) / 1e9,
                       "line_layer_bins_per_s": world * stats["sum_bins"] / (ms_step * 1e-3)},
            "roofline": {"bound": "hbm (distance, not the binding roof)", "kernel": dom_label, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS,
                         "kernels": walk_forms if dom == "k_line_walk" else None,
                         "traffic": traffic, "traffic_uncorrected": traffic_raw, "traffic_source": traffic_file,
                         "bmin_frac": frac_or_none(b_min_run / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS),
                         "balg_frac": frac_or_none(b_alg_run / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS),
                         "alg_bytes_per_launch": alg[dom] / dlaunch, "avg_launch_ms": kern[dom] / dlaunch,
                         "concurrent_launches": bool(walks_overlap and dom == "k_line_walk"),
                         "binding_roof": "fp64-rate vector-instruction issue", "valu_issue_frac": valu_frac,
                         "launches": dlaunch,
                         "note": ("The family's launches of one run overlap on two queues: achieved = the family's bytes / the span from the first "
                                  "launch's start to the last one's end, avg_launch_ms = that span / launches; `kernels` has each launch's own "
                                  "(longer, shared-device) duration, which is what a kernel trace reports.  " if (walks_overlap and dom == "k_line_walk") else "") +
                                 "achieved = bytes the dominant kernel family's data flow must move (32 B per line record -- 40 B in "
                                 "k_line_walk_lanes, which also reads the line's base point --, 4 B per accumulated bin, its "
                                 "partial records) / its measured time (HIP events on its own stream); `kernels` prices each "
                                 "member of the family with its own launches and bytes.  bmin_frac / balg_frac = the whole spectrum against SURVEY 8(d)'s layer-fused "
                                 "minimum / reference-flow byte counts (null where the count exceeds what HBM could "
                                 "deliver in the measured time: the fused layers read the line list once, not once "
                                 "per layer).  The walk is not HBM-bandwidth bound: it issues ~38 (two-bin frames, lanes = "
                                 "layers) and ~42 (eight-bin frames, k_line_walk_lanes) vector instructions per line and step, most at the fp64 "
                                 "rate of one per 4 clocks and SIMD, and the deep step's row gathers are one L2 request per (group, layer) -- valu_issue_frac = SQ_INSTS_VALU x 4 clocks / "
                                 "(launch time x 1024 SIMDs x 2.4 GHz); counters and probes in profiles/, DESIGN.md section 4"},
        }
        out["config"].update(extras)
        if "k_accumulate" in dom and ach > HBM_PEAK_GBS:
            # wide-profile regime (fine grids): neighbouring lines read the same profile rows, staged in
            # LDS once per run of lines (k_accumulate_rows) or served by L1 (k_accumulate_wide) -- the
            # 4 B per accumulated bin never reach HBM.  The kernel is bound by the fp64 vector pipe:
            # one fused multiply-add per bin (2 flop) at one instruction per 4 clocks and SIMD, next to
            # the three lane reads and the address add every (line, tile) pair costs.
            # peak = MI355X vector fp64 spec.  Beside it: the LDS read roof (256 B per clock and CU).
            tf = 2.0 * bins_s / (kern[dom] * 1e-3) / 1e12
            lds_peak = 256 * 256 * 2.4                        # GB/s: 256 CUs x 256 B/clk x 2.4 GHz
            out["roofline"].update(bound="fp64-valu", achieved=tf, peak=FP64_VALU_TFLOPS, unit="TFLOP/s",
                                   frac=tf / FP64_VALU_TFLOPS, binding_roof="fp64-rate vector-instruction issue",
                                   lds_read_GBs=8.0 * bins_s / (kern[dom] * 1e-3) / 1e9, lds_peak_GBs=lds_peak,
                                   lds_frac=8.0 * bins_s / (kern[dom] * 1e-3) / 1e9 / lds_peak,
                                   note="profile rows staged in LDS (reused by neighbouring lines): no HBM fraction is "
                                        "quoted for this kernel.  achieved = 2 flop per accumulated bin / launch time; "
                                        "lds_frac = 8 B per accumulated bin (the staged doubles) / launch time against "
                                        "the LDS read peak -- both count the bins a profile reaches, not the tile bins "
                                        "beyond its ends that are computed as zeros (profiles/r04_c5_*: counters)")
            out["roofline"]["bmin_frac"] = out["roofline"]["balg_frac"] = None
            for k in ("alg_bytes_per_launch",):
                out["roofline"].pop(k, None)
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args, full)
                if extras.get("ms_cli_wall") and out["cpu_baseline"].get("seconds_wall"):
                    out["config"]["cli_wall_speedup_vs_reference"] = 1e3 * out["cpu_baseline"]["seconds_wall"] / extras["ms_cli_wall"]
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
