"""A list of 10^8 lines once: synthetic TLI (2.7 GB) over configs[2]'s band,
trh_load (mmap + range selection) -> trx_create -> spectra; a shard of the handle against the whole; a 100 cm-1
window of the same file against the oracle.  Prints progress, writes gpurun_out/r4_1e8.json."""
import json, os, sys, time, tempfile, shutil
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # (profiles/ -> the repository)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
NL = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
out = {"nlines": NL}
def say(*a):
    print("[%7.1f s]" % (time.time() - T0), *a, flush=True)
T0 = time.time()
import torch
from transit_amd import synth, engine
from transit_amd.engine import Engine
from transit_amd.host import Problem
d = os.path.join(tempfile.gettempdir(), "trx_1e8")
shutil.rmtree(d, ignore_errors=True)
t = time.time()
synth.make_case(d, nlines=NL, wnlow=333.33, wnhigh=10000.0, wndelt=1.0, wnosamp=2160, nlayers=200, solution="eclipse",
                toomuch=10.0, ethresh=1e-50, nwidth=20.0, raygrid="0 20 40 60 80", ncia=1, seed=1234)
out["s_make_case"] = time.time() - t
out["tli_bytes"] = os.path.getsize(os.path.join(d, "case.tli"))
say("case written: TLI %.2f GB in %.0f s" % (out["tli_bytes"] / 1e9, out["s_make_case"]))
t = time.time(); P = Problem.from_cfg(os.path.join(d, "case.cfg")); out["s_trh_load"] = time.time() - t
say("trh_load %.2f s, %d wavenumbers, %d layers, %d lines selected" % (out["s_trh_load"], P.nwn, P.nlayer, int(P.static.nlines)))
free0 = torch.cuda.mem_get_info()[0]
msgs = []
engine.set_log(lambda lvl, m: (msgs.append(m), print("    |", m, flush=True) if ("create" in m or "trx_create" in m) else None), 5)
t = time.time(); eng = Engine(P.static); out["s_trx_create"] = time.time() - t
free1 = torch.cuda.mem_get_info()[0]
out["hbm_bytes_handle"] = free0 - free1
say("trx_create %.2f s, %.2f GB of HBM" % (out["s_trx_create"], out["hbm_bytes_handle"] / 1e9))
t = time.time(); r0 = eng.run(P.atm, P.opts, debug=("last",)); out["s_first_spectrum"] = time.time() - t
say("first spectrum (unhinted) %.3f s, layers needed %d" % (out["s_first_spectrum"], int(r0["last"].max()) + 1))
ts = []
for i in range(3):
    t = time.time(); r1 = eng.run(P.atm, P.opts); ts.append(time.time() - t)
out["s_hinted_spectrum"] = min(ts)
assert np.array_equal(r1["spectrum"], r0["spectrum"])
st = eng.stats()
out["stats"] = {k: st[k] for k in ("nlines_inrange", "ngroups", "nadd", "layers_swept", "walk_steps", "walk_records", "ncandidates", "table_floats")}
out["hbm_bytes_after_runs"] = free0 - torch.cuda.mem_get_info()[0]
say("hinted spectrum %.4f s; %.2f GB of HBM after the runs; stats %s" % (out["s_hinted_spectrum"], out["hbm_bytes_after_runs"] / 1e9, out["stats"]))
for m in msgs:
    if "walk frame" in m or "walk:" in m: print("    |", m[:300], flush=True); break
assert np.all(np.isfinite(r1["spectrum"])) and np.all(r1["spectrum"] > 0)
full = r1["spectrum"].copy()
eng.close(); del eng
# ---- a shard of the same problem: only the ranges that reach it; the same bits as the whole
lo, hi = 2700, 2800
P.set_shard(lo, hi)
t = time.time(); es = Engine(P.static); out["s_trx_create_shard"] = time.time() - t
es.run(P.atm, P.opts); rs = es.run(P.atm, P.opts)
es.close(); P.set_shard(0, P.nwn)
out["shard_equals_whole"] = bool(np.array_equal(rs["spectrum"], full[lo:hi]))
say("shard [%d,%d): create %.2f s, equals the whole run's slice bit for bit: %s" % (lo, hi, out["s_trx_create_shard"], out["shard_equals_whole"]))
# ---- a window of the same FILE against the oracle (the loader selects the window's lines from the mapped file)
import oracle_lib as ol
cfg = open(os.path.join(d, "case.cfg")).read().replace("wnlow 333.33", "wnlow 3000").replace("wnhigh 10000", "wnhigh 3100")
open(os.path.join(d, "win.cfg"), "w").write(cfg)
t = time.time(); W = Problem.from_cfg(os.path.join(d, "win.cfg")); out["s_trh_load_window"] = time.time() - t
say("window 3000-3100 cm-1: trh_load %.3f s, %d lines selected" % (out["s_trh_load_window"], int(W.static.nlines)))
ew = Engine(W.static); gw = ew.run(W.atm, W.opts, debug=("last",)); ew.close()
t = time.time(); ora = ol.OracleEngine(W.static); ow = ora.run(W.atm, W.opts, debug=("last",)); out["s_oracle_window"] = time.time() - t
rel = float(np.max(np.abs(gw["spectrum"] / ow["spectrum"] - 1)))
out["window_lines"] = int(W.static.nlines); out["window_vs_oracle_max_rel"] = rel; out["window_last_equal"] = bool(np.array_equal(gw["last"], ow["last"]))
say("window vs oracle: max rel %.3g, toomuch cut equal: %s (oracle %.1f s)" % (rel, out["window_last_equal"], out["s_oracle_window"]))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r4_1e8.json"), "w"), indent=1)
shutil.rmtree(d, ignore_errors=True)
say("done")
