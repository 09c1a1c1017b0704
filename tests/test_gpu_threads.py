"""Several handles computing hinted spectra at the same time from several host threads (the C ABI's
contract: thread-safe per handle, include/transit_hip.h) -- what bench.py's three-handles section and a
retrieval driver with several chains do.  Every thread runs its own handle through the production path
(k_ray_tail: spectrum and flags stored straight into pinned host memory, or the spectrum left in device
memory), a fourth thread creates and destroys handles meanwhile (allocations and frees next to running
kernels); every spectrum must be the single-threaded one, bit for bit."""
import os
import threading

import numpy as np
import pytest

from transit_amd import engine, synth
from transit_amd.engine import Engine
from transit_amd.host import Problem

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("solution", ["eclipse", "transit"])
def test_handles_on_threads_host_and_device_spectra(tmp_path, solution):
    import torch
    d = str(tmp_path / "c")
    synth.make_case(d, nlines=150_000, wnlow=2500, wnhigh=2900, wndelt=1.0, wnosamp=2160, nlayers=100,
                    solution=solution, toomuch=10.0, ethresh=1e-50, seed=21, ncia=2 if solution == "transit" else 1)
    P = Problem.from_cfg(os.path.join(d, "case.cfg"))
    msgs = []
    engine.set_log(lambda lvl, m: msgs.append(m), 5)
    ref_e = Engine(P.static)
    ref_e.run(P.atm, P.opts)                                  # unhinted
    ref = ref_e.run(P.atm, P.opts)["spectrum"].copy()         # hinted: the tail
    engine.set_log(None)
    assert any("ray tail over" in m for m in msgs), "the production path of this test is the ray tail"
    ref_e.close()

    nth, nrun = 3, 40
    engs = [Engine(P.static) for _ in range(nth)]
    for e in engs:
        e.run(P.atm, P.opts)
    bad, errs = [], []

    def worker(k):
        try:
            e = engs[k]
            dev = torch.zeros(P.nwn, dtype=torch.float64, device="cuda:0")
            for i in range(nrun):
                if (i + k) % 2 == 0:
                    got = e.run(P.atm, P.opts)["spectrum"]                   # spectrum into host memory
                else:
                    dev.fill_(-1.0)
                    torch.cuda.synchronize()
                    e.run_device(P.atm, P.opts, dev.data_ptr())               # spectrum stays on the device
                    got = dev.cpu().numpy()
                if not np.array_equal(got, ref):
                    bad.append((k, i, float(np.max(np.abs(got - ref)))))
        except Exception as ex:                                              # noqa: BLE001
            errs.append(repr(ex))

    def churn():
        try:
            for _ in range(6):
                e = Engine(P.static)
                got = e.run(P.atm, P.opts)["spectrum"]
                again = e.run(P.atm, P.opts)["spectrum"]
                if not (np.array_equal(got, ref) and np.array_equal(again, ref)):
                    bad.append(("churn", 0, 0.0))
                e.close()
        except Exception as ex:                                              # noqa: BLE001
            errs.append(repr(ex))

    th = [threading.Thread(target=worker, args=(k,)) for k in range(nth)] + [threading.Thread(target=churn)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for e in engs:
        e.close()
    assert not errs, errs
    assert not bad, bad[:5]
