// transit_main.cpp -- `transit_hip`: command-line drop-in for the reference's
// `transit` binary on the spectrum path (transit/src/transit.c:233-242):
//
//     transit_hip -c run.cfg [--option value ...] [--gpus N]
//
// Same options, same cfg grammar, same TLI / atmosphere / CIA / molecule files
// in, same spectrum (and toomuch, intensity, dump, detail, sampling) files out.
// transit_init() -> trh_load(), do_transit() -> trx_create() + trx_run() on the GPU,
// free_memory() -> trx_destroy() + trh_free().
//
// --gpus N (the one option the reference does not have): the wavenumber axis is cut into N
// shards of equal work (trh_shard_bounds), one handle per shard, one host thread per handle,
// one GPU per thread; the shards need nothing from each other while they run and the spectrum
// slices are collected by ONE ncclAllGather at the end (trx_gather_host; the communicator is
// made with trx_comm_unique_id / trx_comm_create).  With fewer than N devices the ranks share
// devices and the slices are joined in host memory instead (RCCL refuses two ranks on one
// device): same code path otherwise, for rehearsal.  Files are written once, by rank 0.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "transit_hip.h"
#include "transit_host.h"

static double now_s()
{ return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

namespace {

struct Rank {
  int rank = 0, device = 0;
  trx_static st{};                   // this rank's copy: shard, device, communicator
  trx_handle *h = nullptr;
  int rc = TRX_OK; std::string err;
  std::vector<double> slice, tau, e, ecs, intens, er, es, ec; std::vector<int64_t> last; std::vector<uint8_t> comp;
  trx_stats stats{};
};

}  // namespace

int main(int argc, char **argv)
{
  // three busy queues per handle; next to an RCCL communicator the runtime's default of 4 hardware
  // queues per device makes two of them share one.  Read by the HIP
  // runtime when it initialises, which nothing has made it do yet.
  setenv("GPU_MAX_HW_QUEUES", "8", 0);
  char err[512] = {0};
  trh_problem *P = nullptr;
  const double t_start = now_s();
  double t0 = t_start;
  // --gpus N is ours: take it out before the reference's option table sees the line
  int ngpus = 1;
  std::vector<char *> args;
  for (int i = 0; i < argc; i++) {
    if (std::strcmp(argv[i], "--gpus") == 0 && i + 1 < argc) { ngpus = std::atoi(argv[++i]); continue; }
    if (std::strncmp(argv[i], "--gpus=", 7) == 0) { ngpus = std::atoi(argv[i] + 7); continue; }
    args.push_back(argv[i]);
  }
  if (ngpus < 1) { std::fprintf(stderr, "transit_hip: --gpus needs a positive number\n"); return EXIT_FAILURE; }
  // a one-shot run uses the first N devices: keep the runtime from opening the node's other GPUs (agents,
  // memory pools and queues of all 8 are set up otherwise, whichever one the run uses).  The user's own
  // choice of visible devices stands.
  if (!std::getenv("ROCR_VISIBLE_DEVICES") && !std::getenv("HIP_VISIBLE_DEVICES") && !std::getenv("CUDA_VISIBLE_DEVICES")) {
    std::string vis;
    for (int k = 0; k < ngpus; k++) vis += (k ? "," : "") + std::to_string(k);
    setenv("ROCR_VISIBLE_DEVICES", vis.c_str(), 0);        // (entries past the node's last device are ignored)
  }
  int rc = trh_load((int)args.size(), args.data(), &P, err, sizeof(err));
  if (rc == 1) return EXIT_SUCCESS;                       // --help / --version (argum.c:582-607)
  if (rc != TRX_OK) { std::fprintf(stderr, "transit_hip: %s (%s)\n", err, trx_strerror(rc)); return EXIT_FAILURE; }
  const char *verb = trh_option(P, "verb");
  const int verblevel = verb ? std::atoi(verb) : 2;
  if (verblevel >= 2)                                     // TOUT_WARN is level 2, TOUT_INFO level 3 (flags_tr.h:181-185)
    for (const char *m = trh_messages(P); m && *m; ) {
      const char *e = std::strchr(m, '\n'); const size_t n = e ? (size_t)(e - m) : std::strlen(m);
      if (m[0] == 'W' || verblevel >= 3) std::fprintf(stderr, "transit_hip: %s: %.*s\n", m[0] == 'W' ? "warning" : "note", (int)(n > 3 ? n - 3 : 0), m + 3);
      m = e ? e + 1 : m + n;
    }
  if (verblevel > 3) std::printf("Check point: 00 - 04 inputs read and sampled:  dt = %.4f sec.\n\n", now_s() - t0);

  const int64_t nwn = trh_nwn(P);
  const int nr = trh_atm(P)->nlayer;
  if (ngpus > nwn) { std::fprintf(stderr, "transit_hip: more GPUs than wavenumbers\n"); trh_free(P); return EXIT_FAILURE; }

  // --opacityfile names a file that does not exist yet: build the grid on ONE GPU
  // (calcopacity, opacity.c:282-427), write it, and go on with it as the reference does
  t0 = now_s();
  if (trh_needs_opacity_build(P)) {
    trx_handle *h = nullptr;
    rc = trx_create(trh_static(P), &h);
    if (rc != TRX_OK) { std::fprintf(stderr, "transit_hip: trx_create failed: %s\n", trx_strerror(rc)); trh_free(P); return EXIT_FAILURE; }
    int32_t nv = 0, nslot = 0; const double *gt, *gd, *gz; const int32_t *gs;
    trh_grid_request(P, &nv, &gt, &gd, &gz, &nslot, &gs);
    std::vector<double> grid((size_t)nv * nslot * nwn);
    rc = trx_sweep_permol(h, nv, gt, gd, gz, trh_opts(P)->ethresh, nslot, gs, grid.data());
    if (rc == TRX_OK) rc = trh_install_opacity(P, grid.data());
    if (rc != TRX_OK) {
      std::fprintf(stderr, "transit_hip: opacity-grid build failed: %s (%s)\n", trx_strerror(rc), trx_last_error(h));
      trx_destroy(h); trh_free(P); return EXIT_FAILURE;
    }
    if (verblevel > 3) std::printf("Check point: 00 - 05 opacity grid (%d states x %d molecules):  dt = %.4f sec.\n\n", nv, nslot, now_s() - t0);
    trx_destroy(h);
  }
  if (trh_option(P, "justOpacity")) { trh_free(P); return EXIT_SUCCESS; }   // transit.c:133-136

  // ---- what the run has to hand back
  const bool want_toomuch = trh_option(P, "outtoomuch") != nullptr;
  const char *sf = trh_option(P, "savefiles");
  const bool want_dumps = sf && std::strncmp(sf, "yes", 3) == 0;                // argum.c:461-470
  const bool det_tau = trh_wants_detail(P, 0), det_ext = trh_wants_detail(P, 1), det_cia = trh_wants_detail(P, 2);
  const bool want_intens = trh_option(P, "outintens") != nullptr && trh_opts(P)->solution == TRX_SOL_ECLIPSE;
  // --saveext: the extinction of an earlier run back in (restfile_extinct, tau.c:155-156), this run's out (tau.c:340-341)
  const bool want_saveext = trh_option(P, "saveext") != nullptr;
  std::vector<double> ext_in; std::vector<uint8_t> ext_flags;
  bool restored = false;
  if (want_saveext) {
    ext_in.resize((size_t)nwn * nr); ext_flags.assign((size_t)nr, 0);
    restored = trh_saveext_read(P, ext_in.data(), ext_flags.data()) == TRX_OK;
    if (!restored && verblevel >= 2) std::fprintf(stderr, "transit_hip: note: no extinction restored from '%s'\n", trh_option(P, "saveext"));
  }
  const bool need_tau = want_toomuch || want_dumps || det_tau || det_ext, need_e = want_dumps || det_ext || want_saveext, need_ecs = want_dumps || det_cia;
  const int nang = trh_opts(P)->nangles;
  if (want_dumps && trh_write_sample(P, nullptr) != TRX_OK)                     // makesample.c:598-599
    std::fprintf(stderr, "transit_hip: cannot write the sampling file\n");

  // ---- shards, devices, communicator
  std::vector<int64_t> bounds((size_t)ngpus + 1);
  if (trh_shard_bounds(P, ngpus, bounds.data()) != TRX_OK) { std::fprintf(stderr, "transit_hip: cannot cut %d shards\n", ngpus); trh_free(P); return EXIT_FAILURE; }
  t0 = now_s();
  const int ndev = trx_device_count();                     // (the process's first HIP call: the runtime comes up here)
  if (verblevel > 3) std::printf("Check point: 00 - 06 GPU runtime up, %d device(s) visible:  dt = %.4f sec.\n\n", ndev, now_s() - t0);
  const bool rccl = ngpus > 1 && ndev >= ngpus;
  unsigned char comm_id[TRX_COMM_ID_BYTES];
  if (rccl && (rc = trx_comm_unique_id(comm_id)) != TRX_OK) {
    std::fprintf(stderr, "transit_hip: no RCCL communicator: %s\n", trx_strerror(rc)); trh_free(P); return EXIT_FAILURE;
  }
  if (ngpus > 1 && !rccl && verblevel >= 2)
    std::fprintf(stderr, "transit_hip: note: %d ranks on %d device(s): devices are shared and the slices are joined in host memory\n", ngpus, ndev);
  int64_t count = 0;                                       // slice length every rank contributes
  for (int k = 0; k < ngpus; k++) count = std::max<int64_t>(count, bounds[k + 1] - bounds[k]);

  t0 = now_s();
  std::vector<Rank> R((size_t)ngpus);
  std::vector<double> gathered((size_t)count * ngpus);    // rank 0's copy of the gather
  // Three phases, every rank's status checked between them: a collective (the communicator's
  // start-up, the gather) is entered only when EVERY rank got that far -- a rank that failed alone
  // would leave the others waiting in it for ever.
  //   1 communicator (collective: ncclCommInitRank)     2 create + run (no exchange: a rank may fail alone)
  //   3 gather (collective: ncclAllGather)
  // TRANSIT_HIP_FAIL_RANK=k (tests): rank k reports a failure in phase 2 without running.
  const char *fail_env = std::getenv("TRANSIT_HIP_FAIL_RANK");
  const int fail_rank = fail_env ? std::atoi(fail_env) : -1;
  auto in_parallel = [&](auto &&phase) {
    if (ngpus == 1) { phase(0); return; }
    std::vector<std::thread> th;
    for (int k = 0; k < ngpus; k++) th.emplace_back(phase, k);
    for (auto &t : th) t.join();
  };
  auto cleanup = [&](bool failed) {
    for (auto &r : R) {
      void *c = r.st.comm;
      if (r.h) trx_destroy(r.h);
      if (c) { if (failed) trx_comm_abort(c); else trx_comm_destroy(c); }     // (after a failure: no collective teardown either)
    }
    trh_free(P);
  };
  auto all_ok = [&](const char *phase) {
    bool ok = true;
    for (auto &r : R)
      if (r.rc != TRX_OK) {
        std::fprintf(stderr, "transit_hip: rank %d: %s failed: %s\n", r.rank, r.err.c_str(), trx_strerror(r.rc));
        ok = false;
      }
    if (!ok && ngpus > 1) std::fprintf(stderr, "transit_hip: stopping after the %s phase: no rank enters the next one\n", phase);
    return ok;
  };
  auto setup = [&](int k) {
    Rank &r = R[(size_t)k];
    r.rank = k; r.device = ndev > 0 ? k % ndev : 0;
    r.st = *trh_static(P);
    r.st.wn_lo = bounds[k]; r.st.wn_hi = bounds[k + 1]; r.st.device = r.device;
    r.st.comm = nullptr; r.st.nranks = 1; r.st.rank = 0;
    if (rccl) {
      void *c = nullptr;
      if ((r.rc = trx_comm_create(comm_id, ngpus, k, r.device, &c)) != TRX_OK) { r.err = std::string("trx_comm_create: ") + trx_last_error(nullptr); return; }
      r.st.comm = c; r.st.nranks = ngpus; r.st.rank = k;
    }
  };
  auto work = [&](int k) {
    Rank &r = R[(size_t)k];
    if (k == fail_rank) { r.rc = TRX_E_HIP; r.err = "trx_run (TRANSIT_HIP_FAIL_RANK)"; return; }
    if ((r.rc = trx_create(&r.st, &r.h)) != TRX_OK) { r.err = "trx_create"; return; }
    const int64_t n = bounds[k + 1] - bounds[k];
    if (restored) {                                        // this shard's columns of the restored rows
      std::vector<double> part((size_t)n * nr);
      for (int l = 0; l < nr; l++) std::memcpy(&part[(size_t)l * n], &ext_in[(size_t)l * nwn + bounds[k]], sizeof(double) * (size_t)n);
      if ((r.rc = trx_restore_extinction(r.h, nr, part.data(), ext_flags.data())) != TRX_OK) { r.err = "trx_restore_extinction"; return; }
    }
    r.slice.assign((size_t)count, 0.0);
    trx_debug dbg{};
    if (need_tau) { r.tau.resize((size_t)n * nr); r.last.resize((size_t)n); dbg.tau = r.tau.data(); dbg.last = r.last.data(); }
    if (need_e) { r.e.resize((size_t)n * nr); dbg.e = r.e.data(); }
    if (want_saveext) { r.comp.assign((size_t)nr, 0); dbg.computed = r.comp.data(); }
    if (need_ecs) { r.ecs.resize((size_t)n * nr); dbg.e_cs = r.ecs.data(); }
    if (want_intens) { r.intens.resize((size_t)n * nang); dbg.intens = r.intens.data(); }
    trx_opts opts = *trh_opts(P);
    if (want_dumps) {                                      // the dump writers redo the reference's laziness from `last`
      opts.eager = 1;
      r.er.resize((size_t)n * nr); r.es.resize((size_t)n * nr); r.ec.resize((size_t)n * nr);
      dbg.er = r.er.data(); dbg.e_scat = r.es.data(); dbg.e_cloud = r.ec.data();
    }
    r.rc = trx_run(r.h, trh_atm(P), &opts, r.slice.data(), (dbg.tau || dbg.e || dbg.e_cs || dbg.intens) ? &dbg : nullptr);
    if (r.rc != TRX_OK) { r.err = std::string("trx_run: ") + trx_last_error(r.h); return; }
    trx_get_stats(r.h, &r.stats);
  };
  auto gather = [&](int k) {                               // the one exchange: all slices to every rank
    Rank &r = R[(size_t)k];
    std::vector<double> all((size_t)count * ngpus);
    r.rc = trx_gather_host(r.h, r.slice.data(), all.data(), count);
    if (r.rc != TRX_OK) { r.err = std::string("trx_gather: ") + trx_last_error(r.h); return; }
    if (k == 0) gathered = all;
  };
  in_parallel(setup);
  if (!all_ok("communicator")) { cleanup(true); return EXIT_FAILURE; }
  in_parallel(work);
  if (!all_ok("spectrum")) { cleanup(true); return EXIT_FAILURE; }
  if (rccl) {
    in_parallel(gather);
    if (!all_ok("gather")) { cleanup(true); return EXIT_FAILURE; }
  }
  if (!rccl) for (int k = 0; k < ngpus; k++) std::memcpy(&gathered[(size_t)k * count], R[(size_t)k].slice.data(), sizeof(double) * (size_t)count);

  // ---- rank 0 stitches and writes
  std::vector<double> spectrum((size_t)nwn);
  for (int k = 0; k < ngpus; k++)
    std::memcpy(&spectrum[(size_t)bounds[k]], &gathered[(size_t)k * count], sizeof(double) * (size_t)(bounds[k + 1] - bounds[k]));
  std::vector<double> tau, e, ecs, intens, er, es, ec; std::vector<int64_t> last;
  if (want_dumps) { er.resize((size_t)nwn * nr); es.resize((size_t)nwn * nr); ec.resize((size_t)nwn * nr); }
  if (need_tau) { tau.resize((size_t)nwn * nr); last.resize((size_t)nwn); }
  if (need_e) e.resize((size_t)nwn * nr);
  if (need_ecs) ecs.resize((size_t)nwn * nr);
  if (want_intens) intens.resize((size_t)nwn * nang);
  for (int k = 0; k < ngpus; k++) {                       // debug layouts: tau [wn][height], e/e_cs [layer][wn], intens [angle][wn]
    const Rank &r = R[(size_t)k];
    const int64_t lo = bounds[k], n = bounds[k + 1] - bounds[k];
    if (need_tau) { std::memcpy(&tau[(size_t)lo * nr], r.tau.data(), sizeof(double) * (size_t)n * nr); std::memcpy(&last[(size_t)lo], r.last.data(), sizeof(int64_t) * (size_t)n); }
    for (int l = 0; l < nr; l++) {
      if (need_e) std::memcpy(&e[(size_t)l * nwn + lo], &r.e[(size_t)l * n], sizeof(double) * (size_t)n);
      if (need_ecs) std::memcpy(&ecs[(size_t)l * nwn + lo], &r.ecs[(size_t)l * n], sizeof(double) * (size_t)n);
      if (want_dumps) {
        std::memcpy(&er[(size_t)l * nwn + lo], &r.er[(size_t)l * n], sizeof(double) * (size_t)n);
        std::memcpy(&es[(size_t)l * nwn + lo], &r.es[(size_t)l * n], sizeof(double) * (size_t)n);
        std::memcpy(&ec[(size_t)l * nwn + lo], &r.ec[(size_t)l * n], sizeof(double) * (size_t)n);
      }
    }
    for (int a = 0; a < nang && want_intens; a++) std::memcpy(&intens[(size_t)a * nwn + lo], &r.intens[(size_t)a * n], sizeof(double) * (size_t)n);
  }
  if (want_saveext) {
    // a layer is in the file when EVERY shard swept it or it came out of the file already (whose row it keeps)
    std::vector<uint8_t> flags((size_t)nr, 1);
    for (int l = 0; l < nr; l++) {
      for (auto &r : R) flags[(size_t)l] &= r.comp[(size_t)l];
      if (restored && ext_flags[(size_t)l]) { flags[(size_t)l] = 1; std::memcpy(&e[(size_t)l * nwn], &ext_in[(size_t)l * nwn], sizeof(double) * (size_t)nwn); }
      if (!flags[(size_t)l]) std::fill(e.begin() + (size_t)l * nwn, e.begin() + (size_t)(l + 1) * nwn, 0.0);
    }
    if (trh_saveext_write(P, e.data(), flags.data()) != TRX_OK) std::fprintf(stderr, "transit_hip: cannot write the extinction savefile\n");
  }
  if (need_e && need_tau) {                                // the run may have swept deeper than the deepest ray: give the rows
    int64_t deep = 0;                                      // below it back the zeros the reference's lazy sweep leaves there
    for (int64_t w = 0; w < nwn; w++) deep = std::max(deep, last[(size_t)w]);
    std::fill(e.begin(), e.begin() + (size_t)(nr - 1 - deep) * nwn, 0.0);
  }
  if (verblevel > 3) {
    long long inr = 0, nadd = 0, swept = 0; double dev_ms = 0;
    for (auto &r : R) { inr += r.stats.nlines_inrange; nadd = std::max<long long>(nadd, r.stats.nadd); swept = std::max<long long>(swept, r.stats.layers_swept); dev_ms = std::max(dev_ms, r.stats.ms_run_total); }
    std::printf("Check point: 00 - 14 opacity + spectrum on %d GPU%s (Voigt table, line list, CIA + line sweep + optical depth + %s):  dt = %.4f sec.\n"
                "  lines in range %lld, co-added %lld, layers swept %lld of %d, device time of the spectrum %.3f ms\n\n",
                ngpus, ngpus > 1 ? "s" : "", trh_opts(P)->solution == TRX_SOL_ECLIPSE ? "intensities + flux" : "modulation",
                now_s() - t0, inr, nadd, swept, nr, dev_ms);
  }
  if (want_toomuch) trh_write_toomuch(P, tau.data(), last.data(), nullptr);
  if (want_intens) trh_write_intens(P, intens.data(), nullptr);
  if (want_dumps && (trh_write_dumps_masked(P, e.data(), ecs.data(), tau.data(), last.data(), nullptr) != TRX_OK ||
                     trh_write_ext_dumps(P, e.data(), ecs.data(), last.data(), er.data(), es.data(), ec.data(), nullptr) != TRX_OK))
    std::fprintf(stderr, "transit_hip: cannot write the savefiles dumps\n");
  if ((det_tau && trh_write_detail(P, 0, tau.data()) != TRX_OK) || (det_ext && trh_write_detail(P, 1, e.data()) != TRX_OK) ||
      (det_cia && trh_write_detail(P, 2, ecs.data()) != TRX_OK))
    std::fprintf(stderr, "transit_hip: cannot write a detail file\n");
  rc = trh_write_spectrum(P, spectrum.data(), nullptr);
  if (rc != TRX_OK) std::fprintf(stderr, "transit_hip: cannot write the spectrum file\n");
  if (verblevel > 3) std::printf("Check point: 00 - 15 outputs written:  wall since start = %.4f sec.\n", now_s() - t_start);
  cleanup(false);
  return rc == TRX_OK ? EXIT_SUCCESS : EXIT_FAILURE;
}
