// transit_main.cpp -- `transit_hip`: command-line drop-in for the reference's
// `transit` binary on the spectrum path (transit/src/transit.c:233-242):
//
//     transit_hip -c run.cfg [--option value ...]
//
// Same options, same cfg grammar, same TLI / atmosphere / CIA / molecule files
// in, same spectrum (and toomuch) files out.  transit_init() -> trh_load(),
// do_transit() -> trx_create() + trx_run() on the GPU, free_memory() ->
// trx_destroy() + trh_free().
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "transit_hip.h"
#include <string>
#include "transit_host.h"

static double now_s()
{ return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
  char err[512] = {0};
  trh_problem *P = nullptr;
  const double t_start = now_s();
  double t0 = t_start;
  int rc = trh_load(argc, argv, &P, err, sizeof(err));
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

  t0 = now_s();
  trx_handle *h = nullptr;
  rc = trx_create(trh_static(P), &h);
  if (rc != TRX_OK) { std::fprintf(stderr, "transit_hip: trx_create failed: %s\n", trx_strerror(rc)); trh_free(P); return EXIT_FAILURE; }
  if (verblevel > 3) std::printf("Check point: 00 - 05 opacity (Voigt table on device, line list resident):  dt = %.4f sec.\n\n", now_s() - t0);

  // --opacityfile names a file that does not exist yet: build the grid on the GPU
  // (calcopacity, opacity.c:282-427), write it, and go on with it as the reference does
  if (trh_needs_opacity_build(P)) {
    t0 = now_s();
    int32_t nv = 0, nslot = 0; const double *gt, *gd, *gz; const int32_t *gs;
    trh_grid_request(P, &nv, &gt, &gd, &gz, &nslot, &gs);
    std::vector<double> grid((size_t)nv * nslot * trh_nwn(P));
    rc = trx_sweep_permol(h, nv, gt, gd, gz, trh_opts(P)->ethresh, nslot, gs, grid.data());
    if (rc == TRX_OK) rc = trh_install_opacity(P, grid.data());
    if (rc != TRX_OK) {
      std::fprintf(stderr, "transit_hip: opacity-grid build failed: %s (%s)\n", trx_strerror(rc), trx_last_error(h));
      trx_destroy(h); trh_free(P); return EXIT_FAILURE;
    }
    if (verblevel > 3) std::printf("Check point: 00 - 05 opacity grid (%d states x %d molecules):  dt = %.4f sec.\n\n", nv, nslot, now_s() - t0);
    trx_destroy(h); h = nullptr;
    rc = trx_create(trh_static(P), &h);          // same problem, now in grid mode
    if (rc != TRX_OK) { std::fprintf(stderr, "transit_hip: trx_create failed: %s\n", trx_strerror(rc)); trh_free(P); return EXIT_FAILURE; }
  }
  if (trh_option(P, "justOpacity")) { trx_destroy(h); trh_free(P); return EXIT_SUCCESS; }   // transit.c:133-136

  const int64_t nwn = trh_nwn(P);
  const int nr = trh_atm(P)->nlayer;
  std::vector<double> spectrum((size_t)nwn), tau, e, ecs;
  std::vector<int64_t> last;
  trx_debug dbg{};
  const bool want_toomuch = trh_option(P, "outtoomuch") != nullptr;
  const char *sf = trh_option(P, "savefiles");
  const bool want_dumps = sf && std::strncmp(sf, "yes", 3) == 0;                // argum.c:461-470
  const bool det_tau = trh_wants_detail(P, 0), det_ext = trh_wants_detail(P, 1), det_cia = trh_wants_detail(P, 2);
  if (want_dumps && trh_write_sample(P, nullptr) != TRX_OK)                     // makesample.c:598-599
    std::fprintf(stderr, "transit_hip: cannot write the sampling file\n");
  if (want_toomuch || want_dumps || det_tau) { tau.resize((size_t)nwn * nr); last.resize((size_t)nwn); dbg.tau = tau.data(); dbg.last = last.data(); }
  if (want_dumps || det_ext) { e.resize((size_t)nwn * nr); dbg.e = e.data(); }
  if (want_dumps || det_cia) { ecs.resize((size_t)nwn * nr); dbg.e_cs = ecs.data(); }
  std::vector<double> intens;
  const bool want_intens = trh_option(P, "outintens") != nullptr && trh_opts(P)->solution == TRX_SOL_ECLIPSE;
  if (want_intens) { intens.resize((size_t)nwn * trh_opts(P)->nangles); dbg.intens = intens.data(); }

  t0 = now_s();
  rc = trx_run(h, trh_atm(P), trh_opts(P), spectrum.data(), (dbg.tau || dbg.e || dbg.e_cs || dbg.intens) ? &dbg : nullptr);
  if (rc != TRX_OK) {
    std::fprintf(stderr, "transit_hip: trx_run failed: %s (%s)\n", trx_strerror(rc), trx_last_error(h));
    trx_destroy(h); trh_free(P); return EXIT_FAILURE;
  }
  if (verblevel > 3) {
    trx_stats s{}; trx_get_stats(h, &s);
    std::printf("Check point: 00 - 14 spectrum (CIA + line sweep + optical depth + %s):  dt = %.4f sec.\n"
                "  lines in range %lld, co-added %lld, layers swept %lld of %d, device time %.3f ms\n\n",
                trh_opts(P)->solution == TRX_SOL_ECLIPSE ? "intensities + flux" : "modulation",
                now_s() - t0, (long long)s.nlines_inrange, (long long)s.nadd, (long long)s.layers_swept, nr, s.ms_run_total);
  }
  if (want_toomuch) trh_write_toomuch(P, tau.data(), last.data(), nullptr);
  if (want_intens) trh_write_intens(P, intens.data(), nullptr);
  if (want_dumps && trh_write_dumps(P, e.data(), ecs.data(), tau.data(), nullptr) != TRX_OK)
    std::fprintf(stderr, "transit_hip: cannot write the savefiles dumps\n");
  if ((det_tau && trh_write_detail(P, 0, tau.data()) != TRX_OK) || (det_ext && trh_write_detail(P, 1, e.data()) != TRX_OK) ||
      (det_cia && trh_write_detail(P, 2, ecs.data()) != TRX_OK))
    std::fprintf(stderr, "transit_hip: cannot write a detail file\n");
  rc = trh_write_spectrum(P, spectrum.data(), nullptr);
  if (rc != TRX_OK) std::fprintf(stderr, "transit_hip: cannot write the spectrum file\n");
  if (verblevel > 3) std::printf("Check point: 00 - 15 outputs written:  wall since start = %.4f sec.\n", now_s() - t_start);
  trx_destroy(h);
  trh_free(P);
  return rc == TRX_OK ? EXIT_SUCCESS : EXIT_FAILURE;
}
