// trx_api.hip -- C ABI of include/transit_hip.h on top of the gfx950 kernels.
//
// Host responsibilities (all layer- or geometry-only, O(N) or O(N^2) scalars):
//   * line-list preparation once per handle: wavenumbers, range flags, the
//     greedy co-adding groups (extinction.c:449-462) and their fine-grid
//     indices (:445-447) -- exact reference arithmetic, sequential by nature;
//   * per-run layer prologue: broadening widths, nearest table indices,
//     strength prefactors (extinction.c:364-395, 464, 473);
//   * Simpson weights of the ray geometry (numerical.c:390-425);
//   * the table-only halves of the CIA splines (crosssec.c:272-428), once per handle;
//   * the plan of a run: layers per step, streams, events, the exchanges of a sharded job.
// Everything per (line x layer), per (group x layer x bin) and per
// (wavenumber x layer) runs in the kernels of trx_kernels.hip.h.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <functional>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "transit_hip.h"
#include "trx_kernels.hip.h"
#include "trx_walk.hip.h"
#include "trx_rows.hip.h"
#include "trx_tail.hip.h"
#include "trx_lanes.hip.h"
#include "../trx_groups.h"

using namespace trx;

namespace {

struct DevBuf {
  void *p = nullptr; size_t bytes = 0;
  ~DevBuf() { release(); }
  void release() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
  template <class T> T *as() const { return (T *)p; }
};

// view into another allocation (same accessors as DevBuf, owns nothing)
struct DevView {
  void *p = nullptr;
  template <class T> T *as() const { return (T *)p; }
};

}  // namespace

struct trx_handle {
  int device = 0;
  hipStream_t stream = nullptr, stream2 = nullptr;   // stream2: CIA kernels, overlapped with the first sweep step
  // stream4: the line sweep of step c+1 runs while `stream` integrates the optical depth of
  // step c; ev_ac[c] = extinction of step c complete
  hipStream_t stream4 = nullptr;
  std::vector<hipEvent_t> ev_ac, ev_cb;      // ev_cb[c] = partial records of step c consumed
  hipEvent_t ev_walk1 = nullptr;                     // the first walk of a two-queue run (and the CIA kernels) are done
  hipEvent_t ev_inputs = nullptr, ev_cia = nullptr, ev_join = nullptr, ev_run_a = nullptr, ev_run_b = nullptr;
  std::string err;

  // grids
  double wn_i = 0, wn_d = 0, odwn = 0; int64_t nwn = 0, nown = 0, lo = 0, hi = 0, nsh = 0; int osamp = 1;
  // isotopes / molecules (host copies)
  int niso = 0, nmol = 0;
  std::vector<double> iso_mass, iso_ratio, mol_mass, mol_radius, mol_pol;
  std::vector<double> pair_csd, pair_sqrt;          // [niso][nmol]: r_mol + r_iso's molecule, sqrt(1/m_iso + 1/m_mol) (extinction.c:376-380)
  std::vector<int32_t> iso_imol, mol_is_h2;
  std::vector<double> iso_wmin, iso_wmax;          // anchor wavenumber range per isotope
  // Voigt table
  int ndop = 0, nlor = 0;
  std::vector<double> adop, alor;                   // +1 sentinel
  std::vector<int32_t> psizeT; bool psize_mono = false;   // psize as [nlor][ndop] (prep_layers); no profile narrower than the one a Doppler index below it
  std::vector<double> dopthr;                       // steps of the nearest-index function on adop (build_table; dop_index)
  std::vector<double> lorthr;                       // the same on alor (empty: the grid did not pass the check -- nearest_index is called)
  std::vector<int> guess_dop, guess_lor, guess_a0, guess_a1;      // [niso] where prep_layers found the layer above (its walks start there)
  std::vector<long> guess_npre;                     // [niso] and the layer above's count of groups at or above the refresh cut
  std::vector<double> iso_sqrtm;                    // sqrt(iso_mass)
  std::vector<double> dens_over_m;                  // [nmol] scratch of prep_layers
  std::vector<int32_t> psize; std::vector<long long> poff; int64_t tab_n = 0;
  DevBuf d_adop, d_dopthr, d_e2tab, d_psize, d_poff, d_tab, d_tabT, d_poffT, d_gimod, d_gidiv;
  // both tables carry kTabPad zero floats in front and behind: k_accumulate_wide reads whole
  // 4-float lane segments around a profile row and masks what lies outside the row
  float *tab = nullptr; const float *tabT = nullptr; const long long *poffT = nullptr;
  // the walk's copy: phase-major rows of whole cache lines (walk_row_layout), one WalkProfile per table entry
  DevBuf d_tabW, d_walkprof; const float *tabW = nullptr; bool tabw_ok = false;
  DevBuf d_tabW32, d_wp32; const float *tabW32 = nullptr; unsigned slab32 = 0;      // compact 32-byte rows for frames of 8 bins (k_table_rows32)
  long long row_m8_from = 768;      // profile width (bins) from which a layer's tiles are 512 bins (TRX_ROWS_M8_FROM: measurements)
  std::vector<std::pair<double, double>> recip_ok;     // divisors whose reciprocal quotient_rn may use (checked_reciprocal)
  bool shard_frames = true;                              // frames sized for the Doppler indices the lines in reach can take (TRX_SHARD_FRAMES=0: for the isotope's whole wavenumber range)
  bool cia_window = true;                                // the CIA spline solved for the table rows a run needs, not the whole table (TRX_CIA_WINDOW=0)
  bool cia_sums = true;                                  // the CIA splines' second derivatives as sums per row (TRX_CIA_SUMS=0: the sweeps of k_cia_layers; tests)
  bool cia_segments = true;                              // k_cia_layers in pieces of 128 rows (TRX_CIA_SEGMENTS=0: one sweep per table; tests)
  bool two_queues = true;                                // the second walk of such a run on a queue of its own, next to the first (TRX_TWO_QUEUES=0: behind it)
  bool tail_direct = true;                               // ... which writes spectrum and flags straight into pinned host memory (TRX_TAIL_DIRECT=0: copy commands)
  bool ray_tail = true;                                  // hinted eclipse runs end in k_ray_tail (TRX_RAY_TAIL=0: the step kernels; tests, measurements)
  bool packed_walk = true; int packed_max_layers = 10;   // steps of few layers walk several ranges per wave (TRX_NO_PACKED_WALK, TRX_PACKED_MAX_LAYERS: tests, measurements)
  // steps of at most 32 layers with frames of 8+ bins: lanes = lines for the strengths (trx_lanes.hip.h; TRX_LANES_WALK=0:
  // the one-range / packed forms)
  bool no_binrec = false;                            // TRX_NO_BINREC: k_ray_tail finds a bin's records through the ranges' numbers (A/B, tests)
  int xcd_map = 1;                                   // blocks -> ranges by XCD (xcd_block): bit 0 k_line_walk_lanes, bit 1 k_line_walk (measured: slower there).  TRX_XCD_MAP, A/B
  bool lanes_walk = true, lanes_force = false; int max_gcount = 0; DevBuf d_linebase, d_rinfo;   // (TRX_LANES_WALK=2: also on sparse lists, tests)
  bool no_row_copy = false, no_rows32 = false;
  bool row_staging = true;          // osamp == 1: wide profiles through k_accumulate_rows (TRX_NO_ROW_STAGING at create: tests compare the two forms)
  // lines
  int64_t nlines = 0, ngroups = 0, nadd = 0, ninrange = 0;
  DevBuf d_lgroup, d_wavn, d_elow, d_gf, d_iso, d_inr, d_gfirst, d_gcount, d_giown, d_giso, d_gwavn, d_gblock, d_cntge, d_cntsub;
  int sub_f = 1;                        // sub-buckets per coarse cell of d_cntsub (1: it is d_cntge)
  LinesDev L{};
  HostBuf<double> h_gwavn; std::vector<int32_t> h_gblock; HostBuf<int32_t> h_cntge, h_gfirst, h_gcount;   // host copies for the per-run prologue
  void *comm = nullptr; int nranks = 1, rank = 0;          // RCCL communicator: only trx_gather uses it
  bool windowed() const { return lo > 0 || hi < nwn; }    // a shard sweeps only the lines that can reach it
  // candidates for the layer maximum (k_cand_*): indices into the line arrays; -1 = use every line
  DevBuf d_cand, d_candrec; int64_t ncand = -1;             // candidate line indices / their packed line data
  DevBuf d_kmax;                                            // [2][layer] (runs alternate; k_layer_max) / [layer][nmx] (per-molecule sweeps)
  int kmax_parity = 0, kmax_nr = 0; bool kmax_clean = false;   // clean: the half the next run will use is zero
  // the walk (k_line_walk): one record per line, line ranges of ngw groups, plans per profile reach
  DevBuf d_walk, d_wbase, d_part[2];   // partial records: consecutive steps alternate
  std::vector<int32_t> h_wbase; int nwaves = 0, ngw = 0; bool walk_ok = false;
  bool walk_temp_ok = true;         // this run's layers are all warmer than kWalkMinTemp
  struct Plan { bool built = false; DevBuf blo, bhi, off, binw, binrec; int64_t records = 0; };
  Plan plan[4];                                             // NB = 2, 4, 8, 16 bins per frame
  // CIA (host copies)
  struct Cia { int nspec; int mol[2]; std::vector<double> wn, temp, cs, zt, uw, ruw, rh, wf, wb; DevBuf d_wn, d_temp, d_cs, d_zt, d_uw, d_ruw, d_rh, d_wf, d_wb; };      // wf, wb: weights of k_cia_v / k_cia_z (empty: the sweeps)
  std::vector<Cia> cia;
  DevBuf d_cia_ws;
  // per-run workspaces (grown on demand)
  int ws_nr = 0, ws_chunk = 0;
  DevBuf d_SG, d_idop8, d_sticky, d_part3;
  // per-run inputs: packed into ONE pinned host block and copied with ONE transfer
  // (layer scalars, ray geometry, impact parameters, CIA density products)
  DevBuf d_in; void *h_in = nullptr; size_t h_in_bytes = 0; const double *h_in_dev = nullptr;      // (h_in_dev: the pinned block as the device sees it)
  DevBuf d_pm_f64, d_pm_i32;         // the same scalars of a per-molecule sweep (trx_sweep_permol)
  // what the host reads back after a run, in ONE device block and one pinned host block:
  // flags (8 ints, byte 0), status (4 ints, byte 64), counters (3 per layer, byte 128)
  DevBuf d_xf;                        // scattering / cloud wavenumber factors [2][nsh] + 3 constants
  DevBuf d_small; DevView d_flags, d_status, d_counters; void *h_small = nullptr; size_t h_small_bytes = 0;
  DevBuf d_e, d_ecs, d_er, d_tau, d_last, d_intens, d_spec, d_acc, d_geom;
  // opacity grid (optional)
  bool has_grid = false; long og_nmol = 0, og_ntemp = 0, og_nlayer = 0, og_nwave = 0;
  std::vector<double> og_temp; std::vector<int32_t> og_molidx; DevBuf d_og_o, d_og_layer, d_og_itemp, d_iso_mx, d_pm, d_kmaxpm;
  trx_stats stats{};
  std::vector<double> run_f64, run_geom, run_ipv; std::vector<int32_t> run_i32;      // per-run host arrays (kept: no allocation per run)
  int hint_layers = 0;       // layers the previous run needed (deepest toomuch crossing + 1)
  DevBuf d_e_saved; std::vector<uint8_t> saved;          // trx_restore_extinction: [nlayer][nsh] and the flags (empty: none)
  void *h_spec = nullptr; size_t h_spec_bytes = 0;     // pinned staging of the spectrum (trx_run hands over pageable memory)
  void *h_spec_dev = nullptr, *h_small_dev = nullptr;  // the device's addresses of the pinned blocks (asked for once per allocation, not per run)
  void *h_tailblk = nullptr, *h_tailblk_dev = nullptr; size_t h_tailblk_bytes = 0;   // pinned: what each block of k_ray_tail adds to the run's flags (vertical rays: the host adds them up)
};

namespace {

// process-wide message sink (trx_set_log): the reference's tr_output()/verblevel pair
struct LogSink { trx_log_fn fn = nullptr; void *user = nullptr; int max_level = 0; };
LogSink &log_sink() { static LogSink s; return s; }
void log_msg(int level, const std::string &msg)
{
  const LogSink s = log_sink();
  if (s.fn && level <= s.max_level) s.fn(level, msg.c_str(), s.user);
}

// stage timer of trx_create: logs "create: <stage> x.xx ms" at TRX_LOG_DEBUG
struct StageTimer {
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void lap(const char *what) {
    const auto n = std::chrono::steady_clock::now();
    if (log_sink().fn && log_sink().max_level >= TRX_LOG_DEBUG) {
      char b[128]; std::snprintf(b, sizeof b, "create: %-28s %8.2f ms", what, std::chrono::duration<double, std::milli>(n - t).count());
      log_msg(TRX_LOG_DEBUG, b);
    }
    t = n;
  }
};

int fail(trx_handle *h, int code, const std::string &msg)
{ if (h) h->err = msg; log_msg(TRX_LOG_ERROR, msg); return code; }

#define HIPCHK(h, call)                                                              \
  do { hipError_t e_ = (call);                                                       \
       if (e_ != hipSuccess)                                                         \
         return fail(h, e_ == hipErrorOutOfMemory ? TRX_E_NOMEM : TRX_E_HIP,         \
                     std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

int ensure(trx_handle *h, DevBuf &b, size_t bytes)
{
  if (bytes == 0) bytes = 8;
  if (b.bytes >= bytes) return TRX_OK;
  b.release();
  HIPCHK(h, hipMalloc(&b.p, bytes));
  b.bytes = bytes;
  return TRX_OK;
}

// the small read-back block for nlay layers (device + pinned host mirror)
int ensure_small(trx_handle *h, int nlay)
{
  const size_t bytes = 128 + 24 * (size_t)nlay;
  int rc = ensure(h, h->d_small, bytes);
  if (rc) return rc;
  char *base = (char *)h->d_small.p;
  h->d_flags.p = base; h->d_status.p = base + 64; h->d_counters.p = base + 128;
  if (h->h_small_bytes < bytes) {
    if (h->h_small) (void)hipHostFree(h->h_small);
    h->h_small = nullptr; h->h_small_bytes = 0;
    HIPCHK(h, hipHostMalloc(&h->h_small, bytes, hipHostMallocDefault));
    h->h_small_bytes = bytes;
    HIPCHK(h, hipHostGetDevicePointer(&h->h_small_dev, h->h_small, 0));
  }
  return TRX_OK;
}

template <class T>
int upload_raw(trx_handle *h, DevBuf &b, const T *p, size_t n)
{
  int rc = ensure(h, b, n * sizeof(T));
  if (rc) return rc;
  if (n) HIPCHK(h, hipMemcpyAsync(b.p, p, n * sizeof(T), hipMemcpyHostToDevice, h->stream));
  return TRX_OK;
}

template <class T>
int upload(trx_handle *h, DevBuf &b, const std::vector<T> &v)
{
  int rc = ensure(h, b, v.size() * sizeof(T));
  if (rc) return rc;
  if (!v.empty()) HIPCHK(h, hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, h->stream));
  return TRX_OK;
}

template <class T>
int upload(trx_handle *h, DevBuf &b, const HostBuf<T> &v) { return upload_raw(h, b, v.data(), v.size()); }

// pu/src/iomisc.c:1064-1083 (logspace)
void logspace(double lo, double hi, int n, std::vector<double> &out)
{
  out.resize(n + 1);
  const double l0 = std::log10(lo), l1 = std::log10(hi);
  const double step = (l1 - l0) / (n - 1.0);
  for (int i = 0; i < n; i++) out[i] = std::pow(10, l0 + i * step);
  out[n] = HUGE_VAL;   // the reference searches with hi = n (extinction.c:393-394)
}

// ---- Voigt table plan: opacity.c:219-277 + extinction.c:8-57 ---------------
int plan_table(trx_handle *h, const trx_static *s, std::vector<ProfileJob> &jobs)
{
  const int nd = s->ndop, nl = s->nlor;
  logspace((double)s->dmin, (double)s->dmax, nd, h->adop);
  logspace((double)s->lmin, (double)s->lmax, nl, h->alor);
  h->psize.assign((size_t)nd * nl, 0); h->poff.assign((size_t)nd * nl, 0);
  const double dwn = s->wn_d / s->osamp;
  int64_t total = 0, bins = 0;
  for (int i = 0; i < nd; i++)
    for (int j = 0; j < nl; j++) {
      const size_t k = (size_t)i * nl + j;
      if (h->adop[i] * 10.0 < h->alor[j] && i != 0) {         // opacity.c:262-265
        h->psize[k] = h->psize[k - nl]; h->poff[k] = h->poff[k - nl];
        continue;
      }
      const float dop = (float)h->adop[i], lor = (float)h->alor[j];   // extinction.c:11-12
      double big = dop; if (big < lor) big = lor;
      const double wv = big * s->timesalpha;
      int nv = 2 * (long)(wv / dwn + 0.5) + 1;
      if (nv < 2) nv = 3;
      if (nv > 2 * (int)s->nown) nv = 2 * (int)s->nown + 1;
      if (nv < 0) return fail(h, TRX_E_ARG, "negative Voigt profile size");
      ProfileJob J{};
      J.off = total; J.nv = nv; J.alphaL = lor; J.alphaD = dop;
      J.half = dwn * (long)(nv / 2);
      const bool quick = nv > 99999;                            // voigt.c:109, extinction.c:51
      // voigt.c:399-433
      double step = 2.0 * J.half / (nv - 1);
      int npts = 50; double sub = J.alphaD / (npts - 1);
      if (step < sub || quick) { sub = step; J.regime = quick ? 0 : 1; J.m = 1; }
      else {
        npts = (int)(step / sub) + 1;
        if (npts & 1) npts++;
        J.m = npts; J.regime = 2;
        const long long tot = (long long)nv * npts + 1;
        if (tot > 2000000000LL) return fail(h, TRX_E_UNSUPPORTED, "Voigt sub-sampling exceeds int range");
        sub = 2.0 * J.half / (double)(tot - 1);
      }
      J.sub = sub; J.first_bin = bins;
      jobs.push_back(J);
      h->psize[k] = nv / 2; h->poff[k] = total;
      total += nv; bins += nv;
    }
  h->tab_n = total;
  return TRX_OK;
}

// Environment variables read at trx_create -- ALL of them, in one place.  Each selects between two forms
// of the same computation that give the same bits (the tests named compare them side by side); none is
// needed in production, none changes a result.  (The A/B switches of forms that lost -- run graphs, the
// range size, tile-size tuning beyond what a test forces -- are gone with those forms.)
void test_switches(trx_handle *h)
{
  h->no_row_copy = std::getenv("TRX_NO_ROW_COPY") != nullptr;              // wide frames without the row copy (test_gpu_properties)
  h->no_rows32 = std::getenv("TRX_NO_ROWS32") != nullptr;                  // k_line_walk_lanes<8> on the 64-byte rows (test_gpu_lanes)
  h->row_staging = !std::getenv("TRX_NO_ROW_STAGING");                      // k_accumulate_wide instead of k_accumulate_rows (test_gpu_rows)
  if (const char *v = std::getenv("TRX_ROWS_M8_FROM")) h->row_m8_from = std::atoll(v);      // ... and its tile size per layer (test_gpu_rows)
  h->packed_walk = !std::getenv("TRX_NO_PACKED_WALK");                      // k_line_walk_packed never / for steps of up to N layers (test_gpu_packed)
  if (const char *v = std::getenv("TRX_PACKED_MAX_LAYERS")) h->packed_max_layers = std::max(1, std::min(32, std::atoi(v)));
  if (const char *e = std::getenv("TRX_XCD_MAP")) h->xcd_map = std::atoi(e);
  h->no_binrec = std::getenv("TRX_NO_BINREC") != nullptr;
  if (const char *e = std::getenv("TRX_LANES_WALK")) { h->lanes_walk = std::atoi(e) != 0; h->lanes_force = std::atoi(e) == 2; }      // k_line_walk_lanes never / also on sparse lists (test_gpu_lanes)
  if (const char *e = std::getenv("TRX_RAY_TAIL")) h->ray_tail = std::atoi(e) != 0;            // the step kernels instead of k_ray_tail (test_gpu_tail)
  if (const char *e = std::getenv("TRX_CIA_SUMS")) h->cia_sums = std::atoi(e) != 0;
  if (const char *e = std::getenv("TRX_CIA_SEGMENTS")) h->cia_segments = std::atoi(e) != 0;  // the CIA splines' second derivatives in one sweep per table (test_gpu_cia_window)
  if (const char *e = std::getenv("TRX_TWO_QUEUES")) h->two_queues = std::atoi(e) != 0;       // the walks of a hinted run one behind the other (A/B, tests)
  if (const char *e = std::getenv("TRX_TAIL_DIRECT")) h->tail_direct = std::atoi(e) != 0;      // ... copy commands instead of stores into pinned memory
  if (const char *e = std::getenv("TRX_CIA_WINDOW")) h->cia_window = std::atoi(e) != 0;        // the CIA spline over the whole table (test_gpu_cia_window)
  if (const char *e = std::getenv("TRX_SHARD_FRAMES")) h->shard_frames = std::atoi(e) != 0;    // a shard with the list's frames (test_gpu_shard_frames)
}

int build_table(trx_handle *h, const trx_static *s)
{
  std::vector<ProfileJob> jobs;
  int rc = plan_table(h, s, jobs);
  if (rc) return rc;
  // series coefficients 1/(n!(2n+1)), voigt.c:45-108
  double coef[64]; long double fact = 1.0L;
  for (int n = 0; n < 64; n++) { if (n > 0) fact *= (long double)n; coef[n] = (double)(1.0L / (fact * (long double)(2 * n + 1))); }
  HIPCHK(h, hipMemcpyToSymbolAsync(HIP_SYMBOL(c_voigt_coef), coef, sizeof(coef), 0, hipMemcpyHostToDevice, h->stream));
  DevBuf d_jobs;
  if ((rc = upload(h, d_jobs, jobs))) return rc;
  // (behind the table: kWalkMaxFrame cells of zeros, where the walk's lanes read what a slot does not reach)
  // (and kRowTail more: k_accumulate_rows stages whole 256-float pieces of a row, trx_rows.hip.h)
  const size_t tab_alloc = (size_t)h->tab_n + 2 * kTabPad + (size_t)kWalkMaxFrame * (size_t)std::min<int64_t>(s->osamp, 1 << 21) + kRowTail;
  if ((rc = ensure(h, h->d_tab, sizeof(float) * tab_alloc))) return rc;
  HIPCHK(h, hipMemsetAsync(h->d_tab.p, 0, sizeof(float) * tab_alloc, h->stream));
  h->tab = h->d_tab.as<float>() + kTabPad;
  std::vector<int32_t> ps32(h->psize.begin(), h->psize.end());
  if ((rc = upload(h, h->d_psize, ps32))) return rc;
  if ((rc = upload(h, h->d_poff, h->poff))) return rc;
  if ((rc = upload(h, h->d_adop, h->adop))) return rc;
  {
    // exact steps of the nearest-index function on the Doppler grid (index_from in the
    // kernels): thr[k] = smallest double v with nearest_index(adop, v, 0, ndop) >= k,
    // found by bisection on the bit patterns of the (positive) doubles
    const int nd = s->ndop;
    std::vector<double> thr((size_t)nd + 1);
    thr[0] = -HUGE_VAL; thr[nd] = HUGE_VAL;
    auto idx = [&](double v) { return nearest_index(h->adop.data(), v, 0, nd); };
    for (int k = 1; k < nd; k++) {
      uint64_t a, b; double x;
      std::memcpy(&a, &h->adop[k - 1], 8); std::memcpy(&b, &h->adop[k], 8);      // idx(a) < k <= idx(b)
      while (b - a > 1) {
        const uint64_t m = a + (b - a) / 2;
        std::memcpy(&x, &m, 8);
        if (idx(x) >= k) b = m; else a = m;
      }
      std::memcpy(&thr[k], &b, 8);
      if (idx(thr[k]) != k || idx(std::nextafter(thr[k], 0.0)) != k - 1)
        return fail(h, TRX_E_ARG, "Doppler-width grid is not strictly increasing");
    }
    if ((rc = upload(h, h->d_dopthr, thr))) return rc;
    h->dopthr = thr;
    {   // the Lorentz grid's steps, for the host's prologue (prep_layers): the same bisection; a grid that fails the check keeps nearest_index
      const int nl = s->nlor;
      std::vector<double> lt((size_t)nl + 1);
      lt[0] = -HUGE_VAL; lt[nl] = HUGE_VAL;
      auto lidx = [&](double v) { return nearest_index(h->alor.data(), v, 0, nl); };
      bool ok = true;
      for (int k = 1; k < nl && ok; k++) {
        uint64_t a, b; double x;
        if (!(h->alor[k - 1] > 0) || !(h->alor[k] > h->alor[k - 1])) { ok = false; break; }
        std::memcpy(&a, &h->alor[k - 1], 8); std::memcpy(&b, &h->alor[k], 8);
        while (b - a > 1) {
          const uint64_t m = a + (b - a) / 2;
          std::memcpy(&x, &m, 8);
          if (lidx(x) >= k) b = m; else a = m;
        }
        std::memcpy(&lt[k], &b, 8);
        ok = lidx(lt[k]) == k && lidx(std::nextafter(lt[k], 0.0)) == k - 1;
      }
      if (ok) h->lorthr = lt; else h->lorthr.clear();
    }
    h->psizeT.resize((size_t)nd * h->nlor); h->psize_mono = true;
    for (int d = 0; d < nd; d++)
      for (int l = 0; l < h->nlor; l++) {
        h->psizeT[(size_t)l * nd + d] = h->psize[(size_t)d * h->nlor + l];
        if (d > 0 && h->psize[(size_t)d * h->nlor + l] < h->psize[(size_t)(d - 1) * h->nlor + l]) h->psize_mono = false;
      }
    std::vector<double> e2(64);                       // 2^(j/64) for exp_neg (kernels)
    for (int j = 0; j < 64; j++) e2[j] = (double)exp2l((long double)j / 64.0L);
    if ((rc = upload(h, h->d_e2tab, e2))) return rc;
  }
  hipEvent_t e0, e1;
  HIPCHK(h, hipEventCreate(&e0)); HIPCHK(h, hipEventCreate(&e1));
  HIPCHK(h, hipEventRecord(e0, h->stream));
  const int m_limit = 64;
  for (size_t j0 = 0; j0 < jobs.size(); j0 += 32768) {
    const int nj = (int)std::min<size_t>(32768, jobs.size() - j0);
    int maxnv = 0; bool any_wave = false;
    for (int j = 0; j < nj; j++) { maxnv = std::max(maxnv, jobs[j0 + j].nv); any_wave |= (jobs[j0 + j].regime == 2 && jobs[j0 + j].m > m_limit); }
    const int gx = std::max(1, std::min(64, (maxnv + 255) / 256));
    hipLaunchKernelGGL(k_voigt_bins, dim3(gx, nj), dim3(256), 0, h->stream,
                       d_jobs.as<ProfileJob>() + j0, h->tab, m_limit);
    if (any_wave)
      hipLaunchKernelGGL(k_voigt_bins_wave, dim3(std::max(1, std::min(256, maxnv)), nj), dim3(64), 0, h->stream,
                         d_jobs.as<ProfileJob>() + j0, h->tab, m_limit);
  }
  // phase-major copy for the wide-profile kernel (identical layout when osamp == 1)
  if (s->osamp == 1) { h->tabT = h->tab; h->poffT = h->d_poff.as<long long>(); }
  else {
    std::vector<long long> joffT(jobs.size()), poffT((size_t)s->ndop * s->nlor, 0);
    long long totT = 0;
    for (size_t j = 0; j < jobs.size(); j++) { joffT[j] = totT; totT += (long long)s->osamp * ((jobs[j].nv - 1) / s->osamp + 1); }
    {   // table entries -> job (aliases share the job of the row above)
      size_t j = 0;
      for (int i = 0; i < s->ndop; i++)
        for (int k = 0; k < s->nlor; k++) {
          const size_t e = (size_t)i * s->nlor + k;
          if (h->adop[i] * 10.0 < h->alor[k] && i != 0) poffT[e] = poffT[e - s->nlor];
          else poffT[e] = joffT[j++];
        }
    }
    DevBuf d_joffT;
    if ((rc = upload(h, d_joffT, joffT)) || (rc = upload(h, h->d_poffT, poffT))) return rc;
    if ((rc = ensure(h, h->d_tabT, sizeof(float) * ((size_t)totT + 2 * kTabPad)))) return rc;
    HIPCHK(h, hipMemsetAsync(h->d_tabT.p, 0, sizeof(float) * ((size_t)totT + 2 * kTabPad), h->stream));
    for (size_t j0 = 0; j0 < jobs.size(); j0 += 32768) {
      const int nj = (int)std::min<size_t>(32768, jobs.size() - j0);
      hipLaunchKernelGGL(k_table_phase_major, dim3(32, nj), dim3(256), 0, h->stream, d_jobs.as<ProfileJob>() + j0,
                         d_joffT.as<long long>() + j0, h->tab, h->d_tabT.as<float>() + kTabPad, s->osamp, 0);
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->tabT = h->d_tabT.as<float>() + kTabPad; h->poffT = h->d_poffT.as<long long>();
  }
  // The walk's copy (trx_walk.hip.h): phase-major rows, each between zeros (walk_row_layout) -- the
  // bins of a frame are CONSECUTIVE entries of one row, and what a narrow profile does not reach
  // is zero by position (the pad behind a row is also the pad in front of the next).  One
  // descriptor per table entry.  32-bit byte offsets: no copy when it would pass 4 GB (walk_chunk).
  {
    std::vector<long long> joffW(jobs.size());
    long long totW = 0;
    for (size_t j = 0; j < jobs.size(); j++) {
      int front, stride; walk_row_layout((jobs[j].nv - 1) / s->osamp + 1, front, stride);
      joffW[j] = totW; totW += (long long)s->osamp * stride;
    }
    test_switches(h);
    h->tabw_ok = 4 * (totW + 2 * (long long)kTabPad) < (1LL << 32) && !h->no_row_copy;
    if (h->tabw_ok) {
      std::vector<WalkProfile> desc((size_t)s->ndop * s->nlor);
      size_t j = 0;
      for (int i = 0; i < s->ndop; i++)
        for (int k = 0; k < s->nlor; k++) {
          const size_t e = (size_t)i * s->nlor + k;
          if (h->adop[i] * 10.0 < h->alor[k] && i != 0) { desc[e] = desc[e - s->nlor]; continue; }
          const long long ps = h->psize[e], K = (2 * ps) / s->osamp + 1;
          int front, stride; walk_row_layout((int)K, front, stride);
          desc[e].centre4 = (uint32_t)(4 * (joffW[j++] + front + ps / s->osamp));
          desc[e].rowb = (int32_t)(4 * stride);
          desc[e].psr = (int32_t)(ps % s->osamp);
          desc[e].ps = (int32_t)ps;
        }
      DevBuf d_joffW;
      if ((rc = upload(h, d_joffW, joffW)) || (rc = upload(h, h->d_walkprof, desc))) return rc;
      if ((rc = ensure(h, h->d_tabW, sizeof(float) * ((size_t)totW + 2 * kTabPad)))) return rc;
      HIPCHK(h, hipMemsetAsync(h->d_tabW.p, 0, sizeof(float) * ((size_t)totW + 2 * kTabPad), h->stream));
      for (size_t j0 = 0; j0 < jobs.size(); j0 += 32768) {
        const int nj = (int)std::min<size_t>(32768, jobs.size() - j0);
        hipLaunchKernelGGL(k_table_phase_major, dim3(32, nj), dim3(256), 0, h->stream, d_jobs.as<ProfileJob>() + j0,
                           d_joffW.as<long long>() + j0, h->tab, h->d_tabW.as<float>() + kTabPad, s->osamp, 1);
      }
      // compact rows (32 bytes) of the profiles with at most 8 entries per row: what k_line_walk_lanes<8> gathers
      {
        std::vector<long long> joff32(jobs.size(), -1);
        std::vector<uint32_t> c32((size_t)s->ndop * s->nlor, 0xffffffffu);
        long long nq = 0; size_t jj = 0;                 // profiles with compact rows: [phase][profile][8 floats]
        for (int i = 0; i < s->ndop; i++)
          for (int k = 0; k < s->nlor; k++) {
            const size_t e = (size_t)i * s->nlor + k;
            if (h->adop[i] * 10.0 < h->alor[k] && i != 0) { c32[e] = c32[e - s->nlor]; continue; }
            const long long ps = h->psize[e], K = (2 * ps) / s->osamp + 1;
            if (K <= 8) { joff32[jj] = 8 * nq; c32[e] = (uint32_t)(32 * nq); nq++; }
            jj++;
          }
        const long long tot32 = nq * 8 * (long long)s->osamp;
        // (a slab below 2^24 bytes and the phase below 2^24: the kernel's 24-bit multiply; the whole below 4 GB)
        if (nq > 0 && 32 * nq < (1LL << 24) && s->osamp < (1 << 24) && 4 * tot32 < (1LL << 32) && !h->no_rows32) {
          DevBuf d_joff32;
          if ((rc = upload(h, d_joff32, joff32)) || (rc = upload(h, h->d_wp32, c32))) return rc;
          if ((rc = ensure(h, h->d_tabW32, sizeof(float) * (size_t)tot32))) return rc;
          for (size_t j0 = 0; j0 < jobs.size(); j0 += 32768) {
            const int nj = (int)std::min<size_t>(32768, jobs.size() - j0);
            hipLaunchKernelGGL(k_table_rows32, dim3(16, nj), dim3(256), 0, h->stream, d_jobs.as<ProfileJob>() + j0,
                               d_joffW.as<long long>() + j0, d_joff32.as<long long>() + j0, h->d_tabW.as<float>() + kTabPad, h->d_tabW32.as<float>(), s->osamp, 8 * nq);
          }
          h->slab32 = (unsigned)(32 * nq);
          HIPCHK(h, hipStreamSynchronize(h->stream));
          h->tabW32 = h->d_tabW32.as<float>();
        }
      }
      HIPCHK(h, hipStreamSynchronize(h->stream));
      h->tabW = h->d_tabW.as<float>() + kTabPad;
    }
  }
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipEventRecord(e1, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  float ms = 0; HIPCHK(h, hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  h->stats.ms_create_table = ms;
  h->stats.table_floats = h->tab_n;
  return TRX_OK;
}

// ---- line list preparation (the threaded host loops: ../trx_groups.h) -------
int prepare_lines(trx_handle *h, const trx_static *s)
{
  const int64_t n = s->nlines;
  StageTimer T;
  const int nth = create_threads();
  const double wn0 = s->wn_i, odwn = s->wn_d / s->osamp;
  const double own_last = wn0 + (double)(s->nown - 1) * odwn;
  if (n > 2000000000LL) return fail(h, TRX_E_UNSUPPORTED, "more than 2^31 lines per handle");
  // the three arrays that go up as they are leave now, on a thread of their own, under the grouping
  const double *elow = s->elow, *gf = s->gf;
  int rc_raw = TRX_OK;
  std::thread raw_up([&]() {
    (void)hipSetDevice(h->device);
    if ((rc_raw = upload_raw(h, h->d_elow, elow, (size_t)n)) || (rc_raw = upload_raw(h, h->d_gf, gf, (size_t)n)) ||
        (rc_raw = upload_raw(h, h->d_iso, s->isoid, (size_t)n))) return;
  });
  struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } raw_join{raw_up};
  HostBuf<double> wavn; wavn.alloc((size_t)n);
  HostBuf<uint8_t> inr; inr.alloc((size_t)n);
  {
    std::vector<int64_t> cnt((size_t)nth + 1, 0); std::vector<int> bad((size_t)nth + 1, 0);
    parallel_parts(n, nth, [&](int t, int64_t i0, int64_t i1) {
      int64_t c = 0;
      for (int64_t i = i0; i < i1; i++) {
        if (s->isoid[i] < 0 || s->isoid[i] >= s->niso) { bad[t] = 1; continue; }
        wavn[i] = 1.0 / (s->wl_um[i] * kTliWfct);
        inr[i] = !(wavn[i] < wn0 || wavn[i] > own_last);             // extinction.c:410
        c += inr[i];
      }
      cnt[t] = c;
    });
    for (int t = 0; t <= nth; t++) { if (bad[t]) return fail(h, TRX_E_ARG, "isotope id out of range"); h->ninrange += cnt[t]; }
  }
  T.lap("wavn + range flags");
  // TLI order: isotope blocks in ascending id, wavelength ascending inside a
  // block (pylineread.py:369-383); the gather kernel relies on it.
  {
    std::vector<int> bad((size_t)nth + 1, 0);
    parallel_parts(n, nth, [&](int t, int64_t i0, int64_t i1) {
      for (int64_t i = std::max<int64_t>(i0, 1); i < i1; i++) {
        if (s->isoid[i] < s->isoid[i-1]) bad[t] |= 1;
        else if (s->isoid[i] == s->isoid[i-1] && wavn[i] > wavn[i-1]) bad[t] |= 2;
      }
    });
    int any = 0;
    for (int t = 0; t <= nth; t++) any |= bad[t];
    if (any & 1) return fail(h, TRX_E_ORDER, "isotope blocks are not in ascending order");
    if (any & 2) return fail(h, TRX_E_ORDER, "wavelengths are not ascending inside an isotope block");
  }
  T.lap("order check");
  // co-added groups (extinction.c:445-462), grouped in pieces side by side (trx_groups.h)
  LineGroups LG;
  group_lines(n, s->isoid, wavn.data(), inr.data(), s->niso, wn0, odwn, nth, LG);
  HostBuf<int32_t> &gfirst = LG.first, &gcount = LG.count, &giown = LG.iown; HostBuf<int16_t> &giso = LG.iso; HostBuf<double> &gwavn = LG.wavn;
  h->iso_wmin = LG.iso_wmin; h->iso_wmax = LG.iso_wmax; h->nadd += LG.nadd;
  T.lap("grouping");
  h->nlines = n; h->ngroups = (int64_t)gfirst.size();
  // isotope blocks (groups are in line order: the first group of every isotope by bisection) and
  // the coarse-bin index over them
  std::vector<int32_t> gblock(s->niso + 1, 0);
  for (int b = 0; b <= s->niso; b++) gblock[b] = (int32_t)(std::lower_bound(giso.data(), giso.data() + giso.size(), (int16_t)b) - giso.data());
  {
    std::vector<int> bad((size_t)nth + 1, 0);
    parallel_parts((int64_t)giown.size(), nth, [&](int t, int64_t g0, int64_t g1) {
      for (int64_t g = std::max<int64_t>(g0, 1); g < g1; g++) if (giso[g] == giso[g-1] && giown[g] > giown[g-1]) bad[t] = 1;
    });
    for (int t = 0; t <= nth; t++) if (bad[t]) return fail(h, TRX_E_ORDER, "fine-grid indices are not descending inside an isotope block");
  }
  HostBuf<int32_t> cntge; cntge.alloc((size_t)s->niso * (s->nwn + 1));
  for (int b = 0; b < s->niso; b++) {
    const long long osamp = s->osamp, kmaxc = s->nwn - 1;
    count_ge(giown.data(), gblock[b], gblock[b + 1], s->nwn, [=](int32_t io) { return std::min<long long>(io / osamp, kmaxc); },
             &cntge[(size_t)b * (s->nwn + 1)], nth);
  }
  T.lap("cnt_ge");
  // the same counts at F sub-buckets per coarse cell (key iown*F/osamp): k_accumulate sizes its
  // windows with them, so that it does not stream whole cells of groups that lie between the
  // reach of two bins.  F = 1 (the table above) when the fine grid is no finer or the table
  // would be large.
  HostBuf<int32_t> cntsub;
  {
    int F = (int)std::min<long long>(16, s->osamp);
    while (F > 1 && (size_t)s->niso * (size_t)F * (size_t)s->nwn * 4 > ((size_t)64 << 20)) F /= 2;
    h->sub_f = F;
    if (F > 1) {
      const size_t stride = (size_t)F * s->nwn + 1;
      cntsub.alloc((size_t)s->niso * stride);
      for (int b = 0; b < s->niso; b++) {
        const long long osamp = s->osamp, kmaxs = (long long)stride - 2, FF = F;
        count_ge(giown.data(), gblock[b], gblock[b + 1], (long long)stride - 1,
                 [=](int32_t io) { return std::min<long long>((long long)io * FF / osamp, kmaxs); }, &cntsub[(size_t)b * stride], nth);
      }
    }
  }

  HostBuf<int32_t> lgroup; lgroup.alloc((size_t)n);
  parallel_parts(n, nth, [&](int, int64_t i0, int64_t i1) { std::fill(lgroup.begin() + i0, lgroup.begin() + i1, -1); });
  HostBuf<int32_t> gimod, gidiv; gimod.alloc(giown.size()); gidiv.alloc(giown.size());
  parallel_parts((int64_t)giown.size(), nth, [&](int, int64_t g0, int64_t g1) {
    for (int64_t g = g0; g < g1; g++) { lgroup[(size_t)gfirst[g]] = (int32_t)g; gimod[g] = giown[g] % s->osamp; gidiv[g] = giown[g] / s->osamp; }
  });
  T.lap("cnt_sub, lgroup, gimod");
  int rc;
  // ---- the walk's view of the list (k_line_walk): one 32-byte record per line, and line
  // ranges of ngw consecutive groups per isotope block
  // (32-bit byte offsets into the widened table: 8*tab_n + 64*osamp must stay below 2^32)
  h->walk_ok = s->osamp < (1 << 21) && h->tab_n < ((int64_t)1 << 28) && !gfirst.empty();
  if (h->walk_ok) {
    // Groups per range: ~2 rounds of resident waves (the hardware balances the rounds).  Taken from
    // the WHOLE list, not from what reaches this shard: the range size is part of the order of the
    // sums, and a shard's spectrum must be the same bits as the unsharded one.  (Shorter ranges for
    // small shards were measured: 1/8 of the demo 0.267 -> 0.259 ms with 32, slower with 16.)
    // Lists of more than a million groups get MORE ranges of 64 groups, not longer ones (up to 2^17
    // ranges: their partial records are ~50 MB per step and buffer): a rank of an N-way job walks 1/N
    // of them, and a range is one wave's serial work -- with 512-group ranges one shard of eight of an
    // 8*10^6-line list had 1 700 waves for 1 024 SIMDs (its 2-bin walk 226 us instead of 108).
    int ngw = 32;
    while (ngw < 64 && (int64_t)gfirst.size() / ngw > 16384) ngw *= 2;
    while (ngw < 512 && (int64_t)gfirst.size() / ngw > 131072) ngw *= 2;
    h->ngw = ngw;
    h->h_wbase.assign(s->niso + 1, 0);
    for (int b = 0; b < s->niso; b++) h->h_wbase[b + 1] = h->h_wbase[b] + (gblock[b + 1] - gblock[b] + ngw - 1) / ngw;
    h->nwaves = h->h_wbase[s->niso];
    if ((rc = upload(h, h->d_wbase, h->h_wbase))) return rc;
  }
  raw_up.join();
  if (rc_raw) return rc_raw;
  if ((rc = upload(h, h->d_wavn, wavn)) || (rc = upload(h, h->d_inr, inr)) || (rc = upload(h, h->d_lgroup, lgroup)) || (rc = upload(h, h->d_gfirst, gfirst)) ||
      (rc = upload(h, h->d_gcount, gcount)) || (rc = upload(h, h->d_giown, giown)) || (rc = upload(h, h->d_giso, giso)) ||
      (rc = upload(h, h->d_gwavn, gwavn)) || (rc = upload(h, h->d_gimod, gimod)) || (rc = upload(h, h->d_gidiv, gidiv)) || (rc = upload(h, h->d_gblock, gblock)) || (rc = upload(h, h->d_cntge, cntge)) ||
      (rc = upload(h, h->d_cntsub, cntsub)))
    return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));    // host vectors die at return
  T.lap("uploads + sync");
  h->h_gwavn = std::move(gwavn); h->h_gblock = gblock; h->h_cntge = std::move(cntge); h->h_gfirst = std::move(gfirst); h->h_gcount = std::move(gcount);
  T.lap("host copies");
  LinesDev &L = h->L;
  L.nlines = n; L.wavn = h->d_wavn.as<double>(); L.elow = h->d_elow.as<double>(); L.gf = h->d_gf.as<double>();
  L.iso = h->d_iso.as<int16_t>(); L.inrange = h->d_inr.as<uint8_t>(); L.lgroup = h->d_lgroup.as<int32_t>();
  L.ngroups = h->ngroups; L.gfirst = h->d_gfirst.as<int32_t>(); L.gcount = h->d_gcount.as<int32_t>();
  L.giown = h->d_giown.as<int32_t>(); L.giso = h->d_giso.as<int16_t>(); L.gwavn = h->d_gwavn.as<double>();
  L.gblock = h->d_gblock.as<int32_t>(); L.cnt_ge = h->d_cntge.as<int32_t>();
  if (h->walk_ok) {   // the walk's records (trx_walk.hip.h), built where the arrays already are
    if ((rc = ensure(h, h->d_walk, sizeof(WalkLine) * ((size_t)n + 1))) || (rc = ensure(h, h->d_linebase, sizeof(double) * ((size_t)n + 1)))) return rc;
    HIPCHK(h, hipMemsetAsync(h->d_linebase.p, 0, sizeof(double) * ((size_t)n + 1), h->stream));
    {
      std::vector<int> mg((size_t)nth + 1, 0);
      parallel_parts((int64_t)h->h_gcount.size(), nth, [&](int t, int64_t g0, int64_t g1) {
        int m = 0;
        for (int64_t g = g0; g < g1; g++) m = std::max(m, (int)h->h_gcount[(size_t)g]);
        mg[t] = m;
      });
      h->max_gcount = *std::max_element(mg.begin(), mg.end());
    }
    hipLaunchKernelGGL(k_walk_records, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, h->stream, (long long)n, L.wavn, L.elow, L.gf,
                       L.lgroup, L.giown, s->osamp, h->d_walk.as<WalkLine>());
    hipLaunchKernelGGL(k_walk_marks, dim3((unsigned)((h->nwaves + 63) / 64)), dim3(64), 0, h->stream, h->nwaves, h->ngw, s->niso,
                       h->d_wbase.as<int32_t>(), L.gblock, L.gfirst, L.gcount, h->d_walk.as<WalkLine>(), h->d_linebase.as<double>());
    if ((rc = ensure(h, h->d_rinfo, sizeof(RangeInfo) * (size_t)std::max(h->nwaves, 1)))) return rc;
    hipLaunchKernelGGL(k_range_info, dim3((unsigned)((h->nwaves + 255) / 256)), dim3(256), 0, h->stream, h->nwaves, h->ngw, s->niso,
                       h->d_wbase.as<int32_t>(), L.gblock, L.gfirst, L.gcount, h->d_walk.as<WalkLine>(), h->d_rinfo.as<RangeInfo>());
  }
  h->stats.nlines_inrange = h->ninrange; h->stats.ngroups = h->ngroups; h->stats.nadd = h->nadd;
  // ---- candidates for the layer maximum: lines no other line of their isotope dominates
  // (trx_walk.hip.h).  Falls back to "every line" when the filter would not pay.
  h->ncand = -1;
  if (h->ninrange > 4096 && s->niso > 0) {
    double emin = HUGE_VAL, emax = -HUGE_VAL;
    {
      std::vector<double> lo((size_t)nth + 1, HUGE_VAL), hi((size_t)nth + 1, -HUGE_VAL);
      parallel_parts(n, nth, [&](int t, int64_t i0, int64_t i1) {
        double a = HUGE_VAL, b = -HUGE_VAL;
        for (int64_t i = i0; i < i1; i++) if (inr[i]) { a = std::min(a, elow[i]); b = std::max(b, elow[i]); }
        lo[t] = a; hi[t] = b;
      });
      for (int t = 0; t <= nth; t++) { emin = std::min(emin, lo[t]); emax = std::max(emax, hi[t]); }
    }
    CandGeom Gm{};
    Gm.e_min = emin; Gm.e_scale = emax > emin ? kCandGrid / (emax - emin) : 0.0;
    Gm.w_min = wn0;  Gm.w_scale = own_last > wn0 ? kCandGrid / (own_last - wn0) : 0.0;
    const int cap = (int)std::max<int64_t>(4096, n / 8);
    DevBuf d_M, d_n;
    const size_t mbytes = sizeof(unsigned long long) * (size_t)s->niso * kCandGrid * kCandGrid;
    if ((rc = ensure(h, d_M, mbytes)) || (rc = ensure(h, d_n, sizeof(int))) || (rc = ensure(h, h->d_cand, sizeof(int32_t) * (size_t)cap))) return rc;
    HIPCHK(h, hipMemsetAsync(d_M.p, 0, mbytes, h->stream));
    HIPCHK(h, hipMemsetAsync(d_n.p, 0, sizeof(int), h->stream));
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_cand_cellmax, dim3(nb), dim3(256), 0, h->stream, (long long)n, L.wavn, L.elow, L.gf, L.iso, L.inrange, Gm, d_M.as<unsigned long long>());
    hipLaunchKernelGGL(k_cand_prefix, dim3((unsigned)s->niso), dim3(kCandGrid), 0, h->stream, d_M.as<unsigned long long>());
    hipLaunchKernelGGL(k_cand_select, dim3(nb), dim3(256), 0, h->stream, (long long)n, L.wavn, L.elow, L.gf, L.iso, L.inrange, Gm,
                       d_M.as<unsigned long long>(), h->d_cand.as<int32_t>(), d_n.as<int>(), cap);
    int nc = 0;
    HIPCHK(h, hipMemcpyAsync(&nc, d_n.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    if (nc > 0 && nc <= cap) {
      h->ncand = nc;
      if ((rc = ensure(h, h->d_candrec, sizeof(CandLine) * (size_t)nc))) return rc;
      hipLaunchKernelGGL(k_cand_pack, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, h->stream, nc, h->d_cand.as<int32_t>(),
                         L.wavn, L.elow, L.gf, L.iso, L.inrange, h->d_candrec.as<CandLine>());
      HIPCHK(h, hipStreamSynchronize(h->stream));
    }
  }
  h->stats.ncandidates = h->ncand;
  T.lap("candidates");
  return TRX_OK;
}

// ---- CIA: crosssec.c:272-344 + 354-428, device kernels ----------------------
// Host part: range checks and the no-extrapolation index windows only.
// density product of every CIA table and layer (crosssec.c:318-330), host side
void cia_densities(const trx_handle *h, const trx_atm *a, double *dens /* [ncia][nr] */)
{
  const int nr = a->nlayer;
  for (size_t n = 0; n < h->cia.size(); n++)
    for (int j = 0; j < nr; j++) {
      double d = 1.0;
      for (int k = 0; k < h->cia[n].nspec; k++) {
        const int m = h->cia[n].mol[k];
        d *= a->density[(size_t)m * nr + j] / (kAmu * h->mol_mass[m] * kAmagat);
      }
      dens[n * nr + j] = d;
    }
}

int cia_device(trx_handle *h, const trx_atm *a, const trx_opts *o, const double *d_tlay /* [nr] on device */,
               const double *d_dens /* [ncia][nr] on device */, hipStream_t cst)
{
  const int nr = a->nlayer; const long long nsh = h->nsh;
  if (h->cia.empty()) { HIPCHK(h, hipMemsetAsync(h->d_ecs.p, 0, sizeof(double) * (size_t)nr * nsh, cst)); return TRX_OK; }
  double tmin = 0.0, tmax = 70000.0;                        // crosssec.c:44-45, 175-176
  size_t nwmax = 0;
  for (auto &c : h->cia) { tmin = std::fmax(tmin, c.temp.front()); tmax = std::fmin(tmax, c.temp.back()); nwmax = std::max(nwmax, c.wn.size()); }
  for (int i = 0; i < nr; i++)
    if (a->temp[i] < tmin || a->temp[i] > tmax) return fail(h, TRX_E_RANGE, "layer temperature outside the CIA tables");
  int rc;
  // (per table of a batch: the table at the layers' temperatures, its second derivatives, and the sweeps' scratch -- one
  // piece per segment of k_cia_layers, each with room for its rows and both margins)
  const int seg_rows = h->cia_segments ? 128 : 1 << 30;
  const size_t seg_vrows = h->cia_segments ? (size_t)seg_rows + 2 * kCiaMargin + 8 : nwmax;
  const size_t seg_cap = h->cia_segments ? (nwmax + (size_t)seg_rows - 1) / (size_t)seg_rows : 1;
  const size_t cia_job_doubles = (2 * nwmax + seg_cap * seg_vrows) * (size_t)nr;
  if ((rc = ensure(h, h->d_cia_ws, sizeof(double) * cia_job_doubles * std::min<size_t>(kCiaBatch, h->cia.size())))) return rc;
  auto wn_at = [&](long long i) { return o->wn_fct * (h->wn_i + (double)(h->lo + i) * h->wn_d); };
  CiaBatch B{};
  bool first = true;             // the first batch writes the whole array (k_cia_eval), also when it is empty
  auto flush = [&](bool last) {
    if (B.n == 0 && !(first && last)) return;
    int nwave = 0, fj0 = nr, lj1 = 0, fjl = nr, ljl = 0; long long fi0 = nsh, li1 = 0;
    for (int t = 0; t < B.n; t++) {
      nwave = std::max(nwave, B.J[t].iz - B.J[t].ia + 1);      // (rows of the widest window)
      fj0 = std::min(fj0, B.J[t].fj); lj1 = std::max(lj1, B.J[t].lj);
      fi0 = std::min(fi0, B.J[t].fi); li1 = std::max(li1, B.J[t].li);
    }
    fjl = fj0; ljl = lj1;
    if (first) { fj0 = 0; lj1 = nr; fi0 = 0; li1 = nsh; }
    if (B.n > 0) {
      hipLaunchKernelGGL(k_cia_rows, dim3((unsigned)(((long long)nwave * nr + 255) / 256), (unsigned)B.n), dim3(256), 0, cst, B, nr, d_tlay);
      bool sums = true;                    // every table of the batch has its weights: two sums per row instead of the sweeps
      for (int t = 0; t < B.n; t++) sums = sums && B.J[t].C.wf != nullptr;
      if (sums) {
        hipLaunchKernelGGL(k_cia_v, dim3((unsigned)(((long long)nwave * nr + 255) / 256), (unsigned)B.n), dim3(256), 0, cst, B, nr);
        hipLaunchKernelGGL(k_cia_z, dim3((unsigned)(((long long)nwave * nr + 255) / 256), (unsigned)B.n), dim3(256), 0, cst, B, nr);
      }
      unsigned nseg = 1;
      for (int t = 0; t < B.n && !sums; t++) {
        const long nw = B.J[t].C.nwave;
        const long need_a = B.J[t].ia == 0 ? 0 : B.J[t].ia + kCiaMargin, need_b = B.J[t].iz == nw - 1 ? nw - 1 : B.J[t].iz - kCiaMargin;
        if (h->cia_segments && need_b >= need_a) nseg = std::max<unsigned>(nseg, (unsigned)((need_b - need_a + seg_rows) / seg_rows));
      }
      if (!sums)
        hipLaunchKernelGGL(k_cia_layers, dim3((unsigned)((ljl - fjl + 63) / 64), (unsigned)B.n, nseg), dim3(64), 0, cst, B, nr, seg_rows,
                           (long long)(seg_vrows * (size_t)nr));
    }
    if (nsh > 65536)
      hipLaunchKernelGGL(k_cia_eval<16>, dim3((unsigned)((li1 - fi0 + 255) / 256), (unsigned)((lj1 - fj0 + 15) / 16)), dim3(256), 0, cst,
                         B, nr, nsh, h->lo, h->wn_i, h->wn_d, o->wn_fct, fi0, li1, fj0, lj1, first ? 1 : 0, h->d_ecs.as<double>());
    else
      hipLaunchKernelGGL(k_cia_eval<1>, dim3((unsigned)((li1 - fi0 + 255) / 256), (unsigned)(lj1 - fj0)), dim3(256), 0, cst,
                         B, nr, nsh, h->lo, h->wn_i, h->wn_d, o->wn_fct, fi0, li1, fj0, lj1, first ? 1 : 0, h->d_ecs.as<double>());
    B.n = 0; first = false;
  };
  for (size_t n = 0; n < h->cia.size(); n++) {
    auto &c = h->cia[n];
    const long long nt1 = nsh; const int nt2 = nr;
    const double fx1 = c.wn.front(), lx1 = c.wn.back(), fx2 = c.temp.front(), lx2 = c.temp.back();
    if (wn_at(0) > lx1 || wn_at(nt1 - 1) < fx1 || a->temp[0] > lx2 || a->temp[nt2 - 1] < fx2) continue;   // crosssec.c:376-377
    // first index not below the table, first index above it (crosssec.c:381-393)
    long long fi = (long long)std::floor((fx1 / o->wn_fct - h->wn_i) / h->wn_d) - h->lo - 2;
    if (fi < 0) fi = 0;
    while (fi < nt1 && wn_at(fi) < fx1) fi++;
    long long li = (long long)std::ceil((lx1 / o->wn_fct - h->wn_i) / h->wn_d) - h->lo + 2;
    if (li > nt1) li = nt1;
    while (li > 0 && wn_at(li - 1) > lx1) li--;
    int fj = 0, lj = nt2;
    while (a->temp[fj] < fx2) fj++;
    for (int j = 0; j < lj; j++) if (a->temp[j] > lx2) lj = j;
    if (fi >= li || fj >= lj) continue;
    CiaJob &J = B.J[B.n];
    J.C = CiaDev{(int)c.wn.size(), (int)c.temp.size(), c.d_wn.as<double>(), c.d_temp.as<double>(), c.d_cs.as<double>(),
                 c.d_zt.as<double>(), c.d_uw.as<double>(), c.d_ruw.as<double>(), c.d_rh.as<double>(),
                 (h->cia_sums && !c.wf.empty()) ? c.d_wf.as<double>() : nullptr, (h->cia_sums && !c.wf.empty()) ? c.d_wb.as<double>() : nullptr};
    J.fj = fj; J.lj = lj; J.fi = fi; J.li = li;
    {   // table rows the wavenumber spline is solved for: those the run's wavenumbers bracket, a margin to spare (k_cia_layers)
      const int nw = (int)c.wn.size();
      J.ia = 0; J.iz = nw - 1;
      if (h->cia_window && nw > 4 * kCiaMargin) {
        const double xa = wn_at(fi), xb = wn_at(li - 1);
        const int ra = (int)(std::upper_bound(c.wn.begin(), c.wn.end(), xa) - c.wn.begin()) - 1;      // last row at or below the first wavenumber
        const int rb = (int)(std::lower_bound(c.wn.begin(), c.wn.end(), xb) - c.wn.begin());          // first row at or above the last one
        const int ia = ra - 2 - kCiaMargin, iz = rb + 2 + kCiaMargin;
        if (ia >= 3) J.ia = ia;
        if (iz <= nw - 4) J.iz = iz;
      }
    }
    J.mid = h->d_cia_ws.as<double>() + cia_job_doubles * (size_t)B.n; J.z2 = J.mid + nwmax * nr; J.v = J.z2 + nwmax * nr;
    J.dens = d_dens + n * nr;
    if (++B.n == kCiaBatch) flush(false);
  }
  flush(true);
  HIPCHK(h, hipGetLastError());
  return TRX_OK;
}

// Simpson weights of one abscissa (numerical.c:390-425 geth, 486-495 makeh):
// per interval pair {2-hratio, hfactor, 2-1/hratio, hsum}; h0 = first interval.
// 1 / d for quotient_rn(x, d, 1/d) in place of x / d -- if that IS the division for this divisor:
// Markstein's theorem leaves out divisors with a mantissa of all ones, so the product form is
// compared with the division on a few thousand numerators of every magnitude an optical depth takes,
// exact multiples and their neighbours among them; any difference: 0 (the kernel divides).  Once per
// divisor and handle (a run's angles rarely change).
double checked_reciprocal(trx_handle *h, double d)
{
  for (auto &kv : h->recip_ok) if (kv.first == d) return kv.second;
  double rd = (d > 0 && std::isfinite(d)) ? 1.0 / d : 0.0;
  if (rd != 0.0) {
    uint64_t st = 0x9e3779b97f4a7c15ull ^ (uint64_t)(d * 1e15);
    for (int k = 0; k < 4096 && rd != 0.0; k++) {
      st = st * 6364136223846793005ull + 1442695040888963407ull;
      const double m = 1.0 + (double)(st >> 11) * 0x1.0p-53;                     // mantissa in [1, 2)
      const int ex = (int)((st >> 3) % 120) - 100;                                // 2^-100 .. 2^19
      double x = std::ldexp(m, ex);
      if (k % 4 == 1) x = std::ldexp((double)(1 + (st >> 40) % 4096), ex) * d;    // (near) exact multiples of the divisor
      if (k % 4 == 2) x = std::nextafter(std::ldexp((double)(1 + (st >> 40) % 4096), ex) * d, 0.0);
      if (quotient_rn(-x, d, rd) != -x / d) rd = 0.0;
    }
  }
  if (h->recip_ok.size() >= 64) h->recip_ok.clear();
  h->recip_ok.emplace_back(d, rd);
  return rd;
}

void simpson_weights(const double *x, int n, double *row, double *h0)
{
  *h0 = (n >= 2) ? x[1] - x[0] : 0.0;
  if (n < 3) return;
  const int even = (n % 2 == 0);
  for (int i = 0; i < (n - 1) / 2; i++) {
    const int j = 2 * i + even;
    const double ha = x[j+1] - x[j], hb = x[j+2] - x[j+1];
    const double hsum = ha + hb, hratio = hb / ha, hfactor = hsum * hsum / (ha * hb);
    row[4*i] = 2.0 - hratio; row[4*i+1] = hfactor; row[4*i+2] = 2.0 - 1.0 / hratio; row[4*i+3] = hsum;
  }
}

// smallest wavenumber w for which alphad*w/alphal >= 0.1 holds in double
// arithmetic (extinction.c:480); the predicate is monotone in w, so start from
// the algebraic root and walk the few ulps to the exact floating-point edge.
double doppler_refresh_cut(double alphad, double alphal)
{
  auto cond = [&](double w) { return alphad * w / alphal >= 1e-1; };
  if (!(alphad > 0) || !(alphal > 0)) {                 // degenerate widths: bisection over all doubles
    double lo = 0.0, hi = 1e300;
    if (!cond(hi)) return HUGE_VAL;
    if (cond(std::nextafter(0.0, 1.0))) return 0.0;
    while (std::nextafter(lo, hi) < hi) {
      double mid = lo + (hi - lo) / 2;
      if (mid <= lo || mid >= hi) mid = std::nextafter(lo, hi);
      if (cond(mid)) hi = mid; else lo = mid;
    }
    return hi;
  }
  double w = 1e-1 * alphal / alphad;
  if (!std::isfinite(w)) return HUGE_VAL;
  // (the neighbours of a positive finite double are its bit pattern +- 1: std::nextafter, a library call, was a
  // quarter of the prologue's time per (layer, isotope) pair)
  auto up = [](double x) { uint64_t u; std::memcpy(&u, &x, 8); u++; std::memcpy(&x, &u, 8); return x; };            // x > 0 finite -> next above (inf after DBL_MAX)
  auto down = [](double x) { uint64_t u; std::memcpy(&u, &x, 8); u--; std::memcpy(&x, &u, 8); return x; };          // x > 0 -> next below (+0 after the smallest denormal)
  if (!(w > 0)) w = std::nextafter(0.0, 1.0);
  while (!cond(w)) { w = up(w); if (!std::isfinite(w)) return HUGE_VAL; }
  for (;;) {
    const double p = down(w);
    if (p > 0 && cond(p)) w = p; else break;
  }
  return w;
}

// ---- RCCL, resolved at run time (the library must load on a CPU-only box) ---
struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  const char *(*GetLastError)(ncclComm_t) = nullptr;
  bool ok() const { return lib && GetUniqueId && CommInitRank && CommDestroy && AllGather; }
};
Rccl &rccl()
{
  static Rccl R;
  if (R.lib) return R;
  const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char *n : names) if ((R.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;    // the copy torch already mapped
  if (!R.lib) for (const char *n : names) if ((R.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!R.lib) return R;
  R.GetUniqueId  = (decltype(R.GetUniqueId))dlsym(R.lib, "ncclGetUniqueId");
  R.CommInitRank = (decltype(R.CommInitRank))dlsym(R.lib, "ncclCommInitRank");
  R.CommDestroy  = (decltype(R.CommDestroy))dlsym(R.lib, "ncclCommDestroy");
  R.CommAbort    = (decltype(R.CommAbort))dlsym(R.lib, "ncclCommAbort");
  R.AllGather    = (decltype(R.AllGather))dlsym(R.lib, "ncclAllGather");
  R.GetErrorString = (decltype(R.GetErrorString))dlsym(R.lib, "ncclGetErrorString");
  R.GetLastError = (decltype(R.GetLastError))dlsym(R.lib, "ncclGetLastError");
  return R;
}

// ---- per-layer scalars (extinction.c:364-395), shared by trx_run and trx_sweep_permol ----
struct LayerHost {
  size_t nli = 0, extra_off = 0;
  std::vector<double> &f64; std::vector<int32_t> &i32;      // (the handle's: no allocation per run)
  const int32_t *psmax = nullptr;
  LayerHost(std::vector<double> &f, std::vector<int32_t> &i) : f64(f), i32(i) {}
};

// -c/T per layer and the strength factor SIGCTE*ratio/(m*Z) per (layer, isotope): all k_layer_max needs of a run's
// inputs.  trx_run writes them first, straight into the pinned block, and launches the layer maxima before the rest of
// the prologue (prep_layers, which computes the same values again for the block's device copy).
inline double layer_negct(double temp) { return -kExpCte * kTliEfct / temp; }
inline double layer_strength(const trx_handle *h, int i, double z) { return kSigCte * h->iso_ratio[i] / (h->iso_mass[i] * z); }

// nearest_index(adop, v, 0, ndop) by its exact steps (build_table), walked from a nearby index: the Doppler indices
// one (layer, isotope) pair asks for lie a few steps apart, and a bisection each was most of prep_layers' time.
inline int dop_index(const trx_handle *h, double v, int from)
{
  const double *thr = h->dopthr.data();      // thr[0] = -inf, thr[ndop] = +inf
  while (from + 1 < h->ndop && v >= thr[from + 1]) from++;
  while (v < thr[from]) from--;
  return from;
}
inline int lor_index(const trx_handle *h, double v, int from)
{
  if (h->lorthr.empty()) return nearest_index(h->alor.data(), v, 0, h->nlor);
  const double *thr = h->lorthr.data();      // thr[0] = -inf, thr[nlor] = +inf
  while (from + 1 < h->nlor && v >= thr[from + 1]) from++;
  while (v < thr[from]) from--;
  return from;
}
inline int dop_index_far(const trx_handle *h, double v)
{
  const double *thr = h->dopthr.data();
  return (int)(std::upper_bound(thr + 1, thr + h->ndop, v) - thr) - 1;      // (thr[1 .. ndop-1] <= v, counted)
}

int prep_layers(trx_handle *h, int nr, const double *temp_k, const double *density /* [nmol][nr] */,
                const double *zpart /* [niso][nr] */, size_t extra_doubles, LayerHost &LH)
{
  const int niso = h->niso, nmol = h->nmol;
  const size_t nli = (size_t)nr * std::max(niso, 1);
  LH.nli = nli; LH.extra_off = 7 * nli;
  LH.f64.assign(7 * nli + extra_doubles, 0.0);
  LH.i32.assign(4 * nli, 0);
  double *negct = &LH.f64[0], *strength = negct + nr, *dens = strength + nli, *alphad = dens + nli,
         *alphal = alphad + nli, *wcut = alphal + nli;
  int32_t *idop0 = &LH.i32[0], *ilor = idop0 + nli, *psmax = ilor + nli, *npre = psmax + nli;
  LH.psmax = psmax;
  h->dens_over_m.resize(nmol);
  double *dm = h->dens_over_m.data();
  // (the indices of an isotope change by a step or two from a layer to the next: each isotope's walks start where the
  // layer above ended -- a bisection per pair and grid was a third of the prologue's time)
  h->guess_dop.assign((size_t)std::max(niso, 1), 0); h->guess_lor.assign((size_t)std::max(niso, 1), 0);
  h->guess_a0.assign((size_t)std::max(niso, 1), 0); h->guess_a1.assign((size_t)std::max(niso, 1), 0); h->guess_npre.assign((size_t)std::max(niso, 1), -1);
  for (int r = 0; r < nr; r++) {
    const double temp = temp_k[r];
    if (!(temp > 0)) return fail(h, TRX_E_ARG, "non-positive layer temperature");
    negct[r] = layer_negct(temp);
    const double fdoppler = std::sqrt(2 * kKb * temp / kAmu) * kSqrtLn2 / kLs;
    const double florentz = std::sqrt(2 * kKb * temp / kPi / kAmu) / (kAmu * kLs);
    for (int j = 0; j < nmol; j++) dm[j] = density[(size_t)j * nr + r] / h->mol_mass[j];      // (the first factor of the sum's terms, for every isotope)
    for (int i = 0; i < niso; i++) {
      double al = 0.0;
      const double *csd_i = &h->pair_csd[(size_t)i * nmol], *sq_i = &h->pair_sqrt[(size_t)i * nmol];
      for (int j = 0; j < nmol; j++)       // (the collision diameter and the reduced-mass root of the pair: constants of the handle)
        al += dm[j] * csd_i[j] * csd_i[j] * sq_i[j];
      al *= florentz;
      const double ad = fdoppler / h->iso_sqrtm[i];
      const size_t k = (size_t)r * niso + i;
      alphal[k] = al; alphad[k] = ad;
      idop0[k] = h->guess_dop[i] = dop_index(h, ad * h->wn_i, h->guess_dop[i]);
      ilor[k]  = h->guess_lor[i] = lor_index(h, al, h->guess_lor[i]);
      strength[k] = layer_strength(h, i, zpart[(size_t)i * nr + r]);
      dens[k] = density[(size_t)h->iso_imol[i] * nr + r];
      wcut[k] = doppler_refresh_cut(ad, al);
      // (Doppler indices at the ends of the isotope's lines, and the widest profile between two indices: asked for
      // several times per pair -- each index is looked up once, and on a table whose profiles widen with the Doppler
      // width, the usual case, the widest of a range is its last)
      const bool has_lines = h->iso_wmax[i] > 0;
      const int a0 = has_lines ? (h->guess_a0[i] = dop_index(h, ad * h->iso_wmin[i], h->guess_a0[i])) : idop0[k];
      const int a1 = has_lines ? (h->guess_a1[i] = dop_index(h, ad * h->iso_wmax[i], h->guess_a1[i])) : idop0[k];
      const int32_t *psl = &h->psizeT[(size_t)ilor[k] * h->ndop];
      auto widest_of = [&](int i0, int i1) {
        if (i0 > i1) std::swap(i0, i1);
        if (h->psize_mono) return psl[i1];
        int32_t m = 0;
        for (int d = i0; d <= i1; d++) m = std::max(m, psl[d]);
        return m;
      };
      int32_t pm = widest_of(std::min(idop0[k], std::min(a0, a1)), std::max(idop0[k], std::max(a0, a1)));
      // The bound is then tightened to the Doppler indices a line of this isotope can actually TAKE in
      // this layer, among the lines that can reach this handle's bins (all of the block, or -- a
      // shard -- those within the reach of the widest profile, pm, a cell to spare):
      //   * anchor >= wcut ("own", extinction.c:480-483): the index of its own wavenumber;
      //   * anchor < wcut: the sticky index -- the block's index at wn_i, or that of SOME anchor >= wcut
      //     anywhere in the block (k_sticky_index) -- whatever the window.
      // Doppler widths grow with the wavenumber: on a band that spans a factor of ten the widest
      // profile of the list is several times the widest one a low-wavenumber shard meets, and in the
      // deep layers (no anchor reaches wcut) every line takes the ONE profile of the index at wn_i.
      if (h->shard_frames && has_lines) {
        double lo_w = h->iso_wmin[i], hi_w = h->iso_wmax[i];
        if (h->windowed()) {
          const double reach = ((double)pm + h->osamp) * (h->wn_d / h->osamp) + h->wn_d;
          lo_w = std::max(lo_w, h->wn_i + (double)h->lo * h->wn_d - reach);
          hi_w = std::min(hi_w, h->wn_i + (double)(h->hi - 1) * h->wn_d + reach);
        }
        const double wc = wcut[k];
        auto didx = [&](double w) { return w == h->iso_wmin[i] ? a0 : w == h->iso_wmax[i] ? a1 : dop_index_far(h, ad * w); };
        auto widest = [&](double wa, double wb) { return widest_of(didx(wa), didx(wb)); };
        int32_t pw = 0;
        if (lo_w <= hi_w) {
          const double own_lo = std::max(lo_w, wc), all_lo = std::max(h->iso_wmin[i], wc);
          int32_t w_own = -1;
          if (hi_w >= wc) pw = std::max(pw, w_own = widest(own_lo, hi_w));                      // own indices of the lines in reach
          if (lo_w < wc) {                                                                       // some line in reach takes the sticky index
            pw = std::max(pw, psl[idop0[k]]);
            if (h->iso_wmax[i] >= wc)
              pw = std::max(pw, w_own >= 0 && own_lo == all_lo && hi_w == h->iso_wmax[i] ? w_own : widest(all_lo, h->iso_wmax[i]));
          }
        }
        pm = std::min(pm, pw);
      }
      psmax[k] = pm;
      {   // groups of the block that refresh the Doppler index: wavn >= wcut (descending order)
        const double *gb = h->h_gwavn.data() + h->h_gblock[i];
        const double wc = wcut[k];
        // (the block's groups descend in wavenumber: bracketed by the per-cell counts first -- groups two cells above
        // wcut's are all >= it, groups two cells below all < it -- so that the bisection stays inside ~3 cells of groups
        // instead of walking 20 cold cache lines of a 10^6-entry array)
        const int32_t *cg = &h->h_cntge[(size_t)i * (h->nwn + 1)];
        const double kcd = std::floor((wc - h->wn_i) / h->wn_d);
        const long long kc = kcd < -4 ? -4 : kcd > (double)h->nwn + 4 ? h->nwn + 4 : (long long)kcd;
        const double *pa = gb + cg[std::min<long long>(std::max<long long>(kc + 2, 0), h->nwn)];
        const double *pz = gb + cg[std::min<long long>(std::max<long long>(kc - 2, 0), h->nwn)];
        // (the cut moves a little from a layer to the next: the search starts at the layer above's answer and doubles its
        // step -- the lines it touches are the ones the layer above left in the cache; the same partition point)
        const double *lo = pa, *hi = pz;                       // [lo, hi): all of [gb, lo) >= wc, all of [hi, ..) < wc
        const long g0 = h->guess_npre[i];
        if (g0 >= 0 && gb + g0 >= pa && gb + g0 <= pz) {
          const double *p = gb + g0;
          long stepw = 1;
          if (p < pz && *p >= wc) {                            // the answer lies above p
            lo = p + 1;
            while (lo < hi) { const double *q = lo + stepw - 1; if (q >= hi) break; if (*q >= wc) { lo = q + 1; stepw *= 2; } else { hi = q; break; } }
          } else {                                             // at or below p
            hi = p;
            while (lo < hi) { const double *q = hi - stepw; if (q < lo) break; if (*q >= wc) { lo = q + 1; break; } else { hi = q; stepw *= 2; } }
          }
        }
        const long np = (long)(std::partition_point(lo, hi, [wc](double w) { return w >= wc; }) - gb);
        npre[k] = (int32_t)np; h->guess_npre[i] = np;
      }
    }
  }
  return TRX_OK;
}

void layer_dev(const double *df, const int32_t *di, const LayerHost &LH, int nr, LayerDev &Y, const double *&d_wcut, const int32_t *&d_npre)
{
  const size_t nli = LH.nli;
  Y.negc_over_t = df; Y.strength_f = df + nr; Y.density = Y.strength_f + nli; Y.alphad = Y.density + nli;
  Y.alphal = Y.alphad + nli; d_wcut = Y.alphal + nli;
  Y.idop0 = di; Y.ilor = di + nli; Y.psmax = Y.ilor + nli; d_npre = Y.psmax + nli;
}

// ---- per-kernel timing of a profiled run: (start, end) event pairs on the stream of the kernel ----
struct Spans {
  enum Kind { kSweep = 0, kWalk = 1, kAccum = 2, kTau = 3, kWalkLanes = 4, kWalkPacked = 5, kKinds = 6 };      // (kWalk: k_line_walk itself; the three walk forms add up to trx_stats.ms_k_walk)
  struct Span { hipEvent_t a, b; int kind; };
  std::vector<Span> v;
  int begin(int kind, hipStream_t st) {
    Span sp{nullptr, nullptr, kind};
    if (hipEventCreate(&sp.a) != hipSuccess || hipEventCreate(&sp.b) != hipSuccess) return -1;
    v.push_back(sp);
    return hipEventRecord(sp.a, st) == hipSuccess ? 0 : -1;
  }
  int end(hipStream_t st) { return hipEventRecord(v.back().b, st) == hipSuccess ? 0 : -1; }
  void sum(double out[kKinds]) const {
    for (const Span &sp : v) { float t = 0; if (hipEventElapsedTime(&t, sp.a, sp.b) == hipSuccess) out[sp.kind] += t; }
  }
  static bool is_walk(int k) { return k == kWalk || k == kWalkLanes || k == kWalkPacked; }
  // first walk's start to last walk's end (the walks of one run may share the device on two queues)
  double walk_span() const {
    double best = 0;
    for (const Span &x : v) if (is_walk(x.kind))
      for (const Span &y : v) if (is_walk(y.kind)) { float t = 0; if (hipEventElapsedTime(&t, x.a, y.b) == hipSuccess && t > best) best = t; }
    return best;
  }
  ~Spans() { for (Span &sp : v) { if (sp.a) (void)hipEventDestroy(sp.a); if (sp.b) (void)hipEventDestroy(sp.b); } }
};

// ---- one step of the line sweep ------------------------------------------------------------
struct SweepMode {
  bool eager = false, prof = false, skip_done = false, permol = false;
  double ethresh = 0;
  int nmx = 1; const int32_t *d_iso_mx = nullptr;   // output slot per isotope (per-molecule sweeps)
  double *d_e = nullptr;                             // [layer][nmx][nsh]
  const double *d_kmax = nullptr;                    // [layer][nmx] (k_layer_max)
  const int *d_sticky = nullptr;                     // [layer][iso] (k_sticky_index)
  hipStream_t st = nullptr;                          // null: the handle's stream
};

// widest profile (in fine-grid points) any isotope with lines can use in layer r
long long layer_psmax(const trx_handle *h, const int32_t *psmax, int r)
{
  long long pm = 0;
  for (int b = 0; b < h->niso; b++)
    if (h->h_gblock[b] != h->h_gblock[b + 1]) pm = std::max<long long>(pm, psmax[(size_t)r * h->niso + b]);
  return pm;
}

// Frame size of the walk for layer r (bins a line can reach = 2*Rc + 2 with Rc whole cells of
// profile half-width), or 0 when its profiles are wider than the largest frame: the layer then
// takes the two-kernel path.  A per-LAYER property, so that the path a layer takes -- and with
// it the order of its sums -- does not depend on how the layers are grouped into steps.
int walk_frame_bins(const trx_handle *h, const int32_t *psmax, int r)
{
  if (!h->walk_ok || !h->walk_temp_ok) return 0;
  const long long rc = layer_psmax(h, psmax, r) / h->osamp;
  // (The 16-bin frame -- profiles reaching 4-7 cells -- used to pay only on sparse lists: with a load
  // per bin it cost 1.4 ms for the 9 such layers of configs[2] against ~0.4 ms in the two-kernel
  // form.  Reading rows of the table copy, two lanes per layer, it is the cheaper one there too:
  // 2.32 -> 2.26 ms.)
  return rc <= 0 ? 2 : rc <= 1 ? 4 : rc <= 3 ? 8 : rc <= 7 ? 16 : 0;     // (tab_n >= Rc*osamp follows: a profile that wide is in the table)
}

// plan of the line ranges for a frame of nb bins (built once per handle and frame size)
int walk_plan(trx_handle *h, int nb, hipStream_t st, WalkPlan &P, trx_handle::Plan *&pl)
{
  const int v = nb == 2 ? 0 : nb == 4 ? 1 : nb == 8 ? 2 : 3;
  pl = &h->plan[v];
  int rc;
  if (!pl->built) {
    const size_t nw = (size_t)std::max(h->nwaves, 1);
    if ((rc = ensure(h, pl->blo, 4 * nw)) || (rc = ensure(h, pl->bhi, 4 * nw)) || (rc = ensure(h, pl->off, 8 * (nw + 1)))) return rc;
  }
  P.nwaves = h->nwaves; P.ngw = h->ngw; P.wbase = h->d_wbase.as<int32_t>();
  P.blo = pl->blo.as<int32_t>(); P.bhi = pl->bhi.as<int32_t>(); P.off = pl->off.as<int64_t>();
  // (the per-bin range table: 8 bytes per bin and isotope; above 2^24 entries the combine searches instead)
  const long long nbinw = (long long)h->nsh * h->niso;
  const bool with_binw = nbinw > 0 && nbinw <= (1LL << 24);
  if (!pl->built && with_binw && (rc = ensure(h, pl->binw, 8 * (size_t)nbinw))) return rc;
  P.binw = with_binw ? pl->binw.as<int32_t>() : nullptr;
  // (and the bin's record in each of its first 64 ranges, for k_ray_tail: 256 bytes per bin and isotope, small shards only)
  const bool with_binrec = with_binw && nbinw <= (1LL << 17) && !h->no_binrec;
  if (!pl->built && with_binrec && (rc = ensure(h, pl->binrec, 256 * (size_t)nbinw))) return rc;
  P.binrec = with_binrec ? pl->binrec.as<int32_t>() : nullptr;
  if (!pl->built) {
    hipLaunchKernelGGL(k_wave_plan, dim3(1), dim3(256), 0, st, P, h->niso, h->d_gblock.as<int32_t>(), h->d_gidiv.as<int32_t>(),
                       nb / 2 - 1, (long long)h->lo, (long long)h->hi);
    if (with_binw)
      hipLaunchKernelGGL(k_bin_ranges, dim3((unsigned)((nbinw + 255) / 256)), dim3(256), 0, st, P, h->niso, (long long)h->lo, (long long)h->nsh);
    int64_t total = 0;
    HIPCHK(h, hipMemcpyAsync(&total, P.off + h->nwaves, sizeof total, hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));                 // once per handle and frame size
    pl->records = total; pl->built = true;
  }
  if (pl->records >= (1LL << 31)) P.binrec = nullptr;      // (32-bit record numbers)
  return TRX_OK;
}

template <int NB>
void launch_walk(const WalkArgs &A, bool prof, unsigned nwaves, hipStream_t st)
{
  const dim3 grid((nwaves + kWalkWaves - 1) / kWalkWaves), block(64 * kWalkWaves);
  if (prof) hipLaunchKernelGGL((k_line_walk<NB, true>), grid, block, 0, st, A);
  else if constexpr (NB >= 8) {      // steps of at most 32 layers: two lanes per layer (trx_walk.hip.h)
    if (A.nc <= 32) hipLaunchKernelGGL((k_line_walk<NB, false, 2>), grid, block, 0, st, A);
    else            hipLaunchKernelGGL((k_line_walk<NB, false>), grid, block, 0, st, A);
  }
  else      hipLaunchKernelGGL((k_line_walk<NB, false>), grid, block, 0, st, A);
}

// A combine whose launch has been put off (the host queues the NEXT step's walk first, so that the
// walks sit back to back on their queue whatever else this step still has to queue).
struct PendingCombine {
  bool valid = false;
  CombineArgs C{}; hipStream_t sc = nullptr, st_walk = nullptr; hipEvent_t ev_walk = nullptr, ev_done = nullptr; bool cross = false;
};

int launch_combine(trx_handle *h, PendingCombine &pc, Spans *sp)
{
  if (!pc.valid) return TRX_OK;
  pc.valid = false;
  if (pc.cross) HIPCHK(h, hipStreamWaitEvent(pc.sc, pc.ev_walk, 0));
  if (sp && sp->begin(Spans::kAccum, pc.sc)) return fail(h, TRX_E_HIP, "event");
  hipLaunchKernelGGL(k_walk_combine, dim3((unsigned)((h->nsh + kCombineBins - 1) / kCombineBins)), dim3(64 * kCombineBins), 0, pc.sc, pc.C);
  if (sp && sp->end(pc.sc)) return fail(h, TRX_E_HIP, "event");
  if (pc.cross && pc.ev_done) HIPCHK(h, hipEventRecord(pc.ev_done, pc.sc));
  return TRX_OK;
}

// The walk: layers r_top .. r_top-nc+1 (nc <= 64) in one kernel on M.st, then the combine of its
// partial sums on st_comb (the stream the optical depth runs on; null: M.st).  Consecutive steps
// alternate between two record buffers, so that the next step's walk does not wait for this
// step's combine.
int walk_chunk(trx_handle *h, const LayerDev &Y, const double *d_wcut, int nb, int r_top, int nc, const SweepMode &M, Spans *sp,
               int parity, hipStream_t st_comb, hipEvent_t ev_walk, hipEvent_t ev_reuse = nullptr, hipEvent_t ev_done = nullptr,
               PendingCombine *defer = nullptr /* non-null: the combine is handed back instead of launched */,
               int *form_out = nullptr /* which kernel took the step: 0 k_line_walk, 1 k_line_walk_lanes, 2 k_line_walk_packed */)
{
  hipStream_t st = M.st ? M.st : h->stream;
  WalkPlan P{}; trx_handle::Plan *pl = nullptr;
  int rc = walk_plan(h, nb, st, P, pl);
  if (rc) return rc;
  DevBuf &part = h->d_part[parity & 1];
  h->stats.walk_steps++; h->stats.walk_records += pl->records; h->stats.walk_record_lanes += pl->records * nc;
  const size_t pbytes = sizeof(double) * kWalkLayers * (size_t)std::max<int64_t>(pl->records, 1);
  if (part.bytes < pbytes) {
    HIPCHK(h, hipStreamSynchronize(st));                 // an earlier step may still be using the old buffer
    HIPCHK(h, hipStreamSynchronize(h->stream4));          // (combines of earlier steps run there)
    if ((rc = ensure(h, part, pbytes))) return rc;
  }
  // this buffer's previous records (two steps ago) must have been combined
  if (ev_reuse) HIPCHK(h, hipStreamWaitEvent(st, ev_reuse, 0));
  WalkArgs A{};
  A.lines = h->d_walk.as<WalkLine>(); A.rinfo = h->d_rinfo.as<RangeInfo>(); A.gfirst = h->d_gfirst.as<int32_t>(); A.gcount = h->d_gcount.as<int32_t>();
  A.gblock = h->d_gblock.as<int32_t>(); A.P = P;
  A.niso = h->niso; A.nlor = h->nlor; A.ndop = h->ndop; A.osamp = h->osamp; A.lo = h->lo; A.hi = h->hi;
  A.r_top = r_top; A.nc = nc; A.Y = Y; A.wcut = d_wcut; A.kmax = M.d_kmax; A.ethresh = M.ethresh;
  A.nmx = M.nmx; A.iso_mx = M.d_iso_mx; A.permol = M.permol; A.sticky_idop = M.d_sticky;
  A.dthr = h->d_dopthr.as<double>(); A.e2tab = h->d_e2tab.as<double>();
  A.psize = h->d_psize.as<int32_t>(); A.poff = h->d_poff.as<long long>();
  A.table = h->tab; A.zero_index = h->tab_n;
  A.tabw = h->tabW; A.walkprof = h->d_walkprof.as<WalkProfile>();
  A.tabw32 = h->tabW32; A.wp32 = h->tabW32 ? h->d_wp32.as<uint32_t>() : nullptr; A.slab32 = h->slab32;
  A.xcd_map = h->xcd_map & 2;                        // (bit 1: k_line_walk, bit 0: k_line_walk_lanes)
  A.part = part.as<double>(); A.counters = M.prof ? h->d_counters.as<unsigned long long>() : nullptr;
  A.flags = h->d_flags.as<int>(); A.last = M.skip_done ? h->d_last.as<int>() : nullptr; A.eager = M.eager;
  // a shard launches only the ranges that can reach it: per isotope block the groups whose cells
  // lie within Rc + 1 cells of [lo, hi) are one run of consecutive ranges (cnt_ge look-up)
  unsigned nw = (unsigned)h->nwaves;
  A.nseg = 0;
  if (h->windowed() && h->niso <= kWalkSegs && nw > 0) {
    const int Rc = nb / 2 - 1;
    const long long klo = std::max<long long>(0, h->lo - Rc - 1), khi = std::min<long long>(h->nwn - 1, h->hi - 1 + Rc);
    int cum = 0, ns = 0;
    for (int b = 0; b < h->niso; b++) {
      const int gb0 = h->h_gblock[b], gb1 = h->h_gblock[b + 1];
      if (gb0 == gb1) continue;
      const int32_t *cg = &h->h_cntge[(size_t)b * (h->nwn + 1)];
      const int ga = cg[khi + 1], gz = cg[klo];                 // groups of the block (descending cells) with cell in [klo, khi]
      if (gz <= ga) continue;
      const int wa = h->h_wbase[b] + ga / h->ngw, wz = h->h_wbase[b] + (gz + h->ngw - 1) / h->ngw;
      A.seg_w0[ns] = wa; A.seg_cum[ns] = cum; cum += wz - wa; ns++;
    }
    A.seg_cum[ns] = cum; A.nseg = ns;
    nw = (unsigned)cum;
    if (ns == 0) { A.nseg = 1; A.seg_w0[0] = 0; A.seg_cum[0] = 0; A.seg_cum[1] = 0; }     // nothing reaches: no wave does anything
  }
  // steps of few layers with wide frames on a dense list: lanes = lines for the strengths (trx_lanes.hip.h)
  const bool lanes_prod = nw > 0 && A.tabw != nullptr && nb >= 8 && nc <= kLanesMaxLayers && h->lanes_walk &&
                          h->max_gcount <= kLanesMaxGroup && (h->ngroups >= 8 * h->nwn || h->lanes_force);
  // steps of few layers: several ranges per wave (k_line_walk_packed: an instruction serves S lines)
  const bool packed_prod = !lanes_prod && nw > 0 && A.tabw != nullptr && nc <= h->packed_max_layers && h->packed_walk;
  // (a counting run -- instrumented k_line_walk for every step -- books its layers under the form the production run
  // takes for them: its per-layer counters are what prices each form's bytes, bench.py)
  const bool lanes = lanes_prod && !M.prof, packed = packed_prod && !M.prof;
  const int form_prod = lanes_prod ? 1 : packed_prod ? 2 : 0, form = lanes ? 1 : packed ? 2 : 0;
  if (form_out) *form_out = form_prod;
  h->stats.walk_form_steps[form_prod]++; h->stats.walk_form_layers[form_prod] += nc; h->stats.walk_form_record_lanes[form_prod] += pl->records * nc;
  if (sp && sp->begin(form == 1 ? Spans::kWalkLanes : form == 2 ? Spans::kWalkPacked : Spans::kWalk, st)) return fail(h, TRX_E_HIP, "event");
  if (lanes) {
    // one range per wave (round 4 measured 2, 3, 4 ranges per wave at the demo size: 0.257 / 0.283 / 0.276 ms for the
    // spectrum's two walks against 0.239 -- the last lines of a range already get lanes = (line, 4 or 8 layer sets);
    // the kernel no longer has that form)
    LanesExtra X{h->d_linebase.as<double>()};
    A.xcd_map = h->xcd_map & 1;
    if (log_sink().fn && log_sink().max_level >= TRX_LOG_DEBUG)
      log_msg(TRX_LOG_DEBUG, "walk: lanes = lines, " + std::to_string(nc) + " layers, " + std::to_string(nb) + "-bin frames");
    const dim3 grid((nw + kLanesWaves - 1) / kLanesWaves), block(64 * kLanesWaves);
    const size_t lds = lanes_lds_bytes(nc, h->ndop);
    if (nb == 8) hipLaunchKernelGGL((k_line_walk_lanes<8, 5>), grid, block, lds, st, A, X);       // (blocks of 5 groups: a batch's ~28 are 6 blocks, an even number; 4: 102.2 us, 5: 101.2, 6: 103.6, round 5)
    else         hipLaunchKernelGGL((k_line_walk_lanes<16, 2>), grid, block, lds, st, A, X);      // (blocks of 2 groups -- 8 floats per group and lane, 112 registers: 169.5 us; 3 / 4: 130 / 150 registers, 171.0 / 172.6, round 5)
  }
  else if (packed) {
    const int S = 64 / nc;
    const unsigned pw = (nw + (unsigned)S - 1) / (unsigned)S;
    const dim3 grid((pw + kWalkWaves - 1) / kWalkWaves), block(64 * kWalkWaves);
    if (nb <= 4)      hipLaunchKernelGGL(k_line_walk_packed<4>, grid, block, 0, st, A, S);      // (a 2-bin step in the 4-bin row form: the same values)
    else if (nb == 8) hipLaunchKernelGGL(k_line_walk_packed<8>, grid, block, 0, st, A, S);
    else              hipLaunchKernelGGL(k_line_walk_packed<16>, grid, block, 0, st, A, S);
  }
  else if (nw > 0) {
    // (no row copy -- it would have passed 4 GB: the wide frames run their per-bin form, which is
    // the counting instantiation with the counters switched off)
    const bool per_bin = M.prof || (nb >= kWalkRowsFrom && A.tabw == nullptr);
    if (nb == 2) launch_walk<2>(A, per_bin, nw, st);
    else if (nb == 4) launch_walk<4>(A, per_bin, nw, st);
    else if (nb == 8) launch_walk<8>(A, per_bin, nw, st);
    else launch_walk<16>(A, per_bin, nw, st);
  }
  if (sp && sp->end(st)) return fail(h, TRX_E_HIP, "event");
  PendingCombine pc;
  pc.valid = true; pc.sc = st_comb ? st_comb : st; pc.st_walk = st; pc.cross = st_comb != nullptr && ev_walk != nullptr;
  pc.ev_walk = ev_walk; pc.ev_done = ev_done;
  if (pc.cross) HIPCHK(h, hipEventRecord(ev_walk, st));
  CombineArgs &C = pc.C;
  C.P = P; C.niso = h->niso; C.gblock = h->d_gblock.as<int32_t>(); C.lo = h->lo; C.nsh = h->nsh; C.r_top = r_top; C.nc = nc;
  C.nmx = M.nmx; C.iso_mx = M.d_iso_mx; C.part = part.as<double>(); C.e = M.d_e;
  C.flags = h->d_flags.as<int>(); C.last = A.last; C.eager = M.eager;
  if (defer) { *defer = pc; return TRX_OK; }
  return launch_combine(h, pc, sp);
}

// The two-kernel form (profiles wider than the walk's largest frame): strengths, accumulation.
int sweep_chunk(trx_handle *h, const LayerDev &Y, const double *d_wcut, const int32_t *psmax,
                int r_top, int nc, int nc_max, const SweepMode &M, Spans *sp)
{
  hipStream_t st = M.st ? M.st : h->stream;
  const int niso = h->niso; const int64_t nsh = h->nsh;
  const int ntiles = (int)((nsh + kTileBins - 1) / kTileBins);
  double *d_SG = h->d_SG.as<double>();
  uint8_t *d_idop8 = h->d_idop8.as<uint8_t>();
  // lines whose profiles can reach this shard in any layer of the step (contiguous per isotope block):
  // their number here, for the launch; the runs themselves are found by the kernel (SweepWindow)
  SweepWindow Wn{};
  Wn.windowed = h->windowed() ? 1 : 0; Wn.osamp = h->osamp; Wn.lo = h->lo; Wn.hi = h->hi; Wn.nwn = h->nwn;
  long long seg_lines = 0;
  for (int b = 0; b < niso; b++) {
    const int gb0 = h->h_gblock[b], gb1 = h->h_gblock[b + 1];
    if (gb0 == gb1) continue;
    int ga = gb0, gz = gb1;
    if (h->windowed()) {
      long long psm = 0;
      for (int c = 0; c < nc; c++) psm = std::max<long long>(psm, psmax[(size_t)(r_top - c) * niso + b]);
      const long long lo_f = (long long)h->osamp * h->lo - psm;
      long long klo = lo_f > 0 ? lo_f / h->osamp : 0;
      long long khi = ((long long)h->osamp * (h->hi - 1) + psm) / h->osamp;
      if (khi > h->nwn - 1) khi = h->nwn - 1;
      const int32_t *cg = &h->h_cntge[(size_t)b * (h->nwn + 1)];
      ga = gb0 + cg[khi + 1]; gz = gb0 + cg[klo];
    }
    if (ga >= gz) continue;
    seg_lines += ((long long)h->h_gfirst[gz - 1] + h->h_gcount[gz - 1]) - h->h_gfirst[ga];
  }
  constexpr unsigned kSpan = kXcds * kAccumXcdGroup;
  const unsigned tblocks = (unsigned)(((ntiles + 3) / 4 + kSpan - 1) / kSpan * kSpan);   // multiple of 8*G: xcd_grouped_x()
  if (sp && sp->begin(Spans::kSweep, st)) return fail(h, TRX_E_HIP, "event");
  if (seg_lines > 0) {
    hipLaunchKernelGGL(k_group_sweep, dim3((unsigned)((seg_lines + 255) / 256)), dim3(256), sizeof(long long) * (2 * (size_t)niso + 1), st,
                       h->L, Y, Wn, niso, r_top, nc, h->d_dopthr.as<double>(), h->ndop, h->d_e2tab.as<double>(), d_wcut,
                       d_SG, d_idop8, h->d_flags.as<int>(), (int)M.eager);
  }
  if (sp && (sp->end(st) || sp->begin(Spans::kAccum, st))) return fail(h, TRX_E_HIP, "event");
  if (h->ngroups > 0) {
    AccumArgs A{};
    A.L = h->L; A.Y = Y; A.niso = niso; A.nlor = h->nlor; A.ndop = h->ndop; A.osamp = h->osamp;
    A.nwn = h->nwn; A.lo = h->lo; A.nsh = nsh; A.r_top = r_top; A.nc = nc; A.ntiles = ntiles;
    A.SG = d_SG; A.idop8 = d_idop8; A.sticky_idop = M.d_sticky;
    A.kmaxc = M.d_kmax; A.ethresh = M.ethresh; A.nmx = M.nmx; A.iso_mx = M.d_iso_mx; A.permol = M.permol;
    A.psize = h->d_psize.as<int32_t>(); A.poff = h->d_poff.as<long long>();
    A.table = h->tab; A.e = M.d_e;
    // (profiled runs count every group in the tile of its own coarse cell: whole-cell windows)
    A.sub_f = M.prof ? 1 : h->sub_f; A.cnt_sub = A.sub_f > 1 ? h->d_cntsub.as<int32_t>() : h->d_cntge.as<int32_t>();
    A.part = M.prof ? h->d_part3.as<unsigned long long>() : nullptr; A.part_stride = (int)tblocks;
    A.flags = h->d_flags.as<int>(); A.eager = M.eager;
    A.last = M.skip_done ? h->d_last.as<int>() : nullptr;
    // layers whose profiles span >= 64 coarse bins go to the lanes-own-bins kernel
    unsigned wide_mask = 0;
    for (int c = 0; c < nc; c++)
      if ((2 * layer_psmax(h, psmax, r_top - c)) / h->osamp + 1 >= 64) wide_mask |= 1u << c;   // measured: 16 or 32 here is 4x slower at configs[2] size, 128/256 no better
    A.skip_mask = wide_mask;
    if (M.prof) HIPCHK(h, hipMemsetAsync(h->d_part3.p, 0, 24 * (size_t)nc_max * tblocks, st));
    if (wide_mask != (nc >= 32 ? 0xffffffffu : ((1u << nc) - 1u)))
      hipLaunchKernelGGL(k_accumulate, dim3(tblocks, (unsigned)nc), dim3(256), 0, st, A);
    if (wide_mask) {
      WideArgs W{}; W.A = A; W.tabT = h->tabT; W.poffT = h->poffT;
      W.gimod = h->d_gimod.as<int32_t>(); W.gidiv = h->d_gidiv.as<int32_t>(); W.layer_mask = wide_mask;
      const unsigned wtiles = (unsigned)((nsh + 64 * kWideM - 1) / (64 * kWideM));
      // no oversampling: every group of a profile reads the same row -- staged in LDS (trx_rows.hip.h);
      // layers whose profiles are much wider than a tile take tiles of 512 bins, the others of 256
      if (h->osamp == 1 && h->row_staging) {
        unsigned m8 = 0;
        for (int c = 0; c < nc; c++)
          if (((wide_mask >> c) & 1u) && 2 * layer_psmax(h, psmax, r_top - c) + 1 >= h->row_m8_from) m8 |= 1u << c;
        auto launch = [&](unsigned mask, int m) {
          if (!mask) return;
          W.layer_mask = mask;
          const unsigned tiles = (unsigned)((nsh + 64 * m - 1) / (64 * m));
          const dim3 grid((tiles + 3) / 4, (unsigned)nc);
          const size_t lds = rows_lds_bytes(h->ndop, m);
          if (M.prof) hipLaunchKernelGGL((k_accumulate_rows<true, 4>), grid, dim3(256), lds, st, W);      // (counting runs: one tile size)
          else if (m == 8) hipLaunchKernelGGL((k_accumulate_rows<false, 8>), grid, dim3(256), lds, st, W);
          else hipLaunchKernelGGL((k_accumulate_rows<false, 4>), grid, dim3(256), lds, st, W);
        };
        if (M.prof) launch(wide_mask, 4);
        else { launch(wide_mask & ~m8, 4); launch(m8, 8); }
      }
      else hipLaunchKernelGGL(k_accumulate_wide, dim3((wtiles + 3) / 4, (unsigned)nc), dim3(256), 0, st, W);
    }
  }
  if (sp && sp->end(st)) return fail(h, TRX_E_HIP, "event");
  if (M.prof && h->ngroups > 0) {      // counters (profiling runs only; gated like the sweep itself)
    for (int k = 0; k < 3; k++)
      hipLaunchKernelGGL(k_sum_parts_gated, dim3((unsigned)nc), dim3(256), 0, st, h->d_part3.as<unsigned long long>(),
                         (int)tblocks, 3, k, h->d_counters.as<unsigned long long>(), 3, r_top, h->d_flags.as<int>(), (int)M.eager);
  }
  return TRX_OK;
}

// Strongest single line and sticky Doppler index of ALL nv layers (states) at once: both depend
// on the inputs only, not on how far the rays get, so they leave the per-step chain.
// kmax: [nv][nmx], zero on entry.  init: the run's small buffers, initialised by an extra row of
// blocks of the first launch (null: none); *init_done tells whether that happened.
// Two launches, each usable on its own: trx_run queues the maxima BEFORE the rest of its host prologue
// (they need -c/T and the strength factors only, which it hands over in pinned host memory).
int launch_layer_max(trx_handle *h, const LayerDev &Y, int nv, const double *temp_k, int nmx, const int32_t *d_iso_mx,
                     hipStream_t st, double *kmax, const RunInit *init = nullptr, bool *init_done = nullptr)
{
  if (init_done) *init_done = false;
  if (h->ngroups == 0) return TRX_OK;
  // the pruning argument needs c*nu/T well above the rounding of 1 - exp(-c*nu/T) (trx_walk.hip.h)
  bool pruned = h->ncand > 0;
  for (int r = 0; r < nv && pruned; r++) if (kExpCte * kTliEfct * h->wn_i / temp_k[r] < 1e-5) pruned = false;
  const long long n = pruned ? h->ncand : h->nlines;
  for (int r0 = 0; r0 < nv; r0 += 32768) {
    const int nr = std::min(32768, nv - r0);
    LayerDev Yr = Y; Yr.negc_over_t += r0; Yr.strength_f += (size_t)r0 * h->niso;
    const unsigned ny = (unsigned)((nr + kLayerMaxGroup - 1) / kLayerMaxGroup);
    const bool with_init = init && r0 == 0;
    RunInit R{}; R.nsh = -1;
    if (with_init) { R = *init; *init_done = true; }
    // (at most kLayerMaxWaves waves per group of layers: beyond that a wave takes several chunks of candidates)
    const int xwaves = (int)std::min<long long>((n + 64 * kLayerMaxLines - 1) / (64 * kLayerMaxLines), kLayerMaxWaves);
    hipLaunchKernelGGL(k_layer_max, dim3((unsigned)((xwaves + kLayerMaxBlock - 1) / kLayerMaxBlock), ny + (with_init ? 1u : 0u)), dim3(64 * kLayerMaxBlock), sizeof(double) * (size_t)kLayerMaxGroup * (size_t)std::max(h->niso, 1), st,
                       h->L, Yr, h->niso, nr, pruned ? h->d_candrec.as<CandLine>() : nullptr, n, h->d_e2tab.as<double>(), nmx, d_iso_mx,
                       (unsigned long long *)(kmax + (size_t)r0 * nmx), R, with_init ? (int)ny : -1, xwaves);
  }
  return TRX_OK;
}

// (copy_*: the run's input block from pinned host memory into device memory, by extra blocks of the first launch)
int launch_sticky(trx_handle *h, const LayerDev &Y, const int32_t *d_npre, int nv, int nmx, const int32_t *d_iso_mx, double ethresh,
                  hipStream_t st, const double *kmax, const void *copy_src = nullptr, void *copy_dst = nullptr, size_t copy_bytes = 0)
{
  const long long n16 = (long long)(copy_bytes / 16);
  if (h->ngroups == 0 && n16 == 0) return TRX_OK;
  bool copied = n16 == 0;
  for (int r0 = 0; r0 < nv || !copied; r0 += 4096) {               // one wave per (layer, isotope)
    const int nr = h->ngroups == 0 ? 0 : std::max(0, std::min(4096, nv - r0));
    const int nst = nr * h->niso;
    const int ncp = copied ? 0 : (int)std::min<long long>((n16 + 63) / 64, 64);
    if (nst + ncp == 0) break;
    hipLaunchKernelGGL(k_sticky_index, dim3((unsigned)(nst + ncp)), dim3(64), 0, st,
                       h->L, Y, h->niso, r0 + nr - 1, nr, kmax, nmx, d_iso_mx, ethresh, h->d_dopthr.as<double>(), h->ndop,
                       h->d_e2tab.as<double>(), d_npre, h->d_sticky.as<int>(), h->d_flags.as<int>(), 1,
                       copied ? nullptr : (const uint4 *)copy_src, copied ? nullptr : (uint4 *)copy_dst, copied ? 0LL : n16, nst);
    copied = true;
  }
  HIPCHK(h, hipGetLastError());
  return TRX_OK;
}

int layer_maxima_and_sticky(trx_handle *h, const LayerDev &Y, const int32_t *d_npre, int nv, const double *temp_k,
                            int nmx, const int32_t *d_iso_mx, double ethresh, hipStream_t st, double *kmax,
                            const RunInit *init = nullptr, bool *init_done = nullptr)
{
  int rc;
  if (init_done) *init_done = false;
  if ((rc = ensure(h, h->d_sticky, sizeof(int) * (size_t)nv * std::max(h->niso, 1)))) return rc;
  if (h->ngroups == 0) return TRX_OK;
  if ((rc = launch_layer_max(h, Y, nv, temp_k, nmx, d_iso_mx, st, kmax, init, init_done))) return rc;
  return launch_sticky(h, Y, d_npre, nv, nmx, d_iso_mx, ethresh, st, kmax);
}

}  // namespace

// ============================================================================
extern "C" {

int trx_comm_unique_id(void *id_out)
{
  if (!id_out) return TRX_E_ARG;
  static_assert(sizeof(ncclUniqueId) == TRX_COMM_ID_BYTES, "unique id size");
  Rccl &R = rccl();
  if (!R.ok()) return TRX_E_UNSUPPORTED;
  ncclUniqueId id;
  if (R.GetUniqueId(&id) != ncclSuccess) return TRX_E_HIP;
  std::memcpy(id_out, &id, sizeof(id));
  return TRX_OK;
}

// what went wrong in a call that has no handle to keep its message (trx_last_error(NULL))
thread_local std::string g_comm_err;

int trx_comm_create(const void *idp, int nranks, int rank, int device, void **comm_out)
{
  g_comm_err.clear();
  if (!idp || !comm_out || nranks < 1 || rank < 0 || rank >= nranks) return TRX_E_ARG;
  Rccl &R = rccl();
  if (!R.ok()) { g_comm_err = "librccl.so could not be loaded"; return TRX_E_UNSUPPORTED; }
  if (hipSetDevice(device) != hipSuccess) return TRX_E_NODEVICE;
  ncclUniqueId id; std::memcpy(&id, idp, sizeof(id));
  ncclComm_t c = nullptr;
  const ncclResult_t st = R.CommInitRank(&c, nranks, id, rank);
  if (st != ncclSuccess) {
    g_comm_err = std::string("ncclCommInitRank: ") + (R.GetErrorString ? R.GetErrorString(st) : "failed");
    if (R.GetLastError) { const char *m = R.GetLastError(nullptr); if (m && *m) g_comm_err += std::string(" -- ") + m; }
    return TRX_E_HIP;
  }
  *comm_out = (void *)c;
  return TRX_OK;
}

void trx_comm_destroy(void *comm)
{
  if (comm && rccl().ok()) (void)rccl().CommDestroy((ncclComm_t)comm);
}

void trx_comm_abort(void *comm)
{
  if (!comm || !rccl().ok()) return;
  if (rccl().CommAbort) (void)rccl().CommAbort((ncclComm_t)comm);
  else (void)rccl().CommDestroy((ncclComm_t)comm);
}


int trx_restore_extinction(trx_handle *h, int32_t nlayer, const double *e, const uint8_t *computed)
{
  if (!h || nlayer < 0 || (nlayer > 0 && (!e || !computed))) return TRX_E_ARG;
  h->saved.clear();
  if (nlayer == 0) return TRX_OK;
  HIPCHK(h, hipSetDevice(h->device));
  int rc = ensure(h, h->d_e_saved, sizeof(double) * (size_t)nlayer * (size_t)h->nsh);
  if (rc) return rc;
  HIPCHK(h, hipMemcpy(h->d_e_saved.p, e, sizeof(double) * (size_t)nlayer * (size_t)h->nsh, hipMemcpyHostToDevice));
  h->saved.assign(computed, computed + nlayer);
  return TRX_OK;
}

int trx_abi_version(void) { return TRX_ABI_VERSION; }

// HIP version the library was built with and the one of the runtime it is running on (a process
// may have mapped another copy of libamdhip64.so first, e.g. the one a PyTorch wheel bundles)
int trx_hip_versions(int *built, int *running)
{
  if (built) *built = HIP_VERSION;
  int v = 0;
  if (hipRuntimeGetVersion(&v) != hipSuccess) return TRX_E_HIP;
  if (running) *running = v;
  return TRX_OK;
}

int trx_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char *trx_strerror(int st)
{
  switch (st) {
    case TRX_OK: return "success";
    case TRX_E_ARG: return "bad argument or inconsistent sizes";
    case TRX_E_NOMEM: return "host or device allocation failed";
    case TRX_E_HIP: return "HIP runtime call failed";
    case TRX_E_NODEVICE: return "no usable gfx950 device";
    case TRX_E_RANGE: return "value outside a table range";
    case TRX_E_UNSUPPORTED: return "option combination not implemented";
    case TRX_E_ORDER: return "line list is not sorted as the TLI format requires";
    case TRX_E_NOTREACHED: return "optical depth never reached toomuch (modlevel -1)";
  }
  return "unknown status";
}

const char *trx_last_error(const trx_handle *h) { return h ? h->err.c_str() : g_comm_err.c_str(); }

// The single exchange of a wavenumber-sharded job: every rank contributes `count` doubles (its
// spectrum slice, padded to the same length on every rank) and receives all of them, rank
// order, in d_all -- ncclAllGather on the handle's stream, synchronised on return.  Without a
// communicator (one rank) it is a device copy.
int trx_gather(trx_handle *h, const void *d_slice, void *d_all, int64_t count)
{
  if (!h || !d_slice || !d_all || count < 0) return TRX_E_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  if (h->comm) {
    if (rccl().AllGather(d_slice, d_all, (size_t)count, ncclDouble, (ncclComm_t)h->comm, h->stream) != ncclSuccess)
      return fail(h, TRX_E_HIP, "ncclAllGather failed");
  } else if (d_all != d_slice) {
    HIPCHK(h, hipMemcpyAsync(d_all, d_slice, sizeof(double) * (size_t)count, hipMemcpyDeviceToDevice, h->stream));
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return TRX_OK;
}

// trx_gather for callers that keep their slices in host memory (the command-line driver): the
// slice goes to the device, is gathered there, and the whole comes back -- on every rank.
int trx_gather_host(trx_handle *h, const double *slice, double *all, int64_t count)
{
  if (!h || !slice || !all || count < 0) return TRX_E_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  const int nranks = h->comm ? h->nranks : 1;
  DevBuf d_slice, d_all;
  int rc;
  if ((rc = ensure(h, d_slice, sizeof(double) * (size_t)std::max<int64_t>(count, 1))) ||
      (rc = ensure(h, d_all, sizeof(double) * (size_t)std::max<int64_t>(count, 1) * nranks))) return rc;
  HIPCHK(h, hipMemcpyAsync(d_slice.p, slice, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, h->stream));
  if ((rc = trx_gather(h, d_slice.p, d_all.p, count))) return rc;
  HIPCHK(h, hipMemcpy(all, d_all.p, sizeof(double) * (size_t)count * nranks, hipMemcpyDeviceToHost));
  return TRX_OK;
}

void trx_set_log(trx_log_fn fn, void *user, int max_level)
{
  LogSink &s = log_sink();
  s.fn = fn; s.user = user; s.max_level = max_level;
}

int trx_create(const trx_static *s, trx_handle **out)
{
  if (!s || !out) return TRX_E_ARG;
  *out = nullptr;
  if (s->abi_version != TRX_ABI_VERSION) return TRX_E_ARG;
  if (s->nwn < 2 || s->osamp < 1 || s->nown != (s->nwn - 1) * s->osamp + 1) return TRX_E_ARG;
  if (s->ndop < 2 || s->nlor < 2 || s->ndop > kMaxDop) return TRX_E_UNSUPPORTED;
  if (s->niso < 0 || s->niso > kMaxIso || s->nmol < 1) return TRX_E_UNSUPPORTED;
  if (s->wn_lo < 0 || s->wn_hi > s->nwn || s->wn_lo >= s->wn_hi) return TRX_E_ARG;
  if (s->nlines < 0 || (s->nlines > 0 && (!s->wl_um || !s->isoid || !s->elow || !s->gf))) return TRX_E_ARG;
  if (s->nown > 2000000000LL) return TRX_E_UNSUPPORTED;     // iown/beg_j are int in the reference too
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || s->device < 0 || s->device >= ndev) return TRX_E_NODEVICE;
  trx_handle *h = new (std::nothrow) trx_handle();
  if (!h) return TRX_E_NOMEM;
  h->device = s->device;
  int rc = TRX_OK;
  auto bail = [&](int code) { trx_destroy(h); return code; };
  if (hipSetDevice(h->device) != hipSuccess) return bail(TRX_E_NODEVICE);
  {
    // Three streams that really overlap: the runtime hands out hardware queues round-robin, and with
    // other streams in the process (an RCCL communicator, the caller's own) two of ours can land on
    // ONE queue -- the CIA kernels then ran after the walks instead of under them, +0.11 ms per
    // spectrum.  So every new stream is probed against the ones it must overlap and replaced (the
    // next creation gets the next queue) until it does; after 12 tries it is taken as it is.
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return bail(TRX_E_HIP);
    DevBuf probe;
    if ((rc = ensure(h, probe, 64))) return bail(rc);
    int *d_flag = probe.as<int>(), *d_saw = d_flag + 8;
    auto overlaps = [&](hipStream_t a, hipStream_t b) -> bool {
      int saw = 0;
      if (hipMemsetAsync(d_flag, 0, 64, a) != hipSuccess || hipStreamSynchronize(a) != hipSuccess) return true;
      hipLaunchKernelGGL(k_probe_wait, dim3(1), dim3(1), 0, a, (volatile int *)d_flag, d_saw);
      hipLaunchKernelGGL(k_probe_set, dim3(1), dim3(1), 0, b, (volatile int *)d_flag);
      if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return true;
      if (hipMemcpy(&saw, d_saw, sizeof saw, hipMemcpyDeviceToHost) != hipSuccess) return true;
      return saw != 0;
    };
    auto concurrent_stream = [&](hipStream_t *out, hipStream_t with1, hipStream_t with2) -> int {
      std::vector<hipStream_t> rejected;
      int code = TRX_OK;
      for (int attempt = 0; attempt < 12; attempt++) {
        hipStream_t s2 = nullptr;
        if (hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess) { code = TRX_E_HIP; break; }
        if (attempt == 11 || (overlaps(with1, s2) && (!with2 || overlaps(with2, s2)))) { *out = s2; break; }
        rejected.push_back(s2);                  // kept alive until the end: destroying it would free its slot for the next try
      }
      for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
      return code;
    };
    if ((rc = concurrent_stream(&h->stream4, h->stream, nullptr)) || (rc = concurrent_stream(&h->stream2, h->stream, h->stream4))) return bail(rc);
  }
  if (hipEventCreateWithFlags(&h->ev_walk1, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_inputs, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->ev_cia, hipEventDisableTiming) != hipSuccess) return bail(TRX_E_HIP);
  h->wn_i = s->wn_i; h->wn_d = s->wn_d; h->osamp = s->osamp; h->odwn = s->wn_d / s->osamp;
  h->nwn = s->nwn; h->nown = s->nown; h->lo = s->wn_lo; h->hi = s->wn_hi; h->nsh = s->wn_hi - s->wn_lo;
  h->niso = s->niso; h->nmol = s->nmol; h->ndop = s->ndop; h->nlor = s->nlor;
  h->comm = s->comm; h->nranks = s->comm ? std::max(1, s->nranks) : 1; h->rank = s->rank;
  if (h->comm && !rccl().ok()) return bail(TRX_E_UNSUPPORTED);
  h->iso_mass.assign(s->iso_mass, s->iso_mass + s->niso);
  h->iso_sqrtm.resize(s->niso);
  for (int i = 0; i < s->niso; i++) h->iso_sqrtm[i] = std::sqrt(h->iso_mass[i]);
  h->iso_ratio.assign(s->iso_ratio, s->iso_ratio + s->niso);
  h->iso_imol.assign(s->iso_imol, s->iso_imol + s->niso);
  h->mol_mass.assign(s->mol_mass, s->mol_mass + s->nmol);
  h->mol_radius.assign(s->mol_radius, s->mol_radius + s->nmol);
  if (s->mol_pol) h->mol_pol.assign(s->mol_pol, s->mol_pol + s->nmol); else h->mol_pol.assign(s->nmol, 0.0);
  if (s->mol_is_h2) h->mol_is_h2.assign(s->mol_is_h2, s->mol_is_h2 + s->nmol); else h->mol_is_h2.assign(s->nmol, 0);
  for (int i = 0; i < s->niso; i++) if (s->iso_imol[i] < 0 || s->iso_imol[i] >= s->nmol) return bail(TRX_E_ARG);
  h->pair_csd.assign((size_t)s->niso * s->nmol, 0.0); h->pair_sqrt.assign((size_t)s->niso * s->nmol, 0.0);
  for (int i = 0; i < s->niso; i++)
    for (int j = 0; j < s->nmol; j++) {
      h->pair_csd[(size_t)i * s->nmol + j] = h->mol_radius[j] + h->mol_radius[h->iso_imol[i]];
      h->pair_sqrt[(size_t)i * s->nmol + j] = std::sqrt(1 / h->iso_mass[i] + 1 / h->mol_mass[j]);
    }
  for (int k = 0; k < s->ncia; k++) {
    const trx_cia &c = s->cia[k];
    if (c.nspec < 1 || c.nspec > 2 || c.nwave < 3 || c.ntemp < 3) return bail(TRX_E_ARG);
    trx_handle::Cia t; t.nspec = c.nspec; t.mol[0] = c.mol[0]; t.mol[1] = c.mol[1];
    t.wn.assign(c.wn, c.wn + c.nwave); t.temp.assign(c.temp, c.temp + c.ntemp);
    t.cs.assign(c.cs, c.cs + (size_t)c.nwave * c.ntemp);
    // table-only halves of the two natural splines (spline_init, pu/src/spline.c:186-206):
    // second derivatives along T of every row, and the pivots of the wavenumber sweep
    {
      const size_t nw = (size_t)c.nwave, nt = (size_t)c.ntemp;
      std::vector<double> u(std::max(nw, nt)), v(std::max(nw, nt));
      t.zt.resize(nw * nt);
      for (size_t i = 0; i < nw; i++)
        spline_second_derivs(t.zt.data() + i * nt, t.temp.data(), t.cs.data() + i * nt, (long)nt, u.data(), v.data());
      std::vector<double> zdummy(nw), ydummy(nw, 0.0);
      t.uw.assign(nw, 0.0);
      spline_second_derivs(zdummy.data(), t.wn.data(), ydummy.data(), (long)nw, t.uw.data(), v.data());
      t.ruw.assign(nw, 0.0);                 // reciprocal pivots: the per-layer sweeps multiply instead of dividing
      for (size_t i = 0; i < (size_t)nw; i++) if (t.uw[i] != 0.0) t.ruw[i] = 1.0 / t.uw[i];
      t.rh.assign(nw, 0.0);                  // reciprocal spacings 1/(wn[i+1]-wn[i])
      for (size_t i = 0; i + 1 < (size_t)nw; i++) t.rh[i] = 1.0 / (t.wn[i + 1] - t.wn[i]);
      // weights of the sums that stand for the two sweeps (k_cia_v, k_cia_z): products of the sweeps' row-to-row factors,
      // kept where kCiaTerms of them leave less than 2^-60 everywhere in the table
      if (nw >= 8) {
        const size_t K = kCiaTerms;
        std::vector<double> wf(nw * K, 0.0), wb(nw * K, 0.0);
        double worst = 0.0;
        for (size_t i = 1; i + 1 < nw; i++) {
          double p = 1.0;                                   // forward: (-1)^k c[i] c[i-1] .. c[i-k+1], c[m] = h[m-1] / u[m-1] (m >= 2)
          for (size_t k = 0; k <= K; k++) {
            if (i < 1 + k) break;                           // row i - k < 1
            if (k < K) wf[i * K + k] = p; else worst = std::fmax(worst, std::fabs(p));
            const size_t m = i - k;                         // next factor: -c[m]
            if (m < 2) break;
            p *= -((t.wn[m] - t.wn[m - 1]) * t.ruw[m - 1]);
          }
          p = 1.0;                                          // backward: (-1)^k d[i] .. d[i+k-1] / u[i+k], d[m] = h[m] / u[m]
          for (size_t k = 0; k <= K; k++) {
            const size_t m = i + k;
            if (m + 1 >= nw) break;                         // row i + k > n - 2
            if (k < K) wb[i * K + k] = p * t.ruw[m]; else worst = std::fmax(worst, std::fabs(p));
            p *= -((t.wn[m + 1] - t.wn[m]) * t.ruw[m]);
          }
        }
        if (worst < 0x1p-60) { t.wf = std::move(wf); t.wb = std::move(wb); }
        if (log_sink().fn && log_sink().max_level >= TRX_LOG_DEBUG) {
          char msg[160];
          std::snprintf(msg, sizeof msg, "create: CIA table %d (%zu rows): second derivatives by %s (term %d of the row sums at most %.1e of the first)",
                        k, nw, t.wf.empty() ? "the two sweeps" : "sums per row", kCiaTerms, worst);
          log_msg(TRX_LOG_DEBUG, msg);
        }
      }
    }
    h->cia.push_back(std::move(t));
  }
  if (s->ogrid) {
    const trx_opacity_grid *g = s->ogrid;
    if (g->nmol < 1 || g->ntemp < 2 || g->nlayer < 1 || g->nwave != s->nwn || !g->o || !g->temp || !g->mol_index) return bail(TRX_E_ARG);
    for (long m = 0; m < g->nmol; m++) if (g->mol_index[m] < 0 || g->mol_index[m] >= s->nmol) return bail(TRX_E_ARG);
    h->has_grid = true; h->og_nmol = g->nmol; h->og_ntemp = g->ntemp; h->og_nlayer = g->nlayer; h->og_nwave = g->nwave;
    h->og_temp.assign(g->temp, g->temp + g->ntemp); h->og_temp.push_back(HUGE_VAL);   // searched with hi = Ntemp (extinction.c:562)
    h->og_molidx.assign(g->mol_index, g->mol_index + g->nmol);
    const size_t no = (size_t)g->nlayer * g->ntemp * g->nmol * g->nwave;
    if ((rc = ensure(h, h->d_og_o, sizeof(double) * no))) return bail(rc);
    if (hipMemcpy(h->d_og_o.p, g->o, sizeof(double) * no, hipMemcpyHostToDevice) != hipSuccess) return bail(TRX_E_HIP);
  }
  for (auto &c : h->cia)
    if ((rc = upload(h, c.d_wn, c.wn)) || (rc = upload(h, c.d_temp, c.temp)) || (rc = upload(h, c.d_cs, c.cs)) ||
        (rc = upload(h, c.d_zt, c.zt)) || (rc = upload(h, c.d_uw, c.uw)) || (rc = upload(h, c.d_ruw, c.ruw)) || (rc = upload(h, c.d_rh, c.rh)) ||
        (!c.wf.empty() && ((rc = upload(h, c.d_wf, c.wf)) || (rc = upload(h, c.d_wb, c.wb))))) return bail(rc);
  // the handle does not survive a failed create, so its error text cannot be asked for later:
  // without a message callback it goes to stderr
  auto say = [&]() { if (!log_sink().fn) std::fprintf(stderr, "trx_create: %s\n", h->err.c_str()); };
  StageTimer TC;
  if ((rc = build_table(h, s)) != TRX_OK) { *out = nullptr; say(); return bail(rc); }
  TC.lap("Voigt table (plan+kernels+sync)");
  if ((rc = prepare_lines(h, s)) != TRX_OK) { say(); return bail(rc); }
  {
    char buf[256];
    std::snprintf(buf, sizeof buf, "trx_create: %lld lines (%lld in range, %lld co-added groups), %lld wavenumbers [%lld,%lld), "
                  "Voigt table %dx%d = %lld floats, %zu CIA tables", (long long)s->nlines, (long long)h->stats.nlines_inrange,
                  (long long)h->ngroups, (long long)s->nwn, (long long)h->lo, (long long)h->hi, s->ndop, s->nlor,
                  (long long)h->tab_n, h->cia.size());
    log_msg(TRX_LOG_INFO, buf);
  }
  *out = h;
  return TRX_OK;
}

void trx_destroy(trx_handle *h)
{
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream2) { (void)hipStreamSynchronize(h->stream2); (void)hipStreamDestroy(h->stream2); }
  if (h->stream4) { (void)hipStreamSynchronize(h->stream4); (void)hipStreamDestroy(h->stream4); }
  for (auto e : h->ev_ac) (void)hipEventDestroy(e);
  for (auto e : h->ev_cb) (void)hipEventDestroy(e);
  if (h->stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
  if (h->ev_inputs) (void)hipEventDestroy(h->ev_inputs);
  if (h->ev_run_a) (void)hipEventDestroy(h->ev_run_a);
  if (h->ev_run_b) (void)hipEventDestroy(h->ev_run_b);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->ev_walk1) (void)hipEventDestroy(h->ev_walk1);
  if (h->ev_cia) (void)hipEventDestroy(h->ev_cia);
  if (h->h_small) (void)hipHostFree(h->h_small);
  if (h->h_in) (void)hipHostFree(h->h_in);
  if (h->h_spec) (void)hipHostFree(h->h_spec);
  if (h->h_tailblk) (void)hipHostFree(h->h_tailblk);
  delete h;
}

int trx_get_stats(const trx_handle *h, trx_stats *out)
{ if (!h || !out) return TRX_E_ARG; *out = h->stats; return TRX_OK; }

int trx_table_info(const trx_handle *h, int64_t *ps, int64_t *off, int64_t *total)
{
  if (!h) return TRX_E_ARG;
  const size_t n = (size_t)h->ndop * h->nlor;
  if (ps)  for (size_t k = 0; k < n; k++) ps[k] = h->psize[k];
  if (off) for (size_t k = 0; k < n; k++) off[k] = h->poff[k];
  if (total) *total = h->tab_n;
  return TRX_OK;
}

int trx_table_copy(const trx_handle *hc, float *out)
{
  trx_handle *h = const_cast<trx_handle *>(hc);
  if (!h || !out) return TRX_E_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipMemcpy(out, h->tab, sizeof(float) * (size_t)h->tab_n, hipMemcpyDeviceToHost));
  return TRX_OK;
}

int trx_width_grids(const trx_handle *h, double *adop, double *alor)
{
  if (!h) return TRX_E_ARG;
  if (adop) std::memcpy(adop, h->adop.data(), sizeof(double) * h->ndop);
  if (alor) std::memcpy(alor, h->alor.data(), sizeof(double) * h->nlor);
  return TRX_OK;
}

namespace {
}  // namespace

// ---- a run in three parts: what it is given (run_check, run_host_inputs), what it queues (run_once),
// ---- what it hands back (run_finish)
static int run_check(trx_handle *h, const trx_atm *a, const trx_opts *o)
{
  const int nr = a->nlayer;
  if (nr < 3) return fail(h, TRX_E_ARG, "at least three layers are needed");
  if (!h->saved.empty() && (int)h->saved.size() != nr) return fail(h, TRX_E_ARG, "restored extinction has another number of layers");
  if (!h->saved.empty() && h->has_grid) return fail(h, TRX_E_UNSUPPORTED, "restored extinction together with an opacity grid");
  if (o->solution != TRX_SOL_ECLIPSE && o->solution != TRX_SOL_TRANSIT) return fail(h, TRX_E_ARG, "unknown solution");
  if (o->solution == TRX_SOL_ECLIPSE && (o->nangles < 1 || o->nangles > kMaxAngles || !o->angles_deg))
    return fail(h, TRX_E_ARG, "eclipse needs 1..16 angles");
  if (o->solution == TRX_SOL_TRANSIT && o->modlevel != 1 && o->modlevel != -1) return fail(h, TRX_E_UNSUPPORTED, "modlevel must be 1 or -1");
  if (!(o->ethresh > 0)) return fail(h, TRX_E_ARG, "ethresh must be positive");
  if ((o->cloud_flag >= 2 || o->scat_flag == 2) && !a->abund && o->cloud_flag >= 2) return fail(h, TRX_E_ARG, "cloud model needs abundances");
  for (int i = 1; i < nr; i++) if (!(a->radius[i] > a->radius[i-1])) return fail(h, TRX_E_ARG, "radii must ascend");
  return TRX_OK;
}

// the host's share of a run's inputs beyond prep_layers: the layer scalars of the scattering / cloud
// models, the vertical rays' Simpson weights and per-layer chain constants, the impact parameters
static void run_host_inputs(trx_handle *h, const trx_atm *a, LayerHost &LH, bool vertical, int gstride, size_t mw_doubles,
                            size_t n_geom_all, std::vector<double> &geom, std::vector<double> &ipv)
{
  const int nr = a->nlayer, nmol = h->nmol;
  std::vector<double> &f64 = LH.f64;
  // layer-only scalars of the scattering / cloud models (tau.c:193-214, extinction.c:617-621)
  double *press = &f64[LH.extra_off], *tempk = press + nr, *mdens = tempk + nr, *nH = mdens + nr,
         *scat_pol = nH + nr, *radv = scat_pol + nr;
  for (int r = 0; r < nr; r++) {
    press[r] = a->press ? a->press[r] : 0.0; tempk[r] = a->temp[r]; radv[r] = a->radius[r];
    double mm = 0, md = 0, sp = 0;
    for (int j = 0; j < nmol; j++) {
      const double d = a->density[(size_t)j * nr + r];
      if (a->abund) {
        const double q = a->abund[(size_t)j * nr + r];
        md += d / h->mol_mass[j] * q;
        if (h->mol_is_h2[j]) nH[r] = d / h->mol_mass[j] * q * kNavo;
        mm += h->mol_mass[j] * q;
      }
      sp += kPi * 8e-32 / 3. * std::pow(h->mol_pol[j], 2) * d / h->mol_mass[j] * kNavo;
    }
    mdens[r] = md * mm; scat_pol[r] = sp;
  }

  geom.assign(vertical ? n_geom_all : 1, 0.0);                    // (transit: device-built, k_slant_geometry)
  {
    std::vector<double> sx(nr + 1);
    if (vertical) {
      double *gw = &geom[0], *gh0 = gw + (size_t)(nr + 1) * gstride;
      double *pw = gh0 + (nr + 1) + mw_doubles + (nr + 1);    // pair weights by starting layer (vertical rays)
      // only the two-point ray (start layer nr-2) integrates with tabulated weights (eclipse.c:68-80);
      // all others run on the pair weights below
      const int rs = nr - 2;
      const double r3[3] = {a->radius[rs], (a->radius[rs] + a->radius[rs+1]) / 2.0, a->radius[rs+1]};
      sx[0] = 0.0;
      for (int i = 1; i < 3; i++) sx[i] = sx[i-1] + (r3[i] - r3[i-1]);
      simpson_weights(sx.data(), 3, gw + (size_t)rs * gstride, gh0 + rs);
      for (int k = 0; k + 2 < nr; k++) { double h0; simpson_weights(a->radius + k, 3, pw + 4 * (size_t)k, &h0); }
      // what the chain of bottom-point parabolas and Simpson sums needs per layer (VertLayer, trx_kernels.hip.h):
      // the same IEEE operations the kernels used to repeat in every block
      double *lay = pw + 6 * (size_t)nr;
      for (int rs = 0; rs < nr; rs++) {
        double *L = lay + (size_t)kVertLay * rs;
        if (rs + 1 < nr) {
          const double step = a->radius[rs + 1] - a->radius[rs];
          L[0] = step; L[1] = a->radius[rs] / step; L[2] = 2.0 * step * step;
          L[8] = 1.0 / step; L[9] = 1.0 / L[2]; L[10] = L[1] + 1.5;
        }
        for (int q = 0; q < 4; q++) L[3 + q] = pw[4 * (size_t)rs + q];
        L[7] = a->radius[rs]; L[11] = a->radius[rs] * a->radius[rs];
      }
    } else {
      // transit geometry: built on the device (k_slant_geometry), nothing to prepare or ship here
    }
  }
  ipv.resize((size_t)nr);
  for (int i = 0; i < nr; i++) ipv[i] = a->radius[nr - 1 - i];

}

// what a run hands back: the handle's statistics and depth hint, the debug copies, the status raised on the device
struct RunOutcome {
  int flags[8], status[4];
  const std::vector<unsigned long long> *counters; const std::vector<uint8_t> *layer_walked;
  Spans *spans; hipEvent_t ev_a, ev_b;                    // profiled runs (null: not profiled)
  int nchunks; double ms_cia;
  std::chrono::steady_clock::time_point t_host0, t_host_prep, t_host_queued; const std::string *laps;
};

static int run_finish(trx_handle *h, const trx_atm *a, const trx_opts *o, trx_debug *dbg, const RunOutcome &Q,
                      const std::function<void(TauArgs &)> &model_args)
{
  const int nr = a->nlayer; const int64_t nsh = h->nsh;
  hipStream_t st = h->stream;
  int rc;
  trx_stats &S = h->stats;
  S.layers_swept = Q.flags[2];
  h->hint_layers = Q.flags[4];
  S.neval = S.nskip = S.sum_bins = S.sum_bins_walk = S.walk_layers = 0;
  S.walk_form_bins[0] = S.walk_form_bins[1] = S.walk_form_bins[2] = 0;
  for (int r = 0; r < nr; r++) {
    S.sum_bins += (int64_t)(*Q.counters)[3*r]; S.neval += (int64_t)(*Q.counters)[3*r+1]; S.nskip += (int64_t)(*Q.counters)[3*r+2];
    if (const int fw = (*Q.layer_walked)[(size_t)r]) { S.walk_layers++; S.sum_bins_walk += (int64_t)(*Q.counters)[3*r]; S.walk_form_bins[fw - 1] += (int64_t)(*Q.counters)[3*r]; }
  }
  float ms = 0; if (Q.spans) (void)hipEventElapsedTime(&ms, Q.ev_a, Q.ev_b); S.ms_run_total = ms;
  S.ms_cia = Q.ms_cia;
  S.ms_host_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - Q.t_host0).count();
  if (log_sink().fn && log_sink().max_level >= TRX_LOG_DEBUG) {
    char b[160];
    std::snprintf(b, sizeof b, "run: host %.0f us preparing inputs, %.0f us queueing, %.0f us waiting for the device",
                  1e3 * std::chrono::duration<double, std::milli>(Q.t_host_prep - Q.t_host0).count(),
                  1e3 * std::chrono::duration<double, std::milli>(Q.t_host_queued - Q.t_host_prep).count(),
                  1e3 * (S.ms_host_total - std::chrono::duration<double, std::milli>(Q.t_host_queued - Q.t_host0).count()));
    log_msg(TRX_LOG_DEBUG, std::string(b) + "; queueing by phase (us):" + *Q.laps);
  }
  S.ms_k_sweep = S.ms_k_walk = S.ms_k_accum = S.ms_tau = S.ms_sweep = 0; S.sweep_launches = 0;
  S.ms_k_walk_form[0] = S.ms_k_walk_form[1] = S.ms_k_walk_form[2] = 0; S.ms_walk_span = 0;
  if (Q.spans) {
    // every launch counts (also the ~4 us gated ones after all rays stopped), so that
    // sum / launches is the average a kernel trace reports
    double t[Spans::kKinds] = {0, 0, 0, 0, 0, 0};
    Q.spans->sum(t);
    S.ms_k_walk_form[0] = t[Spans::kWalk]; S.ms_k_walk_form[1] = t[Spans::kWalkLanes]; S.ms_k_walk_form[2] = t[Spans::kWalkPacked];
    S.ms_k_sweep = t[Spans::kSweep]; S.ms_k_walk = t[Spans::kWalk] + t[Spans::kWalkLanes] + t[Spans::kWalkPacked]; S.ms_k_accum = t[Spans::kAccum]; S.ms_tau = t[Spans::kTau];
    S.sweep_launches = Q.nchunks;
    S.ms_sweep = S.ms_k_sweep + S.ms_k_walk + S.ms_k_accum;
    S.ms_walk_span = Q.spans->walk_span();
  }

  if (dbg) {
    if (dbg->e)    HIPCHK(h, hipMemcpy(dbg->e, h->d_e.p, sizeof(double) * nr * nsh, hipMemcpyDeviceToHost));
    if (dbg->e_cs) HIPCHK(h, hipMemcpy(dbg->e_cs, h->d_ecs.p, sizeof(double) * nr * nsh, hipMemcpyDeviceToHost));
    if (dbg->tau) {
      std::vector<double> t((size_t)nr * nsh);
      HIPCHK(h, hipMemcpy(t.data(), h->d_tau.p, sizeof(double) * nr * nsh, hipMemcpyDeviceToHost));
      for (int64_t w = 0; w < nsh; w++) for (int i = 0; i < nr; i++) dbg->tau[(size_t)w * nr + i] = t[(size_t)i * nsh + w];
    }
    if (dbg->last) {
      std::vector<int> l(nsh);
      HIPCHK(h, hipMemcpy(l.data(), h->d_last.p, sizeof(int) * nsh, hipMemcpyDeviceToHost));
      for (int64_t w = 0; w < nsh; w++) dbg->last[w] = l[w];
    }
    if (dbg->intens && o->solution == TRX_SOL_ECLIPSE)
      HIPCHK(h, hipMemcpy(dbg->intens, h->d_intens.p, sizeof(double) * o->nangles * nsh, hipMemcpyDeviceToHost));
    if (dbg->computed) for (int r = 0; r < nr; r++) dbg->computed[r] = (r >= nr - S.layers_swept) ? 1 : 0;
    if (dbg->er) HIPCHK(h, hipMemcpy(dbg->er, h->d_er.p, sizeof(double) * nr * nsh, hipMemcpyDeviceToHost));
    if (dbg->e_scat || dbg->e_cloud) {
      DevBuf d_x;
      if ((rc = ensure(h, d_x, sizeof(double) * 2 * (size_t)nr * nsh))) return rc;
      TauArgs T{};
      T.nr = nr; T.nsh = nsh; T.lo = h->lo; T.wn_i = h->wn_i; T.wn_d = h->wn_d; T.wn_fct = o->wn_fct;
      model_args(T);
      double *xs = d_x.as<double>(), *xc = xs + (size_t)nr * nsh;
      hipLaunchKernelGGL(k_extras_dump, dim3((unsigned)((nsh + 255) / 256), (unsigned)nr), dim3(256), 0, st, T, xs, xc);
      HIPCHK(h, hipStreamSynchronize(st));
      if (dbg->e_scat)  HIPCHK(h, hipMemcpy(dbg->e_scat, xs, sizeof(double) * nr * nsh, hipMemcpyDeviceToHost));
      if (dbg->e_cloud) HIPCHK(h, hipMemcpy(dbg->e_cloud, xc, sizeof(double) * nr * nsh, hipMemcpyDeviceToHost));
    }
  }
  if (Q.status[0] == 1) return fail(h, TRX_E_NOTREACHED, "optical depth never reached toomuch (modlevel -1)");
  if (Q.status[0] == 2) return fail(h, TRX_E_ARG, "fewer than three points for the radial integration");
  if (Q.status[0] == 3) return fail(h, TRX_E_RANGE, "closest approach of a ray lies below the bottom layer (slantpath.c:39-44)");
  return TRX_OK;
}

static int run_once(trx_handle *h, const trx_atm *a, const trx_opts *o, double *spectrum, void *d_spectrum, trx_debug *dbg)
{
  if (!h || !a || !o) return TRX_E_ARG;
  const auto t_host0 = std::chrono::steady_clock::now();
  auto t_host_queued = t_host0;
  const int nr = a->nlayer, nmol = h->nmol;
  const int64_t nsh = h->nsh;
  { const int rcc = run_check(h, a, o); if (rcc) return rcc; }
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t st = h->stream;
  const bool eager = o->eager != 0, prof = o->profile != 0, count = o->profile >= 2;
  // A handle remembers how deep the previous spectrum went (hint_layers) and plans its steps to
  // end exactly there; the run returns at that depth and goes on only if rays are still open.
  const bool stop_at_hint_ok = !h->has_grid && !eager && h->hint_layers > 0 && h->hint_layers <= nr;
  int rc;

  hipStream_t st_sweep = st;
  const auto t_host_first = std::chrono::steady_clock::now();
  const bool lap_on = log_sink().fn && log_sink().max_level >= TRX_LOG_DEBUG;
  std::string laps; auto t_lap = t_host_first;
  auto lap = [&](const char *what) {
    if (!lap_on) return;
    const auto n = std::chrono::steady_clock::now();
    char b[64]; std::snprintf(b, sizeof b, " %s %.0f", what, 1e3 * std::chrono::duration<double, std::milli>(n - t_lap).count());
    laps += b; t_lap = n;
  };
  // ---- ray geometry: Simpson weights per start layer (eclipse.c:82-96, slantpath.c:76-95)
  // (eclipse geometry uses tabulated weights for its one three-point ray only: rows of one pair,
  // no modulation table -- 9 KB instead of 330 KB to build, copy and ship per run at 100 layers)
  const bool vertical = o->solution == TRX_SOL_ECLIPSE;
  const int gstride = vertical ? 4 : 4 * (nr / 2 + 1);
  const size_t mw_doubles = vertical ? 0 : (size_t)(nr + 1) * gstride;
  const size_t n_geom_all = (size_t)(nr + 1) * gstride + mw_doubles + 2 * (size_t)(nr + 1) + 4 * (size_t)nr + 2 * (size_t)nr +
                            (vertical ? (size_t)kVertLay * nr : 0);       // (vertical rays: the chain's per-layer constants behind the rest)
  // ---- per-run inputs: one pinned block (its layout is the run's shape alone)
  // [layer scalars f64 | ray geometry | impact parameters | CIA density products | layer scalars i32]
  const size_t nli = (size_t)nr * std::max(h->niso, 1);
  const size_t n_f64 = 7 * nli + 8 * (size_t)nr, n_geom = vertical ? n_geom_all : 1, n_ip = (size_t)nr, n_cd = (size_t)nr * h->cia.size(), n_i32 = 4 * nli;
  const size_t off_geom = n_f64, off_ip = off_geom + n_geom, off_cd = off_ip + n_ip, off_i32 = off_cd + n_cd;    // in doubles
  const size_t in_bytes = (8 * off_i32 + 4 * n_i32 + 8 + 15) & ~(size_t)15;
  if (h->h_in_bytes < in_bytes) {
    if (h->h_in) (void)hipHostFree(h->h_in);
    h->h_in = nullptr; h->h_in_bytes = 0; h->h_in_dev = nullptr;
    HIPCHK(h, hipHostMalloc(&h->h_in, in_bytes, hipHostMallocDefault));
    h->h_in_bytes = in_bytes;
    void *dp = nullptr;
    HIPCHK(h, hipHostGetDevicePointer(&dp, h->h_in, 0));
    h->h_in_dev = (const double *)dp;
  }
  if ((rc = ensure(h, h->d_in, in_bytes)) || (rc = ensure_small(h, nr)) ||
      (rc = ensure(h, h->d_last, sizeof(int) * nsh)) || (rc = ensure(h, h->d_acc, sizeof(double) * 2 * nsh)) ||
      (rc = ensure(h, h->d_sticky, sizeof(int) * nli)))
    return rc;
  // the layer maxima of consecutive runs alternate between two arrays: the run's start-up pass
  // (which rides along with k_layer_max) zeroes the NEXT run's
  {
    const size_t had = h->d_kmax.bytes;
    if ((rc = ensure(h, h->d_kmax, sizeof(double) * 2 * (size_t)nr))) return rc;
    if (h->d_kmax.bytes != had || !h->kmax_clean || h->kmax_nr != nr) {
      HIPCHK(h, hipMemsetAsync(h->d_kmax.p, 0, sizeof(double) * 2 * (size_t)nr, st_sweep));
      h->kmax_parity = 0; h->kmax_nr = nr;
    }
    h->kmax_clean = false;               // until this run has got through (an error return leaves the halves in doubt)
  }
  double *kmax_run = h->d_kmax.as<double>() + (size_t)h->kmax_parity * nr;
  RunInit R{};
  R.last = h->d_last.as<int>(); R.nsh = nsh; R.acc = h->d_acc.as<double>();
  R.counters = h->d_counters.as<unsigned long long>(); R.ncounters = 3 * nr;
  R.kmax = h->d_kmax.as<double>() + (size_t)(h->kmax_parity ^ 1) * nr; R.nkmax = nr;
  R.status = h->d_status.as<int>(); R.flags = h->d_flags.as<int>(); R.rays = (int)std::min<int64_t>(nsh, 0x7fffffff);
  h->kmax_parity ^= 1;
  // an error return below must not leave work of this run in flight (the next run would overwrite its inputs underneath it)
  struct Drain {
    trx_handle *h; bool armed = true;
    ~Drain() { if (armed) { (void)hipStreamSynchronize(h->stream4); (void)hipStreamSynchronize(h->stream2); (void)hipStreamSynchronize(h->stream); } }
  } drain{h};
  // ---- the device's first kernel, ahead of the host's prologue.  The strongest line of every layer (k_layer_max) needs
  // -c/T and SIGCTE*ratio/(m*Z) only: they go into the pinned block first, the kernel reads them THERE (a few hundred
  // doubles over the host link, once per block) and runs while the host computes widths, table indices, frames and ray
  // geometry -- 12 us that used to lie in front of the device's first instruction.  The rest of the block reaches device
  // memory by extra blocks of the next kernel (k_sticky_index, which reads its own few inputs from the pinned block too):
  // no copy engine, no wait of a kernel for a copy's completion signal.  (Opacity-grid runs and runs without lines have
  // no such kernels: the block is copied as before.)
  // (the start-up pass rides along with k_layer_max only where it is small next to it: a few
  // thousand threads striding over 10^7 rays took 20 ms at configs[4])
  const bool ride_along = nsh <= 65536;
  const bool early_front = !h->has_grid && h->ngroups > 0 && h->h_in_dev != nullptr;
  bool init_done = false;
  if (early_front) {
    double *hin = (double *)h->h_in;
    for (int r = 0; r < nr; r++) {
      if (!(a->temp[r] > 0)) return fail(h, TRX_E_ARG, "non-positive layer temperature");
      hin[r] = layer_negct(a->temp[r]);
      for (int i = 0; i < h->niso; i++) hin[(size_t)nr + (size_t)r * h->niso + i] = layer_strength(h, i, a->zpart[(size_t)i * nr + r]);
    }
    if (!ride_along) {
      hipLaunchKernelGGL(k_run_init, dim3((unsigned)std::min<long long>((std::max<long long>(nsh, 3LL * nr) + 255) / 256, 65536)), dim3(256), 0, st_sweep, R);
      init_done = true;
    }
    LayerDev Yp{}; Yp.negc_over_t = h->h_in_dev; Yp.strength_f = h->h_in_dev + nr;
    bool rode = false;
    if ((rc = launch_layer_max(h, Yp, nr, a->temp, 1, nullptr, st_sweep, kmax_run, ride_along ? &R : nullptr, &rode))) return rc;
    init_done = init_done || rode;
    lap("max");
  }

  // ---- layer prologue (extinction.c:364-395) --------------------------------
  LayerHost LH(h->run_f64, h->run_i32);
  if ((rc = prep_layers(h, nr, a->temp, a->density, a->zpart, 8 * (size_t)nr, LH))) return rc;
  lap("layers");
  h->walk_temp_ok = true;
  for (int r = 0; r < nr; r++) if (a->temp[r] < kWalkMinTemp) h->walk_temp_ok = false;
  std::vector<double> &f64 = LH.f64;
  const int32_t *psmax = LH.psmax;
  // Layers per step.  The walk (narrow profiles) takes up to 64 layers, one per lane; its cost
  // hardly depends on how many lanes are busy, so its steps are as large as the plan allows.
  // The two-kernel form keeps a strength buffer per layer in flight: at most kMaxChunk, and
  // 8 where the profiles are wide (a tile only learns between steps that its rays stopped).
  // Optical depths are integrated in sub-steps of at most tau_cap layers.
  const int tau_cap = o->solution == TRX_SOL_TRANSIT ? kTauH : kMaxChunk;
  const int user_chunk = o->layer_chunk > 0 ? std::max(3, o->layer_chunk) : 0;
  std::vector<double> &geom = h->run_geom, &ipv = h->run_ipv;
  run_host_inputs(h, a, LH, vertical, gstride, mw_doubles, n_geom_all, geom, ipv);
  lap("rays");
  if (f64.size() != n_f64 || geom.size() != n_geom || ipv.size() != n_ip || LH.i32.size() != n_i32)
    return fail(h, TRX_E_HIP, "internal: the input block's layout");

  double ms_cia = 0;

  // ---- workspaces -------------------------------------------------------------
  // Two streams: the line sweep of step c+1 (saturates the machine) runs on stream4 while the
  // optical depth of step c (a latency chain on a few waves) is integrated on the main stream.
  const bool pipelined = !h->has_grid;
  bool any_wide = false;                      // some layer needs the two-kernel form
  for (int r = 0; r < nr && !any_wide; r++) any_wide = walk_frame_bins(h, psmax, r) == 0;
  const size_t gr_b = (size_t)std::max<int64_t>(h->ngroups, 1);
  const int sg_layers = any_wide ? (user_chunk ? std::min(user_chunk, kMaxChunk) : kMaxChunk) : 1;
  if ((rc = ensure(h, h->d_SG, sizeof(double) * gr_b * sg_layers)) || (rc = ensure(h, h->d_idop8, gr_b * sg_layers)) ||
      (rc = ensure(h, h->d_e, sizeof(double) * nr * nsh)) || (rc = ensure(h, h->d_er, sizeof(double) * nr * nsh)) ||
      (rc = ensure(h, h->d_tau, sizeof(double) * nr * nsh)) ||
      (rc = ensure(h, h->d_intens, sizeof(double) * kMaxAngles * nsh)) || (rc = ensure(h, h->d_spec, sizeof(double) * nsh)))
    return rc;
  if (pipelined)
    while ((int)h->ev_ac.size() < nr + 1) {
      hipEvent_t e1, e2;
      if (hipEventCreateWithFlags(&e1, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&e2, hipEventDisableTiming) != hipSuccess) return fail(h, TRX_E_HIP, "event");
      h->ev_ac.push_back(e1); h->ev_cb.push_back(e2);
    }
  if (count && any_wide && (rc = ensure(h, h->d_part3, 24 * (size_t)kMaxChunk * ((((size_t)((nsh + kTileBins - 1) / kTileBins) + 3) / 4) + kXcds * kAccumXcdGroup))))
    return rc;
  if ((rc = ensure(h, h->d_ecs, sizeof(double) * (size_t)nr * nsh))) return rc;
  {
    double *hin = (double *)h->h_in;
    // (-c/T and the strength factors are in place already where the layer maxima were launched from them -- the same
    // values: they are not written a second time under a kernel that may be reading them)
    const size_t skip = early_front ? (size_t)nr + nli : 0;
    std::memcpy(hin + skip, f64.data() + skip, 8 * (n_f64 - skip));
    std::memcpy(hin + off_geom, geom.data(), 8 * n_geom);
    std::memcpy(hin + off_ip, ipv.data(), 8 * n_ip);
    cia_densities(h, a, hin + off_cd);
    std::memcpy(hin + off_i32, LH.i32.data(), 4 * n_i32);
  }
  lap("block");
  // the whole front end of a run goes to the stream the line sweep runs on (the main stream
  // joins it at the first optical depth): no cross-stream hop before the first sweep kernel
  // Streams.  The front end (inputs, layer maxima), the walks and everything that follows the
  // LAST walk of the plan -- its combine, optical depth, the spectrum, the copies back -- sit on
  // ONE queue: that chain is the critical path, and a hop between queues costs it ~30 us of
  // signalling.  The combines and optical depths of the earlier steps go to a second queue, where
  // they overlap the next step's walk.
  hipStream_t st_early = pipelined ? h->stream4 : st;
  bool early_dirty = false;                    // work queued on st_early that st has not waited for
  auto join_early = [&]() -> int {
    if (!early_dirty) return TRX_OK;
    if (hipEventRecord(h->ev_join, st_early) != hipSuccess || hipStreamWaitEvent(st, h->ev_join, 0) != hipSuccess) return fail(h, TRX_E_HIP, "event");
    early_dirty = false;
    return TRX_OK;
  };
  const auto t_host_prep = std::chrono::steady_clock::now();
  lap("prep");
  if (!early_front) { HIPCHK(h, hipMemcpyAsync(h->d_in.p, h->h_in, in_bytes, hipMemcpyHostToDevice, st_sweep)); lap("h2d"); }
  // With lines, every element of e the path reads is written first (the accumulation kernels
  // store every bin of a swept layer) and zeros only matter in the dumps.  Without any
  // in-range line (empty list, all lines outside the band, a CIA-only run) no kernel writes
  // e, but the optical-depth kernels still read it: it must be zero then.
  if (dbg || eager || (h->ngroups == 0 && !h->has_grid))
    HIPCHK(h, hipMemsetAsync(h->d_e.p, 0, sizeof(double) * nr * nsh, st_sweep));
  if (dbg || eager)
    HIPCHK(h, hipMemsetAsync(h->d_tau.p, 0, sizeof(double) * nr * nsh, st_sweep));

  const double *df = h->d_in.as<double>();
  LayerDev Y{}; const double *d_wcut; const int32_t *d_npre;
  layer_dev(df, (const int32_t *)(df + off_i32), LH, nr, Y, d_wcut, d_npre);
  const double *d_press = df + LH.extra_off, *d_tempk = d_press + nr, *d_mdens = d_tempk + nr, *d_nH = d_mdens + nr,
               *d_scatpol = d_nH + nr, *d_rad = d_scatpol + nr;
  // (ray geometry: part of the input block in eclipse geometry, a device-built buffer of its own in transit geometry)
  if (!vertical && (rc = ensure(h, h->d_geom, sizeof(double) * n_geom_all))) return rc;
  const double *d_gw = vertical ? df + off_geom : h->d_geom.as<double>(), *d_gh0 = d_gw + (size_t)(nr + 1) * gstride,
               *d_mw = d_gh0 + (nr + 1), *d_mh0 = d_mw + mw_doubles, *d_pw = d_mh0 + (nr + 1);
  const double *d_ipv = df + off_ip, *d_ciadens = df + off_cd;

  // ---- opacity grid: temperature bracket and weights per layer (extinction.c:549-574) ----
  std::vector<double> og_layer; std::vector<int> og_itemp;
  if (h->has_grid) {
    if (h->og_nlayer != nr) return fail(h, TRX_E_ARG, "opacity grid has a different number of layers");
    const int nt = (int)h->og_ntemp, nm = (int)h->og_nmol;
    og_layer.assign((size_t)(3 + nm) * nr, 0.0); og_itemp.assign(nr, 0);
    for (int r = 0; r < nr; r++) {
      const double temp = a->temp[r];
      if (temp < h->og_temp[0] || !(temp < h->og_temp[nt - 1])) return fail(h, TRX_E_RANGE, "layer temperature outside the opacity grid");
      int it = nearest_index(h->og_temp.data(), temp, 0, nt);
      if (temp < h->og_temp[it]) it--;
      og_itemp[r] = it;
      og_layer[r] = h->og_temp[it + 1] - temp; og_layer[nr + r] = temp - h->og_temp[it];
      og_layer[2 * (size_t)nr + r] = h->og_temp[it + 1] - h->og_temp[it];
      for (int m = 0; m < nm; m++) og_layer[(size_t)(3 + m) * nr + r] = a->density[(size_t)h->og_molidx[m] * nr + r];
    }
    if ((rc = upload(h, h->d_og_layer, og_layer)) || (rc = upload(h, h->d_og_itemp, og_itemp))) return rc;
  }

  // ---- the sticky Doppler index of every layer (with the block's copy into device memory riding along), or -- no early
  // front end -- both front kernels behind the copy command.  The event marks the place of the main queue behind which
  // the inputs are in device memory: the CIA queue waits for it when its kernels are queued (behind the first walk's
  // launch, not on the host's way to the device's first kernel), the side queue in front of its first work of the run.
  if (early_front) {
    // (what k_sticky_index itself reads of the block, it reads from the pinned copy: nothing of a grid can wait for the
    // copy blocks of the same grid)
    const double *pf = h->h_in_dev;
    LayerDev Yp{}; const double *p_wcut; const int32_t *p_npre;
    layer_dev(pf, (const int32_t *)(pf + off_i32), LH, nr, Yp, p_wcut, p_npre);
    if ((rc = launch_sticky(h, Yp, p_npre, nr, 1, nullptr, o->ethresh, st_sweep, kmax_run, h->h_in_dev, h->d_in.p, in_bytes))) return rc;
  } else {
    if (!ride_along) {
      hipLaunchKernelGGL(k_run_init, dim3((unsigned)std::min<long long>((std::max<long long>(nsh, 3LL * nr) + 255) / 256, 65536)), dim3(256), 0, st_sweep, R);
      init_done = true;
    }
    bool rode = false;
    if (!h->has_grid &&
        (rc = layer_maxima_and_sticky(h, Y, d_npre, nr, a->temp, 1, nullptr, o->ethresh, st_sweep, kmax_run, ride_along ? &R : nullptr, &rode))) return rc;
    if (!init_done && !rode)               // no line kernel to ride along with (opacity-grid mode, no in-range line)
      hipLaunchKernelGGL(k_run_init, dim3((unsigned)std::min<long long>((std::max<long long>(nsh, 3LL * nr) + 255) / 256, 65536)), dim3(256), 0, st_sweep, R);
  }
  HIPCHK(h, hipEventRecord(h->ev_inputs, st_sweep));
  lap("kmax");
  // CIA extinction (device), on a second stream: only the first optical-depth kernel needs
  // e_cs, so the (latency-bound) spline kernels overlap the first sweep step.  Queued right
  // after that step's kernels, which are what the GPU is waiting for.
  // scattering / cloud models: the parameters of tau.c:193-214, extinction.c:587-693, and the per-ray
  // wavenumber factors the optical-depth kernels multiply the layer parts with (k_extras_factors)
  const bool extras_on = o->scat_flag != 0 || o->cloud_flag != 0;
  if (extras_on && (rc = ensure(h, h->d_xf, sizeof(double) * (2 * (size_t)nsh + 8)))) return rc;
  auto model_args = [&](TauArgs &T) {
    T.scat_flag = o->scat_flag; T.cloud_flag = o->cloud_flag; T.nmol = nmol;
    T.scat_pref = std::pow(10.0, o->scat_logext) * kE0H2;
    T.press = d_press; T.temp = d_tempk; T.scat_pol = d_scatpol;
    T.cloud_top = o->cloud_top; T.cloud_bot = o->cloud_bot; T.cloud_ext = o->cloud_ext; T.cloud_gamma = o->cloud_gamma;
    T.cloud_Q = o->cloud_Q; T.cloud_r = o->cloud_r; T.cloud_sig = o->cloud_sig; T.cloud_refwn = o->cloud_refwn;
    T.mdens = d_mdens; T.nH = d_nH;
    if (extras_on) { T.xf_scat = h->d_xf.as<double>(); T.xf_cloud = T.xf_scat + nsh; T.xf_const = T.xf_cloud + nsh; }
  };
  auto queue_cia = [&]() -> int {
    const auto t0 = std::chrono::steady_clock::now();
    if (hipStreamWaitEvent(h->stream2, h->ev_inputs, 0) != hipSuccess) return fail(h, TRX_E_HIP, "event");
    if (extras_on) {        // (ahead of the first optical depth like everything on this queue)
      TauArgs X{};
      X.nr = nr; X.nsh = nsh; X.lo = h->lo; X.wn_i = h->wn_i; X.wn_d = h->wn_d; X.wn_fct = o->wn_fct;
      model_args(X);
      hipLaunchKernelGGL(k_extras_factors, dim3((unsigned)((nsh + 255) / 256)), dim3(256), 0, h->stream2, X,
                         h->d_xf.as<double>(), h->d_xf.as<double>() + nsh, h->d_xf.as<double>() + 2 * nsh);
    }
    if (!vertical) {        // the slant rays' geometry, ahead of the CIA kernels: both are waited for by the first optical depth
      SlantGeomArgs G{};
      G.rad = d_rad; G.nr = nr; G.fct = a->rad_fct; G.gstride = gstride;
      G.gw = const_cast<double *>(d_gw); G.gh0 = const_cast<double *>(d_gh0); G.mw = const_cast<double *>(d_mw); G.mh0 = const_cast<double *>(d_mh0);
      G.hrs = const_cast<double *>(d_pw) + 4 * (size_t)nr; G.hr0 = G.hrs + nr;
      hipLaunchKernelGGL(k_slant_geometry, dim3((unsigned)(nr + nr - 2)), dim3(64), 0, h->stream2, G);
    }
    const int rcc = cia_device(h, a, o, d_tempk, d_ciadens, h->stream2);
    if (rcc) return rcc;
    if (hipEventRecord(h->ev_cia, h->stream2) != hipSuccess) return fail(h, TRX_E_HIP, "event");
    ms_cia = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return TRX_OK;
  };

  // ---- events -----------------------------------------------------------------
  Spans spans;
  // (the run's two timing events live with the handle: creating and destroying a pair per run was
  // ~10 us of host time, which a small shard waits for)
  struct { hipEvent_t a, b; } ev{h->ev_run_a, h->ev_run_b};
  if (!ev.a) {
    HIPCHK(h, hipEventCreate(&h->ev_run_a)); HIPCHK(h, hipEventCreate(&h->ev_run_b));
    ev.a = h->ev_run_a; ev.b = h->ev_run_b;
  }
  // (device time of the run, trx_stats.ms_run_total: profiled runs only -- an event record is a
  // packet of its own between the spectrum kernel and the copy back)
  if (prof) HIPCHK(h, hipEventRecord(ev.a, st));

  h->stats.walk_steps = 0; h->stats.walk_records = 0; h->stats.walk_record_lanes = 0;
  for (int k = 0; k < 3; k++) h->stats.walk_form_steps[k] = h->stats.walk_form_layers[k] = h->stats.walk_form_record_lanes[k] = 0;
  std::vector<uint8_t> layer_walked((size_t)nr, 0);         // layers swept by walk steps: 1 + the form that took them (stats)
  if (!h->has_grid && log_sink().fn && log_sink().max_level >= TRX_LOG_DEBUG) {
    std::string ln = "run: walk frame (bins) per layer, top first; 0 = two-kernel form:";
    for (int r = nr - 1; r >= 0; r--) ln += " " + std::to_string(walk_frame_bins(h, psmax, r));
    log_msg(TRX_LOG_DEBUG, ln);
  }
  // ---- top-down sweep in steps of layers (tau.c:235-290) ----
  int nchunks = 0, nwalks = 0, r_top = nr - 1;
  bool stop_at_hint = stop_at_hint_ok, resumed = false;
  int flags_host[8] = {0, 0, 0, 0, 0, 0, 0, 0}, status_host[4] = {0, 0, 0, 0};
  std::vector<unsigned long long> counters(3 * (size_t)nr);
  double *d_out = d_spectrum ? (double *)d_spectrum : h->d_spec.as<double>();
  auto tau_args = [&](TauArgs &T, int r_top_, int nc_) {
    T.nr = nr; T.solution = o->solution; T.nsh = nsh; T.lo = h->lo;
    T.wn_i = h->wn_i; T.wn_d = h->wn_d; T.wn_fct = o->wn_fct; T.rad_fct = a->rad_fct; T.toomuch = o->toomuch;
    T.r_top = r_top_; T.nc = nc_; T.rad = d_rad; T.e = h->d_e.as<double>(); T.ecs = h->d_ecs.as<double>();
    T.er = h->d_er.as<double>(); T.tau = h->d_tau.as<double>(); T.last = h->d_last.as<int>();
    T.er_all = (dbg != nullptr || eager) ? 1 : 0;
    T.gw = d_gw; T.gstride = gstride; T.gh0 = d_gh0;
    model_args(T);
    T.flags = h->d_flags.as<int>(); T.eager = eager;
    T.pw = d_pw; T.acc = h->d_acc.as<double>(); T.lay = vertical ? d_pw + 6 * (size_t)nr : nullptr;
    T.hrs = d_pw + 4 * (size_t)nr; T.hr0 = T.hrs + nr; T.status = h->d_status.as<int>();
  };
  auto emis_args = [&](EmisArgs &E) {
    E.nr = nr; E.nang = o->nangles; E.nsh = nsh; E.lo = h->lo; E.wn_i = h->wn_i; E.wn_d = h->wn_d; E.wn_fct = o->wn_fct;
    E.tau = h->d_tau.as<double>(); E.last = h->d_last.as<int>(); E.temp = d_tempk;
    std::vector<double> grid(o->nangles + 1);                    // eclipse.c:262-269
    grid[0] = 0.0 * kDeg; grid[o->nangles] = 90.0 * kDeg;
    for (int i = 1; i < o->nangles; i++) grid[i] = (o->angles_deg[i-1] + o->angles_deg[i]) * kDeg / 2.0;
    for (int i = 0; i < o->nangles; i++) {
      E.cosang[i] = std::cos(o->angles_deg[i] * kDeg);
      E.area[i] = std::pow(std::sin(grid[i+1]), 2.0) - std::pow(std::sin(grid[i]), 2.0);
      E.rcos[i] = checked_reciprocal(h, E.cosang[i]);
    }
    E.intens = h->d_intens.as<double>(); E.flux = d_out; E.e2tab = h->d_e2tab.as<double>();
  };
  auto mod_args = [&](ModArgs &M) {
    M.nr = nr; M.modlevel = o->modlevel; M.transparent = o->transparent; M.nsh = nsh; M.toomuch = o->toomuch;
    M.ip_fct = a->rad_fct; M.srad = o->starrad_cm; M.tau = h->d_tau.as<double>(); M.last = h->d_last.as<int>();
    M.ip = d_ipv; M.gw = d_mw; M.gstride = gstride; M.gh0 = d_mh0; M.out = d_out; M.status = h->d_status.as<int>();
  };
  struct SideWork { bool active = false, first = false; int r_top = 0, nc = 0, swept = 0; hipStream_t st_tau = nullptr; PendingCombine pc; };
  SideWork pending;
  bool early_behind_inputs = false;            // the side queue has waited for this run's input copy
  auto run_side = [&](SideWork &S) -> int {
    // The side queue's first work of a run waits for the inputs explicitly.  (Every step kind of today queues it behind
    // an event recorded on the main queue after the copy -- a walk's, a sweep's -- but that is a rule nothing states.)
    if (S.st_tau != st && !early_behind_inputs) { HIPCHK(h, hipStreamWaitEvent(S.st_tau, h->ev_inputs, 0)); early_behind_inputs = true; }
    int rc = launch_combine(h, S.pc, prof ? &spans : nullptr);
    if (rc) return rc;
    if (S.first) {
      if ((rc = queue_cia())) return rc;
      lap("cia");
      HIPCHK(h, hipStreamWaitEvent(S.st_tau, h->ev_cia, 0));
    }
    if (S.st_tau == st) { if ((rc = join_early())) return rc; }       // the optical depths of the earlier steps
    else early_dirty = true;
    if (!h->saved.empty())        // layers restored from an earlier run (trx_restore_extinction): their rows as they were saved
      for (int c = 0; c < S.nc; c++) {
        const int r = S.r_top - c;
        if (h->saved[(size_t)r])
          HIPCHK(h, hipMemcpyAsync(h->d_e.as<double>() + (size_t)r * nsh, h->d_e_saved.as<double>() + (size_t)r * nsh, sizeof(double) * (size_t)nsh,
                                   hipMemcpyDeviceToDevice, S.st_tau));
      }
    if (prof && spans.begin(Spans::kTau, S.st_tau)) return fail(h, TRX_E_HIP, "event");
    for (int done = 0; done < S.nc; ) {          // optical depth in sub-steps of at most tau_cap layers
      int nt = std::min(tau_cap, S.nc - done);
      if (S.swept == 0 && done == 0) nt = std::min(S.nc, std::max(nt, 3));
      TauArgs T{};
      tau_args(T, S.r_top - done, nt);
      if (o->solution == TRX_SOL_ECLIPSE) {
        // small shards: one wave per block spreads the (latency-bound) chains over more CUs
        const bool small = nsh <= 64 * 1024, extras = o->scat_flag != 0 || o->cloud_flag != 0;
        const dim3 grid((unsigned)std::min<int64_t>((nsh + (small ? 63 : 255)) / (small ? 64 : 256), kTauMaxBlocks)), block(small ? 64 : 256);
        if (small && extras)       hipLaunchKernelGGL((k_optical_depth_vertical<true, true>), grid, block, 0, S.st_tau, T);
        else if (small)            hipLaunchKernelGGL((k_optical_depth_vertical<true, false>), grid, block, 0, S.st_tau, T);
        else if (extras)           hipLaunchKernelGGL((k_optical_depth_vertical<false, true>), grid, block, 0, S.st_tau, T);
        else                       hipLaunchKernelGGL((k_optical_depth_vertical<false, false>), grid, block, 0, S.st_tau, T);
      } else if (o->scat_flag != 0 || o->cloud_flag != 0)
        hipLaunchKernelGGL(k_optical_depth<true>, dim3((unsigned)std::min<int64_t>((nsh + kTauW - 1) / kTauW, kTauMaxBlocks)),
                           dim3(256), 0, S.st_tau, T);
      else
        hipLaunchKernelGGL(k_optical_depth<false>, dim3((unsigned)std::min<int64_t>((nsh + kTauW - 1) / kTauW, kTauMaxBlocks)),
                           dim3(256), 0, S.st_tau, T);
      done += nt;
    }
    if (prof && spans.end(S.st_tau)) return fail(h, TRX_E_HIP, "event");
    return TRX_OK;
  };
  // Step plan.  Layers still to go: down to the previous spectrum's depth when it is known
  // (retrieval loops re-run near-identical atmospheres), else to the bottom.  The step takes
  // the layers of ONE kind from r_top down -- walk or two-kernel form -- up to that kind's cap,
  // in equal parts when more than one step is needed.
  auto plan_step = [&](int r_top, bool stop_at_hint, int &nb, int &nc, bool &last_step) {
    const int swept = nr - 1 - r_top;
    int togo = r_top + 1;
    if (!eager && h->hint_layers > swept) togo = std::min(togo, h->hint_layers - swept);
    nb = 0;
    if (h->has_grid) nc = std::min(togo, user_chunk ? user_chunk : kMaxChunk);
    else {
      nb = walk_frame_bins(h, psmax, r_top);
      int run = 1;                                 // consecutive layers of the same kind below r_top
      while (run < togo && (walk_frame_bins(h, psmax, r_top - run) == 0) == (nb == 0)) run++;
      int cap = nb ? kWalkLayers : kMaxChunk;
      if (!nb && (2 * layer_psmax(h, psmax, std::max(0, r_top - run + 1))) / h->osamp + 1 >= 64) cap = 8;
      if (user_chunk) cap = std::min(cap, user_chunk);
      else if (!nb && !stop_at_hint_ok) cap = std::min(cap, 12);      // depth unknown, expensive layers: small steps
      const int steps = (run + cap - 1) / cap;
      nc = (run + steps - 1) / steps;
      if (nb && steps > 1 && !user_chunk) {
        // A walk step costs what its WIDEST layer's frame costs, whatever the number of layers
        // (<= 64, one per lane), and frames grow with depth.  So this step takes the layers it
        // cannot leave to the later steps, and then as many more as share their frame: the wide
        // frames further down are paid for by as few lanes as possible.
        const int must = run - (steps - 1) * cap;
        int f = 0;
        for (int c = 0; c < must; c++) f = std::max(f, walk_frame_bins(h, psmax, r_top - c));
        nc = must;
        while (nc < cap && nc < run && walk_frame_bins(h, psmax, r_top - nc) <= f) nc++;
      }
      if (nb) for (int c = 1; c < nc; c++) nb = std::max(nb, walk_frame_bins(h, psmax, r_top - c));
    }
    if (swept == 0) nc = std::max(nc, 3);          // the first step holds the 2- and 3-point rays (eclipse.c:65-80)
    nc = std::min(nc, r_top + 1);
    if (nb && swept == 0) for (int c = 1; c < nc; c++) {          // (a widened first step stays one kind)
      const int v = walk_frame_bins(h, psmax, r_top - c);
      if (v == 0) { nb = 0; break; }
      nb = std::max(nb, v);
    }
    if (!nb && !h->has_grid && nc > sg_layers) nc = sg_layers;
    // the plan's last step (bottom reached, or the depth the previous spectrum needed): its
    // combine and optical depth stay on the walk's queue
    last_step = r_top - nc < 0 || (stop_at_hint && nr - 1 - (r_top - nc) >= h->hint_layers);
  };
  // ---- the ray tail (trx_tail.hip.h): a hinted eclipse run whose plan is one or two walk steps from
  // the top ends in ONE kernel behind its walks -- no side queue, no combine / optical depth /
  // emission launches.  Decided from the plan, which a hinted run knows beforehand.
  bool tail_mode = false;
  TailArgs TA{};
  if (h->ray_tail && stop_at_hint_ok && !count && h->ngroups > 0 && h->saved.empty() &&      // (profile 1: the same plan with events around its kernels)
      nsh <= 65536 && h->nwn <= kEmisRowsAbove && nsh < 0x7fffffffLL / kTailRays) {
    int r = nr - 1, steps = 0; bool ok = true;
    for (; r >= 0 && ok; ) {
      int nb_, nc_; bool last_;
      plan_step(r, true, nb_, nc_, last_);
      if (!nb_ || ++steps > kTailSteps) ok = false;
      r -= nc_;
      if (nr - 1 - r >= h->hint_layers) break;
    }
    tail_mode = ok && steps >= 1;
  }
  // (flags into the pinned block the host reads; the spectrum into pinned memory too when the caller wants it on the host)
  const bool tail_direct = tail_mode && h->tail_direct, tail_spec = tail_direct && spectrum && !d_spectrum;
  // (any run that hands its spectrum to pageable host memory stages it in the handle's pinned buffer when it is small:
  // the copy command into pageable memory is staged by the runtime anyway, 25 us behind the copy of the flags at configs[2])
  const bool stage_spec = spectrum && !d_spectrum && nsh <= (1 << 20);
  if ((tail_spec || stage_spec) && h->h_spec_bytes < sizeof(double) * (size_t)nsh) {
    if (h->h_spec) (void)hipHostFree(h->h_spec);
    h->h_spec = nullptr; h->h_spec_bytes = 0;
    HIPCHK(h, hipHostMalloc(&h->h_spec, sizeof(double) * (size_t)nsh, hipHostMallocDefault));
    h->h_spec_bytes = sizeof(double) * (size_t)nsh;
    HIPCHK(h, hipHostGetDevicePointer(&h->h_spec_dev, h->h_spec, 0));
  }
  // Vertical rays: what the blocks of the tail add to the run's flags -- rays still open, deepest layer reached -- goes
  // into a pinned array, one entry per block, and the HOST adds it up behind the kernel.  The device-side sum was three
  // dependent device-scope atomics per block and, for the last block to arrive, four more round trips and a system-scope
  // fence: the kernel's last wave ended 5 us behind its last emission -- and every one of those fences writes back and
  // invalidates the L2 under the emission waves (8 us of emission with them, 3 without).  The run's status (slant
  // rays: what the reference exits on) goes into four slots behind the blocks' entries, one per code.
  const size_t tail_blocks = (size_t)((nsh + kTailRays - 1) / kTailRays);
  const bool tail_hostsum = tail_direct;
  if (tail_hostsum && h->h_tailblk_bytes < 8 * tail_blocks + 16) {
    if (h->h_tailblk) (void)hipHostFree(h->h_tailblk);
    h->h_tailblk = nullptr; h->h_tailblk_bytes = 0;
    HIPCHK(h, hipHostMalloc(&h->h_tailblk, 8 * tail_blocks + 16, hipHostMallocDefault));
    h->h_tailblk_bytes = 8 * tail_blocks + 16;
    HIPCHK(h, hipHostGetDevicePointer(&h->h_tailblk_dev, h->h_tailblk, 0));
  }
  int tail_nct = 0;
  const bool two_queues = h->two_queues && tail_direct && pipelined;      // (tail_direct: nothing behind the tail on the main queue)
  bool tail_on_side = false, cia_queued = false;
  for (;;) {
  {
  for (; r_top >= 0; ) {
    const int swept = nr - 1 - r_top;
    int nb = 0, nc = 0; bool last_step = false;
    plan_step(r_top, stop_at_hint, nb, nc, last_step);
    hipStream_t st_tau = (pipelined && !last_step) ? st_early : st;
    SideWork S;
    bool step_walked = false;
    S.first = nchunks == 0; S.r_top = r_top; S.nc = nc; S.swept = swept; S.st_tau = st_tau;
    if (h->has_grid) {
      if (prof && spans.begin(Spans::kSweep, st)) return fail(h, TRX_E_HIP, "event");
      GridArgs Gd{};
      Gd.o = h->d_og_o.as<double>(); Gd.nt = (int)h->og_ntemp; Gd.nm = (int)h->og_nmol; Gd.nr = nr;
      Gd.nwave = h->og_nwave; Gd.lo = h->lo; Gd.nsh = nsh; Gd.r_top = r_top; Gd.nc = nc;
      Gd.itemp = h->d_og_itemp.as<int>();
      Gd.w_lo = h->d_og_layer.as<double>(); Gd.w_hi = Gd.w_lo + nr; Gd.dg = Gd.w_hi + nr; Gd.dens = Gd.dg + nr;
      Gd.e = h->d_e.as<double>(); Gd.flags = h->d_flags.as<int>(); Gd.eager = eager;
      hipLaunchKernelGGL(k_grid_extinction, dim3((unsigned)((nsh + 255) / 256), (unsigned)nc), dim3(256), 0, st, Gd);
      if (prof && spans.end(st)) return fail(h, TRX_E_HIP, "event");
    } else {
      SweepMode M{};
      bool walked = false; int form = 0;
      M.eager = eager; M.prof = count; M.ethresh = o->ethresh;
      M.skip_done = (!eager && !(dbg && dbg->e)); M.nmx = 1; M.d_iso_mx = nullptr; M.permol = false;
      M.d_e = h->d_e.as<double>(); M.d_kmax = kmax_run; M.d_sticky = h->d_sticky.as<int>();
      M.st = st_sweep;
      bool all_saved = !h->saved.empty();
      for (int c = 0; c < nc && all_saved; c++) all_saved = h->saved[(size_t)(r_top - c)] != 0;
      if (h->ngroups > 0 && !all_saved) {
        if (nb && tail_mode && !resumed) {
          // (no combine of its own: its records wait for the tail, each step in its own buffer)
          // Two walks of one run do not depend on each other, and a walk alone leaves a third of the device idle for
          // the last third of its time: waves of one launch start together and end apart -- the oldest wave of a SIMD is
          // served first, a step's ~13 600 ranges are under two generations of resident waves, and the kernel behind it
          // on the queue cannot start before the last wave has ended (in-kernel clocks, round 5: 7168 waves in flight
          // for the first 60 us of k_line_walk<2>, then 5000, 3900, 3000, 2200, 1200, 370 at 5 us steps).  The plan's
          // second walk therefore goes to the side queue, behind the event that marks the inputs, and fills what the
          // first one leaves; the tail follows it THERE (same queue: no signal between them) and waits for the first
          // walk's event, long satisfied by then.  Demo: 0.269 -> 0.251 ms, the same bits.
          const bool side_walk = two_queues && nchunks == 1;
          if (side_walk) { HIPCHK(h, hipStreamWaitEvent(h->stream4, h->ev_inputs, 0)); M.st = h->stream4; tail_on_side = true; }
          rc = walk_chunk(h, Y, d_wcut, nb, r_top, nc, M, prof ? &spans : nullptr, nwalks, nullptr, nullptr, nullptr, nullptr, &S.pc, &form);
          if (!rc) {
            TailStep &TS = TA.S[TA.nsteps++];
            TS.P = S.pc.C.P; TS.part = S.pc.C.part; TS.nc = nc;
            TA.skip = S.pc.C.last;
            S.pc.valid = false;
          }
          nwalks++;
        }
        else if (nb) {
          rc = walk_chunk(h, Y, d_wcut, nb, r_top, nc, M, prof ? &spans : nullptr, nwalks, st_tau != st ? st_tau : nullptr, h->ev_ac[nchunks],
                          nwalks >= 2 ? h->ev_cb[(nwalks - 2) % h->ev_cb.size()] : nullptr, h->ev_cb[nwalks % h->ev_cb.size()], &S.pc, &form);
          nwalks++;
        }
        else    rc = sweep_chunk(h, Y, d_wcut, psmax, r_top, nc, sg_layers, M, prof ? &spans : nullptr);
        if (rc) return rc;
        walked = nb != 0;
        step_walked = walked;
        if (walked) for (int c = 0; c < nc; c++) layer_walked[(size_t)(r_top - c)] = (uint8_t)(1 + form);
      }
      if (st_tau != st && !walked) {     // the optical depth of this step follows its extinction
        HIPCHK(h, hipEventRecord(h->ev_ac[nchunks], st));
        HIPCHK(h, hipStreamWaitEvent(st_tau, h->ev_ac[nchunks], 0));
      }
    }
    lap("sweep");
    if (tail_mode && !resumed) {
      // the CIA kernels go to their queue behind the first walk (what the device is waiting for) -- ahead of a second
      // walk on the side queue too: next to TWO walks they take three times as long, and configs[3]'s two tables then
      // end after the walks.  With two queues the main one waits for them behind its walk and marks the place: one
      // event for the tail to wait for.
      if (!cia_queued) {
        if ((rc = queue_cia())) return rc;
        cia_queued = true;
        if (two_queues && !last_step) { HIPCHK(h, hipStreamWaitEvent(st, h->ev_cia, 0)); HIPCHK(h, hipEventRecord(h->ev_walk1, st)); }
        lap("cia");
      }
      r_top -= nc; nchunks++;
      if (nr - 1 - r_top >= h->hint_layers) break;
      continue;
    }
    // The rest of the step -- its combine, the CIA kernels ahead of the first optical depth, the
    // optical depth itself -- goes to the side queue for every step but the plan's last, and is
    // QUEUED only after the next step's walk: the walks then sit back to back on the main queue
    // however long the host takes over the rest (small shards are host-bound otherwise).
    if (pending.active) { if ((rc = run_side(pending))) return rc; pending.active = false; }
    S.active = true;
    // (walk steps only: with the two-kernel form's long steps the later queueing of an optical
    // depth measurably delays the stop information the next step's tile skipping reads --
    // configs[4] 0.223 -> 0.246 s)
    if (st_tau != st && step_walked) pending = S;
    else if ((rc = run_side(S))) return rc;
    lap("tau");
    r_top -= nc; nchunks++;
    // the previous spectrum stopped here: compute the spectrum now and look at the outcome
    // on the host (which this call waits for anyway) instead of queueing gated no-op steps
    if (stop_at_hint && nr - 1 - r_top >= h->hint_layers) break;
  }

  if (pending.active) { if ((rc = run_side(pending))) return rc; pending.active = false; }

  // ---- spectrum ---------------------------------------------------------------
  if ((rc = join_early())) return rc;
  if (resumed) HIPCHK(h, hipMemsetAsync(h->d_status.p, 0, 16, st));       // (a resumed run computes the spectrum a second time)
  hipStream_t tst = st;                          // the tail's queue
  if (tail_mode && !resumed) {
    if (tail_on_side) { tst = h->stream4; HIPCHK(h, hipStreamWaitEvent(tst, h->ev_walk1, 0)); }
    else {
      if (!cia_queued) { if ((rc = queue_cia())) return rc; cia_queued = true; }
      HIPCHK(h, hipStreamWaitEvent(st, h->ev_cia, 0));
    }
    TA.niso = h->niso; TA.gblock = h->d_gblock.as<int32_t>(); TA.e = h->d_e.as<double>();
    for (int b = 0; b < h->niso && b < 64; b++) if (h->h_gblock[b] != h->h_gblock[b + 1]) TA.blocks |= 1ull << b;
    int nct = 0;
    for (int k = 0; k < TA.nsteps; k++) nct += TA.S[k].nc;
    tau_args(TA.T, nr - 1, nct);
    if (vertical) emis_args(TA.E); else mod_args(TA.M);
    if (tail_direct) {              // spectrum and flags straight into pinned host memory: no copy commands behind the kernel
      if (tail_spec) (vertical ? TA.E.flux : TA.M.out) = (double *)h->h_spec_dev;
      TA.host_flags = (int *)h->h_small_dev;
      if (tail_hostsum) {
        TA.host_blocks = (int *)h->h_tailblk_dev;
        std::memset((char *)h->h_tailblk + 8 * tail_blocks, 0, 16);
        if (!vertical) TA.M.status_slots = TA.host_blocks + 2 * tail_blocks;
      }
      tail_nct = nct;
    }
    // Every address the tail reads or writes, checked on the host before the launch: a null or stale one
    // here is a wild access of a whole grid (round 3's one memory fault -- a work-in-progress tail storing
    // the spectrum through a pinned buffer that no branch had allocated yet -- was exactly this kind).
    {
      bool ok = TA.nsteps >= 1 && TA.nsteps <= kTailSteps && nct >= 3 && nct <= kTailLayers && TA.gblock && TA.e &&
                TA.T.ecs && TA.T.er && TA.T.tau && TA.T.last && TA.T.flags && TA.T.status && TA.T.rad &&
                (vertical ? (TA.T.acc && TA.T.lay && TA.T.gw && TA.E.flux && TA.E.intens && TA.E.temp && TA.E.e2tab)
                          : (TA.T.hrs && TA.T.hr0 && TA.T.gw && TA.M.out && TA.M.ip && TA.M.gw && TA.M.gh0 && TA.M.status)) &&
                (!extras_on || (TA.T.xf_scat && TA.T.xf_cloud)) && (!tail_direct || TA.host_flags) && (!tail_hostsum || TA.host_blocks);
      for (int k = 0; k < TA.nsteps && ok; k++)
        ok = TA.S[k].part && TA.S[k].nc >= 1 && TA.S[k].nc <= kWalkLayers && TA.S[k].P.blo && TA.S[k].P.bhi && TA.S[k].P.off && TA.S[k].P.wbase;
      if (!ok) return fail(h, TRX_E_HIP, "internal: incomplete arguments for the ray tail (not launched)");
    }
    const dim3 tgrid((unsigned)((nsh + kTailRays - 1) / kTailRays)), tblock(kTailThreads);
    if (prof && spans.begin(Spans::kTau, tst)) return fail(h, TRX_E_HIP, "event");
    if (!vertical) {
      if (extras_on) hipLaunchKernelGGL((k_ray_tail<0, true>), tgrid, tblock, 0, tst, TA);
      else           hipLaunchKernelGGL((k_ray_tail<0, false>), tgrid, tblock, 0, tst, TA);
    } else if (extras_on) {
      if (o->nangles <= 8) hipLaunchKernelGGL((k_ray_tail<8, true>), tgrid, tblock, 0, tst, TA);
      else                 hipLaunchKernelGGL((k_ray_tail<kMaxAngles, true>), tgrid, tblock, 0, tst, TA);
    } else {
      if (o->nangles <= 8) hipLaunchKernelGGL((k_ray_tail<8, false>), tgrid, tblock, 0, tst, TA);
      else                 hipLaunchKernelGGL((k_ray_tail<kMaxAngles, false>), tgrid, tblock, 0, tst, TA);
    }
    if (prof && spans.end(tst)) return fail(h, TRX_E_HIP, "event");
    if (lap_on) log_msg(TRX_LOG_DEBUG, "run: ray tail over " + std::to_string(TA.nsteps) + " walk steps, " + std::to_string(nct) + " layers");
  }
  else if (o->solution == TRX_SOL_ECLIPSE) {
    EmisArgs E{};
    emis_args(E);
    if (h->nwn > kEmisRowsAbove)       // (by the job's grid, not the shard: all shards of a job add in the same order)
      hipLaunchKernelGGL(k_emission_rows, dim3((unsigned)((nsh + 255) / 256)), dim3(256), 0, st, E);
    else
      hipLaunchKernelGGL(k_emission, dim3((unsigned)((nsh + kEmisWaves - 1) / kEmisWaves)), dim3(64 * kEmisWaves), 0, st, E);
  } else {
    ModArgs M{};
    mod_args(M);
    if (h->nwn > kEmisRowsAbove)
      hipLaunchKernelGGL(k_modulation_rows, dim3((unsigned)((nsh + 255) / 256)), dim3(256), 0, st, M);
    else
      hipLaunchKernelGGL(k_modulation, dim3((unsigned)((nsh + kModWaves - 1) / kModWaves)), dim3(64 * kModWaves), 0, st, M);
  }
  HIPCHK(h, hipGetLastError());
  if (prof) HIPCHK(h, hipEventRecord(ev.b, tst));

  // ---- results back -----------------------------------------------------------
  {   // one copy into pinned memory: flags, status and (profiled runs) the counters
    const size_t nb = count ? 128 + 24 * (size_t)nr : 128;
    if (!(tail_direct && !resumed)) HIPCHK(h, hipMemcpyAsync(h->h_small, h->d_small.p, nb, hipMemcpyDeviceToHost, st));
    if (spectrum && !(tail_spec && !resumed)) HIPCHK(h, hipMemcpyAsync(stage_spec ? h->h_spec : (void *)spectrum, d_out, sizeof(double) * nsh, hipMemcpyDeviceToHost, st));
    lap("spectrum+copies");
  }
  t_host_queued = std::chrono::steady_clock::now();
  }
  {
    const bool staged = spectrum && ((tail_spec && !resumed) || stage_spec);       // the tail stored the spectrum into the handle's pinned buffer, or the copy command did
    if (tail_on_side && !resumed) HIPCHK(h, hipStreamSynchronize(h->stream4));      // (the tail has waited for the main queue's walk: nothing is left there)
    HIPCHK(h, hipStreamSynchronize(st));
    if (staged) std::memcpy(spectrum, h->h_spec, sizeof(double) * (size_t)nsh);
    if (tail_hostsum && !resumed) {
      // the flags as tau_publish leaves them after the plan's last step (all of them zero but the rays' number
      // when the tail started: it covers the run from its first layer), from the blocks' entries
      const int32_t *tb = (const int32_t *)h->h_tailblk;
      int still = 0, deep = 0;
      for (size_t b = 0; b < tail_blocks; b++) { still += tb[2 * b]; deep = std::max(deep, tb[2 * b + 1]); }
      const int f[8] = {still, 0, tail_nct, 0, deep, 0, 0, 0};
      std::memcpy(h->h_small, f, sizeof f);
      int st4[4] = {0, 0, 0, 0};
      for (int code = 1; code < 4; code++) if (tb[2 * tail_blocks + code]) st4[0] = code;      // (vertical rays raise none)
      std::memcpy((char *)h->h_small + 64, st4, 16);
    }
    std::memcpy(flags_host, h->h_small, sizeof(flags_host));
    std::memcpy(status_host, (const char *)h->h_small + 64, sizeof(status_host));
    if (count) std::memcpy(counters.data(), (const char *)h->h_small + 128, 24 * (size_t)nr);
    else std::fill(counters.begin(), counters.end(), 0ull);
  }
  // rays still descending below the expected depth (the atmosphere changed): go on from there
  if (stop_at_hint && flags_host[0] > 0 && r_top >= 0) {
    // (the step kernels that go on from here read the flags on the device: where the host has added them up, they go there first)
    if (tail_hostsum && !resumed) HIPCHK(h, hipMemcpyAsync(h->d_flags.p, h->h_small, 32, hipMemcpyHostToDevice, st));
    stop_at_hint = false; resumed = true; h->hint_layers = 0; continue;
  }
  break;
  }

  drain.armed = false;                     // everything was joined into the main stream and waited for
  h->kmax_clean = true;
  RunOutcome Q{};
  std::memcpy(Q.flags, flags_host, sizeof Q.flags); std::memcpy(Q.status, status_host, sizeof Q.status);
  Q.counters = &counters; Q.layer_walked = &layer_walked; Q.spans = prof ? &spans : nullptr; Q.ev_a = ev.a; Q.ev_b = ev.b;
  Q.nchunks = nchunks; Q.ms_cia = ms_cia; Q.t_host0 = t_host0; Q.t_host_prep = t_host_prep; Q.t_host_queued = t_host_queued; Q.laps = &laps;
  return run_finish(h, a, o, dbg, Q, model_args);
}

int trx_sweep_permol(trx_handle *h, int32_t nv, const double *temp, const double *density, const double *zpart,
                     double ethresh, int32_t nslot, const int32_t *iso_slot, double *out)
{
  if (!h || nv < 1 || !temp || !density || !zpart || !out || nslot < 1 || !iso_slot) return TRX_E_ARG;
  if (!(ethresh > 0)) return fail(h, TRX_E_ARG, "ethresh must be positive");
  const int niso = h->niso; const int64_t nsh = h->nsh;
  for (int i = 0; i < niso; i++) {
    if (iso_slot[i] < 0 || iso_slot[i] >= nslot) return fail(h, TRX_E_ARG, "isotope slot out of range");
    if (i > 0 && iso_slot[i] < iso_slot[i-1]) return fail(h, TRX_E_UNSUPPORTED, "isotopes of one molecule must be contiguous");
  }
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t st = h->stream;
  int rc;
  LayerHost LH(h->run_f64, h->run_i32);
  if ((rc = prep_layers(h, nv, temp, density, zpart, 0, LH))) return rc;
  h->walk_temp_ok = true;
  for (int r = 0; r < nv; r++) if (temp[r] < kWalkMinTemp) h->walk_temp_ok = false;
  const size_t gr_b = (size_t)std::max<int64_t>(h->ngroups, 1);
  std::vector<int32_t> slots(iso_slot, iso_slot + std::max(niso, 1));
  bool any_wide = false;
  for (int r = 0; r < nv && !any_wide; r++) any_wide = walk_frame_bins(h, LH.psmax, r) == 0;
  const int sg_layers = any_wide ? 12 : 1;
  if ((rc = ensure(h, h->d_SG, sizeof(double) * gr_b * sg_layers)) || (rc = ensure(h, h->d_idop8, gr_b * sg_layers)) ||
      (rc = ensure_small(h, nv)) ||
      (rc = ensure(h, h->d_pm, sizeof(double) * (size_t)nv * nslot * nsh)) ||
      (rc = upload(h, h->d_pm_f64, LH.f64)) || (rc = upload(h, h->d_pm_i32, LH.i32)) || (rc = upload(h, h->d_iso_mx, slots)))
    return rc;
  HIPCHK(h, hipMemsetAsync(h->d_pm.p, 0, sizeof(double) * (size_t)nv * nslot * nsh, st));
  LayerDev Y{}; const double *d_wcut; const int32_t *d_npre;
  layer_dev(h->d_pm_f64.as<double>(), h->d_pm_i32.as<int32_t>(), LH, nv, Y, d_wcut, d_npre);
  // (its own maxima array: [state][slot]; the runs' two halves are left dirty, trx_run re-zeroes them)
  if ((rc = ensure(h, h->d_kmax, sizeof(double) * (size_t)nv * nslot))) return rc;
  HIPCHK(h, hipMemsetAsync(h->d_kmax.p, 0, sizeof(double) * (size_t)nv * nslot, st));
  h->kmax_clean = false;
  if ((rc = layer_maxima_and_sticky(h, Y, d_npre, nv, temp, nslot, h->d_iso_mx.as<int32_t>(), ethresh, st, h->d_kmax.as<double>()))) return rc;
  for (int r_top = nv - 1; r_top >= 0 && h->ngroups > 0; ) {
    int nb = walk_frame_bins(h, LH.psmax, r_top);
    int nc = 1;
    const int cap = nb ? kWalkLayers : sg_layers;
    while (nc < cap && nc <= r_top && (walk_frame_bins(h, LH.psmax, r_top - nc) == 0) == (nb == 0)) nc++;
    if (nb) for (int c = 1; c < nc; c++) nb = std::max(nb, walk_frame_bins(h, LH.psmax, r_top - c));
    SweepMode M{};
    M.eager = true; M.ethresh = ethresh; M.nmx = nslot; M.d_iso_mx = h->d_iso_mx.as<int32_t>();
    M.permol = true; M.d_e = h->d_pm.as<double>(); M.d_kmax = h->d_kmax.as<double>(); M.d_sticky = h->d_sticky.as<int>();
    if (nb) rc = walk_chunk(h, Y, d_wcut, nb, r_top, nc, M, nullptr, 0, nullptr, nullptr);
    else    rc = sweep_chunk(h, Y, d_wcut, LH.psmax, r_top, nc, sg_layers, M, nullptr);
    if (rc) return rc;
    r_top -= nc;
  }
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(out, h->d_pm.p, sizeof(double) * (size_t)nv * nslot * nsh, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  return TRX_OK;
}

int trx_run(trx_handle *h, const trx_atm *a, const trx_opts *o, double *spectrum, trx_debug *dbg)
{
  if (!spectrum) return TRX_E_ARG;
  return run_once(h, a, o, spectrum, nullptr, dbg);
}

int trx_run_device(trx_handle *h, const trx_atm *a, const trx_opts *o, void *d_spectrum, trx_debug *dbg)
{
  if (!d_spectrum) return TRX_E_ARG;
  return run_once(h, a, o, nullptr, d_spectrum, dbg);
}

// ---- several atmospheres per call -----------------------------------------------------------------
// A retrieval driver runs many chains over one line list (the reference: one run_transit per atmosphere,
// transit.c:118-122, one process each).  One spectrum leaves the device idle between its kernels and while
// the host prepares and queues (about a fifth of a CH4-demo spectrum), and another
// spectrum's kernels fit there: a batch keeps `ways` handles made from ONE description, each with a host
// thread of its own, and deals the K atmospheres of a call to them.  Every spectrum is what trx_run gives
// for its atmosphere, bit for bit -- it IS a trx_run, on whichever handle was free (a run's sums do not
// depend on the handle's history: the depth hint only changes the step plan).
struct trx_batch {
  std::vector<trx_handle *> hs;
  std::vector<std::thread> workers;
  std::mutex mu; std::condition_variable cv_work, cv_done;
  // the call being served (under mu)
  uint64_t epoch = 0; bool quit = false;
  int32_t k = 0; const trx_atm *atm = nullptr; const trx_opts *opts = nullptr; double *const *spectra = nullptr;
  std::atomic<int32_t> next{0};
  int32_t busy = 0; int rc = TRX_OK; std::string err;
};

int trx_batch_create(const trx_static *st, int32_t ways, trx_batch **out)
{
  g_comm_err.clear();
  if (!st || !out || ways < 1 || ways > TRX_BATCH_MAX_WAYS) { g_comm_err = "batch: bad argument (ways 1.." + std::to_string(TRX_BATCH_MAX_WAYS) + ")"; return TRX_E_ARG; }
  *out = nullptr;
  std::unique_ptr<trx_batch> B(new (std::nothrow) trx_batch);
  if (!B) { g_comm_err = "batch: out of host memory"; return TRX_E_NOMEM; }
  // (no C++ exception crosses the C boundary: a failed handle, allocation or thread start gives the handles back,
  // joins the workers that did start and returns a code, its text through trx_last_error(NULL))
  auto give_up = [&](int rc, const std::string &why) {
    { std::lock_guard<std::mutex> lk(B->mu); B->quit = true; }
    B->cv_work.notify_all();
    for (std::thread &t : B->workers) if (t.joinable()) t.join();
    for (trx_handle *x : B->hs) trx_destroy(x);
    B->hs.clear(); B->workers.clear();
    g_comm_err = why;
    return rc;
  };
  for (int i = 0; i < ways; i++) {
    trx_handle *h = nullptr;
    const int rc = trx_create(st, &h);
    if (rc != TRX_OK) return give_up(rc, "batch: handle " + std::to_string(i) + ": " + trx_strerror(rc));
    try { B->hs.push_back(h); } catch (...) { trx_destroy(h); return give_up(TRX_E_NOMEM, "batch: out of host memory"); }
  }
  trx_batch *b = B.get();
  try {
  for (int i = 0; i < ways; i++)
    b->workers.emplace_back([b, i]() {
      uint64_t seen = 0;
      for (;;) {
        {
          std::unique_lock<std::mutex> lk(b->mu);
          b->cv_work.wait(lk, [&] { return b->quit || b->epoch != seen; });
          if (b->quit) return;
          seen = b->epoch;
        }
        for (;;) {
          const int32_t j = b->next.fetch_add(1);
          if (j >= b->k) break;
          const int rc = trx_run(b->hs[(size_t)i], b->atm + j, b->opts, b->spectra[j], nullptr);
          if (rc != TRX_OK) {
            std::lock_guard<std::mutex> lk(b->mu);
            if (b->rc == TRX_OK) { b->rc = rc; b->err = "atmosphere " + std::to_string(j) + ": " + b->hs[(size_t)i]->err; }
            b->next.store(b->k);                           // (the others finish the spectrum they are on and stop)
          }
        }
        std::lock_guard<std::mutex> lk(b->mu);
        if (--b->busy == 0) b->cv_done.notify_all();
      }
    });
  } catch (const std::bad_alloc &) { return give_up(TRX_E_NOMEM, "batch: out of host memory");
  } catch (const std::exception &e) { return give_up(TRX_E_HIP, std::string("batch: worker thread: ") + e.what()); }
  *out = B.release();
  return TRX_OK;
}

int trx_run_batch(trx_batch *b, int32_t k, const trx_atm *atm, const trx_opts *opts, double *const *spectra)
{
  g_comm_err.clear();
  if (!b || k < 0 || (k > 0 && (!atm || !opts || !spectra))) return TRX_E_ARG;
  for (int32_t j = 0; j < k; j++) if (!spectra[j]) return TRX_E_ARG;
  if (k == 0) return TRX_OK;
  std::unique_lock<std::mutex> lk(b->mu);
  b->k = k; b->atm = atm; b->opts = opts; b->spectra = spectra;
  b->next.store(0); b->rc = TRX_OK; b->err.clear();
  b->busy = (int32_t)b->workers.size();
  b->epoch++;
  b->cv_work.notify_all();
  b->cv_done.wait(lk, [&] { return b->busy == 0; });
  if (b->rc != TRX_OK) g_comm_err = b->err;                // trx_last_error(NULL)
  return b->rc;
}

int trx_batch_ways(const trx_batch *b) { return b ? (int)b->hs.size() : 0; }

void trx_batch_destroy(trx_batch *b)
{
  if (!b) return;
  { std::lock_guard<std::mutex> lk(b->mu); b->quit = true; }
  b->cv_work.notify_all();
  for (std::thread &t : b->workers) t.join();
  for (trx_handle *h : b->hs) trx_destroy(h);
  delete b;
}

}  // extern "C"
