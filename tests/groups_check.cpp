// groups_check -- the threaded host loops of trx_create (transit_amd/csrc/trx_groups.h) against
// one-thread loops written out here: co-added groups (extinction.c:445-462) and the per-key
// counts, on line lists that are sparse, as dense as the fine grid, much denser than it, made of
// many small isotope blocks, partly out of range, and with ties.  Prints "<cases> cases, <bad> differ".
#include <cstdio>
#include <random>
#include "trx_groups.h"

struct Plain { std::vector<int32_t> first, count, iown; std::vector<int16_t> iso; std::vector<double> wv; int64_t nadd = 0; };

static Plain plain_groups(int64_t n, const int16_t *isoid, const double *wavn, const uint8_t *inr, double wn0, double odwn)
{
  Plain P;
  for (int64_t ln = 0; ln < n; ln++) {
    if (!inr[ln]) continue;
    const double w = wavn[ln];
    int k = (int)((w - wn0) / odwn);
    if (std::fabs(w - (wn0 + (k + 1) * odwn)) < std::fabs(w - (wn0 + k * odwn))) k++;
    const int64_t f = ln;
    while (ln != n - 1 && isoid[ln + 1] == isoid[f] && std::fabs(wavn[ln + 1] - (wn0 + k * odwn)) < odwn) { ln++; P.nadd++; }
    P.first.push_back((int32_t)f); P.count.push_back((int32_t)(ln - f + 1)); P.iown.push_back(k);
    P.iso.push_back(isoid[f]); P.wv.push_back(w);
  }
  return P;
}

int main(int argc, char **argv)
{
  const int rounds = argc > 1 ? std::atoi(argv[1]) : 40;
  std::mt19937_64 rng(20260);
  int cases = 0, bad = 0;
  for (int r = 0; r < rounds; r++) {
    const int64_t n = 1 + (int64_t)(rng() % 60000);
    const int niso = 1 + (int)(rng() % (r % 5 == 0 ? 200 : 6));
    const double wn0 = 2000.0, odwn = (r % 3 == 0) ? 0.004 : (r % 3 == 1 ? 0.1 : 1.0);
    // mean spacing of the lines in units of the fine step: 0.01 (every group takes ~100) .. 30
    const double dens[] = {0.01, 0.3, 1.0, 1.7, 30.0};
    const double mean = dens[r % 5] * odwn;
    std::vector<int16_t> iso((size_t)n); std::vector<double> wv((size_t)n); std::vector<uint8_t> inr((size_t)n);
    std::exponential_distribution<double> gap(1.0 / mean);
    int64_t at = 0;
    for (int b = 0; b < niso && at < n; b++) {              // blocks of one isotope each, wavenumbers descending
      const int64_t m = (b == niso - 1) ? n - at : std::min<int64_t>(n - at, 1 + (int64_t)(rng() % (2 * n / niso + 1)));
      double w = wn0 + mean * (double)m * 1.05 + 3 * odwn;
      for (int64_t i = 0; i < m; i++, at++) {
        if (rng() % 7) w -= gap(rng);                         // (one in seven: the same wavenumber again)
        if (r % 4 == 1) w = wn0 + std::round((w - wn0) / (odwn / 2)) * (odwn / 2);   // on grid points and half-way between them
        iso[(size_t)at] = (int16_t)b; wv[(size_t)at] = w;
      }
    }
    const double hi = wn0 + mean * (double)n / niso * 0.9;
    for (int64_t i = 0; i < n; i++) inr[(size_t)i] = wv[(size_t)i] >= wn0 && wv[(size_t)i] <= hi;
    const Plain P = plain_groups(n, iso.data(), wv.data(), inr.data(), wn0, odwn);
    for (int nth : {1, 2, 3, 8, 16}) {
      for (int64_t grain : {(int64_t)65536, (int64_t)257, (int64_t)16}) {
        trx::LineGroups G;
        trx::group_lines(n, iso.data(), wv.data(), inr.data(), niso, wn0, odwn, nth, G, grain);
        cases++;
        bool ok = G.first.size() == P.first.size() && G.nadd == P.nadd;
        for (size_t g = 0; ok && g < P.first.size(); g++)
          ok = G.first[g] == P.first[g] && G.count[g] == P.count[g] && G.iown[g] == P.iown[g] && G.iso[g] == P.iso[g] && G.wavn[g] == P.wv[g];
        std::vector<double> mn((size_t)niso, HUGE_VAL), mx((size_t)niso, 0.0);
        for (size_t g = 0; g < P.first.size(); g++) { mn[P.iso[g]] = std::min(mn[P.iso[g]], P.wv[g]); mx[P.iso[g]] = std::max(mx[P.iso[g]], P.wv[g]); }
        ok = ok && mn == G.iso_wmin && mx == G.iso_wmax;
        if (!ok) { bad++; std::printf("groups differ: round %d n %lld niso %d odwn %g mean %g threads %d grain %lld\n", r, (long long)n, niso, odwn, mean, nth, (long long)grain); }
      }
    }
    // counts per key over one isotope block's groups (keys descend along the block)
    size_t g0 = 0;
    while (g0 < P.first.size()) {
      size_t g1 = g0; while (g1 < P.first.size() && P.iso[g1] == P.iso[g0]) g1++;
      const long long nkey = P.iown[g0] + 3;
      auto key = [](int32_t k) { return (long long)std::max(k, 0); };
      std::vector<int32_t> want((size_t)nkey + 1), got((size_t)nkey + 1);
      for (long long k = 0; k <= nkey; k++) { int c = 0; for (size_t g = g0; g < g1; g++) c += key(P.iown[g]) >= k; want[(size_t)k] = c; if (g1 - g0 > 4000 || nkey > 50000) break; }
      if (g1 - g0 <= 4000 && nkey <= 50000)
        for (int nth : {1, 3, 16}) {
          trx::count_ge(P.iown.data(), (int)g0, (int)g1, nkey, key, got.data(), nth, 64);
          cases++;
          if (got != want) { bad++; std::printf("counts differ: round %d block at %zu threads %d\n", r, g0, nth); }
        }
      g0 = g1;
      if (cases > 4000 * (r + 1)) break;
    }
  }
  std::printf("%d cases, %d differ\n", cases, bad);
  return bad != 0;
}
