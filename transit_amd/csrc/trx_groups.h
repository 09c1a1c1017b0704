// trx_groups.h -- host side of trx_create that is linear in the line list, on a few host threads:
// the co-added groups of the reference (extinction.c:445-462) and the per-bin group counts.
// Plain C++ (no HIP): shared by trx_api.hip and by tests/groups_check.cpp, which checks the
// threaded forms against the one-thread loops on the CPU.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

namespace trx {

// host array WITHOUT initialisation (a std::vector of 10^7 doubles spends 10 ms of one thread
// zeroing what the next loop overwrites; these are written by several threads, first touch included)
template <class T>
struct HostBuf {
  std::unique_ptr<T[]> p; size_t n = 0;
  void alloc(size_t m) { p.reset(m ? new T[m] : nullptr); n = m; }
  T *data() { return p.get(); }
  const T *data() const { return p.get(); }
  T &operator[](size_t i) { return p[i]; }
  const T &operator[](size_t i) const { return p[i]; }
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
  T *begin() { return p.get(); }
  T *end() { return p.get() + n; }
};

inline int create_threads()
{
  if (const char *e = std::getenv("TRX_CREATE_THREADS")) return std::max(1, std::atoi(e));
  const unsigned hc = std::thread::hardware_concurrency();
  return (int)std::min<unsigned>(std::max<unsigned>(hc, 1u), 16u);
}

// f(part, begin, end) over [0, n) cut into `parts` contiguous pieces, one thread each
template <class F>
void parallel_parts(int64_t n, int parts, F f, int64_t grain = 65536)
{
  parts = (int)std::max<int64_t>(1, std::min<int64_t>(parts, n / grain + 1));     // (small lists: one thread)
  if (parts == 1) { f(0, (int64_t)0, n); return; }
  std::vector<std::thread> th;
  for (int t = 0; t < parts; t++) th.emplace_back([=]() { f(t, n * t / parts, n * (t + 1) / parts); });
  for (auto &x : th) x.join();
}

// cnt[k] = number of the block's groups with key >= k, k = 0..nkey (keys descend along the block)
template <class Key>
void count_ge(const int32_t *giown, int g0, int g1, long long nkey, Key key, int32_t *cnt, int nth, int64_t grain = 65536)
{
  parallel_parts(nkey + 1, nth, [&](int, int64_t k0, int64_t k1) {
    // groups with key >= k1 - 1: found by bisection, then the pointer only moves forward as k falls
    int a = g0, z = g1;
    while (a < z) { const int m = (a + z) >> 1; if (key(giown[m]) >= k1 - 1) a = m + 1; else z = m; }
    int p = a;
    for (int64_t k = k1 - 1; k >= k0; k--) {
      while (p < g1 && key(giown[p]) >= k) p++;
      cnt[k] = p - g0;
    }
  }, grain);
}

// Co-added groups (extinction.c:445-462): a greedy chain -- a group's anchor decides which of the
// following lines join it, and the line after them is the next anchor -- so it is sequential by
// nature.  But a line that starts an isotope block, or lies more than 1.6 fine-grid steps below
// its predecessor, can belong to NO earlier group (an anchor's grid point is at most half a step
// from it): the chain restarts there whatever came before.  The list is cut at such lines and the
// pieces are grouped side by side, each with the reference's own loop; a piece without such a
// line nearby (a list much denser than the fine grid) is simply left to the piece before it.
struct LineGroups {
  HostBuf<int32_t> first, count, iown; HostBuf<int16_t> iso; HostBuf<double> wavn;   // per group, in line order
  int64_t nadd = 0;                                   // lines co-added to an earlier one
  std::vector<double> iso_wmin, iso_wmax;             // anchor wavenumber range per isotope
};

inline void group_lines(int64_t n, const int16_t *isoid, const double *wavn, const uint8_t *inr, int niso,
                        double wn0, double odwn, int nth, LineGroups &G, int64_t grain = 65536)
{
  auto own = [&](long long k) { return wn0 + (double)k * odwn; };
  G.iso_wmin.assign(niso, HUGE_VAL); G.iso_wmax.assign(niso, 0.0); G.nadd = 0;
  int parts = (int)std::max<int64_t>(1, std::min<int64_t>(nth, n / grain + 1));
  std::vector<int64_t> cut;                       // piece p = lines [cut[p], cut[p+1])
  cut.push_back(0);
  for (int t = 1; t < parts; t++) {
    const int64_t from = std::max<int64_t>(n * t / parts, cut.back() + 1), to = std::min<int64_t>(n, from + n / parts / 2);
    for (int64_t l = from; l < to; l++)
      if (isoid[l] != isoid[l-1] || wavn[l-1] - wavn[l] >= 1.6 * odwn) { cut.push_back(l); break; }
  }
  cut.push_back(n);
  parts = (int)cut.size() - 1;
  struct Piece { std::vector<int32_t> first, count, iown; std::vector<int16_t> iso; std::vector<double> wv; int64_t nadd = 0; std::vector<double> wmin, wmax; };
  std::vector<Piece> P((size_t)parts);
  auto work = [&](int p) {
    Piece &Q = P[(size_t)p];
    Q.wmin.assign(niso, HUGE_VAL); Q.wmax.assign(niso, 0.0);
    const int64_t l0 = cut[(size_t)p], l1 = cut[(size_t)p + 1];
    const size_t guess = (size_t)(l1 - l0) / 2 + 16;
    Q.first.reserve(guess); Q.count.reserve(guess); Q.iown.reserve(guess); Q.iso.reserve(guess); Q.wv.reserve(guess);
    for (int64_t ln = l0; ln < l1; ln++) {
      if (!inr[ln]) continue;
      const double w = wavn[ln]; const int iso = isoid[ln];
      int iown = (int)((w - wn0) / odwn);                          // extinction.c:445-447
      if (std::fabs(w - own(iown + 1)) < std::fabs(w - own(iown))) iown++;
      const int64_t first = ln;
      while (ln != n - 1 && isoid[ln + 1] == iso) {                // extinction.c:449-462
        if (std::fabs(wavn[ln + 1] - own(iown)) < odwn) { Q.nadd++; ln++; }
        else break;
      }
      Q.first.push_back((int32_t)first); Q.count.push_back((int32_t)(ln - first + 1));
      Q.iown.push_back(iown); Q.iso.push_back((int16_t)iso); Q.wv.push_back(w);
      Q.wmin[iso] = std::min(Q.wmin[iso], w); Q.wmax[iso] = std::max(Q.wmax[iso], w);
    }
  };
  if (parts == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (int p = 0; p < parts; p++) th.emplace_back(work, p);
    for (auto &x : th) x.join();
  }
  std::vector<size_t> at((size_t)parts + 1, 0);
  for (int p = 0; p < parts; p++) at[(size_t)p + 1] = at[(size_t)p] + P[(size_t)p].first.size();
  const size_t ng = at[(size_t)parts];
  G.first.alloc(ng); G.count.alloc(ng); G.iown.alloc(ng); G.iso.alloc(ng); G.wavn.alloc(ng);
  auto gather = [&](int p) {
    const Piece &Q = P[(size_t)p]; const size_t o = at[(size_t)p], m = Q.first.size();
    if (!m) return;
    std::memcpy(&G.first[o], Q.first.data(), 4 * m); std::memcpy(&G.count[o], Q.count.data(), 4 * m);
    std::memcpy(&G.iown[o], Q.iown.data(), 4 * m); std::memcpy(&G.iso[o], Q.iso.data(), 2 * m); std::memcpy(&G.wavn[o], Q.wv.data(), 8 * m);
  };
  if (parts == 1) gather(0);
  else {
    std::vector<std::thread> th;
    for (int p = 0; p < parts; p++) th.emplace_back(gather, p);
    for (auto &x : th) x.join();
  }
  for (int p = 0; p < parts; p++) {
    G.nadd += P[(size_t)p].nadd;
    for (int b = 0; b < niso; b++) { G.iso_wmin[b] = std::min(G.iso_wmin[b], P[(size_t)p].wmin[b]); G.iso_wmax[b] = std::max(G.iso_wmax[b], P[(size_t)p].wmax[b]); }
  }
}

}  // namespace trx
