// trx_rows.hip.h -- pass 2b for wide profiles on a grid WITHOUT oversampling (osamp == 1):
// k_accumulate_rows, the profile accumulation of extinction.c:485-509 for BASELINE configs[4]
// (10^7 bins x 150 layers x 10^7 lines, profiles hundreds to thousands of bins wide).
//
// With osamp == 1 bin j of a group in cell c reads entry j - c + ps of its profile: every group
// of one profile reads the SAME row, shifted by its cell.  k_accumulate_wide (the general form,
// trx_kernels.hip.h) fetches 1 KB of that row from L1 per (group, 256-bin tile) and is bound by
// exactly that: 64 groups touch ~400 different floats and issue 64 KB of reads.  Here a wavefront
//   phase A (one lane per group, as before): threshold, profile, distance of the tile's first
//     bin from the start of the row -- the integer arithmetic of extinction.c:476-501;
//   cuts the surviving groups, IN LINE ORDER, into runs that read one row within kRowSpan cells;
//   stages the piece of the row a run can touch (256 + span floats) in LDS ONCE -- twice in
//     fact, the second copy one float on, so that every read below is an aligned 8-byte read --
//     and zeroes what lies outside the row while doing so: no masks afterwards, a bin a
//     profile does not reach adds strength x 0 = +0;
//   phase B (lanes = bins): per group one distance and one strength, wave-uniform, taken from
//     the group's lane into scalar registers (v_readlane), ONE vector add for the LDS address,
//     two ds_read_b64 (lane l owns bins 2l, 2l+1, 128+2l, 129+2l) and four convert + fused
//     multiply-adds.  No global load, no LDS broadcast, no mask in the loop.
// Sums: one owner per bin, groups in line order, the same fused multiply-add as
// k_accumulate_wide => the same bits (tests/test_gpu_parity.py compares the two forms).
#pragma once
#include "trx_kernels.hip.h"

namespace trx {

constexpr int kRowSpan = 256;                 // cells between the first and the last group of a run, at most
constexpr int kRowMaxT = 512;                 // bins of the larger tile
constexpr int kRowTail = kRowMaxT + kRowSpan + 8;   // zero floats the table carries behind its last profile
static_assert(kRowMaxT <= kTabPad, "a tile's first bin lies at most T - 1 floats in front of a row");

// bytes of dynamic LDS of a 4-wave block whose lanes own M bins each
__host__ __device__ inline size_t rows_lds_bytes(int ndop, int M)
{
  const size_t ndp = ((size_t)ndop + 1) & ~(size_t)1;
  return 4 * (size_t)(64 * M + kRowSpan) * sizeof(double) + 4 * ndp * (sizeof(long long) + sizeof(int32_t));
}

__device__ __forceinline__ long long readlane_i64(long long v, int l)
{
  const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffll), l);
  const int hi = __builtin_amdgcn_readlane((int)(v >> 32), l);
  return ((long long)hi << 32) | (unsigned int)lo;
}

struct alignas(4) RowQuad { float v[4]; };
__device__ __forceinline__ RowQuad row_load4(const float *p)
{
  RowQuad q;
  __builtin_memcpy(&q, p, sizeof(q));
  return q;
}

// lowest set bit of a wave-uniform mask, cleared (s_ff1_i32_b64 + s_bitset0_b64)
__device__ __forceinline__ int take_lowest(unsigned long long &m)
{
  const int l = __builtin_ctzll(m);
  asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(l));
  return l;
}

#if defined(__HIP_DEVICE_COMPILE__)
#define TRX_SEPARATE_LDS_READS __attribute__((target("no-load-store-opt")))
#else
#define TRX_SEPARATE_LDS_READS          /* (a device-code attribute: the host pass does not know it) */
#endif

// One group's share of a tile: M bins per lane (bins lane, lane + 64, ...), each one 8-byte LDS
// read and one fused multiply-add with the group's strength in scalar registers.
template <int M> struct RowVals { double v[M]; };
template <int M>
__device__ __forceinline__ RowVals<M> row_fetch(const char *p)
{
  RowVals<M> v;
#pragma unroll
  for (int m = 0; m < M; m++) v.v[m] = *(const double *)(p + 512 * m);
  return v;
}
template <int M>
__device__ __forceinline__ void row_apply(double (&acc)[M], double s, const RowVals<M> &v)
{
#pragma unroll
  for (int m = 0; m < M; m++) acc[m] = __builtin_fma(s, v.v[m], acc[m]);
}

// grid: x = blocks of 4 tiles of 64 M bins, y = layer of the chunk; dynamic LDS rows_lds_bytes(ndop, M).
// M = 4 or 8: what a group costs besides its bins (three v_readlane, one add, its share of phase A
// and of the staging) is the same for a tile of 256 and of 512 bins, but a group whose profile ends
// inside a tile still pays for the whole tile: 8 bins per lane for layers whose profiles are much
// wider than a tile, 4 for the others (sweep_chunk decides per layer; the sums do not depend on it).
// (no-load-store-opt: the pass would pair the 8-byte LDS reads into ds_read2st64_b64, which the
// LDS serves at half the bytes per clock of two ds_read_b64)
template <bool COUNT, int M>
__global__ __launch_bounds__(256) TRX_SEPARATE_LDS_READS
void k_accumulate_rows(WideArgs W)
{
  constexpr int T = 64 * M;                                  // bins per wavefront tile
  constexpr int kSeg = T + kRowSpan;                         // doubles staged per run, at most
  const AccumArgs &A = W.A;
  if (!A.eager && A.flags[0] == 0) return;
  const int c = blockIdx.y, bx = (int)blockIdx.x;
  if (!((W.layer_mask >> c) & 1u)) return;
  extern __shared__ __align__(16) unsigned char s_rows[];
  __shared__ long long s_nb[4][3];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ndp = (A.ndop + 1) & ~1;
  char      *s_seg = (char *)s_rows + wv * (kSeg * sizeof(double));                     // this wave's staged piece
  long long *s_po  = (long long *)(s_rows + 4 * kSeg * sizeof(double)) + wv * ndp;
  int32_t   *s_ps  = (int32_t *)(s_rows + 4 * kSeg * sizeof(double) + 4 * (size_t)ndp * sizeof(long long)) + wv * ndp;
  const int tile = bx * 4 + wv;
  const int ntiles = (int)((A.nsh + T - 1) / T);
  const int r = A.r_top - c;
  bool live = tile < ntiles;
  const long long j0 = A.lo + (long long)tile * T;
  const long long j1 = min(j0 + T, A.lo + A.nsh) - 1;
  const int jcount = (int)(j1 - j0 + 1);
  if (live && A.last) {
    bool open = false;
#pragma unroll
    for (int m = 0; m < M; m++) {
      const int t = lane + 64 * m;
      open |= (t < jcount) && (A.last[j0 - A.lo + t] < 0);
    }
    live = __ballot(open) != 0ull;
  }
  double acc[M];
#pragma unroll
  for (int m = 0; m < M; m++) acc[m] = 0.0;
  long long nb = 0, nev = 0, nsk = 0;
  int cur_mx = -1;
  auto flush = [&](int mx) {
    double *dst = A.e + ((long long)r * A.nmx + mx) * A.nsh + (j0 - A.lo);
#pragma unroll
    for (int m = 0; m < M; m++) {
      const int t = lane + 64 * m;
      if (t < jcount) dst[t] = acc[m];
      acc[m] = 0.0;
    }
  };
  const char *segl = s_seg + 8 * lane;

  if (live)
  for (int b = 0; b < A.niso; b++) {
    const int gb0 = A.L.gblock[b], gb1 = A.L.gblock[b + 1];
    if (gb0 == gb1) continue;
    const int mx = A.nmx == 1 ? 0 : A.iso_mx[b];
    if (mx != cur_mx) { if (cur_mx >= 0) flush(cur_mx); cur_mx = mx; }
    const double lim = A.ethresh * A.kmaxc[(long long)r * A.nmx + mx];
    const int ri = r * A.niso + b;
    const int il = A.Y.ilor[ri];
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < A.ndop; i += 64) {                // profile column of this (layer, isotope)
      s_ps[i] = A.psize[i * A.nlor + il];
      s_po[i] = W.poffT[i * A.nlor + il];
    }
    const int idst = A.sticky_idop[ri];
    const long long psm = A.Y.psmax[ri];
    const long long lo_f = j0 - psm;
    const long long klo = lo_f > 0 ? lo_f : 0;
    long long khi = j1 + psm;
    if (khi > A.nwn - 1) khi = A.nwn - 1;
    const int32_t *cg = A.L.cnt_ge + (long long)b * (A.nwn + 1);
    const int ga = gb0 + cg[khi + 1], gz = gb0 + cg[klo];
    const double  *SGr = A.SG    + (long long)c * A.L.ngroups;
    const uint8_t *idr = A.idop8 + (long long)c * A.L.ngroups;
    const double dens = A.permol ? 1.0 : A.Y.density[ri];
    __builtin_amdgcn_wave_barrier();
    for (int gbase = ga; gbase < gz; gbase += 64) {
      // ---- phase A: one lane per group
      const int g = gbase + lane;
      bool keep = false;
      double sg = 0; long long off = 0; int id = 0;
      if (g < gz) {
        const double sg0 = SGr[g];
        const int iown = A.L.giown[g];
        const bool below = sg0 < lim;                          // extinction.c:467
        if (COUNT && iown >= j0 && iown <= j1) { if (below) nsk++; else nev++; }
        if (!below) {
          id = idr[g];
          if (id == 0xFF) id = idst;
          const int ps = s_ps[id];
          // entry of the row that bin j reads: k = j - iown + ps, valid for 0 <= k <= 2 ps
          const long long kk0 = j0 + ps - iown;               // k of the tile's first bin
          const long long ta = kk0 < 0 ? -kk0 : 0;
          long long tb = 2ll * ps - kk0;
          if (tb > jcount - 1) tb = jcount - 1;
          if (ta <= tb) {
            keep = true;
            sg = sg0 * dens;                                   // extinction.c:472-473
            off = s_po[id] + kk0;
            if (COUNT) nb += tb - ta + 1;
          }
        }
      }
      unsigned long long km = __ballot(keep);
      while (km) {
        // ---- a run: consecutive surviving groups of one row, within kRowSpan cells of the first
        const int i0 = __builtin_ctzll(km);
        const int id0 = __builtin_amdgcn_readlane(id, i0);
        const long long off0 = readlane_i64(off, i0);
        const long long d = off - off0;
        const bool inb = keep && id == id0 && (unsigned long long)d <= (unsigned long long)kRowSpan;
        const unsigned long long bad = km & ~__ballot(inb);
        unsigned long long sub = bad ? (km & ((1ull << __builtin_ctzll(bad)) - 1ull)) : km;
        km &= ~sub;
        const int dsa = (int)d * (int)sizeof(double);          // what this group adds to every lane's LDS address
        const int span = __builtin_amdgcn_readlane((int)d, 63 - __builtin_clzll(sub));
        // the row occupies staged indices [rlo, rlo + 2 ps0]
        const int ps0 = __builtin_amdgcn_readfirstlane(s_ps[id0]);
        const int rlo = (int)(readlane_i64(s_po[id0], 0) - off0);
        const unsigned rw = 2u * (unsigned)ps0;
        const float *src = W.tabT + off0;
        const bool inside = rlo <= 0 && (long long)rlo + rw >= kSeg - 1;     // the staged piece lies inside the row: nothing to zero
        __builtin_amdgcn_wave_barrier();                       // (the previous run's reads are done: LDS is in order per wave)
        for (int i = 4 * lane; i < T + span; i += 256) {       // (T + span <= kSeg; lanes past the end skip their last piece)
          RowQuad q = row_load4(src + i);
          if (!inside) {
#pragma unroll
            for (int e = 0; e < 4; e++)
              if ((unsigned)(i + e - rlo) > rw) q.v[e] = 0.f;
          }
          *(double2 *)(s_seg + 8 * i) = make_double2((double)q.v[0], (double)q.v[1]);
          *(double2 *)(s_seg + 8 * i + 16) = make_double2((double)q.v[2], (double)q.v[3]);
        }
        __builtin_amdgcn_wave_barrier();
        // ---- phase B: lanes = bins, groups in line order
        constexpr int U = M == 4 ? 4 : 2;                      // groups whose reads are in flight together
        while (__builtin_popcountll(sub) >= U) {
          int l[U]; RowVals<M> v[U];
#pragma unroll
          for (int u = 0; u < U; u++) l[u] = take_lowest(sub);
#pragma unroll
          for (int u = 0; u < U; u++) v[u] = row_fetch<M>(segl + __builtin_amdgcn_readlane(dsa, l[u]));
#pragma unroll
          for (int u = 0; u < U; u++) row_apply<M>(acc, readlane_f64(sg, l[u]), v[u]);
        }
        while (sub) {
          const int l0 = take_lowest(sub);
          const RowVals<M> v0 = row_fetch<M>(segl + __builtin_amdgcn_readlane(dsa, l0));
          row_apply<M>(acc, readlane_f64(sg, l0), v0);
        }
      }
    }
  }
  if (live && cur_mx >= 0) flush(cur_mx);
  if (COUNT) {
    nb = wave_sum_ll(nb); nev = wave_sum_ll(nev); nsk = wave_sum_ll(nsk);
    if (lane == 0) { s_nb[wv][0] = nb; s_nb[wv][1] = nev; s_nb[wv][2] = nsk; }
    __syncthreads();
    if (threadIdx.x < 3)
      A.part[((long long)c * A.part_stride + bx) * 3 + threadIdx.x] =
          (unsigned long long)(s_nb[0][threadIdx.x] + s_nb[1][threadIdx.x] + s_nb[2][threadIdx.x] + s_nb[3][threadIdx.x]);
  }
}

}  // namespace trx
