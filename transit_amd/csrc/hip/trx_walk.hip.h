// trx_walk.hip.h -- the line sweep of the narrow-profile regime as ONE kernel, and the exact
// per-layer maximum line strength from a pruned candidate set.
//
//   k_cand_cellmax / _prefix / _select   create time: lines that can never be a layer's
//                                        strongest line are pruned (dominance on a grid)
//   k_layer_max                          strongest single line of every layer (extinction.c:399-427)
//   k_wave_plan                          bin interval and partial-sum slot of every line range
//   k_line_walk<NB>                      passes 2a + 2b of computemolext (extinction.c:429-511)
//                                        for up to 64 layers at once: LANES ARE LAYERS (frames of
//                                        8+ bins in steps of <= 32 layers: two lanes per layer);
//                                        frames of 4+ bins read a row copy of the Voigt table
//   k_walk_combine                       adds the line ranges' partial sums into e[layer][wn]
//
// Why lanes = layers.  At 1 cm-1 output resolution a line reaches 0-2 coarse bins per layer, so
// the sweep is per-(line, layer) arithmetic: two exponentials, a threshold, a table look-up.
// With one lane per LAYER a wavefront walks a contiguous range of the (wavenumber-sorted) line
// list sequentially; the line's data is wave-uniform (scalar loads, one 32-byte record per
// line), everything that depends on the layer (-c/T, partition function, widths, profile
// column, threshold) is loaded once per wave and stays in registers, every lane owns its
// layer's accumulators for the few bins around the walk's current cell, and:
//   * no cross-lane reduction anywhere (the line sum of a bin is a per-lane running sum, in
//     line order like the reference's);
//   * no intermediate strength array (the two-kernel form writes and re-reads 9 + 13 bytes per
//     group and layer, 82 % of a demo-sized run's traffic);
//   * exp(-c nu/T) costs a 5-term polynomial: consecutive lines are < 1e-3 apart in c nu/T, so
//     it is rebased on a value computed once per ~100 lines.
// The price: the layer maximum (needed by the threshold test BEFORE a line is accumulated) must
// be known up front -- k_layer_max, exact, from the few thousand lines that survive the
// dominance filter -- and bins near the ends of a line range get contributions from two
// neighbouring ranges: every wave writes its bins as partial records and k_walk_combine adds
// them in a fixed order (isotope, then range, i.e. line order), so results are bitwise
// reproducible and independent of the step size and of the wavenumber shard.
#pragma once
#include "trx_kernels.hip.h"

namespace trx {

// ---------------------------------------------------------------------------
// candidate lines for the layer maximum
// ---------------------------------------------------------------------------
// Strength of a line in a layer (pass 1): K = gf * exp(-c Elow/T) * (1 - exp(-c nu/T)) * F_iso(T).
// For lines A, B of the SAME isotope with gf_A >= gf_B (1 + eps), Elow_A <= Elow_B and
// nu_A >= nu_B the exact K_A exceeds K_B by the factor (1 + eps) at EVERY temperature, and so
// does the computed one as long as eps (1e-9) is far above the relative error of the evaluation
// (~1e-15, and ~4e-16/(c nu/T) for the 1-exp factor: the host checks c nu_min/T_max >= 1e-5
// per run and falls back to all lines otherwise).  B can then never be a layer's strongest
// line.  Dominators are looked for on a G x G grid over (Elow, nu): M = largest gf per cell,
// D[a][b] = max of M over cells with strictly smaller Elow index and strictly larger nu index.
// Survivors: gf (1 + eps) > D[cell].  For random lists about 2/G of the lines survive.
constexpr int    kCandGrid = 128;
constexpr double kCandEps  = 1e-9;

struct CandGeom { double e_min, e_scale, w_min, w_scale; };   // cell = clamp((x - min) * scale)

__device__ __forceinline__ int cand_cell(double x, double x0, double scale)
{
  const double c = (x - x0) * scale;
  int k = (c > 0) ? (int)fmin(c, (double)(kCandGrid - 1)) : 0;
  return k;
}

__global__ __launch_bounds__(256)
void k_cand_cellmax(long long n, const double *__restrict__ wavn, const double *__restrict__ elow,
                    const double *__restrict__ gf, const int16_t *__restrict__ iso, const uint8_t *__restrict__ inr,
                    CandGeom Gm, unsigned long long *__restrict__ M /* [niso][G][G] */)
{
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n || !inr[i]) return;
  const double g = gf[i];
  if (!(g > 0)) return;
  const int a = cand_cell(elow[i], Gm.e_min, Gm.e_scale), b = cand_cell(wavn[i], Gm.w_min, Gm.w_scale);
  unsigned long long *slot = &M[((long long)iso[i] * kCandGrid + a) * kCandGrid + b];
  const unsigned long long bits = (unsigned long long)__double_as_longlong(g);
  if (bits > *slot) atomicMax(slot, bits);
}

// one block per isotope: D[a][b] = max M[a' < a][b' > b]  (in place: M becomes D)
__global__ __launch_bounds__(kCandGrid)
void k_cand_prefix(unsigned long long *__restrict__ M)
{
  __shared__ unsigned long long s[kCandGrid][kCandGrid + 1];
  unsigned long long *Mi = M + (long long)blockIdx.x * kCandGrid * kCandGrid;
  const int t = threadIdx.x;
  for (int a = 0; a < kCandGrid; a++) s[a][t] = Mi[a * kCandGrid + t];
  __syncthreads();
  // inclusive running maximum along a (thread = column b)
  for (int a = 1; a < kCandGrid; a++) s[a][t] = max(s[a][t], s[a - 1][t]);
  __syncthreads();
  // inclusive running maximum along descending b (thread = row a)
  for (int b = kCandGrid - 2; b >= 0; b--) s[t][b] = max(s[t][b], s[t][b + 1]);
  __syncthreads();
  for (int a = 0; a < kCandGrid; a++)
    Mi[a * kCandGrid + t] = (a > 0 && t < kCandGrid - 1) ? s[a - 1][t + 1] : 0ull;
}

__global__ __launch_bounds__(256)
void k_cand_select(long long n, const double *__restrict__ wavn, const double *__restrict__ elow,
                   const double *__restrict__ gf, const int16_t *__restrict__ iso, const uint8_t *__restrict__ inr,
                   CandGeom Gm, const unsigned long long *__restrict__ D,
                   int32_t *__restrict__ cand, int *__restrict__ ncand, int cap)
{
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n || !inr[i]) return;
  const double g = gf[i];
  if (!(g > 0)) return;                    // a line without strength is nobody's maximum (K = 0)
  const int a = cand_cell(elow[i], Gm.e_min, Gm.e_scale), b = cand_cell(wavn[i], Gm.w_min, Gm.w_scale);
  const double d = __longlong_as_double((long long)D[((long long)iso[i] * kCandGrid + a) * kCandGrid + b]);
  if (g * (1.0 + kCandEps) > d) {
    const int k = atomicAdd(ncand, 1);
    if (k < cap) cand[k] = (int32_t)i;
  }
}

// Start of a run: every small per-run buffer in ONE pass (a dozen memsets cost more than the
// spectrum's arithmetic at demo size) -- and that pass rides along with k_layer_max as one extra
// row of blocks, so that it is not a launch of its own on the critical path.  kmax here is the
// array of the NEXT run (two alternate): this run's was zeroed a run ago.
struct RunInit {
  int *last; long long nsh;            // -1: ray still descending   (nsh < 0: nothing to initialise)
  double *acc;                         // [2][nsh] running Simpson sums of the vertical rays
  unsigned long long *counters; int ncounters;
  double *kmax; int nkmax;
  int *status, *flags; int rays;
};

__device__ __forceinline__ void run_init_elements(const RunInit &R, long long t0, long long stride)
{
  const long long n = max(max(R.nsh, (long long)R.ncounters), max((long long)R.nkmax, 8LL));
  for (long long t = t0; t < n; t += stride) {
    if (t < R.nsh) { R.last[t] = -1; R.acc[t] = 0.0; R.acc[R.nsh + t] = 0.0; }
    if (t < R.ncounters) R.counters[t] = 0ull;
    if (t < R.nkmax) R.kmax[t] = 0.0;
    if (t < 4) R.status[t] = 0;
    if (t < 8) R.flags[t] = t == 0 ? R.rays : 0;
  }
}

__global__ __launch_bounds__(256)
void k_run_init(RunInit R)
{
  run_init_elements(R, (long long)blockIdx.x * 256 + threadIdx.x, (long long)gridDim.x * 256);
}

// the candidates' line data in one place: k_layer_max then needs one load round trip, not three
// (index -> range flag -> four arrays)
struct alignas(32) CandLine { double gf, elow, wavn; int32_t iso, pad; };

__global__ __launch_bounds__(256)
void k_cand_pack(int n, const int32_t *__restrict__ cand, const double *__restrict__ wavn, const double *__restrict__ elow,
                 const double *__restrict__ gf, const int16_t *__restrict__ iso, const uint8_t *__restrict__ inr, CandLine *__restrict__ out)
{
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const int ln = cand[t];
  CandLine c{0.0, 0.0, 0.0, 0, 0};
  if (inr[ln]) { c.gf = gf[ln]; c.elow = elow[ln]; c.wavn = wavn[ln]; c.iso = iso[ln]; }
  out[t] = c;
}

// Strongest single line of every layer.  One wavefront per 256 candidates (four per lane) and
// kLayerMaxGroup layers: the lines are read once, the lanes keep running maxima per layer and the
// wave reduces them once at the end.  cand == null: all lines.  kmax is [layer][nmx] bit
// patterns, zeroed beforehand.
constexpr int kLayerMaxGroup = 8;
constexpr int kLayerMaxLines = 4;
constexpr int kLayerMaxWaves = 128;   // waves per group of layers at most (each takes every xwaves-th chunk of 256 candidates)

constexpr int kLayerMaxBlock = 4;     // waves per block: their maxima meet in LDS, ONE atomic per block and layer (the atomics of
                                      // all waves of a layer go to one word, ~0.17 us each one after the other: 62 of them were
                                      // most of this kernel's 13 us at the demo size)

__global__ __launch_bounds__(64 * kLayerMaxBlock)
void k_layer_max(LinesDev L, LayerDev Y, int niso, int nr, const CandLine *__restrict__ cand, long long ncand,
                 const double *__restrict__ e2tab, int nmx, const int32_t *__restrict__ iso_mx,
                 unsigned long long *__restrict__ kmax_bits, RunInit R, int init_row, int xwaves)
{
  if ((int)blockIdx.y == init_row) {     // the extra row of blocks: the run's small buffers
    run_init_elements(R, (long long)blockIdx.x * (64 * kLayerMaxBlock) + threadIdx.x, (long long)gridDim.x * (64 * kLayerMaxBlock));
    return;
  }
  __shared__ double s_e2[64];
  __shared__ double s_ct[kLayerMaxGroup];
  __shared__ double s_best[kLayerMaxBlock][kLayerMaxGroup];
  extern __shared__ double s_f[];                      // [kLayerMaxGroup][niso] (dynamic: the launch sizes it)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int xw = blockIdx.x * kLayerMaxBlock + wv;     // this wave's place among the xwaves waves of its group of layers
  const int r0 = blockIdx.y * kLayerMaxGroup, r1 = min(r0 + kLayerMaxGroup, nr);
  // the block's layer scalars in LDS, one round trip for all of them (read per layer from global
  // memory they were a dependent round trip per layer of this one-wave chain)
  if (threadIdx.x < 64) s_e2[threadIdx.x] = e2tab[threadIdx.x];
  for (int t = threadIdx.x; t < (r1 - r0) * niso; t += 64 * kLayerMaxBlock) s_f[t] = Y.strength_f[r0 * niso + t];
  if ((int)threadIdx.x < r1 - r0) s_ct[threadIdx.x] = Y.negc_over_t[r0 + threadIdx.x];
  __syncthreads();
  // A wave takes the chunks xw, xw + xwaves, ... of 256 candidates and keeps its layers' running maxima over
  // all of them (a list of 8*10^6 lines has 1.2*10^5 candidates: one chunk per wave would be 488 waves)
  double best[kLayerMaxGroup];
#pragma unroll
  for (int q = 0; q < kLayerMaxGroup; q++) best[q] = 0.0;
  const long long nchunk = (ncand + 64 * kLayerMaxLines - 1) / (64 * kLayerMaxLines);
  for (long long chunk = xw; chunk < nchunk && xw < xwaves; chunk += xwaves) {
    double gf[kLayerMaxLines], elow[kLayerMaxLines], wavn[kLayerMaxLines]; int iso[kLayerMaxLines];
#pragma unroll
    for (int u = 0; u < kLayerMaxLines; u++) {
      const long long t = (chunk * kLayerMaxLines + u) * 64 + lane;
      gf[u] = 0.0; elow[u] = 0.0; wavn[u] = 0.0; iso[u] = 0;
      if (t < ncand) {
        if (cand) { const CandLine c = cand[t]; gf[u] = c.gf; elow[u] = c.elow; wavn[u] = c.wavn; iso[u] = c.iso; }
        else if (L.inrange[t]) { gf[u] = L.gf[t]; elow[u] = L.elow[t]; wavn[u] = L.wavn[t]; iso[u] = L.iso[t]; }
      }
    }
#pragma unroll
    for (int q = 0; q < kLayerMaxGroup; q++) {
      const int r = r0 + q;
      if (r < r1) {
        const double ct = s_ct[q];
#pragma unroll
        for (int u = 0; u < kLayerMaxLines; u++) {
          const double s = gf[u] * exp_neg(ct * elow[u], s_e2) * (1 - exp_neg(ct * wavn[u], s_e2));
          const double k = s * s_f[q * niso + iso[u]];
          if (nmx == 1) best[q] = fmax(best[q], k);
          else if (k > 0) {       // per-molecule maxima (extinction.c:406-407, permol)
            unsigned long long *slot = &kmax_bits[(long long)r * nmx + iso_mx[iso[u]]];
            const unsigned long long kb = (unsigned long long)__double_as_longlong(k);
            if (kb > *slot) atomicMax(slot, kb);
          }
        }
      }
    }
  }
  if (nmx == 1) {
#pragma unroll
    for (int q = 0; q < kLayerMaxGroup; q++) {
      const double m = wave_max(best[q]);
      if (lane == 0) s_best[wv][q] = m;
    }
    __syncthreads();
    if ((int)threadIdx.x < r1 - r0) {
      double m = s_best[0][threadIdx.x];
#pragma unroll
      for (int w = 1; w < kLayerMaxBlock; w++) m = fmax(m, s_best[w][threadIdx.x]);
      if (m > 0) atomicMax(&kmax_bits[r0 + threadIdx.x], (unsigned long long)__double_as_longlong(m));
    }
  }
}

// ---------------------------------------------------------------------------
// line ranges ("waves") of the walk
// ---------------------------------------------------------------------------
// Range w of isotope block b covers the groups [g0, g1) = kWalkGroups consecutive co-added
// groups (fewer at the end of the block).  Groups are sorted by descending wavenumber, so a
// range's groups sit in the cells gidiv[g1-1] .. gidiv[g0] and can reach the bins
//     blo = gidiv[g1-1] - Rc  ..  bhi = gidiv[g0] + Rc + 1          (clipped to the shard)
// when every profile of the step is at most Rc whole cells wide on either side.  The range
// writes one partial record (64 lanes = layers) per bin of that interval, at record index
// off[w] + (j - blo); off is the running sum of the interval lengths.
struct WalkPlan {
  int nwaves, ngw;                  // ranges, groups per range
  const int32_t *wbase;             // [niso + 1] first range of every isotope block
  int32_t *blo, *bhi;               // [nwaves] (bhi < blo: the range cannot reach the shard)
  int64_t *off;                     // [nwaves + 1]
  int32_t *binw;                    // [niso][bins of the shard][2]: first and one-past-last range of the
                                    // block that touches the bin (null: the combine searches blo/bhi itself)
  int32_t *binrec;                  // [niso][bins of the shard][64]: the bin's record in each of its first 64 ranges
                                    // (off[w] + j - blo[w]); null: none.  One look-up instead of two dependent ones in k_ray_tail
};

__device__ __forceinline__ int walk_block_of(const int32_t *wbase, int niso, int w)
{
  int b = 0;
  while (b + 1 < niso && w >= wbase[b + 1]) b++;
  return b;
}

__global__ __launch_bounds__(256)
void k_wave_plan(WalkPlan P, int niso, const int32_t *__restrict__ gblock, const int32_t *__restrict__ gidiv,
                 int Rc, long long lo, long long hi)
{
  __shared__ long long s_scan[256];
  __shared__ long long s_carry;
  if (threadIdx.x == 0) { s_carry = 0; P.off[0] = 0; }
  __syncthreads();
  for (int w0 = 0; w0 < P.nwaves; w0 += 256) {
    const int w = w0 + threadIdx.x;
    long long len = 0;
    if (w < P.nwaves) {
      const int b = walk_block_of(P.wbase, niso, w);
      const int g0 = gblock[b] + (w - P.wbase[b]) * P.ngw, g1 = min(g0 + P.ngw, gblock[b + 1]);
      long long bl = (long long)gidiv[g1 - 1] - Rc, bh = (long long)gidiv[g0] + Rc + 1;
      if (bl < lo) bl = lo;
      if (bh > hi - 1) bh = hi - 1;
      P.blo[w] = (int32_t)bl; P.bhi[w] = (int32_t)bh;
      len = bh >= bl ? bh - bl + 1 : 0;
    }
    s_scan[threadIdx.x] = len;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
      const long long v = threadIdx.x >= o ? s_scan[threadIdx.x - o] : 0;
      __syncthreads();
      s_scan[threadIdx.x] += v;
      __syncthreads();
    }
    if (w < P.nwaves) P.off[w + 1] = s_carry + s_scan[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 255) s_carry += s_scan[255];
    __syncthreads();
  }
}

// The ranges of an isotope block that touch a bin, for the combine: blo and bhi descend with the
// range index, so they are one run [wa, wz).  Found once per plan here -- the combine used to do
// these two bisections itself, ~28 dependent loads at the head of every wave.
__device__ __forceinline__ void ranges_of_bin(const WalkPlan &P, int b, long long j, int &wa, int &wz)
{
  const int w0 = P.wbase[b], w1 = P.wbase[b + 1];
  int a = w0, z = w1;                      // first w with blo[w] <= j
  while (a < z) { const int m = (a + z) >> 1; if (P.blo[m] <= j) z = m; else a = m + 1; }
  wa = a;
  z = w1;                                  // first w >= wa with bhi[w] < j
  while (a < z) { const int m = (a + z) >> 1; if (P.bhi[m] < j) z = m; else a = m + 1; }
  wz = a;
}

__global__ __launch_bounds__(256)
void k_bin_ranges(WalkPlan P, int niso, long long lo, long long nsh)
{
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= nsh * niso) return;
  const int b = (int)(t / nsh); const long long j = lo + t % nsh;
  int wa = 0, wz = 0;
  if (P.wbase[b] != P.wbase[b + 1]) ranges_of_bin(P, b, j, wa, wz);
  P.binw[2 * t] = wa; P.binw[2 * t + 1] = wz;
  if (P.binrec)
    for (int u = 0; u < min(wz - wa, 64); u++) P.binrec[t * 64 + u] = (int32_t)(P.off[wa + u] + (j - P.blo[wa + u]));
}

// ---------------------------------------------------------------------------
// the walk
// ---------------------------------------------------------------------------
// One 32-byte record per line, read with ONE scalar load per line.
//   meta bit 0: the line anchors a co-added group (extinction.c:449-462), bit 1: it is the
//   group's last member, bit 2: the exponential is rebased on this line (below);
//   bits 3..: iown % osamp of the group's anchor; cell = iown / osamp.
// The array carries one record of padding: the walk requests record n+1 while it works on n.
struct alignas(32) WalkLine { double wavn, elow, gf; int32_t meta, cell; };

constexpr int kWalkLayers = 64;      // layers per step = lanes
constexpr int kWalkSegs = 16;        // isotope blocks whose reaching ranges a launch can address separately
// exp(ct*wavn) along a wavenumber-sorted range: with a base point t0 = fl(ct*w0), c0 = -t0 and
// E0 = exp(t0), x = fma(ct, wavn, c0) is the EXACT difference ct*wavn - t0 (one rounding) and
// exp(ct*wavn) = E0 * P5(x) to 5e-18 while 0 <= x <= 2^-8.  Base points are a property of the
// LINE LIST (meta bit 2, set by trx_create: a range's first line, then every line more than
// kRebaseSpan below the last base), so that every rounding depends on the list and the lane's
// layer alone -- not on which other layers share a step nor on which groups a step can skip.
// c*span/T <= 2^-8 needs T >= kWalkMinTemp (colder runs take the two-kernel path).
constexpr double kRebaseSpan = 0.03125;     // cm-1
constexpr double kWalkMinTemp = 1.4387752 * kRebaseSpan * 256.0 * 1.05;

constexpr int kWalkMaxFrame = 16;    // bins of the widest frame (k_line_walk<16>)

// The walk's copy of the Voigt table ("tabW", built by trx_create): per profile `osamp` rows, row
// `ph` holding the entries q = osamp*kk + ph, kk = 0..K-1, between zeros (walk_row_layout,
// trx_kernels.hip.h: rows are whole 64-byte lines).  The bins of a frame sit a whole cell apart: they are
// CONSECUTIVE entries of one row, one or two wide loads per lane instead of a load per bin, inside
// one cache line for frames of up to 8 bins, and where a profile does not reach the entries are
// zero by position (kTabPad zeros around the whole).
struct alignas(16) WalkProfile {
  uint32_t centre4;                  // byte offset of (row 0, kk = ps / osamp)
  int32_t rowb;                      // bytes per row (walk_row_layout)
  int32_t psr;                       // ps % osamp
  int32_t ps;                        // half-width in table samples
};
// frames of at least this many bins read tabW (the narrower ones: a load per bin from the table itself)
constexpr int kWalkRowsFrom = 4;

// trx_create builds the records on the device from the line and group arrays it has just uploaded
__global__ __launch_bounds__(256)
void k_walk_records(long long n, const double *__restrict__ wavn, const double *__restrict__ elow, const double *__restrict__ gf,
                    const int32_t *__restrict__ lgroup, const int32_t *__restrict__ giown, int osamp, WalkLine *__restrict__ out)
{
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i > n) return;
  WalkLine R{0.0, 0.0, 0.0, 0, 0};                         // i == n: the padding record
  if (i < n) {
    R.wavn = wavn[i]; R.elow = elow[i]; R.gf = gf[i];
    const int g = lgroup[i];
    if (g >= 0) { const int io = giown[g]; R.meta = 1 | ((io % osamp) << 3); R.cell = io / osamp; }
  }
  out[i] = R;
}

// one lane per line range: last members of the groups, base points of the rebased exponential
__global__ __launch_bounds__(64)
void k_walk_marks(int nwaves, int ngw, int niso, const int32_t *__restrict__ wbase, const int32_t *__restrict__ gblock,
                  const int32_t *__restrict__ gfirst, const int32_t *__restrict__ gcount, WalkLine *__restrict__ out,
                  double *__restrict__ linebase /* [nlines]: wavenumber of every line's base point (k_line_walk_lanes) */)
{
  const int w = blockIdx.x * 64 + threadIdx.x;
  if (w >= nwaves) return;
  int b = 0;
  while (b + 1 < niso && w >= wbase[b + 1]) b++;
  const int g0 = gblock[b] + (w - wbase[b]) * ngw, g1 = min(g0 + ngw, gblock[b + 1]);
  for (int g = g0; g < g1; g++) {
    const int la = gfirst[g], lz = la + gcount[g] - 1;
    out[lz].meta |= 2;
    for (int l = la + 1; l <= lz; l++) out[l].cell = out[la].cell;      // (members carry their group's cell: k_line_walk_lanes cuts its blocks where the cell changes)
  }
  const int l0 = gfirst[g0], l1 = gfirst[g1 - 1] + gcount[g1 - 1];
  double w0 = HUGE_VAL;
  for (int l = l0; l < l1; l++) {
    const double wv = out[l].wavn;
    if (l == l0 || w0 - wv > kRebaseSpan) { out[l].meta |= 4; w0 = wv; }
    linebase[l] = w0;
  }
}

// what a wave needs to know about its range before it can ask for anything else (k_line_walk_lanes: one load
// instead of a chain of five through wbase, gblock, gfirst, gcount and the first line's record)
struct alignas(32) RangeInfo { int32_t l0, l1, b, cell0; double wavn0; int32_t cell1, pad; };      // cell0 / cell1: of the first / last group

__global__ __launch_bounds__(256)
void k_range_info(int nwaves, int ngw, int niso, const int32_t *__restrict__ wbase, const int32_t *__restrict__ gblock,
                  const int32_t *__restrict__ gfirst, const int32_t *__restrict__ gcount, const WalkLine *__restrict__ lines,
                  RangeInfo *__restrict__ out)
{
  const int w = blockIdx.x * 256 + threadIdx.x;
  if (w >= nwaves) return;
  int b = 0;
  while (b + 1 < niso && w >= wbase[b + 1]) b++;
  const int g0 = gblock[b] + (w - wbase[b]) * ngw, g1 = min(g0 + ngw, gblock[b + 1]);
  RangeInfo R{};
  R.l0 = gfirst[g0]; R.l1 = gfirst[g1 - 1] + gcount[g1 - 1]; R.b = b;
  R.cell0 = lines[R.l0].cell; R.wavn0 = lines[R.l0].wavn; R.cell1 = lines[gfirst[g1 - 1]].cell;
  out[w] = R;
}

// Workgroups go to the 8 XCDs in turn (block i runs on XCD i % 8), each XCD with an L2 of its own.  The
// ranges are in wavenumber order and the table rows a range gathers depend on its wavenumber (Doppler index)
// and its layers: with block i working on ranges xcd_block(i), an XCD takes ONE contiguous eighth of the
// launch in order, so the rows its L2 holds at any time are those of a narrow band -- an eighth of what the
// launch order asks every L2 to hold.
__device__ __forceinline__ int xcd_block(int i, int n)
{
  const int x = i & 7, k = i >> 3, q = n >> 3, r = n & 7;     // XCD x has q + (x < r) blocks
  return x * q + min(x, r) + k;
}

struct WalkArgs {
  const WalkLine *lines;
  const RangeInfo *rinfo;           // [nwaves]
  const int32_t *gfirst, *gcount, *gblock;
  WalkPlan P;
  int niso, nlor, ndop, osamp;
  long long lo, hi;                 // shard [lo, hi)
  int r_top, nc;
  LayerDev Y; const double *wcut;   // [layer][iso]
  const double *kmax;               // [layer][nmx] strongest single line
  double ethresh;
  int nmx; const int32_t *iso_mx; int permol;
  const int *sticky_idop;           // [layer][iso]
  const double *dthr;               // [ndop + 1] steps of the nearest-Doppler-index function
  const double *e2tab;              // [64]
  const int32_t *psize; const long long *poff;
  long long zero_index;             // index (in `table`) where kWalkMaxFrame*osamp zeros begin
  const float *table;               // the Voigt table, followed by kWalkMaxFrame cells of zeros (k_table_padded)
  const float *tabw; const WalkProfile *walkprof;   // the walk's row copy and its descriptors [ndop][nlor]
  int xcd_map;                                      // blocks -> ranges by xcd_block (0: in launch order)
  const float *tabw32; const uint32_t *wp32;        // its compact 32-byte rows (frames of 8 bins), [phase][profile][8], and the profiles' byte offsets inside a phase's slab [ndop][nlor] (null: none)
  unsigned slab32;                                  // bytes of a slab (profiles with such rows x 32)
  double *part;                     // [records][64]
  unsigned long long *counters;     // [layer][3] {bins, evaluated, skipped} or null
  const int *flags; const int *last; int eager;
  // the ranges that can reach the shard: per isotope block one run of consecutive ranges
  // (nseg = 0: every range is launched).  Wave L of the grid walks range seg_w0[s] + L - seg_cum[s].
  int nseg; int seg_w0[kWalkSegs]; int seg_cum[kWalkSegs + 1];
};

__device__ __forceinline__ double exp_small(double x, double c24)       // (c24 = 1/24 from a vector register: exp_neg's note)
{
  double p = __builtin_fma(0x1.1111111111111p-7, x, c24);
  p = __builtin_fma(p, x, 0x1.5555555555555p-3);   // 1/6
  p = __builtin_fma(p, x, 0.5);
  p = __builtin_fma(p, x, 1.0);
  p = __builtin_fma(p, x, 1.0);
  return p;
}
__device__ __forceinline__ double exp_small(double x)
{
  double p = 0x1.1111111111111p-7;                 // 1/120
  p = __builtin_fma(p, x, 0x1.5555555555555p-5);   // 1/24
  p = __builtin_fma(p, x, 0x1.5555555555555p-3);   // 1/6
  p = __builtin_fma(p, x, 0.5);
  p = __builtin_fma(p, x, 1.0);
  p = __builtin_fma(p, x, 1.0);
  return p;
}

constexpr int kWalkWaves = 4;        // independent waves per workgroup (a CU holds few workgroups)

// LPL = 2 (steps of at most 32 layers, row form): TWO lanes per layer, each with half of the frame --
// lane = 2*layer + half: the two halves of a frame are neighbouring lanes reading one cache line.  The frame's row segment is then fetched by ONE load instruction per
// group instead of two (the wide frames are bound by their table loads: a further 16-byte load per
// lane and group costs 35 us of 153 even when it hits the same cache line); the strength
// arithmetic is done twice, on lanes that would be idle.
template <int NB, bool PROF, int LPL = 1>
__global__ __launch_bounds__(64 * kWalkWaves)
void k_line_walk(WalkArgs A)
{
  constexpr int Rc = NB / 2 - 1;
  constexpr int NS = NB / LPL;                             // slots (bins of the frame) per lane
  constexpr bool ROWS = !PROF && NB >= kWalkRowsFrom;     // (counting runs keep the per-bin form: they count per bin)
  static_assert(LPL == 1 || (LPL == 2 && ROWS && NS >= 4), "lane pairs: row form, frames of 8+ bins");
  if (!A.eager && A.flags[0] == 0) return;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int w = (A.xcd_map ? xcd_block(blockIdx.x, gridDim.x) : (int)blockIdx.x) * kWalkWaves + wv;      // wave-uniform
  if (A.nseg > 0) {                                       // only the ranges that can reach the shard were launched
    if (w >= A.seg_cum[A.nseg]) return;
    int sgm = 0;
    while (sgm + 1 < A.nseg && w >= A.seg_cum[sgm + 1]) sgm++;
    w = A.seg_w0[sgm] + (w - A.seg_cum[sgm]);
  }
  if (w >= A.P.nwaves) return;
  const int lane = threadIdx.x & 63;
  const int blo = A.P.blo[w], bhi = A.P.bhi[w];
  if (bhi < blo) return;                                  // nothing of this range reaches the shard
  if (A.last) {   // every ray of the range's bins has stopped (tau.c:277-287): nobody reads them
    bool open = false;
    for (int j = blo + lane; j <= bhi; j += 64) open |= A.last[j - A.lo] < 0;
    if (__ballot(open) == 0ull) return;
  }
  __shared__ double s_thr_w[kWalkWaves][kMaxDop + 1];     // per wave: waves come and go on their own
  __shared__ double s_e2_w[kWalkWaves][64];
  double *s_thr = s_thr_w[wv], *s_e2 = s_e2_w[wv];
  for (int i = lane; i <= A.ndop; i += 64) s_thr[i] = A.dthr[i];
  s_e2[lane] = A.e2tab[lane];
  __builtin_amdgcn_wave_barrier();

  const RangeInfo RI = A.rinfo[w];                          // (one load: trx_create made the record)
  const int b = __builtin_amdgcn_readfirstlane(RI.b), l0 = __builtin_amdgcn_readfirstlane(RI.l0), l1 = __builtin_amdgcn_readfirstlane(RI.l1);
  const long long rec0 = A.P.off[w];

  // ---- this lane's layer (and, with lane pairs, its part of the frame: slots part*NS ...)
  const int li = LPL == 1 ? lane : (lane >> 1), part = LPL == 1 ? 0 : (lane & 1);
  const bool valid = li < A.nc;
  const int r = A.r_top - (valid ? li : 0), ri = r * A.niso + b;
  const int mx = A.nmx == 1 ? 0 : A.iso_mx[b];
  const double ct = valid ? A.Y.negc_over_t[r] : 0.0;      // idle lanes: strength 0
  const double f = A.Y.strength_f[ri], dens = A.permol ? 1.0 : A.Y.density[ri];
  const double lim = A.ethresh * A.kmax[(long long)r * A.nmx + mx];
  const double wc = A.wcut[ri], ad = A.Y.alphad[ri];
  const int il = A.Y.ilor[ri];
  const int idst = A.sticky_idop[ri];
  const int ps_st = A.psize[idst * A.nlor + il];
  const unsigned vo_st = (unsigned)(A.poff[idst * A.nlor + il] + ps_st);     // table index of the sticky profile's centre
  // widest profile any lane of this step can use for this isotope: wave-uniform bound for the bin
  // loop, split into whole cells and the rest (psm_s = psq*osamp + psr)
  const int psm_s = __builtin_amdgcn_readfirstlane(wave_max_i(valid ? A.Y.psmax[ri] : 0));
  const int psq = psm_s / A.osamp, psr = psm_s - psq * A.osamp;
  const int lo32 = (int)A.lo, hi32 = (int)A.hi;            // (the grid has < 2^31 bins: trx_create)
  // every bin this range can reach lies inside the shard: no clipping of the slot masks
  const bool interior = __builtin_amdgcn_readfirstlane(RI.cell1) - Rc >= lo32 && __builtin_amdgcn_readfirstlane(RI.cell0) + Rc + 1 < hi32;

  // Doppler index of the first anchor, then followed downwards (wavenumbers descend => it never rises)
  int lo_i = index_from(s_thr, ad * RI.wavn0, A.Y.idop0[ri]);
  double thr_lo = s_thr[lo_i];
  int ps_cur = A.psize[lo_i * A.nlor + il];
  unsigned vo_cur = 4u * (unsigned)(A.poff[lo_i * A.nlor + il] + ps_cur);   // byte offset of the profile centre
  const unsigned vo_st8 = 4u * vo_st;
  WalkProfile wp_cur{}, wp_st{};
  if (ROWS) { wp_cur = A.walkprof[lo_i * A.nlor + il]; wp_st = A.walkprof[idst * A.nlor + il]; }

  double acc[NS];
#pragma unroll
  for (int k = 0; k < NS; k++) acc[k] = 0.0;
  int jc = __builtin_amdgcn_readfirstlane(RI.cell0);           // frame: acc[k] <-> bin jc - Rc + k
  unsigned long long nb = 0, nev = 0, nsk = 0;

  auto flush = [&](int k, double v, bool mine = true) {    // bin of this lane's slot k leaves the frame
    const int j = jc - Rc + part * NS + k;
    if (valid && mine && j >= blo && j <= bhi) A.part[(rec0 + (j - blo)) * kWalkLayers + li] = v;     // (idle lanes: nobody reads their entries)
  };

  // bins of the interval that no frame position covers (a jump over empty or skipped cells, the
  // tail below the last evaluated group) still get their record: zeros
  auto fill_zero = [&](int ja, int jb) {                   // bins ja..jb inclusive
    ja = max(ja, blo); jb = min(jb, bhi);
    if (valid && part == 0) for (int j = ja; j <= jb; j++) A.part[(rec0 + (j - blo)) * kWalkLayers + li] = 0.0;
  };

  double pk = 0.0;                   // strength of the group so far (0 between groups)
  double c24 = 0x1.5555555555555p-5;                       // 1/24, kept in a vector register (exp_neg's note)
  asm volatile("" : "+v"(c24));
  int cell = jc, imod = 0;
  unsigned cand = 0u;                // slots of the frame (once it sits on the group's cell) that some
                                     // layer of this step can reach from the current group; 0: skip it
  // table entries requested for the previous group, added when the next group is complete (or
  // the frame moves): a gather's round trip then overlaps a line's worth of arithmetic
  // (every slot takes part, a slot the line does not reach with a table value of 0: adding
  // kk * 0 leaves the sum as it is, and eight unconditional multiply-adds are fewer instructions
  // than eight scalar tests around six of them)
  float pv_p[NS]; double kk_p = 0.0;                       // (before the first group: weight 0 on zeros -- settle() adds nothing, and needs no "is something pending" select around its multiply-adds)
#pragma unroll
  for (int k = 0; k < NS; k++) pv_p[k] = 0.f;
  auto settle = [&]() {
#pragma unroll
    for (int k = 0; k < NS; k++) acc[k] = __builtin_fma(kk_p, (double)pv_p[k], acc[k]);     // :507
  };
  // Slot k reads the table at (profile centre) + d, d = (k - Rc)*osamp - imod.  The part of d that
  // does not depend on the group goes into a per-slot base pointer (wave-uniform, set up once),
  // the rest into ONE per-lane offset per group, kept >= 0 by moving a whole cell into the
  // pointer: entry = tab_k[k] + (vo8 - 8*imod + 8*osamp).  A lane the slot does not reach reads
  // zero padding instead -- at the same offset vzero for every slot: the widened table ends in
  // kWalkMaxFrame cells of zeros, slot k lands k cells into them.  No branch, no mask, one load
  // per slot.
  const char *tab_k[NB];
#pragma unroll
  for (int k = 0; k < NB; k++) tab_k[k] = (const char *)A.table + 4LL * (k - Rc - 1) * A.osamp;
  const unsigned vzero = 4u * (unsigned)(A.zero_index + (long long)(Rc + 1) * A.osamp);
  const unsigned cell8 = 4u * (unsigned)A.osamp;
  const char *tabw_base = (const char *)A.tabw - 4 * (Rc + 1);     // (slot 0 of a row segment; the copy has a front pad)

  // line records through the scalar cache: the address is wave-uniform and the data constant
  typedef const __attribute__((address_space(4))) double *ScalarF64;
  typedef const __attribute__((address_space(4))) int32_t *ScalarI32;
  ScalarF64 rp = (ScalarF64)(A.lines + l0);
  double L_wavn = rp[0], L_elow = rp[1], L_gf = rp[2];
  int L_meta = ((ScalarI32)rp)[6] | 4, L_cell = ((ScalarI32)rp)[7];    // (a range starts on a base point)
  double c0 = 0.0, E0 = 1.0, wav_a = L_wavn;
  for (int left = l1 - l0; left > 0; left--) {
    rp += 4;                                               // next record, requested a line ahead
    const double N_wavn = rp[0], N_elow = rp[1], N_gf = rp[2];
    const int N_meta = ((ScalarI32)rp)[6], N_cell = ((ScalarI32)rp)[7];
    if (L_meta & 4) {                                      // base point of the rebased exponential
      const double t0 = ct * L_wavn;
      E0 = exp_neg(t0, s_e2, c24); c0 = -t0;
    }
    if (L_meta & 1) {
      // slots some layer of this step can reach from this group: |(k - Rc)*osamp - imod| <= psm_s
      // and the bin inside the shard.  All wave-uniform integer arithmetic.
      cell = L_cell; imod = L_meta >> 3; wav_a = L_wavn;
      if (NB == 2) cand = (imod <= psm_s ? 1u : 0u) | (A.osamp - imod <= psm_s ? 2u : 0u);
      else {
        const int klo = Rc - psq + (imod > psr ? 1 : 0), khi = Rc + psq + (imod + psr >= A.osamp ? 1 : 0);
        cand = ((2u << khi) - 1u) & ~((1u << klo) - 1u);   // (0 <= klo <= khi < NB by the choice of NB)
      }
      if (!interior) {
#pragma unroll
        for (int k = 0; k < NB; k++) { const int j = cell - Rc + k; if (j < lo32 || j >= hi32) cand &= ~(1u << k); }
      }
      if (PROF) cand |= 0x80000000u;                       // (counting runs evaluate every group)
    }
    if (cand) {
      // ---- strength of the line in every layer
      const double e1 = exp_neg(ct * L_elow, s_e2, c24);
      const double q = __builtin_fma(-E0, exp_small(__builtin_fma(ct, L_wavn, c0), c24), 1.0);
      pk += L_gf * e1 * q;
      if (L_meta & 2) {
        // ---- the group is complete: threshold, density, profile, bins (extinction.c:464-509)
        const double pkf = pk * f;
        pk = 0.0;
        const bool below = pkf < lim;                          // :467
        if (PROF && cell >= lo32 && cell < hi32 && valid) { if (below) nsk++; else nev++; }
        const double kk = pkf * dens;                          // :472-473
        // nearest Doppler-width index: own one while alphad*wn/alphal >= 0.1, else the sticky one (:480-483)
        const double v = ad * wav_a;
        if (__any(v < thr_lo)) {
          do {
            if (v < thr_lo) {
              lo_i--; thr_lo = s_thr[lo_i];
              ps_cur = A.psize[lo_i * A.nlor + il];
              vo_cur = 4u * (unsigned)(A.poff[lo_i * A.nlor + il] + ps_cur);
              if (ROWS) wp_cur = A.walkprof[lo_i * A.nlor + il];
            }
          } while (__any(v < thr_lo));
        }
        const bool own = wav_a >= wc;
        const int ps = own ? ps_cur : ps_st;
        const unsigned vo8 = own ? vo_cur : vo_st8;
        const bool act = valid && !below;
        settle();
        // ---- move the frame down to the group's cell
        if (cell != jc) {
          int sh = jc - cell;
          if (sh >= NB) {
#pragma unroll
            for (int k = 0; k < NS; k++) { flush(k, acc[k]); acc[k] = 0.0; }
            fill_zero(cell + Rc + 2, jc - Rc - 1);
            jc = cell;
          } else {
            for (; sh > 0; sh--) {
              flush(NS - 1, acc[NS - 1], part == LPL - 1);            // the frame's last bin
              double carry = 0.0;                                      // the lower part's last bin moves up a lane
              if (LPL == 2) carry = dpp_f64<0xB1>(acc[NS - 1]);       // quad_perm [1,0,3,2]: the pair's other lane
#pragma unroll
              for (int k = NS - 1; k > 0; k--) acc[k] = acc[k - 1];
              acc[0] = (LPL == 2 && part == 1) ? carry : 0.0;
              jc--;
            }
          }
        }
        // ---- bins: slot k is bin jc - Rc + k at fine distance d = (k - Rc)*osamp - imod from the line
        // (a bin outside the shard may be accumulated too: it never leaves the frame, see flush)
        if constexpr (ROWS) {
          // one row of the lane's profile holds all the frame's bins: row ph = (ps - imod) mod osamp,
          // entry kk = (ps - imod) div osamp for the centre slot -- from the profile's ps = psq*osamp + psr
          // without a division: psr >= imod ? (psr - imod, psq) : (psr - imod + osamp, psq - 1).
          // A lane that sits out adds 0 * (whatever it reads).
          kk_p = act ? kk : 0.0;
          const unsigned c4 = own ? wp_cur.centre4 : wp_st.centre4;
          const int rb = own ? wp_cur.rowb : wp_st.rowb;
          const int d = (own ? wp_cur.psr : wp_st.psr) - imod, sgn = d >> 31;          // sgn = -1: borrowed a cell
          const unsigned at = c4 + (unsigned)(d + (sgn & A.osamp)) * (unsigned)rb + (unsigned)((sgn + 1) << 2);     // (unsigned product: a profile's rows may pass 2 GB, the copy as a whole stays below 4 GB)
          struct alignas(4) Row { float v[NS]; } row;
          __builtin_memcpy(&row, tabw_base + at + 4u * (unsigned)(NS * part), sizeof row);       // 4-byte aligned wide loads
#pragma unroll
          for (int k = 0; k < NS; k++) pv_p[k] = row.v[k];
        } else {
          kk_p = kk;
          const int ps_act = act ? ps : -1;                       // a lane that sits out reaches no slot
          const unsigned vbase = vo8 + cell8 - 4u * (unsigned)imod;
#pragma unroll
          for (int k = 0; k < NB; k++) {
            // |d| with the sign known per slot: d <= 0 up to the centre slot, > 0 beyond (imod < osamp)
            const int dist = k <= Rc ? (Rc - k) * A.osamp + imod : (k - Rc) * A.osamp - imod;      // wave-uniform
            const bool ok = dist <= ps_act;
            pv_p[k] = *(const float *)(tab_k[k] + (ok ? vbase : vzero));
            if (PROF && ok && cell - Rc + k >= lo32 && cell - Rc + k < hi32) nb++;
          }
        }
      }
    }
    L_wavn = N_wavn; L_elow = N_elow; L_gf = N_gf; L_meta = N_meta; L_cell = N_cell;
  }
  settle();
#pragma unroll
  for (int k = 0; k < NS; k++) flush(k, acc[k]);
  fill_zero(blo, jc - Rc - 1);
  if (PROF && valid && A.counters) {
    if (nb)  atomicAdd(&A.counters[(long long)r * 3 + 0], nb);
    if (nev) atomicAdd(&A.counters[(long long)r * 3 + 1], nev);
    if (nsk) atomicAdd(&A.counters[(long long)r * 3 + 2], nsk);
  }
}

// ---------------------------------------------------------------------------
// the walk for steps of few layers: several line ranges per wavefront
// ---------------------------------------------------------------------------
// k_line_walk costs what its instructions cost whatever the number of busy lanes: the demo's second
// step (17 layers) pays 140 us for 27 % of the lanes, a step of 3 layers the same.  Here a wave
// takes S = 64 / nc CONSECUTIVE ranges at once, lane = slot * nc + layer: every slot walks its own
// range exactly as a wave of k_line_walk would -- same lines, same order, same frame, same partial
// records (one per (range, bin), written by the slot that owns the range), so the sums and the
// combine are untouched and the results are the same bits -- but an instruction now serves S lines.
// What k_line_walk keeps wave-uniform in scalar registers (the line's record, its cell, the frame's
// position, which bins it can reach) is per lane here: the records come through the vector
// memory path (the lanes of a slot ask for the same 32 bytes: one request), the frame moves under
// the lanes' own masks.  Row form only (frames of 4+ bins; a step of narrower layers runs its
// 4-bin frame: the same values from the row copy), production only (counting runs: k_line_walk).
template <int NB>
__global__ __launch_bounds__(64 * kWalkWaves)
void k_line_walk_packed(WalkArgs A, int S)
{
  constexpr int Rc = NB / 2 - 1;
  if (!A.eager && A.flags[0] == 0) return;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int W = (A.xcd_map ? xcd_block(blockIdx.x, gridDim.x) : (int)blockIdx.x) * kWalkWaves + wv;     // wave-uniform: ranges W*S .. W*S + S - 1 of the launch
  const int lane = threadIdx.x & 63;
  const int slot = lane / A.nc, li = lane - slot * A.nc;
  int widx = W * S + slot;                                // index into the launched ranges
  bool valid = slot < S;
  int w = widx;
  if (A.nseg > 0) {                                       // only the ranges that can reach the shard were launched
    if (widx >= A.seg_cum[A.nseg]) valid = false;
    int w0 = A.seg_w0[0], c0 = A.seg_cum[0];
#pragma unroll
    for (int sg = 1; sg < kWalkSegs; sg++)
      if (sg < A.nseg && widx >= A.seg_cum[sg]) { w0 = A.seg_w0[sg]; c0 = A.seg_cum[sg]; }
    w = w0 + (widx - c0);
  }
  if (w >= A.P.nwaves) valid = false;
  if (!valid) w = 0;
  int blo = A.P.blo[w], bhi = A.P.bhi[w];
  if (bhi < blo) valid = false;                           // nothing of this range reaches the shard
  if (A.last) {   // every ray of the range's bins has stopped (tau.c:277-287): nobody reads them
    bool open = false;
    if (valid) for (int j = blo + li; j <= bhi; j += A.nc) open |= A.last[j - A.lo] < 0;
    bool slot_open = false;
    for (int sl = 0; sl < S; sl++) {                      // (S <= 32 ballots: any lane of the slot saw an open ray)
      const unsigned long long m = __ballot(open && slot == sl);
      if (slot == sl) slot_open = m != 0ull;
    }
    valid = valid && slot_open;
  }
  if (__ballot(valid) == 0ull) return;
  __shared__ double s_thr_w[kWalkWaves][kMaxDop + 1];
  __shared__ double s_e2_w[kWalkWaves][64];
  double *s_thr = s_thr_w[wv], *s_e2 = s_e2_w[wv];
  for (int i = lane; i <= A.ndop; i += 64) s_thr[i] = A.dthr[i];
  s_e2[lane] = A.e2tab[lane];
  __builtin_amdgcn_wave_barrier();

  const int b = walk_block_of(A.P.wbase, A.niso, w);
  const int g0 = A.gblock[b] + (w - A.P.wbase[b]) * A.P.ngw, g1 = min(g0 + A.P.ngw, A.gblock[b + 1]);
  const int l0 = A.gfirst[g0], l1 = A.gfirst[g1 - 1] + A.gcount[g1 - 1];
  const long long rec0 = A.P.off[w];

  const int r = A.r_top - (valid ? li : 0), ri = r * A.niso + b;
  const int mx = A.nmx == 1 ? 0 : A.iso_mx[b];
  const double ct = valid ? A.Y.negc_over_t[r] : 0.0;      // idle lanes: strength 0
  const double f = A.Y.strength_f[ri], dens = A.permol ? 1.0 : A.Y.density[ri];
  const double lim = A.ethresh * A.kmax[(long long)r * A.nmx + mx];
  const double wc = A.wcut[ri], ad = A.Y.alphad[ri];
  const int il = A.Y.ilor[ri];
  const int idst = A.sticky_idop[ri];
  // this slot's lines, requested kAhead lines ahead (vector loads: the slot's lanes share the address; a
  // packed step has few waves, a wave hides its own memory latency): ring Q, line t in Q[t % kAhead]
  constexpr int kAhead = 4;
  const WalkLine *lp = A.lines + l0;
  const int nlines_slot = valid ? l1 - l0 : 0;
  WalkLine Q[kAhead];
#pragma unroll
  for (int u = 0; u < kAhead; u++) Q[u] = lp[min(u, max(nlines_slot - 1, 0))];
  WalkLine L = Q[0];
  L.meta |= 4;                                              // (a range starts on a base point)

  // Doppler index of the first anchor, then followed downwards (wavenumbers descend => it never rises)
  int lo_i = index_from(s_thr, ad * L.wavn, A.Y.idop0[ri]);
  double thr_lo = s_thr[lo_i];
  WalkProfile wp_cur = A.walkprof[lo_i * A.nlor + il];
  const WalkProfile wp_st = A.walkprof[idst * A.nlor + il];

  double acc[NB];
#pragma unroll
  for (int k = 0; k < NB; k++) acc[k] = 0.0;
  int jc = L.cell;                                          // frame: acc[k] <-> bin jc - Rc + k
  auto flush = [&](int k, double v, bool on) {              // bin of slot k leaves the frame
    const int j = jc - Rc + k;
    if (on && j >= blo && j <= bhi) A.part[(rec0 + (j - blo)) * kWalkLayers + li] = v;
  };
  // Rows of the table: a group's row segment depends on its anchor line alone (profile from the anchor's
  // wavenumber, phase from its fine-grid index), not on the strength -- so it is requested when the anchor is
  // TWO lines ahead in the ring (stage "ahead"), into the row buffer of that ring place, and used when the
  // group completes: the wait for the table, 63 % of this kernel's wave-cycles with a request per completed
  // group, is gone from the chain.  grow: the row of the group in progress (taken over at its anchor line).
  struct alignas(4) Row { float v[NB]; };
  Row R[kAhead], grow;
#pragma unroll
  for (int k = 0; k < NB; k++) grow.v[k] = 0.f;
  const char *tabw_base = (const char *)A.tabw - 4 * (Rc + 1);
  auto ahead = [&](const WalkLine &La, bool on_a, Row &dst) {
    if (on_a && (La.meta & 1)) {
      // nearest Doppler-width index: own one while alphad*wn/alphal >= 0.1, else the sticky one (:480-483)
      const double v = ad * La.wavn;
      while (v < thr_lo) { lo_i--; thr_lo = s_thr[lo_i]; wp_cur = A.walkprof[lo_i * A.nlor + il]; }
      const bool own = La.wavn >= wc;
      const unsigned c4 = own ? wp_cur.centre4 : wp_st.centre4;
      const int rb = own ? wp_cur.rowb : wp_st.rowb;
      const int d = (own ? wp_cur.psr : wp_st.psr) - (La.meta >> 3), sgn = d >> 31;
      const unsigned at = c4 + (unsigned)(d + (sgn & A.osamp)) * (unsigned)rb + (unsigned)((sgn + 1) << 2);
      __builtin_memcpy(&dst, tabw_base + at, sizeof dst);
    }
  };
  ahead(Q[0], nlines_slot > 0, R[0]);                       // (the first two lines' rows: nobody was ahead of them)
  ahead(Q[1], nlines_slot > 1, R[1]);

  double pk = 0.0, c0 = 0.0, E0 = 1.0;
  int cell = jc;
  const int trips = wave_max_i(nlines_slot);
  auto step = [&](const WalkLine &L, bool on, double e1, const Row &mine) {
    if (on && (L.meta & 4)) {                               // base point of the rebased exponential
      const double t0 = ct * L.wavn;
      E0 = exp_neg(t0, s_e2); c0 = -t0;
    }
    if (on && (L.meta & 1)) { cell = L.cell; grow = mine; }
    // ---- strength of the line in the lane's layer
    const double q = __builtin_fma(-E0, exp_small(__builtin_fma(ct, L.wavn, c0)), 1.0);
    if (on) pk += L.gf * e1 * q;
    if (on && (L.meta & 2)) {
      // ---- the group is complete: threshold, density, bins (extinction.c:464-509)
      const double pkf = pk * f;
      pk = 0.0;
      const bool below = pkf < lim;                          // :467
      const double kk = below ? 0.0 : pkf * dens;            // :472-473 (a group below the threshold adds 0 x its row)
      // ---- move the frame down to the group's cell
      int sh = jc - cell;
      if (sh >= NB) {
#pragma unroll
        for (int k = 0; k < NB; k++) { flush(k, acc[k], valid); acc[k] = 0.0; }
        // bins of the interval that no frame position covers still get their record: zeros
        for (int j = max(cell + Rc + 2, blo); j <= min(jc - Rc - 1, bhi); j++) A.part[(rec0 + (j - blo)) * kWalkLayers + li] = 0.0;
        jc = cell;
      } else {
        for (; sh > 0; sh--) {
          flush(NB - 1, acc[NB - 1], valid);
#pragma unroll
          for (int k = NB - 1; k > 0; k--) acc[k] = acc[k - 1];
          acc[0] = 0.0;
          jc--;
        }
      }
      // ---- bins: one row of the lane's profile holds all the frame's bins (k_line_walk, row form)
#pragma unroll
      for (int k = 0; k < NB; k++) acc[k] = __builtin_fma(kk, (double)grow.v[k], acc[k]);     // :507
    }
  };
  Q[0] = L;
  for (int t = 0; t < trips; t += kAhead) {
    // the four lines' exp(-c Elow/T) side by side: four independent chains instead of one after the other
    double e1[kAhead];
#pragma unroll
    for (int u = 0; u < kAhead; u++) e1[u] = exp_neg(ct * Q[u].elow, s_e2);
#pragma unroll
    for (int u = 0; u < kAhead; u++) {
      const WalkLine Lu = Q[u];
      const Row Ru = R[u];
      ahead(Q[(u + 2) % kAhead], t + u + 2 < nlines_slot, R[(u + 2) % kAhead]);      // line t+u+2: its row, two lines early
      const int nx = t + u + kAhead;                          // the line that takes this place in the ring
      Q[u] = lp[min(nx, max(nlines_slot - 1, 0))];            // (past the range's end: its last line again, never used)
      step(Lu, t + u < nlines_slot, e1[u], Ru);
    }
  }
#pragma unroll
  for (int k = 0; k < NB; k++) flush(k, acc[k], valid);
  if (valid) for (int j = blo; j <= min(jc - Rc - 1, bhi); j++) A.part[(rec0 + (j - blo)) * kWalkLayers + li] = 0.0;
}

// e[layer][j] = sum of the partial records of bin j, isotope blocks in order, ranges in order.
// A block handles 8 consecutive bins, one wavefront (lanes = layers) per bin; the records of a
// bin are requested four at a time and added in order; the 8 x 64 results cross an LDS tile so
// that every layer's row receives its 8 bins as one 64-byte segment.
struct CombineArgs {
  WalkPlan P; int niso; const int32_t *gblock;
  long long lo, nsh; int r_top, nc;
  int nmx; const int32_t *iso_mx;
  const double *part; double *e;
  const int *flags; const int *last; int eager;
};

constexpr int kCombineBins = 4;

__global__ __launch_bounds__(64 * kCombineBins)
void k_walk_combine(CombineArgs C)
{
  if (!C.eager && C.flags[0] == 0) return;
  latency_critical();
  __shared__ double s_t[kCombineBins][kWalkLayers + 1];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long j0 = C.lo + (long long)blockIdx.x * kCombineBins;
  const int nbins = (int)min((long long)kCombineBins, C.lo + C.nsh - j0);
  if (C.last) {   // every ray of these bins has stopped: nobody reads them (block-uniform)
    bool open = false;
    for (int t = 0; t < nbins; t++) open |= C.last[j0 - C.lo + t] < 0;
    if (!open) return;
  }
  const long long j = j0 + wv;                       // this wave's bin
  // (a bin whose ray has stopped: the walk may have skipped the whole range that would have written
  // its records -- nothing reads such a bin again, and it gets a zero rather than whatever the record
  // buffer held two steps ago)
  const bool have = wv < nbins && !(C.last && C.last[j - C.lo] >= 0);
  int cur_mx = -1;
  double sum = 0.0;
  auto store = [&](int mx) {                         // all waves of the block arrive here together
    s_t[wv][lane] = sum;
    __syncthreads();
    const int layer = threadIdx.x / kCombineBins, t = threadIdx.x % kCombineBins;
    if (layer < C.nc && t < nbins)
      C.e[((long long)(C.r_top - layer) * C.nmx + mx) * C.nsh + (j0 - C.lo) + t] = s_t[t][layer];
    __syncthreads();
  };
  for (int b = 0; b < C.niso; b++) {
    if (C.gblock[b] == C.gblock[b + 1]) continue;
    const int mx = C.nmx == 1 ? 0 : C.iso_mx[b];
    if (mx != cur_mx) {
      if (cur_mx >= 0) store(cur_mx);
      cur_mx = mx; sum = 0.0;
    }
    if (!have) continue;
    // ranges of the block that touch bin j
    int wa, wz;
    if (C.P.binw) { const long long t = (long long)b * C.nsh + (j - C.lo); wa = C.P.binw[2 * t]; wz = C.P.binw[2 * t + 1]; }
    else ranges_of_bin(C.P, b, j, wa, wz);
    // The record indices of all ranges touching the bin in ONE round trip (lane u: range w0 + u), then
    // the records eight loads at a time -- added in range order all the same.  (One range after the
    // other, each record waited for its range's off/blo: ~20 dependent round trips per bin at the
    // demo size, where ~45 ranges touch a bin of the 8-bin frame.)
    wa = __builtin_amdgcn_readfirstlane(wa); wz = __builtin_amdgcn_readfirstlane(wz);     // (one bin per wave)
    for (int w0 = wa; w0 < wz; w0 += 64) {
      const int n = min(64, wz - w0);
      long long myrec = 0;
      if (lane < n) myrec = C.P.off[w0 + lane] + (j - C.P.blo[w0 + lane]);
      const int rec_lo = (int)(myrec & 0xffffffffLL), rec_hi = (int)(myrec >> 32);
      for (int u0 = 0; u0 < n; u0 += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int uu = min(u0 + u, n - 1);                       // wave-uniform
          const long long rec = ((long long)__builtin_amdgcn_readlane(rec_hi, uu) << 32) |
                                (unsigned)__builtin_amdgcn_readlane(rec_lo, uu);
          v[u] = (u0 + u < n && lane < C.nc) ? C.part[rec * kWalkLayers + lane] : 0.0;     // (a step's idle lanes are never written)
        }
#pragma unroll
        for (int u = 0; u < 8; u++) if (u0 + u < n) sum += v[u];
      }
    }
  }
  if (cur_mx >= 0) store(cur_mx);
}

}  // namespace trx
