// trx_lanes.hip.h -- the line walk for steps of FEW layers with WIDE frames (8 or 16 bins, at most 32
// layers: the deep layers of a run), lanes = LINES for everything that is per (line, layer).
//
// k_line_walk (trx_walk.hip.h) puts the step's layers on the lanes and walks a range's lines one after
// the other: whatever the number of layers, a line costs the wave its ~58 vector and ~36 scalar
// instructions -- the demo's deep step (17 layers, two lanes each: 34 of 64 lanes busy, the two lanes of
// a pair computing the SAME strength) paid 136 us for 17 layers where the 64-layer step pays 104.
// Here a wavefront takes ONE range and goes through its lines 32 at a time:
//
//   phase 1   lanes = (line, set of layers): 32 lines x 2 sets, the last lines of the range 16 x 4 or 8 x 8, the
//             step's layers in a loop.  Everything per layer comes from one 96-byte record per layer in LDS,
//             per line: the two exponentials (exp(ct * base point) once per (base point, layer) of the
//             batch), the co-added group's sum (in line order: the anchor's lane adds its members'
//             strengths one after the other by DPP, 87 % of the groups are single lines), the threshold
//             (extinction.c:467), the nearest Doppler index (:480-483, followed downwards from the
//             batch before: lines descend in wavenumber; searched per line only in a batch in which it
//             steps), the profile and the byte offset of the row the group's bins read (compact rows
//             [phase][profile][8 floats] for frames of 8 bins: the profile's offset + phase x slab).  Per
//             (GROUP, layer): the weight kk (0 below the threshold) and that offset go to LDS -- 12 bytes,
//             in the row of the group's RANK among the batch's groups (members of a group write a spare row).
//   phase 2   lanes = (layer, half of the frame) exactly as k_line_walk<NB, false, 2>: the frame of NB
//             accumulators slides down with the groups' cells, a bin that leaves it becomes one entry of
//             the range's partial record -- but a group now costs a lane two LDS reads, one 16- or
//             32-byte load and NB/2 multiply-adds.  Groups that share a cell (a dense list has hundreds
//             per cell) are taken D at a time, consecutive ranks: their row segments are requested together,
//             one block ahead of the block being added.  Two register sets take turns; WHICH of them is in
//             flight when a batch ends is part of the control flow (the loop exists once per parity), so
//             that no register of a set is ever copied or reused while its loads are in flight: the
//             compiler's conservative s_waitcnt in front of such a reuse had cut the prefetch distance to
//             nothing (round 5).  The next batch's line records are requested before
//             phase 2 and arrive under it.
// Workgroups take their ranges by XCD (xcd_block): an L2 then holds the rows of one contiguous eighth of
// the list.  What a wave's time goes to, what bounds the kernel and the forms that were measured and not
// kept: DESIGN.md section 4.
//
// Same lines in the same order, same base points of the rebased exponential (their wavenumbers per line:
// WalkArgs-side array `wbase`, made by k_walk_marks), same frame, same records: the SAME BITS as
// k_line_walk / k_line_walk_packed, so the combine, the tail and every property that rests on the order
// of the sums are untouched (tests/test_gpu_lanes.py forces one form or the other on the same steps).
// Taken by the host for steps of at most 32 layers with frames of 8+ bins on lists whose co-added groups
// have at most kLanesMaxGroup members (longer ones: k_line_walk); production only.
#pragma once
#include "trx_walk.hip.h"

namespace trx {

constexpr int kLanesWaves = 2;         // waves per workgroup (each with 12 bytes x 32 lines x layers of LDS)
constexpr int kLanesBatch = 32;        // lines per batch: a lane of phase 1 is (line, set of layers)
constexpr int kLanesMaxGroup = 16;     // members of the longest co-added group this kernel takes
constexpr int kLanesMaxLayers = 32;
constexpr int kLanesLayK = 12;         // doubles per layer record
constexpr int kLanesBases = 4;         // base points of the rebased exponential a batch's table holds (more: computed per line)
constexpr int kLanesRows = kLanesBatch + 2;      // rows of the two per-(group, layer) tables: the batch's groups by rank, a row of zeros (what a block's empty places read), a spare row (what the members of a group write)

struct LanesExtra {
  const double *wbase;                 // [nlines] wavenumber of the line's base point (rebased exponential)
};

// per wave: kk [rows][ne] doubles, at [rows][ne] words, the layers' records (ne: layers rounded up to even), exp(ct * base point) [bases][ne], the groups' cells [batch] words
__host__ __device__ inline int lanes_at_doubles(int ne) { return ((kLanesRows * ne / 2) + 1) & ~1; }      // (the offsets' words, a whole number of 16-byte units)
__host__ __device__ inline int lanes_wave_doubles(int nc) { const int ne = (nc + 1) & ~1; return kLanesRows * ne + lanes_at_doubles(ne) + kLanesLayK * ne + kLanesBases * ne + kLanesBatch / 2; }
__host__ __device__ inline size_t lanes_lds_bytes(int nc, int ndop) { return (size_t)kLanesWaves * 8 * (size_t)lanes_wave_doubles(nc) + 8 * (size_t)(ndop + 1); }

// lane i <- lane i + 1 of the wave (DPP wave_shl:1; lane 63 gets 0)
__device__ __forceinline__ double wave_shl1(double v)
{
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

template <int NB, int D>                                     // D: groups per block (two blocks in flight)
__global__ __launch_bounds__(64 * kLanesWaves)
void k_line_walk_lanes(WalkArgs A, LanesExtra X)
{
  constexpr int Rc = NB / 2 - 1, NS = NB / 2, LPL = 2, BL = kLanesBatch;
  static_assert(NB == 8 || NB == 16, "frames of 8 or 16 bins");
  if (!A.eager && A.flags[0] == 0) return;
  __shared__ double s_e2[64];
  extern __shared__ double s_dyn[];
  double *s_thr = s_dyn + (size_t)kLanesWaves * lanes_wave_doubles(A.nc);      // [ndop + 1] steps of the nearest-Doppler-index function (the workgroup's)
  // (a wave's start is a chain of dependent round trips -- 18 % of its life when each waited for the one before: the two
  // tables for LDS, the range's records and, as soon as the range's isotope block is known, the layers' scalars are asked
  // for together; the closed-ray test runs under them)
  const double e2_mine = threadIdx.x < 64 ? A.e2tab[threadIdx.x] : 0.0;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int nc = A.nc, ne = (nc + 1) & ~1;                   // (ne: layers rounded up to even -- a lane pair of phase 1 takes two)
  double *s_kk = s_dyn + (size_t)wv * lanes_wave_doubles(nc);                // [rows][ne]
  uint32_t *s_at = (uint32_t *)(s_kk + kLanesRows * ne);                      // [rows][ne]
  double (*LK)[kLanesLayK] = (double (*)[kLanesLayK])(s_kk + kLanesRows * ne + lanes_at_doubles(ne));     // [ne] the layers' records
  double *s_E0 = s_kk + kLanesRows * ne + lanes_at_doubles(ne) + kLanesLayK * ne;            // [kLanesBases][ne] exp(ct * base point)
  int *s_cellr = (int *)(s_E0 + kLanesBases * ne);                            // [BL] cell of the batch's group of rank r

  // ---- the wave's range (wave-uniform: scalar loads)
  const int nlaunch = A.nseg > 0 ? A.seg_cum[A.nseg] : A.P.nwaves;
  int w = (A.xcd_map ? xcd_block(blockIdx.x, gridDim.x) : (int)blockIdx.x) * kLanesWaves + wv;
  bool have = w < nlaunch;                                   // (a wave past the launch's last range still fills the tables and meets the barrier)
  if (have && A.nseg > 0) {
    int sgm = 0;
    while (sgm + 1 < A.nseg && w >= A.seg_cum[sgm + 1]) sgm++;
    w = A.seg_w0[sgm] + (w - A.seg_cum[sgm]);
  }
  have = have && w < A.P.nwaves;
  int r_b = 0, l0 = 0, l1 = 0, blo = 0, bhi = -1, cell0 = 0; long long rec0 = 0; double wavn0 = 0.0;
  if (have) {
    const RangeInfo RI = A.rinfo[w];                         // (one load: trx_create made the record)
    r_b = RI.b; l0 = RI.l0; l1 = RI.l1; wavn0 = RI.wavn0; cell0 = RI.cell0;
    blo = A.P.blo[w]; bhi = A.P.bhi[w]; rec0 = A.P.off[w];
  }
  for (int i = threadIdx.x; i <= A.ndop; i += 64 * kLanesWaves) s_thr[i] = A.dthr[i];
  if (threadIdx.x < 64) s_e2[threadIdx.x] = e2_mine;
  // the layers' scalars of the range's isotope block (lane c = layer c): in flight under the closed-ray test
  struct LayerScalars { double ad, ct, f, dens, kmax, wcut; int il, idst, idop0; } pre{};
  if (have && lane < ne) {
    const int cl = min(lane, nc - 1);                        // (the odd layer out: a copy of the last one, its results are never read)
    const int r = A.r_top - cl, ri = r * A.niso + r_b;
    pre.ad = A.Y.alphad[ri]; pre.il = A.Y.ilor[ri]; pre.idst = A.sticky_idop[ri];
    pre.ct = A.Y.negc_over_t[r]; pre.f = A.Y.strength_f[ri]; pre.dens = A.permol ? 1.0 : A.Y.density[ri];
    pre.kmax = A.kmax[(long long)r * A.nmx + (A.nmx == 1 ? 0 : A.iso_mx[r_b])]; pre.wcut = A.wcut[ri]; pre.idop0 = A.Y.idop0[ri];
  }
  bool open = have && bhi >= blo;                            // (else: nothing of this range reaches the shard)
  if (open && A.last) {   // every ray of the range's bins has stopped (tau.c:277-287): nobody reads them
    bool o = false;
    for (int j = blo + lane; j <= bhi; j += 64) o |= A.last[j - A.lo] < 0;
    open = __ballot(o) != 0ull;
  }
  __syncthreads();                                           // (the only workgroup barrier: from here the waves are on their own)
  if (!open) return;

  // ---- phase 2's lane: layer li, part of the frame (slots part*NS ...)
  const int li = lane >> 1, part = lane & 1;
  const bool valid = li < nc;
  const int lic = valid ? li : 0;
  const bool r32 = NB == 8 && A.tabw32 != nullptr;           // compact 32-byte rows (frames of 8 bins): offset = profile's + phase * slab, no borrow term
  const char *tabw_base = r32 ? (const char *)A.tabw32 : (const char *)A.tabw - 4 * (Rc + 1);     // (wave-uniform; a lane adds its part's 4 * NS * part)
  const unsigned part_off = 4u * NS * (unsigned)part;
  const double *s_kk_lane = s_kk + lic; const uint32_t *s_at_lane = s_at + lic;
  double acc[NS];
#pragma unroll
  for (int k = 0; k < NS; k++) acc[k] = 0.0;
  int jc = cell0;                                            // the frame: acc[k] <-> bin jc - Rc + part*NS + k (a range's first line anchors a group: its cell)
  auto flush = [&](int k, double v, bool mine) {             // bin of this lane's slot k leaves the frame
    const int j = jc - Rc + part * NS + k;
    if (valid && mine && j >= blo && j <= bhi) A.part[(rec0 + (j - blo)) * kWalkLayers + li] = v;
  };
  auto fill_zero = [&](int ja, int jb) {                     // bins no frame position covered: zeros
    ja = max(ja, blo); jb = min(jb, bhi);
    if (valid && part == 0) for (int j = ja; j <= jb; j++) A.part[(rec0 + (j - blo)) * kWalkLayers + li] = 0.0;
  };
  auto shift_to = [&](int cell) {                            // move the frame down to the group's cell (k_line_walk)
    int sh = jc - cell;
    if (sh >= NB) {
#pragma unroll
      for (int k = 0; k < NS; k++) { flush(k, acc[k], true); acc[k] = 0.0; }
      fill_zero(cell + Rc + 2, jc - Rc - 1);
      jc = cell;
    } else {
      for (; sh > 0; sh--) {
        flush(NS - 1, acc[NS - 1], part == LPL - 1);
        const double carry = dpp_f64<0xB1>(acc[NS - 1]);          // quad_perm [1,0,3,2]: the pair's other lane
#pragma unroll
        for (int k = NS - 1; k > 0; k--) acc[k] = acc[k - 1];
        acc[0] = part == 1 ? carry : 0.0;
        jc--;
      }
    }
  };

  // two blocks of up to D groups of one cell: one is being fetched while the other is added.  A set is consumed exactly
  // once per produce (and once, all zeros, before the first one): nothing below tests whether its loads have landed
  // except the multiply-adds that use them
  struct alignas(16) Row { float v[NS]; };
  double bk_kk[2][D]; Row bk_row[2][D];
  int bn[2] = {0, 0}, bcell[2] = {0, 0};                     // per set: groups in it, their cell
#pragma unroll
  for (int s = 0; s < 2; s++)
#pragma unroll
    for (int u = 0; u < D; u++) {
      bk_kk[s][u] = 0.0;
#pragma unroll
      for (int k = 0; k < NS; k++) bk_row[s][u].v[k] = 0.f;
    }
  auto consume = [&](auto SET) {
    constexpr int s = decltype(SET)::value;
    if (bn[s] != 0 && bcell[s] != jc) shift_to(bcell[s]);
#pragma unroll
    for (int u = 0; u < D; u++)
#pragma unroll
      for (int k = 0; k < NS; k++) acc[k] = __builtin_fma(bk_kk[s][u], (double)bk_row[s][u].v[k], acc[k]);     // :507
  };

  // (rows BL, BL + 1 of the two tables: the weight 0 and the offset 0 -- what a block's empty places read; what a group's members write)
  if (lane < ne) { s_kk[BL * ne + lane] = 0.0; s_at[BL * ne + lane] = 0u; s_at[(BL + 1) * ne + lane] = 0u; }

  // ---- the layers' records (lane c = layer c of the step): constants of (layer, isotope), the Doppler
  // index at the range's first line and its profile.  Doubles 0 ct, 1 f, 2 density, 3 threshold, 4 wcut,
  // 5 alphad, 6 lower end of the current Doppler index' interval; words from double 7: index, ilor;
  // from 8: {centre + 4, row bytes, ps % osamp, -} of the current profile; from 10: of the sticky one
  if (lane < ne) {
    double *K = LK[lane];
    K[0] = pre.ct; K[1] = pre.f; K[2] = pre.dens;
    K[3] = A.ethresh * pre.kmax; K[4] = pre.wcut; K[5] = pre.ad;
    const int cur = index_from(s_thr, pre.ad * wavn0, pre.idop0);
    const WalkProfile wc_ = A.walkprof[cur * A.nlor + pre.il], ws_ = A.walkprof[pre.idst * A.nlor + pre.il];
    K[6] = s_thr[cur];
    int *KI = (int *)(K + 7);
    KI[0] = cur; KI[1] = pre.il;
    KI[2] = r32 ? (int)A.wp32[cur * A.nlor + pre.il] : (int)wc_.centre4 + 4; KI[3] = wc_.rowb; KI[4] = wc_.psr; KI[5] = 0;
    KI[6] = r32 ? (int)A.wp32[pre.idst * A.nlor + pre.il] : (int)ws_.centre4 + 4; KI[7] = ws_.rowb; KI[8] = ws_.psr; KI[9] = pre.idst;
  }

  // a batch's line records: lanes = (line t1, layer set h1), 2^lg lines
  struct Lines { double wavn, elow, gf, wb; int meta, cell; };
  auto geometry = [&](int l_at, int &lg) { const int left = l1 - l_at; lg = left > 16 ? 5 : left > 8 ? 4 : 3; };      // log2(lines per batch)
  auto fetch = [&](int l_at) -> Lines {
    int lg; geometry(l_at, lg);
    const int BLx = 1 << lg, t1 = lane & (BLx - 1);
    const bool in0 = t1 < min(BLx, l1 - l_at);
    const int at = l_at + (in0 ? t1 : 0);
    const WalkLine *lp = A.lines + at;
    Lines b; b.wavn = lp->wavn; b.elow = lp->elow; b.gf = lp->gf; b.meta = in0 ? lp->meta : 0; b.cell = lp->cell; b.wb = X.wbase[at];
    return b;
  };
  double c24 = 0x1.5555555555555p-5;                         // 1/24, kept in a vector register (exp_neg's note, trx_kernels.hip.h)
  asm volatile("" : "+v"(c24));
  Lines cur = fetch(l0);
  int e0_prev = -1;                                          // row of the exp(ct * base point) table that holds the last base point of the batch before (-1: none)
  __builtin_amdgcn_wave_barrier();
  // set 1 starts as an empty block IN FLIGHT (weights 0): the state every later batch begins in -- the first batch's wait
  // for its line records then leaves these loads outstanding like any other's (one wait count for the loop, not "all")
  bn[1] = 0; bcell[1] = cell0;
#pragma unroll
  for (int u = 0; u < D; u++) {
    bk_kk[1][u] = s_kk_lane[BL * ne];
    __builtin_memcpy(&bk_row[1][u], tabw_base + (s_at_lane[BL * ne] + part_off + 32u * u), sizeof(Row));     // (D loads, not one and its copies: any bytes of the table will do under a weight of 0)
  }

  for (int l = l0; l < l1; ) {
    // ---- phase 1: lines l .. l + n - 1, cut at the end of a group.  Lanes = (line t1, layer set h1): 32
    // lines x 2 sets of layers, or -- the last lines of the range -- 16 x 4 or 8 x 8: an iteration of the
    // layer loop then serves 4 or 8 layers of the few lines that are left
    int lg; geometry(l, lg);
    const int BLx = 1 << lg, nsets = 64 >> lg;
    const int t1 = lane & (BLx - 1), h1 = lane >> lg;
    const int n0 = min(BLx, l1 - l);
    const bool in0 = t1 < n0;
    const double wavn = cur.wavn, elow = cur.elow, wb = cur.wb;
    const int meta = cur.meta, v_cell = cur.cell;
    const unsigned lowm = (unsigned)((1ull << BLx) - 1ull);
    const unsigned ends = (unsigned)__ballot(in0 && (meta & 2)) & lowm;      // (every layer set holds the same lines: the first set's bits)
    const int n = ends ? 32 - __builtin_clz(ends) : n0;      // (a group has at most kLanesMaxGroup members -- the host checks --: ends != 0)
    const bool in = t1 < n;
    const double gf = in ? cur.gf : 0.0;                      // (lanes past the batch: strength 0)
    const bool is_anchor = in && (meta & 1);
    // the batch's groups by rank (their anchors in line order), and the ranks at which a block must begin: a new cell
    const unsigned amask = (unsigned)__ballot(is_anchor) & lowm;
    const int nanch = __builtin_popcount(amask);
    const int rank = __builtin_popcount(amask & (unsigned)((1ull << t1) - 1ull));
    const int prev_cell = __builtin_amdgcn_update_dpp(0, v_cell, 0x138, 0xF, 0xF, false);      // wave_shr:1 -- the line before
    unsigned bstart_r = 0u;                                   // (rank 0 begins a block anyway)
    for (unsigned m = (unsigned)__ballot(is_anchor && t1 > 0 && prev_cell != v_cell) & lowm; m; m &= m - 1u)
      bstart_r |= 1u << __builtin_popcount(amask & ((1u << __builtin_ctz(m)) - 1u));
    if (is_anchor && h1 == 0) s_cellr[rank] = v_cell;
    const int rowk = (is_anchor ? rank : BL + 1) * ne;        // where this lane's (group, layer) entries go
    int len = 1;
    { const unsigned e = ends >> t1; if (is_anchor && e) len = __builtin_ctz(e) + 1; }
    int kmax_len = 1;
    while (__any(kmax_len < len)) kmax_len++;                 // longest group of the batch
    const double m1 = len > 1 ? 1.0 : 0.0;                    // the group has a second member
    const int imod = meta >> 3;
    // ---- exp(ct * base point) per (base point of the batch, layer): the batch's lines share a few base points
    // (a new one every 1/32 cm-1), so the value is made once per pair -- lanes = layers -- and read per line
    const unsigned leaders = ((unsigned)__ballot(in && ((meta & 4) || t1 == 0))) & lowm;
    const int nlead = __builtin_popcount(leaders);
    const bool e0_tab = nlead <= kLanesBases;
    const int bid = __builtin_popcount(leaders & (unsigned)((2ull << t1) - 1ull)) - 1;
    if (e0_tab) {
      unsigned lm = leaders; int j = 0;
      if (e0_prev >= 0 && !(__builtin_amdgcn_readfirstlane(meta) & 4)) {
        // the batch's first line goes on from the base point the batch before ended on: its row of the table is that one
        if (e0_prev != 0 && lane < ne) s_E0[lane] = s_E0[e0_prev * ne + lane];
        lm &= lm - 1u; j = 1;
      }
      for (; j < nlead; j++) {
        const int tl = __builtin_ctz(lm); lm &= lm - 1u;
        const double wbj = readlane_f64(wb, tl);
        if (lane < ne) s_E0[j * ne + lane] = exp_neg(LK[lane][0] * wbj, s_e2);
      }
      e0_prev = nlead - 1;
    } else e0_prev = -1;
    // does some layer's Doppler index step inside this batch?  (lines descend: the last line decides)
    bool step = false;
    { const double wl = readlane_f64(wavn, n - 1); if (lane < ne) step = LK[lane][5] * wl < LK[lane][6]; }
    const bool slow = __any(step);
    __builtin_amdgcn_wave_barrier();
    auto layer = [&](int c2, auto E0TAB, auto SLOW, auto CLAMP) {
      // (two sets of layers -- a full batch -- never pass the last layer, ne is even; four or eight sets do: they repeat it, the same values to the same places)
      const int c = decltype(CLAMP)::value ? min(c2 + h1, ne - 1) : c2 + h1;
      double *K = LK[c];
      struct alignas(16) D2 { double a, b; }; struct alignas(16) I4 { int x, y, z, w; };
      const D2 k01 = *(const D2 *)(K + 0), k23 = *(const D2 *)(K + 2);
      const double ct = k01.a, f = k01.b, dens = k23.a, lim = k23.b, wc = K[4];
      I4 pc = *(const I4 *)(K + 8); const I4 ps = *(const I4 *)(K + 10);
      // ---- strength of the line in layer c (k_line_walk's arithmetic: same base points, same roundings)
      const double e1 = exp_neg(ct * elow, s_e2, c24);
      const double t0 = ct * wb;
      double E0;
      if constexpr (decltype(E0TAB)::value) E0 = s_E0[bid * ne + c]; else E0 = exp_neg(t0, s_e2, c24);
      const double q = __builtin_fma(-E0, exp_small(__builtin_fma(ct, wavn, -t0), c24), 1.0);
      const double s = gf * e1 * q;
      // ---- the group's sum on its anchor's lane, members in line order (extinction.c:449-462).  The second member
      // through a multiply-add with 1 or 0 (s + sh in one rounding, or s): most batches have no longer group
      double pk = s;
      if (kmax_len > 1) {
        double sh = wave_shl1(s);                            // strength of the line one lane on
        pk = __builtin_fma(sh, m1, pk);
        for (int k = 2; k < kmax_len; k++) {
          sh = wave_shl1(sh);                                // ... k lanes on
          pk += k < len ? sh : 0.0;
        }
      }
      const double pkf = pk * f;
      const double kk = pkf < lim ? 0.0 : pkf * dens;        // :467, :472-473
      if constexpr (decltype(SLOW)::value) {
        // ---- nearest Doppler index (:480-483) where it steps inside the batch: per line, downwards
        const double v = K[5] * wavn;
        int *KI = (int *)(K + 7);
        const int cur_i = KI[0];
        int idx = cur_i; double th = K[6];
        while (__any(in && v < th)) { if (in && v < th) { idx--; th = s_thr[idx]; } }
        if (idx != cur_i) { const WalkProfile wp = A.walkprof[idx * A.nlor + KI[1]]; pc.x = r32 ? (int)A.wp32[idx * A.nlor + KI[1]] : (int)wp.centre4 + 4; pc.y = wp.rowb; pc.z = wp.psr; }
        const int ncur = __shfl(idx, (n - 1) + (h1 << lg), 64);
        __builtin_amdgcn_wave_barrier();
        if (t1 == 0 && c2 + h1 < ne && ncur != cur_i) {      // the next batch starts from the last line's index
          const WalkProfile wp = A.walkprof[ncur * A.nlor + KI[1]];
          K[6] = s_thr[ncur]; KI[0] = ncur; KI[2] = r32 ? (int)A.wp32[ncur * A.nlor + KI[1]] : (int)wp.centre4 + 4; KI[3] = wp.rowb; KI[4] = wp.psr;
        }
      }
      const bool own = wavn >= wc;                           // own index while alphad*wn/alphal >= 0.1, else the sticky one
      const int c4p = own ? pc.x : ps.x, rb = own ? pc.y : ps.y, psr = own ? pc.z : ps.z;
      const int d = psr - imod, sgn = d >> 31;              // sgn = -1: borrowed a cell
      const unsigned rowsel = (unsigned)(d + (sgn & A.osamp));
      // (compact rows: [phase][profile][8 floats] -- the row of phase imod of every profile in one slab, so the
      // rows of one group's layers, whose profiles are neighbours in the table, share cache lines)
      const unsigned at = r32 ? (unsigned)c4p + __umul24((unsigned)imod, A.slab32)
                              : __umul24(rowsel, (unsigned)rb) + (unsigned)(c4p + (sgn << 2));     // (osamp < 2^21, a row < 2^24 bytes)
      s_kk[rowk + c] = kk; s_at[rowk + c] = at;
    };
    using T = std::true_type; using F = std::false_type;
    if (slow)             for (int c2 = 0; c2 < ne; c2 += nsets) layer(c2, F{}, T{}, T{});
    else if (!e0_tab)     for (int c2 = 0; c2 < ne; c2 += nsets) layer(c2, F{}, F{}, T{});
    else if (nsets != 2)  for (int c2 = 0; c2 < ne; c2 += nsets) layer(c2, T{}, F{}, T{});
    else                  for (int c2 = 0; c2 < ne; c2 += 2)     layer(c2, T{}, F{}, F{});
    __builtin_amdgcn_wave_barrier();

    // ---- the next batch's lines: requested here, they arrive under phase 2
    const int l_next = l + n;
    if (l_next < l1) cur = fetch(l_next);

    // ---- phase 2: the batch's groups by rank, blocks of up to D consecutive ones that share a cell.  A batch takes an
    // EVEN number of blocks -- an empty one in front where the count is odd -- so that the block in flight between batches
    // is always set 1's: which set is in flight is then a fact of the program's structure, not a run-time value the
    // compiler must guard against with waits (or a copy of registers whose loads are still on their way)
    const int c_rank = s_cellr[lane & (BL - 1)];              // lane r: cell of the group of rank r
    int j = 0;                                                // next rank to take
    auto block_end = [&](int a) {                             // end of the run of ranks from a that may share a block: the next rank at which one must begin
      const unsigned nxt = bstart_r >> a >> 1;
      return nxt ? min(nanch, a + 1 + __builtin_ctz(nxt)) : nanch;
    };
    int nblk = 0;
    for (int a = 0; a < nanch; ) { const int e = block_end(a); nblk += (e - a + D - 1) / D; a = e; }
    bool pad = nblk & 1;
    auto produce = [&](auto SET, bool empty) {
      constexpr int s = decltype(SET)::value;
      const int cnt = empty ? 0 : min(D, block_end(j) - j);
      bcell[s] = __builtin_amdgcn_readlane(c_rank, min(j, BL - 1)); bn[s] = cnt;
#pragma unroll
      for (int u = 0; u < D; u++) {
        const int row = (u < cnt ? j + u : BL) * ne;          // (an empty place: weight 0, offset 0)
        bk_kk[s][u] = s_kk_lane[row];
        __builtin_memcpy(&bk_row[s][u], tabw_base + (s_at_lane[row] + part_off), sizeof(Row));
      }
      j += cnt;
    };
    int it = (nblk + 1) >> 1;                                 // (a batch begins with an anchor: at least one pair of blocks)
    do {
      produce(std::integral_constant<int, 0>{}, pad); pad = false;
      consume(std::integral_constant<int, 1>{});
      produce(std::integral_constant<int, 1>{}, false);
      consume(std::integral_constant<int, 0>{});
    } while (--it > 0);
    __builtin_amdgcn_wave_barrier();                         // (the next batch overwrites the LDS entries)
    l = l_next;
  }
  // the block still in flight, then the range's last bins
  consume(std::integral_constant<int, 1>{});
#pragma unroll
  for (int k = 0; k < NS; k++) flush(k, acc[k], true);
  fill_zero(blo, jc - Rc - 1);
}

}  // namespace trx
