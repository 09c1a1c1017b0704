// trx_lanes.hip.h -- the line walk for steps of FEW layers with WIDE frames (8 or 16 bins, at most 32
// layers: the deep layers of a run), lanes = LINES for everything that is per (line, layer).
//
// k_line_walk (trx_walk.hip.h) puts the step's layers on the lanes and walks a range's lines one after
// the other: whatever the number of layers, a line costs the wave its ~58 vector and ~36 scalar
// instructions -- the demo's deep step (17 layers, two lanes each: 34 of 64 lanes busy, the two lanes of
// a pair computing the SAME strength) paid 136 us for 17 layers where the 64-layer step pays 104.
// Here a wavefront takes S consecutive ranges (production: one) and goes through their lines 32 at a time:
//
//   phase 1   lanes = (line, set of layers): 32 lines x 2 sets, the last lines of a run 16 x 4 or 8 x 8, the
//             step's layers in a loop.  Everything per layer comes from one 96-byte record per layer in LDS,
//             per line: the two exponentials (exp(ct * base point) once per (base point, layer) of the
//             batch), the co-added group's sum (in line order: the anchor's lane adds its members'
//             strengths one after the other by DPP, 87 % of the groups are single lines), the threshold
//             (extinction.c:467), the nearest Doppler index (:480-483, followed downwards from the
//             batch before: lines descend in wavenumber; searched per line only in a batch in which it
//             steps), the profile and the byte offset of the row the group's bins read (compact rows
//             [phase][profile][8 floats] for frames of 8 bins: the profile's offset + phase x slab).  Per
//             (line, layer): the weight kk (0 below the threshold) and that offset go to LDS -- 12 bytes.
//   phase 2   lanes = (layer, half of the frame) exactly as k_line_walk<NB, false, 2>: the frame of NB
//             accumulators slides down with the groups' cells, a bin that leaves it becomes one entry of
//             the range's partial record -- but a group now costs a lane two LDS reads, one 16- or
//             32-byte load and NB/2 multiply-adds.  Groups that share a cell (a dense list has hundreds
//             per cell) are taken D at a time: their row segments are requested together, one block
//             ahead of the block being added (two register sets in turn; the block in flight at a batch's
//             end stays in flight under the next batch's lines and strengths).
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
constexpr int kLanesMaxS = 8;          // ranges per wave at most
constexpr int kLanesMaxGroup = 16;     // members of the longest co-added group this kernel takes
constexpr int kLanesMaxLayers = 32;
constexpr int kLanesLayK = 12;         // doubles per layer record
constexpr int kLanesBases = 4;         // base points of the rebased exponential a batch's table holds (more: computed per line)

struct LanesExtra {
  const double *wbase;                 // [nlines] wavenumber of the line's base point (rebased exponential)
  int S;                               // ranges per wave
};

// per wave: kk [32][nc] doubles, at [32][nc] words, the layers' records (nc rounded up to even)
__host__ __device__ inline int lanes_at_doubles(int ne) { return (((kLanesBatch + 1) * ne / 2) + 1) & ~1; }      // (the offsets' words, a whole number of 16-byte units)
__host__ __device__ inline int lanes_wave_doubles(int nc) { const int ne = (nc + 1) & ~1; return (kLanesBatch + 1) * ne + lanes_at_doubles(ne) + kLanesLayK * ne + kLanesBases * ne; }
__host__ __device__ inline size_t lanes_lds_bytes(int nc, int ndop) { return (size_t)kLanesWaves * 8 * (size_t)lanes_wave_doubles(nc) + 8 * (size_t)(ndop + 1); }

// lane i <- lane i + 1 of the wave (DPP wave_shl:1; lane 63 keeps `old`)
__device__ __forceinline__ double wave_shl1(double v)
{
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xF, 0xF, false);
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
  double *s_kk = s_dyn + (size_t)wv * lanes_wave_doubles(nc);                // [BL lines + 1][ne]
  uint32_t *s_at = (uint32_t *)(s_kk + (BL + 1) * ne);                        // [BL lines + 1][ne]
  double (*LK)[kLanesLayK] = (double (*)[kLanesLayK])(s_kk + (BL + 1) * ne + lanes_at_doubles(ne));     // [ne] the layers' records
  double *s_E0 = s_kk + (BL + 1) * ne + lanes_at_doubles(ne) + kLanesLayK * ne;              // [kLanesBases][ne] exp(ct * base point)

  // ---- the wave's ranges: launched indices L0 .. L0 + nq - 1, lane q holds range q's numbers
  const int nlaunch = A.nseg > 0 ? A.seg_cum[A.nseg] : A.P.nwaves;
  const int L0 = ((A.xcd_map ? xcd_block(blockIdx.x, gridDim.x) : (int)blockIdx.x) * kLanesWaves + wv) * X.S;
  const int nq = max(0, min(X.S, nlaunch - L0));             // (0: a wave past the launch's last range; it still fills the tables and meets the barrier)
  int r_b = 0, r_l0 = 0, r_l1 = 0, r_blo = 0, r_bhi = -1, r_open = 0, r_rlo = 0, r_rhi = 0; double r_wavn0 = 0.0;
  if (lane < nq) {
    int w = L0 + lane;
    if (A.nseg > 0) {
      int w0 = A.seg_w0[0], c0 = A.seg_cum[0];
#pragma unroll
      for (int sg = 1; sg < kWalkSegs; sg++)
        if (sg < A.nseg && w >= A.seg_cum[sg]) { w0 = A.seg_w0[sg]; c0 = A.seg_cum[sg]; }
      w = w0 + (w - c0);
    }
    if (w < A.P.nwaves) {
      const RangeInfo RI = A.rinfo[w];                       // (one load: trx_create made the record)
      r_b = RI.b; r_l0 = RI.l0; r_l1 = RI.l1; r_wavn0 = RI.wavn0;
      r_blo = A.P.blo[w]; r_bhi = A.P.bhi[w];
      const long long rec = A.P.off[w];
      r_rlo = (int)(rec & 0xffffffffLL); r_rhi = (int)(rec >> 32);
      r_open = r_bhi >= r_blo;                               // (else: nothing of this range reaches the shard)
    }
  }
  for (int i = threadIdx.x; i <= A.ndop; i += 64 * kLanesWaves) s_thr[i] = A.dthr[i];
  if (threadIdx.x < 64) s_e2[threadIdx.x] = e2_mine;
  auto RL = [&](int v, int q) { return __builtin_amdgcn_readlane(v, q); };
  // the layers' scalars of the first range's isotope block (lane c = layer c): in flight under the closed-ray test
  const int b0 = RL(r_b, 0);
  struct LayerScalars { double ad, ct, f, dens, kmax, wcut; int il, idst, idop0; } pre{};
  auto layer_scalars = [&](int b) {
    LayerScalars v{};
    if (lane < ne) {
      const int cl = min(lane, nc - 1);                      // (the odd layer out: a copy of the last one, its results are never read)
      const int r = A.r_top - cl, ri = r * A.niso + b;
      v.ad = A.Y.alphad[ri]; v.il = A.Y.ilor[ri]; v.idst = A.sticky_idop[ri];
      v.ct = A.Y.negc_over_t[r]; v.f = A.Y.strength_f[ri]; v.dens = A.permol ? 1.0 : A.Y.density[ri];
      v.kmax = A.kmax[(long long)r * A.nmx + (A.nmx == 1 ? 0 : A.iso_mx[b])]; v.wcut = A.wcut[ri]; v.idop0 = A.Y.idop0[ri];
    }
    return v;
  };
  if (nq > 0) pre = layer_scalars(b0);
  if (A.last && lane < nq && r_open) {   // every ray of the range's bins has stopped (tau.c:277-287): nobody reads them
    bool open = false;
    for (int j = r_blo; j <= r_bhi && !open; j++) open = A.last[j - A.lo] < 0;
    r_open = open;
  }
  __syncthreads();                                           // (the only workgroup barrier: from here the waves are on their own)
  const unsigned long long open_q = __ballot(r_open != 0);
  if (open_q == 0ull) return;

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
  int cq = -1, jc = 0, blo = 0, bhi = -1; long long rec0 = 0;   // the range being accumulated (wave-uniform)
  auto flush = [&](int k, double v, bool mine) {             // bin of this lane's slot k leaves the frame
    const int j = jc - Rc + part * NS + k;
    if (valid && mine && j >= blo && j <= bhi) A.part[(rec0 + (j - blo)) * kWalkLayers + li] = v;
  };
  auto fill_zero = [&](int ja, int jb) {                     // bins no frame position covered: zeros
    ja = max(ja, blo); jb = min(jb, bhi);
    if (valid && part == 0) for (int j = ja; j <= jb; j++) A.part[(rec0 + (j - blo)) * kWalkLayers + li] = 0.0;
  };
  auto end_range = [&]() {
    if (cq < 0) return;
#pragma unroll
    for (int k = 0; k < NS; k++) { flush(k, acc[k], true); acc[k] = 0.0; }
    fill_zero(blo, jc - Rc - 1);
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

  // two blocks of up to D groups of one cell: one is being fetched while the other is added
  // (between iterations the block in flight is set 1; a batch with an odd number of blocks moves it there)
  struct alignas(16) Row { float v[NS]; };
  double bk_kk[2][D]; Row bk_row[2][D];
  int bn0 = 0, bn1 = 0, bcell0 = 0, bcell1 = 0, bq0 = 0, bq1 = 0;     // per set: groups in it, their cell, their range
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
    int &bn = s ? bn1 : bn0; const int bcell = s ? bcell1 : bcell0, bq = s ? bq1 : bq0;
    if (bn == 0) return;
    if (bq != cq) {
      end_range();
      cq = bq;
      blo = RL(r_blo, cq); bhi = RL(r_bhi, cq);
      rec0 = ((long long)RL(r_rhi, cq) << 32) | (unsigned)RL(r_rlo, cq);
      jc = bcell;                                            // (a range's first line anchors a group: its cell)
    } else if (bcell != jc) shift_to(bcell);
#pragma unroll
    for (int u = 0; u < D; u++)
#pragma unroll
      for (int k = 0; k < NS; k++) acc[k] = __builtin_fma(bk_kk[s][u], (double)bk_row[s][u].v[k], acc[k]);     // :507
    bn = 0;
  };

  int v_cell = 0;                                            // phase 1: the cell of the lane's line
  // (row BL of the two tables: the weight 0 and the offset 0 -- what a block's empty places read)
  if (lane < ne) { s_kk[BL * ne + lane] = 0.0; s_at[BL * ne + lane] = 0u; }

  // ---- runs of ranges that share an isotope block and follow each other in the list
  // the layers' records (lane c = layer c of the step): constants of (layer, isotope), the Doppler
  // index at the run's first line and its profile.  Doubles 0 ct, 1 f, 2 density, 3 threshold, 4 wcut,
  // 5 alphad, 6 lower end of the current Doppler index' interval; words from double 7: index, ilor;
  // from 8: {centre + 4, row bytes, ps % osamp, -} of the current profile; from 10: of the sticky one
  auto write_records = [&](const LayerScalars &ls, double wavn_first) {
    if (lane < ne) {
      double *K = LK[lane];
      const double ad = ls.ad;
      const int il = ls.il, idst = ls.idst;
      K[0] = ls.ct; K[1] = ls.f; K[2] = ls.dens;
      K[3] = A.ethresh * ls.kmax; K[4] = ls.wcut; K[5] = ad;
      const int cur = index_from(s_thr, ad * wavn_first, ls.idop0);
      const WalkProfile wc_ = A.walkprof[cur * A.nlor + il], ws_ = A.walkprof[idst * A.nlor + il];
      K[6] = s_thr[cur];
      int *KI = (int *)(K + 7);
      KI[0] = cur; KI[1] = il;
      KI[2] = r32 ? (int)A.wp32[cur * A.nlor + il] : (int)wc_.centre4 + 4; KI[3] = wc_.rowb; KI[4] = wc_.psr; KI[5] = 0;
      KI[6] = r32 ? (int)A.wp32[idst * A.nlor + il] : (int)ws_.centre4 + 4; KI[7] = ws_.rowb; KI[8] = ws_.psr; KI[9] = idst;
    }
  };
  // (the first run's, from the scalars asked for at the start: they are not kept beyond this point)
  write_records(pre, readlane_f64(r_wavn0, __builtin_ctzll(open_q)));
  const int q_first = __builtin_ctzll(open_q);               // the first run is the one that holds this range
  for (int qs = 0; qs < nq; ) {
    const int b = RL(r_b, qs);
    int qe = qs + 1;
    while (qe < nq && RL(r_b, qe) == b && RL(r_l0, qe) == RL(r_l1, qe - 1)) qe++;
    if (((open_q >> qs) & ((1ull << (qe - qs)) - 1ull)) == 0ull) { qs = qe; continue; }      // the whole run is closed
    if (!(b == b0 && q_first >= qs && q_first < qe)) {
      int q0 = qs; while (!((open_q >> q0) & 1ull)) q0++;
      __builtin_amdgcn_wave_barrier();                       // (the run before may still read its records)
      write_records(layer_scalars(b), readlane_f64(r_wavn0, q0));
    }
    __builtin_amdgcn_wave_barrier();

    const int lend = RL(r_l1, qe - 1);
    int pq = qs, pl1 = RL(r_l1, qs);                         // producer's range
    for (int l = RL(r_l0, qs); l < lend; ) {
      // (a closed range at the cursor is stepped over; one met inside a batch is computed and masked)
      if (qe - qs > 1 || !((open_q >> qs) & 1ull)) {
        int qx = pq; while (qx < qe && l >= RL(r_l1, qx)) qx++;
        if (qx < qe && !((open_q >> qx) & 1ull)) { l = RL(r_l1, qx); continue; }
      }
      // ---- phase 1: lines l .. l + n - 1, cut at the end of a group.  Lanes = (line t1, layer set h1): 32
      // lines x 2 sets of layers, or -- the last lines of a run -- 16 x 4 or 8 x 8: an iteration of the
      // layer loop then serves 4 or 8 layers of the few lines that are left
      const int left = lend - l;
      const int lg = left > 16 ? 5 : left > 8 ? 4 : 3;       // log2(lines per batch)
      const int BLx = 1 << lg, nsets = 64 >> lg;
      const int t1 = lane & (BLx - 1), h1 = lane >> lg;
      const int n0 = min(BLx, left);
      const bool in0 = t1 < n0;
      const WalkLine *lp = A.lines + l + (in0 ? t1 : 0);
      const double wavn = lp->wavn, elow = lp->elow, gf_l = lp->gf;
      const int meta = in0 ? lp->meta : 0;
      const double wb = X.wbase[l + (in0 ? t1 : 0)];
      v_cell = lp->cell;
      const unsigned lowm = (unsigned)((1ull << BLx) - 1ull);
      const unsigned ends = (unsigned)__ballot(in0 && (meta & 2)) & lowm;      // (every layer set holds the same lines: the first set's bits)
      const int n = 32 - __builtin_clz(ends);                 // (a group has at most kLanesMaxGroup members: ends != 0)
      const bool in = t1 < n;
      const double gf = in ? gf_l : 0.0;                      // (lanes past the batch: strength 0)
      const bool is_anchor = in && (meta & 1);
      // anchors of open ranges, and the places where a block must begin: a new cell, a new range
      unsigned mask = (unsigned)__ballot(is_anchor) & lowm;
      const int prev_cell = __builtin_amdgcn_update_dpp(0, v_cell, 0x138, 0xF, 0xF, false);      // wave_shr:1 -- the line before
      unsigned bstart = (unsigned)__ballot(is_anchor && (t1 == 0 || prev_cell != v_cell)) & lowm;
      if (qe - qs > 1) {
        int qmine = pq; bool first = false;
        for (int qx = pq; qx + 1 < qe; qx++) { const int e = RL(r_l1, qx); if (l + t1 >= e) qmine = qx + 1; first |= l + t1 == e; }
        mask = (unsigned)__ballot(is_anchor && ((open_q >> qmine) & 1ull)) & lowm;
        bstart |= (unsigned)__ballot(is_anchor && first) & lowm;
      }
      int len = 1;
      { const unsigned e = ends >> t1; if (is_anchor && e) len = __builtin_ctz(e) + 1; }
      int kmax_len = 1;
      while (__any(kmax_len < len)) kmax_len++;               // longest group of the batch
      const int imod = meta >> 3;
      // ---- exp(ct * base point) per (base point of the batch, layer): the batch's lines share a few base points
      // (a new one every 1/32 cm-1), so the value is made once per pair -- lanes = layers -- and read per line
      const unsigned leaders = ((unsigned)__ballot(in && ((meta & 4) || t1 == 0))) & lowm;
      const int nlead = __builtin_popcount(leaders);
      const bool e0_tab = nlead <= kLanesBases;
      const int bid = __builtin_popcount(leaders & (unsigned)((2ull << t1) - 1ull)) - 1;
      if (e0_tab) {
        unsigned lm = leaders;
        for (int j = 0; j < nlead; j++) {
          const int tl = __builtin_ctz(lm); lm &= lm - 1u;
          const double wbj = readlane_f64(wb, tl);
          if (lane < ne) s_E0[j * ne + lane] = exp_neg(LK[lane][0] * wbj, s_e2);
        }
      }
      // does some layer's Doppler index step inside this batch?  (lines descend: the last line decides)
      bool step = false;
      { const double wl = readlane_f64(wavn, n - 1); if (lane < ne) step = LK[lane][5] * wl < LK[lane][6]; }
      const bool slow = __any(step);
      __builtin_amdgcn_wave_barrier();
      const int row8 = t1 * ne;
      auto layer = [&](int c2, auto E0TAB, auto SLOW) {
        const int c = min(c2 + h1, ne - 1);                   // (sets past the last layer repeat it: the same values to the same places)
        double *K = LK[c];
        struct alignas(16) D2 { double a, b; }; struct alignas(16) I4 { int x, y, z, w; };
        const D2 k01 = *(const D2 *)(K + 0), k23 = *(const D2 *)(K + 2);
        const double ct = k01.a, f = k01.b, dens = k23.a, lim = k23.b, wc = K[4];
        I4 pc = *(const I4 *)(K + 8); const I4 ps = *(const I4 *)(K + 10);
        // ---- strength of the line in layer c (k_line_walk's arithmetic: same base points, same roundings)
        const double e1 = exp_neg(ct * elow, s_e2);
        const double t0 = ct * wb;
        double E0;
        if constexpr (decltype(E0TAB)::value) E0 = s_E0[bid * ne + c]; else E0 = exp_neg(t0, s_e2);
        const double q = __builtin_fma(-E0, exp_small(__builtin_fma(ct, wavn, -t0)), 1.0);
        const double s = gf * e1 * q;
        // ---- the group's sum on its anchor's lane, members in line order (extinction.c:449-462)
        double pk = s, sh = s;
        for (int k = 1; k < kmax_len; k++) {
          sh = wave_shl1(sh);                                  // strength of the line k lanes on
          pk += k < len ? sh : 0.0;
        }
        const double pkf = pk * f;
        const double kk = pkf < lim ? 0.0 : pkf * dens;        // :467, :472-473
        if constexpr (decltype(SLOW)::value) {
          // ---- nearest Doppler index (:480-483) where it steps inside the batch: per line, downwards
          const double v = K[5] * wavn;
          int *KI = (int *)(K + 7);
          const int cur = KI[0];
          int idx = cur; double th = K[6];
          while (__any(in && v < th)) { if (in && v < th) { idx--; th = s_thr[idx]; } }
          if (idx != cur) { const WalkProfile wp = A.walkprof[idx * A.nlor + KI[1]]; pc.x = r32 ? (int)A.wp32[idx * A.nlor + KI[1]] : (int)wp.centre4 + 4; pc.y = wp.rowb; pc.z = wp.psr; }
          const int ncur = __shfl(idx, (n - 1) + (h1 << lg), 64);
          __builtin_amdgcn_wave_barrier();
          if (t1 == 0 && c2 + h1 < ne && ncur != cur) {        // the next batch starts from the last line's index
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
        s_kk[row8 + c] = kk; s_at[row8 + c] = at;
      };
      using T = std::true_type; using F = std::false_type;
      if (slow)        for (int c2 = 0; c2 < ne; c2 += nsets) layer(c2, F{}, T{});
      else if (e0_tab) for (int c2 = 0; c2 < ne; c2 += nsets) layer(c2, T{}, F{});
      else             for (int c2 = 0; c2 < ne; c2 += nsets) layer(c2, F{}, F{});
      __builtin_amdgcn_wave_barrier();

      // ---- phase 2: the batch's groups, blocks of up to D that share a cell and a range
      auto produce = [&](auto SET) {
        constexpr int s = decltype(SET)::value;
        const int t0 = __builtin_ctz(mask);
        while (l + t0 >= pl1) { pq++; pl1 = RL(r_l1, pq); }
        (s ? bcell1 : bcell0) = RL(v_cell, t0); (s ? bq1 : bq0) = pq;
        // the block: the anchors from t0 up to the next place where a block must begin, D at most
        const unsigned nxt = bstart & ~((2u << t0) - 1u);
        unsigned bm = nxt ? mask & ((1u << __builtin_ctz(nxt)) - 1u) : mask;
        int cnt = 0;
#pragma unroll
        for (int u = 0; u < D; u++) {
          int row = BL * ne;                                   // (an empty place: weight 0, offset 0)
          if (bm) { row = __builtin_ctz(bm) * ne; bm &= bm - 1u; cnt++; }
          bk_kk[s][u] = s_kk_lane[row];
          __builtin_memcpy(&bk_row[s][u], tabw_base + (s_at_lane[row] + part_off), sizeof(Row));
        }
        // (what was taken: the cnt lowest anchors)
        for (int k = 0; k < cnt; k++) mask &= mask - 1u;
        (s ? bn1 : bn0) = cnt;
      };
      while (mask) {
        produce(std::integral_constant<int, 0>{}); consume(std::integral_constant<int, 1>{});
        if (!mask) {                                           // set 0 stays in flight: it becomes set 1
          bn1 = bn0; bcell1 = bcell0; bq1 = bq0; bn0 = 0;
#pragma unroll
          for (int u = 0; u < D; u++) { bk_kk[1][u] = bk_kk[0][u]; bk_row[1][u] = bk_row[0][u]; }
          break;
        }
        produce(std::integral_constant<int, 1>{}); consume(std::integral_constant<int, 0>{});
      }
      __builtin_amdgcn_wave_barrier();                         // (the next batch overwrites the LDS entries)
      l += n;
    }
    qs = qe;
  }
  // the block still in flight, then the last range's bins
  consume(std::integral_constant<int, 1>{});
  end_range();
}

}  // namespace trx
