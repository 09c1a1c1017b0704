// trx_tail.hip.h -- everything a hinted eclipse run does AFTER its walks, as one kernel.
//
// A demo-sized spectrum (2501 rays, 81 layers in two walk steps) used to end in six launches on
// two queues -- combine, optical depth (x2) of the first step on a side queue under the second
// walk; combine, optical depth, emission of the last step behind it, two copies back -- and a
// third of its time was what lies between kernels: a dependent launch costs 5-10 us, a wait for
// another queue's event 10-20 us even when long satisfied, every event record between two walks
// ~15 us, and each of those latency-bound kernels begins with its own chain of memory round trips
// (~2-3 us each right behind a walk: plan, records, layer scalars, flags).  None of that work needs
// more than the ray's own data:
//
//   k_ray_tail      block = 10 rays (coarse bins) = 10 waves + the chain wave
//     waves 0..9   lanes = layers: the plan look-ups of BOTH steps side by side, then the partial
//                  records of their bin added in line order (k_walk_combine's sums) -> e (global,
//                  for dumps and saveext), e + e_cs -> LDS; the first step's before barrier X1,
//                  the second's while the chain wave is on its way down the first step's layers
//     wave 10      the layers' scalars (made by the host once per run: TauArgs.lay) and the rays'
//                  carried state -> LDS; then B1, one lane per ray: the chain of bottom-point
//                  parabolas (eclipse.c:66), the only part of an optical depth that depends on
//                  the layer above -- ~30 fp64 instructions per layer, one wave issuing
//     all          B2 the Simpson terms, B4 the division and the stopping test: one (layer, ray)
//                  pair per thread; B3 (chain wave) the running sums between them
//     waves 0..9   emission_ray (k_emission's code; lanes = heights; the Planck function of the
//                  ray's heights was made while the chain wave was on its way), this run's optical
//                  depths from LDS, the flux straight into pinned host memory;  chain wave: last, the
//                  running sums, and what the block adds to the run's flags -- rays still open, deepest
//                  layer -- as ONE store into a pinned array (host_blocks): the host adds the blocks
//                  up.  No atomic, no fence, no copy command behind the kernel.  (TRX_TAIL_DIRECT=0:
//                  the flags are summed on the device as tau_publish does, the last block to arrive
//                  publishes them, and copy commands bring flags and spectrum back.)
//
// Same operations in the same order on the same values as the kernels it stands for: the same
// bits (tests/test_gpu_tail.py compares every output with TRX_RAY_TAIL=0).  The side queue, its
// events and the waits on them disappear from a hinted run; a plan's second walk runs NEXT to the
// first on the side queue with the tail behind it there (trx_api.hip, "two queues"), and the one
// wait in front of the tail is for an event the main queue records behind its walk and the CIA
// kernels.  Not covered (the step kernels run as before): unhinted and resumed runs, plans of more
// than two steps or with two-kernel steps, shards of more than 65 536 rays, counting runs
// (trx_opts.profile 2), restored extinction.
#pragma once
#include "trx_walk.hip.h"

namespace trx {

constexpr int kTailRays = 10;                                 // rays per block: one combine / emission wave each.  10 + 1 waves: a demo-sized run is 251 blocks, ONE per CU, each
                                                              // alone with its latency chains -- with 7 + 1 (358 blocks, two to a CU on 102 of the 256 CUs) the spectrum took 5 us longer
                                                              // (0.2915 -> 0.2864 ms; 12 + 1: 0.2890; round 5)
constexpr int kTailWaves = kTailRays + 1;                     // and the wave that runs the chain of parabolas
constexpr int kTailThreads = 64 * kTailWaves;
constexpr int kTailSteps = 2;
constexpr int kTailLayers = kTailSteps * kWalkLayers;
constexpr int kTailBatch = 24;                                // records requested together (a bin of the 8-bin frame: ~45)

struct TailStep { WalkPlan P; const double *part; int nc; };
struct TailArgs {
  int nsteps; TailStep S[kTailSteps];                         // top step first; layers T.r_top, T.r_top-1, ... in that order
  int niso; const int32_t *gblock;
  unsigned long long blocks;                                  // bit b: isotope block b has groups (b < 64; beyond: gblock is looked at)
  double *e;                                                  // [nr][nsh]
  const int *skip;                                            // null, or the rays' `last`: a closed ray's bins are not combined (zero)
  int *host_flags;                                            // null, or pinned host memory: the run's flags [0..7] and status [16..19] land there too (no copy command behind the kernel)
  int *host_blocks;                                           // null, or pinned host memory [blocks][2]: (rays still open, deepest layer + 1) of each block -- the host adds them up, nothing is published on the device
                                                              // (slant rays: M.status_slots, behind the blocks' entries, takes what the device's status word would)
  TauArgs T;                                                  // r_top: the first step's; nc: all layers of the plan
  EmisArgs E;                                                 // eclipse geometry (NANG > 0)
  ModArgs M;                                                  // transit geometry (NANG == 0)
};

// Where the records of bin j (wave-uniform) are: per step the ranges [wa, wz) of the isotope block
// that touch it (the plan's per-bin table) and, lane u, the record of range wa + u -- for both
// steps side by side: two dependent round trips instead of four.
struct TailBin { int wa[kTailSteps], wz[kTailSteps]; long long myrec[kTailSteps]; };

__device__ __forceinline__ void tail_bin_plan(const TailArgs &A, int b, long long j, int lane, TailBin &B)
{
  const TauArgs &T = A.T;
  int32_t rec32[kTailSteps] = {};                            // (the plan's per-bin record table: asked for next to the range numbers, one round trip for both)
#pragma unroll
  for (int s = 0; s < kTailSteps; s++) {
    B.wa[s] = B.wz[s] = 0;
    if (s < A.nsteps) {
      const WalkPlan &P = A.S[s].P;
      if (P.binw) {
        const long long t = (long long)b * T.nsh + (j - T.lo);
        B.wa[s] = P.binw[2 * t]; B.wz[s] = P.binw[2 * t + 1];
        if (P.binrec) rec32[s] = P.binrec[t * 64 + lane];
      }
      else ranges_of_bin(P, b, j, B.wa[s], B.wz[s]);
    }
  }
#pragma unroll
  for (int s = 0; s < kTailSteps; s++) {
    B.wa[s] = __builtin_amdgcn_readfirstlane(B.wa[s]); B.wz[s] = __builtin_amdgcn_readfirstlane(B.wz[s]);
    B.myrec[s] = 0;
    if (lane < B.wz[s] - B.wa[s]) {                          // (the first 64 ranges)
      if (s < A.nsteps && A.S[s].P.binrec) B.myrec[s] = rec32[s];
      else B.myrec[s] = A.S[s].P.off[B.wa[s] + lane] + (j - A.S[s].P.blo[B.wa[s] + lane]);
    }
  }
}

// the records of one step added in range order, this lane's layer (k_walk_combine's sum), kTailBatch
// loads at a time: eight at a time a bin of the 8-bin frame is six dependent round trips
__device__ __forceinline__ double tail_add_records(const TailStep &S, const TailBin &B, int s, long long j, int lane, double sum)
{
  for (int w0 = B.wa[s]; w0 < B.wz[s]; w0 += 64) {
    const int n = min(64, B.wz[s] - w0);
    long long mr = B.myrec[s];
    if (w0 != B.wa[s]) { mr = 0; if (lane < n) mr = S.P.off[w0 + lane] + (j - S.P.blo[w0 + lane]); }
    const int rec_lo = (int)(mr & 0xffffffffLL), rec_hi = (int)(mr >> 32);
    for (int u0 = 0; u0 < n; u0 += kTailBatch) {
      double v[kTailBatch];
#pragma unroll
      for (int u = 0; u < kTailBatch; u++) {
        const int uu = min(u0 + u, n - 1);                       // wave-uniform
        const long long rec = ((long long)__builtin_amdgcn_readlane(rec_hi, uu) << 32) |
                              (unsigned)__builtin_amdgcn_readlane(rec_lo, uu);
        v[u] = (u0 + u < n && lane < S.nc) ? S.part[rec * kWalkLayers + lane] : 0.0;     // (a step's idle lanes are never written)
      }
#pragma unroll
      for (int u = 0; u < kTailBatch; u++) if (u0 + u < n) sum += v[u];
    }
  }
  return sum;
}

template <int NANG, bool EXTRAS>                            // angles of the emission (8: half the state of 16; 0: slant rays and the modulation of a transit); a scattering or cloud model is on
__global__ __launch_bounds__(kTailThreads, 4)                // (at most 128 registers: a block's 11 waves are 3 + 3 + 3 + 2 on a CU's SIMDs)
void k_ray_tail(TailArgs A)
{
  const TauArgs &T = A.T;
  constexpr bool SLANT = NANG == 0;
  latency_critical();
  __shared__ double s_x[kTailLayers + 1][kTailRays];          // total extinction e + e_cs (tau.c:231-232); row nct: zeros
  __shared__ double s_cs[SLANT ? kTailLayers : 1][kTailRays];  // slant rays: the layer's extinction without its lines (what an unswept layer holds, tau.c:231-232)
  __shared__ double s_p[kTailLayers][kTailRays];              // the Simpson term of the pair that starts at the layer
  __shared__ double s_a[kTailLayers][kTailRays];              // running Simpson sums A(layer)
  __shared__ double s_er[kTailLayers][kTailRays], s_tau[kTailLayers][kTailRays];
  __shared__ double s_rad[kTailLayers + 3];
  __shared__ double s_lay[(kTailLayers + 1) * kVertLay];
  __shared__ double s_ray[4][kTailRays];                       // a1, a2, y1, y2 on entry
  __shared__ double s_acc[2][kTailRays];                       // a1, a2 on exit
  __shared__ int s_last[kTailRays], s_stop[kTailRays];       // where the ray stopped before this run (-1: open); first layer of the plan that stops it
  __shared__ double s_e2[64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int nr = T.nr, nct = T.nc;
  const long long w0 = (long long)blockIdx.x * kTailRays;     // first ray of the block (index in the shard)
  const int nrays = (int)min((long long)kTailRays, T.nsh - w0);
  const int n0 = nr - T.r_top;                                // points of the ray that ends in the plan's first layer (layer c: n0 + c)
  const int nc0 = A.S[0].nc;                                  // layers of the first step
  const bool chain_wave = wv == kTailRays;
  double bpre[2] = {0.0, 0.0};                                // (ray waves, vertical rays) Planck function at heights lane, lane + 64

  if (!chain_wave) {
    // ---- the ray waves: bin w0 + wv, lanes = layers.  The plan look-ups of both steps, the first
    // step's records, and -- while the chain wave goes down the first step's layers -- the second's.
    const long long j = T.lo + w0 + wv;
    const bool have = wv < nrays && !(A.skip && A.skip[w0 + wv] >= 0);
    double ecs[kTailSteps] = {}, sum[kTailSteps] = {};
    {
      int c0 = 0;
#pragma unroll
      for (int s = 0; s < kTailSteps; s++)
        if (s < A.nsteps) {
          if (wv < nrays && lane < A.S[s].nc) ecs[s] = T.ecs[(long long)(T.r_top - c0 - lane) * T.nsh + w0 + wv];
          c0 += A.S[s].nc;
        }
    }
    // (what the emission needs of the inputs, requested here: used long after)
    const double r_e2 = (!SLANT && wv == 1) ? A.E.e2tab[lane] : 0.0;
    double tk[2] = {1.0, 1.0};                                  // the temperatures at this lane's heights (height = lane, lane + 64 from the top)
    if (!SLANT) {
#pragma unroll
      for (int q = 0; q < 2; q++) if (lane + 64 * q < nr) tk[q] = A.E.temp[nr - 1 - lane - 64 * q];
    }
    double xfs = 0.0, xfc = 0.0;                              // this ray's wavenumber factors of the two models
    if (EXTRAS && wv < nrays) { xfs = T.xf_scat[w0 + wv]; xfc = T.xf_cloud[w0 + wv]; }
    auto put = [&](int s, int c0) {                           // this bin's layers of step s: e to memory, the total extinction (tau.c:231-232) to the chain
      if (lane < A.S[s].nc) {
        const int rs = T.r_top - c0 - lane;
        if (wv < nrays) A.e[(long long)rs * T.nsh + w0 + wv] = sum[s];
        if (EXTRAS) s_x[c0 + lane][wv] = sum[s] + scat_layer(T, rs, xfs) + cloud_layer(T, rs, xfc) + ecs[s];
        else        s_x[c0 + lane][wv] = sum[s] + ecs[s];
        if (SLANT) s_cs[c0 + lane][wv] = EXTRAS ? scat_layer(T, rs, xfs) + cloud_layer(T, rs, xfc) + ecs[s] : ecs[s];
      }
    };
    auto empty = [&](int b) { return b < 64 ? !((A.blocks >> b) & 1ull) : A.gblock[b] == A.gblock[b + 1]; };   // no group of this isotope block in range
    TailBin B{}; int b_kept = -1;                              // (the plan look-ups of the last block are kept for the second step)
    if (have)
      for (int b = 0; b < A.niso; b++) {
        if (empty(b)) continue;
        tail_bin_plan(A, b, j, lane, B);
        b_kept = b;
        sum[0] = tail_add_records(A.S[0], B, 0, j, lane, sum[0]);
      }
    put(0, 0);
    if (wv == 1) s_e2[lane] = r_e2;
    __syncthreads();                                            // X1: the first step's extinction is in LDS
    if (A.nsteps > 1) {
      if (have)
        for (int b = 0; b < A.niso; b++) {
          if (empty(b)) continue;
          if (b != b_kept) { TailBin B2; tail_bin_plan(A, b, j, lane, B2); sum[1] = tail_add_records(A.S[1], B2, 1, j, lane, sum[1]); }
          else sum[1] = tail_add_records(A.S[1], B, 1, j, lane, sum[1]);
        }
      put(1, nc0);
    }
    // the Planck function at this ray's heights, while the chain wave is on its way: the emission's only part that
    // does not wait for the optical depths (a division, an exponential and a division per height, eclipse.c:131-134)
    if (!SLANT && wv < nrays) {
      const RayPlanck PL(A.E, w0 + wv, s_e2);
      bpre[0] = PL.at_temp(tk[0]); bpre[1] = PL.at_temp(tk[1]);
    }
    __syncthreads();                                            // X2: and the second's
  } else {
    // ---- the chain wave: the layers' and rays' scalars into LDS, then down the first step's layers
    // behind X1 while the other waves fetch the second step's records
    if (!SLANT) {
      for (int k = lane; k < (nct + 1) * kVertLay; k += 64) {
        const int c = k / kVertLay, rs = T.r_top - c;
        s_lay[k] = (c < nct && rs >= 0) ? T.lay[(long long)kVertLay * rs + (k - c * kVertLay)] : 0.0;
      }
      for (int k = lane; k < nct + 3; k += 64) { const int r = T.r_top + 1 - k; s_rad[k] = (r >= 0 && r < nr) ? T.rad[r] : 0.0; }
    }
    if (lane < kTailRays) {
      double a1 = 0, a2 = 0, y1 = 0, y2 = 0; int last = 0;
      if (lane < nrays) {
        const long long w = w0 + lane;
        if (!SLANT) {
          a1 = T.acc[w]; a2 = T.acc[T.nsh + w];
          if (T.r_top + 1 < nr) y1 = T.er[(long long)(T.r_top + 1) * T.nsh + w];
          if (T.r_top + 2 < nr) y2 = T.er[(long long)(T.r_top + 2) * T.nsh + w];
        }
        last = T.last[w];
      }
      s_ray[0][lane] = a1; s_ray[1][lane] = a2; s_ray[2][lane] = y1; s_ray[3][lane] = y2;
      s_last[lane] = last; s_stop[lane] = nct;
      s_x[nct][lane] = 0.0;
    }
    __syncthreads();                                            // X1
    // B1: er[c] = parabola through (yraw_c, er[c-1], er[c-2]) at the layer's radius (eclipse.c:66, kept;
    // the one- and two-point rays of the top two layers: :45-46, :65, not kept).  Only this depends
    // on the layer above: one lane per ray, ~30 instructions per layer; what a layer reads from LDS
    // is fetched two layers ahead -- behind the chain's own latency, not in front of every layer.
    const bool open_ray = lane < nrays && s_last[lane] < 0;
    struct In { double l0, l8, l1, l10, l2, l9, l7, l11, yraw; };
    auto fetch = [&](int c) -> In {
      const double *L = s_lay + kVertLay * min(c, nct);
      return In{L[0], L[8], L[1], L[10], L[2], L[9], L[7], L[11], s_x[min(c, nct)][lane]};
    };
    double y1 = s_ray[2][lane], y2 = s_ray[3][lane];
    auto one = [&](int c, const In &I, bool general) {
      const double y0 = general ? parab3_chain(I.l0, I.l8, I.l1, I.l10, I.l2, I.l9, I.yraw, y1, y2, I.l7, I.l11) : I.yraw;
      s_er[c][lane] = y0;
      y2 = y1; y1 = y0;
    };
    auto down = [&](int ca, int cz) {                          // layers ca .. cz-1
      if (!open_ray) return;
      int c = ca;
      for (; c < cz && n0 + c < 3; c++) one(c, fetch(c), false);
      In a0 = fetch(c), a1 = fetch(c + 1);
      for (; c + 4 <= cz; c += 4) {                            // (two sets of registers in turn: nothing is moved)
        const In b0 = fetch(c + 2), b1 = fetch(c + 3);
        one(c, a0, true); one(c + 1, a1, true);
        a0 = fetch(c + 4); a1 = fetch(c + 5);
        one(c + 2, b0, true); one(c + 3, b1, true);
      }
      if (c < cz) one(c, a0, true);
      if (c + 1 < cz) one(c + 1, a1, true);
      if (c + 2 < cz) one(c + 2, fetch(c + 2), true);
    };
    if (!SLANT) down(0, A.nsteps > 1 ? nc0 : nct);
    __syncthreads();                                            // X2
    if (!SLANT && A.nsteps > 1) down(nc0, nct);
  }
  __syncthreads();                                              // X3: er of all layers

  if constexpr (SLANT) {
  // ---- slant rays (slantpath.c:29-107; k_optical_depth's arithmetic: slant_bottom, slant_integral):
  // every (height, ray) pair is on its own -- the parabola at the closest approach, then Simpson
  // over the layers above with the height's weights.  What a layer of the NEXT step holds while a
  // step's heights are integrated is its extinction without lines (the reference sweeps lazily,
  // tau.c:231-232): s_cs, as the step kernels read it.
  const int r_bottom = T.r_top - nct + 1;
  for (int k = tid; k < nct * kTailRays; k += kTailThreads) {
    const int c = k / kTailRays, t = k % kTailRays;
    const int kl = T.r_top - c;                                  // the height's own layer
    const int rs = (int)T.hrs[kl];
    double y0 = 0.0;
    if (t < nrays && s_last[t] < 0 && rs >= 0) {
      const int r_low = (A.nsteps > 1 && c < nc0) ? T.r_top - nc0 + 1 : r_bottom;      // lowest layer swept when this height's step is integrated
      double ylow;
      if (rs >= r_low) ylow = s_x[T.r_top - rs][t];
      else if (rs >= r_bottom) ylow = s_cs[T.r_top - rs][t];
      else if (EXTRAS) ylow = scat_layer(T, rs, T.xf_scat[w0 + t]) + cloud_layer(T, rs, T.xf_cloud[w0 + t]) + T.ecs[(long long)rs * T.nsh + w0 + t];
      else             ylow = T.ecs[(long long)rs * T.nsh + w0 + t];
      y0 = slant_bottom(T, rs, T.hr0[kl], ylow, [&](int L) { return s_x[T.r_top - L][t]; });
    }
    s_p[c][t] = y0;
  }
  __syncthreads();
  for (int k = tid; k < nct * kTailRays; k += kTailThreads) {
    const int c = k / kTailRays, t = k % kTailRays;
    if (t < nrays && s_last[t] < 0) {
      const int kl = T.r_top - c;
      const int rs = (int)T.hrs[kl];
      double tv;
      if (rs == -1) tv = 0.0;                                    // slantpath.c:37-38: the outermost layer
      else if (rs < 0) { tv = nan(""); A.M.raise(3); }           // slantpath.c:39-44: the reference exits here (M.status is T.status)
      else tv = 2 * slant_integral(T, kl, rs, s_p[c][t], [&](int L) { return s_x[T.r_top - L][t]; });      // slantpath.c:107
      tv = T.rad_fct * tv;
      s_tau[c][t] = tv;
      const int ri = nr - 1 - kl;
      if ((tv > T.toomuch) | (ri == nr - 1)) atomicMin(&s_stop[t], c);       // tau.c:277-287, 299-304
      s_er[c][t] = s_x[c][t];                                    // (transit geometry leaves the total extinction in er as it is)
    }
  }
  __syncthreads();
  } else {
  // ---- the rest of k_optical_depth_vertical's arithmetic (vertical_layer), taken off the chain:
  // the Simpson terms (B2), the division and the stopping test (B4) are one (layer, ray) pair per
  // thread, and the running sums between them (B3) one addition per layer.
  for (int k = tid; k < nct * kTailRays; k += kTailThreads) {      // B2
    const int c = k / kTailRays, t = k % kTailRays, n = n0 + c;
    const double *L = s_lay + kVertLay * c;
    const double y0 = s_er[c][t];
    const double y1 = c >= 1 ? s_er[c - 1][t] : s_ray[2][t];
    const double y2 = c >= 2 ? s_er[c - 2][t] : (c == 1 ? s_ray[2][t] : s_ray[3][t]);
    if (n >= 3) s_p[c][t] = (y0 * L[3] + y1 * L[4] + y2 * L[5]) * L[6];
    else if (n == 2) {                                               // eclipse.c:65, 68-80: needs the layer below (a first step has >= 3 layers)
      const double yraw = s_x[c][t], ybelow = s_x[c + 1][t];
      const double yp = parab3(s_rad[c + 2], s_rad[c + 1], ybelow, yraw, y1, s_rad[c + 1]);
      const double *g = T.gw + (long long)(T.r_top - c) * T.gstride;
      s_tau[c][t] = T.rad_fct * (((yp * g[0] + ((y1 + yp) / 2.0) * g[1] + y1 * g[2]) * g[3]) / 6.0);
    } else s_tau[c][t] = 0.0;                                        // eclipse.c:45-46
  }
  __syncthreads();
  if (chain_wave && lane < nrays && s_last[lane] < 0) {              // B3: A(layer) = term + A(layer + 2)
    double a1 = s_ray[0][lane], a2 = s_ray[1][lane];
    int c = max(0, 3 - n0);
    for (; c + 4 <= nct; c += 4) {
      const double p0 = s_p[c][lane], p1 = s_p[c + 1][lane], p2 = s_p[c + 2][lane], p3 = s_p[c + 3][lane];
      const double a0 = p0 + a2, b0 = p1 + a1, c0 = p2 + a0, d0 = p3 + b0;
      s_a[c][lane] = a0; s_a[c + 1][lane] = b0; s_a[c + 2][lane] = c0; s_a[c + 3][lane] = d0;
      a2 = c0; a1 = d0;
    }
    for (; c < nct; c++) {
      const double a0 = s_p[c][lane] + a2;
      s_a[c][lane] = a0;
      a2 = a1; a1 = a0;
    }
    s_acc[0][lane] = a1; s_acc[1][lane] = a2;
  }
  __syncthreads();
  for (int k = tid; k < nct * kTailRays; k += kTailThreads) {      // B4
    const int c = k / kTailRays, t = k % kTailRays, n = n0 + c;
    if (n >= 3) {
      const double *L = s_lay + kVertLay * c;
      const double y0 = s_er[c][t], y1 = c >= 1 ? s_er[c - 1][t] : s_ray[2][t];
      const double a0 = s_a[c][t], a1 = (c >= 1 && n - 1 >= 3) ? s_a[c - 1][t] : s_ray[0][t];
      const bool odd = n & 1;
      const double sixth = quotient_rn(odd ? a0 : a1, 6.0, 1.0 / 6.0);
      const double with_first = sixth + L[0] * (y0 + y1) / 2;
      s_tau[c][t] = T.rad_fct * (odd ? sixth : with_first);
    }
    // a ray ends at the first layer whose optical depth passes toomuch, or at the bottom (tau.c:277-287, 299-304)
    const int ri = nr - 1 - (T.r_top - c);
    if ((s_tau[c][t] > T.toomuch) | (ri == nr - 1)) atomicMin(&s_stop[t], c);
  }
  __syncthreads();

  }

  // ---- results.  Waves 0..6: the emission of their ray (k_emission: lanes = heights; this run's
  // optical depths from LDS); the chain wave: the rays' state and the run's flags; everybody: er and tau.
  int pub_still = 0, pub_deep = 0;                                    // (chain wave) what this block adds to the run's flags
  auto outcome = [&](int t, int &last, int &done, bool &still) {      // of ray t, from the first stopping layer
    last = s_last[t]; done = 0; still = false;
    if (last < 0) {
      const int cs = s_stop[t];
      if (cs < nct) { last = nr - 1 - (T.r_top - cs); done = cs + 1; }
      else { done = nct; still = true; }
    }
  };
  if (chain_wave) {
    int nstill = 0, deep = 0;
    if (lane < nrays) {
      int last, done; bool still;
      const bool was_open = s_last[lane] < 0;
      outcome(lane, last, done, still);
      if (was_open) {
        if (!still) T.last[w0 + lane] = last;
        if (!SLANT) { T.acc[w0 + lane] = s_acc[0][lane]; T.acc[T.nsh + w0 + lane] = s_acc[1][lane]; }
      }
      nstill = still; if (last >= 0) deep = last + 1;
    }
    nstill = (int)wave_sum_ll(nstill);
    deep = wave_max_i(deep);
    pub_still = nstill; pub_deep = deep;
  }
  for (int k = tid; k < nct * kTailRays; k += kTailThreads) {
    const int c = k / kTailRays, t = k % kTailRays;
    int last, done; bool still;
    outcome(t, last, done, still);
    if (t < nrays && c < done) {
      const int rs = T.r_top - c;
      T.er[(long long)rs * T.nsh + w0 + t] = s_er[c][t];
      T.tau[(long long)(nr - 1 - rs) * T.nsh + w0 + t] = s_tau[c][t];
    }
  }
  // The run's flags as the step kernels' tau_publish leaves them after the plan's last step (the
  // last block to arrive publishes the totals), into device memory and into the pinned block the
  // host reads.  Vertical rays: next to the other waves' emission, not behind it; slant rays: behind
  // the block's modulation, which may raise a status.
  auto publish = [&]() {
    if (pub_still) atomicAdd(&T.flags[1], pub_still);
    if (pub_deep) atomicMax(&T.flags[4], pub_deep);
    __threadfence();
    const int ticket = atomicAdd(&T.flags[3], 1);
    if (ticket == (int)gridDim.x - 1) {
      __threadfence();
      const int act = atomicAdd(&T.flags[1], 0), dp = atomicAdd(&T.flags[4], 0);
      const int status = SLANT ? atomicAdd(T.status, 0) : 0;     // (vertical rays raise none)
      const int swept = T.flags[2] + nct;                    // layers swept so far
      T.flags[2] = swept; T.flags[1] = 0; T.flags[3] = 0;
      __threadfence();
      atomicExch(&T.flags[0], act);
      if (A.host_flags) {
        volatile int *hf = A.host_flags;
        hf[0] = act; hf[1] = 0; hf[2] = swept; hf[3] = 0; hf[4] = dp; hf[5] = 0; hf[6] = 0; hf[7] = 0;
        hf[16] = status; hf[17] = 0; hf[18] = 0; hf[19] = 0;
        __threadfence_system();
      }
    }
  };
  if (!SLANT && chain_wave && lane == 0) {
    if (A.host_blocks) *(int2 *)(A.host_blocks + 2 * (long long)blockIdx.x) = make_int2(pub_still, pub_deep);
    else publish();
  }
  if (!chain_wave && wv < nrays) {
    const long long w = w0 + wv;
    int last, done; bool still;
    outcome(wv, last, done, still);
    last = __builtin_amdgcn_readfirstlane(last); done = __builtin_amdgcn_readfirstlane(done);
    const int i_first = nr - 1 - T.r_top;                  // heights above it: earlier steps (global memory)
    const int i_end = i_first + done;                      // (a ray that was closed on entry left nothing in LDS)
    const double *tau_g = T.tau;
    auto tau_at = [&](int i) { return (i >= i_first && i < i_end) ? s_tau[i - i_first][wv] : tau_g[(long long)i * T.nsh + w]; };
    if constexpr (SLANT) modulation_ray(A.M, w, last, lane, tau_at);
    else {
      // (the tail covers a run from its first layer: at most kTailLayers = 128 heights, two passes of 64)
      const RayPlanck PL(A.E, w, s_e2);
      emission_ray<(NANG > 0 ? NANG : 1)>(A.E, w, last, lane, s_e2, tau_at,
                                          [&](int i0, int i) { return i0 == 0 ? bpre[0] : i0 == 64 ? bpre[1] : PL(i0, i); });
    }
  }
  if (SLANT) {
    if (A.host_blocks) { if (chain_wave && lane == 0) *(int2 *)(A.host_blocks + 2 * (long long)blockIdx.x) = make_int2(pub_still, pub_deep); }
    else {
      __syncthreads();
      if (chain_wave && lane == 0) publish();
    }
  }
}

}  // namespace trx
