// trx_kernels.hip.h -- hand-written gfx950 kernels of the spectrum path.
//
// All of it is HBM/latency-bound fp64 accumulation over a line list -- no
// dense contraction, hence no MFMA.  Wavefront = 64 lanes.
//
//   k_voigt_bins / k_voigt_bins_wave  Voigt-profile table      (opacity.c:219-277, voigt.c)
//   (the line sweep of layers whose profiles reach <= 7 cells is trx_walk.hip.h: k_line_walk)
//   k_group_sweep                     two-kernel form, wider profiles: passes 1+2a, co-added
//                                     group strength and Doppler index (extinction.c:399-483)
//   k_sticky_index                    sticky Doppler index, all layers of a run up front
//                                                               (extinction.c:393, 480-483)
//   k_accumulate                      pass 2b: threshold + profile accumulation into
//                                     e[layer][wn], gather per 4-bin tile (extinction.c:467-509)
//   k_accumulate_wide                 the same for profiles >= 64 coarse bins: lanes own bins,
//                                     phase-major table (k_table_phase_major, which also makes the
//                                     walk's row copy)
//   k_grid_extinction                 opacity-grid mode          (extinction.c:535-581)
//   k_cia_rows/_layers/_eval          CIA extinction            (crosssec.c:272-428)
//   k_slant_geometry                  transit rays: impact parameters, bracket layers, Simpson weights
//                                                               (tau.c:274, slantpath.c:36-95, 399-408)
//   k_optical_depth(_vertical)        total extinction + ray quadrature + toomuch cut
//                                                               (tau.c:216-305, eclipse.c:29-105,
//                                                                slantpath.c:19-108)
//   k_extras_factors                  scattering / cloud models: their wavenumber factors, once per ray
//   k_extras_dump                     scattering / cloud terms alone (savefiles dumps, tau.c:180-190)
//   k_emission / k_modulation         spectrum from tau         (eclipse.c:118-287, slantpath.c:351-473);
//                                     k_emission: one wave per wavenumber, lanes = heights;
//                                     k_emission_rows / k_modulation_rows: one lane per wavenumber
//                                     (grids of more than 65 536 wavenumbers)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "trx_device.h"
#include "../trx_numerics.h"

namespace trx {

// ---------------------------------------------------------------------------
// do two streams run concurrently?
// ---------------------------------------------------------------------------
// The HIP runtime maps a process's streams onto a few hardware queues per device; two streams
// on one queue run their kernels one after the other, whatever the program intended.  trx_create
// probes: k_probe_wait on one stream waits (bounded: ~0.5 ms of the 100 MHz wall clock) for a flag
// that k_probe_set on the other stream raises.  If the waiter saw the flag, the two overlap.
__global__ void k_probe_wait(volatile int *flag, int *saw)
{
  const unsigned long long t0 = wall_clock64();
  int v = 0;
  while ((v = *flag) == 0 && wall_clock64() - t0 < 50000ull) __builtin_amdgcn_s_sleep(8);
  *saw = v;
}
__global__ void k_probe_set(volatile int *flag) { *flag = 1; }

// ---------------------------------------------------------------------------
// wavefront helpers (64 lanes, fixed butterfly order => deterministic sums)
// ---------------------------------------------------------------------------
// Wave maximum of non-negative doubles without LDS traffic: rotate-and-max inside
// the four rows of 16 lanes (DPP row_ror 1,2,4,8 leaves every lane with its row's
// maximum), then the four row results through SGPRs.  Result is wave-uniform.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
// The butterfly sum of a wave: every lane gets the total, partner lane ^ 32, 16, 8, 4, 2, 1 in that order.  The
// four steps inside a row of 16 lanes are DPP moves on the vector pipe instead of lane permutes through the LDS
// pipe (which a block of k_ray_tail's seven emission waves keeps busy): lane ^ 8 is row_ror:8, lane ^ 4 is
// row_shl:4 for the lanes of banks 0 and 2 and row_shr:4 for banks 1 and 3, lane ^ 2 and lane ^ 1 are
// quad_perm -- the same partner in every step, so the same bits as six __shfl_xor.
__device__ __forceinline__ double lane_xor4(double v)
{
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x104, 0xF, 0x5, false);       // row_shl:4 -> lanes of banks 0, 2 read lane + 4
  lo = __builtin_amdgcn_update_dpp(lo, __double2loint(v), 0x114, 0xF, 0xA, false);           // row_shr:4 -> lanes of banks 1, 3 read lane - 4
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x104, 0xF, 0x5, false);
  hi = __builtin_amdgcn_update_dpp(hi, __double2hiint(v), 0x114, 0xF, 0xA, false);
  return __hiloint2double(hi, lo);
}
// v[lane] + v[lane ^ 32] and v[lane] + v[lane ^ 16] in every lane without the LDS pipe: gfx950's v_permlane32_swap
// exchanges the upper half of one register with the lower half of another (v_permlane16_swap: the odd rows of 16
// with the even ones) -- two copies of v go in, (lower halves twice, upper halves twice) come out, and their sum is
// the butterfly step.  (The upper lanes add their partner FIRST: the same bits, the addition commutes.)
__device__ __forceinline__ double swap_sum32(double v)
{
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
__device__ __forceinline__ double swap_sum16(double v)
{
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
__device__ __forceinline__ double wave_sum(double v)
{
  v = swap_sum32(v);           // lane ^ 32
  v = swap_sum16(v);           // lane ^ 16
  v += dpp_f64<0x128>(v);      // row_ror:8 = lane ^ 8
  v += lane_xor4(v);
  v += dpp_f64<0x4E>(v);       // quad_perm [2,3,0,1]: lane ^ 2
  v += dpp_f64<0xB1>(v);       // quad_perm [1,0,3,2]: lane ^ 1
  return v;
}
__device__ __forceinline__ double readlane_f64(double v, int l)
{
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wave_max(double v)
{
  v = fmax(v, dpp_f64<0x121>(v));      // row_ror:1
  v = fmax(v, dpp_f64<0x122>(v));      // row_ror:2
  v = fmax(v, dpp_f64<0x124>(v));      // row_ror:4
  v = fmax(v, dpp_f64<0x128>(v));      // row_ror:8
  return fmax(fmax(readlane_f64(v, 0), readlane_f64(v, 16)), fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}
__device__ __forceinline__ int wave_max_i(int v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
  return v;
}
// Workgroups are dealt to the 8 XCDs round-robin (linear id mod 8) and each XCD has its own
// L2.  Tiles next to each other read overlapping windows of the group arrays (and the same
// profile rows), so out of every 8*G consecutive tile blocks of a layer each XCD gets G
// CONSECUTIVE ones.  (One contiguous eighth of the layer per XCD was measured too: the
// stopped-ray tile skipping is spectrally clustered and then unbalances the XCDs.)
// Needs gridDim.x to be a multiple of 8*G (the host pads; surplus blocks find no tile), so
// that blockIdx.x mod 8 is the XCD for every blockIdx.y.
constexpr int kXcds = 8;
constexpr int kAccumXcdGroup = 4;   // tile blocks (of 4 tiles) per XCD run in k_accumulate
template <int G>
__device__ __forceinline__ int xcd_grouped_x()
{
  const int x = (int)blockIdx.x, span = kXcds * G;
  return (x / span) * span + (x % kXcds) * G + (x % span) / kXcds;
}

// Latency chains on a handful of waves (optical depth, CIA splines, the spectrum) share the
// machine with the line sweep of the NEXT step, which keeps every SIMD's issue slots busy:
// at equal priority a chain gets one issue slot in seven.  Raised issue priority lets its
// instructions go first; the sweep loses almost nothing (the chain is a few dozen waves).
__device__ __forceinline__ void latency_critical() { __builtin_amdgcn_s_setprio(3); }

// Gate of the line-sweep kernels: rays still descending (the device counter flags[0],
// maintained by the optical-depth kernel of the step before).
__device__ __forceinline__ bool sweep_active(const int *flags, int eager)
{
  return eager || flags[0] != 0;
}

// item t of the flat iteration space over `n` contiguous runs: run s with base[s] <= t < base[s+1]
// (runs may be empty), mapped to start[s] + (t - base[s]); -1 past the end
__device__ __forceinline__ long long seg_index(const long long *start, const long long *base, int n, long long t)
{
  if (t >= base[n]) return -1;
  int a = 0, z = n - 1;              // last s with base[s] <= t
  while (a < z) { const int m = (a + z + 1) >> 1; if (base[m] <= t) a = m; else z = m - 1; }
  return start[a] + (t - base[a]);
}
__device__ __forceinline__ long long wave_sum_ll(long long v)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---------------------------------------------------------------------------
// Voigt table
// ---------------------------------------------------------------------------
__constant__ double c_voigt_coef[64];     // 1/(n!(2n+1)), voigt.c:45-108

constexpr double kTwoOSqrtPi = 1.12837916709551257389;
constexpr double kSqrtLn2Pi  = 0.46971863934982566689;

// voigt.c:132-200 (voigtxy).  The reference runs the Region-I recurrence in
// long double; double carries ~1e-12 relative error through its worst
// cancellation (x -> 3), far below the float the result is rounded to.
__device__ __forceinline__ float voigt_point(double x, double y, double alphaD)
{
  const double x2y2 = x*x - y*y;
  const double xy2  = 2*x*y;
  if (x < 3 && y < 1.8) {
    double sn, cs;
    sincos(xy2, &sn, &cs);
    const int nterm = (x < 1 ? 15 : (int)(6.842*x + 8.0)) + 1;
    double pr = y, pi = -x, sr = y, si = -x;
    for (int k = 1; k <= nterm; k++) {
      const double ti = pr*xy2 + pi*x2y2;
      const double tr = pr*x2y2 - pi*xy2;
      si += ti * c_voigt_coef[k];
      sr += tr * c_voigt_coef[k];
      pi = ti; pr = tr;
    }
    return (float)(kSqrtLn2Pi/alphaD * exp(-x2y2) *
                   (cs*(1 - sr*kTwoOSqrtPi) - sn*si*kTwoOSqrtPi));
  }
  const double q = xy2*xy2, p = xy2*x;
  if (x < 5 && y < 5) {
    const double t1 = x2y2 - 0.19016350, t2 = x2y2 - 1.78449270, t3 = x2y2 - 5.52534370;
    return (float)(kSqrtLn2Pi/alphaD * (0.46131350 *((p - t1*y)/(t1*t1 + q)) +
                                        0.09999216 *((p - t2*y)/(t2*t2 + q)) +
                                        0.002883894*((p - t3*y)/(t3*t3 + q))));
  }
  const double t1 = x2y2 - 0.27525510, t2 = x2y2 - 2.72474500;
  return (float)(kSqrtLn2Pi/alphaD * (0.51242424*((p - t1*y)/(t1*t1 + q)) +
                                      0.05176536*((p - t2*y)/(t2*t2 + q))));
}

__device__ __forceinline__ float voigt_at(const ProfileJob &J, double yv, long long i)
{
  return voigt_point(kSqrtLn2 * fabs(J.sub * (double)i - J.half) / J.alphaD, yv, J.alphaD);
}

// One thread per output bin; blockIdx.y selects the profile (voigt.c:369-483).
// The float accumulation order of the two averaging rules (:489-554) is kept.
__global__ __launch_bounds__(256)
void k_voigt_bins(const ProfileJob *jobs, float *table, int m_limit)
{
  const ProfileJob J = jobs[blockIdx.y];
  if (J.regime == 2 && J.m > m_limit) return;          // done by k_voigt_bins_wave
  const double yv = kSqrtLn2 * J.alphaL / J.alphaD;
  for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < J.nv;
       k += (long long)gridDim.x * blockDim.x) {
    float r;
    if (J.regime == 0) {
      r = voigt_at(J, yv, k);
    } else if (J.regime == 1) {
      const float a = voigt_at(J, yv, k), b = voigt_at(J, yv, k + 1);
      float s = 0;
      s = (float)((s + (a + b) / 2.0) / (double)1);
      r = s;
    } else {
      const int m = J.m;
      const long long base = k * m;
      float s = 0;
      for (int i = 1; i < m; i += 2) s += voigt_at(J, yv, base + i);
      s *= 2;
      for (int i = 2; i < m; i += 2) s += voigt_at(J, yv, base + i);
      s *= 2;
      s += voigt_at(J, yv, base) + voigt_at(J, yv, base + m);
      s = (float)(s / (m * 3.0));
      r = s;
    }
    table[J.off + k] = r;
  }
}

// Coarse grids: thousands of sub-intervals per bin.  One wavefront per bin: the
// 64 lanes evaluate 64 points at a time, lane 0 adds them in the reference order.
__global__ __launch_bounds__(64)
void k_voigt_bins_wave(const ProfileJob *jobs, float *table, int m_limit)
{
  const ProfileJob J = jobs[blockIdx.y];
  if (J.regime != 2 || J.m <= m_limit) return;
  __shared__ float buf[64];
  const int lane = threadIdx.x;
  const double yv = kSqrtLn2 * J.alphaL / J.alphaD;
  const int m = J.m;
  for (long long k = blockIdx.x; k < J.nv; k += gridDim.x) {
    const long long base = k * m;
    float s = 0;
    for (int pass = 0; pass < 2; pass++) {             // odd points, then even points
      for (int i0 = 1 + pass; i0 < m; i0 += 128) {
        const int i = i0 + 2*lane;
        buf[lane] = (i < m) ? voigt_at(J, yv, base + i) : 0.f;
        __syncthreads();
        if (lane == 0) {
          const int cnt = min(64, (m - i0 + 1) / 2);
          for (int t = 0; t < cnt; t++) s += buf[t];
        }
        __syncthreads();
      }
      if (lane == 0) s *= 2;
    }
    if (lane == 0) {
      s += voigt_at(J, yv, base) + voigt_at(J, yv, base + m);
      s = (float)(s / (m * 3.0));
      table[J.off + k] = s;
    }
  }
}

// ---------------------------------------------------------------------------
// line sweep, strength part (two-kernel form, profiles wider than the walk's window)
// ---------------------------------------------------------------------------
// extinction.c:429-483 (co-added group strength, Doppler-width index).  One lane per LINE
// (uniform work: two exponentials per line and layer); the lanes that anchor a co-added group
// (anchor line + the following lines of the same isotope inside one fine bin, built on the
// host) then add their members' strengths out of LDS.  Per (layer, group) it emits
//     SG    = sum_members gf*exp(-c*Elow/T)*(1-exp(-c*wn/T)) * SIGCTE*ratio/(m*Z)
//     idop8 = nearest Doppler-width index, or 0xFF = "use the isotope's sticky one"
// The threshold test against ethresh*kmax (:467) is applied where SG is consumed
// (k_accumulate); kmax itself comes from k_layer_max (trx_walk.hip.h).
// The nearest Doppler-width index of pu/src/iomisc.c:1088-1108 (nearest_index in
// trx_numerics.h) is a monotone step function of the width; trx_create finds its
// steps exactly (thr[k] = smallest double whose index is >= k, thr[0] = -inf,
// thr[n] = +inf), so that index(v) = the k with thr[k] <= v < thr[k+1].  Walked from
// the previous layer's answer: two compares in the common case.
__device__ __forceinline__ int index_from(const double *thr, double v, int lo)
{
  while (v >= thr[lo + 1]) lo++;
  while (v < thr[lo]) lo--;
  return lo;
}

// exp(x) for x <= 0, the only arguments the line strengths have (-c*Elow/T, -c*wn/T):
//   x = (64 m + j) ln2/64 + r, |r| <= ln2/128;  exp(x) = 2^m * 2^(j/64) * e^r
// with 2^(j/64) from a 64-entry table (LDS) and e^r by a degree-5 polynomial (truncation
// 3e-17).  16 instructions instead of the 29 of the general-range library routine; ~1.5 ulp.
// Underflow goes through ldexp (denormals, then 0) like the library's.
__device__ __forceinline__ double exp_neg(double x, const double *e2tab)
{
  const double kd = __builtin_rint(x * 0x1.71547652b82fep+6);           // 64/ln2
  double r = __builtin_fma(-kd, 0x1.62e42fee00000p-7, x);               // ln2/64, high 32 bits: exact product
  r = __builtin_fma(-kd, 0x1.a39ef35793c76p-39, r);                     // ln2/64, rest
  const int ki = (int)kd;
  const double t = e2tab[ki & 63];
  double p = 0x1.1111111111111p-7;                                      // 1/120
  p = __builtin_fma(p, r, 0x1.5555555555555p-5);                        // 1/24
  p = __builtin_fma(p, r, 0x1.5555555555555p-3);                        // 1/6
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return ldexp(t * p, ki >> 6);
}

// the same with the coefficient 1/24 handed in (a vector register the caller keeps: the first step of the
// polynomial takes two constants, and only one operand of an instruction can come from the scalar side --
// the compiler otherwise copies 1/24 into vector registers in front of every evaluation)
__device__ __forceinline__ double exp_neg(double x, const double *e2tab, double c24)
{
  const double kd = __builtin_rint(x * 0x1.71547652b82fep+6);           // 64/ln2
  double r = __builtin_fma(-kd, 0x1.62e42fee00000p-7, x);               // ln2/64, high 32 bits: exact product
  r = __builtin_fma(-kd, 0x1.a39ef35793c76p-39, r);                     // ln2/64, rest
  const int ki = (int)kd;
  const double t = e2tab[ki & 63];
  double p = __builtin_fma(0x1.1111111111111p-7, r, c24);               // 1/120, 1/24
  p = __builtin_fma(p, r, 0x1.5555555555555p-3);                        // 1/6
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return ldexp(t * p, ki >> 6);
}

constexpr int kSweepIsoLds = 8;          // isotopes whose per-layer scalars are staged in LDS

// The lines of every isotope block that can reach the shard (SweepWindow): one flat iteration space.
__global__ __launch_bounds__(256)
void k_group_sweep(LinesDev L, LayerDev Y, SweepWindow Wn, int niso, int r_top, int nc,
                   const double *__restrict__ dthr, int ndop,   // steps of the Doppler index, [ndop + 1]
                   const double *__restrict__ e2tab,            // 2^(j/64), j = 0..63
                   const double *__restrict__ wcut,     // [layer][iso] refresh <=> wavn >= wcut
                   double *__restrict__ SG, uint8_t *__restrict__ idop8,
                   const int *__restrict__ flags, int eager)
{
  if (!sweep_active(flags, eager)) return;
  __shared__ double s_thr[kMaxDop + 1];
  __shared__ double s_e2[64];
  __shared__ double s_ct[kMaxChunk];
  __shared__ double s_f[kMaxChunk][kSweepIsoLds], s_wc[kMaxChunk][kSweepIsoLds], s_ad[kMaxChunk][kSweepIsoLds];
  __shared__ double s_s[2][256];                         // line strengths of the layer in flight
  extern __shared__ long long s_runs[];                  // [niso] first line, [niso + 1] running count (dynamic: 16 niso + 8 bytes)
  long long *s_start = s_runs, *s_base = s_runs + niso;
  for (int i = threadIdx.x; i <= ndop; i += 256) s_thr[i] = dthr[i];
  if (threadIdx.x < 64) s_e2[threadIdx.x] = e2tab[threadIdx.x];
  if (threadIdx.x < nc) s_ct[threadIdx.x] = Y.negc_over_t[r_top - threadIdx.x];
  const int nst = min(niso, kSweepIsoLds);
  for (int i = threadIdx.x; i < nc * nst; i += 256) {
    const int c = i / nst, b = i - c * nst, ri = (r_top - c) * niso + b;
    s_f[c][b] = Y.strength_f[ri]; s_wc[c][b] = wcut[ri]; s_ad[c][b] = Y.alphad[ri];
  }
  // the runs of lines: block b -> [la, lz) (the same arithmetic as the host's launch size, sweep_chunk)
  const long long t0 = (long long)blockIdx.x * 256;
  for (int b = threadIdx.x; b < niso; b += 256) {
    const int gb0 = L.gblock[b], gb1 = L.gblock[b + 1];
    long long la = 0, lz = 0;
    if (gb0 != gb1) {
      int ga = gb0, gz = gb1;
      if (Wn.windowed) {
        long long psm = 0;
        for (int c = 0; c < nc; c++) psm = max(psm, (long long)Y.psmax[(long long)(r_top - c) * niso + b]);
        const long long lo_f = (long long)Wn.osamp * Wn.lo - psm;
        const long long klo = lo_f > 0 ? lo_f / Wn.osamp : 0;
        long long khi = ((long long)Wn.osamp * (Wn.hi - 1) + psm) / Wn.osamp;
        if (khi > Wn.nwn - 1) khi = Wn.nwn - 1;
        const int32_t *cg = L.cnt_ge + (long long)b * (Wn.nwn + 1);
        ga = gb0 + cg[khi + 1]; gz = gb0 + cg[klo];
      }
      if (ga < gz) { la = L.gfirst[ga]; lz = (long long)L.gfirst[gz - 1] + L.gcount[gz - 1]; }
    }
    s_start[b] = la; s_base[b + 1] = lz - la;              // (lengths first; the running sum below)
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    long long tot = 0;
    for (int b = 0; b < niso; b++) { const long long len = s_base[b + 1]; s_base[b] = tot; tot += len; }
    s_base[niso] = tot;
  }
  __syncthreads();
  const long long ln = niso > 0 ? seg_index(s_start, s_base, niso, t0 + threadIdx.x) : -1;
  const bool ok = ln >= 0;
  int g = -1, cnt = 0, iso = 0;
  double wavn = 0, elow = 0, gf = 0;
  if (ok) {
    wavn = L.wavn[ln]; elow = L.elow[ln]; gf = L.gf[ln];
    iso = L.iso[ln]; g = L.lgroup[ln];
    if (g >= 0) cnt = L.gcount[g];
  }
  // members that sit in this block's LDS window: lines ln+1 .. ln+cnt-1 are consecutive
  // lanes as long as they stay inside the block AND inside the same segment; the anchor
  // and its members always share a segment (segments are cut at group boundaries)
  int in_lds = cnt;
  if (g >= 0 && (int)threadIdx.x + cnt > 256) in_lds = 256 - (int)threadIdx.x;
  const bool all_staged = niso <= kSweepIsoLds;             // kernel-uniform: plain LDS reads below
  const bool staged = iso < kSweepIsoLds;
  int lo = 0;                                               // Doppler index carried across layers
  __syncthreads();
  for (int c = 0; c < nc; c++) {
    const int r = r_top - c;
    const double ct = s_ct[c];
    const int ri = r * niso + iso;
    double s = 0;
    if (ok) s = gf * exp_neg(ct * elow, s_e2) * (1 - exp_neg(ct * wavn, s_e2));
    s_s[c & 1][threadIdx.x] = s;
    __syncthreads();                                        // (two buffers: one barrier per layer)
    if (g >= 0) {
      double f;
      if (all_staged) f = s_f[c][iso];
      else f = staged ? s_f[c][iso] : Y.strength_f[ri];
      double pk = s;
      for (int m = 1; m < in_lds; m++) pk += s_s[c & 1][threadIdx.x + m];
      for (int m = in_lds; m < cnt; m++) {                  // members beyond the block: recompute
        const long long lm = ln + m;
        pk += L.gf[lm] * exp_neg(ct * L.elow[lm], s_e2) * (1 - exp_neg(ct * L.wavn[lm], s_e2));
      }
      pk *= f;
      double wc, ad;
      if (all_staged) { wc = s_wc[c][iso]; ad = s_ad[c][iso]; }
      else { wc = staged ? s_wc[c][iso] : wcut[ri]; ad = staged ? s_ad[c][iso] : Y.alphad[ri]; }
      uint8_t id = 0xFF;
      if (wavn >= wc) {
        lo = index_from(s_thr, ad * wavn, lo);
        id = (uint8_t)lo;
      }
      SG[(long long)c * L.ngroups + g] = pk;
      idop8[(long long)c * L.ngroups + g] = id;
    }
  }
}

// Sticky Doppler index of every (layer, isotope): extinction.c:393 and :480-483.
// idop[i] is refreshed by every evaluated line with alphad*wn/alphal >= 0.1 and
// keeps its last value for the others.  Inside an isotope block wavenumbers
// descend, so the refreshing lines are a prefix of the block (npre groups,
// found on the host) and the index every other line sees is the one of the
// LAST evaluated group of that prefix.  One wavefront per (layer, isotope)
// walks back from the end of the prefix, 64 candidates at a time, evaluating
// their strengths exactly as pass 1/2a do.
__global__ __launch_bounds__(64)
void k_sticky_index(LinesDev L, LayerDev Y, int niso, int r_top, int nc,
                    const double *__restrict__ kmax /* [layer][nmx] */, int nmx, const int32_t *__restrict__ iso_mx, double ethresh,
                    const double *__restrict__ dthr, int ndop,   // the steps of the nearest-Doppler-index function (index_from), [ndop + 1]
                    const double *__restrict__ e2tab,    // 2^(j/64): the same exp as the sweep => the same decisions
                    const int *__restrict__ npre,        // [layer][iso] refreshing groups
                    int *__restrict__ sticky_idop,       // [layer][iso]
                    const int *__restrict__ flags, int eager,
                    const uint4 *__restrict__ copy_src = nullptr, uint4 *__restrict__ copy_dst = nullptr, long long copy_n = 0, int copy_from = 0)
{
  // blocks copy_from ...: the run's input block from the pinned host memory the kernels of the front end read it from
  // (Y, npre here) into device memory, where every later kernel of the run reads it -- 16 bytes per lane and trip; no
  // copy engine, no wait of a kernel for a copy's completion signal
  if (copy_n > 0 && (int)blockIdx.x >= copy_from) {
    const long long stride = (long long)(gridDim.x - copy_from) * 64;
    for (long long i = (long long)((int)blockIdx.x - copy_from) * 64 + threadIdx.x; i < copy_n; i += stride) copy_dst[i] = copy_src[i];
    return;
  }
  if (!sweep_active(flags, eager)) return;
  __shared__ double s_e2[64];
  s_e2[threadIdx.x] = e2tab[threadIdx.x];
  __builtin_amdgcn_wave_barrier();
  const int c = blockIdx.x / niso, b = blockIdx.x - c * niso;
  if (c >= nc) return;
  const int r = r_top - c, ri = r * niso + b, lane = threadIdx.x;
  const int gb0 = L.gblock[b];
  const double ct = Y.negc_over_t[r], lim = ethresh * kmax[(long long)r * nmx + (nmx == 1 ? 0 : iso_mx[b])], f = Y.strength_f[ri];
  // (what the end of this one-wave chain needs, asked for at its start: the layer's own index and width, and the index
  // function's steps 1 .. 64, one per lane -- the bisection of nearest_index was six dependent round trips of one lane)
  const int id0 = Y.idop0[ri]; const double ad = Y.alphad[ri];
  const double thr_mine = 1 + lane < ndop ? dthr[1 + lane] : 0.0;
  int found = -1;
  for (int base = npre[ri] - 1; base >= 0 && found < 0; base -= 64) {
    const int k = base - lane;
    bool ev = false;
    if (k >= 0) {
      const int g = gb0 + k, first = L.gfirst[g], cnt = L.gcount[g];
      double pk = 0;
      for (int m = 0; m < cnt; m++) {
        const int ln = first + m;
        const double s = L.gf[ln] * exp_neg(ct * L.elow[ln], s_e2) * (1 - exp_neg(ct * L.wavn[ln], s_e2));
        pk = (m == 0) ? s : pk + s;
      }
      pk *= f;
      ev = !(pk < lim);
    }
    const unsigned long long mask = __ballot(ev);
    if (mask) found = base - (__ffsll((long long)mask) - 1);
  }
  int id = id0;
  if (found >= 0) {                                          // (wave-uniform)
    // the index of the width: the number of steps 1 .. ndop - 1 at or below it (index_from's definition)
    const double v = ad * L.gwavn[gb0 + found];
    id = __builtin_popcountll(__ballot(1 + lane < ndop && v >= thr_mine));
    for (int k0 = 65; k0 < ndop; k0 += 64) id += __builtin_popcountll(__ballot(k0 + lane < ndop && v >= dthr[min(k0 + lane, ndop)]));
  }
  if (lane == 0) sticky_idop[ri] = id;
}

// sum per-block partial counters: out[slot] += sum_k parts[slot*nparts + k]
__global__ __launch_bounds__(256)
void k_sum_parts_gated(const unsigned long long *__restrict__ parts, int nparts, int stride, int offset,
                       unsigned long long *__restrict__ out, int out_stride, int r_top,
                       const int *__restrict__ flags, int eager)
{
  if (!eager && flags[0] == 0) return;
  __shared__ unsigned long long red[256];
  const int slot = blockIdx.x;                       // layer of the chunk
  unsigned long long s = 0;
  for (int k = threadIdx.x; k < nparts; k += 256) s += parts[((long long)slot * nparts + k) * stride + offset];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) out[(long long)(r_top - slot) * out_stride + offset] += red[0];
}

// ---------------------------------------------------------------------------
// pass 2b: profile accumulation (gather form, deterministic)
// ---------------------------------------------------------------------------
// extinction.c:485-509.  The reference scatters every line into the bins its
// profile covers; bin j receives a term iff |osamp*j - iown| <= profsize (the
// minj/maxj/beg_j arithmetic of :486-509 reduces to exactly that set).  Here
// one wavefront owns a tile of kTileBins coarse bins of one layer, its lanes
// stride over the groups whose window can reach the tile (contiguous per
// isotope block because the TLI is wavelength-sorted), each lane keeps private
// partial sums, and a shuffle reduction closes the line sum.
struct AccumArgs {
  LinesDev L;
  LayerDev Y;
  int niso, nlor, ndop, osamp;
  long long nwn, lo, nsh;           // full grid, shard origin, shard bins
  int r_top, nc, ntiles;
  const double  *SG;                // [chunk][ngroups] group strength before threshold and density
  const uint8_t *idop8;             // [chunk][ngroups]
  const double  *kmaxc;             // [layer][nmx] strongest single line (k_layer_max)
  double ethresh;
  int nmx; const int32_t *iso_mx;   // per-molecule sweeps: output slot of every isotope (else 1 / null)
  int permol;                       // 1: no density factor, one output row per slot (extinction.c:472, 507)
  const int     *sticky_idop;       // [layer][iso]
  const int32_t *psize;             // [ndop][nlor]
  const long long *poff;            // [ndop][nlor]
  const float   *table;
  double        *e;                 // [layer][nsh]
  unsigned long long *part;         // [nc][part_stride][3] {bins, evaluated, skipped} or null (profiling)
  int part_stride;
  const int *flags;
  const int *last;                  // [nsh] or null: skip tiles whose rays all stopped above this chunk
  int eager;
  unsigned skip_mask;               // bit c set: layer c of the chunk belongs to k_accumulate_wide
  int sub_f; const int32_t *cnt_sub;  // [niso][sub_f*nwn + 1]: groups with iown*sub_f/osamp >= k (sub_f == 1: cnt_ge)
};

// grid: x = groups of 4 tiles, y = layer of the chunk
__global__ __launch_bounds__(256)
void k_accumulate(AccumArgs A)
{
  if (!A.eager && A.flags[0] == 0) return;
  __shared__ int32_t  s_ps[4][kMaxDop];
  __shared__ long long s_po[4][kMaxDop];
  __shared__ long long s_nb[4][3];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = blockIdx.y, bx = xcd_grouped_x<kAccumXcdGroup>();
  if ((A.skip_mask >> c) & 1u) return;
  const int tile = bx * 4 + wv;
  const int r = A.r_top - c;
  bool live = tile < A.ntiles;                       // wave-uniform
  const long long j0 = A.lo + (long long)tile * kTileBins;      // global coarse bin
  const long long j1 = min(j0 + kTileBins, A.lo + A.nsh) - 1;
  if (live && A.last) {
    // every ray of the tile already crossed toomuch (tau.c:277-287): nothing
    // below will ever read e[layer][j] for these bins
    const bool open = (lane <= j1 - j0) && (A.last[(j0 - A.lo) + lane] < 0);
    live = __ballot(open) != 0ull;
  }
  double acc[kTileBins];
#pragma unroll
  for (int t = 0; t < kTileBins; t++) acc[t] = 0.0;
  long long nb = 0, nev = 0, nsk = 0;
  // counters: a group is counted by the tile that holds its own coarse bin
  const long long hk0 = j0, hk1 = (j1 == A.nwn - 1) ? (long long)1 << 60 : j1;
  int cur_mx = -1;                                    // output slot being accumulated
  auto flush = [&](int mx) {
#pragma unroll
    for (int t = 0; t < kTileBins; t++) {
      const double s = wave_sum(acc[t]);
      if (lane == 0 && j0 + t <= j1) A.e[((long long)r * A.nmx + mx) * A.nsh + (j0 - A.lo) + t] = s;
      acc[t] = 0.0;
    }
  };

  if (live)
  for (int b = 0; b < A.niso; b++) {
    const int gb0 = A.L.gblock[b], gb1 = A.L.gblock[b + 1];
    if (gb0 == gb1) continue;
    const int mx = A.nmx == 1 ? 0 : A.iso_mx[b];
    if (mx != cur_mx) { if (cur_mx >= 0) flush(cur_mx); cur_mx = mx; }
    const double lim = A.ethresh * A.kmaxc[(long long)r * A.nmx + mx];
    const int ri = r * A.niso + b;
    const int il = A.Y.ilor[ri];
    // profile column of this (layer, isotope): size and offset per Doppler index
    for (int i = lane; i < A.ndop; i += 64) {
      s_ps[wv][i] = A.psize[i * A.nlor + il];
      s_po[wv][i] = A.poff [i * A.nlor + il];
    }
    const int idst = A.sticky_idop[ri];
    // window of groups that can reach the tile
    const long long psm = A.Y.psmax[ri];
    // groups with iown in [osamp*j0 - psm, osamp*j1 + psm]; keys are sub-buckets iown*F/osamp
    const long long lo_f = (long long)A.osamp * j0 - psm, hi_f = (long long)A.osamp * j1 + psm;
    const long long nkey = (long long)A.sub_f * A.nwn;
    long long klo = lo_f > 0 ? lo_f * A.sub_f / A.osamp : 0;
    long long khi = hi_f * A.sub_f / A.osamp;
    if (khi > nkey - 1) khi = nkey - 1;
    const int32_t *cg = A.cnt_sub + (long long)b * (nkey + 1);
    const int ga = gb0 + cg[khi + 1], gz = gb0 + cg[klo];
    const double  *SGr = A.SG    + (long long)c * A.L.ngroups;
    const uint8_t *idr = A.idop8 + (long long)c * A.L.ngroups;
    __builtin_amdgcn_wave_barrier();
    const double dens = A.permol ? 1.0 : A.Y.density[ri];
    // four groups per lane and trip: their streamed operands (13 B each) are requested
    // together, so a trip costs one memory round trip instead of four
    constexpr int U = 4;
    for (int gb = ga; gb < gz; gb += 64 * U) {
      double sgv[U]; int iov[U], idv[U]; bool in[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int g = gb + 64 * u + lane;
        in[u] = g < gz;
        const int gc = in[u] ? g : gz - 1;
        sgv[u] = SGr[gc]; iov[u] = A.L.giown[gc]; idv[u] = idr[gc];
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (!in[u]) continue;
        double sg_k = sgv[u];
        const int iown = iov[u];
        const bool below = sg_k < lim;                       // extinction.c:467
        if (A.part) {
          const long long hk = iown / A.osamp;
          if (hk >= hk0 && hk <= hk1) { if (below) nsk++; else nev++; }
        }
        if (below) continue;
        if (!A.permol) sg_k *= dens;                         // extinction.c:472-473
        int id = idv[u];
        if (id == 0xFF) id = idst;
        const long long ps = s_ps[wv][id];
        const long long d0 = (long long)A.osamp * j0 - iown;
        // bins of the tile inside the profile: -ps <= d0 + t*osamp <= ps
        bool ok[kTileBins]; bool any = false;
#pragma unroll
        for (int t = 0; t < kTileBins; t++) {
          const long long d = d0 + (long long)t * A.osamp;
          ok[t] = d >= -ps && d <= ps && j0 + t <= j1;
          any |= ok[t];
        }
        if (!any) continue;
        const float *prof = A.table + s_po[wv][id] + ps;           // centre of the profile
        float pv[kTileBins];
#pragma unroll
        for (int t = 0; t < kTileBins; t++) pv[t] = prof[ok[t] ? d0 + (long long)t * A.osamp : 0];
#pragma unroll
        for (int t = 0; t < kTileBins; t++)
          if (ok[t]) { acc[t] = __builtin_fma(sg_k, (double)pv[t], acc[t]); nb++; }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (live && cur_mx >= 0) flush(cur_mx);
  if (A.part) {
    nb = wave_sum_ll(nb); nev = wave_sum_ll(nev); nsk = wave_sum_ll(nsk);
    if (lane == 0) { s_nb[wv][0] = nb; s_nb[wv][1] = nev; s_nb[wv][2] = nsk; }
    __syncthreads();
    if (threadIdx.x < 3)
      A.part[((long long)c * A.part_stride + bx) * 3 + threadIdx.x] =
          (unsigned long long)(s_nb[0][threadIdx.x] + s_nb[1][threadIdx.x] + s_nb[2][threadIdx.x] + s_nb[3][threadIdx.x]);
  }
}

// ---------------------------------------------------------------------------
// pass 2b for wide profiles (fine output grids): lanes own bins
// ---------------------------------------------------------------------------
// When a profile covers >= ~64 coarse bins (Delta_wn << line width: the
// high-resolution regime, BASELINE configs[4]) the gather above would revisit
// every group for thousands of 4-bin tiles.  Here a wavefront owns 256
// consecutive bins, four consecutive bins per lane, and walks the groups of its
// window in line order, 64 at a time:
//   phase A (one lane per group): threshold test, profile lookup, the group's
//     valid bin range inside the tile [ta, tb] and the table offset of the tile's
//     first bin -- all the integer arithmetic of extinction.c:476-501, once per
//     group; survivors are compacted into LDS in line order;
//   phase B (lanes = bins): per surviving group ONE 16-byte table load per lane
//     and four multiply-adds; groups whose profile covers the whole tile (almost
//     all of them when profiles are much wider than 256 bins) take no masks.
// Bin j reads profile entry q = osamp*j - iown + ps = osamp*(j + kb) + phase, so
// the table is kept phase-major ("tabT": [phase][k], k = q / osamp): a fixed
// group walks ONE row with unit stride instead of striding by osamp floats, and
// neighbouring groups (sorted by wavenumber) re-read nearly the same row segment
// out of L1.  Lanes load their four floats unconditionally -- the tables are
// padded by kTabPad zeros at both ends -- and what lies outside [ta, tb] is masked.
// Each bin is owned by one lane => plain stores, sums in the reference's own
// (line) order.
constexpr int kWideM = 4;
constexpr int kWideT = 64 * kWideM;
static_assert(kWideT <= kTabPad, "table padding must cover one wide tile");

struct WideArgs {
  AccumArgs A;
  const float     *tabT;            // phase-major copy of the table (== table when osamp == 1)
  const long long *poffT;           // [ndop][nlor]
  const int32_t   *gimod, *gidiv;   // [ngroups] iown % osamp, iown / osamp
  unsigned layer_mask;              // bit c set: layer c of the chunk is done here (clear: k_accumulate)
};

struct alignas(4) WideQuad { float v[kWideM]; };   // kWideM = 8 was measured: 1.6x slower (registers, fewer waves)

__device__ __forceinline__ WideQuad wide_load(const float *p)
{
  WideQuad q;
  __builtin_memcpy(&q, p, sizeof(q));      // 4-byte aligned 16-byte loads (global_load_dwordx4)
  return q;
}

__device__ __forceinline__ void wide_apply(double (&acc)[kWideM], double sg, const WideQuad &q, int tt, int t0)
{
  const int ta = tt & 0xFFFF, tb = tt >> 16;
  if (ta == 0 && tb == kWideT - 1) {                         // wave-uniform: the profile covers the tile
#pragma unroll
    for (int m = 0; m < kWideM; m++) acc[m] = __builtin_fma(sg, (double)q.v[m], acc[m]);     // (fused, as in the walk)
  } else {
#pragma unroll
    for (int m = 0; m < kWideM; m++)
      if (t0 + m >= ta && t0 + m <= tb) acc[m] = __builtin_fma(sg, (double)q.v[m], acc[m]);
  }
}

__global__ __launch_bounds__(256)
void k_accumulate_wide(WideArgs W)
{
  const AccumArgs &A = W.A;
  if (!A.eager && A.flags[0] == 0) return;
  const int c = blockIdx.y, bx = (int)blockIdx.x;
  if (!((W.layer_mask >> c) & 1u)) return;
  __shared__ int32_t   s_K[4][kMaxDop], s_psd[4][kMaxDop], s_psm[4][kMaxDop], s_r2[4][kMaxDop];
  __shared__ long long s_po[4][kMaxDop];
  __shared__ double    s_sg[4][64];
  __shared__ long long s_off[4][64];
  __shared__ int32_t   s_tt[4][64];
  __shared__ long long s_nb[4][3];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int tile = bx * 4 + wv;
  const int ntiles = (int)((A.nsh + kWideT - 1) / kWideT);
  const int r = A.r_top - c;
  bool live = tile < ntiles;
  const long long j0 = A.lo + (long long)tile * kWideT;
  const long long j1 = min(j0 + kWideT, A.lo + A.nsh) - 1;
  const int jcount = (int)(j1 - j0 + 1);                     // bins of this tile (< kWideT in the last one)
  const int t0 = kWideM * lane;                              // this lane's first bin, tile-relative
  if (live && A.last) {
    bool open = false;
#pragma unroll
    for (int m = 0; m < kWideM; m++) open |= (t0 + m < jcount) && (A.last[j0 - A.lo + t0 + m] < 0);
    live = __ballot(open) != 0ull;
  }
  double acc[kWideM];
#pragma unroll
  for (int m = 0; m < kWideM; m++) acc[m] = 0.0;
  long long nb = 0, nev = 0, nsk = 0;
  const int of = A.osamp;
  int cur_mx = -1;
  auto flush = [&](int mx) {
    double *dst = A.e + ((long long)r * A.nmx + mx) * A.nsh + (j0 - A.lo) + t0;
#pragma unroll
    for (int m = 0; m < kWideM; m++) {
      if (t0 + m < jcount) dst[m] = acc[m];
      acc[m] = 0.0;
    }
  };

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
    for (int i = lane; i < A.ndop; i += 64) {
      const int ps = A.psize[i * A.nlor + il];
      s_K [wv][i] = (2 * ps) / of + 1;
      s_r2[wv][i] = (2 * ps) % of;
      s_psd[wv][i] = ps / of;
      s_psm[wv][i] = ps % of;
      s_po[wv][i] = W.poffT[i * A.nlor + il];
    }
    const int idst = A.sticky_idop[ri];
    const long long psm = A.Y.psmax[ri];
    const long long lo_f = (long long)of * j0 - psm;
    long long klo = lo_f > 0 ? lo_f / of : 0;
    long long khi = ((long long)of * j1 + psm) / of;
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
      double sg = 0; long long off = 0; int tt = 0;
      if (g < gz) {
        const double sg0 = SGr[g];
        const int im = W.gimod[g], idv = W.gidiv[g];
        const bool below = sg0 < lim;                        // extinction.c:467
        if (A.part && idv >= j0 && idv <= j1) { if (below) nsk++; else nev++; }
        if (!below) {
          int id = idr[g];
          if (id == 0xFF) id = idst;
          const int K = s_K[wv][id];
          // q = of*j - iown + ps = of*(j + kb) + ph with 0 <= ph < of
          int ph = s_psm[wv][id] - im, kb = s_psd[wv][id] - idv;
          if (ph < 0) { ph += of; kb -= 1; }
          // valid k = j + kb: 0 <= k and of*k + ph <= 2*ps
          const int kv = (ph <= s_r2[wv][id]) ? K - 1 : K - 2;
          const long long kk0 = j0 + kb;                     // k of the tile's first bin
          const long long ta = kk0 < 0 ? -kk0 : 0;
          long long tb = (long long)kv - kk0;
          if (tb > jcount - 1) tb = jcount - 1;
          if (ta <= tb) {
            keep = true;
            sg = sg0 * dens;                                 // extinction.c:472-473
            off = s_po[wv][id] + (long long)ph * K + kk0;
            tt = (int)ta | ((int)tb << 16);
            nb += tb - ta + 1;
          }
        }
      }
      const unsigned long long km = __ballot(keep);
      const int n = __popcll(km);
      if (keep) {
        const int pos = __popcll(km & ((1ull << lane) - 1ull));
        s_sg[wv][pos] = sg; s_off[wv][pos] = off; s_tt[wv][pos] = tt;
      }
      __builtin_amdgcn_wave_barrier();
      // ---- phase B: lanes = bins, groups in line order, four table loads in flight
      const float *base = W.tabT + t0;
      int i = 0;
      for (; i + 4 <= n; i += 4) {
        const WideQuad q0 = wide_load(base + s_off[wv][i]);
        const WideQuad q1 = wide_load(base + s_off[wv][i + 1]);
        const WideQuad q2 = wide_load(base + s_off[wv][i + 2]);
        const WideQuad q3 = wide_load(base + s_off[wv][i + 3]);
        wide_apply(acc, s_sg[wv][i],     q0, s_tt[wv][i],     t0);
        wide_apply(acc, s_sg[wv][i + 1], q1, s_tt[wv][i + 1], t0);
        wide_apply(acc, s_sg[wv][i + 2], q2, s_tt[wv][i + 2], t0);
        wide_apply(acc, s_sg[wv][i + 3], q3, s_tt[wv][i + 3], t0);
      }
      for (; i < n; i++) {
        const WideQuad q0 = wide_load(base + s_off[wv][i]);
        wide_apply(acc, s_sg[wv][i], q0, s_tt[wv][i], t0);
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (live && cur_mx >= 0) flush(cur_mx);
  if (A.part) {
    nb = wave_sum_ll(nb); nev = wave_sum_ll(nev); nsk = wave_sum_ll(nsk);
    if (lane == 0) { s_nb[wv][0] = nb; s_nb[wv][1] = nev; s_nb[wv][2] = nsk; }
    __syncthreads();
    if (threadIdx.x < 3)
      A.part[((long long)c * A.part_stride + bx) * 3 + threadIdx.x] =
          (unsigned long long)(s_nb[0][threadIdx.x] + s_nb[1][threadIdx.x] + s_nb[2][threadIdx.x] + s_nb[3][threadIdx.x]);
  }
}

// The walk's rows (trx_walk.hip.h): every row is a whole number of 64-byte lines, its K entries
// behind `front` zeros and in front of at least as many -- the bins of a frame are consecutive
// entries of ONE row, inside ONE or two cache lines, and what a narrow profile does not reach is
// zero by position.  Profiles of up to 8 entries per row: 4 zeros, the entries, zeros to 16 floats
// (a frame of 8 bins = one aligned 64-byte line); up to 16 entries: 8 zeros, the entries, zeros to
// 32 floats; wider ones (no frame reads them) the same with whole lines.
__host__ __device__ inline void walk_row_layout(int K, int &front, int &stride)
{
  if (K <= 8) { front = 4; stride = 16; }
  else if (K <= 16) { front = 8; stride = 32; }
  else { front = 8; stride = (K + 16 + 15) & ~15; }
}

// phase-major copy of one profile: tabT[offT + ph*stride + front + k] = tab[off + osamp*k + ph]
// (walk = 0: the wide-profile kernel's copy, rows back to back; 1: the walk's, walk_row_layout)
__global__ __launch_bounds__(256)
void k_table_phase_major(const ProfileJob *jobs, const long long *joffT, const float *__restrict__ tab,
                         float *__restrict__ tabT, int of, int walk)
{
  const ProfileJob J = jobs[blockIdx.y];
  const long long offT = joffT[blockIdx.y];
  const int npt = J.nv;                                  // 2*ps + 1
  const int K = (npt - 1) / of + 1;
  int front = 0, stride = K;
  if (walk) walk_row_layout(K, front, stride);
  const long long total = (long long)of * K;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int ph = (int)(t / K), k = (int)(t - (long long)ph * K);
    const long long q = (long long)of * k + ph;
    tabT[offT + (long long)ph * stride + front + k] = (q < npt) ? tab[J.off + q] : 0.f;
  }
}

// The compact rows of the walk's copy for frames of 8 bins (k_line_walk_lanes): of a profile with at most
// 8 entries per row, a row keeps exactly the 8 floats a group of that row reads -- its window starts one
// float earlier when the row index borrowed a cell (r > ps % osamp) -- so a row is 32 bytes.  Laid out
// [phase][profile][8]: the row a group of phase ph reads (row (ps % osamp - ph) mod osamp of the 64-byte
// copy) of EVERY profile in one slab, profiles in table order (Lorentz index fastest).  The layers of a
// step differ by a step or none of the Lorentz index and share or nearly share the Doppler index: a group's
// rows of its 17 layers are then a few cache lines, not 17 (L1 -> L2 requests of the deep step 1.13e7 ->
// 5.9e6 per launch).
__global__ __launch_bounds__(256)
void k_table_rows32(const ProfileJob *jobs, const long long *joffW, const long long *joff32, const float *__restrict__ tabW,
                    float *__restrict__ tab32, int of, long long slab_floats)
{
  const ProfileJob J = jobs[blockIdx.y];
  const long long o32 = joff32[blockIdx.y];                // the profile's place inside a slab (floats)
  if (o32 < 0) return;                                   // (more than 8 entries per row: no frame of 8 bins reads it)
  const long long offW = joffW[blockIdx.y];
  const int ps = (J.nv - 1) / 2, psq = ps / of, psr = ps - psq * of;
  const long long total = (long long)of * 8;
  for (long long t = (long long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long long)gridDim.x * 256) {
    const int r = (int)(t >> 3), k = (int)(t & 7);
    const int start = 1 + psq - (r > psr ? 1 : 0);
    int ph = psr - r;                                      // the phase whose groups read row r
    if (ph < 0) ph += of;
    tab32[(long long)ph * slab_floats + o32 + k] = tabW[offW + (long long)r * 16 + start + k];
  }
}

// ---------------------------------------------------------------------------
// opacity-grid mode: extinction by temperature interpolation
// ---------------------------------------------------------------------------
// extinction.c:535-581 (interpolmolext): e[r][w] = sum_m rho_m(r) * linear-in-T
// interpolation of the grid o[layer][temp][mol][wn].  One lane per (wn, layer).
struct GridArgs {
  const double *o; int nt, nm, nr; long long nwave, lo, nsh;
  int r_top, nc;
  const int *itemp;            // [nr] lower temperature node
  const double *w_lo, *w_hi, *dg;   // [nr] (g1 - T), (T - g0), (g1 - g0)
  const double *dens;          // [nm][nr] density of each grid molecule
  double *e;                   // [nr][nsh]
  const int *flags; int eager;
};

__global__ __launch_bounds__(256)
void k_grid_extinction(GridArgs G)
{
  if (!G.eager && G.flags[0] == 0) return;
  const long long w = (long long)blockIdx.x * 256 + threadIdx.x;
  if (w >= G.nsh) return;
  const int r = G.r_top - (int)blockIdx.y;
  const int it = G.itemp[r];
  double acc = 0.0;
  for (int m = 0; m < G.nm; m++) {
    const double olo = G.o[(((long long)r * G.nt + it    ) * G.nm + m) * G.nwave + G.lo + w];
    const double ohi = G.o[(((long long)r * G.nt + it + 1) * G.nm + m) * G.nwave + G.lo + w];
    const double ext = (olo * G.w_lo[r] + ohi * G.w_hi[r]) / G.dg[r];
    acc += G.dens[(long long)m * G.nr + r] * ext;
  }
  G.e[(long long)r * G.nsh + w] = acc;
}

// ---------------------------------------------------------------------------
// CIA / cross-section extinction
// ---------------------------------------------------------------------------
// crosssec.c:272-344 (interpcs) + :354-428 (bicubicinterpolate): natural cubic
// splines (pu/src/spline.c), first along temperature for every table row, then
// along wavenumber for every layer; no extrapolation; negative values dropped.
// The table-only halves of both splines are computed once at trx_create
// (zt = second derivatives along T of every table row, uw = the tridiagonal
// pivots u[] of the wavenumber spline, which depend on the wn grid alone).
struct CiaDev { int nwave, ntemp; const double *wn, *temp, *cs, *zt, *uw, *ruw, *rh; const double *wf, *wb; };      // wf, wb: [nwave][kCiaTerms] weights of k_cia_v / k_cia_z, or null

// one lane per (table row, layer): the T spline evaluated at the layer temperature.
// mid is [nwave][nr] so that the next kernel walks it with unit stride per lane.
// The tables of a run are independent until their values are added: up to kCiaBatch of them go
// through each kernel side by side (blockIdx.y = table), and k_cia_eval adds them in table order.
constexpr int kCiaBatch = 4;
struct CiaJob {
  CiaDev C;
  int fj, lj; long long fi, li;      // layers / wavenumbers inside the table (crosssec.c:376-393)
  int ia, iz;                        // table rows the wavenumber spline is solved for (0, nwave-1: all of them; see k_cia_layers)
  double *mid, *z2, *v;              // [nwave][nr] each
  const double *dens;                // [nr] density product
};
struct CiaBatch { int n; CiaJob J[kCiaBatch]; };

__global__ __launch_bounds__(256)
void k_cia_rows(CiaBatch B, int nr, const double *__restrict__ tlay)
{
  const CiaDev &C = B.J[blockIdx.y].C;
  const int fj = B.J[blockIdx.y].fj, lj = B.J[blockIdx.y].lj;
  double *__restrict__ mid = B.J[blockIdx.y].mid;
  // (the launch covers the rows ia..iz of the widest window of the batch: k_cia_layers)
  const int ia = B.J[blockIdx.y].ia, iz = B.J[blockIdx.y].iz;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int i = ia + (int)(t / nr), j = (int)(t % nr);
  if (i > iz || j < fj || j >= lj) return;
  latency_critical();
  const long long idx = (long long)i * nr + j;
  mid[idx] = spline_eval_pt(C.zt + (long long)i * C.ntemp, C.ntemp, C.temp, C.cs + (long long)i * C.ntemp, tlay[j]);
}

// one lane per layer: second derivatives of the wavenumber spline -- the sequential
// tridiagonal sweep of spline_second_derivs (trx_numerics.h), pivots from the table.
// With one wave per 64 layers nothing hides a memory round trip, and a load one step
// ahead of its use still waited ~400 cycles per step (and for the step's store: loads and
// stores return in order on this part).  So the sweeps run in chunks of kCiaChunk steps:
// all of a chunk's operands are requested together, the chain then runs out of registers,
// and the chunk's results are stored together.  The divisions by table constants (pivots,
// spacings) are multiplications by reciprocals made at create: the chain is mul-fma
// (e_cs agrees with the oracle to 1e-13).  z2/v are [nwave][nr].
constexpr int kCiaChunk = 16;
constexpr int kCiaMargin = 128;          // rows solved beyond the ones a run's wavenumbers bracket (k_cia_layers)

// Segments (blockIdx.z): the rows a run needs, cut into pieces of seg_rows, each piece a window of its own -- the
// same margins, the same argument, the same doubles -- with its own scratch for v (the pieces' margins overlap) and
// storing the second derivatives of ITS rows only.  Two sweeps of ~1700 dependent steps on two waves were 50 us of
// latency next to one walk and 80 next to two, the longest thing on the CIA queue; pieces of 128 rows are sweeps of
// 384 + 256 steps (tests/test_gpu_cia_window.py: bit for bit against one segment).
__global__ __launch_bounds__(64)
void k_cia_layers(CiaBatch B, int nr, int seg_rows, long long seg_vstride)
{
  const CiaDev &C = B.J[blockIdx.y].C;
  const int fj = B.J[blockIdx.y].fj, lj = B.J[blockIdx.y].lj;
  const double *__restrict__ mid = B.J[blockIdx.y].mid;
  double *__restrict__ z2 = B.J[blockIdx.y].z2, *__restrict__ v = B.J[blockIdx.y].v + (long long)blockIdx.z * seg_vstride;
  const int j = fj + blockIdx.x * 64 + threadIdx.x;
  if (j >= lj) return;
  const long n = C.nwave;
  // The rows ia..iz of the table: all of them, or -- where the run's wavenumbers lie inside a small
  // part of a long table (a shard of a wide band; a band inside a 20-10000 cm-1 table) -- the rows
  // around them with kCiaMargin to spare on either side, the sweeps started from zero there.  What
  // a sweep carries from row to row shrinks by the pivots' ratio (< 0.3) every row: after the margin
  // a different start is 10^-60 of the value, the doubles inside are those of the whole table's
  // sweeps (tests/test_gpu_cia_window.py: bit for bit against them).
  const long ja = B.J[blockIdx.y].ia, jz = B.J[blockIdx.y].iz;        // the job's window (trx_api.hip)
  const long need_a = ja == 0 ? 0 : ja + kCiaMargin, need_b = jz == n - 1 ? n - 1 : jz - kCiaMargin;      // the rows it is solved for
  const long sa = need_a + (long)blockIdx.z * seg_rows;              // this segment's rows sa..sb, its window ia..iz
  if (sa > need_b) return;
  const long sb = min(need_b, sa + (long)seg_rows - 1);
  long ia = sa - kCiaMargin, iz = sb + kCiaMargin;
  if (ia < 3) ia = 0;
  if (iz > n - 4) iz = n - 1;
  latency_critical();
  const double *__restrict__ x = C.wn, *__restrict__ ru = C.ruw, *__restrict__ rh = C.rh;
  const double *y = mid + j; double *z = z2 + j, *vs = v + j;
#define TRX_CIA_V(i) vs[((i) - ia) * nr]
  auto own = [&](long i) { return i >= sa && i <= sb; };
  double vp = 0;
  long i0 = 2;                                             // first row of the forward recurrence
  double yi, bim;
  if (ia == 0) {
    if (n > 2) {
      const double h0 = x[1] - x[0], h1 = x[2] - x[1];
      const double b0 = (y[1L*nr] - y[0]) / h0, b1 = (y[2L*nr] - y[1L*nr]) / h1;
      vp = 6 * (b1 - b0);
      TRX_CIA_V(1L) = vp;
    }
    yi = n > 2 ? y[2L*nr] : 0.0; bim = n > 2 ? (yi - y[1L*nr]) / (x[2] - x[1]) : 0.0;
  } else {
    i0 = ia + 1;                                           // (ia >= 3)
    yi = y[i0 * nr]; bim = (yi - y[(i0 - 1) * nr]) * rh[i0 - 1];
    TRX_CIA_V(ia) = 0.0;
  }
  const long iend = iz == n - 1 ? n - 1 : iz;              // the recurrence runs for rows i0 .. iend-1
  if (n > 3) {
    // forward: v[i] = 6 (b[i] - b[i-1]) - v[i-1] h[i-1] / u[i-1],  b[i] = (y[i+1] - y[i]) / h[i]
    for (long base = i0; base < iend; base += kCiaChunk) {
      double yb[kCiaChunk], vo[kCiaChunk];
#pragma unroll
      for (int k = 0; k < kCiaChunk; k++) { const long i = base + k + 1; yb[k] = y[(i < n ? i : n - 1) * nr]; }     // y[i+1] of step i
#pragma unroll
      for (int k = 0; k < kCiaChunk; k++) {
        const long i = base + k;
        if (i < iend) {
          const double bi = (yb[k] - yi) * rh[i];
          const double vn = 6*(bi - bim) - vp * (x[i] - x[i-1]) * ru[i-1];
          vo[k] = vn; vp = vn; bim = bi; yi = yb[k];
        }
      }
#pragma unroll
      for (int k = 0; k < kCiaChunk; k++) if (base + k < iend) TRX_CIA_V(base + k) = vo[k];
    }
  }
  if (ia == 0 && own(0)) z[0] = 0;
  if (own(iz)) z[iz * nr] = 0;                             // (row n-1 of the table; a window's pretended end is nobody's row)
  if (n > 2) {
    // backward: z[i] = (v[i] - h[i] z[i+1]) / u[i]
    double zn = 0;
    const long ibot = ia == 0 ? 1 : ia + 1;               // lowest row the recurrence gives
    for (long top = iz - 1; top >= ibot; top -= kCiaChunk) {
      double vb[kCiaChunk], zo[kCiaChunk];
#pragma unroll
      for (int k = 0; k < kCiaChunk; k++) { const long i = top - k; vb[k] = TRX_CIA_V(i >= ibot ? i : ibot); }
#pragma unroll
      for (int k = 0; k < kCiaChunk; k++) {
        const long i = top - k;
        if (i >= ibot) {
          const double zi = (vb[k] - (x[i+1] - x[i]) * zn) * ru[i];
          zo[k] = zi; zn = zi;
        }
      }
#pragma unroll
      for (int k = 0; k < kCiaChunk; k++) if (top - k >= ibot && own(top - k)) z[(top - k) * nr] = zo[k];
    }
  }
#undef TRX_CIA_V
}

// The same second derivatives without a sweep.  The forward recurrence v[i] = g[i] - c[i] v[i-1] (g[i] = 6 (b[i] - b[i-1]),
// c[i] = h[i-1] / u[i-1]) is v[i] = g[i] - c[i] g[i-1] + c[i] c[i-1] g[i-2] - ..., and c < 0.3 on any sensible grid: after
// kCiaTerms terms the products are below 2^-60 (trx_create checks that on the table's own pivots and leaves the weights
// out otherwise: the sweeps then run).  Likewise backwards, z[i] = v[i]/u[i] - d[i] v[i+1]/u[i+1] + ..., d[i] = h[i] / u[i].
// The products are table constants (weights wf, wb, made at create); every (row, layer) pair is then a sum of
// kCiaTerms terms on its own, smallest first: two launches of a few microseconds instead of two chains of a thousand
// dependent steps on two waves (50 us next to one walk, 80 next to two, and proportional to the table's rows).  A row
// depends on its 2 x 48 neighbours only: a shard's window, a segment, the whole table give the same doubles by
// construction.  Rounding differs from the sweeps' (and the oracle's) in the last place; e_cs agrees with the oracle
// to 1e-13 as before (tests/test_gpu_cia_window.py compares the two forms).
constexpr int kCiaTerms = 48;

__device__ __forceinline__ void cia_need(const CiaJob &J, long n, long &need_a, long &need_b)
{
  need_a = J.ia == 0 ? 0 : J.ia + kCiaMargin; need_b = J.iz == n - 1 ? n - 1 : J.iz - kCiaMargin;
}

__global__ __launch_bounds__(256)
void k_cia_v(CiaBatch B, int nr)
{
  const CiaJob &J = B.J[blockIdx.y];
  const CiaDev &C = J.C;
  const long n = C.nwave;
  long need_a, need_b;
  cia_need(J, n, need_a, need_b);
  const long va = max(1L, need_a), vb = min(n - 2, need_b + kCiaTerms - 1);          // rows of v that k_cia_z reads
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long i = va + (long)(t / nr); const int j = (int)(t % nr);
  if (i > vb || j < J.fj || j >= J.lj) return;
  latency_critical();
  const double *__restrict__ y = J.mid + j, *__restrict__ rh = C.rh, *__restrict__ w = C.wf + i * kCiaTerms;
  // rows i - kCiaTerms .. i + 1 go by once, eight at a time (the kernel runs next to the walks: few registers, or its
  // waves wait for a SIMD with room); a row outside the window or the table: the nearest one inside -- its weight is zero
  auto yrow = [&](long r) { return y[min(max(r, (long)J.ia), (long)J.iz) * nr]; };
  double acc = 0.0;
  double y0 = yrow(i - kCiaTerms), y1 = yrow(i - kCiaTerms + 1);
  double bm = (y1 - y0) * rh[min(max(i - kCiaTerms, 0L), n - 2)];      // b of the row before the first term's
#pragma unroll 8
  for (int q = 1; q <= kCiaTerms; q++) {                     // row m = i - kCiaTerms + q: the term k = kCiaTerms - q, smallest first
    const long m = i - kCiaTerms + q;
    const double y2 = yrow(m + 1);
    const double b = (y2 - y1) * rh[min(max(m, 0L), n - 2)];
    acc = __builtin_fma(w[kCiaTerms - q], 6 * (b - bm), acc);
    bm = b; y1 = y2;
  }
  J.v[i * nr + j] = acc;
}

__global__ __launch_bounds__(256)
void k_cia_z(CiaBatch B, int nr)
{
  const CiaJob &J = B.J[blockIdx.y];
  const CiaDev &C = J.C;
  const long n = C.nwave;
  long need_a, need_b;
  cia_need(J, n, need_a, need_b);
  const long va = max(1L, need_a), vb = min(n - 2, need_b + kCiaTerms - 1);
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const long i = need_a + (long)(t / nr); const int j = (int)(t % nr);
  if (i > need_b || j < J.fj || j >= J.lj) return;
  latency_critical();
  double acc = 0.0;
  if (i >= 1 && i <= n - 2) {
    const double *__restrict__ v = J.v + j, *__restrict__ w = C.wb + i * kCiaTerms;
#pragma unroll 8
    for (int k = kCiaTerms - 1; k >= 0; k--) acc = __builtin_fma(w[k], v[min(max(i + k, va), vb) * nr], acc);
  }
  J.z2[i * nr + j] = acc;                                   // (rows 0 and n - 1: the natural spline's zeros)
}

// one lane per (wavenumber, layer): evaluate every table of the batch, scale by its density
// product, add in table order.  The FIRST batch of a run covers the whole [layer][wavenumber]
// array and starts every sum from zero (a memset of the array ahead of the kernels was a dispatch
// of its own on this queue, 12-35 us next to a walk); a later batch covers the union of its
// tables' ranges and adds to what is there.
// G layers per lane: a table interval is located once for them.  16 on large grids (10^7 rays:
// 17 -> 10 ms at configs[4]); on small ones the searches are nothing and the lanes are few: 1.
template <int G>
__global__ __launch_bounds__(256)
void k_cia_eval(CiaBatch B, int nr, long long nsh, long long lo, double wn_i, double wn_d, double wn_fct,
                long long fi0, long long li1, int fj0, int lj1, int first, double *__restrict__ ecs)
{
  const long long w = fi0 + (long long)blockIdx.x * 256 + threadIdx.x;
  const int j0 = fj0 + blockIdx.y * G, nj = min(G, lj1 - j0);
  if (w >= li1) return;
  latency_critical();
  const double xo = wn_fct * (wn_i + (double)(lo + w) * wn_d);
  double sum[G];
  unsigned any = first ? 0xffffu : 0u;
#pragma unroll
  for (int q = 0; q < G; q++) sum[q] = (first || q >= nj) ? 0.0 : ecs[(long long)(j0 + q) * nsh + w];
  for (int t = 0; t < B.n; t++) {
    const CiaJob &J = B.J[t];
    if (w < J.fi || w >= J.li) continue;
    const int k = spline_interval(J.C.wn, J.C.nwave, xo);          // the same for every layer
#pragma unroll
    for (int q = 0; q < G; q++) {
      const int j = j0 + q;
      if (q >= nj || j < J.fj || j >= J.lj) continue;
      const double val = spline_eval_at(k, J.z2 + j, J.C.wn, J.mid + j, xo, nr, 1, nr);
      if (val > 0) { sum[q] += val * J.dens[j]; any |= 1u << q; }
    }
  }
#pragma unroll
  for (int q = 0; q < G; q++)
    if (q < nj && ((any >> q) & 1u)) ecs[(long long)(j0 + q) * nsh + w] = sum[q];
}

// ---------------------------------------------------------------------------
// optical depth
// ---------------------------------------------------------------------------
struct TauArgs {
  int nr, solution;                  // layers; TRX_SOL_*
  long long nsh, lo;
  double wn_i, wn_d, wn_fct, rad_fct, toomuch;
  int r_top, nc;                     // layers of this chunk: r_top, r_top-1, ...
  const double *rad;                 // [nr]
  const double *e, *ecs;             // [nr][nsh]
  double *er;                        // [nr][nsh] total extinction, edited in place (eclipse.c:65-66)
  int er_all;                        // vertical rays: 1 = every layer's bottom-point value goes to er (dumps); 0 = only a launch's last two layers, which
                                     // the next launch's parabolas start from -- nothing else reads them, and they were a fifth of the kernel's bytes
  double *tau;                       // [nr(height)][nsh]
  int *last;                         // [nsh], -1 while the ray is still descending
  // Simpson weights per start layer rs: row rs holds, per interval pair,
  // {2-hratio, hfactor, 2-1/hratio, hsum} (numerical.c:390-425, 500-525)
  const double *gw; int gstride;     // gw[rs*gstride + 4*i + k]
  const double *gh0;                 // [nr] first interval (trapezoid when the count is even)
  // scattering / clouds (extinction.c:587-693)
  int scat_flag, cloud_flag, nmol;
  double scat_pref;                  // 10^logext * E0H2
  const double *press, *temp;        // [nr]
  const double *scat_pol;            // [nr] sum_j pi*8e-32/3*pol_j^2*rho_j/m_j*N_A  (flag 2)
  double cloud_top, cloud_bot, cloud_ext, cloud_gamma, cloud_Q, cloud_r, cloud_sig, cloud_refwn;
  const double *mdens, *nH;          // [nr]
  // the wavenumber-only factors of both models, one pair per ray (k_extras_factors; null: models off),
  // and their three run constants {10^cloud_top, 10^cloud_bot, cloud_refwn^cloud_gamma}
  const double *xf_scat, *xf_cloud, *xf_const;
  int *flags;                        // [0] active rays (gate), [1] rays still active after this chunk
  int eager;
  // vertical (eclipse) rays only: weights of the interval pair that STARTS at layer k,
  // {2-hratio, hfactor, 2-1/hratio, hsum} from h_k = rad[k+1]-rad[k], and the running
  // Simpson sums of both parities carried from chunk to chunk
  const double *pw;                  // [nr][4]
  const double *lay;                 // [nr][kVertLay] per-layer constants of the chain (vertical_stage_layers)
  double *acc;                       // [2][nsh]
  // slant rays only.  The reference hands its ray solution b = h*hfct/rfct (tau.c:274), which its
  // object code evaluates as (h*hfct)*(1/rfct): b comes out on the layer radius it was made from
  // or an ulp beside it, and the bracket search (slantpath.c:36) then integrates from b instead of
  // the radius (b high) or starts at the layer below (b low; not seen with fct = 1e5, handled).
  // Per height k (layer index of the height): hrs = lowest layer of the bracket (-1: optical depth
  // 0, -3: closest approach below the bottom layer), hr0 = b; the Simpson weights gw/gh0 are then
  // those of the point set {b, rad[hrs+1], ...} and are indexed by k.
  const double *hrs, *hr0;
  int *status;
};

// (out of line: a dozen inlined pow() bodies per call site would push the optical-depth
// loops past the instruction cache; most runs have both models switched off)
__device__ __noinline__ double scat_term(const TauArgs &T, int r, double wn)
{
  if (T.scat_flag == 1) return T.scat_pref * T.press[r] / T.temp[r] * pow(wn, 4);
  if (T.scat_flag == 2) return T.scat_pol[r] * pow(2. * kPi * wn * kMicron, 4);
  return 0.0;
}

// extinction.c:630-693: zero above the top and below the bottom of the deck
__device__ __noinline__ double cloud_term(const TauArgs &T, int r, double wn)
{
  if (T.cloud_flag == 0 || T.cloud_ext == 0.0) return 0.0;
  const double ctop = pow(10, T.cloud_top), cbot = pow(10, T.cloud_bot);
  const double p = T.press[r];
  // the reference walks down from the top: layers with p < ctop are clear, then
  // cloudy until the first layer with p >= cbot, clear below (pressure is monotonic)
  if (p < ctop || p >= cbot) return 0.0;
  switch (T.cloud_flag) {
    case 1: return T.cloud_ext;
    case 2: return T.cloud_ext * T.mdens[r];
    case 3: return T.cloud_ext * pow(wn, T.cloud_gamma) * T.mdens[r];
    case 4: {
      const double x = 2 * kPi * T.cloud_r * wn;
      return T.cloud_ext / (T.cloud_Q * pow(x, -1 * T.cloud_gamma) + pow(x, 0.2)) * T.mdens[r];
    }
    case 5: return T.nH[r] * (T.cloud_ext * pow(wn, T.cloud_gamma)) * T.cloud_sig /
                   pow(T.cloud_refwn, T.cloud_gamma) * T.mdens[r];
  }
  return 0.0;
}

// The same two terms split into a wavenumber-only factor (the pow() calls: once per ray, by
// k_extras_factors) and what is left per layer -- the products in the order of scat_term / cloud_term,
// so the same bits.  The optical-depth kernels evaluate the terms per (ray, layer): with the
// out-of-line bodies above that was two calls, several pow() and a 344-byte scratch copy of the
// argument block per lane (eclipse with both models on: 0.60 ms per spectrum against 0.41 without).
__device__ __forceinline__ double scat_factor(const TauArgs &T, double wn)
{
  if (T.scat_flag == 1) return pow(wn, 4);
  if (T.scat_flag == 2) return pow(2. * kPi * wn * kMicron, 4);
  return 0.0;
}
__device__ __forceinline__ double cloud_factor(const TauArgs &T, double wn)
{
  switch (T.cloud_flag) {
    case 3: case 5: return T.cloud_ext * pow(wn, T.cloud_gamma);
    case 4: { const double x = 2 * kPi * T.cloud_r * wn; return T.cloud_ext / (T.cloud_Q * pow(x, -1 * T.cloud_gamma) + pow(x, 0.2)); }
  }
  return T.cloud_ext;
}
__device__ __forceinline__ double scat_layer(const TauArgs &T, int r, double fs)
{
  if (T.scat_flag == 1) return T.scat_pref * T.press[r] / T.temp[r] * fs;
  if (T.scat_flag == 2) return T.scat_pol[r] * fs;
  return 0.0;
}
__device__ __forceinline__ double cloud_layer(const TauArgs &T, int r, double fc)
{
  if (T.cloud_flag == 0 || T.cloud_ext == 0.0) return 0.0;
  const double p = T.press[r];
  if (p < T.xf_const[0] || p >= T.xf_const[1]) return 0.0;
  switch (T.cloud_flag) {
    case 1: return T.cloud_ext;
    case 2: return T.cloud_ext * T.mdens[r];
    case 3: case 4: return fc * T.mdens[r];
    case 5: return T.nH[r] * fc * T.cloud_sig / T.xf_const[2] * T.mdens[r];
  }
  return 0.0;
}
__global__ __launch_bounds__(256)
void k_extras_factors(TauArgs T, double *__restrict__ fs, double *__restrict__ fc, double *__restrict__ consts)
{
  const long long w = (long long)blockIdx.x * 256 + threadIdx.x;
  if (w == 0) { consts[0] = pow(10, T.cloud_top); consts[1] = pow(10, T.cloud_bot); consts[2] = pow(T.cloud_refwn, T.cloud_gamma); }
  if (w >= T.nsh) return;
  const double wcgs = (T.wn_i + (double)(T.lo + w) * T.wn_d) * T.wn_fct;
  fs[w] = scat_factor(T, wcgs);
  fc[w] = cloud_factor(T, wcgs);
}

// the two model terms on their own, for the reference's scatt_extion.dat / cloud_extion.dat dumps
__global__ __launch_bounds__(256)
void k_extras_dump(TauArgs T, double *__restrict__ e_scat, double *__restrict__ e_cloud)
{
  const long long w = (long long)blockIdx.x * 256 + threadIdx.x;
  const int r = blockIdx.y;
  if (w >= T.nsh) return;
  const double wcgs = (T.wn_i + (double)(T.lo + w) * T.wn_d) * T.wn_fct;
  if (e_scat)  e_scat [(long long)r * T.nsh + w] = scat_term(T, r, wcgs);
  if (e_cloud) e_cloud[(long long)r * T.nsh + w] = cloud_term(T, r, wcgs);
}

// End of an optical-depth launch: every block adds its number of rays that are still
// descending and its deepest stopping height ONCE (blocks loop over tiles, so the
// same-address atomics stay in the low thousands at any grid size); the last block to
// arrive publishes the total for the gating of the next step's kernels.
constexpr int kTauMaxBlocks = 2048;

__device__ __forceinline__ void tau_publish(const TauArgs &T, int nstill, int deep)
{
  __shared__ int s_still[4], s_deep[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = (blockDim.x + 63) >> 6;
  nstill = (int)wave_sum_ll(nstill);
  deep = wave_max_i(deep);
  if (lane == 0) { s_still[wv] = nstill; s_deep[wv] = deep; }
  __syncthreads();
  if (threadIdx.x == 0) {
    int st = 0, dp = 0;
    for (int k = 0; k < nwv; k++) { st += s_still[k]; dp = max(dp, s_deep[k]); }
    if (st) atomicAdd(&T.flags[1], st);
    if (dp > T.flags[4]) atomicMax(&T.flags[4], dp);
    __threadfence();
    const int ticket = atomicAdd(&T.flags[3], 1);
    if (ticket == (int)gridDim.x - 1) {
      __threadfence();
      const int act = atomicAdd(&T.flags[1], 0);
      T.flags[2] += T.nc;                                // layers swept so far
      T.flags[1] = 0; T.flags[3] = 0;
      __threadfence();
      atomicExch(&T.flags[0], act);
    }
  }
}

// Ray geometry of the transit solution on the device: per height the impact parameter as the
// reference's object code forms it (tau.c:274), its bracket layer (slantpath.c:36-44), the
// Simpson weights of its point set (slantpath.c:62-95; numerical.c:390-425), and per point count
// the weights of the modulation integral (slantpath.c:399-408).  O(nlayer^2) numbers that depend
// on the radii alone: the host used to build them (~25 us) and ship them (330 KB) every run.
// Same operations in the same order as that host code (IEEE sqrt and division, no contraction):
// the same bits.  One block per height / point count, one lane per Simpson pair.
struct SlantGeomArgs {
  const double *rad; int nr; double fct;
  double *gw, *gh0, *hrs, *hr0;       // per height k: weights [k][gstride], first interval, bracket layer, closest approach
  double *mw, *mh0;                   // per point count cnt: weights [cnt][gstride], first interval
  int gstride;
};

__device__ __forceinline__ void simpson_pair(double xa, double xb, double xc, double *row)
{
  const double ha = xb - xa, hb = xc - xb;
  const double hsum = ha + hb, hratio = hb / ha, hfactor = hsum * hsum / (ha * hb);
  row[0] = 2.0 - hratio; row[1] = hfactor; row[2] = 2.0 - 1.0 / hratio; row[3] = hsum;
}

__global__ __launch_bounds__(64)
void k_slant_geometry(SlantGeomArgs G)
{
  const int nr = G.nr, lane = threadIdx.x;
  latency_critical();
  if ((int)blockIdx.x < nr) {                              // ---- height k
    const int k = blockIdx.x;
    const double recip = 1.0 / G.fct;
    const double b = (G.rad[k] * G.fct) * recip;
    const int rs = bracket_ie(G.rad, 0, nr - 1, b);        // slantpath.c:36
    if (lane == 0) { G.hr0[k] = b; G.hrs[k] = (rs == -5 || rs == -2) ? -1.0 : (rs < 0 ? -3.0 : (double)rs); }
    if (rs < 0) return;
    int n = nr - rs;
    // point i of the set: rr[0] = b, rr[i] = rad[rs + i]; two points become three (slantpath.c:62-74)
    auto rr = [&](int i) -> double {
      if (n == 2) return i == 0 ? b : (i == 2 ? G.rad[rs + 1] : (b + G.rad[rs + 1]) / 2.0);
      return i == 0 ? b : G.rad[rs + i];
    };
    const int np = n == 2 ? 3 : n;
    auto sx = [&](int i) -> double { if (i == 0) return 0.0; const double r = rr(i); return sqrt(r * r - b * b); };   // :82-84
    if (lane == 0) G.gh0[k] = sx(1) - sx(0);
    const int even = (np % 2 == 0);
    for (int i = lane; i < (np - 1) / 2; i += 64) {
      const int j = 2 * i + even;
      simpson_pair(sx(j), sx(j + 1), sx(j + 2), G.gw + (long long)k * G.gstride + 4 * i);
    }
  } else {                                                 // ---- point count cnt of the modulation integral
    const int cnt = (int)blockIdx.x - nr + 3;
    if (cnt > nr) return;
    auto x = [&](int q) -> double { return G.rad[nr - 1 - (cnt - 1 - q)] * G.fct; };
    if (lane == 0) G.mh0[cnt] = x(1) - x(0);
    const int even = (cnt % 2 == 0);
    for (int i = lane; i < (cnt - 1) / 2; i += 64) {
      const int j = 2 * i + even;
      simpson_pair(x(j), x(j + 1), x(j + 2), G.mw + (long long)cnt * G.gstride + 4 * i);
    }
  }
}

// The two per-height pieces of a slant ray's optical depth (slantpath.c:55-107), with the total
// extinction of layer L handed in as er(L) -- global memory, a block's LDS tile, or k_ray_tail's.
// slantpath.c:55-58: extinction at the closest-approach radius r0, a parabola through the bracket's
// lowest layer rs (value ylow: what that layer holds at this point of the sweep) and the two above it
template <class ErOf>
__device__ __forceinline__ double slant_bottom(const TauArgs &T, int rs, double r0, double ylow, ErOf er)
{
  if (T.nr - rs == 2) return parab3(T.rad[rs-1], T.rad[rs], er(rs - 1), ylow, er(rs + 1), r0);
  return parab3(T.rad[rs], T.rad[rs+1], ylow, er(rs + 1), er(rs + 2), r0);
}
// slantpath.c:62-107: Simpson over the point set {r0, rad[rs+1], ...} with the weights of height k
// (TauArgs.gw/gh0); returns the integral over half the chord (the caller doubles it)
template <class ErOf>
__device__ __forceinline__ double slant_integral(const TauArgs &T, int k, int rs, double y0, ErOf er)
{
  const int n = T.nr - rs;
  const double *g = T.gw + (long long)k * T.gstride;            // weights of THIS height's point set
  double res;
  if (n == 2) {                                       // slantpath.c:62-74
    const double y2 = er(rs + 1);
    const double y1 = (y2 + y0) / 2.0;
    res = ((y0 * g[0] + y1 * g[1] + y2 * g[2]) * g[3]) / 6.0;
  } else {
    const int even = (n % 2 == 0);
    double acc = 0.0;
    const int npair = (n - 1) / 2;
    // four interval pairs per trip: their nine points and sixteen weights are requested together
    // (pair by pair, every trip waited for its own loads -- ~30 round trips for a deep ray), the
    // sum keeps the pairs' order
    double gn[16];                                     // the NEXT trip's weights, requested a trip ahead
#pragma unroll
    for (int q = 0; q < 16; q++) gn[q] = g[4 * min(q / 4, npair - 1) + (q & 3)];
    for (int i0 = 0; i0 < npair; i0 += 4) {
      const int jb = 2 * i0 + even;
      double yv[9], gg[16];
#pragma unroll
      for (int q = 0; q < 16; q++) gg[q] = gn[q];
      if (i0 + 4 < npair) {
#pragma unroll
        for (int q = 0; q < 16; q++) gn[q] = g[4 * min(i0 + 4 + q / 4, npair - 1) + (q & 3)];
      }
#pragma unroll
      for (int q = 0; q < 9; q++) yv[q] = er(rs + min(jb + q, n - 1));
      if (jb == 0) yv[0] = y0;
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (i0 + u < npair) acc += (yv[2*u] * gg[4*u] + yv[2*u+1] * gg[4*u+1] + yv[2*u+2] * gg[4*u+2]) * gg[4*u+3];
    }
    res = acc / 6.0;
    if (even) res += T.gh0[k] * (y0 + er(rs + 1)) / 2;
  }
  return res;
}

// Heights are visited top-down, a chunk of layers per launch (tau.c:235-290).
// A 256-thread block owns 256/kTauH wavenumbers x kTauH heights:
//   phase 1 (one lane per wavenumber): total extinction of the chunk's layers
//           (tau.c:231-232) and the bottom-point parabola of every height, which
//           the eclipse geometry leaves in er (eclipse.c:65-66) -- a short
//           sequential chain down the chunk;
//   phase 2 (one lane per wavenumber x height): the Simpson sums, independent
//           once the chain is known; weights are wave-uniform loads, er reads are
//           coalesced along the wavenumber axis;
//   phase 3 (one lane per wavenumber): toomuch cut in height order (tau.c:277-287).
constexpr int kTauH = 32;                 // heights per block (>= layers per chunk)
constexpr int kTauW = 256 / kTauH;        // wavenumbers per block
constexpr int kTauStageRows = 256;        // rows of a tile kept in LDS (deeper atmospheres read global memory)

template <bool EXTRAS>                    // EXTRAS: a scattering or cloud model is switched on (their out-of-line
__global__ __launch_bounds__(256)         // bodies take TauArgs by reference: a copy in scratch memory per lane)
void k_optical_depth(TauArgs T)
{
  if (!T.eager && T.flags[0] == 0) return;
  latency_critical();
  __shared__ double s_y0[kTauH][kTauW];
  __shared__ double s_tv[kTauH][kTauW];
  __shared__ int s_alive[kTauW];
  // the tile's total extinction from the step's lowest layer up, fetched once per block: every
  // height of the tile integrates over (nearly) the same rows
  __shared__ double s_y[kTauStageRows][kTauW];
  const bool staged = T.nr - (T.r_top - T.nc + 1) <= kTauStageRows;
  const int wi = threadIdx.x % kTauW, hc = threadIdx.x / kTauW;
  const int nr = T.nr;
  const int r_low = T.r_top - T.nc + 1;                  // lowest layer whose extinction exists so far
  const long long ntiles = (T.nsh + kTauW - 1) / kTauW;
  int nstill = 0, deep = 0;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  const long long w = tile * kTauW + wi;
  const bool ok = w < T.nsh;

  // phase 1a: every (wavenumber, height) lane adds up the total extinction of ITS layer of the
  // chunk (tau.c:231-232) -- it used to be one lane per wavenumber going down the chunk
  const bool alive = ok && T.last[w] < 0;
  if (hc == 0) s_alive[wi] = alive;
  if (alive && hc < T.nc) {
    const int r = T.r_top - hc;
    const long long k = (long long)r * T.nsh + w;
    if (EXTRAS) T.er[k] = T.e[k] + scat_layer(T, r, T.xf_scat[w]) + cloud_layer(T, r, T.xf_cloud[w]) + T.ecs[k];
    else        T.er[k] = T.e[k] + T.ecs[k];
  }
  __syncthreads();                                       // (the block's own global writes: visible to it after the barrier)
  if (staged) {
    const int rows = nr - r_low;
    for (int t = threadIdx.x; t < rows * kTauW; t += 256) {
      const int row = t / kTauW, col = t % kTauW;
      const long long ww = tile * kTauW + col;
      s_y[row][col] = ww < T.nsh ? T.er[(long long)(r_low + row) * T.nsh + ww] : 0.0;
    }
    __syncthreads();
  }
  // phase 1b: extinction at the closest-approach radius of this lane's height (slantpath.c:55-58).
  // The bracket's lowest layer rs is the height's own layer, or -- when the impact parameter came
  // out an ulp below the layer radius (TauArgs.hrs) -- the layer under it.  That one may lie
  // below this step: the reference, sweeping lazily, has no line extinction there yet either
  // (tau.c:231-232 with an unswept row), and its weight in the parabola is ~1e-14.
  if (alive && hc < T.nc) {
    const int c = hc;
    const int k = T.r_top - c;
    const int rs = (int)T.hrs[k];
    double y0 = 0.0;
    if (rs >= 0) {
      const double r0 = T.hr0[k];
      double ylow;
      if (rs >= r_low) ylow = T.er[(long long)rs * T.nsh + w];
      else if (EXTRAS) ylow = scat_layer(T, rs, T.xf_scat[w]) + cloud_layer(T, rs, T.xf_cloud[w]) + T.ecs[(long long)rs * T.nsh + w];
      else             ylow = T.ecs[(long long)rs * T.nsh + w];
      y0 = slant_bottom(T, rs, r0, ylow, [&](int L) { return T.er[(long long)L * T.nsh + w]; });
    }
    s_y0[c][wi] = y0;
  }
  __syncthreads();

  if (hc < T.nc && s_alive[wi]) {
    const int c = hc;
    const int k = T.r_top - c;
    const int rs = (int)T.hrs[k];
    double t;
    if (rs == -1) {
      t = 0.0;                                            // slantpath.c:37-38: the outermost layer
    } else if (rs < 0) {
      t = nan(""); *T.status = 3;                         // slantpath.c:39-44: the reference exits here
    } else {
      const double y0 = s_y0[c][wi];
      double res;
      if (staged) res = slant_integral(T, k, rs, y0, [&](int L) { return s_y[max(L - r_low, 0)][wi]; });      // (row rs itself: only as the first point, replaced by y0)
      else        res = slant_integral(T, k, rs, y0, [&](int L) { return T.er[(long long)L * T.nsh + w]; });
      t = 2 * res;                                        // slantpath.c:107
    }
    s_tv[c][wi] = T.rad_fct * t;
  }
  __syncthreads();

  if (hc == 0 && s_alive[wi]) {
    bool still = true;
    for (int c = 0; c < T.nc; c++) {
      const int ri = nr - 1 - (T.r_top - c);
      const double tv = s_tv[c][wi];
      T.tau[(long long)ri * T.nsh + w] = tv;
      if (tv > T.toomuch || ri == nr - 1) { T.last[w] = ri; still = false; break; }   // tau.c:277-287, 299-304
    }
    nstill += still;
  }
  if (hc == 0 && ok && T.last[w] >= 0) deep = max(deep, T.last[w] + 1);
  __syncthreads();                                       // s_y0/s_tv/s_alive are reused by the next tile
  }
  tau_publish(T, nstill, deep);
}

// Vertical rays (eclipse geometry, eclipse.c:29-105) in O(1) per height.
// The path abscissa is the radius itself, so the Simpson term of the interval
// pair that starts at layer k,
//     P(k) = (y[k](2-hr) + y[k+1] hf + y[k+2](2-1/hr)) hsum,
// does not depend on where the ray starts, and -- because eclipsetau leaves its
// bottom-point parabola value in er (eclipse.c:65-66) -- neither do the y's once
// layer k has been the bottom.  Hence with A(k) = P(k) + A(k+2):
//     odd  point count:  tau(rs) = A(rs)/6
//     even point count:  tau(rs) = A(rs+1)/6 + h_rs (y[rs]+y[rs+1])/2
// which is the reference's sum re-associated (top-down instead of bottom-up) with
// interval lengths taken as rad[k+1]-rad[k] rather than differences of their running
// sum: ~1e-15 relative.  One lane per wavenumber walks the chunk's layers.
// STAGED (one wave per block, small shards): the step's inputs are fetched into LDS before
// the chain starts -- all loads of a ray independent and in flight together, one memory round
// trip per step instead of one per layer -- and its outputs wait in LDS until the chain is
// done (on this part stores and loads share one in-order counter, so a store in the loop would
// make every wait for a load wait for the store's round trip too).  The chain itself then runs
// at arithmetic latency.
// What a chain of vertical rays needs per layer that does not depend on the wavenumber -- {step, t0,
// twice_sq (parab3_nodes), pair weights p0..p3, rad[rs], 1/step, 1/twice_sq (parab3_recip), t0 + 1.5,
// rad[rs]^2 (parab3_chain)}, kVertLay doubles per layer, made by the host once per run (T.lay; the kernels used to repeat these
// divisions in every block, behind two dependent loads) -- goes to LDS for the step's layers:
// s_lay[kVertLay*c ...] for layer c of the step (ray bottom rs = r_top - c), and the radii
// s_rad[j] = rad[r_top + 1 - j].  Read per layer from global memory (the same for every lane, but
// fetched as vector loads) they put a memory round trip -- ~1 us next to a running walk -- into
// every step of the chain.  The caller synchronises the block afterwards.
constexpr int kVertLay = 12;
__device__ __forceinline__ void vertical_stage_layers(const TauArgs &T, double *s_rad, double *s_lay)
{
  const int nr = T.nr;
  for (int j = threadIdx.x; j < T.nc + 3; j += blockDim.x) {
    const int r = T.r_top + 1 - j;
    s_rad[j] = (r >= 0 && r < nr) ? T.rad[r] : 0.0;
  }
  for (int k = threadIdx.x; k < (T.nc + 1) * kVertLay; k += blockDim.x) {
    const int c = k / kVertLay, rs = T.r_top - c;
    s_lay[k] = (c < T.nc && rs >= 0) ? T.lay[(long long)kVertLay * rs + (k - c * kVertLay)] : 0.0;
  }
}

// a ray's state between layers: the (edited) extinction of the two layers above the current bottom,
// and the running Simpson sums A(rs+1), A(rs+2)
struct VertRay { double y1, y2, a1, a2; };

// One layer of the chain.  SHORT: the 1- and 2-point rays of the top two layers may occur.
// The two Simpson forms are selected, not branched on, so that consecutive layers make one
// straight piece of code.
template <bool SHORT>
__device__ __forceinline__ void vertical_layer(const TauArgs &T, const double *s_rad, const double *s_lay, int c,
                                               double yraw, double ybelow, VertRay &R, double &y0, double &tv)
{
  const int rs = T.r_top - c, n = T.nr - rs;
  const double *L = s_lay + kVertLay * c;
  y0 = yraw;
  if (SHORT && n == 1) {
    tv = 0.0;                                           // eclipse.c:45-46
  } else if (SHORT && n == 2) {                         // eclipse.c:65, 68-80 (value not kept)
    // needs the layer below: it belongs to this chunk (the first chunk has >= 3 layers)
    const double yp = parab3(s_rad[c + 2], s_rad[c + 1], ybelow, yraw, R.y1, s_rad[c + 1]);
    const double *g = T.gw + (long long)rs * T.gstride;
    tv = T.rad_fct * (((yp * g[0] + ((R.y1 + yp) / 2.0) * g[1] + R.y1 * g[2]) * g[3]) / 6.0);
  } else {
    // (the three divisions of a layer as residual-corrected products, trx_numerics.h: the
    // chain is one wave issuing ~6 clocks per instruction, a division is 13 of them)
    y0 = parab3_chain(L[0], L[8], L[1], L[10], L[2], L[9], yraw, R.y1, R.y2, L[7], L[11]);   // kept: eclipse.c:66
    const double a0 = (y0 * L[3] + R.y1 * L[4] + R.y2 * L[5]) * L[6] + R.a2;
    const bool odd = n & 1;
    const double sixth = quotient_rn(odd ? a0 : R.a1, 6.0, 1.0 / 6.0);
    const double with_first = sixth + L[0] * (y0 + R.y1) / 2;
    tv = T.rad_fct * (odd ? sixth : with_first);
    R.a2 = R.a1; R.a1 = a0;
  }
  R.y2 = R.y1; R.y1 = y0;
}

template <bool STAGED, bool EXTRAS>      // EXTRAS: a scattering or cloud model is switched on
__global__ __launch_bounds__(256)
void k_optical_depth_vertical(TauArgs T)
{
  if (!T.eager && T.flags[0] == 0) return;
  latency_critical();
  __shared__ double s_out[STAGED ? 2 * kMaxChunk * 64 : 1];
  __shared__ double s_in[STAGED ? (kMaxChunk + 1) * 64 : 1];
  __shared__ double s_rad[kMaxChunk + 3];
  __shared__ double s_lay[(kMaxChunk + 1) * kVertLay];
  const int nr = T.nr;
  vertical_stage_layers(T, s_rad, s_lay);
  __syncthreads();
  int nstill = 0, deep = 0;
  for (long long w = (long long)blockIdx.x * blockDim.x + threadIdx.x; w < T.nsh; w += (long long)gridDim.x * blockDim.x) {
  if (T.last[w] < 0) {
    bool still = true;
    VertRay R{0.0, 0.0, T.acc[w], T.acc[T.nsh + w]};      // A(rs+1), A(rs+2) on entry of a step
    if (T.r_top + 1 < nr) R.y1 = T.er[(long long)(T.r_top + 1) * T.nsh + w];
    if (T.r_top + 2 < nr) R.y2 = T.er[(long long)(T.r_top + 2) * T.nsh + w];
    double xfs = 0.0, xfc = 0.0;                           // this ray's wavenumber factors of the two models
    if (EXTRAS) { xfs = T.xf_scat[w]; xfc = T.xf_cloud[w]; }
    auto total_ext = [&](int c) -> double {
      if (c >= T.nc) return 0.0;
      const int rs = T.r_top - c;
      const long long k = (long long)rs * T.nsh + w;
      if (EXTRAS) return T.e[k] + scat_layer(T, rs, xfs) + cloud_layer(T, rs, xfc) + T.ecs[k];   // tau.c:231-232
      return T.e[k] + T.ecs[k];
    };
    double q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0, q5 = 0;
    if (STAGED) {
#pragma unroll 8
      for (int c = 0; c < T.nc; c++) s_in[c * 64 + threadIdx.x] = total_ext(c);
      s_in[T.nc * 64 + threadIdx.x] = 0.0;
    } else {
      // total extinction of the layers ahead is requested six steps before it is used
      q0 = total_ext(0); q1 = total_ext(1); q2 = total_ext(2); q3 = total_ext(3); q4 = total_ext(4); q5 = total_ext(5);
    }
    int done = 0;
    if constexpr (STAGED) {
      // A ray ends at the first layer whose optical depth passes toomuch (tau.c:277-287, 299-304),
      // but a test per layer would put that layer's whole Simpson sum and division between one
      // parabola and the next.  Results go to LDS whether the ray is still alive or not (only the
      // first `done` are copied out), so the test can wait: once per four layers, the four
      // Simpson sums overlapping the parabola chain.
      bool alive = true;
      auto one = [&](auto short_rays, int c) {
        double y0, tv;
        vertical_layer<decltype(short_rays)::value>(T, s_rad, s_lay, c, s_in[c * 64 + threadIdx.x], s_in[(c + 1) * 64 + threadIdx.x], R, y0, tv);
        s_out[(2 * c) * 64 + threadIdx.x] = y0; s_out[(2 * c + 1) * 64 + threadIdx.x] = tv;
        const int ri = nr - 1 - (T.r_top - c);
        done += alive;                                                    // (no short cuts: no branches)
        alive = alive & !((tv > T.toomuch) | (ri == nr - 1));
      };
      int c = 0;
      for (; alive && c < T.nc && nr - (T.r_top - c) < 3; c++) one(std::true_type(), c);
      for (; alive && c + 4 <= T.nc; c += 4) {
        one(std::false_type(), c); one(std::false_type(), c + 1);
        one(std::false_type(), c + 2); one(std::false_type(), c + 3);
      }
      for (; alive && c < T.nc; c++) one(std::false_type(), c);
      if (!alive) { T.last[w] = nr - 1 - (T.r_top - (done - 1)); still = false; }
      for (int d = 0; d < done; d++) {
        const int rs = T.r_top - d;
        if (T.er_all || d >= T.nc - 2) T.er[(long long)rs * T.nsh + w] = s_out[(2 * d) * 64 + threadIdx.x];
        T.tau[(long long)(nr - 1 - rs) * T.nsh + w] = s_out[(2 * d + 1) * 64 + threadIdx.x];
      }
    } else {
      for (int c = 0; c < T.nc; c++) {
        const int rs = T.r_top - c, ri = nr - 1 - rs;
        const double yraw = q0, ybelow = q1;
        q0 = q1; q1 = q2; q2 = q3; q3 = q4; q4 = q5; q5 = total_ext(c + 6);
        double y0, tv;
        vertical_layer<true>(T, s_rad, s_lay, c, yraw, ybelow, R, y0, tv);
        if (T.er_all || c >= T.nc - 2) T.er[(long long)rs * T.nsh + w] = y0;
        T.tau[(long long)ri * T.nsh + w] = tv;
        if (tv > T.toomuch || ri == nr - 1) { T.last[w] = ri; still = false; break; }   // tau.c:277-287, 299-304
      }
    }
    T.acc[w] = R.a1; T.acc[T.nsh + w] = R.a2;
    nstill += still;
  }
  if (T.last[w] >= 0) deep = max(deep, T.last[w] + 1);
  }
  tau_publish(T, nstill, deep);
}

// ---------------------------------------------------------------------------
// spectrum
// ---------------------------------------------------------------------------
struct EmisArgs {
  int nr, nang; long long nsh, lo;
  double wn_i, wn_d, wn_fct;
  const double *tau; const int *last; const double *temp;
  double cosang[kMaxAngles], area[kMaxAngles];
  double rcos[kMaxAngles];           // 1 / cosang where the host has checked that quotient_rn(x, cos, 1/cos) is the division (else 0: divide)
  double *intens;                    // [nang][nsh]
  double *flux;                      // [nsh]
  const double *e2tab;               // [64] 2^(j/64)
};

// -tau / cos(angle) (eclipse.c:140): the correctly rounded quotient either way -- five multiply-adds
// with the reciprocal the host has made and checked (trx_numerics.h quotient_rn), or the division
__device__ __forceinline__ double slant_depth(const EmisArgs &E, int a, double tv)
{
  return E.rcos[a] != 0.0 ? quotient_rn(-tv, E.cosang[a], E.rcos[a]) : -tv / E.cosang[a];      // (wave-uniform choice)
}

// B = num / (e^x - 1) = num e^-x / (1 - e^-x),  x = h nu / k T > 0, with the kernels' own exponential
__device__ __forceinline__ double planck_from(double num, double x, const double *e2tab)
{
  const double em = exp_neg(-x, e2tab);
  return num * em / (1.0 - em);
}

// eclipse.c:118-160 (eclipse_intens) + eclipse.c:243-287 (flux).
// One wavefront per wavenumber, lanes = heights: every lane evaluates the Planck function and
// the nang attenuations exp(-tau/cos) of its own height, takes its upper neighbour's through a
// lane shift, and the trapezoid terms (numerical.c:168-170) are added over the wave in a fixed
// butterfly order.  (The ray's ~80 heights used to be one lane's serial loop of ~500 dependent
// exponentials: 38 us of pure latency at the end of every spectrum.)
constexpr int kEmisWaves = 4;

// 2 h nu^3 c^2 of the Planck function (the cube as two products: the library's general power was 450 instructions
// at the head of every ray's emission)
__device__ __forceinline__ double planck_num(double wv) { return 2.0 * kH * (wv * wv * wv) * kLs * kLs; }

// One ray by one wavefront; tau_at(i): optical depth at height i of this ray; s_e2: 2^(j/64) in LDS.
// A lane adds the trapezoid terms of its heights (one per pass of 64) for itself; ONE butterfly sum per angle
// behind the last pass.
//
// emission_ray_exact<N>: exactly N angles, every one with a checked reciprocal of its cosine -- no branch per angle
// (the angles' chains of exponentials are issued side by side), no test of a lane's height inside the angle loop (a
// lane past the ray's end computes exp(-0) and multiplies it with a zero weight).
template <int N, class TauAt, class PlanckAt>
__device__ __forceinline__ void emission_ray_exact(const EmisArgs &E, long long w, int last, int lane, const double *s_e2, TauAt tau_at, PlanckAt planck_at)
{
  double acc[N], dt[N], dt_c[N];
#pragma unroll
  for (int a = 0; a < N; a++) { acc[a] = 0.0; dt[a] = 0.0; dt_c[a] = 0.0; }
  double B = 0.0, B_c = 0.0;                  // B_c, dt_c: the previous pass's last height (wave-uniform)
  int top = 0;
  for (int i0 = 0; i0 <= last; i0 += 64) {
    const int i = i0 + lane;
    const bool have = i <= last;
    double tv = 0.0;
    B = 0.0;
    if (have) {
      tv = tau_at(i);
      B = planck_at(i0, i);
    }
    double Bp = dpp_f64<0x138>(B);              // wave_shr:1 -- the lane before (lane 0: the carry)
    if (lane == 0) Bp = B_c;
    const double Bs = (have && i > 0) ? B + Bp : 0.0;
    const double ntv = -tv;
#pragma unroll
    for (int a = 0; a < N; a++) {
      const double d = exp_neg(quotient_rn(ntv, E.cosang[a], E.rcos[a]), s_e2);
      double dp = dpp_f64<0x138>(d);
      if (lane == 0) dp = dt_c[a];
      acc[a] += (d - dp) * Bs;
      dt_c[a] = readlane_f64(d, 63);
      dt[a] = d;
    }
    B_c = readlane_f64(B, 63);
    top = min(63, last - i0);                   // lane of the ray's last height
  }
  const double B_last = readlane_f64(B, top);
  double fl = 0.0;
#pragma unroll
  for (int a = 0; a < N; a++) {
    const double I = B_last * readlane_f64(dt[a], top) - 0.5 * wave_sum(acc[a]);
    if (lane == a) E.intens[(long long)a * E.nsh + w] = I;
    fl += kPi * I * E.area[a];
  }
  if (lane == 0) E.flux[w] = fl;
}

// Any number of angles up to NMAX, a division where an angle has no checked reciprocal: the same terms and sums,
// angle after angle.
template <int NMAX, class TauAt, class PlanckAt>
__device__ __forceinline__ void emission_ray_any(const EmisArgs &E, long long w, int last, int lane, const double *s_e2, TauAt tau_at, PlanckAt planck_at)
{
  double acc[NMAX], dt_c[NMAX], dt_last[NMAX];
#pragma unroll
  for (int a = 0; a < NMAX; a++) { acc[a] = 0.0; dt_c[a] = 0.0; dt_last[a] = 0.0; }
  double B_c = 0.0, B_last = 0.0;             // carry: the previous pass's last height
  for (int i0 = 0; i0 <= last; i0 += 64) {
    const int i = i0 + lane;
    const bool have = i <= last;
    double B = 0.0, tv = 0.0;
    if (have) {
      tv = tau_at(i);
      B = planck_at(i0, i);
    }
    double Bp = dpp_f64<0x138>(B);              // wave_shr:1 -- the lane before (lane 0: set below)
    if (lane == 0) Bp = B_c;
    const double Bs = (have && i > 0) ? B + Bp : 0.0;
    const int top = min(63, last - i0);        // lane of the pass's last height
#pragma unroll
    for (int a = 0; a < NMAX; a++) {
      if (a < E.nang) {
        const double d = exp_neg(slant_depth(E, a, tv), s_e2);
        double dp = dpp_f64<0x138>(d);
        if (lane == 0) dp = dt_c[a];
        acc[a] += (d - dp) * Bs;
        dt_c[a] = readlane_f64(d, 63);
        dt_last[a] = readlane_f64(d, top);
      }
    }
    B_c = readlane_f64(B, 63);
    B_last = readlane_f64(B, top);
  }
  // B_last and dt_last are wave-uniform: lane a writes angle a, lane 0 the flux in angle order
  double fl = 0.0;
#pragma unroll
  for (int a = 0; a < NMAX; a++) {
    if (a < E.nang) {
      const double I = B_last * dt_last[a] - 0.5 * wave_sum(acc[a]);
      if (lane == a) E.intens[(long long)a * E.nsh + w] = I;
      fl += kPi * I * E.area[a];
    }
  }
  if (lane == 0) E.flux[w] = fl;
}

// (NMAX: the angles the general form holds state for -- 16 of them are 96 registers)
// (planck_at(i0, i): the Planck function of the ray's wavenumber at height i = i0 + lane -- ray_planck below, or
// values the caller has made ahead of the optical depths, k_ray_tail)
template <int NMAX = kMaxAngles, class TauAt, class PlanckAt>
__device__ __forceinline__ void emission_ray(const EmisArgs &E, long long w, int last, int lane, const double *s_e2, TauAt tau_at, PlanckAt planck_at)
{
  bool recip = E.nang <= 8;
#pragma unroll
  for (int a = 0; a < 8; a++) if (a < E.nang && E.rcos[a] == 0.0) recip = false;      // (wave-uniform: kernel arguments)
  if (recip) {
    switch (E.nang) {
      case 1: return emission_ray_exact<1>(E, w, last, lane, s_e2, tau_at, planck_at);
      case 2: return emission_ray_exact<2>(E, w, last, lane, s_e2, tau_at, planck_at);
      case 3: return emission_ray_exact<3>(E, w, last, lane, s_e2, tau_at, planck_at);
      case 4: return emission_ray_exact<4>(E, w, last, lane, s_e2, tau_at, planck_at);
      case 5: return emission_ray_exact<5>(E, w, last, lane, s_e2, tau_at, planck_at);
      case 6: return emission_ray_exact<6>(E, w, last, lane, s_e2, tau_at, planck_at);
      case 7: return emission_ray_exact<7>(E, w, last, lane, s_e2, tau_at, planck_at);
      case 8: return emission_ray_exact<8>(E, w, last, lane, s_e2, tau_at, planck_at);
      default: break;
    }
  }
  emission_ray_any<NMAX>(E, w, last, lane, s_e2, tau_at, planck_at);
}

// the Planck function of ray w at height i (eclipse.c:131-134)
struct RayPlanck {
  double num, ex; const double *temp; int nr; const double *e2;
  __device__ __forceinline__ RayPlanck(const EmisArgs &E, long long w, const double *s_e2)
  {
    const double wv = (E.wn_i + (double)(E.lo + w) * E.wn_d) * E.wn_fct;
    num = planck_num(wv); ex = kH * wv * kLs; temp = E.temp; nr = E.nr; e2 = s_e2;
  }
  __device__ __forceinline__ double at_temp(double tk) const { return planck_from(num, ex / (kKb * tk), e2); }
  __device__ __forceinline__ double operator()(int, int i) const { return at_temp(temp[nr - 1 - i]); }
};

__global__ __launch_bounds__(64 * kEmisWaves)
void k_emission(EmisArgs E)
{
  latency_critical();
  __shared__ double s_e2[64];                 // 2^(j/64) for exp_neg (the kernels' own exponential, ~1.5 ulp)
  if (threadIdx.x < 64) s_e2[threadIdx.x] = E.e2tab[threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const long long w = (long long)blockIdx.x * kEmisWaves + (threadIdx.x >> 6);     // wave-uniform
  if (w >= E.nsh) return;
  const int last = __builtin_amdgcn_readfirstlane(E.last[w]);      // (< 0: the ray is still descending -- provisional spectrum, zero)
  emission_ray(E, w, last, lane, s_e2, [&](int i) { return E.tau[(long long)i * E.nsh + w]; }, RayPlanck(E, w, s_e2));
}

// The same for large grids (more than kEmisRowsAbove wavenumbers in the JOB's grid, so that every
// shard of a job takes the same form): one LANE per wavenumber going down its heights -- the loads
// of a height are one coalesced row segment, no lane shifts, no wave sums; with 10^7 rays there are
// waves enough to hide a lane's chain of exponentials (configs[4]: 27 -> 5 ms).  The trapezoid
// terms are added in height order (the wave form adds them in its butterfly order).
constexpr long long kEmisRowsAbove = 65536;

__global__ __launch_bounds__(256)
void k_emission_rows(EmisArgs E)
{
  __shared__ double s_e2[64];
  if (threadIdx.x < 64) s_e2[threadIdx.x] = E.e2tab[threadIdx.x];
  __syncthreads();
  const long long w = (long long)blockIdx.x * 256 + threadIdx.x;
  if (w >= E.nsh) return;
  const double wv = (E.wn_i + (double)(E.lo + w) * E.wn_d) * E.wn_fct;
  const int last = E.last[w];                  // (< 0: the ray is still descending -- provisional spectrum, zero)
  const double pl_num = planck_num(wv);
  const double pl_exp = kH * wv * kLs;
  double sum[kMaxAngles], dtp[kMaxAngles];
#pragma unroll
  for (int a = 0; a < kMaxAngles; a++) { sum[a] = 0.0; dtp[a] = 0.0; }
  double Bp = 0.0;
  for (int i = 0; i <= last; i++) {
    const double tv = E.tau[(long long)i * E.nsh + w];
    const double B = planck_from(pl_num, pl_exp / (kKb * E.temp[E.nr - 1 - i]), s_e2);
#pragma unroll
    for (int a = 0; a < kMaxAngles; a++) {
      if (a < E.nang) {
        const double dt = exp_neg(slant_depth(E, a, tv), s_e2);
        if (i > 0) sum[a] += (dt - dtp[a]) * (B + Bp);
        dtp[a] = dt;
      }
    }
    Bp = B;
  }
  double fl = 0.0;
#pragma unroll
  for (int a = 0; a < kMaxAngles; a++) {
    if (a < E.nang) {
      const double I = Bp * dtp[a] - 0.5 * sum[a];
      E.intens[(long long)a * E.nsh + w] = I;
      fl += kPi * I * E.area[a];
    }
  }
  E.flux[w] = fl;
}

struct ModArgs {
  int nr, modlevel, transparent; long long nsh;
  double toomuch, ip_fct, srad;
  const double *tau; const int *last;
  const double *ip;                  // [nr] impact parameters, top first (makesample.c:564-574)
  // Simpson weights per point count (slantpath.c:399-408): row "cnt"
  const double *gw; int gstride; const double *gh0;
  double *out;                       // [nsh]
  int *status;                       // set to 1 when modlevel -1 cannot be evaluated
  int *status_slots;                 // null, or [4] in pinned host memory: slot "code" is set instead of the atomic maximum on status (k_ray_tail with the host adding up; the host takes the highest slot set)
  __device__ __forceinline__ void raise(int code) const { if (status_slots) status_slots[code] = 1; else atomicMax(status, code); }
};

// slantpath.c:351-436 (modulation1) and :447-473 (modulationm1).
// One wavefront per wavenumber, lanes = Simpson interval pairs of the ray's radial integral:
// every lane evaluates the three integrand points of its pair (exp(-tau) * b), the pairs are
// added over the wave in a fixed butterfly.  (One lane per wavenumber looping over ~35 pairs of
// three exponentials each was 40 us of latency at the end of every transmission spectrum.)
constexpr int kModWaves = 4;

// one ray by one wavefront; tau_at(i): optical depth at height i of this ray; status codes are
// raised with an atomic maximum (several rays may raise them at once; k_ray_tail reads the word back)
template <class TauAt>
__device__ __forceinline__ void modulation_ray(const ModArgs &M, long long w, int last, int lane, TauAt tau_at)
{
  const int nr = M.nr;
  // a ray that is still descending (only possible in the provisional spectrum of a run that
  // stopped at the previous run's depth and will go on): nothing to integrate yet
  if (last < 0) { if (lane == 0) M.out[w] = 0.0; return; }
  if (M.modlevel == -1) {
    if (lane != 0) return;
    const double tl = tau_at(last);
    if (tl < M.toomuch) { M.out[w] = -1; M.raise(1); return; }
    int ini = last + 1 - 2; if (ini < 0) ini = 0;
    // interp_line(tau+ini, ipv, toomuch), numerical.c:202-211
    const double x0 = tau_at(ini), x1 = tau_at(ini + 1);
    const double y0 = M.ip[ini] * M.ip_fct, y1 = M.ip[ini+1] * M.ip_fct;
    const double r = y0 + (M.toomuch - x0) * ((y1 - y0) / (x1 - x0));
    M.out[w] = r * r / (M.srad * M.srad);
    return;
  }
  const double tlast = readlane_f64(tau_at(last), 0);
  const double maxtau = tlast > M.toomuch ? tlast : M.toomuch;
  // integrand on ascending radius: index q = 0..cnt-1 maps to height i = cnt-1-q
  int lastp = last + 1; if (lastp > nr - 1) lastp = nr - 1;
  const int cnt = lastp + 1;                               // points, including the zero pad
  if (cnt < 3) { if (lane == 0) { M.out[w] = nan(""); M.raise(2); } return; }
  const double *g = M.gw + (long long)cnt * M.gstride;
  auto val = [&](int q) -> double {
    const int i = cnt - 1 - q;
    if (i > last) return 0.0;                              // slantpath.c:383-386
    const double b = M.ip[i] * M.ip_fct;
    return exp(-tau_at(i)) * b;
  };
  const int even = (cnt % 2 == 0), npair = (cnt - 1) / 2;
  double acc = 0.0;
  for (int i0 = 0; i0 < npair; i0 += 64) {
    const int i = i0 + lane;
    double term = 0.0;
    if (i < npair) {
      const int j = 2*i + even;
      term = (val(j) * g[4*i] + val(j+1) * g[4*i+1] + val(j+2) * g[4*i+2]) * g[4*i+3];
    }
    acc += wave_sum(term);
  }
  if (lane != 0) return;
  double res = acc / 6.0;
  if (even) res += M.gh0[cnt] * (val(0) + val(1)) / 2;
  const double rtop = M.ip[0] * M.ip_fct;
  res = rtop * rtop - 2.0 * res;
  if (M.transparent) {
    const double bmin = M.ip[cnt - 1] * M.ip_fct;
    res -= exp(-maxtau) * bmin * bmin;
  }
  res *= 1.0 / (M.srad * M.srad);
  M.out[w] = res;
}

__global__ __launch_bounds__(64 * kModWaves)
void k_modulation(ModArgs M)
{
  latency_critical();
  const int lane = threadIdx.x & 63;
  const long long w = (long long)blockIdx.x * kModWaves + (threadIdx.x >> 6);      // wave-uniform
  if (w >= M.nsh) return;
  const int last = __builtin_amdgcn_readfirstlane(M.last[w]);
  const double *tw = M.tau + w;                           // tw[i*nsh] = tau[i][w]
  modulation_ray(M, w, last, lane, [&](int i) { return tw[(long long)i * M.nsh]; });
}

// The same for large grids (like k_emission_rows, by the job's grid): one lane per wavenumber,
// the interval pairs of its radial integral in order.
__global__ __launch_bounds__(256)
void k_modulation_rows(ModArgs M)
{
  const long long w = (long long)blockIdx.x * 256 + threadIdx.x;
  if (w >= M.nsh) return;
  const int nr = M.nr;
  const int last = M.last[w];
  if (last < 0) { M.out[w] = 0.0; return; }
  const double *tw = M.tau + w;                           // tw[i*nsh] = tau[i][w]
  if (M.modlevel == -1) {
    const double tl = tw[(long long)last * M.nsh];
    if (tl < M.toomuch) { M.out[w] = -1; *M.status = 1; return; }
    int ini = last + 1 - 2; if (ini < 0) ini = 0;
    const double x0 = tw[(long long)ini * M.nsh], x1 = tw[(long long)(ini+1) * M.nsh];
    const double y0 = M.ip[ini] * M.ip_fct, y1 = M.ip[ini+1] * M.ip_fct;
    const double r = y0 + (M.toomuch - x0) * ((y1 - y0) / (x1 - x0));
    M.out[w] = r * r / (M.srad * M.srad);
    return;
  }
  const double tlast = tw[(long long)last * M.nsh];
  const double maxtau = tlast > M.toomuch ? tlast : M.toomuch;
  int lastp = last + 1; if (lastp > nr - 1) lastp = nr - 1;
  const int cnt = lastp + 1;                               // points, including the zero pad
  if (cnt < 3) { M.out[w] = nan(""); *M.status = 2; return; }
  const double *g = M.gw + (long long)cnt * M.gstride;
  auto val = [&](int q) -> double {
    const int i = cnt - 1 - q;
    if (i > last) return 0.0;                              // slantpath.c:383-386
    const double b = M.ip[i] * M.ip_fct;
    return exp(-tw[(long long)i * M.nsh]) * b;
  };
  const int even = (cnt % 2 == 0), npair = (cnt - 1) / 2;
  double acc = 0.0;
  double va = val(even);                                   // a pair's last point is the next pair's first
  for (int i = 0; i < npair; i++) {
    const int j = 2*i + even;
    const double vb = val(j+1), vc = val(j+2);
    acc += (va * g[4*i] + vb * g[4*i+1] + vc * g[4*i+2]) * g[4*i+3];
    va = vc;
  }
  double res = acc / 6.0;
  if (even) res += M.gh0[cnt] * (val(0) + val(1)) / 2;
  const double rtop = M.ip[0] * M.ip_fct;
  res = rtop * rtop - 2.0 * res;
  if (M.transparent) {
    const double bmin = M.ip[cnt - 1] * M.ip_fct;
    res -= exp(-maxtau) * bmin * bmin;
  }
  res *= 1.0 / (M.srad * M.srad);
  M.out[w] = res;
}

}  // namespace trx
