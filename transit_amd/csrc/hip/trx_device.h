// trx_device.h -- plain structs shared by the host API (trx_api.hip) and the
// kernels (trx_kernels.hip.h).  Everything here is laid out for HBM:
//   * SoA, 8/4/2/1-byte columns, no row pointers;
//   * per-layer arrays are [layer][x] with x contiguous so that consecutive
//     lanes (lines, groups or wavenumbers) read consecutive addresses.
#pragma once
#include <stdint.h>

namespace trx {

constexpr int kMaxChunk   = 32;    // layers swept per top-down step (upper bound)
constexpr int kMaxIso     = 256;   // isotopes per run (per-isotope tables of a block live in LDS)
constexpr int kMaxAngles  = 16;
constexpr int kMaxDop     = 256;   // Doppler-width samples (ndop)
constexpr int kTileBins   = 4;     // coarse bins per wavefront tile in the accumulate kernel
constexpr int kTabPad     = 512;   // zero floats in front of and behind the Voigt tables (>= bins per wide tile, trx_rows.hip.h's too)

// One distinct Voigt profile of the table (opacity.c:258-270, getprofile).
struct ProfileJob {
  int64_t off;        // first float of the profile in the table
  int32_t nv;         // points (odd)
  int32_t regime;     // 0 quick (point samples), 1 fine (edge mean), 2 coarse (Simpson mean)
  int32_t m;          // sub-intervals per bin (regime 2)
  int32_t pad;
  double  half;       // dwn * (nv/2)
  double  sub;        // spacing of the evaluation points
  double  alphaL, alphaD;
  int64_t first_bin;  // prefix sum of nv over jobs (global bin index of bin 0)
};

// Static line-list description (layer independent), device pointers.
struct LinesDev {
  int64_t nlines;
  const double  *wavn;     // [nlines] 1/(wl*1e-4), cm-1
  const double  *elow;     // [nlines]
  const double  *gf;       // [nlines]
  const int16_t *iso;      // [nlines]
  const uint8_t *inrange;  // [nlines] extinction.c:410
  const int32_t *lgroup;   // [nlines] group index when the line anchors a co-added group, else -1
  int64_t ngroups;
  const int32_t *gfirst;   // [ngroups] first line of the co-added group (extinction.c:449-462)
  const int32_t *gcount;   // [ngroups] members
  const int32_t *giown;    // [ngroups] fine-grid index of the anchor (extinction.c:445-447)
  const int16_t *giso;     // [ngroups]
  const double  *gwavn;    // [ngroups] anchor wavenumber
  // isotope blocks: groups of isotope b are [gblock[b], gblock[b+1]) sorted by descending iown
  const int32_t *gblock;   // [niso+1]
  // cnt_ge[b*(nwn+1) + k] = number of groups of block b with iown/osamp >= k
  const int32_t *cnt_ge;
};

// Per-run, per-layer scalars prepared on the host in the reference's own
// arithmetic (extinction.c:364-395) -- [layer][iso] unless noted.
struct LayerDev {
  const double *negc_over_t;  // [layer]  -EXPCTE/T
  const double *strength_f;   // SIGCTE*isoratio/(m*Z)          (extinction.c:464)
  const double *density;      // density of the isotope's molecule (extinction.c:473)
  const double *alphad;       // Doppler width / wavenumber
  const double *alphal;       // Lorentz width
  const int32_t *idop0;       // nearest aDop index for alphad*wn[0] (extinction.c:393)
  const int32_t *ilor;        // nearest aLor index              (extinction.c:394)
  const int32_t *psmax;       // largest profile half-size the isotope can use in this layer
};

// What k_group_sweep needs to find, per isotope block, the lines whose profiles can reach the
// shard in some layer of the step (one contiguous run per block: the TLI is wavelength-sorted):
// every workgroup builds the table of those runs in LDS itself (up to kMaxIso blocks), from
// psmax of the step's layers and cnt_ge -- a shard of the grid (windowed) or the whole grid.
// (The runs as a kernel argument for lists of few isotopes were measured: no difference.)
struct SweepWindow {
  int windowed, osamp;
  long long lo, hi, nwn;            // shard [lo, hi) of nwn coarse bins
};

}  // namespace trx
