/* transit_hip.h -- C ABI of the MI355X line-by-line spectrum core.
 *
 * Drop-in boundary for the spectrum-computation path of exosports/transit
 * (reference transit/src/transit.c:125-214, do_transit(): interpcs -> extwn ->
 * tau -> emergent_intens/flux | modulation, plus the Voigt-table build of
 * opacity.c:219-277 done once per handle).
 *
 * The reference dispatches this path through a per-ray plugin vtable
 *   ray_solution { name, file, monospace, optdepth(tr,b,ex), spectrum(tr,tau,w,last,toomuch,r) }
 *   (transit/include/structures_tr.h:68-83; instances eclipse.c:408, slantpath.c:573;
 *    selected by name in argum.c:752-765)
 * and through per-layer operators
 *   int computemolext(struct transit*, PREC_RES **kiso, PREC_ATM temp,
 *                     PREC_ATM *density, double *Z, int permol)   (extinction.h:25)
 * which are one-scalar-per-call and cannot feed a GPU.  This ABI keeps the same
 * selection surface (solution = "eclipse" | "transit") and the same data
 * contract, batched: one create (static data: line list, grids, Voigt grid,
 * CIA tables) and one run per atmosphere.  Plain pointers and sizes only; the
 * caller owns every host buffer, the library owns device memory.
 *
 * All functions return 0 on success and a negative trx_status on failure; they
 * never exit()/abort (reference: fw() macro transit.h:91-98 exits).
 */
#ifndef TRANSIT_HIP_H
#define TRANSIT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRX_ABI_VERSION 5

typedef enum {
  TRX_OK            =  0,
  TRX_E_ARG         = -1,   /* bad argument / inconsistent sizes               */
  TRX_E_NOMEM       = -2,   /* host or device allocation failed                */
  TRX_E_HIP         = -3,   /* a HIP runtime call failed (see trx_last_error)  */
  TRX_E_NODEVICE    = -4,   /* no usable gfx950 device                         */
  TRX_E_RANGE       = -5,   /* layer temperature outside CIA / table range     */
  TRX_E_UNSUPPORTED = -6,   /* option combination not implemented              */
  TRX_E_ORDER       = -7,   /* line list not sorted as the TLI format requires */
  TRX_E_NOTREACHED  = -8    /* modlevel -1 and tau never reached toomuch       */
} trx_status;

typedef enum { TRX_SOL_ECLIPSE = 0, TRX_SOL_TRANSIT = 1 } trx_solution;

/* One collision-induced-absorption / cross-section table
 * (reference struct cross, structures_tr.h:296-307; file grammar crosssec.c:87-233). */
typedef struct {
  int32_t nspec;          /* 1 or 2 species                                    */
  int32_t mol[2];         /* indices into the atmosphere species list          */
  int32_t nwave, ntemp;
  const double *wn;       /* [nwave] cm-1                                      */
  const double *temp;     /* [ntemp] K                                         */
  const double *cs;       /* [nwave][ntemp] cm-1 amagat^-nspec                 */
} trx_cia;

/* Pre-computed opacity grid (reference struct opacity, structures_tr.h:154-171;
 * file layout opacity.c:405-421): extinction per unit density of each molecule
 * on (layer, temperature, wavenumber).  When trx_static.ogrid is set, trx_run
 * interpolates it in temperature (interpolmolext, extinction.c:535-581) instead
 * of sweeping the line list. */
typedef struct {
  int64_t nmol, ntemp, nlayer, nwave;
  const int32_t *mol_index;  /* [nmol] atmosphere species index of each grid molecule */
  const double  *temp;       /* [ntemp] K, ascending                                   */
  const double  *o;          /* [nlayer][ntemp][nmol][nwave] cm2 g-1 ... cm-1 per g cm-3 */
} trx_opacity_grid;

/* Everything that does not change between spectra
 * (reference: transit_init(), transit.c:25-74). */
typedef struct {
  int32_t abi_version;    /* TRX_ABI_VERSION                                   */
  int32_t device;         /* HIP device ordinal                                */

  /* wavenumber sampling (makewnsample, makesample.c:309-400)                 */
  double  wn_i;           /* wns.i  first coarse wavenumber, cm-1              */
  double  wn_d;           /* wns.d  coarse spacing, cm-1                       */
  int64_t nwn;            /* wns.n  coarse samples                             */
  int32_t osamp;          /* owns.o oversampling factor (wnosamp)              */
  int64_t nown;           /* owns.n fine samples = (nwn-1)*osamp+1             */
  /* this handle computes coarse bins [wn_lo, wn_hi) only (wavenumber shard of
   * a multi-GPU job); 0, nwn for the whole grid                               */
  int64_t wn_lo, wn_hi;

  /* Voigt-profile grid (opacity.c:219-277; hint fields are float,
   * structures_tr.h:326-333)                                                  */
  int32_t ndop, nlor;
  float   dmin, dmax, lmin, lmax;
  float   timesalpha;     /* nwidth                                            */

  /* line transitions, SoA as in struct line_transition (structures_tr.h:95-102);
   * TLI order: isotope blocks, ascending wavelength inside a block            */
  int64_t nlines;
  const double  *wl_um;   /* [nlines] wavelength, microns                      */
  const int16_t *isoid;   /* [nlines] cumulative isotope index                 */
  const double  *elow;    /* [nlines] lower-state energy, cm-1                 */
  const double  *gf;      /* [nlines]                                          */

  /* isotopes (struct isotopes, structures_tr.h)                               */
  int32_t niso;
  const double  *iso_mass;   /* [niso] amu                                     */
  const double  *iso_ratio;  /* [niso]                                         */
  const int32_t *iso_imol;   /* [niso] index into the species list             */

  /* atmosphere species (struct molecules)                                     */
  int32_t nmol;
  const double  *mol_mass;   /* [nmol] amu                                     */
  const double  *mol_radius; /* [nmol] cm                                      */
  const double  *mol_pol;    /* [nmol] polarizability A^3 (scattering flag 2)  */
  const int32_t *mol_is_h2;  /* [nmol] 1 for the species named "H2" (cloud P19)*/

  /* CIA tables                                                                */
  int32_t ncia;
  const trx_cia *cia;

  /* multi-GPU job: communicator from trx_comm_create (NULL for one GPU); used by
   * trx_gather only.  A handle whose shard [wn_lo, wn_hi) is a part of the grid
   * sweeps only the lines whose profiles can reach it. */
  void   *comm;
  int32_t nranks, rank;

  /* optional pre-computed opacity grid (NULL: line-by-line)                   */
  const trx_opacity_grid *ogrid;
} trx_static;

/* Per-spectrum atmosphere, already on transit's layer grid, bottom layer first
 * (reference: makeradsample(), makesample.c:409-549; the density[]/Z[] vectors
 * handed to computemolext at tau.c:164-167, 254-257). */
typedef struct {
  int32_t nlayer;
  double  rad_fct;        /* rads.fct: radius units -> cm                      */
  const double *radius;   /* [nlayer] rads.v, units of rad_fct                 */
  const double *temp;     /* [nlayer] K (atm.t * tfct; the reference's Planck
                             and scattering terms read atm.t without tfct,
                             eclipse.c:155 -- identical when ut = 1)            */
  const double *press;    /* [nlayer] atm.p as the reference stores it, i.e. in
                             atmosphere-file units; its cloud and scattering
                             models (extinction.c:608,661) assume these are bar */
  const double *density;  /* [nmol][nlayer] g cm-3                             */
  const double *abund;    /* [nmol][nlayer] mixing ratio q (cloud/scatter only; may be NULL) */
  const double *zpart;    /* [niso][nlayer] partition function Z_i(T_layer)    */
} trx_atm;

/* Per-spectrum options (reference: struct transithint fields accepted in
 * argum.c:774-911 and tau.c:12-52). */
typedef struct {
  int32_t solution;       /* trx_solution                                      */
  double  toomuch;        /* optical-depth cut (tau.c:277)                     */
  double  ethresh;        /* line-strength threshold (extinction.c:467)        */
  double  wn_fct;         /* wns.fct (output wavenumber units factor)          */
  /* eclipse */
  int32_t nangles;
  const double *angles_deg;   /* raygrid                                       */
  /* transit */
  double  starrad_cm;     /* sg->starrad * sg->starradfct                      */
  int32_t transparent;    /* sg->transpplanet                                  */
  int32_t modlevel;       /* 1 or -1                                           */
  /* clouds (struct extcloud) and scattering (struct extscat)                  */
  int32_t cloud_flag;     /* 0 none, 1 ext, 2 opa, 3 B17, 4 F18, 5 P19         */
  double  cloud_ext, cloud_top, cloud_bot, cloud_gamma, cloud_Q, cloud_r,
          cloud_sig, cloud_refwn;
  int32_t scat_flag;      /* 0 none, 1 Lecavelier, 2 polarizability            */
  double  scat_logext;
  /* execution knobs (no effect on results)                                    */
  int32_t layer_chunk;    /* layers swept per top-down step; 0 = automatic: up to 64 where the
                             profiles are narrow (one kernel walks the line list with one lane
                             per layer), up to 32 otherwise; the depth the previous spectrum
                             reached is split into equal steps                          */
  int32_t eager;          /* 1 = sweep every layer (debug dumps of all layers) */
  int32_t profile;        /* 1 = bracket the production kernels with HIP events on their own
                             streams (trx_stats ms_* timings: the run's own plan, queues and kernels,
                             with the events' packets between them); 2 = also count evaluated /
                             skipped groups and bins with the instrumented kernel variants
                             (trx_stats neval/nskip/sum_bins; those kernels run slower)   */
} trx_opts;

/* Optional intermediate outputs (host buffers, any may be NULL).  They mirror
 * the reference's --savefiles dumps (tau.c:180-190, 293-329). */
typedef struct {
  double  *e;             /* [nlayer][nwn_shard] molecular extinction          */
  double  *e_cs;          /* [nlayer][nwn_shard] CIA extinction (ref: [wn][layer]) */
  double  *tau;           /* [nwn_shard][nlayer] optical depth                 */
  int64_t *last;          /* [nwn_shard]                                       */
  double  *intens;        /* [nangles][nwn_shard] (eclipse)                    */
  uint8_t *computed;      /* [nlayer] 1 if the layer was swept                 */
  /* the arrays behind the reference's total/cloud/scatt_extion.dat dumps (tau.c:293-329):      */
  double  *er;            /* [nlayer][nwn_shard] total extinction as the ray solution left it:
                             defined for the layers a ray went through (layer >= nlayer-1-last);
                             eclipse geometry keeps its bottom-point parabola values in it
                             (eclipse.c:65-66)                                               */
  double  *e_scat;        /* [nlayer][nwn_shard] scattering extinction (extinction.c:587-624) */
  double  *e_cloud;       /* [nlayer][nwn_shard] cloud extinction (extinction.c:630-693)     */
} trx_debug;

/* Counters and device timings of the last trx_run (reference DEBUG counters
 * extinction.c:513-518; stage timers transitstd.c:359-374). */
typedef struct {
  int64_t nlines_inrange; /* lines passing the range test (extinction.c:410)   */
  int64_t ngroups;        /* co-added groups (anchors), layer independent      */
  int64_t nadd;           /* co-added lines per layer (layer independent)      */
  int64_t layers_swept;
  int64_t neval;          /* evaluated groups, summed over swept layers (these three
                             counters are collected by counting runs only: profile 2) */
  int64_t nskip;          /* groups below ethresh*kmax, summed over layers     */
  int64_t sum_bins;       /* accumulated (group,layer,bin) triples             */
  int64_t table_floats;   /* Voigt table size                                  */
  double  ms_create_table;/* device time of the Voigt-table build              */
  double  ms_run_total;   /* device time of the last run, first to last kernel (profiled runs, trx_opts.profile; else 0) */
  double  ms_sweep;       /* line-sweep kernels (profile >= 1; as ms_k_* and ms_tau) */
  double  ms_k_sweep;      /* sum over launches of k_group_sweep (the steps whose profiles are wider than the walk's widest frame) */
  double  ms_k_walk;       /* sum over launches of the walk kernels (k_line_walk, k_line_walk_lanes, k_line_walk_packed)      */
  double  ms_k_accum;     /* sum over launches of k_walk_combine and k_accumulate(_wide/_rows) */
  int64_t sweep_launches; /* launches of each sweep kernel (gated no-op ones too) */
  double  ms_tau;         /* optical-depth kernels                             */
  double  ms_cia;         /* host wall time of queueing the CIA kernels          */
  double  ms_host_total;  /* host wall time of the whole trx_run call           */
  double  ms_spectrum;    /* intensity/flux or modulation                      */
  int64_t ncandidates;    /* lines that can be a layer's strongest line (the others are
                             dominated, trx_walk.hip.h); -1: every line is looked at */
  int64_t walk_steps;     /* steps of the last run taken by the one-kernel line walk (the
                             rest, sweep_launches - walk_steps, took the two-kernel form)  */
  int64_t walk_records;   /* partial-sum records (64 lane slots each) those steps wrote   */
  int64_t walk_record_lanes; /* lane slots of them actually written and read: sum over steps of
                             records x layers of the step (8 bytes each)                 */
  int64_t walk_layers;    /* layers of the last run swept by walk steps (the rest of layers_swept: two-kernel steps) */
  int64_t sum_bins_walk;  /* the part of sum_bins accumulated by walk steps (counting runs)                          */
  /* the walk steps by kernel form (ABI 4): [0] k_line_walk (lanes = layers, one range per wave),
     [1] k_line_walk_lanes (lanes = lines for the strengths), [2] k_line_walk_packed (several ranges per wave) */
  int64_t walk_form_steps[3];        /* steps of the last run                                                   */
  int64_t walk_form_layers[3];       /* layers they swept                                                       */
  int64_t walk_form_record_lanes[3]; /* their share of walk_record_lanes                                        */
  int64_t walk_form_bins[3];         /* their share of sum_bins_walk (counting runs)                            */
  double  ms_k_walk_form[3];         /* their share of ms_k_walk (profile >= 1)                                 */
  double  ms_walk_span;              /* (ABI 5) first walk's start to last walk's end of the last run: where the walks of a hinted run share
                                        the device on two queues, less than ms_k_walk, the sum of their own durations (profile >= 1) */
} trx_stats;

typedef struct trx_handle trx_handle;

int  trx_abi_version(void);
/* Extinction of layers computed by an earlier run (the reference's --saveext file: restfile_extinct,
 * extinction.c:97-137, called at the head of tau(), tau.c:155-156).  The following trx_run calls take
 * e[layer][.] of every flagged layer from here instead of sweeping it (a step whose layers are all
 * flagged launches no line kernel); flags and values stay until replaced (nlayer = 0: forget them).
 * e: [nlayer][nwn_shard] host memory, this handle's shard; computed: [nlayer]. */
int  trx_restore_extinction(trx_handle *h, int32_t nlayer, const double *e, const uint8_t *computed);
int  trx_device_count(void);
/* HIP version the library was built with / of the runtime it runs on (HIP_VERSION encoding:
 * major*10000000 + minor*100000 + patch); a caller that maps a runtime of its own first can check. */
int  trx_hip_versions(int *built, int *running);

int  trx_create (const trx_static *st, trx_handle **out);
int  trx_run    (trx_handle *h, const trx_atm *atm, const trx_opts *opts,
                 double *spectrum /* [wn_hi-wn_lo], host */, trx_debug *dbg /* may be NULL */);
/* Same as trx_run but leaves the spectrum in device memory (d_spectrum is a
 * device pointer on the handle's device, e.g. a torch tensor for an RCCL
 * gather); work is enqueued on the handle's stream and synchronised on return. */
int  trx_run_device(trx_handle *h, const trx_atm *atm, const trx_opts *opts,
                    void *d_spectrum, trx_debug *dbg);
void trx_destroy(trx_handle *h);

/* Several atmospheres per call -- what a retrieval driver does with the reference by calling run_transit
 * (transit.c:118-122) once per atmosphere.  A batch keeps `ways` handles made from one description (line list
 * and tables `ways` times in device memory) and a host thread for each; trx_run_batch deals the k
 * atmospheres to them and returns when all are done.  spectra[j] ([wn_hi-wn_lo], host) is what
 * trx_run(h, &atm[j], opts, spectra[j], NULL) gives, bit for bit.  On failure: the first error's code, its
 * text through trx_last_error(NULL); spectra of other atmospheres may or may not have been written. */
#define TRX_BATCH_MAX_WAYS 8
typedef struct trx_batch trx_batch;
int  trx_batch_create(const trx_static *st, int32_t ways, trx_batch **out);
int  trx_run_batch(trx_batch *b, int32_t k, const trx_atm *atm /* [k] */, const trx_opts *opts,
                   double *const *spectra /* [k] */);
int  trx_batch_ways(const trx_batch *b);
void trx_batch_destroy(trx_batch *b);

/* The per-layer operator of the reference in its per-molecule form,
 *   computemolext(tr, kiso, temp, density, Z, permol = 1)   (extinction.c:282)
 * batched over nv independent thermodynamic states -- what calcopacity()
 * (opacity.c:387-403) loops over (layer x temperature) to fill an opacity grid.
 * iso_slot[i] = output row of isotope i (isotopes of one molecule share a row and
 * must be contiguous); out is [nv][nslot][wn_hi-wn_lo], extinction per unit
 * density (no density factor, extinction.c:472), thresholded per molecule. */
int  trx_sweep_permol(trx_handle *h, int32_t nv, const double *temp /* [nv] */,
                      const double *density /* [nmol][nv] */, const double *zpart /* [niso][nv] */,
                      double ethresh, int32_t nslot, const int32_t *iso_slot /* [niso] */, double *out);

int  trx_get_stats(const trx_handle *h, trx_stats *out);

/* Voigt table access for parity tests (reference struct opacity.profile /
 * .profsize, structures_tr.h:154-171). */
int  trx_table_info(const trx_handle *h, int64_t *profsize /* [ndop*nlor] */,
                    int64_t *offset /* [ndop*nlor] float offset of each profile */,
                    int64_t *total_floats);
int  trx_table_copy(const trx_handle *h, float *out /* [total_floats] */);
int  trx_width_grids(const trx_handle *h, double *adop /* [ndop] */, double *alor /* [nlor] */);

/* RCCL communicator for the wavenumber-sharded job (one process per GPU).
 * Rank 0 calls trx_comm_unique_id and ships the 128 bytes to the other ranks
 * by any channel (torch.distributed, MPI, a file); every rank then calls
 * trx_comm_create on its own device. */
#define TRX_COMM_ID_BYTES 128
int  trx_comm_unique_id(void *id_out /* TRX_COMM_ID_BYTES */);
int  trx_comm_create(const void *id, int nranks, int rank, int device, void **comm_out);
void trx_comm_destroy(void *comm);
/* Give a communicator up without the collective teardown (ncclCommAbort): for the error path of a
 * job in which some rank failed and the others must not wait for it. */
void trx_comm_abort(void *comm);

/* The one exchange of a wavenumber-sharded job ("one ncclAllGather of the
 * spectrum slices at the end"): every rank hands in `count` doubles in device memory (its slice,
 * padded to the same count on every rank) and receives all ranks' slices in rank order in d_all
 * (nranks * count doubles, device memory).  ncclAllGather on the handle's stream over the
 * communicator given in trx_static.comm; synchronised on return.  A handle without a
 * communicator (single rank) copies its slice.  Nothing else is exchanged between ranks: the
 * per-layer maximum line strength, the only global quantity of the path (extinction.c:399-427),
 * is computed by every rank from the same small set of candidate lines. */
int  trx_gather(trx_handle *h, const void *d_slice, void *d_all, int64_t count);
/* The same with host buffers on both sides (slice: count doubles; all: nranks * count). */
int  trx_gather_host(trx_handle *h, const double *slice, double *all, int64_t count);

const char *trx_strerror(int status);
const char *trx_last_error(const trx_handle *h);   /* detail of the last failure; h NULL: of this thread's last
                                                      trx_comm_create */

/* Messages.  The reference prints through tr_output(level, ...) filtered by the global
 * `verblevel` (transit.h:70-75, levels flags_tr.h:107-111) and exit()s on errors; the
 * library never prints or exits by itself: it hands the text to this process-wide callback
 * for levels <= max_level.  NULL = silent, except that the reason of a failed trx_create --
 * whose handle does not survive to be asked -- then goes to stderr.  Called on the thread
 * that made the API call. */
enum { TRX_LOG_ERROR = 1, TRX_LOG_WARN = 2, TRX_LOG_INFO = 3, TRX_LOG_RESULT = 4, TRX_LOG_DEBUG = 5 };
typedef void (*trx_log_fn)(int level, const char *message, void *user);
void trx_set_log(trx_log_fn fn, void *user, int max_level);

#ifdef __cplusplus
}
#endif
#endif /* TRANSIT_HIP_H */
