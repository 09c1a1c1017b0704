/* transit_host.h -- C ABI of the host side of the drop-in: everything the
 * reference does in transit_init() *before* the spectrum path
 * (transit/src/transit.c:25-74: processparameters, acceptgenhints,
 * makewnsample, getatm, readlineinfo, makeradsample, readcs) and the writers
 * after it (printflux eclipse.c:356-380, printmod slantpath.c:511-555,
 * printtoomuch tau.c:612-640), restated in C++ with the reference's CLI/cfg
 * option names and file formats.  It produces the plain structs that
 * include/transit_hip.h consumes; it contains no GPU code and no spectrum
 * arithmetic.
 */
#ifndef TRANSIT_HOST_H
#define TRANSIT_HOST_H
#include "transit_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct trh_problem trh_problem;

/* Parse argv exactly like `transit [options]` (-c file | --config_file file,
 * --name value, cfg lines "name value" with the reference's prefix matching,
 * procopt.c:651-705), read every input file and build the samplings.
 * On failure returns a negative code and, if err != NULL, a message.
 * Returns 1 (and no problem) when --help or --version was served: the text went to stdout and
 * there is nothing to run (the reference exits with success there, argum.c:582-607). */
int  trh_load(int argc, const char *const *argv, trh_problem **out, char *err, int errlen);
void trh_free(trh_problem *p);

/* Notes ("I: ...") and warnings ("W: ...") the host side recorded while loading, one per line:
 * what the reference prints through tr_output(TOUT_WARN / TOUT_INFO, ...) for the same input
 * (abundance sums off by more than allowq, readatm.c:545-549; options that are accepted but
 * cannot change this path's outputs). */
const char *trh_messages(const trh_problem *p);

/* The option table (same names, order and defaults as argum.c:112-320), entry i:
 * kind 'a' acted on, 'n' parsed like the reference and like there without effect on any output,
 * 'w' accepted with a warning (results unaffected), 'x' rejected with TRX_E_UNSUPPORTED.
 * Returns TRX_E_ARG past the end. */
int  trh_option_table(int i, const char **name, int *has_arg, char *kind);

/* Every file a run of this problem writes, one "kind value" line each (what the output options
 * turned into); valid until the next call on this problem. */
const char *trh_output_plan(trh_problem *p);

const trx_static *trh_static(const trh_problem *p);
const trx_atm    *trh_atm   (const trh_problem *p);
const trx_opts   *trh_opts  (const trh_problem *p);

int64_t trh_nwn(const trh_problem *p);                       /* get_no_samples()  transit.c:77 */
void    trh_wavenumbers(const trh_problem *p, double *out);  /* get_waveno_arr()  transit.c:82 */

/* restrict the static description to coarse bins [lo,hi) (one GPU's shard) */
void trh_set_shard(trh_problem *p, int64_t lo, int64_t hi);
/* Shards of a multi-GPU job: bins [bounds[k], bounds[k+1]) for rank k, cut so that every rank
 * gets about the same work (lines to walk + rays to integrate), not the same number of bins. */
int  trh_shard_bounds(const trh_problem *p, int nranks, int64_t *bounds /* [nranks + 1] */);

/* BART-style re-entry (run_transit -> reloadatm, readatm.c:722-784):
 * input = [T(nlayer), q_0(nlayer), ..., q_{nmol-1}(nlayer)] */
int  trh_reload_atm(trh_problem *p, const double *input, int n);
void trh_set_radius(trh_problem *p, double refradius);       /* transit.c:98  */
void trh_set_cloudtop(trh_problem *p, double cloudtop);      /* transit.c:103 */
void trh_set_scattering(trh_problem *p, int flag, double logext); /* transit.c:112 */

/* Opacity-grid mode (--opacityfile, reference opacity.c:9-214).
 * If the file exists it was read by trh_load and trh_static()->ogrid points at it.
 * If it does not, trh_needs_opacity_build() is 1: ask for the (layer x temperature)
 * states with trh_grid_request, run them through trx_sweep_permol (the batched
 * computemolext(permol=1) of calcopacity, opacity.c:387-403) and hand the result to
 * trh_install_opacity, which writes the file (opacity.c:405-421) and switches the
 * problem to grid mode -- the reference, too, goes on with the grid it just made. */
int  trh_needs_opacity_build(const trh_problem *p);
int  trh_grid_request(const trh_problem *p, int32_t *nv, const double **temp, const double **density,
                      const double **zpart, int32_t *nslot, const int32_t **iso_slot);
int  trh_install_opacity(trh_problem *p, const double *o /* [nv][nslot][nwn] */);

/* writers in the reference's formats */
int  trh_write_spectrum(const trh_problem *p, const double *spectrum, const char *path /* NULL = cfg outspec */);
int  trh_write_toomuch(const trh_problem *p, const double *tau, const int64_t *last, const char *path);
/* per-angle intensities (eclipse geometry), printintens eclipse.c:293-350; intens = trx_debug.intens */
int  trh_write_intens(const trh_problem *p, const double *intens, const char *path /* NULL = cfg outintens */);
/* `savefiles yes`: tau.dat, CIA.dat, mol_extion.dat in the reference's dump formats (tau.c:386-515),
 * from the trx_debug arrays (any of them may be NULL); dir NULL = next to the cfg */
int  trh_write_dumps(const trh_problem *p, const double *e, const double *e_cs, const double *tau, const char *dir);
/* the same with the reference's zero rows in mol_extion.dat: layers below the deepest ray
 * (last = trx_debug.last) were never swept by its lazy sweep and are written as zeros */
int  trh_write_dumps_masked(const trh_problem *p, const double *e, const double *e_cs, const double *tau, const int64_t *last,
                            const char *dir);
/* the other three `savefiles` dumps: total_extion.dat, cloud_extion.dat, scatt_extion.dat
 * (tau.c:180-190, 293-297), from trx_debug.e / e_cs / last / er / e_scat / e_cloud of a run
 * whose e covers every needed layer (trx_opts.eager = 1) */
int  trh_write_ext_dumps(const trh_problem *p, const double *e, const double *e_cs, const int64_t *last, const double *er,
                         const double *e_scat, const double *e_cloud, const char *dir);
/* detailtau / detailext / detailcia (detailout, tau.c:526-605): which = 0 tau [wn][height],
 * 1 molecular extinction [layer][wn], 2 CIA extinction [layer][wn] (trx_debug layouts).  Writes
 * the file named in the option; no-op when the option was not given. */
int  trh_wants_detail(const trh_problem *p, int which);
int  trh_write_detail(const trh_problem *p, int which, const double *arr);
/* outsample (makesample.c:744-770): the samplings file; the reference writes it only together
 * with `savefiles yes` (makesample.c:598-599).  path NULL = the outsample option. */
int  trh_write_sample(const trh_problem *p, const char *path);
/* --saveext FILE (savefile_extinct / restfile_extinct, extinction.c:62-137; tau.c:155-156, 340-341):
 * e [nlayer][nwn] and one flag per layer.  read: TRX_OK, or 1 when there is no valid file (a note goes
 * to trh_messages, as the reference warns and continues); hand the arrays to trx_restore_extinction. */
int  trh_saveext_read(trh_problem *p, double *e, uint8_t *computed);
int  trh_saveext_write(const trh_problem *p, const double *e, const uint8_t *computed);
const char *trh_option(const trh_problem *p, const char *name);  /* accepted value of an option */

#ifdef __cplusplus
}
#endif
#endif
