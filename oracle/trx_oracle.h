/* trx_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C; one thread but for the OpenMP'd eager layer sweep) of the reference's spectrum path, used
 * as the parity checker by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.  It is never linked into, imported by, or
 * called from the product (transit_amd/): the product fails loudly when its
 * HIP library is missing.
 *
 * The entry points take the very same POD structs as the product's C ABI
 * (include/transit_hip.h) so that a test can hand identical inputs to both.
 *
 * Pinning: checked against the compiled reference (oracle/_ref/transit and
 * oracle/_ref/libpu_ref.so, built by oracle/Makefile from the sources where
 * they lie) through the fixtures in tests/golden/ -- see tests/test_oracle_*.py.
 */
#ifndef TRX_ORACLE_H
#define TRX_ORACLE_H
#include "../include/transit_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct trxo_handle trxo_handle;

int  trxo_create (const trx_static *st, trxo_handle **out);
int  trxo_run    (trxo_handle *h, const trx_atm *atm, const trx_opts *opts,
                  double *spectrum, trx_debug *dbg);
void trxo_destroy(trxo_handle *h);
int  trxo_sweep_permol(trxo_handle *h, int32_t nv, const double *temp, const double *density,
                       const double *zpart, double ethresh, int32_t nslot, const int32_t *iso_slot, double *out);
int  trxo_restore_extinction(trxo_handle *h, int32_t nlayer, const double *e, const unsigned char *computed);
int  trxo_get_stats(const trxo_handle *h, trx_stats *out);
int  trxo_table_info(const trxo_handle *h, int64_t *profsize, int64_t *offset, int64_t *total);
int  trxo_table_copy(const trxo_handle *h, float *out);
int  trxo_width_grids(const trxo_handle *h, double *adop, double *alor);

/* leaf functions exposed for unit checks against oracle/_ref/libpu_ref.so */
int    trxo_voigt_profile(int nwn, double half, double alphaL, double alphaD, float *out, int quick);
double trxo_parab3(const double *x, const double *y, double xr);
double trxo_simpson(const double *x, const double *y, int n);
double trxo_tau_slant(const double *rad, long nlay, double b, double *ex);
double trxo_modulation(const double *tau, long last, double toomuch, const double *ip, long ipn,
                       double ipfct, double srad, int transparent);
int    trxo_nearest(const double *a, double v, int lo, int hi);
void   trxo_spline_init(double *z, const double *x, const double *y, long n);
double trxo_spline_eval(const double *z, long n, const double *x, const double *y, double xo);

#ifdef __cplusplus
}
#endif
#endif
