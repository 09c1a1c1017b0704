/* trx_oracle.c -- TEST INFRASTRUCTURE ONLY (see trx_oracle.h).
 *
 * CPU restatement of the reference's spectrum path in its exact discretisation; one thread,
 * except that the eager sweep of all layers (trx_opts.eager) runs its independent layers
 * under OpenMP -- bench.py's all-core CPU figure.  Every function names the reference lines it follows
 * (paths relative to the reference tree).  Written from the algorithm, not
 * from the text: data layout, naming and control flow are this project's.
 *
 * Arithmetic: double everywhere except (as in the reference) the Voigt table,
 * which is float with long-double Region-I series (pu/src/voigt.c:139-181).
 * Built without -ffast-math and without FMA contraction; the reference's own
 * build (with -ffast-math) was measured to agree to its 9-10 printed digits.
 */
#include "trx_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

/* constants: transit/include/constants_tr.h:35-63 (values, cgs) */
#define C_PI      3.141592653589793
#define C_AMU     1.66053886e-24
#define C_EC      4.8032068e-10
#define C_LS      2.99792458e10
#define C_ME      9.1093897e-28
#define C_KB      1.380658e-16
#define C_H       6.6260755e-27
#define C_AMAGAT  2.68678e19
#define C_SIGCTE  (C_PI*C_EC*C_EC/C_LS/C_LS/C_ME/C_AMU)
#define C_EXPCTE  (C_H*C_LS/C_KB)
#define C_SQRTLN2 0.83255461115769775635
#define C_DEG     (C_PI/180.0)
#define C_NAVO    6.02214076e23
#define C_E0H2    4.911e-23
#define C_MICRON  1e-4
#define TLI_WFCT  1e-4   /* readlineinfo.c:6  microns -> cm */
#define TLI_EFCT  1.0    /* readlineinfo.c:7 */

/* ------------------------------------------------------------------------ */
/* small numerics                                                           */
/* ------------------------------------------------------------------------ */

/* pu/src/iomisc.c:1088-1108 -- index of the element nearest to v by the
 * reference's recursive bisection (ties and unsorted corner cases included). */
int trxo_nearest(const double *a, double v, int lo, int hi)
{
  while (hi - lo > 1) {
    int mid = (hi + lo) / 2;
    if (a[mid] > v) hi = mid; else lo = mid;
  }
  if (hi - lo == 1)
    return (fabs(a[hi] - v) < fabs(a[lo] - v)) ? hi : lo;
  return lo;   /* hi == lo: the reference exits; callers never get here */
}

/* pu/src/numerical.c:16-45 (binsearchie): a[i] <= v < a[i+1]; -1 below,
 * -2 above, -5 when v equals the last element. */
static int bsearch_ie(const double *a, long i, long f, double v)
{
  if (a[i] > v) return -1;
  if (a[f] < v) return -2;
  if (a[f] == v) return -5;
  if (i == f && a[i] != v) return -3;
  while (f - i > 1) {
    long m = (f + i) >> 1;
    if (a[m] > v) f = m; else i = m;
  }
  return (int)i;
}

/* pu/src/numerical.c:182-195 (interp_parab): parabola through three nodes taken as
 * equispaced (only node[1]-node[0] is used), in the coordinates t = x/step.  Same operations in
 * the same order as the reference: the rounding noise is part of the result (tests/tolerances.py). */
double trxo_parab3(const double *node, const double *v, double at)
{
  const double step = node[1] - node[0];
  const double t0   = node[0] / step;
  const double bend = v[0] + v[2] - 2*v[1];
  const double quad = bend / (2.0 * step * step);
  const double lin  = (v[2] - v[1] - (t0 + 1.5) * bend) / step;
  const double cst  = v[0] + t0 * (v[2] - 4*v[1] + 3*v[0] + t0 * bend) / 2.0;
  return at * at * quad + at * lin + cst;
}

/* pu/src/numerical.c:202-211 (interp_line) */
static double line2(const double *x, const double *y, double xr)
{
  const double m = (y[1] - y[0]) / (x[1] - x[0]);
  return y[0] + (xr - x[0]) * m;
}

/* pu/src/numerical.c:486-495 (makeh), 390-425 (geth), 454-481 (simps),
 * 500-525 (simpson): Simpson rule on a non-uniform abscissa; with an even
 * number of points the first interval is a trapezoid. */
double trxo_simpson(const double *x, const double *y, int n)
{
  if (n == 1) return 0.0;
  if (n == 2) return (x[1] - x[0]) * (y[0] + y[1]) / 2;
  const int even = (n % 2 == 0);
  double acc = 0.0;
  for (int i = 0; i < (n - 1) / 2; i++) {
    const int j = 2*i + even;
    const double h0 = x[j+1] - x[j], h1 = x[j+2] - x[j+1];
    const double hsum = h0 + h1;
    const double hratio = h1 / h0;
    const double hfactor = hsum * hsum / (h0 * h1);
    acc += (y[j] * (2.0 - hratio) + y[j+1] * hfactor + y[j+2] * (2.0 - 1.0/hratio)) * hsum;
  }
  acc = acc / 6.0;
  if (even) acc += (x[1] - x[0]) * (y[0] + y[1]) / 2;
  return acc;
}

/* pu/src/numerical.c:154-172 (integ_trapz) */
static double trapz(const double *x, const double *y, long n)
{
  double r = 0;
  for (long i = 0; i < n - 1; i++) r += (x[i+1] - x[i]) * (y[i+1] + y[i]);
  return 0.5 * r;
}

/* pu/src/spline.c:12-48 (tri) + 186-206 (spline_init): second derivatives of
 * the natural cubic spline. */
void trxo_spline_init(double *z, const double *x, const double *y, long n)
{
  double *h = calloc(n, sizeof(double)), *b = calloc(n, sizeof(double)),
         *u = calloc(n, sizeof(double)), *v = calloc(n, sizeof(double));
  for (long i = 0; i < n - 1; i++) h[i] = x[i+1] - x[i];
  for (long i = 0; i < n - 1; i++) b[i] = (y[i+1] - y[i]) / h[i];
  if (n > 2) {
    u[1] = 2 * (h[1] + h[0]);
    v[1] = 6 * (b[1] - b[0]);
  }
  for (long i = 2; i < n - 1; i++) {
    u[i] = 2*(h[i] + h[i-1]) - h[i-1]*h[i-1]/u[i-1];
    v[i] = 6*(b[i] - b[i-1]) - v[i-1]*h[i-1]/u[i-1];
  }
  z[0] = z[n-1] = 0;
  for (long i = n - 2; i > 0; i--) z[i] = (v[i] - h[i]*z[i+1]) / u[i];
  free(h); free(b); free(u); free(v);
}

/* pu/src/spline.c:131-183 (splinterp_pt) */
double trxo_spline_eval(const double *z, long n, const double *x, const double *y, double xo)
{
  int k = trxo_nearest(x, xo, 0, (int)n - 1);
  if (k == n - 1 || xo < x[k]) k--;
  const double h = x[k+1] - x[k], dy = y[k+1] - y[k];
  if (x[k] == xo) return y[k];
  if (h > 0) {
    const double dx = xo - x[k];
    const double a = (z[k+1] - z[k]) / (6*h);
    const double b = 0.5 * z[k];
    const double c = dy/h - h/6 * (z[k+1] + 2*z[k]);
    return y[k] + dx*(c + dx*(b + dx*a));
  }
  return 0;
}

/* pu/src/iomisc.c:1064-1083 (logspace) */
static void logspace(double lo, double hi, int n, double *out)
{
  const double l0 = log10(lo), l1 = log10(hi);
  const double step = (l1 - l0) / (n - 1.0);
  for (int i = 0; i < n; i++) out[i] = pow(10, l0 + i*step);
}

/* ------------------------------------------------------------------------ */
/* Voigt profiles                                                           */
/* ------------------------------------------------------------------------ */
#define V_TWOOSQRTPI 1.12837916709551257389
#define V_SQRTLN2PI  0.46971863934982566689
#define V_NCOEF 64
static double v_coef[V_NCOEF];     /* 1/(n!(2n+1)), pu/src/voigt.c:45-108 */
static int v_coef_ready = 0;

static void voigt_coefs(void)
{
  if (v_coef_ready) return;
  long double fact = 1.0L;
  for (int n = 0; n < V_NCOEF; n++) {
    if (n > 0) fact *= (long double)n;
    v_coef[n] = (double)(1.0L / (fact * (long double)(2*n + 1)));
  }
  v_coef_ready = 1;
}

/* pu/src/voigt.c:132-200 (voigtxy): Re w(x+iy) by region, scaled by
 * sqrt(ln2/pi)/alphaD, rounded to float. */
static float voigt_point(double x, double y, double alphaD)
{
  const long double x2y2 = x*x - y*y;
  const long double xy2  = 2*x*y;
  if (x < 3 && y < 1.8) {                       /* Region I: power series */
    const long double cs = cosl(xy2), sn = sinl(xy2);
    const int nterm = (x < 1 ? 15 : (int)(6.842*x + 8.0)) + 1;
    long double pr = y, pi = -x, sr = y, si = -x;
    for (int k = 1; k <= nterm; k++) {
      const long double ti = pr*xy2 + pi*x2y2;
      const long double tr = pr*x2y2 - pi*xy2;
      si += ti * v_coef[k];
      sr += tr * v_coef[k];
      pi = ti; pr = tr;
    }
    return (float)(V_SQRTLN2PI/alphaD * exp((double)(-x2y2)) *
                   (cs*(1 - sr*V_TWOOSQRTPI) - sn*si*V_TWOOSQRTPI));
  }
  const long double q = xy2*xy2, p = xy2*x;
  if (x < 5 && y < 5) {                         /* Region II: 3-term rational */
    const long double t1 = x2y2 - (long double)0.19016350,
                      t2 = x2y2 - (long double)1.78449270,
                      t3 = x2y2 - (long double)5.52534370;
    return (float)(V_SQRTLN2PI/alphaD * (0.46131350  *((p - t1*y)/(t1*t1 + q)) +
                                         0.09999216  *((p - t2*y)/(t2*t2 + q)) +
                                         0.002883894 *((p - t3*y)/(t3*t3 + q))));
  }
  {                                             /* Region III: 2-term rational */
    const long double t1 = x2y2 - (long double)0.27525510,
                      t2 = x2y2 - (long double)2.72474500;
    return (float)(V_SQRTLN2PI/alphaD * (0.51242424*((p - t1*y)/(t1*t1 + q)) +
                                         0.05176536*((p - t2*y)/(t2*t2 + q))));
  }
}

/* pu/src/voigt.c:369-483 (voigtn) with its two averaging helpers :489-554.
 * nwn bins spanning [-half, +half]; three regimes:
 *   quick  : point samples                                  (:456-459)
 *   fine   : bin spacing below alphaD/49 -> mean of the two bin edges
 *   coarse : Simpson mean over an even number of sub-intervals per bin     */
int trxo_voigt_profile(int nwn, double half, double aL, double aD, float *out, int quick)
{
  voigt_coefs();
  const double y = C_SQRTLN2 * aL / aD;
  double step = 2.0 * half / (nwn - 1);
  int npts = 50;
  double sub = aD / (npts - 1);
  if (step < sub || quick) {
    sub = step;
    npts = nwn + 1;
  } else {
    npts = (int)(step / sub) + 1;
    if (npts & 1) npts++;
    npts = nwn * npts + 1;
    sub = 2.0 * half / (npts - 1);
  }
  float *fine = calloc((size_t)npts, sizeof(float));
  if (!fine) return -1;
  for (int i = 0; i < npts; i++)
    fine[i] = voigt_point(C_SQRTLN2 * fabs(sub*i - half) / aD, y, aD);

  if (quick) {
    for (int i = 0; i < nwn; i++) out[i] = fine[i];
  } else {
    const double ratio = (float)(npts - 1) / nwn;
    const int per = (int)ratio + 1;
    if (ratio + 1 != per) { free(fine); return -2; }
    const int m = per - 1;                 /* sub-intervals per bin */
    const float *in = fine;
    if (per & 1) {                         /* meanintegSimp, float accumulation */
      for (int k = 0; k < nwn; k++, in += m) {
        float s = 0;
        for (int i = 1; i < m; i += 2) s += in[i];
        s *= 2;
        for (int i = 2; i < m; i += 2) s += in[i];
        s *= 2;
        s += in[0] + in[m];
        s /= (m * 3.0);
        out[k] = s;
      }
    } else {                               /* meanintegTrap */
      for (int k = 0; k < nwn; k++, in += m) {
        float s = 0;
        for (int i = 1; i < m; i++) s += in[i];
        s = (s + (in[0] + in[m]) / 2.0) / (double)m;
        out[k] = s;
      }
    }
  }
  free(fine);
  return 1;
}

/* ------------------------------------------------------------------------ */
/* handle                                                                   */
/* ------------------------------------------------------------------------ */
struct trxo_handle {
  trx_static st;           /* shallow copy; arrays below are deep copies */
  double *wl, *elow, *gf; int16_t *isoid;
  double *iso_mass, *iso_ratio; int32_t *iso_imol;
  double *mol_mass, *mol_radius, *mol_pol; int32_t *mol_is_h2;
  trx_cia *cia;
  trx_opacity_grid grid;
  /* Voigt table (struct opacity, structures_tr.h:154-171) */
  double *adop, *alor;
  int64_t *psize;          /* [ndop*nlor] half sizes */
  int64_t *poff;           /* [ndop*nlor] offset of each profile in tab */
  float   *tab;
  int64_t  tab_n;
  trx_stats stats;
  /* restfile_extinct (extinction.c:97-137, called at tau.c:155-156): rows and flags of layers an
   * earlier run computed; [nsaved][nsh] and [nsaved], NULL when nothing was restored */
  double *e_saved; unsigned char *c_saved; long nsaved;
};

static void *dupmem(const void *p, size_t n)
{
  if (!p || !n) return NULL;
  void *q = malloc(n);
  if (q) memcpy(q, p, n);
  return q;
}

/* transit/src/extinction.c:8-57 (getprofile): size rule for one profile.
 * dop/lor arrive as float (PREC_VOIGT), extinction.c:11-12. */
static int64_t profile_points(double dwn, float dop, float lor, float ta, int nwave)
{
  double big = dop;
  if (big < lor) big = lor;
  const double wv = big * ta;
  int nv = 2*(long)(wv/dwn + 0.5) + 1;
  if (nv < 2) nv = 3;
  if (nv > 2*nwave) nv = 2*nwave + 1;
  return nv;
}

/* transit/src/opacity.c:219-277 (calcprofiles) */
static int build_table(trxo_handle *h)
{
  const trx_static *s = &h->st;
  const int nd = s->ndop, nl = s->nlor;
  /* one sentinel past the end: the reference searches with hi = nDop / nLor
   * (extinction.c:393-394, 482), i.e. may compare against the element after
   * the array; +inf makes that comparison pick the last real sample. */
  h->adop = malloc(sizeof(double)*(nd+1));
  h->alor = malloc(sizeof(double)*(nl+1));
  logspace((double)s->dmin, (double)s->dmax, nd, h->adop);
  logspace((double)s->lmin, (double)s->lmax, nl, h->alor);
  h->adop[nd] = HUGE_VAL; h->alor[nl] = HUGE_VAL;
  h->psize = calloc((size_t)nd*nl, sizeof(int64_t));
  h->poff  = calloc((size_t)nd*nl, sizeof(int64_t));
  const double dwn = s->wn_d / s->osamp;
  int64_t total = 0;
  for (int i = 0; i < nd; i++)
    for (int j = 0; j < nl; j++) {
      if (h->adop[i]*10.0 < h->alor[j] && i != 0) {      /* opacity.c:262-265 */
        h->psize[i*nl+j] = h->psize[(i-1)*nl+j];
        h->poff [i*nl+j] = h->poff [(i-1)*nl+j];
      } else {
        int64_t nv = profile_points(dwn, (float)h->adop[i], (float)h->alor[j],
                                    s->timesalpha, (int)s->nown);
        h->psize[i*nl+j] = nv/2;
        h->poff [i*nl+j] = total;
        total += nv;
      }
    }
  h->tab = malloc(sizeof(float)*(size_t)total);
  if (!h->tab) return TRX_E_NOMEM;
  h->tab_n = total;
  for (int i = 0; i < nd; i++)
    for (int j = 0; j < nl; j++) {
      if (h->adop[i]*10.0 < h->alor[j] && i != 0) continue;
      const int nv = (int)(2*h->psize[i*nl+j] + 1);
      const float dop = (float)h->adop[i], lor = (float)h->alor[j];
      int rc = trxo_voigt_profile(nv, dwn*(long)(nv/2), lor, dop,
                                  h->tab + h->poff[i*nl+j], nv > 99999);
      if (rc != 1) return TRX_E_ARG;
    }
  h->stats.table_floats = total;
  return TRX_OK;
}

int trxo_create(const trx_static *st, trxo_handle **out)
{
  if (!st || !out || st->abi_version != TRX_ABI_VERSION) return TRX_E_ARG;
  if (st->nwn < 2 || st->osamp < 1 || st->ndop < 2 || st->nlor < 2) return TRX_E_ARG;
  trxo_handle *h = calloc(1, sizeof(*h));
  if (!h) return TRX_E_NOMEM;
  h->st = *st;
  const size_t nl = (size_t)st->nlines;
  h->wl    = dupmem(st->wl_um, nl*8);  h->elow = dupmem(st->elow, nl*8);
  h->gf    = dupmem(st->gf, nl*8);     h->isoid = dupmem(st->isoid, nl*2);
  h->iso_mass = dupmem(st->iso_mass, st->niso*8);
  h->iso_ratio = dupmem(st->iso_ratio, st->niso*8);
  h->iso_imol = dupmem(st->iso_imol, st->niso*4);
  h->mol_mass = dupmem(st->mol_mass, st->nmol*8);
  h->mol_radius = dupmem(st->mol_radius, st->nmol*8);
  h->mol_pol = dupmem(st->mol_pol, st->nmol*8);
  h->mol_is_h2 = dupmem(st->mol_is_h2, st->nmol*4);
  if (st->ncia > 0) {
    h->cia = calloc(st->ncia, sizeof(trx_cia));
    for (int k = 0; k < st->ncia; k++) {
      h->cia[k] = st->cia[k];
      h->cia[k].wn   = dupmem(st->cia[k].wn,   sizeof(double)*st->cia[k].nwave);
      h->cia[k].temp = dupmem(st->cia[k].temp, sizeof(double)*st->cia[k].ntemp);
      h->cia[k].cs   = dupmem(st->cia[k].cs,   sizeof(double)*st->cia[k].nwave*st->cia[k].ntemp);
    }
  }
  h->st.wl_um = h->wl; h->st.elow = h->elow; h->st.gf = h->gf; h->st.isoid = h->isoid;
  h->st.iso_mass = h->iso_mass; h->st.iso_ratio = h->iso_ratio; h->st.iso_imol = h->iso_imol;
  h->st.mol_mass = h->mol_mass; h->st.mol_radius = h->mol_radius; h->st.mol_pol = h->mol_pol;
  h->st.mol_is_h2 = h->mol_is_h2; h->st.cia = h->cia;
  if (st->ogrid) {
    const trx_opacity_grid *g = st->ogrid;
    h->grid = *g;
    h->grid.mol_index = dupmem(g->mol_index, sizeof(int32_t)*g->nmol);
    h->grid.temp = dupmem(g->temp, sizeof(double)*g->ntemp);
    h->grid.o = dupmem(g->o, sizeof(double)*g->nlayer*g->ntemp*g->nmol*g->nwave);
    h->st.ogrid = &h->grid;
  }
  int rc = build_table(h);
  if (rc != TRX_OK) { trxo_destroy(h); return rc; }
  *out = h;
  return TRX_OK;
}

void trxo_destroy(trxo_handle *h)
{
  if (!h) return;
  free(h->wl); free(h->elow); free(h->gf); free(h->isoid);
  free(h->iso_mass); free(h->iso_ratio); free(h->iso_imol);
  free(h->mol_mass); free(h->mol_radius); free(h->mol_pol); free(h->mol_is_h2);
  if (h->cia) {
    for (int k = 0; k < h->st.ncia; k++) {
      free((void*)h->cia[k].wn); free((void*)h->cia[k].temp); free((void*)h->cia[k].cs);
    }
    free(h->cia);
  }
  if (h->st.ogrid) { free((void*)h->grid.mol_index); free((void*)h->grid.temp); free((void*)h->grid.o); }
  free(h->adop); free(h->alor); free(h->psize); free(h->poff); free(h->tab);
  free(h->e_saved); free(h->c_saved);
  free(h);
}

/* transit/src/extinction.c:97-137 + tau.c:155-156: e[][] and comp[] as the save file holds them are what
 * tau() starts from; a layer with comp set is never swept (tau.c:158, 246). */
int trxo_restore_extinction(trxo_handle *h, int32_t nlayer, const double *e, const unsigned char *computed)
{
  if (!h || nlayer < 0 || (nlayer > 0 && (!e || !computed))) return TRX_E_ARG;
  free(h->e_saved); free(h->c_saved); h->e_saved = NULL; h->c_saved = NULL; h->nsaved = 0;
  if (nlayer == 0) return TRX_OK;
  const trx_static *s = &h->st;
  const long nsh = (s->wn_hi > s->wn_lo) ? (long)(s->wn_hi - s->wn_lo) : (long)s->nwn;
  h->e_saved = dupmem(e, sizeof(double) * (size_t)nlayer * (size_t)nsh);
  h->c_saved = dupmem(computed, (size_t)nlayer);
  if (!h->e_saved || !h->c_saved) return TRX_E_NOMEM;
  h->nsaved = nlayer;
  return TRX_OK;
}

int trxo_get_stats(const trxo_handle *h, trx_stats *out)
{ if (!h || !out) return TRX_E_ARG; *out = h->stats; return TRX_OK; }

int trxo_table_info(const trxo_handle *h, int64_t *ps, int64_t *off, int64_t *total)
{
  if (!h) return TRX_E_ARG;
  const size_t n = (size_t)h->st.ndop * h->st.nlor;
  if (ps)  memcpy(ps,  h->psize, n*sizeof(int64_t));
  if (off) memcpy(off, h->poff,  n*sizeof(int64_t));
  if (total) *total = h->tab_n;
  return TRX_OK;
}
int trxo_table_copy(const trxo_handle *h, float *out)
{ if (!h || !out) return TRX_E_ARG; memcpy(out, h->tab, sizeof(float)*(size_t)h->tab_n); return TRX_OK; }
int trxo_width_grids(const trxo_handle *h, double *adop, double *alor)
{
  if (!h) return TRX_E_ARG;
  if (adop) memcpy(adop, h->adop, sizeof(double)*h->st.ndop);
  if (alor) memcpy(alor, h->alor, sizeof(double)*h->st.nlor);
  return TRX_OK;
}

/* ------------------------------------------------------------------------ */
/* line sweep for one layer                                                 */
/* ------------------------------------------------------------------------ */

/* transit/src/extinction.c:282-529 (computemolext, permol = 0).
 * kout[0..nwn) receives the molecular extinction of the layer. */
static void layer_extinction(trxo_handle *h, double ethresh, double *kout, double temp,
                             const double *dens /* [nmol] */, const double *zp /* [niso] */,
                             int nslot, const int32_t *iso_slot /* NULL: collapsed (permol = 0) */)
{
  const trx_static *s = &h->st;
  const int niso = s->niso, nmol = s->nmol, nl = s->nlor, ofac = s->osamp;
  const int64_t nwn = s->nwn, nown = s->nown, nlines = s->nlines;
  const double dwn = s->wn_d / 1, odwn = s->wn_d / s->osamp;
  const double wn0 = s->wn_i;
  const double own_last = wn0 + (double)(nown - 1) * odwn;
#define OWN(k) (wn0 + (double)(k) * odwn)

  double *alphal = calloc(niso, sizeof(double)), *alphad = calloc(niso, sizeof(double));
  int *idop = calloc(niso, sizeof(int)), *ilor = calloc(niso, sizeof(int));
  const int permol = iso_slot != NULL;
  for (int64_t j = 0; j < nwn * (permol ? nslot : 1); j++) kout[j] = 0.0;

  /* widths: extinction.c:364-395 */
  const double fdoppler = sqrt(2*C_KB*temp/C_AMU) * C_SQRTLN2 / C_LS;
  const double florentz = sqrt(2*C_KB*temp/C_PI/C_AMU) / (C_AMU*C_LS);
  for (int i = 0; i < niso; i++) {
    alphal[i] = 0.0;
    for (int j = 0; j < nmol; j++) {
      const double csd = s->mol_radius[j] + s->mol_radius[s->iso_imol[i]];
      alphal[i] += dens[j]/s->mol_mass[j] * csd * csd *
                   sqrt(1/s->iso_mass[i] + 1/s->mol_mass[j]);
    }
    alphal[i] *= florentz;
    alphad[i] = fdoppler / sqrt(s->iso_mass[i]);
    idop[i] = trxo_nearest(h->adop, alphad[i]*wn0, 0, s->ndop);
    ilor[i] = trxo_nearest(h->alor, alphal[i],     0, s->nlor);
  }

  /* pass 1: strongest line per output slot, extinction.c:399-427 (one bucket when permol=0) */
  double *kmaxv = calloc(permol ? nslot : 1, sizeof(double));
  int64_t ninr = 0;
  for (int64_t ln = 0; ln < nlines; ln++) {
    const double wavn = 1.0 / (s->wl_um[ln] * TLI_WFCT);
    const int i = s->isoid[ln];
    if (wavn < wn0 || wavn > own_last) continue;
    ninr++;
    const double pk = s->iso_ratio[i] * C_SIGCTE * s->gf[ln] *
                      exp(-C_EXPCTE*TLI_EFCT*s->elow[ln]/temp) *
                      (1 - exp(-C_EXPCTE*wavn/temp)) / s->iso_mass[i] / zp[i];
    const int mslot = permol ? iso_slot[i] : 0;
    if (kmaxv[mslot] == 0) kmaxv[mslot] = pk; else kmaxv[mslot] = fmax(kmaxv[mslot], pk);
  }
  h->stats.nlines_inrange = ninr;

  /* pass 2: extinction.c:429-511 */
  int64_t nadd = 0, nskip = 0, neval = 0, nbins = 0;
  for (int64_t ln = 0; ln < nlines; ln++) {
    const double wavn = 1.0 / (s->wl_um[ln] * TLI_WFCT);
    const int i = s->isoid[ln];
    if (wavn < wn0 || wavn > own_last) continue;

    double pk = s->gf[ln] * exp(-C_EXPCTE*TLI_EFCT*s->elow[ln]/temp) *
                (1 - exp(-C_EXPCTE*wavn/temp));
    int iown = (int)((wavn - wn0) / odwn);
    if (fabs(wavn - OWN(iown+1)) < fabs(wavn - OWN(iown))) iown++;

    /* greedy co-adding of following lines of the same isotope that fall within
     * one fine bin of the anchor: extinction.c:449-462 */
    while (ln != nlines - 1 && s->isoid[ln+1] == i) {
      const double nxt = 1.0 / (s->wl_um[ln+1] * TLI_WFCT);
      if (fabs(nxt - OWN(iown)) < odwn) {
        nadd++; ln++;
        pk += s->gf[ln] * exp(-C_EXPCTE*TLI_EFCT*s->elow[ln]/temp) *
              (1 - exp(-C_EXPCTE*nxt/temp));
      } else break;
    }
    pk *= C_SIGCTE * s->iso_ratio[i] / (s->iso_mass[i] * zp[i]);
    const int mslot = permol ? iso_slot[i] : 0;
    if (pk < ethresh * kmaxv[mslot]) { nskip++; continue; }
    if (!permol) pk *= dens[s->iso_imol[i]];
    double *krow = kout + (size_t)mslot * nwn;

    const int idwn = (int)((wavn - wn0) / dwn);
    if (alphad[i]*wavn/alphal[i] >= 1e-1)            /* sticky per isotope :480-483 */
      idop[i] = trxo_nearest(h->adop, alphad[i]*wavn, 0, s->ndop);

    const int64_t ps = h->psize[idop[i]*nl + ilor[i]];
    const float *prof = h->tab + h->poff[idop[i]*nl + ilor[i]];
    const int64_t subw = iown - (int64_t)idwn*ofac;
    const long offset = iown - ps;
    long minj = idwn - (ps - subw) / ofac;
    long maxj = idwn + (ps + subw) / ofac;
    if (minj < 0) minj = 0;
    if (maxj >= nwn) maxj = nwn - 1;
    int bj = (int)(ofac*minj - offset);
    for (long j = minj; j <= maxj; ++j) {
      if (bj > 2*ps) break;
      if (bj >= 0) { krow[j] += pk * prof[bj]; nbins++; }
      bj += ofac;
    }
    neval++;
  }
  h->stats.nadd = nadd;
  /* (layers may be swept by several threads: trxo_run with opts.eager, see there) */
#pragma omp atomic
  h->stats.nskip += nskip;
#pragma omp atomic
  h->stats.neval += neval;
#pragma omp atomic
  h->stats.sum_bins += nbins;
  free(alphal); free(alphad); free(idop); free(ilor); free(kmaxv);
#undef OWN
}

/* ------------------------------------------------------------------------ */
/* CIA                                                                      */
/* ------------------------------------------------------------------------ */

/* transit/src/crosssec.c:354-428 (bicubicinterpolate): natural splines, first
 * along temperature for every table row, then along wavenumber; no
 * extrapolation (points outside the table stay 0). res is [nt1][nt2]. */
static void spline2d(double *res, const double *src, const double *x1, long nx1,
                     const double *x2, long nx2, const double *t1, long nt1,
                     const double *t2, long nt2)
{
  memset(res, 0, sizeof(double)*nt1*nt2);
  const double fx1 = x1[0], fx2 = x2[0], lx1 = x1[nx1-1], lx2 = x2[nx2-1];
  if (t1[0] > lx1 || t1[nt1-1] < fx1 || t2[0] > lx2 || t2[nt2-1] < fx2) return;
  long fi = 0, li = nt1, fj = 0, lj = nt2;
  while (t1[fi++] < fx1); fi--;
  for (long i = 0; i < li; i++) if (t1[i] > lx1) li = i;
  while (t2[fj++] < fx2); fj--;
  for (long j = 0; j < lj; j++) if (t2[j] > lx2) lj = j;

  double *z1 = calloc(nx2, sizeof(double)), *z2 = calloc(nx1, sizeof(double));
  double *mid = malloc(sizeof(double)*nt2*nx1);          /* [nt2][nx1] */
  for (long i = 0; i < nx1; i++) {
    trxo_spline_init(z1, x2, src + i*nx2, nx2);
    for (long j = fj; j < lj; j++)
      mid[j*nx1 + i] = trxo_spline_eval(z1, nx2, x2, src + i*nx2, t2[j]);
  }
  for (long j = fj; j < lj; j++) {
    trxo_spline_init(z2, x1, mid + j*nx1, nx1);
    for (long i = fi; i < li; i++)
      res[i*nt2 + j] += trxo_spline_eval(z2, nx1, x1, mid + j*nx1, t1[i]);
  }
  free(z1); free(z2); free(mid);
}

/* transit/src/crosssec.c:272-344 (interpcs): ecs is [nwn][nlayer] */
static int cia_extinction(const trxo_handle *h, const trx_atm *a, const trx_opts *o, double *ecs)
{
  const trx_static *s = &h->st;
  const long nwn = s->nwn, nr = a->nlayer;
  memset(ecs, 0, sizeof(double)*nwn*nr);
  if (s->ncia == 0) return TRX_OK;
  double tmin = 0.0, tmax = 70000.0;                       /* crosssec.c:44-45,175-176 */
  for (int n = 0; n < s->ncia; n++) {
    tmin = fmax(tmin, s->cia[n].temp[0]);
    tmax = fmin(tmax, s->cia[n].temp[s->cia[n].ntemp-1]);
  }
  for (long i = 0; i < nr; i++)
    if (a->temp[i] < tmin || a->temp[i] > tmax) return TRX_E_RANGE;
  double *w = malloc(sizeof(double)*nwn), *e = malloc(sizeof(double)*nwn*nr);
  for (long i = 0; i < nwn; i++) w[i] = o->wn_fct * (s->wn_i + (double)i * s->wn_d);
  for (int n = 0; n < s->ncia; n++) {
    const trx_cia *c = &s->cia[n];
    spline2d(e, c->cs, c->wn, c->nwave, c->temp, c->ntemp, w, nwn, a->temp, nr);
    for (long i = 0; i < nr; i++) {
      double dens = 1.0;
      for (int k = 0; k < c->nspec; k++) {
        const int m = c->mol[k];
        dens *= a->density[m*nr + i] / (C_AMU * s->mol_mass[m] * C_AMAGAT);
      }
      for (long j = 0; j < nwn; j++)
        if (e[j*nr + i] > 0) ecs[j*nr + i] += e[j*nr + i] * dens;
    }
  }
  free(w); free(e);
  return TRX_OK;
}

/* ------------------------------------------------------------------------ */
/* scattering and clouds                                                    */
/* ------------------------------------------------------------------------ */


/* transit/src/extinction.c:587-624 (computeextscat).  press/temp are the
 * reference's raw tr->atm.p / tr->atm.t (tau.c:113-114, 226). */
static void scat_extinction(double *e, long n, const trxo_handle *h, const trx_atm *a,
                            const trx_opts *o, double wn)
{
  const trx_static *s = &h->st;
  switch (o->scat_flag) {
  case 1:
    for (long i = 0; i < n; i++)
      e[i] = pow(10.0, o->scat_logext) * C_E0H2 * a->press[i] / a->temp[i] * pow(wn, 4);
    break;
  case 2:
    memset(e, 0, n*sizeof(double));
    for (long i = 0; i < n; i++)
      for (int j = 0; j < s->nmol; j++)
        e[i] += C_PI * 8e-32 / 3. * pow(s->mol_pol[j], 2) * pow(2. * C_PI * wn * C_MICRON, 4) *
                a->density[j*n + i] / s->mol_mass[j] * C_NAVO;
    break;
  default:
    memset(e, 0, n*sizeof(double));
    break;
  }
}

/* transit/src/extinction.c:630-693 (computeextcloud) */
static void cloud_extinction(double *e, long n, const trx_atm *a, const trx_opts *o,
                             const double *mdens, const double *nH, double wn)
{
  const double ctop = pow(10, o->cloud_top), cbot = pow(10, o->cloud_bot);
  const double ext = o->cloud_ext, gamma = o->cloud_gamma;
  const double x = 2 * C_PI * o->cloud_r * wn;
  const double refwn = pow(o->cloud_refwn, gamma);
  const double kBP = ext * pow(wn, gamma);
  const double kFH = ext / (o->cloud_Q * pow(x, -1 * gamma) + pow(x, 0.2));
  long i;
  if (!ext) { memset(e, 0, n*sizeof(double)); return; }
  for (i = n - 1; i >= 0; i--) {
    if (a->press[i] >= ctop) break;
    e[i] = 0.0;
  }
  for (; i >= 0; i--) {
    if (a->press[i] >= cbot) break;
    switch (o->cloud_flag) {
    case 1: e[i] = ext; break;
    case 2: e[i] = ext * mdens[i]; break;
    case 3: e[i] = kBP * mdens[i]; break;
    case 4: e[i] = kFH * mdens[i]; break;
    case 5: e[i] = nH[i] * kBP * o->cloud_sig / refwn * mdens[i]; break;
    }
  }
  for (; i >= 0; i--) e[i] = 0.0;
}

/* ------------------------------------------------------------------------ */
/* optical depth along one ray                                              */
/* ------------------------------------------------------------------------ */

/* transit/src/eclipse.c:29-105 (eclipsetau).  ex is modified in place exactly
 * as the reference does: ex[rs] is replaced by the parabola value and NOT
 * restored when three or more points are available (:65-66 vs :75-76). */
static double tau_vertical(const double *radv, long nlay, double height, double *exv)
{
  const int rs = trxo_nearest(radv, height, 0, (int)nlay - 1);
  if (rs == nlay - 1) return 0.0;
  const double *rad = radv + rs;
  double *ex = exv + rs;
  int n = (int)(nlay - rs);
  double x3[3], r3[3];
  const double keep_ex = ex[0];
  if (n == 2) ex[0] = trxo_parab3(rad - 1, ex - 1, rad[0]);
  else        ex[0] = trxo_parab3(rad,     ex,     rad[0]);
  const double *yy = ex;
  if (n == 2) {
    x3[0] = ex[0]; x3[2] = ex[1]; x3[1] = (ex[1] + ex[0]) / 2.0;
    r3[0] = rad[0]; r3[2] = rad[1]; r3[1] = (rad[0] + rad[1]) / 2.0;
    ex[0] = keep_ex;
    rad = r3; yy = x3; n = 3;
  }
  double *s = malloc(sizeof(double)*n);
  s[0] = 0.0;
  for (int i = 1; i < n; i++) s[i] = s[i-1] + (rad[i] - rad[i-1]);
  const double res = trxo_simpson(s, yy, n);
  free(s);
  return res;
}

/* transit/src/slantpath.c:19-108 (totaltau1, refraction index 1) */
static double tau_slant(const double *radv, long nlay, double b, double *exv)
{
  const double r0 = b / 1.0;
  const int rs = bsearch_ie(radv, 0, nlay - 1, r0);
  if (rs == -5 || rs == -2) return 0;
  if (rs < 0) return NAN;
  double *rad = (double *)radv + rs;      /* temporarily edited and restored */
  double *ex = exv + rs;
  int n = (int)(nlay - rs);
  double x3[3], r3[3];
  const double keep_ex = ex[0], keep_rad = rad[0];
  if (n == 2) ex[0] = trxo_parab3(rad - 1, ex - 1, r0);
  else        ex[0] = trxo_parab3(rad,     ex,     r0);
  rad[0] = r0;
  const double *rr = rad, *yy = ex;
  if (n == 2) {
    x3[0] = ex[0]; x3[2] = ex[1]; x3[1] = (ex[0] + ex[1]) / 2.0;
    r3[0] = rad[0]; r3[2] = rad[1]; r3[1] = (rad[0] + rad[1]) / 2.0;
    rad[0] = keep_rad; ex[0] = keep_ex;
    rr = r3; yy = x3; n = 3;
  }
  double *s = malloc(sizeof(double)*n);
  s[0] = 0;
  for (int i = 1; i < n; i++) s[i] = sqrt(rr[i]*rr[i] - r0*r0);
  const double res = trxo_simpson(s, yy, n);
  free(s);
  ex[0] = keep_ex; rad[0] = keep_rad;
  return 2 * res;
}

/* ------------------------------------------------------------------------ */
/* spectrum from the optical depth                                          */
/* ------------------------------------------------------------------------ */

/* transit/src/eclipse.c:118-160 (eclipse_intens) */
static double intensity(const double *tau, double w, long last, double angle_deg,
                        const double *temp, long nlay)
{
  const double ang = angle_deg * C_DEG;
  double *B = malloc(sizeof(double)*(last+1)), *dt = malloc(sizeof(double)*(last+1));
  for (long i = 0; i <= last; i++) {
    dt[i] = exp(-tau[i] / cos(ang));
    B[i]  = (2.0 * C_H * pow(w, 3.0) * C_LS * C_LS) /
            (exp(C_H * w * C_LS / (C_KB * temp[nlay-1-i])) - 1.0);
  }
  const double r = B[last]*dt[last] - trapz(dt, B, last + 1);
  free(B); free(dt);
  return r;
}

/* transit/src/slantpath.c:351-436 (modulation1) */
static double modulation_int(const double *tau, long last, double toomuch, const double *ip,
                             long ipn, double ipfct, double srad, int transparent)
{
  const long ipn1 = ipn - 1;
  const double maxtau = tau[last] > toomuch ? tau[last] : toomuch;
  double *rint = calloc(ipn, sizeof(double)), *ipv = calloc(ipn, sizeof(double));
  long i;
  for (i = 0; i <= last; i++) {
    ipv [ipn1-i] = ip[i] * ipfct;
    rint[ipn1-i] = exp(-tau[i]) * ipv[ipn1-i];
  }
  last += 1;
  if (last > ipn1) last = ipn1;
  for (; i <= last; i++) {
    ipv [ipn1-i] = ip[i] * ipfct;
    rint[ipn1-i] = 0;
  }
  last++;
  if (last < 3) { free(rint); free(ipv); return NAN; }
  double res = trxo_simpson(ipv + ipn - last, rint + ipn - last, (int)last);
  res = ipv[ipn1]*ipv[ipn1] - 2.0*res;
  if (transparent) res -= exp(-maxtau) * ipv[ipn-last] * ipv[ipn-last];
  res *= 1.0 / (srad*srad);
  free(rint); free(ipv);
  return res;
}

/* The ray solutions on their own, for the analytic known-answer tests the reference keeps in
 * transit/test/test_slantpath.c (:177-198 optical depth through a sphere, :231-307 modulation
 * for prescribed optical depths): same arguments as totaltau1 / modulation1. */
double trxo_tau_slant(const double *rad, long nlay, double b, double *ex) { return tau_slant(rad, nlay, b, ex); }
double trxo_modulation(const double *tau, long last, double toomuch, const double *ip, long ipn,
                       double ipfct, double srad, int transparent)
{ return modulation_int(tau, last, toomuch, ip, ipn, ipfct, srad, transparent); }

/* transit/src/slantpath.c:447-473 (modulationm1) */
static double modulation_rad(const double *tau, long last, double toomuch, const double *ip,
                             double ipfct, double srad)
{
  double ipv[2];
  if (tau[last] < toomuch) return -1;
  long ini = ++last - 2;
  if (ini < 0) ini = 0;
  for (long i = ini; i < last; i++) ipv[i-ini] = ip[i] * ipfct;
  const double r = line2(tau + ini, ipv, toomuch);
  return r * r / (srad*srad);
}

/* ------------------------------------------------------------------------ */
/* opacity grid                                                             */
/* ------------------------------------------------------------------------ */

/* transit/src/opacity.c:387-403 (the loop body of calcopacity): the per-layer
 * operator in its per-molecule form, for nv independent states. */
int trxo_sweep_permol(trxo_handle *h, int32_t nv, const double *temp, const double *density,
                      const double *zpart, double ethresh, int32_t nslot, const int32_t *iso_slot,
                      double *out)
{
  if (!h || nv < 1 || !temp || !density || !zpart || nslot < 1 || !iso_slot || !out) return TRX_E_ARG;
  const trx_static *s = &h->st;
  double *dens = malloc(sizeof(double)*s->nmol), *zp = malloc(sizeof(double)*(s->niso > 0 ? s->niso : 1));
  h->stats.neval = h->stats.nskip = h->stats.sum_bins = 0;
  for (int v = 0; v < nv; v++) {
    for (int m = 0; m < s->nmol; m++) dens[m] = density[(size_t)m*nv + v];
    for (int i = 0; i < s->niso; i++) zp[i]   = zpart[(size_t)i*nv + v];
    layer_extinction(h, ethresh, out + (size_t)v*nslot*s->nwn, temp[v], dens, zp, nslot, iso_slot);
  }
  free(dens); free(zp);
  return TRX_OK;
}

/* transit/src/extinction.c:535-581 (interpolmolext): linear in temperature,
 * times the density of each molecule. */
static void grid_extinction(const trxo_handle *h, const trx_atm *a, long r, double *kout)
{
  const trx_opacity_grid *g = h->st.ogrid;
  const long nwn = h->st.nwn, nr = a->nlayer, nt = g->ntemp, nm = g->nmol;
  const double temp = a->temp[r];
  double *gt = malloc(sizeof(double)*(nt + 1));
  memcpy(gt, g->temp, sizeof(double)*nt); gt[nt] = HUGE_VAL;     /* searched with hi = Ntemp (:562) */
  int it = trxo_nearest(gt, temp, 0, (int)nt);
  if (temp < gt[it]) it--;
  for (long i = 0; i < nwn; i++) {
    kout[i] = 0.0;
    for (long m = 0; m < nm; m++) {
      const double *olo = g->o + (((size_t)r*nt + it    )*nm + m)*nwn;
      const double *ohi = g->o + (((size_t)r*nt + it + 1)*nm + m)*nwn;
      const double ext = (olo[i] * (gt[it+1] - temp) + ohi[i] * (temp - gt[it])) / (gt[it+1] - gt[it]);
      kout[i] += a->density[(size_t)g->mol_index[m]*nr + r] * ext;
    }
  }
  free(gt);
}

/* ------------------------------------------------------------------------ */
/* one spectrum                                                             */
/* ------------------------------------------------------------------------ */

/* transit/src/transit.c:125-214 (do_transit): interpcs, extwn, tau,
 * emergent_intens x angles + flux | modulation.
 * tau() itself: transit/src/tau.c:60-356 including the lazy per-layer sweep. */
int trxo_run(trxo_handle *h, const trx_atm *a, const trx_opts *o, double *spectrum,
             trx_debug *dbg)
{
  if (!h || !a || !o || !spectrum) return TRX_E_ARG;
  const trx_static *s = &h->st;
  const long nwn = s->nwn, nr = a->nlayer;
  const int niso = s->niso, nmol = s->nmol;
  if (nr < 1 || nwn < 2) return TRX_E_ARG;
  /* a wavenumber shard [w0, w1) of the grid (trx_static.wn_lo / wn_hi, one rank of a sharded job):
   * the extinction of a layer is computed on the whole grid as always, the rays -- and with them
   * the lazy sweep's depth -- are the shard's own, and every output holds the shard's w1 - w0 bins */
  const long w0 = (s->wn_hi > s->wn_lo) ? (long)s->wn_lo : 0, w1 = (s->wn_hi > s->wn_lo) ? (long)s->wn_hi : nwn;
  if (w0 < 0 || w1 > nwn) return TRX_E_ARG;
  const long nsh = w1 - w0;
  if (o->solution != TRX_SOL_ECLIPSE && o->solution != TRX_SOL_TRANSIT) return TRX_E_ARG;
  h->stats.neval = h->stats.nskip = h->stats.sum_bins = 0;

  int rc = TRX_OK;
  double *ecs = malloc(sizeof(double)*nwn*nr);            /* [wn][layer] */
  double *e   = calloc((size_t)nwn*nr, sizeof(double));   /* [layer][wn] */
  double *tau = calloc((size_t)nwn*nr, sizeof(double));   /* [wn][height] */
  long   *last = calloc(nwn, sizeof(long));
  unsigned char *comp = calloc(nr, 1);
  double *dens = malloc(sizeof(double)*nmol), *zp = malloc(sizeof(double)*niso);
  double *er = malloc(sizeof(double)*nr), *es = malloc(sizeof(double)*nr),
         *ec = malloc(sizeof(double)*nr), *hh = malloc(sizeof(double)*nr),
         *mdens = calloc(nr, sizeof(double)), *nH = calloc(nr, sizeof(double));
  double *rad = malloc(sizeof(double)*nr);
  memcpy(rad, a->radius, sizeof(double)*nr);

  if (s->ogrid) {
    const trx_opacity_grid *g = s->ogrid;
    if (g->nwave != nwn || g->nlayer != nr || g->ntemp < 2) { rc = TRX_E_ARG; goto done; }
    for (long i = 0; i < nr; i++)
      if (a->temp[i] < g->temp[0] || !(a->temp[i] < g->temp[g->ntemp-1])) { rc = TRX_E_RANGE; goto done; }
  }
  if ((rc = cia_extinction(h, a, o, ecs)) != TRX_OK) goto done;

  /* heights (eclipse) / impact parameters (transit) from the top down:
   * tau.c:92-104 and makesample.c:564-574 (ips = reversed radii) */
  for (long i = 0; i < nr; i++) hh[i] = rad[nr-1-i];
  const double rfct = a->rad_fct, hfct = a->rad_fct;
  /* The impact parameter handed to the ray solution, tau.c:274 `h[ri]*hfct/rfct`, as the
   * reference is BUILT (its Makefile's -ffast-math lets gcc hoist the reciprocal out of the
   * loop: the object code multiplies h*hfct by a stored 1.0/rfct; checked in the disassembly
   * of oracle/_ref/obj/tr_tau.o).  The result is the layer radius or an ulp above it; a true
   * division instead puts it an ulp BELOW for some hydrostatic radii, and the slant-path bracket
   * search (slantpath.c:36) then starts one layer lower -- a 1e-3 effect the reference does not
   * have (tests/test_reentry.py). */
  const double rfct_recip = 1.0 / rfct;

#define SWEEP(L) do { \
    for (int m_ = 0; m_ < nmol; m_++) dens[m_] = a->density[m_*nr + (L)]; \
    for (int i_ = 0; i_ < niso; i_++) zp[i_]   = a->zpart[i_*nr + (L)];   \
    if (s->ogrid) grid_extinction(h, a, (L), e + (size_t)(L)*nwn);            \
    else layer_extinction(h, o->ethresh, e + (size_t)(L)*nwn, a->temp[(L)], dens, zp, 1, NULL); \
    comp[(L)] = 1; } while (0)

  if (h->e_saved) {                                        /* restfile_extinct, tau.c:155-156 */
    if (h->nsaved != nr) { rc = TRX_E_ARG; goto done; }
    for (long L = 0; L < nr; L++)
      if (h->c_saved[L]) { memcpy(e + (size_t)L*nwn + w0, h->e_saved + (size_t)L*nsh, sizeof(double)*nsh); comp[L] = 1; }
  }
  if (!comp[nr-1]) SWEEP(nr-1);                            /* tau.c:158-177 */
  /* eager: every layer, independent of each other -- the one place where this restatement uses
   * more than one core (OMP_NUM_THREADS; bench.py's all-core CPU figure).  Each thread has its
   * own density / partition-function scratch; a layer's sums are what one thread computes. */
  if (o->eager) {
#pragma omp parallel
    {
      double *dens_t = malloc(sizeof(double) * (nmol > 0 ? nmol : 1)), *zp_t = malloc(sizeof(double) * (niso > 0 ? niso : 1));
#pragma omp for schedule(dynamic, 1)
      for (long L = nr - 2; L >= 0; L--) {
        for (int m_ = 0; m_ < nmol; m_++) dens_t[m_] = a->density[m_*nr + L];
        for (int i_ = 0; i_ < niso; i_++) zp_t[i_] = a->zpart[i_*nr + L];
        if (s->ogrid) grid_extinction(h, a, L, e + (size_t)L*nwn);
        else layer_extinction(h, o->ethresh, e + (size_t)L*nwn, a->temp[L], dens_t, zp_t, 1, NULL);
        comp[L] = 1;
      }
      free(dens_t); free(zp_t);
    }
  }

  /* mean mass density and H2 number density for the cloud models,
   * tau.c:193-214.  (The reference accumulates mean_dens into an
   * uninitialised VLA; here it starts from 0.) */
  if (a->abund)
    for (long i = 0; i < nr; i++) {
      double mm = 0;
      for (int j = 0; j < nmol; j++) {
        mdens[i] += a->density[j*nr+i] / s->mol_mass[j] * a->abund[j*nr+i];
        if (s->mol_is_h2 && s->mol_is_h2[j])
          nH[i] = a->density[j*nr+i] / s->mol_mass[j] * a->abund[j*nr+i] * C_NAVO;
        mm += s->mol_mass[j] * a->abund[j*nr+i];
      }
      mdens[i] *= mm;
    }

  long lastr = nr - 1;
  for (long wi = w0; wi < w1; wi++) {
    double *tw = tau + (size_t)wi*nr;
    const double wcgs = (s->wn_i + (double)wi * s->wn_d) * o->wn_fct;
    scat_extinction(es, nr, h, a, o, wcgs);
    if (o->cloud_flag) cloud_extinction(ec, nr, a, o, mdens, nH, wcgs);
    else memset(ec, 0, sizeof(double)*nr);
    for (long ri = 0; ri < nr; ri++)
      er[ri] = e[(size_t)ri*nwn + wi] + es[ri] + ec[ri] + ecs[(size_t)wi*nr + ri];

    long ri;
    for (ri = 0; ri < nr; ri++) {
      if (hh[ri]*hfct < rad[lastr]*rfct) {                 /* tau.c:238-271 */
        do {
          if (!comp[--lastr]) {
            SWEEP(lastr);
            er[lastr] = e[(size_t)lastr*nwn + wi] + es[lastr] + ec[lastr] +
                        ecs[(size_t)wi*nr + lastr];
          }
        } while (hh[ri]*hfct < rad[lastr]*rfct);
      }
      const double bb = (hh[ri]*hfct) * rfct_recip;
      const double t = (o->solution == TRX_SOL_ECLIPSE) ? tau_vertical(rad, nr, bb, er)
                                                        : tau_slant(rad, nr, bb, er);
      tw[ri] = rfct * t;
      if (tw[ri] > o->toomuch) { last[wi] = ri; break; }   /* tau.c:277-287 */
    }
    if (ri == nr) last[wi] = ri - 1;                       /* tau.c:299-304 */
    /* the reference's total / cloud / scattering dumps: the per-wavenumber arrays as they stand
     * after the height loop (save1Darray, tau.c:293-297) */
    if (dbg && dbg->er)      for (long r = 0; r < nr; r++) dbg->er[(size_t)r*nsh + (wi - w0)] = er[r];
    if (dbg && dbg->e_scat)  for (long r = 0; r < nr; r++) dbg->e_scat[(size_t)r*nsh + (wi - w0)] = es[r];
    if (dbg && dbg->e_cloud) for (long r = 0; r < nr; r++) dbg->e_cloud[(size_t)r*nsh + (wi - w0)] = ec[r];
  }
#undef SWEEP

  /* spectrum */
  if (o->solution == TRX_SOL_ECLIPSE) {
    const int an = o->nangles;
    double *grid = calloc(an + 1, sizeof(double));          /* eclipse.c:262-269 */
    grid[0] = 0.0 * C_DEG; grid[an] = 90.0 * C_DEG;
    for (int i = 1; i < an; i++) grid[i] = (o->angles_deg[i-1] + o->angles_deg[i]) * C_DEG / 2.0;
    for (long w = 0; w < nsh; w++) spectrum[w] = 0.0;
    for (int i = 0; i < an; i++) {
      const double area = pow(sin(grid[i+1]), 2.0) - pow(sin(grid[i]), 2.0);
      for (long w = w0; w < w1; w++) {
        const double wv = s->wn_i + (double)w * s->wn_d;
        const double I = intensity(tau + (size_t)w*nr, wv * o->wn_fct, last[w],
                                   o->angles_deg[i], a->temp, nr);
        if (dbg && dbg->intens) dbg->intens[(size_t)i*nsh + (w - w0)] = I;
        spectrum[w - w0] += C_PI * I * area;                     /* eclipse.c:275-279 */
      }
    }
    free(grid);
  } else {
    for (long w = w0; w < w1; w++) {
      double m;
      if (o->modlevel == -1)
        m = modulation_rad(tau + (size_t)w*nr, last[w], o->toomuch, hh, a->rad_fct, o->starrad_cm);
      else
        m = modulation_int(tau + (size_t)w*nr, last[w], o->toomuch, hh, nr, a->rad_fct,
                           o->starrad_cm, o->transparent);
      if (o->modlevel == -1 && m < 0) { rc = TRX_E_NOTREACHED; goto done; }
      spectrum[w - w0] = m;
    }
  }

  {
    long swept = 0;
    for (long i = 0; i < nr; i++) swept += comp[i];
    h->stats.layers_swept = swept;
  }
  if (dbg) {
    if (dbg->e)    for (long r = 0; r < nr; r++) memcpy(dbg->e + (size_t)r*nsh, e + (size_t)r*nwn + w0, sizeof(double)*nsh);
    if (dbg->tau)  memcpy(dbg->tau, tau + (size_t)w0*nr, sizeof(double)*nsh*nr);
    if (dbg->last) for (long w = w0; w < w1; w++) dbg->last[w - w0] = last[w];
    if (dbg->computed) memcpy(dbg->computed, comp, nr);
    if (dbg->e_cs)                                          /* transposed to [layer][wn] */
      for (long w = w0; w < w1; w++)
        for (long r = 0; r < nr; r++) dbg->e_cs[(size_t)r*nsh + (w - w0)] = ecs[(size_t)w*nr + r];
  }
done:
  free(ecs); free(e); free(tau); free(last); free(comp); free(dens); free(zp);
  free(er); free(es); free(ec); free(hh); free(mdens); free(nH); free(rad);
  return rc;
}
