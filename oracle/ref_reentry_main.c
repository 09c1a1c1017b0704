/* ref_reentry_main.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Our own driver around the compiled reference's library API
 * (transit/src/transit.c:14-22, exported to BART through transit.i:97-105):
 *     transit_init(argc, argv); run_transit(input, n, out, nout) x K; free_memory();
 * It links the reference's objects as built by oracle/Makefile (transit.c with
 * -DTEST_TRANSIT so that its own main() steps aside, transit.c:230-234) and
 * contains no reference code.  Usage:
 *     transit_reentry <cfg> <inputs.txt> <out_prefix>
 * inputs.txt: one run per line, (1+nmol)*nlayer numbers: T(nlayer), q_0(nlayer), ...; a line
 * that starts with a word calls one of the reference's setters between runs (transit.c:97-116):
 *     radius <refradius>   |   cloudtop <log10 bar>   |   scattering <flag> <log10 factor>
 * Writes <out_prefix><k>.dat with one flux/modulation value per line (%.17g) and
 * <out_prefix><k>_radii.dat: per layer the reference's radius sampling after the reload
 * (tr->rads.v, the output of radpress + makeradsample, readatm.c:787-865, makesample.c:409-549)
 * at %.17g, so that the host side's restatement of radpress can be checked bit for bit.  (The
 * impact parameters are these radii reversed, makesample.c:563-572, and are freed again at the
 * end of every run, transit.c:203.)
 * The reference's headers are included from where they lie (oracle/Makefile passes -I); the
 * only thing read through them is the global `struct transit transit` (transit.h:36).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <transit.h>

int main(int argc, char **argv)
{
  if (argc < 4) { fprintf(stderr, "usage: %s cfg inputs.txt out_prefix\n", argv[0]); return 2; }
  char *targv[] = {"transit", "-c", argv[1], NULL};
  transit_init(3, targv);
  const int nwn = get_no_samples();
  FILE *in = fopen(argv[2], "r");
  if (!in) { perror(argv[2]); return 1; }
  size_t cap = 1 << 22;
  char *line = malloc(cap);
  double *out = malloc(sizeof(double) * nwn);
  int run = 0;
  while (fgets(line, (int)cap, in)) {
    double a = 0, b = 0;
    if (sscanf(line, " radius %lf", &a) == 1) { set_radius(a); continue; }
    if (sscanf(line, " cloudtop %lf", &a) == 1) { set_cloudtop(a); continue; }
    if (sscanf(line, " scattering %lf %lf", &a, &b) == 2) { set_scattering((int)a, b); continue; }
    size_t n = 0, room = 1024;
    double *v = malloc(sizeof(double) * room);
    char *p = line, *e;
    for (;;) {
      const double x = strtod(p, &e);
      if (e == p) break;
      if (n == room) { room *= 2; v = realloc(v, sizeof(double) * room); }
      v[n++] = x; p = e;
    }
    if (n == 0) { free(v); continue; }
    run_transit(v, (int)n, out, nwn);
    char name[1024];
    snprintf(name, sizeof name, "%s%d.dat", argv[3], ++run);
    FILE *o = fopen(name, "w");
    for (int i = 0; i < nwn; i++) fprintf(o, "%.17g\n", out[i]);
    fclose(o);
    snprintf(name, sizeof name, "%s%d_radii.dat", argv[3], run);
    o = fopen(name, "w");
    fprintf(o, "# rads.v  (rads.fct %.17g)\n", transit.rads.fct);
    for (long i = 0; i < transit.rads.n; i++) fprintf(o, "%.17g\n", transit.rads.v[i]);
    fclose(o);
    free(v);
  }
  fclose(in);
  free_memory();
  free(line); free(out);
  return 0;
}
