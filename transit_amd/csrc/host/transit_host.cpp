// transit_host.cpp -- host side of the drop-in (see include/transit_host.h).
//
// Restates, in C++, what the reference does before and after its spectrum path:
//   options ............ transit/src/argum.c:112-320 (table), :361-739 (parse),
//                        pu/src/procopt.c:651-705 (cfg lines, prefix match)
//   wavenumber grids ... transit/src/makesample.c:28-120, 309-400
//   atmosphere ......... transit/src/readatm.c:24-118, 277-428, 444-620, 626-717
//   TLI line list ...... transit/src/readlineinfo.c:17-77, 88-244, 250-278, 416-537
//   layer resampling ... transit/src/makesample.c:409-549 (+ pu/src/spline.c)
//   CIA tables ......... transit/src/crosssec.c:10-268
//   writers ............ eclipse.c:356-380, slantpath.c:511-555, tau.c:612-640
// No spectrum arithmetic lives here; the output is the POD description that
// include/transit_hip.h consumes.
#include "../../../include/transit_host.h"
#include "../trx_numerics.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

// What the host side does with an accepted option:
//   'a' acted on (changes the problem description, an output file, or the run's verbosity)
//   'n' parsed exactly as the reference parses it and, as in the reference, without any
//       effect on what this path writes (a note is recorded, trh_messages)
//   'w' accepted with a warning: results are unaffected, a resource detail differs
//   'x' rejected with TRX_E_UNSUPPORTED and a message
struct OptDef { const char *name; char shortc; bool has_arg; const char *def; char kind; const char *help; };

// Same names, order and defaults as the reference's table (argum.c:112-320);
// order matters because cfg tokens are matched as prefixes, first hit wins.
const OptDef kOptions[] = {
  {"version", 'V', false, nullptr, 'a', "print the version and exit"},
  {"help", 'h', false, nullptr, 'a', "print this list and exit"},
  {"quiet", 'q', false, nullptr, 'a', "verbosity 1"}, {"verb", 'v', true, "2", 'a', "verbosity level"},
  {"config_file", 'c', true, nullptr, 'a', "read options from a file"},
  {"atm", 0, true, nullptr, 'a', "atmosphere file"}, {"linedb", 0, true, nullptr, 'a', "TLI line database"},
  {"outtoomuch", 0, true, nullptr, 'a', "file: depth where toomuch was reached"},
  {"outsample", 0, true, nullptr, 'a', "file: sampling information (written with savefiles)"},
  {"outspec", 0, true, "outspectrum", 'a', "spectrum file"}, {"outintens", 0, true, nullptr, 'a', "per-angle intensity file"},
  {"molfile", 0, true, "../inputs/molecules.dat", 'a', "molecule table"},
  {"savefiles", 0, true, nullptr, 'a', "dump tau, extinctions and CIA under fixed names"},
  {"raddelt", 0, true, "-1", 'a', "radius spacing (-1: keep the file's layers)"},
  {"radlow", 0, true, "0", 'a', "lower radius"}, {"radhigh", 0, true, "0", 'a', "upper radius"},
  {"radfct", 0, true, "0", 'a', "radius units factor of a resampled grid"},
  {"allowq", 0, true, "0.00001", 'a', "allowed departure of the abundance sum from 1 (warning)"},
  {"refpress", 0, true, nullptr, 'a', "reference pressure"},
  {"refradius", 0, true, nullptr, 'a', "reference radius"}, {"gsurf", 0, true, nullptr, 'a', "surface gravity"},
  {"qmol", 0, true, nullptr, 'a', "species whose abundance is scaled"}, {"qscale", 0, true, nullptr, 'a', "log10 scale factors"},
  {"wllow", 0, true, nullptr, 'a', "lower wavelength"}, {"wlhigh", 0, true, nullptr, 'a', "upper wavelength"},
  {"wlfct", 0, true, "1e-4", 'a', "wavelength units factor"},
  {"wnlow", 0, true, nullptr, 'a', "lower wavenumber"}, {"wnhigh", 0, true, nullptr, 'a', "upper wavenumber"},
  {"wndelt", 0, true, "0", 'a', "wavenumber spacing"},
  {"wnosamp", 0, true, "2160", 'a', "wavenumber oversampling"}, {"wnfct", 0, true, "0", 'a', "wavenumber units factor"},
  {"ndop", 0, true, "60", 'a', "Doppler-width samples"}, {"nlor", 0, true, "60", 'a', "Lorentz-width samples"},
  {"dmin", 0, true, "1e-3", 'a', "smallest Doppler width"}, {"dmax", 0, true, "0.25", 'a', "largest Doppler width"},
  {"lmin", 0, true, "1e-4", 'a', "smallest Lorentz width"}, {"lmax", 0, true, "10.0", 'a', "largest Lorentz width"},
  {"nwidth", 'a', true, "20", 'a', "profile half-size in widths"},
  {"ethreshold", 0, true, "1e-8", 'a', "line-strength threshold"}, {"cloud", 0, true, nullptr, 'a', "cloud model"},
  {"cloudtop", 0, true, nullptr, 'a', "cloud-deck top (log bar)"}, {"scattering", 0, true, nullptr, 'a', "scattering model"},
  {"detailext", 0, true, nullptr, 'a', "file:wn,... extinction at the given wavenumbers"},
  {"detailcia", 0, true, nullptr, 'a', "file:wn,... CIA extinction at the given wavenumbers"},
  {"csfile", 0, true, nullptr, 'a', "cross-section files"},
  {"saveext", 0, true, nullptr, 'a', "extinction save/restore file (extinction.c:62-137)"},
  {"opacityfile", 0, true, nullptr, 'a', "opacity-grid file"}, {"tlow", 0, true, "500", 'a', "grid: lowest temperature"},
  {"thigh", 0, true, "3000", 'a', "grid: highest temperature"},
  {"tempdelt", 0, true, "100.0", 'a', "grid: temperature spacing"}, {"justOpacity", 0, false, nullptr, 'a', "stop after the grid"},
  {"shareOpacity", 0, false, nullptr, 'w', "no effect: every handle keeps its own copy of the grid in device memory"},
  {"solution", 's', true, "eclipse", 'a', "eclipse or transit"}, {"toomuch", 0, true, "20", 'a', "optical-depth cut"},
  {"taulevel", 0, true, "1", 'a', "1 (2 is rejected)"}, {"modlevel", 0, true, "1", 'a', "1 or -1"},
  {"detailtau", 0, true, nullptr, 'a', "file:wn,... optical depth at the given wavenumbers"},
  {"starrad", 0, true, "1.125", 'a', "stellar radius"},
  {"gorbpar", 0, true, nullptr, 'n', "orbital parameters (no output of this path depends on them)"},
  {"gorbparfct", 0, true, nullptr, 'n', "units of the orbital parameters (no output depends on them)"},
  {"transparent", 0, false, nullptr, 'a', "transparent planet"},
  {"raygrid", 0, true, "0 20 40 60 80", 'a', "emission angles"},
};
constexpr int kNumOptions = sizeof(kOptions) / sizeof(kOptions[0]);

struct Fail : std::runtime_error { int code; Fail(int c, const std::string &m) : std::runtime_error(m), code(c) {} };

std::string join_path(const std::string &dir, const std::string &f)
{
  if (f.empty() || f[0] == '/' || dir.empty()) return f;
  return dir + "/" + f;
}
std::string rstrip(std::string s)
{
  while (!s.empty() && (s.back() == '\n' || s.back() == '\r' || s.back() == ' ' || s.back() == '\t')) s.pop_back();
  return s;
}
std::vector<std::string> split_ws(const std::string &s)
{
  std::vector<std::string> out; std::istringstream is(s); std::string t;
  while (is >> t) out.push_back(t);
  return out;
}

// natural cubic spline resampling, pu/src/spline.c:97-128 (splinterp -> tri + spline3)
void splinterp(const std::vector<double> &xi, const std::vector<double> &yi,
               const std::vector<double> &xo, std::vector<double> &yo)
{
  const long n = (long)xi.size();
  std::vector<double> z(n), u(n), v(n);
  trx::spline_second_derivs(z.data(), xi.data(), yi.data(), n, u.data(), v.data());
  yo.resize(xo.size());
  // Resampling onto the spline's own abscissa (a retrieval loop: the layers' radii are the
  // atmosphere's): at a node below the last one spline3 takes the interval that starts there, the
  // distance is +0, both powers are +0 and every term it adds to y[k] is a zero: y[k] itself (a
  // negative zero aside, which no temperature, pressure, density or abundance is).  The last node is
  // reached from the interval before it and goes through the polynomial.
  const bool same = xo.size() == xi.size() && n >= 2 && std::equal(xo.begin(), xo.end(), xi.begin());
  for (size_t k = 0; k < xo.size(); k++)
    yo[k] = (same && (long)k < n - 1 && !(yi[k] == 0.0 && std::signbit(yi[k])))
              ? yi[k] : trx::spline_eval_pow(z.data(), n, xi.data(), yi.data(), xo[k]);
}

struct CiaTable { int nspec = 0; int mol[2] = {0, 0}; std::vector<double> wn, temp, cs; };

}  // namespace

struct trh_problem {
  std::map<std::string, std::string> opt;     // accepted option values
  std::string base_dir = ".";                 // relative paths resolve against the working directory, as in the reference

  // wavenumber sampling
  double wn_i = 0, wn_f = 0, wn_d = 0; int64_t nwn = 0, nown = 0; int osamp = 1;

  // atmosphere file (struct atm_data)
  std::vector<std::string> species;
  std::vector<double> a_rad, a_p, a_t, a_mm;            // [nlayer], file units
  std::vector<std::vector<double>> a_q, a_d;            // [nmol][nlayer]
  double rad_fct = 1, p_fct = 1, t_fct = 1, zerorad = 0; bool by_mass = true;
  float allowq = 1e-5f;
  // abundance scaling while the atmosphere file is read (qmol/qscale, argum.c:881-890)
  std::vector<std::string> qmol; std::vector<double> qscale;
  // molecules
  std::vector<int> mol_id; std::vector<double> mol_mass, mol_radius, mol_pol; std::vector<int32_t> mol_is_h2;
  // line list
  std::vector<double> wl, elow, gf; std::vector<int16_t> isoid;
  std::vector<std::string> iso_name; std::vector<double> iso_mass, iso_ratio; std::vector<int32_t> iso_imol;
  std::vector<int> iso_db;
  struct Db { std::string name, molname; std::vector<double> T; int first = 0, niso = 0; };
  std::vector<Db> dbs;
  std::vector<std::vector<double>> iso_z;               // [niso][nT(db)]
  std::vector<std::vector<double>> iso_z2;              // second derivatives of the partition-function splines (make_layer_sampling)
  double tli_tmin = 0, tli_tmax = 70000;
  // sampled layers (makeradsample)
  std::vector<double> rad, t, p, mm, dens, q, zpart, t_k;
  // CIA
  std::vector<CiaTable> cia; std::vector<trx_cia> cia_pod;
  // eclipse angles
  std::vector<double> angles;
  // BART re-entry
  double p0 = 0, r0 = 0, gsurf = 0;

  // opacity grid (struct opacity, structures_tr.h:154-171)
  bool grid_mode = false, needs_build = false;
  std::string opa_path;
  std::vector<double> og_temp, og_press, og_wns, og_o; std::vector<int32_t> og_molid, og_molidx;
  long og_nmol = 0, og_ntemp = 0, og_nlayer = 0, og_nwave = 0;
  trx_opacity_grid og_pod{};
  // build request: nv = nlayer*ntemp states
  std::vector<double> rq_temp, rq_dens, rq_z; std::vector<int32_t> rq_slot; int rq_nslot = 0;

  double out_rad_fct = 1;                               // rads.fct of the sampled layers (radfct when resampling)
  // detailext / detailtau / detailcia: file + requested wavenumbers (argum.c:385-415)
  struct Detail { std::string file; std::vector<double> wn; };
  Detail det_ext, det_tau, det_cia;
  // notes and warnings of the host side ("W: ..." / "I: ..."), newline separated
  std::string messages; int nwarn_q = 0;
  bool exit_requested = false;                          // --help / --version were served

  trx_static st{}; trx_atm atm{}; trx_opts opts{};
  std::string err;
};

namespace {

const OptDef *match_option(const std::string &token)
{
  for (int i = 0; i < kNumOptions; i++)
    if (std::strncmp(kOptions[i].name, token.c_str(), token.size()) == 0) return &kOptions[i];
  return nullptr;
}

void read_cfg(trh_problem &P, const std::string &path);

void note(trh_problem &P, char level, const std::string &msg)
{ P.messages += level; P.messages += ": "; P.messages += msg; P.messages += '\n'; }

// "filename:wn1,wn2,..." of detailext / detailtau / detailcia (argum.c:385-415)
void parse_detail(trh_problem::Detail &d, const std::string &value, const char *what)
{
  const size_t c = value.find(':');
  d.file = value.substr(0, c); d.wn.clear();
  if (c != std::string::npos) {
    std::string tok; std::istringstream is(value.substr(c + 1));
    while (std::getline(is, tok, ',')) {
      char *e; const double v = std::strtod(tok.c_str(), &e);
      if (e != tok.c_str()) d.wn.push_back(v);
    }
  }
  if (d.wn.empty() || d.file.empty())
    throw Fail(TRX_E_ARG, std::string("bad format for detailed ") + what + " parameter, no valid wavenumbers");
}

void print_help()
{
  std::printf("transit_hip: drop-in for `transit` on the spectrum path (MI355X).\n"
              "Options (--name value on the command line, `name value` lines in a -c file):\n");
  for (int i = 0; i < (int)(sizeof(kOptions) / sizeof(kOptions[0])); i++) {
    const OptDef &o = kOptions[i];
    std::printf("  --%-13s%s %-8s %s%s%s\n", o.name, o.shortc ? (std::string(" -") + o.shortc).c_str() : "   ",
                o.has_arg ? "<value>" : "", o.help, o.def ? "  [default " : "", o.def ? (std::string(o.def) + "]").c_str() : "");
  }
}

void accept(trh_problem &P, const OptDef *o, const std::string &value, const std::string &ctx_dir)
{
  const std::string name = o->name;
  if (name == "config_file") { read_cfg(P, join_path(ctx_dir, value)); return; }
  if (name == "quiet") { P.opt["verb"] = "1"; return; }
  if (name == "help") { print_help(); P.exit_requested = true; return; }                 // argum.c:605-607
  if (name == "version") {                                                               // argum.c:582-587
    std::printf("This is 'transit_hip' (MI355X spectrum path behind transit's interface), ABI %d\n\n", TRX_ABI_VERSION);
    P.exit_requested = true; return;
  }
  if (o->kind == 'x') throw Fail(TRX_E_UNSUPPORTED, "option '" + name + "': " + o->help);
  if (o->kind == 'w') note(P, 'W', "option '" + name + "': " + o->help);
  if (name == "gorbpar" || name == "gorbparfct") {          // six comma-separated numbers (argum.c:612-621)
    int n = 0; std::string tok; std::istringstream is(value);
    while (std::getline(is, tok, ',')) { char *e; std::strtod(tok.c_str(), &e); if (e == tok.c_str()) throw Fail(TRX_E_ARG, "bad number in " + name); n++; }
    if (n != 6) throw Fail(TRX_E_ARG, name + " needs six comma-separated values");
    note(P, 'I', "option '" + name + "' accepted; like in the reference, no output of the spectrum path depends on it");
  }
  if (name == "savefiles" && value.compare(0, 3, "yes") != 0 && value.compare(0, 2, "no") != 0)
    throw Fail(TRX_E_ARG, "allowed arguments for savefiles are: 'yes' or 'no'");                // argum.c:461-470
  if (name == "detailext") parse_detail(P.det_ext, value, "Extinction");
  if (name == "detailtau") parse_detail(P.det_tau, value, "Optical depth");
  if (name == "detailcia") parse_detail(P.det_cia, value, "CIA extinction");
  P.opt[o->name] = o->has_arg ? value : "1";
}

// pu/src/procopt.c:651-705: "name value" lines; '#' starts a comment line
void read_cfg(trh_problem &P, const std::string &path)
{
  std::ifstream f(path);
  if (!f) throw Fail(TRX_E_ARG, "cannot open configuration file '" + path + "'");
  std::string line;
  while (std::getline(f, line)) {
    line = rstrip(line);
    if (line.empty() || line[0] == '#') continue;
    size_t k = 0;
    while (k < line.size() && line[k] != ' ' && line[k] != '\t') k++;
    const std::string token = line.substr(0, k);
    while (k < line.size() && (line[k] == ' ' || line[k] == '\t')) k++;
    const OptDef *o = match_option(token);
    if (!o) throw Fail(TRX_E_ARG, "unknown option '" + token + "' in " + path);
    if (o->has_arg && k >= line.size()) throw Fail(TRX_E_ARG, "option '" + token + "' needs a value");
    accept(P, o, line.substr(k), ".");
  }
}

void parse_args(trh_problem &P, int argc, const char *const *argv)
{
  for (int i = 0; i < kNumOptions; i++)
    if (kOptions[i].def) P.opt[kOptions[i].name] = kOptions[i].def;
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    const OptDef *o = nullptr; std::string val; bool have_val = false;
    if (a.size() > 2 && a[0] == '-' && a[1] == '-') {
      std::string name = a.substr(2);
      const size_t eq = name.find('=');
      if (eq != std::string::npos) { val = name.substr(eq + 1); name = name.substr(0, eq); have_val = true; }
      o = match_option(name);
    } else if (a.size() >= 2 && a[0] == '-') {
      for (int k = 0; k < kNumOptions; k++) if (kOptions[k].shortc == a[1]) o = &kOptions[k];
      if (a.size() > 2) { val = a.substr(2); have_val = true; }
    }
    if (!o) throw Fail(TRX_E_ARG, "unknown argument '" + a + "'");
    if (o->has_arg && !have_val) {
      if (i + 1 >= argc) throw Fail(TRX_E_ARG, "missing value for '" + a + "'");
      val = argv[++i];
    }
    accept(P, o, val, ".");
  }
}

bool has(const trh_problem &P, const char *k) { return P.opt.count(k) != 0; }
double num(const trh_problem &P, const char *k, double dflt = 0)
{ auto it = P.opt.find(k); return it == P.opt.end() ? dflt : std::atof(it->second.c_str()); }
std::string str(const trh_problem &P, const char *k)
{ auto it = P.opt.find(k); return it == P.opt.end() ? std::string() : it->second; }

// makesample.c:28-120 (makesample1): number of points of an equispaced grid
int64_t sample_count(double ini, double fin, double d, int o)
{
  double excess = 1e-8;
  if (d < 0) excess = -excess;
  int64_t n = (int64_t)(((1.0 + excess) * fin - ini) / d + 1);
  if (n < 0) n = -n;
  return (n - 1) * o + 1;
}

// makesample.c:309-400 (makewnsample)
void make_wn_sampling(trh_problem &P)
{
  const double wni = num(P, "wnlow"), wnf = num(P, "wnhigh"), wnfct = num(P, "wnfct");
  const double wli = num(P, "wllow"), wlf = num(P, "wlhigh"), wlfct = num(P, "wlfct");
  double ri, rf;
  if (wni > 0) {
    if (wnfct <= 0) throw Fail(TRX_E_ARG, "wnlow given but wnfct is not positive");
    ri = wni * wnfct;
  } else if (wlf > 0) {
    if (wlfct <= 0) throw Fail(TRX_E_ARG, "wlfct is not positive");
    ri = 1.0 / (wlf * wlfct);
  } else throw Fail(TRX_E_ARG, "initial wavenumber (nor final wavelength) provided");
  if (wnf > 0) {
    if (wnfct < 0) throw Fail(TRX_E_ARG, "wnfct is negative");
    rf = wnf * wnfct;
  } else if (wli > 0) {
    if (wlfct < 0) throw Fail(TRX_E_ARG, "wlfct is negative");
    rf = 1.0 / (wli * wlfct);
  } else throw Fail(TRX_E_ARG, "final wavenumber (nor initial wavelength) provided");
  const double d = num(P, "wndelt");
  if (d <= 0) throw Fail(TRX_E_ARG, "wndelt must be positive");
  const int o = (int)num(P, "wnosamp");
  if (o <= 0) throw Fail(TRX_E_ARG, "wnosamp must be positive");
  if (rf < ri) throw Fail(TRX_E_ARG, "final wavenumber smaller than initial");
  P.wn_i = ri; P.wn_f = rf; P.wn_d = d; P.osamp = o;
  P.nown = sample_count(ri, rf, d, o);
  P.nwn  = sample_count(ri, rf, d, 1);
}

// molecules.dat, readatm.c:626-717
void read_molfile(trh_problem &P, const std::string &path)
{
  std::ifstream f(path);
  if (!f) throw Fail(TRX_E_ARG, "cannot open molecule file '" + path + "'");
  std::map<std::string, std::vector<double>> tab;   // name -> {id, mass, radius, pol}
  std::string line;
  while (std::getline(f, line)) {
    if (line.empty() || line[0] == '#' || line[0] == '\n') { if (!tab.empty()) break; continue; }
    auto w = split_ws(line);
    if (w.size() < 6) continue;
    tab[w[1]] = {(double)std::strtol(w[0].c_str(), nullptr, 10), std::strtod(w[2].c_str(), nullptr),
                 std::strtod(w[3].c_str(), nullptr) / 2.0, std::strtod(w[5].c_str(), nullptr)};
  }
  const size_t nm = P.species.size();
  P.mol_id.resize(nm); P.mol_mass.resize(nm); P.mol_radius.resize(nm); P.mol_pol.resize(nm);
  P.mol_is_h2.resize(nm);
  for (size_t i = 0; i < nm; i++) {
    auto it = tab.find(P.species[i]);
    if (it == tab.end()) throw Fail(TRX_E_ARG, "species '" + P.species[i] + "' not in " + path);
    P.mol_id[i] = (int)it->second[0];
    P.mol_mass[i] = it->second[1];
    P.mol_radius[i] = it->second[2] * 1e-8;            // Angstrom -> cm
    P.mol_pol[i] = it->second[3];
    P.mol_is_h2[i] = (P.species[i] == "H2");
  }
}

// readatm.c:121-158 (checkaddmm) + transit.h:58-69 (stateeqnford)
void layer_state(trh_problem &P, size_t r)
{
  const size_t nm = P.species.size();
  double mm = 0, sumq = 0;
  for (size_t i = 0; i < nm; i++) {
    if (P.by_mass) mm += P.a_q[i][r] / P.mol_mass[i];
    else           mm += P.a_q[i][r] * P.mol_mass[i];
    sumq += P.a_q[i][r];
  }
  if (P.by_mass) mm = 1.0 / mm;
  if (sumq > 1.001) throw Fail(TRX_E_ARG, "abundances add up to more than 1");
  if (std::fabs(sumq - 1.0) > P.allowq && P.nwarn_q++ < 8) {        // readatm.c:545-549, 754-757 (a warning)
    char b[160];
    std::snprintf(b, sizeof b, "in layer %zu (%g), abundances don't add up to 1.0: %.9g%s", r, P.a_rad[r], sumq,
                  P.nwarn_q == 8 ? " (further warnings of this kind are not recorded)" : "");
    note(P, 'W', b);
  }
  P.a_mm[r] = mm;
  const double p = P.a_p[r] * P.p_fct, t = P.a_t[r] * P.t_fct;
  for (size_t i = 0; i < nm; i++) {
    const double rho = trx::kAmu * P.a_q[i][r] * p / trx::kKb / t;
    P.a_d[i][r] = P.by_mass ? rho * mm : rho * P.mol_mass[i];
  }
}

// atmosphere file, readatm.c:277-428 (keywords) and :444-620 (rows)
void read_atmosphere(trh_problem &P, const std::string &path, const std::string &molpath)
{
  std::ifstream f(path);
  if (!f) throw Fail(TRX_E_ARG, "cannot open atmosphere file '" + path + "'");
  std::string line;
  std::vector<std::vector<double>> rows;
  bool in_data = false;
  while (std::getline(f, line)) {
    line = rstrip(line);
    if (line.empty()) continue;
    if (!in_data) {
      if (line[0] == '#') {
        auto w = split_ws(line.substr(1));
        if (!w.empty() && w[0] == "SPECIES") {
          if (!std::getline(f, line)) throw Fail(TRX_E_ARG, "EOF after #SPECIES");
          P.species = split_ws(line);
        }
        continue;
      }
      if (line[0] == 'q') {
        size_t k = 1; while (k < line.size() && line[k] == ' ') k++;
        const char c = (char)(line[k] | 0x20);
        if (c == 'n') P.by_mass = false; else if (c == 'm') P.by_mass = true;
        continue;
      }
      if (line[0] == 'z') { P.zerorad = std::atof(line.c_str() + 1); continue; }
      if (line[0] == 'u') {
        const double v = std::atof(line.c_str() + 2);
        if (line[1] == 'r') P.rad_fct = v; else if (line[1] == 'p') P.p_fct = v;
        else if (line[1] == 't') P.t_fct = v; else throw Fail(TRX_E_ARG, "bad unit line in atmosphere file");
        continue;
      }
      if (line[0] == 'n') continue;
      in_data = true;
    }
    if (line[0] == '#') continue;
    auto w = split_ws(line);
    std::vector<double> v; for (auto &s : w) v.push_back(std::strtod(s.c_str(), nullptr));
    rows.push_back(v);
  }
  if (P.species.empty()) throw Fail(TRX_E_ARG, "no #SPECIES header in atmosphere file");
  if (rows.empty()) throw Fail(TRX_E_ARG, "no layers in atmosphere file");
  const size_t nm = P.species.size(), nr = rows.size();
  read_molfile(P, molpath);
  P.a_rad.resize(nr); P.a_p.resize(nr); P.a_t.resize(nr); P.a_mm.resize(nr);
  P.a_q.assign(nm, std::vector<double>(nr)); P.a_d.assign(nm, std::vector<double>(nr));
  // qmol/qscale (readatm.c:394-405): species named in qmol get their abundance multiplied by
  // 10^qscale while the rows are read (:519-522); names that are not atmosphere species are
  // ignored, as in the reference.  H2 and He (molecule IDs 105 and 2, :465-466) are then
  // re-balanced to fill what the other species leave, keeping their ratio (:535-540).
  std::vector<int> qidx(P.qmol.size(), -1);
  int iH2 = -1, iHe = -1;
  if (!P.qscale.empty()) {
    for (size_t k = 0; k < P.qmol.size(); k++)
      for (size_t j = 0; j < nm; j++) if (P.qmol[k] == P.species[j]) { qidx[k] = (int)j; break; }
    for (size_t j = 0; j < nm; j++) { if (iH2 < 0 && P.mol_id[j] == 105) iH2 = (int)j; if (iHe < 0 && P.mol_id[j] == 2) iHe = (int)j; }
    if (iH2 < 0 || iHe < 0)    // the reference indexes its abundance table with -1 here
      throw Fail(TRX_E_ARG, "qscale needs H2 and He among the atmosphere species (they absorb the re-normalisation)");
  }
  for (size_t r = 0; r < nr; r++) {
    if (rows[r].size() < 3 + nm) throw Fail(TRX_E_ARG, "atmosphere row with too few columns");
    P.a_rad[r] = rows[r][0] + P.zerorad; P.a_p[r] = rows[r][1]; P.a_t[r] = rows[r][2];
    double metals = 0.0;
    for (size_t i = 0; i < nm; i++) {
      double q = rows[r][3 + i];
      for (size_t k = 0; k < qidx.size(); k++)
        if (qidx[k] == (int)i) { q *= std::pow(10.0, P.qscale[k]); break; }
      P.a_q[i][r] = q;
      if ((int)i != iH2 && (int)i != iHe) metals += q;
    }
    if (!P.qscale.empty()) {
      const double ratio = P.a_q[iH2][r] / P.a_q[iHe][r];
      P.a_q[iHe][r] =         (1 - metals) / (1.0 + ratio);
      P.a_q[iH2][r] = ratio * (1 - metals) / (1.0 + ratio);
    }
    layer_state(P, r);
  }
  bool sorted = true, reversed = true;                     // readatm.c:583-617
  for (size_t i = 0; i + 1 < nr; i++) {
    if (P.a_rad[i] >= P.a_rad[i+1] || P.a_p[i] <= P.a_p[i+1]) sorted = false;
    if (P.a_rad[i] <= P.a_rad[i+1] || P.a_p[i] >= P.a_p[i+1]) reversed = false;
  }
  if (!sorted && !reversed && nr > 1) throw Fail(TRX_E_ARG, "atmosphere layers are not monotonic");
  if (reversed && nr > 1) {
    auto rev = [](std::vector<double> &v) { for (size_t i = 0, j = v.size() - 1; i < j; i++, j--) std::swap(v[i], v[j]); };
    rev(P.a_rad); rev(P.a_p); rev(P.a_t); rev(P.a_mm);
    for (size_t i = 0; i < nm; i++) { rev(P.a_q[i]); rev(P.a_d[i]); }
  }
}

// readlineinfo.c:17-77 (datafileBS): bisection on the file's ascending wavelength block of one
// isotope -- here on the memory-mapped file, so only the pages the bisection and the selected
// range touch are ever read (the block sits at an arbitrary byte offset: unaligned loads)
struct WlBlock {
  const unsigned char *p;
  double operator[](int64_t i) const { double v; std::memcpy(&v, p + 8 * (size_t)i, 8); return v; }
};
int64_t tli_search(WlBlock w, int64_t n, double target, bool up)
{
  int64_t lo = 0, hi = n - 1, loc;
  do {
    loc = (hi + lo) / 2;
    if (target > w[loc]) lo = loc; else hi = loc;
  } while (hi - lo > 1);
  if (up) { loc = lo; while (loc < n - 1) { if (w[loc+1] > target) break; loc++; } }
  else    { loc = hi; while (loc > 0)     { if (w[loc-1] < target) break; loc--; } }
  return loc;
}

// read-only mapping of a whole file
struct MappedFile {
  const unsigned char *p = nullptr; size_t n = 0; int fd = -1;
  explicit MappedFile(const std::string &path) {
    fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) throw Fail(TRX_E_ARG, "cannot open TLI file '" + path + "'");
    struct stat st;
    if (::fstat(fd, &st) != 0 || st.st_size <= 0) { ::close(fd); throw Fail(TRX_E_ARG, "cannot stat TLI file '" + path + "'"); }
    n = (size_t)st.st_size;
    void *m = ::mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) { ::close(fd); throw Fail(TRX_E_ARG, "cannot map TLI file '" + path + "'"); }
    p = (const unsigned char *)m;
  }
  ~MappedFile() { if (p) ::munmap((void *)p, n); if (fd >= 0) ::close(fd); }
  MappedFile(const MappedFile &) = delete;
  MappedFile &operator=(const MappedFile &) = delete;
};

// TLI v6: readlineinfo.c:88-244 (header), :416-537 (range selection)
void read_tli(trh_problem &P, const std::string &path)
{
  // the file is mapped, not read: line lists run to 10^9 transitions (tens of GB) and a run
  // needs the header, ~60 probes per isotope and the selected wavelength window
  MappedFile mf(path);
  const unsigned char *buf = mf.p; const size_t bufsz = mf.n;
  size_t pos = 0;
  auto need = [&](size_t n) { if (pos + n > bufsz) throw Fail(TRX_E_ARG, "TLI file truncated"); };
  auto rd = [&](void *dst, size_t n) { need(n); std::memcpy(dst, buf + pos, n); pos += n; };
  auto rstr = [&]() { uint16_t n; rd(&n, 2); need(n); std::string s((const char *)buf + pos, n); pos += n; return s; };
  unsigned char magic[4]; rd(magic, 4);
  uint16_t ver[3]; rd(ver, 6);
  if (ver[0] != 6) throw Fail(TRX_E_ARG, "TLI version is not 6");
  double wli, wlf; rd(&wli, 8); rd(&wlf, 8);
  uint16_t ndb; rd(&ndb, 2);
  int cum = 0;
  for (int d = 0; d < ndb; d++) {
    trh_problem::Db db; db.name = rstr(); db.molname = rstr();
    uint16_t nT, nI; rd(&nT, 2); rd(&nI, 2);
    if (nT < 1) throw Fail(TRX_E_ARG, "TLI database '" + db.name + "' has no partition-function temperatures");
    db.T.resize(nT); rd(db.T.data(), 8 * (size_t)nT);
    P.tli_tmin = std::fmax(P.tli_tmin, db.T[0]);
    P.tli_tmax = std::fmin(P.tli_tmax, db.T[nT-1]);
    db.first = cum; db.niso = nI;
    for (int k = 0; k < nI; k++) {
      P.iso_name.push_back(rstr());
      double m, r; rd(&m, 8); rd(&r, 8);
      P.iso_mass.push_back(m); P.iso_ratio.push_back(r); P.iso_db.push_back(d);
      std::vector<double> z(nT); rd(z.data(), 8 * (size_t)nT); P.iso_z.push_back(z);
    }
    cum += nI; P.dbs.push_back(db);
  }
  // range check, readlineinfo.c:306-352 (checkrange)
  {
    const double wlmin = 1.0 / P.wn_f, wlmax = 1.0 / P.wn_i;
    if (wli * trx::kTliWfct > wlmax) throw Fail(TRX_E_ARG, "spectrum ends below the TLI wavelength range");
    if (wlf * trx::kTliWfct < wlmin) throw Fail(TRX_E_ARG, "spectrum starts above the TLI wavelength range");
  }
  // setimol, readlineinfo.c:250-278
  for (size_t i = 0; i < P.iso_name.size(); i++) {
    int im = -1;
    for (size_t m = 0; m < P.species.size(); m++) if (P.species[m] == P.dbs[P.iso_db[i]].molname) { im = (int)m; break; }
    if (im < 0) throw Fail(TRX_E_ARG, "TLI molecule '" + P.dbs[P.iso_db[i]].molname + "' is not an atmosphere species");
    P.iso_imol.push_back(im);
  }
  // the counts index straight into the mapping: nothing is read through them until they add up
  int64_t nlines; int32_t nisol; rd(&nlines, 8); rd(&nisol, 4);
  if (nlines < 0 || nisol < 0 || (size_t)nisol > P.iso_name.size() || (uint64_t)nlines > bufsz / 26)
    throw Fail(TRX_E_ARG, "TLI line counts are not consistent with the file (corrupt header?)");
  std::vector<int64_t> cnt((size_t)nisol); rd(cnt.data(), 8 * (size_t)nisol);
  {
    int64_t sum = 0;
    for (int k = 0; k < nisol; k++) {
      if (cnt[k] < 0 || cnt[k] > nlines) throw Fail(TRX_E_ARG, "TLI per-isotope line count is negative or exceeds the total");
      sum += cnt[k];
    }
    if (sum != nlines) throw Fail(TRX_E_ARG, "TLI per-isotope line counts do not add up to the number of lines");
  }
  const size_t wl0 = pos, iso0 = wl0 + 8 * (size_t)nlines, el0 = iso0 + 2 * (size_t)nlines, gf0 = el0 + 8 * (size_t)nlines;
  if (gf0 + 8 * (size_t)nlines > bufsz) throw Fail(TRX_E_ARG, "TLI line blocks truncated");
  const double iniw = 1.0 / (P.wn_f * 1.0) / trx::kTliWfct, finw = 1.0 / (P.wn_i * 1.0) / trx::kTliWfct;
  int64_t off = 0;
  for (int k = 0; k < nisol; k++) {
    if (cnt[k] > 0) {
      const WlBlock w{buf + wl0 + 8 * (size_t)off};
      const int64_t a = tli_search(w, cnt[k], iniw, false);
      const int64_t b = tli_search(w, cnt[k], finw, true);
      const int64_t nread = b - a + 1;
      if (nread > 0) {
        const size_t o0 = P.wl.size();
        P.wl.resize(o0 + nread); P.elow.resize(o0 + nread); P.gf.resize(o0 + nread); P.isoid.resize(o0 + nread);
        std::memcpy(&P.wl[o0],    buf + wl0  + 8 * (size_t)(off + a), 8 * (size_t)nread);
        std::memcpy(&P.isoid[o0], buf + iso0 + 2 * (size_t)(off + a), 2 * (size_t)nread);
        std::memcpy(&P.elow[o0],  buf + el0  + 8 * (size_t)(off + a), 8 * (size_t)nread);
        std::memcpy(&P.gf[o0],    buf + gf0  + 8 * (size_t)(off + a), 8 * (size_t)nread);
      }
    }
    off += cnt[k];
  }
}

// makesample.c:409-549 (makeradsample): layer grid + per-layer T, p, mm, q, rho, Z
void make_layer_sampling(trh_problem &P)
{
  const size_t nm = P.species.size(), na = P.a_rad.size();
  const double raddelt = num(P, "raddelt", -1);
  P.out_rad_fct = P.rad_fct;
  if (na == 1 || raddelt == -1) P.rad = P.a_rad;
  else {
    if (num(P, "radfct") > 0) P.out_rad_fct = num(P, "radfct");      // makesample.c:158-161 (hint fct wins when set)
    // makesample.c:144-300 (makesample) with the atmosphere sampling as reference
    double ini = num(P, "radlow"), fin = num(P, "radhigh");
    if (ini <= 0) ini = P.a_rad.front();
    if (fin <= 0) fin = P.a_rad.back();
    if (raddelt <= 0) throw Fail(TRX_E_ARG, "raddelt must be -1 or positive");
    if (fin <= ini) throw Fail(TRX_E_ARG, "radhigh must exceed radlow");
    const int64_t n = sample_count(ini, fin, raddelt, 1);
    P.rad.resize((size_t)n);
    for (int64_t k = 0; k < n; k++) P.rad[(size_t)k] = ini + (double)k * raddelt;
  }
  const size_t nr = P.rad.size();
  if (na == 1) {
    P.t = P.a_t; P.p = P.a_p; P.mm = P.a_mm;
    P.dens.resize(nm); P.q.resize(nm);
    for (size_t i = 0; i < nm; i++) { P.dens[i] = P.a_d[i][0]; P.q[i] = P.a_q[i][0]; }
  } else {
    splinterp(P.a_rad, P.a_t, P.rad, P.t);
    splinterp(P.a_rad, P.a_p, P.rad, P.p);
    splinterp(P.a_rad, P.a_mm, P.rad, P.mm);
    P.dens.resize(nm * nr); P.q.resize(nm * nr);
    std::vector<double> tmp;
    for (size_t i = 0; i < nm; i++) {
      splinterp(P.a_rad, P.a_d[i], P.rad, tmp); std::copy(tmp.begin(), tmp.end(), P.dens.begin() + i * nr);
      splinterp(P.a_rad, P.a_q[i], P.rad, tmp); std::copy(tmp.begin(), tmp.end(), P.q.begin() + i * nr);
    }
  }
  for (size_t r = 0; r < nr; r++)
    if (P.t[r] < P.tli_tmin || P.t[r] > P.tli_tmax)
      throw Fail(TRX_E_RANGE, "layer temperature outside the TLI partition-function range");
  const size_t ni = P.iso_name.size();
  P.zpart.resize(ni * nr);
  std::vector<double> tmp;
  // makesample.c:534-544 (T without tfct): splinterp of every isotope's partition function onto the
  // layer temperatures.  The spline's second derivatives depend on the TLI's table alone: they are
  // made once per isotope and kept (a retrieval loop calls this for every spectrum; four tridiagonal
  // sweeps over ~2000 temperatures were 80 of its ~100 us) -- the same doubles as splinterp's.
  if (P.iso_z2.size() != ni) {
    P.iso_z2.assign(ni, std::vector<double>());
    for (size_t i = 0; i < ni; i++) {
      const std::vector<double> &T = P.dbs[P.iso_db[i]].T;
      const long n = (long)T.size();
      std::vector<double> u(n), v(n);
      P.iso_z2[i].resize(n);
      trx::spline_second_derivs(P.iso_z2[i].data(), T.data(), P.iso_z[i].data(), n, u.data(), v.data());
    }
  }
  for (size_t i = 0; i < ni; i++) {
    const std::vector<double> &T = P.dbs[P.iso_db[i]].T;
    for (size_t r = 0; r < nr; r++)
      P.zpart[i * nr + r] = trx::spline_eval_pow(P.iso_z2[i].data(), (long)T.size(), T.data(), P.iso_z[i].data(), P.t[r]);
  }
  P.t_k.resize(nr);
  for (size_t r = 0; r < nr; r++) P.t_k[r] = P.t[r] * P.t_fct;
}

// crosssec.c:87-233
void read_cia(trh_problem &P, const std::string &path)
{
  std::ifstream f(path);
  if (!f) throw Fail(TRX_E_ARG, "cannot read cross-section file '" + path + "'");
  CiaTable c; std::string line; bool header = true;
  while (std::getline(f, line)) {
    line = rstrip(line);
    if (line.empty() || line[0] == '#') continue;
    if (header && line[0] == 'i') {
      auto w = split_ws(line.substr(1));
      if (w.size() != 1 && w.size() != 2) throw Fail(TRX_E_ARG, "bad 'i' line in " + path);
      c.nspec = (int)w.size();
      for (int k = 0; k < c.nspec; k++) {
        int im = -1;
        for (size_t m = 0; m < P.species.size(); m++) if (P.species[m] == w[k]) im = (int)m;
        if (im < 0) throw Fail(TRX_E_ARG, "CIA species '" + w[k] + "' is not an atmosphere species");
        c.mol[k] = im;
      }
      continue;
    }
    if (header && line[0] == 't') {
      const char *s = line.c_str() + 1; char *e;
      while (true) {
        const double v = std::strtod(s, &e);
        if (e == s) break;
        c.temp.push_back(v);
        if ((*e | 0x20) == 'k') e++;
        s = e;
      }
      if (c.temp.empty()) throw Fail(TRX_E_ARG, "no temperatures in " + path);
      continue;
    }
    header = false;
    const char *s = line.c_str(); char *e;
    const double w = std::strtod(s, &e);
    if (e == s) throw Fail(TRX_E_ARG, "bad wavenumber row in " + path);
    c.wn.push_back(w); s = e;
    for (size_t k = 0; k < c.temp.size(); k++) {
      const double v = std::strtod(s, &e);
      if (e == s) throw Fail(TRX_E_ARG, "short row in " + path);
      c.cs.push_back(v); s = e;
    }
  }
  if (c.wn.empty() || c.nspec == 0) throw Fail(TRX_E_ARG, "empty cross-section file " + path);
  const double wlast = P.wn_i + (double)(P.nwn - 1) * P.wn_d;
  if (c.wn.front() > P.wn_i || c.wn.back() < wlast)          // crosssec.c:251-259
    throw Fail(TRX_E_RANGE, "cross-section file does not cover the wavenumber range: " + path);
  P.cia.push_back(std::move(c));
}

// readatm.c:787-865 (radpress): hydrostatic radii around the reference level
// (p0 in atmosphere-file pressure units, r0 in radius units, gsurf in cm s-2).
void hydrostatic_radii(trh_problem &P)
{
  const int n = (int)P.a_rad.size();
  const double rfct = P.rad_fct, g0 = P.gsurf, p0 = P.p0, r0 = P.r0;
  std::vector<double> &rad = P.a_rad; const std::vector<double> &t = P.a_t, &mu = P.a_mm, &p = P.a_p;
  int i0 = -1; double best = 1e37;
  for (int i = 0; i < n; i++)
    if (std::fabs(p[i] - p0) < best) { i0 = i; best = std::fabs(p[i] - p0); }
  if (i0 < 0) throw Fail(TRX_E_RANGE, "reference pressure level not found");
  const int nb = (p[i0] > p0) ? i0 + 1 : i0 - 1;            // neighbour on the p0 side
  if (nb < 0 || nb >= n) throw Fail(TRX_E_RANGE, "reference pressure outside the layer range");
  const double lr = std::log(p[nb] / p[i0]);
  const double temp0 = t[i0]  + ((t[nb]  - t[i0])  / lr) * std::log(p0 / p[i0]);
  const double mu0   = mu[i0] + ((mu[nb] - mu[i0]) / lr) * std::log(p0 / p[i0]);
  const double kam = trx::kKb / trx::kAmu;
  if (p[i0] > p0) rad[i0] = r0 + 0.5 * (t[i0] / mu[i0] + temp0 / mu0) * (kam * std::log(p0 / p[i0]) / g0) / rfct;
  else            rad[i0] = r0 - 0.5 * (t[i0] / mu[i0] + temp0 / mu0) * (kam * std::log(p[i0] / p0) / g0) / rfct;
  double g = g0 * std::pow(r0 / rad[i0], 2);
  for (int i = i0 - 1; i >= 0; i--) {
    rad[i] = rad[i+1] - 0.5 * (t[i] / mu[i] + t[i+1] / mu[i+1]) * (kam * std::log(p[i] / p[i+1]) / g) / rfct;
    g = g * std::pow(rad[i+1] / rad[i], 2);
  }
  g = g0 * std::pow(r0 / rad[i0], 2);
  for (int i = i0 + 1; i < n; i++) {
    rad[i] = rad[i-1] + 0.5 * (t[i] / mu[i] + t[i-1] / mu[i-1]) * (kam * std::log(p[i-1] / p[i]) / g) / rfct;
    g = g * std::pow(rad[i-1] / rad[i], 2);
  }
}


// opacity.c:433-503 (readopacity): 4 longs, molID ints, temp/press/wns doubles, o[layer][temp][mol][wn]
void read_opacity(trh_problem &P, const std::string &path)
{
  FILE *fp = std::fopen(path.c_str(), "rb");
  if (!fp) throw Fail(TRX_E_ARG, "cannot open opacity file '" + path + "'");
  long dims[4];
  if (std::fread(dims, sizeof(long), 4, fp) != 4) { std::fclose(fp); throw Fail(TRX_E_ARG, "opacity file truncated"); }
  P.og_nmol = dims[0]; P.og_ntemp = dims[1]; P.og_nlayer = dims[2]; P.og_nwave = dims[3];
  if (P.og_nmol < 1 || P.og_ntemp < 2 || P.og_nlayer < 1 || P.og_nwave < 1) { std::fclose(fp); throw Fail(TRX_E_ARG, "bad opacity-file dimensions"); }
  std::vector<int> ids((size_t)P.og_nmol);
  P.og_temp.resize((size_t)P.og_ntemp); P.og_press.resize((size_t)P.og_nlayer); P.og_wns.resize((size_t)P.og_nwave);
  const size_t no = (size_t)P.og_nlayer * P.og_ntemp * P.og_nmol * P.og_nwave;
  P.og_o.resize(no);
  bool ok = std::fread(ids.data(), sizeof(int), ids.size(), fp) == ids.size() &&
            std::fread(P.og_temp.data(), 8, P.og_temp.size(), fp) == P.og_temp.size() &&
            std::fread(P.og_press.data(), 8, P.og_press.size(), fp) == P.og_press.size() &&
            std::fread(P.og_wns.data(), 8, P.og_wns.size(), fp) == P.og_wns.size() &&
            std::fread(P.og_o.data(), 8, no, fp) == no;
  std::fclose(fp);
  if (!ok) throw Fail(TRX_E_ARG, "opacity file truncated");
  P.og_molid.assign(ids.begin(), ids.end());
}

// makesample.c:613-636 (maketempsample) + opacity.c:297-403: the (layer x temperature) states
void make_grid_request(trh_problem &P)
{
  const double ti = num(P, "tlow"), tf = num(P, "thigh"), td = num(P, "tempdelt");
  if (tf < ti || td <= 0) throw Fail(TRX_E_ARG, "wrong temperature sampling (tlow, thigh, tempdelt)");
  const int64_t nt = sample_count(ti, tf, td, 1);
  P.og_temp.resize((size_t)nt);
  for (int64_t k = 0; k < nt; k++) P.og_temp[(size_t)k] = ti + (double)k * td;
  if (P.og_temp.front() < P.tli_tmin) throw Fail(TRX_E_RANGE, "opacity-grid temperature below the TLI range");
  if (P.og_temp.back() > P.tli_tmax) throw Fail(TRX_E_RANGE, "opacity-grid temperature above the TLI range");
  const size_t nr = P.rad.size(), nm = P.species.size(), ni = P.iso_mass.size();
  P.og_ntemp = (long)nt; P.og_nlayer = (long)nr; P.og_nwave = (long)P.nwn;
  P.og_press.resize(nr);
  for (size_t r = 0; r < nr; r++) P.og_press[r] = P.p[r] * P.p_fct;
  P.og_wns.resize((size_t)P.nwn);
  for (int64_t k = 0; k < P.nwn; k++) P.og_wns[(size_t)k] = P.wn_i + (double)k * P.wn_d;
  // molecules with line transitions, in isotope order (opacity.c:348-361)
  P.og_molid.clear(); P.og_molidx.clear(); P.rq_slot.assign(ni, 0);
  for (size_t i = 0; i < ni; i++) {
    const int id = P.mol_id[P.iso_imol[i]];
    int slot = -1;
    for (size_t k = 0; k < P.og_molid.size(); k++) if (P.og_molid[k] == id) slot = (int)k;
    if (slot < 0) { slot = (int)P.og_molid.size(); P.og_molid.push_back(id); P.og_molidx.push_back(P.iso_imol[i]); }
    P.rq_slot[i] = slot;
  }
  P.og_nmol = (long)P.og_molid.size(); P.rq_nslot = (int)P.og_molid.size();
  for (size_t i = 1; i < ni; i++)
    if (P.rq_slot[i] < P.rq_slot[i-1]) throw Fail(TRX_E_UNSUPPORTED, "isotopes of one molecule must be contiguous in the TLI");
  // partition function on the grid temperatures (opacity.c:324-339: spline_init + splinterp_pt)
  std::vector<double> ziso(ni * (size_t)nt);
  for (size_t i = 0; i < ni; i++) {
    const std::vector<double> &T = P.dbs[P.iso_db[i]].T;
    std::vector<double> z(T.size()), u(T.size()), v(T.size());
    trx::spline_second_derivs(z.data(), T.data(), P.iso_z[i].data(), (long)T.size(), u.data(), v.data());
    for (int64_t k = 0; k < nt; k++)
      ziso[i * nt + k] = trx::spline_eval_pt(z.data(), (long)T.size(), T.data(), P.iso_z[i].data(), P.og_temp[(size_t)k]);
  }
  const size_t nv = nr * (size_t)nt;
  P.rq_temp.resize(nv); P.rq_dens.resize(nm * nv); P.rq_z.resize(ni * nv);
  for (size_t r = 0; r < nr; r++)
    for (int64_t t = 0; t < nt; t++) {
      const size_t v = r * nt + t;
      P.rq_temp[v] = P.og_temp[(size_t)t];
      for (size_t j = 0; j < nm; j++) {                   // stateeqnford, opacity.c:393-394
        const double rho = trx::kAmu * P.q[j * nr + r] * P.og_press[r] / trx::kKb / P.og_temp[(size_t)t];
        P.rq_dens[j * nv + v] = P.by_mass ? rho * P.mm[r] : rho * P.mol_mass[j];
      }
      for (size_t i = 0; i < ni; i++) P.rq_z[i * nv + v] = ziso[i * nt + t];
    }
}

void set_grid_pod(trh_problem &P)
{
  // molecule IDs -> atmosphere species (extinction.c:575)
  P.og_molidx.assign(P.og_molid.size(), -1);
  for (size_t k = 0; k < P.og_molid.size(); k++) {
    for (size_t m = 0; m < P.mol_id.size(); m++) if (P.mol_id[m] == P.og_molid[k]) P.og_molidx[k] = (int32_t)m;
    if (P.og_molidx[k] < 0) throw Fail(TRX_E_ARG, "opacity-file molecule is not an atmosphere species");
  }
  if (P.og_nlayer != (long)P.rad.size() || P.og_nwave != (long)P.nwn)
    throw Fail(TRX_E_ARG, "opacity file does not match the layer / wavenumber sampling of this run");
  P.og_pod = trx_opacity_grid{};
  P.og_pod.nmol = P.og_nmol; P.og_pod.ntemp = P.og_ntemp; P.og_pod.nlayer = P.og_nlayer; P.og_pod.nwave = P.og_nwave;
  P.og_pod.mol_index = P.og_molidx.data(); P.og_pod.temp = P.og_temp.data(); P.og_pod.o = P.og_o.data();
  P.grid_mode = true;
}

void fill_pods(trh_problem &P)
{
  trx_static &s = P.st;
  s = trx_static{};
  s.abi_version = TRX_ABI_VERSION; s.device = 0;
  s.wn_i = P.wn_i; s.wn_d = P.wn_d; s.nwn = P.nwn; s.osamp = P.osamp; s.nown = P.nown;
  s.wn_lo = 0; s.wn_hi = P.nwn;
  s.ndop = (int)num(P, "ndop"); s.nlor = (int)num(P, "nlor");
  s.dmin = (float)num(P, "dmin"); s.dmax = (float)num(P, "dmax");
  s.lmin = (float)num(P, "lmin"); s.lmax = (float)num(P, "lmax");
  s.timesalpha = (float)num(P, "nwidth");
  s.nlines = (int64_t)P.wl.size();
  s.wl_um = P.wl.data(); s.isoid = P.isoid.data(); s.elow = P.elow.data(); s.gf = P.gf.data();
  s.niso = (int)P.iso_mass.size();
  s.iso_mass = P.iso_mass.data(); s.iso_ratio = P.iso_ratio.data(); s.iso_imol = P.iso_imol.data();
  s.nmol = (int)P.species.size();
  s.mol_mass = P.mol_mass.data(); s.mol_radius = P.mol_radius.data(); s.mol_pol = P.mol_pol.data();
  s.mol_is_h2 = P.mol_is_h2.data();
  P.cia_pod.clear();
  for (auto &c : P.cia) {
    trx_cia t{}; t.nspec = c.nspec; t.mol[0] = c.mol[0]; t.mol[1] = c.mol[1];
    t.nwave = (int)c.wn.size(); t.ntemp = (int)c.temp.size();
    t.wn = c.wn.data(); t.temp = c.temp.data(); t.cs = c.cs.data();
    P.cia_pod.push_back(t);
  }
  s.ncia = (int)P.cia_pod.size(); s.cia = P.cia_pod.data();
  s.comm = nullptr; s.nranks = 1; s.rank = 0; s.ogrid = P.grid_mode ? &P.og_pod : nullptr;

  trx_atm &a = P.atm;
  a = trx_atm{};
  a.nlayer = (int)P.rad.size(); a.rad_fct = P.out_rad_fct;
  a.radius = P.rad.data(); a.temp = P.t_k.data(); a.press = P.p.data();
  a.density = P.dens.data(); a.abund = P.q.data(); a.zpart = P.zpart.data();

  trx_opts &o = P.opts;
  const int keep_chunk = o.layer_chunk, keep_eager = o.eager;
  const int cf = o.cloud_flag; const double ce = o.cloud_ext, ct = o.cloud_top, cb = o.cloud_bot;
  const int sf = o.scat_flag; const double sl = o.scat_logext;
  const bool had = (o.toomuch != 0);
  o = trx_opts{};
  const std::string sol = str(P, "solution");
  if (sol == "eclipse") o.solution = TRX_SOL_ECLIPSE;
  else if (sol == "transit") o.solution = TRX_SOL_TRANSIT;
  else throw Fail(TRX_E_ARG, "solution kind '" + sol + "' is invalid (eclipse, transit)");
  o.toomuch = num(P, "toomuch"); o.ethresh = num(P, "ethreshold");
  if (o.ethresh <= 0) throw Fail(TRX_E_ARG, "ethreshold has to be positive");
  if ((float)num(P, "nwidth") < 1) throw Fail(TRX_E_ARG, "nwidth has to be greater than one");
  o.wn_fct = 1.0;                                          // makesample.c:369 (reference grid is cm-1)
  P.angles.clear();
  if (o.solution == TRX_SOL_ECLIPSE)
    for (auto &w : split_ws(str(P, "raygrid"))) P.angles.push_back(std::atof(w.c_str()));
  o.nangles = (int)P.angles.size(); o.angles_deg = P.angles.data();
  o.starrad_cm = num(P, "starrad") * trx::kSunRadius;      // geometry.c:33-47
  o.transparent = has(P, "transparent"); o.modlevel = (int)num(P, "modlevel", 1);
  if ((int)num(P, "taulevel", 1) != 1) throw Fail(TRX_E_UNSUPPORTED, "taulevel 2 is not implemented (neither in the reference, slantpath.c:135)");
  // clouds, argum.c:653-716
  if (has(P, "cloud")) {
    const std::string c = str(P, "cloud");
    if (c.find("ext") != std::string::npos) o.cloud_flag = 1;
    else if (c.find("opa") != std::string::npos) o.cloud_flag = 2;
    else if (c.find("B17") != std::string::npos) o.cloud_flag = 3;
    else if (c.find("F18") != std::string::npos) o.cloud_flag = 4;
    else if (c.find("P19") != std::string::npos) o.cloud_flag = 5;
    std::vector<double> v; std::string rest = c.size() > 4 ? c.substr(4) : ""; std::string tok; std::istringstream is(rest);
    while (std::getline(is, tok, ',')) v.push_back(std::atof(tok.c_str()));
    if (v.size() < 3) throw Fail(TRX_E_ARG, "cloud needs cloudtype,cloudext,cloudtop,cloudbot");
    o.cloud_ext = v[0]; o.cloud_top = v[1]; o.cloud_bot = v[2];
    o.cloud_gamma = o.cloud_Q = o.cloud_r = o.cloud_sig = o.cloud_refwn = 1;
    if (o.cloud_flag >= 3 && v.size() > 3) o.cloud_gamma = v[3];
    if (o.cloud_flag == 4 && v.size() > 5) { o.cloud_Q = v[4]; o.cloud_r = v[5]; }
    if (o.cloud_flag == 5 && v.size() > 5) { o.cloud_sig = v[4]; o.cloud_refwn = v[5]; }
    if (o.cloud_top > o.cloud_bot) throw Fail(TRX_E_ARG, "cloud top must be less than cloud bottom");
  }
  if (has(P, "cloudtop")) {                                 // argum.c:718-724
    o.cloud_top = num(P, "cloudtop"); o.cloud_bot = o.cloud_top + 10; o.cloud_ext = 100.0; o.cloud_flag = 1;
  }
  if (has(P, "scattering")) {                               // argum.c:726-740
    if (str(P, "scattering") == "polar") { o.scat_flag = 2; o.scat_logext = 0; }
    else { o.scat_flag = 1; o.scat_logext = num(P, "scattering"); }
  }
  if (had) {   // keep run-time overrides set through trh_set_* across reloads
    if (cf) { o.cloud_flag = cf; o.cloud_ext = ce; o.cloud_top = ct; o.cloud_bot = cb; }
    if (sf) { o.scat_flag = sf; o.scat_logext = sl; }
  }
  o.layer_chunk = keep_chunk; o.eager = keep_eager;
}

void load(trh_problem &P, int argc, const char *const *argv)
{
  parse_args(P, argc, argv);
  if (P.exit_requested) return;
  make_wn_sampling(P);
  if (!has(P, "atm") || str(P, "atm") == "NULL" || str(P, "atm") == "-")
    throw Fail(TRX_E_ARG, "no atmospheric file specified");
  P.allowq = (float)num(P, "allowq");
  if (has(P, "qscale")) {                                   // argum.c:881-890
    for (auto &w : split_ws(str(P, "qscale"))) P.qscale.push_back(std::atof(w.c_str()));
    if (!has(P, "qmol")) throw Fail(TRX_E_ARG, "qscale given without qmol");
    P.qmol = split_ws(str(P, "qmol"));
    if (P.qmol.size() != P.qscale.size())
      throw Fail(TRX_E_ARG, "qscale (" + std::to_string(P.qscale.size()) + ") and qmol (" + std::to_string(P.qmol.size()) +
                            ") should have the same number of elements");
  }
  read_atmosphere(P, join_path(P.base_dir, str(P, "atm")), join_path(P.base_dir, str(P, "molfile")));
  bool have_grid = false;
  if (has(P, "opacityfile")) {
    P.opa_path = join_path(P.base_dir, str(P, "opacityfile"));
    if (FILE *f = std::fopen(P.opa_path.c_str(), "rb")) { std::fclose(f); have_grid = true; }
  }
  // the TLI is only read when there is no opacity file yet (readlineinfo.c:584-596)
  if (has(P, "linedb") && !have_grid) read_tli(P, join_path(P.base_dir, str(P, "linedb")));
  make_layer_sampling(P);
  if (have_grid) { read_opacity(P, P.opa_path); set_grid_pod(P); }
  else if (has(P, "opacityfile")) { make_grid_request(P); P.needs_build = true; }
  if (has(P, "csfile")) {
    std::string tok; std::istringstream is(str(P, "csfile"));
    while (std::getline(is, tok, ',')) if (!tok.empty()) read_cia(P, join_path(P.base_dir, tok));
  }
  P.p0 = num(P, "refpress"); P.r0 = num(P, "refradius"); P.gsurf = num(P, "gsurf");
  fill_pods(P);
}

}  // namespace

extern "C" {

int trh_load(int argc, const char *const *argv, trh_problem **out, char *err, int errlen)
{
  if (!out) return TRX_E_ARG;
  trh_problem *P = new trh_problem();
  try { load(*P, argc, argv); }
  catch (const Fail &f) {
    if (err && errlen > 0) std::snprintf(err, (size_t)errlen, "%s", f.what());
    delete P; return f.code;
  }
  catch (const std::exception &e) {
    if (err && errlen > 0) std::snprintf(err, (size_t)errlen, "%s", e.what());
    delete P; return TRX_E_NOMEM;
  }
  if (P->exit_requested) { delete P; return 1; }      // --help / --version: printed, nothing to run
  *out = P;
  return TRX_OK;
}

void trh_free(trh_problem *p) { delete p; }
const char *trh_messages(const trh_problem *p) { return p ? p->messages.c_str() : ""; }

int trh_option_table(int i, const char **name, int *has_arg, char *kind)
{
  if (i < 0 || i >= kNumOptions) return TRX_E_ARG;
  if (name) *name = kOptions[i].name;
  if (has_arg) *has_arg = kOptions[i].has_arg ? 1 : 0;
  if (kind) *kind = kOptions[i].kind;
  return TRX_OK;
}
const trx_static *trh_static(const trh_problem *p) { return p ? &p->st : nullptr; }
const trx_atm *trh_atm(const trh_problem *p) { return p ? &p->atm : nullptr; }
const trx_opts *trh_opts(const trh_problem *p) { return p ? &p->opts : nullptr; }
int64_t trh_nwn(const trh_problem *p) { return p ? p->nwn : 0; }
void trh_wavenumbers(const trh_problem *p, double *out)
{ if (p && out) for (int64_t k = 0; k < p->nwn; k++) out[k] = p->wn_i + (double)k * p->wn_d; }
void trh_set_shard(trh_problem *p, int64_t lo, int64_t hi)
{ if (p) { p->st.wn_lo = lo; p->st.wn_hi = hi; } }

// Contiguous split of the coarse bins into nranks shards of near-equal WORK (SURVEY section 8e:
// balance by the work, not by the number of bins).  Work of a bin = its ray's optical depth and
// spectrum (~ layers) + the lines whose cell it is (every line is walked once per step whatever
// its bin); the two weights are the measured times of the demo-sized run.
// The k-th cut goes where the running sum crosses k/nranks of the total, to the nearer side.
int trh_shard_bounds(const trh_problem *p, int nranks, int64_t *bounds /* [nranks + 1] */)
{
  if (!p || !bounds || nranks < 1 || nranks > p->nwn) return TRX_E_ARG;
  const int64_t nwn = p->nwn;
  std::vector<double> run((size_t)nwn + 1, 0.0), c((size_t)nwn, 0.04 * (double)p->rad.size());
  for (size_t i = 0; i < p->wl.size(); i++) {
    const int64_t cell = (int64_t)std::floor((1e4 / p->wl[i] - p->wn_i) / p->wn_d + 0.5);
    if (cell >= 0 && cell < nwn) c[(size_t)cell] += 3.4e-4;
  }
  for (int64_t j = 0; j < nwn; j++) run[(size_t)j + 1] = run[(size_t)j] + c[(size_t)j];
  bounds[0] = 0; bounds[nranks] = nwn;
  for (int k = 1; k < nranks; k++) {
    const double target = run[(size_t)nwn] * k / nranks;
    int64_t j = std::lower_bound(run.begin(), run.end(), target) - run.begin();      // run[j-1] < target <= run[j]
    if (j > 0 && target - run[(size_t)j - 1] < run[(size_t)j] - target) j--;
    j = std::min<int64_t>(std::max<int64_t>(j, bounds[k - 1] + 1), nwn - (nranks - k));
    bounds[k] = j;
  }
  return TRX_OK;
}

int trh_reload_atm(trh_problem *p, const double *input, int n)
{
  if (!p || !input) return TRX_E_ARG;
  const size_t nl = p->a_rad.size(), nm = p->species.size();
  if ((size_t)n != (1 + nm) * nl) return TRX_E_ARG;
  try {
    for (size_t i = 0; i < nl; i++) p->a_t[i] = input[i];
    for (size_t j = 0; j < nm; j++) for (size_t i = 0; i < nl; i++) p->a_q[j][i] = input[nl * (j + 1) + i];
    for (size_t i = 0; i < nl; i++) layer_state(*p, i);
    if (p->p0 == 0 || p->r0 == 0 || p->gsurf == 0)
      throw Fail(TRX_E_ARG, "refpress, refradius and gsurf must be defined to reload an atmosphere");
    hydrostatic_radii(*p);
    make_layer_sampling(*p);
    fill_pods(*p);
  } catch (const Fail &f) { p->err = f.what(); return f.code; }
  return TRX_OK;
}
void trh_set_radius(trh_problem *p, double r) { if (p) p->r0 = r; }
void trh_set_cloudtop(trh_problem *p, double c)
{ if (p) { p->opts.cloud_top = c; p->opts.cloud_bot = c + 10; p->opts.cloud_ext = 100; p->opts.cloud_flag = 1; } }
void trh_set_scattering(trh_problem *p, int flag, double logext)
{ if (p) { p->opts.scat_flag = flag; p->opts.scat_logext = logext; } }

int trh_needs_opacity_build(const trh_problem *p) { return p && p->needs_build ? 1 : 0; }

int trh_grid_request(const trh_problem *p, int32_t *nv, const double **temp, const double **density,
                     const double **zpart, int32_t *nslot, const int32_t **iso_slot)
{
  if (!p || !p->needs_build) return TRX_E_ARG;
  if (nv) *nv = (int32_t)p->rq_temp.size();
  if (temp) *temp = p->rq_temp.data();
  if (density) *density = p->rq_dens.data();
  if (zpart) *zpart = p->rq_z.data();
  if (nslot) *nslot = p->rq_nslot;
  if (iso_slot) *iso_slot = p->rq_slot.data();
  return TRX_OK;
}

int trh_install_opacity(trh_problem *p, const double *o)
{
  if (!p || !o || !p->needs_build) return TRX_E_ARG;
  const size_t no = (size_t)p->og_nlayer * p->og_ntemp * p->og_nmol * p->og_nwave;
  p->og_o.assign(o, o + no);
  // opacity.c:405-421
  FILE *fp = std::fopen(p->opa_path.c_str(), "wb");
  if (!fp) return TRX_E_ARG;
  const long dims[4] = {p->og_nmol, p->og_ntemp, p->og_nlayer, p->og_nwave};
  std::vector<int> ids(p->og_molid.begin(), p->og_molid.end());
  bool ok = std::fwrite(dims, sizeof(long), 4, fp) == 4 &&
            std::fwrite(ids.data(), sizeof(int), ids.size(), fp) == ids.size() &&
            std::fwrite(p->og_temp.data(), 8, p->og_temp.size(), fp) == p->og_temp.size() &&
            std::fwrite(p->og_press.data(), 8, p->og_press.size(), fp) == p->og_press.size() &&
            std::fwrite(p->og_wns.data(), 8, p->og_wns.size(), fp) == p->og_wns.size() &&
            std::fwrite(p->og_o.data(), 8, no, fp) == no;
  std::fclose(fp);
  if (!ok) return TRX_E_ARG;
  try { set_grid_pod(*p); } catch (const Fail &f) { p->err = f.what(); return f.code; }
  p->needs_build = false;
  p->st.ogrid = &p->og_pod;
  return TRX_OK;
}

const char *trh_option(const trh_problem *p, const char *name)
{
  if (!p || !name) return nullptr;
  auto it = p->opt.find(name);
  return it == p->opt.end() ? nullptr : it->second.c_str();
}

int trh_write_spectrum(const trh_problem *p, const double *sp, const char *path)
{
  if (!p || !sp) return TRX_E_ARG;
  std::string f = path ? std::string(path) : join_path(p->base_dir, str(*p, "outspec"));
  FILE *out = (f.empty() || f[0] == '-') ? stdout : std::fopen(f.c_str(), "w");
  if (!out) return TRX_E_ARG;
  if (p->opts.solution == TRX_SOL_ECLIPSE) {                 // eclipse.c:370-375
    std::fprintf(out, "#wvl [um]%*sFlux [erg/s/cm]\n", 6, " ");
    for (int64_t k = 0; k < p->nwn; k++)
      std::fprintf(out, "%-15.10g%-18.9g\n", 1e4 / ((p->wn_i + (double)k * p->wn_d) / 1.0), sp[k]);
  } else {                                                  // slantpath.c:545-551
    std::fprintf(out, "#wvl [um]        modulation\n");
    for (int64_t k = 0; k < p->nwn; k++)
      std::fprintf(out, "%-17.9g%-18.9g\n", 1 / ((p->wn_i + (double)k * p->wn_d) / 1.0 * 1e-4), sp[k]);
  }
  if (out != stdout) std::fclose(out);
  return TRX_OK;
}

// --saveext FILE (tau.c:155-156, 340-341; extinction.c:62-137): the molecular extinction of the layers a
// run computed, kept "for a possible next run" -- magic "@E@S@", e[nlayer][nwn] doubles, one flag per
// layer (a short: the reference defines _Bool as short, transit.h:129).  Read: TRX_OK and the arrays filled, or 1 with a note in trh_messages when there is no
// (valid) file, as the reference warns and goes on.
int trh_saveext_read(trh_problem *p, double *e, uint8_t *computed)
{
  if (!p || !e || !computed) return TRX_E_ARG;
  if (!has(*p, "saveext")) return 1;
  const std::string f = join_path(p->base_dir, str(*p, "saveext"));
  const size_t nr = (size_t)p->atm.nlayer, ne = nr * (size_t)p->nwn;
  FILE *in = std::fopen(f.c_str(), "rb");
  if (!in) { p->messages += "W: extinction savefile '" + f + "' cannot be opened for reading: continuing without restoring\n"; return 1; }
  char mn[5];
  bool ok = std::fread(mn, 1, 5, in) == 5 && std::memcmp(mn, "@E@S@", 5) == 0;
  std::vector<int16_t> c16(nr);                              // (transit.h:129: "#define _Bool short" -- the flags are shorts)
  if (ok) ok = std::fread(e, sizeof(double), ne, in) == ne && std::fread(c16.data(), sizeof(int16_t), nr, in) == nr;
  if (ok) ok = std::fgetc(in) == EOF;                        // (a file of another grid is longer or shorter)
  std::fclose(in);
  for (size_t i = 0; ok && i < nr; i++) computed[i] = c16[i] != 0;
  if (!ok) { p->messages += "W: '" + f + "' is not a valid extinction savefile for this grid: not restored\n"; return 1; }
  return TRX_OK;
}

int trh_saveext_write(const trh_problem *p, const double *e, const uint8_t *computed)
{
  if (!p || !e || !computed) return TRX_E_ARG;
  if (!has(*p, "saveext")) return TRX_OK;
  const std::string f = join_path(p->base_dir, str(*p, "saveext"));
  const size_t nr = (size_t)p->atm.nlayer, ne = nr * (size_t)p->nwn;
  FILE *out = std::fopen(f.c_str(), "wb");
  if (!out) return TRX_E_ARG;
  std::vector<int16_t> c16(nr);
  for (size_t i = 0; i < nr; i++) c16[i] = computed[i] ? 1 : 0;
  const bool ok = std::fwrite("@E@S@", 1, 5, out) == 5 && std::fwrite(e, sizeof(double), ne, out) == ne &&
                  std::fwrite(c16.data(), sizeof(int16_t), nr, out) == nr;
  std::fclose(out);
  return ok ? TRX_OK : TRX_E_ARG;
}

int trh_write_toomuch(const trh_problem *p, const double *tau, const int64_t *last, const char *path)
{
  if (!p || !tau || !last) return TRX_E_ARG;
  std::string f = path ? std::string(path) : join_path(p->base_dir, str(*p, "outtoomuch"));
  if (f.empty()) return TRX_OK;
  FILE *out = f[0] == '-' ? stdout : std::fopen(f.c_str(), "w");
  if (!out) return TRX_E_ARG;
  const int64_t nr = (int64_t)p->rad.size();
  std::fprintf(out, "# Wavelength   Max Optical   Radius at the    Radius\n"
                    "   (microns)         depth   max depth (km)    index\n");
  for (int64_t w = 0; w < p->nwn; w++) {                     // tau.c:634-638 (ips = reversed radii)
    const double wn = p->wn_i + (double)w * p->wn_d;
    std::fprintf(out, "%12.7f   %.5e     %12.4f     %04ld\n", 1.0 / wn * 1.0 * 1e4,
                 tau[w * nr + last[w]], p->rad[(size_t)(nr - 1 - last[w])] * p->out_rad_fct / 1e5, (long)last[w]);
  }
  if (out != stdout) std::fclose(out);
  return TRX_OK;
}

// printintens (eclipse.c:293-350): wavelength + one column per ray angle
int trh_write_intens(const trh_problem *p, const double *intens, const char *path)
{
  if (!p || !intens) return TRX_E_ARG;
  std::string f = path ? std::string(path) : join_path(p->base_dir, str(*p, "outintens"));
  if (f.empty() || f[0] == '-') return TRX_OK;               // eclipse.c:311-318: no file requested
  FILE *out = std::fopen(f.c_str(), "w");
  if (!out) return TRX_E_ARG;
  const int an = p->opts.nangles;
  std::fprintf(out, "#wvl %*s", 10, " ");
  for (int i = 0; i < an; i++) std::fprintf(out, "I[%4.1lf deg]%*s", p->opts.angles_deg[i], 7, " ");
  std::fprintf(out, "\n#[um]%*s", 10, " ");
  for (int i = 0; i < an; i++) std::fprintf(out, "[erg/s/cm/sr]%*s", 5, " ");
  std::fprintf(out, "\n");
  for (int64_t w = 0; w < p->nwn; w++) {
    const double wn = p->wn_i + (double)w * p->wn_d;
    std::fprintf(out, "%-15.10g", 1e4 / (wn / 1.0));                     // wns.fct is 1 after makewnsample
    for (int i = 0; i < an; i++) std::fprintf(out, "%-18.9g", intens[(size_t)i * p->nwn + w]);
    std::fprintf(out, "\n");
  }
  std::fclose(out);
  return TRX_OK;
}

// detailout (tau.c:526-605): the optical depth, the molecular extinction or the CIA extinction
// at the wavenumbers requested with detailtau / detailext / detailcia, one row per radius
// (per impact parameter for the optical depth).  which: 0 tau, 1 ext, 2 cia.  Arrays in the
// layouts of trx_debug: tau [wn][height], e and e_cs [layer][wn].
// The reference prints the CIA table through a float pointer laid over its array of doubles
// (CIA_DOFLOAT with PREC_CS = double, tau.c:540,591): column m of a row shows 32 of the 64
// bits of value m/2.  That is what its files contain, so that is what is written here.
int trh_write_detail(const trh_problem *p, int which, const double *arr)
{
  if (!p || !arr || which < 0 || which > 2) return TRX_E_ARG;
  const trh_problem::Detail &d = which == 0 ? p->det_tau : which == 1 ? p->det_ext : p->det_cia;
  if (d.wn.empty()) return TRX_OK;
  const int64_t nr = (int64_t)p->rad.size(), nw = p->nwn;
  FILE *out = std::fopen(join_path(p->base_dir, d.file).c_str(), "w");
  if (!out) return TRX_E_ARG;
  auto wn_at = [&](int64_t w) { return p->wn_i + (double)w * p->wn_d; };
  std::vector<int64_t> idx;
  std::fprintf(out, "#Radius-w=>    ");
  for (double val : d.wn) {                                  // tau.c:554-572
    int64_t u = nw - 1, lo;
    if (val == wn_at(u)) lo = u;
    else { lo = 0; while (u - lo > 1) { const int64_t m = (u + lo) / 2; if (wn_at(m) > val) u = m; else lo = m; } }
    idx.push_back(lo);
    std::fprintf(out, "%-15.8g", wn_at(lo));
  }
  std::fprintf(out, "\n");
  std::vector<double> row((size_t)nr);
  for (int64_t m = 0; m < nr; m++) {
    std::fprintf(out, "%-15.7g", which == 0 ? p->rad[(size_t)(nr - 1 - m)] : p->rad[(size_t)m]);
    for (int64_t w : idx) {
      double val;
      if (which == 0) val = arr[w * nr + m];
      else if (which == 1) val = arr[m * nw + w];
      else {
        for (int64_t k = 0; k < nr; k++) row[(size_t)k] = arr[k * nw + w];   // the reference's e_cs[wn][layer] row
        float f; std::memcpy(&f, (const char *)row.data() + 4 * (size_t)m, 4);
        val = f;
      }
      std::fprintf(out, "%-15.7g", val);
    }
    std::fprintf(out, "\n");
  }
  std::fclose(out);
  return TRX_OK;
}
int trh_wants_detail(const trh_problem *p, int which)
{
  if (!p) return 0;
  return !(which == 0 ? p->det_tau : which == 1 ? p->det_ext : p->det_cia).wn.empty();
}

// outsample (makesample.c:642-672, 744-770; written by makeipsample when `savefiles` is set,
// :598-599): the four samplings as the reference holds them at that point -- the wavelength
// sampling is never filled on this path, its block is all zeros.
int trh_write_sample(const trh_problem *p, const char *path)
{
  if (!p) return TRX_E_ARG;
  const std::string f = path ? std::string(path) : str(*p, "outsample");
  if (f.empty()) return TRX_OK;
  FILE *out = f == "-" ? stdout : std::fopen(join_path(p->base_dir, f).c_str(), "w");
  if (!out) return TRX_E_ARG;
  auto head = [&](const char *desc, double fct, double i, double fin, double d) {
    std::fprintf(out, "############################\n   %-12s Sampling\n----------------------------\n", desc);
    std::fprintf(out, "Factor to cgs units: %g\n", fct);
    std::fprintf(out, "Initial value: %g\nFinal value: %g\n", i, fin);
    std::fprintf(out, "Spacing: %g\n", d);
  };
  const size_t nr = p->rad.size();
  const double raddelt = num(*p, "raddelt", -1);
  const bool kept = (p->a_rad.size() == 1 || raddelt == -1);
  head("Wavenumber", 1.0, p->wn_i, p->wn_f, p->wn_d);
  std::fprintf(out, "Oversample: %i\nNumber of elements: %lli\n", 1, (long long)p->nwn);
  head("Wavelength", 0.0, 0.0, 0.0, 0.0);
  std::fprintf(out, "Oversample: %i\nNumber of elements: %lli\n", 0, 0LL);
  head("Radius", p->out_rad_fct, p->rad.front(), p->rad.back(), kept ? 0.0 : raddelt);
  std::fprintf(out, "Number of elements: %lli\nValues: ", (long long)nr);
  for (size_t k = 0; k < nr; k++) std::fprintf(out, " %12.8g", p->rad[k]);
  std::fprintf(out, "\n");
  head("Impact parameter", p->out_rad_fct, p->rad.back(), p->rad.front(), kept ? 0.0 : -raddelt);
  std::fprintf(out, "Oversample: %i\nNumber of elements: %lli\nValues: ", kept ? 0 : 1, (long long)nr);
  for (size_t k = 0; k < nr; k++) std::fprintf(out, " %12.8g", p->rad[nr - 1 - k]);
  std::fprintf(out, "\n");
  if (out != stdout) std::fclose(out);
  return TRX_OK;
}

// every file a run of this problem writes, one "kind path" per line (what each output option
// turned into): spectrum, toomuch, intens, sample, dumps, detailtau/ext/cia
const char *trh_output_plan(trh_problem *p)
{
  if (!p) return "";
  std::string &o = p->err; o.clear();
  auto add = [&](const char *k, const std::string &v) { o += k; o += ' '; o += v; o += '\n'; };
  add("spectrum", str(*p, "outspec"));
  if (has(*p, "outtoomuch")) add("toomuch", str(*p, "outtoomuch"));
  if (has(*p, "outintens") && p->opts.solution == TRX_SOL_ECLIPSE) add("intens", str(*p, "outintens"));
  const bool dumps = str(*p, "savefiles").compare(0, 3, "yes") == 0;
  if (dumps) add("dumps", "tau.dat CIA.dat mol_extion.dat total_extion.dat cloud_extion.dat scatt_extion.dat");
  if (dumps && has(*p, "outsample")) add("sample", str(*p, "outsample"));
  if (!p->det_tau.wn.empty()) add("detailtau", p->det_tau.file + " " + std::to_string(p->det_tau.wn.size()));
  if (!p->det_ext.wn.empty()) add("detailext", p->det_ext.file + " " + std::to_string(p->det_ext.wn.size()));
  if (!p->det_cia.wn.empty()) add("detailcia", p->det_cia.file + " " + std::to_string(p->det_cia.wn.size()));
  if (has(*p, "opacityfile")) add("opacity", str(*p, "opacityfile"));
  if (has(*p, "saveext")) add("saveext", str(*p, "saveext"));
  return o.c_str();
}

// `savefiles yes` (tau.c:180-190, 311-335): the reference's dumps of the intermediates, in its
// formats and under its fixed file names in the working directory -- tau.dat (savetau,
// tau.c:491-515), CIA.dat (saveCIA, :420-446), mol_extion.dat (savemolExtion, :386-416).
// Inputs in the layouts of trx_debug: e, e_cs [layer][wn]; tau [wn][height].
// The reference leaves the rows of layers its lazy sweep never reached at zero in mol_extion.dat;
// pass `last` (trx_debug.last) to get the same: rows below the deepest ray are written as zeros.
int trh_write_dumps(const trh_problem *p, const double *e, const double *e_cs, const double *tau, const char *dir)
{ return trh_write_dumps_masked(p, e, e_cs, tau, nullptr, dir); }

int trh_write_dumps_masked(const trh_problem *p, const double *e, const double *e_cs, const double *tau, const int64_t *last,
                           const char *dir)
{
  if (!p) return TRX_E_ARG;
  const int64_t nr = (int64_t)p->rad.size(), nw = p->nwn;
  int64_t lowest = 0;                                       // lowest layer the reference would have swept
  if (last) { int64_t deep = 0; for (int64_t w = 0; w < nw; w++) deep = std::max(deep, last[w]); lowest = nr - 1 - deep; }
  const std::string base = dir ? std::string(dir) : p->base_dir;
  auto wn_at = [&](int64_t w) { return p->wn_i + (double)w * p->wn_d; };
  auto row = [&](FILE *f, const double *v, int64_t n, int64_t stride) {          // print1dArrayDouble, tau.c:361-367
    for (int64_t k = 0; k < n; k++) std::fprintf(f, "%-20.10g", v[k * stride]);
    std::fprintf(f, "\n");
  };
  if (tau) {
    FILE *f = std::fopen(join_path(base, "tau.dat").c_str(), "w");
    if (!f) return TRX_E_ARG;
    std::fprintf(f, "\n# 2D optical depth\n# tau [wn][rad]; wn[0]=min(wn); rad[0]=top (min(p))\n\n");
    for (int64_t w = 0; w < nw; w++) {                                            // print2dArrayDouble, tau.c:371-382
      std::fprintf(f, "wavenumber: %-20.10g\n", wn_at(w));
      row(f, tau + w * nr, nr, 1);
      std::fprintf(f, "\n");
    }
    std::fclose(f);
  }
  if (e_cs) {
    FILE *f = std::fopen(join_path(base, "CIA.dat").c_str(), "w");
    if (!f) return TRX_E_ARG;
    std::fprintf(f, "\n# 2D CIA extinction\n# e_cs [wn][rad]; wn[0]=min(wn); row[0]=bottom (max(p))\n\n");
    for (int64_t w = 0; w < nw; w++) {
      std::fprintf(f, "wavenumber: %-20.10g\n", wn_at(w));
      row(f, e_cs + w, nr, nw);
      std::fprintf(f, "\n");
    }
    std::fclose(f);
  }
  if (e) {
    FILE *f = std::fopen(join_path(base, "mol_extion.dat").c_str(), "w");
    if (!f) return TRX_E_ARG;
    std::fprintf(f, "\n# mol-line extinction\n# e [rad][wn]; rad[0]=bottom (max(p)); wn[0]=min(wn)\n\n");
    const std::vector<double> zeros((size_t)nw, 0.0);
    for (int64_t r = 0; r < nr; r++) {
      std::fprintf(f, "radius: %-20.10g\n", p->rad[(size_t)r]);
      row(f, r >= lowest ? e + r * nw : zeros.data(), nw, 1);
      std::fprintf(f, "\n");
    }
    std::fclose(f);
  }
  return TRX_OK;
}

// total_extion.dat, cloud_extion.dat, scatt_extion.dat (tau.c:180-190, 293-297, 456-470): per
// wavenumber the arrays er / e_c / e_s as they stand after that wavenumber's height loop.  The
// reference builds er from the molecular extinction swept SO FAR -- layers no earlier or current
// ray has needed still count as zero (tau.c:231-232 with unswept rows) -- and its eclipse
// geometry leaves the bottom-point parabola values in the layers the ray went through
// (eclipse.c:65-66).  Inputs in the trx_debug layouts ([layer][wn]); e must cover every layer a
// ray needed (a run with trx_opts.eager = 1 covers all).
int trh_write_ext_dumps(const trh_problem *p, const double *e, const double *e_cs, const int64_t *last, const double *er,
                        const double *e_scat, const double *e_cloud, const char *dir)
{
  if (!p || !e || !e_cs || !last || !er || !e_scat || !e_cloud) return TRX_E_ARG;
  const int64_t nr = (int64_t)p->rad.size(), nw = p->nwn;
  const std::string base = dir ? std::string(dir) : p->base_dir;
  auto wn_at = [&](int64_t w) { return p->wn_i + (double)w * p->wn_d; };
  struct Out { const char *name, *head; } outs[3] = {
    {"total_extion.dat", "# 2D total extinction\n# er [wn][rad]; wn[0]=min(wn), row[0]=bottom (max(p))\n"},
    {"cloud_extion.dat", "# 2D cloud extinction\n# e_c [wn][rad]; wn[0]=min(wn), row[0]=bottom (max(p))\n"},
    {"scatt_extion.dat", "# 2D scatt extinction\n# e_s [wn][rad]; wn[0]=min(wn), row[0]=bottom (max(p))\n"}};
  FILE *f[3];
  for (int k = 0; k < 3; k++) {
    f[k] = std::fopen(join_path(base, outs[k].name).c_str(), "w");
    if (!f[k]) { for (int j = 0; j < k; j++) std::fclose(f[j]); return TRX_E_ARG; }
    std::fprintf(f[k], "\n%s", outs[k].head);                // openFile, tau.c:473-480
  }
  const bool eclipse = p->opts.solution == TRX_SOL_ECLIPSE;
  int64_t lowest = nr - 1;                                  // lowest layer swept so far (the top one always is, tau.c:158-177)
  std::vector<double> tot((size_t)nr);
  for (int64_t w = 0; w < nw; w++) {
    const int64_t mine = nr - 1 - last[w];                  // lowest layer this ray went through
    lowest = std::min(lowest, mine);
    for (int64_t r = 0; r < nr; r++) {
      const size_t k = (size_t)(r * nw + w);
      if (eclipse && r >= mine) tot[(size_t)r] = er[k];     // as the ray solution left it
      else tot[(size_t)r] = (r >= lowest ? e[k] : 0.0) + e_scat[k] + e_cloud[k] + e_cs[k];     // tau.c:231-232
    }
    const double *src[3] = {tot.data(), nullptr, nullptr};
    for (int k = 0; k < 3; k++) {                           // save1Darray, tau.c:456-470
      std::fprintf(f[k], "\nwavenumber: %-20.10g\n", wn_at(w));
      for (int64_t r = 0; r < nr; r++)
        std::fprintf(f[k], "%-20.10g", k == 0 ? src[0][r] : (k == 1 ? e_cloud : e_scat)[(size_t)(r * nw + w)]);
      std::fprintf(f[k], "\n");
    }
  }
  for (int k = 0; k < 3; k++) std::fclose(f[k]);
  return TRX_OK;
}

}  // extern "C"
