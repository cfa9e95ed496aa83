// TEST-ONLY: compiles the product's device functions (c2-ray3dm1d_helium_amd/csrc/c2ray_device.hpp)
// with the host C++ compiler so that the CPU test-suite can run them against the golden vectors
// before the code ever reaches a GPU.  Nothing in the product links this file.
//   g++ -O2 -ffp-contract=off -mfma -fPIC -shared -o _host_harness.so host_harness.cpp
// (-mfma only so that __builtin_fma is one instruction; no expression is contracted)
#include <cstddef>
#include <cstring>
#include <vector>

#include "../c2-ray3dm1d_helium_amd/csrc/c2ray_device.hpp"
#include "../c2-ray3dm1d_helium_amd/csrc/c2ray_shell.hpp"

using namespace c2r;

namespace {
struct Tables {
  BandDataByRow bd; // (a BandData with its band-by-band copy filled: band_rows_fill)
  std::vector<double> pthick, pthin, hthick, hthin, cool;
  std::vector<double> hthick_il, hthin_il; // as the product's kernels read them (heat_interleave)
  double mintemp, dtemp;
};
Tables T;

void pitch(const double *src, int ncol, std::vector<double> &dst) {
  dst.assign((size_t)ncol * NTAUP, 0.0);
  for (int c = 0; c < ncol; c++) {
    std::memcpy(&dst[(size_t)c * NTAUP], src + (size_t)c * (NTAU + 1), sizeof(double) * (NTAU + 1));
    dst[(size_t)c * NTAUP + NTAU + 1] = src[(size_t)c * (NTAU + 1) + NTAU];
  }
}
// what c2r_set_tables does on the host: tau_zero of every band of SED `sed` from its four tables
void set_tau_zero(int sed, const std::vector<double> &pt, const std::vector<double> &pn, const std::vector<double> &ht,
                  const std::vector<double> &hn) {
  for (int b = 0; b < NFREQ; b++) {
    const double *cols[8];
    int n = 0;
    cols[n++] = &pt[(size_t)b * NTAUP];
    cols[n++] = &pn[(size_t)b * NTAUP];
    const int nh = b < NB1 ? 1 : (b < NB1 + NB2 ? 2 : 3);
    const int c0 = b < NB1 ? 0 : (b < NB1 + NB2 ? 2 * (b + 1) - NB1 - 2 : 3 * (b + 1) - NB2 - 2 * NB1 - 3);
    for (int k = 0; k < nh; k++) {
      cols[n++] = &ht[(size_t)(c0 + k) * NTAUP];
      cols[n++] = &hn[(size_t)(c0 + k) * NTAUP];
    }
    T.bd.tau_zero[sed][b] = band_tau_zero(cols, n);
  }
}
} // namespace

extern "C" {

void hh_set_tables(const double *pthick, const double *pthin, const double *hthick, const double *hthin,
                   const double *sHI, const double *sHeI, const double *sHeII, const double *const f[12],
                   int bb_upper, const double *cool, double mintemp, double dtemp) {
  std::memset(&T.bd, 0, sizeof T.bd);
  pitch(pthick, NFREQ, T.pthick);
  pitch(pthin, NFREQ, T.pthin);
  pitch(hthick, NHEAT, T.hthick);
  pitch(hthin, NHEAT, T.hthin);
  std::memcpy(T.bd.sigma_HI, sHI, sizeof T.bd.sigma_HI);
  std::memcpy(T.bd.sigma_HeI, sHeI, sizeof T.bd.sigma_HeI);
  std::memcpy(T.bd.sigma_HeII, sHeII, sizeof T.bd.sigma_HeII);
  double *dst[12] = {T.bd.f1ion_HI, T.bd.f1ion_HeI, T.bd.f1ion_HeII, T.bd.f2ion_HI, T.bd.f2ion_HeI, T.bd.f2ion_HeII,
                     T.bd.f1heat_HI, T.bd.f1heat_HeI, T.bd.f1heat_HeII, T.bd.f2heat_HI, T.bd.f2heat_HeI, T.bd.f2heat_HeII};
  for (int i = 0; i < 12; i++) std::memcpy(dst[i], f[i], sizeof(double) * (NFREQ - 1));
  T.bd.bb_upper = bb_upper;
  band_rows_fill(T.bd);
  set_tau_zero(0, T.pthick, T.pthin, T.hthick, T.hthin);
  T.hthick_il.resize(T.hthick.size());
  T.hthin_il.resize(T.hthin.size());
  heat_interleave(T.hthick.data(), T.hthick_il.data());
  heat_interleave(T.hthin.data(), T.hthin_il.data());
  for (int s = 1; s < 3; s++)
    for (int b = 0; b < NFREQ; b++) T.bd.tau_zero[s][b] = (double)INFINITY;
  T.cool.assign(cool, cool + 5 * NCOOL);
  T.mintemp = mintemp;
  T.dtemp = dtemp;
}

void hh_reccoef(double temperature, double *out12) {
  RecCoef rc;
  ini_rec_colion_factors(temperature, rc);
  std::memcpy(out12, &rc, sizeof rc);
}

// out5 = photo_HI, photo_HeI, photo_HeII, heat, photo_out
void hh_photoion(const double *cin6, double vol, double nflux, double i_state, int heat, double *out5) {
  PhotoOut o;
  Ricotti ric = {};
  if (heat) ric = ricotti_parameters(i_state);
  if (heat)
    photoion_rates<true>(T.bd, T.pthick.data(), T.pthin.data(), T.hthick_il.data(), T.hthin_il.data(), cin6[0], cin6[1],
                         cin6[2], cin6[3], cin6[4], cin6[5], vol, nflux, ric, o);
  else
    photoion_rates<false>(T.bd, T.pthick.data(), T.pthin.data(), T.hthick.data(), T.hthin.data(), cin6[0], cin6[1],
                          cin6[2], cin6[3], cin6[4], cin6[5], vol, nflux, ric, o);
  out5[0] = o.photo_HI; out5[1] = o.photo_HeI; out5[2] = o.photo_HeII; out5[3] = o.heat; out5[4] = o.photo_out;
}

// the three-SED variant: tables of SED 1 (pl) and 2 (qpl) in the same layout as the BB ones
namespace {
std::vector<double> S_pt[2], S_pn[2], S_ht[2], S_hn[2], S_ht_il[2], S_hn_il[2];
int S_lo[2] = {0, 0}, S_hi[2] = {0, 0};
}
void hh_set_sed(int sed, const double *pthick, const double *pthin, const double *hthick, const double *hthin, int lower,
                int upper) {
  const int k = sed - 1;
  pitch(pthick, NFREQ, S_pt[k]);
  pitch(pthin, NFREQ, S_pn[k]);
  pitch(hthick, NHEAT, S_ht[k]);
  pitch(hthin, NHEAT, S_hn[k]);
  S_lo[k] = lower - 1;
  S_hi[k] = upper;
  set_tau_zero(sed, S_pt[k], S_pn[k], S_ht[k], S_hn[k]);
  S_ht_il[k].resize(S_ht[k].size());
  S_hn_il[k].resize(S_hn[k].size());
  heat_interleave(S_ht[k].data(), S_ht_il[k].data());
  heat_interleave(S_hn[k].data(), S_hn_il[k].data());
}
static SedSet make_sedset() {
  SedSet ss;
  ss.photo_thick[0] = T.pthick.data(); ss.photo_thin[0] = T.pthin.data();
  ss.heat_thick[0] = T.hthick_il.data(); ss.heat_thin[0] = T.hthin_il.data();
  ss.lo[0] = 0; ss.hi[0] = T.bd.bb_upper;
  for (int k = 0; k < 2; k++) {
    ss.photo_thick[k + 1] = S_pt[k].data(); ss.photo_thin[k + 1] = S_pn[k].data();
    ss.heat_thick[k + 1] = S_ht_il[k].data(); ss.heat_thin[k + 1] = S_hn_il[k].data();
    ss.lo[k + 1] = S_lo[k]; ss.hi[k + 1] = S_hi[k];
  }
  return ss;
}
void hh_photoion_multi(const double *cin6, double vol, const double *nflux3, double i_state, int heat, double *out5) {
  PhotoOut o;
  const SedSet ss = make_sedset();
  Ricotti ric = {};
  if (heat) ric = ricotti_parameters(i_state);
  if (heat) photoion_rates_multi<true>(T.bd, ss, cin6[0], cin6[1], cin6[2], cin6[3], cin6[4], cin6[5], vol, nflux3, ric, o);
  else photoion_rates_multi<false>(T.bd, ss, cin6[0], cin6[1], cin6[2], cin6[3], cin6[4], cin6[5], vol, nflux3, ric, o);
  out5[0] = o.photo_HI; out5[1] = o.photo_HeI; out5[2] = o.photo_HeII; out5[3] = o.heat; out5[4] = o.photo_out;
}
// the same through BandDataByRow (what the three-SED heating kernel hands down: cross sections and factors read band by band)
void hh_photoion_multi_rows(const double *cin6, double vol, const double *nflux3, double i_state, int heat, double *out5) {
  PhotoOut o;
  const SedSet ss = make_sedset();
  const BandDataByRow &bdr = static_cast<const BandDataByRow &>(T.bd);
  Ricotti ric = {};
  if (heat) ric = ricotti_parameters(i_state);
  if (heat) photoion_rates_multi<true>(bdr, ss, cin6[0], cin6[1], cin6[2], cin6[3], cin6[4], cin6[5], vol, nflux3, ric, o);
  else photoion_rates_multi<false>(bdr, ss, cin6[0], cin6[1], cin6[2], cin6[3], cin6[4], cin6[5], vol, nflux3, ric, o);
  out5[0] = o.photo_HI; out5[1] = o.photo_HeI; out5[2] = o.photo_HeII; out5[3] = o.heat; out5[4] = o.photo_out;
}
double hh_photo_out_multi(const double *cin6, const double *nflux3) {
  const SedSet ss = make_sedset();
  return photo_out_multi(T.bd, ss, cin6[0], cin6[1], cin6[2], cin6[3], cin6[4], cin6[5], nflux3);
}

double hh_photo_out_only(const double *cin6, double nflux) {
  return photo_out_only(T.bd, T.pthick.data(), T.pthin.data(), cin6[0], cin6[1], cin6[2], cin6[3], cin6[4], cin6[5],
                        nflux);
}

void hh_doric(double dt, double de, double *ion15, const double *phi3, const double *fr4, const double *rc12,
              double clumping) {
  IonStates ion;
  RecCoef rc;
  std::memcpy(&ion, ion15, sizeof ion);
  std::memcpy(&rc, rc12, sizeof rc);
  doric(dt, de, ion, phi3[0], phi3[1], phi3[2], fr4[0], fr4[1], fr4[2], fr4[3], rc, clumping);
  std::memcpy(ion15, &ion, sizeof ion);
}

void hh_prepare_doric_factors(double NH, double NHe0, double NHe1, double *out4) {
  prepare_doric_factors(NH, NHe0, NHe1, out4[0], out4[1], out4[2], out4[3]);
}

void hh_thermal(double dt, double *tend, double *tavg, double de, double nd, const double *ion15, double heat,
                double zred, double H0, double Omega0) {
  IonStates ion;
  std::memcpy(&ion, ion15, sizeof ion);
  CoolData cd{T.cool.data(), T.mintemp, T.dtemp, zred, H0, Omega0, nullptr};
  thermal(cd, dt, *tend, *tavg, de, nd, ion, heat);
}

// cinterp (column_density.f90:28-345) assembled from short_characteristic + interp_column the way
// k_sweep_shell does it; pos/srcpos are 1-based, pos may lie outside the mesh.
void hh_cinterp(const int *mesh, const double *cH, const double *cHe, const int *pos, const int *src, double *out4) {
  const int n1 = mesh[0], n2 = mesh[1], n3 = mesh[2];
  const size_t nc = (size_t)n1 * n2 * n3;
  ShortChar s4;
  short_characteristic(src[0], src[1], src[2], pos[0] - src[0], pos[1] - src[1], pos[2] - src[2], s4);
  size_t qc[4];
  for (int c = 0; c < 4; c++) {
    int i = ((src[0] - 1 + s4.ci[c]) % n1 + n1) % n1, j = ((src[1] - 1 + s4.cj[c]) % n2 + n2) % n2,
        k = ((src[2] - 1 + s4.ck[c]) % n3 + n3) % n3;
    qc[c] = (size_t)i + (size_t)n1 * ((size_t)j + (size_t)n2 * k);
  }
  out4[0] = interp_column(s4, cH[qc[0]], cH[qc[1]], cH[qc[2]], cH[qc[3]], sigma_HI_at_ion_freq);
  out4[1] = interp_column(s4, cHe[qc[0]], cHe[qc[1]], cHe[qc[2]], cHe[qc[3]], sigma_HeI_at_ion_freq);
  out4[2] = interp_column(s4, cHe[nc + qc[0]], cHe[nc + qc[1]], cHe[nc + qc[2]], cHe[nc + qc[3]], sigma_HeII_at_ion_freq);
  out4[3] = s4.path;
}

// the restated libm (csrc/c2ray_math.hpp) on arrays: op 0 exp, 1 log10, 2 pow
void hh_math(int op, int n, const double *x, const double *y, double *out) {
  for (int i = 0; i < n; i++)
    out[i] = op == 0 ? C2R_MATH_EXP(x[i])
                     : (op == 1 ? C2R_MATH_LOG10(x[i]) : (op == 3 ? C2R_MATH_LOG10P(x[i]) : C2R_MATH_POW(x[i], y[i])));
}
// the platform libm the reference links (glibc): same ops
void hh_libm(int op, int n, const double *x, const double *y, double *out) {
  for (int i = 0; i < n; i++) out[i] = op == 0 ? exp(x[i]) : ((op == 1 || op == 3) ? log10(x[i]) : pow(x[i], y[i]));
}

// div_recip(a, make_recip(b)) on arrays (must equal a/b bit for bit)
void hh_div_recip(int n, const double *a, const double *b, double *out) {
  for (int i = 0; i < n; i++) out[i] = div_recip(a[i], make_recip(b[i]));
}

int hh_constants(double *out, int n) {
  const double c[] = {pi, abu_he, abu_c, (1.0 - abu_he) + 4.0 * abu_he, gamma1, hplanck, k_B, 1.672661e-24, temph0,
                      temphe0, temphe1, colh0, colhe0, colhe1, ev2k, ev2fr, eth0, ethe0, ethe1, sigma_HI_at_ion_freq,
                      sigma_HeI_at_ion_freq, sigma_HeII_at_ion_freq, ion_freq_HI, ion_freq_HeI, ev2fr * ethe1,
                      sigma_H_heth, sigma_H_heLya, sigma_He_heLya, sigma_He_he2, sigma_H_he2, epsilon,
                      convergence_fraction, minimum_fractional_change, minimum_fraction_of_atoms, minitemp,
                      relative_denergy, minlogtau, dlogtau};
  const int m = (int)(sizeof c / sizeof c[0]);
  for (int i = 0; i < m && i < n; i++) out[i] = c[i];
  return m;
}

// The per-shell geometry of k_sweep_shell_fast (csrc/c2ray_shell.hpp) against the general functions it replaces, for
// every cell of the shells 2..smax around a source at (i0,j0,k0): the thread -> cell map with its magic divisions,
// the bilinear weights and the path length (bit for bit), and the corner positions -- equal to the general inverse
// map wherever the corner's weight is not exactly zero, and inside shell s-1 where it is.  Returns the number of
// mismatches (first one described in what[0..7]: shell, t, kind, corner).
int hh_check_shell_geometry(int smin, int smax, int i0, int j0, int k0, int *what) {
  int bad = 0;
  auto note = [&](int s, int t, int kind, int c) {
    if (bad++ == 0 && what) { what[0] = s; what[1] = t; what[2] = kind; what[3] = c; }
  };
  for (int s = smin; s <= smax; s++) {
    const ShellGeom G = shell_geometry(s);
    if (G.alam * (double)s != (double)s - 0.5) note(s, -1, 0, 0); // the weight-zero argument rests on this
    const int cnt = (int)shell_count(s);
    for (int t = 0; t < cnt; t++) {
      int di, dj, dk, fi, fj, fk;
      shell_decode(s, t, di, dj, dk);
      const int face = shell_decode_fast(G, t, fi, fj, fk);
      if (fi != di || fj != dj || fk != dk) { note(s, t, 1, 0); continue; }
      const int ia = di < 0 ? -di : di, ja = dj < 0 ? -dj : dj, ka = dk < 0 ? -dk : dk;
      const int want_face = ka == s ? 0 : (ja == s ? 1 : 2);
      if (face != want_face) { note(s, t, 2, 0); continue; }
      if ((long long)shell_position(di, dj, dk) != G.off + t) note(s, t, 3, 0);
      ShortChar ref;
      short_characteristic(i0, j0, k0, di, dj, dk, ref);
      ShellCorners got;
      shell_short_characteristic(G, face, i0, j0, k0, di, dj, dk, got);
      if (ref.diag != 1.0) note(s, t, 4, 0);
      if (std::memcmp(&ref.path, &got.path, sizeof(double)) != 0) note(s, t, 5, 0);
      for (int c = 0; c < 4; c++) {
        if (std::memcmp(&ref.s[c], &got.s[c], sizeof(double)) != 0) note(s, t, 6, c);
        const long long pref = (long long)shell_position(ref.ci[c], ref.cj[c], ref.ck[c]);
        if (ref.s[c] != 0.0) {
          if (pref != got.p[c]) note(s, t, 7, c);
        } else if (got.p[c] < G.offp || got.p[c] >= G.off) {
          note(s, t, 8, c);
        }
      }
    }
  }
  return bad;
}

// log10 of positive normal numbers through the table path of the restated __log_fma, with the plain (invc, logc)
// table and with the table that has the power of two folded in (gm::LogEntry, what k_rates keeps in LDS): the
// arguments outside the near-1 interval must give the same bits.  Returns the number of mismatches.
int hh_check_log_table4(int n, const double *x) {
  static gm::LogEntry tab4[256];
  for (int E = 0; E < 256; E++) tab4[E] = gm::make_log_entry(E);
  int bad = 0;
  for (int i = 0; i < n; i++) {
    const gm::Log10Arg a = gm::log10_split(x[i]);
    if (gm::log10_near1(a)) continue;
    const double u = gm::log_table_path(a, gm::log_table()), v = gm::log_table_path4(a, tab4);
    if (std::memcmp(&u, &v, sizeof u) != 0) bad++;
    const double f = gm::log10_finish(a, v), g = log10(x[i]);
    if (std::memcmp(&f, &g, sizeof f) != 0) bad++;
  }
  return bad;
}
}
