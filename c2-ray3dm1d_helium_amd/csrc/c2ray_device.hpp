// Per-cell physics of the C2-Ray hot path as device functions for gfx950.
//
// Everything here is fp64 and is written so that, with -ffp-contract=off, each function
// performs the same IEEE operations in the same order as the routine it replaces (cited
// per function, paths relative to the reference's code/ directory).  The numerics rule
// of the reference matters: Fortran literals without _dp are REAL(4) and dp parameters
// initialised from them carry the float-rounded value; C2R_F(x) reproduces that.
//
// The file is also compilable by a host C++ compiler (C2R_HD expands to nothing) so that
// tests/ can run the very same functions on the CPU against the golden vectors; the
// product library never does that.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define C2R_HD __host__ __device__ __forceinline__
#else
#define C2R_HD inline
#endif

// Transcendental functions: by default the bit-exact restatement of the libm the reference links
// (csrc/c2ray_math.hpp); -DC2R_USE_PLATFORM_LIBM selects the platform's exp/log10/pow instead
// (ocml on the device), which differ from the reference's in the last bit in 1-23 % of calls.
#if !defined(C2R_USE_PLATFORM_LIBM)
#include "c2ray_math.hpp"
#else
#define C2R_MATH_EXP(x) exp(x)
#define C2R_MATH_LOG10(x) log10(x)
#define C2R_MATH_LOG10P(x) log10(x)
#define C2R_MATH_LOG10PT(x, tab) log10(x)
#define C2R_MATH_LOG10N(x) log10(x)
#define C2R_MATH_POW(x, y) pow(x, y)
#endif

#define C2R_F(x) ((double)(x##f))
#define C2R_FX(x) C2R_F(x) // the same for a literal that arrives through a macro (expanded before the suffix is pasted)

// The numerical parameters of the reference's c2ray_parameters.f90 and abundances.f90 that are compiled into the device
// code.  A host built with another c2ray_parameters*.f90 rebuilds the library with the same values:
//     _build.build(params={"subboxsize": 5, "convergence_fraction": "1.0e-4"})    or   C2R_PARAMS="subboxsize=5,..."
// (-DC2R_PARAM_<NAME>=<the Fortran literal without kind suffix>; REAL(4) parameters are rounded to single first, as the
// reference's bare literals are).  c2r_get_constants reports what a library was built with, and the Fortran shim refuses
// a library whose values differ from its modules' (evolve_data.F90: check_compiled_constants).
#ifndef C2R_PARAM_SUBBOXSIZE
#define C2R_PARAM_SUBBOXSIZE 10                  /* integer, c2ray_parameters.f90:51 */
#endif
#ifndef C2R_PARAM_MAX_SUBBOX
#define C2R_PARAM_MAX_SUBBOX 1150                /* integer, :56 */
#endif
#ifndef C2R_PARAM_EPSILON
#define C2R_PARAM_EPSILON 1.0e-20                /* real(dp), :32 */
#endif
#ifndef C2R_PARAM_CONVERGENCE_FRACTION
#define C2R_PARAM_CONVERGENCE_FRACTION 2.5e-4    /* real(4) literal, :26 */
#endif
#ifndef C2R_PARAM_MINIMUM_FRACTIONAL_CHANGE
#define C2R_PARAM_MINIMUM_FRACTIONAL_CHANGE 1.0e-2 /* :36 */
#endif
#ifndef C2R_PARAM_MINIMUM_FRACTION_OF_ATOMS
#define C2R_PARAM_MINIMUM_FRACTION_OF_ATOMS 1.0e-8 /* :44 */
#endif
#ifndef C2R_PARAM_MINITEMP
#define C2R_PARAM_MINITEMP 1.0                   /* :87 */
#endif
#ifndef C2R_PARAM_RELATIVE_DENERGY
#define C2R_PARAM_RELATIVE_DENERGY 0.1           /* :89 */
#endif
#ifndef C2R_PARAM_ABU_HE
#define C2R_PARAM_ABU_HE 0.074                   /* abundances.f90:23 */
#endif
#ifndef C2R_PARAM_ABU_C
#define C2R_PARAM_ABU_C 7.1e-7                   /* abundances.f90:26 */
#endif

#if !defined(C2R_USE_PLATFORM_LIBM)
#define C2R_LOGTAB_DEFAULT ::c2r::gm::log_table()
#else
#define C2R_LOGTAB_DEFAULT nullptr
#endif

namespace c2r {

constexpr int NFREQ = 47;    // radiation_sizes.f90:22
constexpr int NHEAT = 113;   // radiation_sizes.f90:23
constexpr int NTAU = 2000;   // radiation_sizes.f90:18
constexpr int NTAUP = 2002;  // device column pitch: rows 0..2000 plus one duplicate of row 2000
constexpr int NCOOL = 801;   // cooling_h.f90:25
constexpr int NB1 = 1, NB2 = 26, NB3 = 20;
constexpr int SUBBOXSIZE = C2R_PARAM_SUBBOXSIZE;   // c2ray_parameters.f90:51
constexpr int MAX_SUBBOX = C2R_PARAM_MAX_SUBBOX; // c2ray_parameters.f90:56

// mathconstants.f90:21, abundances.f90:23-29
constexpr double pi = C2R_F(3.141592654);
constexpr double abu_he = C2R_FX(C2R_PARAM_ABU_HE);
constexpr double abu_c = C2R_FX(C2R_PARAM_ABU_C);
// cgsconstants.f90:26-103
constexpr double hplanck = 6.6260755e-27;
constexpr double k_B = 1.381e-16;
constexpr double eth0 = C2R_F(13.598);
constexpr double ethe0 = C2R_F(24.587);
constexpr double ethe1 = C2R_F(54.416);
constexpr double ev2fr = C2R_F(0.241838e15);
constexpr double ev2k = (double)(1.0f / 8.617e-05f);
constexpr double temph0 = eth0 * ev2k;
constexpr double temphe0 = ethe0 * ev2k;
constexpr double temphe1 = ethe1 * ev2k;
constexpr double colh0 = C2R_F(1.3e-8) * C2R_F(0.83) * C2R_F(1.0) / (eth0 * eth0);
constexpr double colhe0 = C2R_F(1.3e-8) * C2R_F(0.63) * C2R_F(2.0) / (ethe0 * ethe0);
constexpr double colhe1 = C2R_F(1.3e-8) * C2R_F(1.30) * C2R_F(1.0) / (ethe1 * ethe1);
constexpr double gamma1 = 5.0 / 3.0 - 1.0;
constexpr double ion_freq_HI = ev2fr * eth0;
constexpr double ion_freq_HeI = ev2fr * ethe0;
// c2ray_parameters.f90:26-89
constexpr double epsilon = C2R_PARAM_EPSILON;
constexpr double convergence_fraction = C2R_FX(C2R_PARAM_CONVERGENCE_FRACTION);
constexpr double minimum_fractional_change = C2R_FX(C2R_PARAM_MINIMUM_FRACTIONAL_CHANGE);
constexpr double minimum_fraction_of_atoms = C2R_FX(C2R_PARAM_MINIMUM_FRACTION_OF_ATOMS);
constexpr double minitemp = C2R_FX(C2R_PARAM_MINITEMP);
constexpr double relative_denergy = C2R_FX(C2R_PARAM_RELATIVE_DENERGY);
// cgsphotoconstants.f90:25-50
constexpr double sigma_HI_at_ion_freq = C2R_F(6.346e-18);
constexpr double sigma_HeI_at_ion_freq = C2R_F(7.430e-18);
constexpr double sigma_HeII_at_ion_freq = C2R_F(1.589e-18);
constexpr double sigma_H_heth = 1.238e-18;
constexpr double sigma_H_heLya = 9.907e-22;
constexpr double sigma_He_heLya = 1.301e-20;
constexpr double sigma_He_he2 = 1.690780687052975e-18;
constexpr double sigma_H_he2 = 1.230695924714239e-19;
// radiation_tables.f90:59-61
constexpr double minlogtau = -20.0;
constexpr double dlogtau = (4.0 - (-20.0)) / (double)(float)NTAU;
// evolve_point.F90:91, radiation_photoionrates.f90:342,482
constexpr double max_coldensh = C2R_F(2e29);
constexpr double tau_photo_limit = C2R_F(1.0e-7);
constexpr double tau_heat_limit = C2R_F(1.0e-4);

// module-global recombination / collisional coefficients (cgsconstants.f90:106-133)
struct RecCoef {
  double arech0, brech0, areche0, breche0, oreche0, areche1, breche1, treche1;
  double colli_HI, colli_HeI, colli_HeII, v;
};

// type ionstates (files_for_3D/mat_ini_test.F90:70-77)
struct IonStates {
  double h[2], he[3], h_av[2], he_av[3], h_old[2], he_old[3];
};

// per-band vectors and table pointers the kernels read (uniform data)
struct BandData {
  double sigma_HI[NFREQ], sigma_HeI[NFREQ], sigma_HeII[NFREQ];
  // secondary ionisation fractions, index band-2 (bands 2..47); radiation_sizes.f90:198-370
  double f1ion_HI[NFREQ - 1], f1ion_HeI[NFREQ - 1], f1ion_HeII[NFREQ - 1];
  double f2ion_HI[NFREQ - 1], f2ion_HeI[NFREQ - 1], f2ion_HeII[NFREQ - 1];
  double f1heat_HI[NFREQ - 1], f1heat_HeI[NFREQ - 1], f1heat_HeII[NFREQ - 1];
  double f2heat_HI[NFREQ - 1], f2heat_HeI[NFREQ - 1], f2heat_HeII[NFREQ - 1];
  int bb_upper;
  // per SED (black body, power law, quasar) and band: the optical depth from which every table entry a lookup
  // of that band can touch is exactly 0 (band_tau_zero); +inf when there is no such depth
  double tau_zero[3][NFREQ];
  // The same cross sections and factors once more, band by band (band_rows_fill): what one band of the rates kernels
  // reads lies in 128 consecutive bytes -- two lines of the scalar cache and a few wide scalar loads instead of fifteen
  // loads from fifteen lines.  f: f1ion_{HI,HeI,HeII}, f2ion_.., f1heat_.., f2heat_.. of band b (0 for band 1, whose
  // arrays start at band 2).
  struct Row {
    double sigma[3], pad;
    double f[12];
  } rows[NFREQ];
};
// fills BandData::rows from the arrays above (host side, before the struct goes to the device)
inline void band_rows_fill(BandData &bd) {
  for (int b = 0; b < NFREQ; b++) {
    BandData::Row &r = bd.rows[b];
    r.sigma[0] = bd.sigma_HI[b]; r.sigma[1] = bd.sigma_HeI[b]; r.sigma[2] = bd.sigma_HeII[b];
    r.pad = 0.0;
    const double *src[12] = {bd.f1ion_HI, bd.f1ion_HeI, bd.f1ion_HeII, bd.f2ion_HI, bd.f2ion_HeI, bd.f2ion_HeII,
                             bd.f1heat_HI, bd.f1heat_HeI, bd.f1heat_HeII, bd.f2heat_HI, bd.f2heat_HeI, bd.f2heat_HeII};
    for (int k = 0; k < 12; k++) r.f[k] = b >= 1 ? src[k][b - 1] : 0.0;
  }
}
// A kernel says which of the two copies its band loops read by the TYPE it hands down: BandData (the per-species
// arrays) or BandDataByRow (the same object, read band by band).  Measured on one box: the rows are worth 4 % to the
// heating kernel with three SEDs (355 -> 341 ms per 128-source pass: fifteen scalar loads per band and SED become
// four wide ones) and cost the isothermal and the one-SED heating kernels 0.5 % (wide loads want aligned blocks of
// scalar registers, which those kernels are short of: 24 instead of 12 spilled), so only the former asks for them.
struct BandDataByRow : BandData {};
C2R_HD double band_sigma(const BandData &bd, int b, int k) { return k == 0 ? bd.sigma_HI[b] : (k == 1 ? bd.sigma_HeI[b] : bd.sigma_HeII[b]); }
C2R_HD double band_sigma(const BandDataByRow &bd, int b, int k) { return bd.rows[b].sigma[k]; }
// factor n = 0..11 of band b >= 1: f1ion_{HI,HeI,HeII}, f2ion_.., f1heat_.., f2heat_..
C2R_HD double band_f(const BandData &bd, int b, int n) {
  const double *const v[12] = {bd.f1ion_HI, bd.f1ion_HeI, bd.f1ion_HeII, bd.f2ion_HI, bd.f2ion_HeI, bd.f2ion_HeII,
                               bd.f1heat_HI, bd.f1heat_HeI, bd.f1heat_HeII, bd.f2heat_HI, bd.f2heat_HeI, bd.f2heat_HeII};
  return v[n][b - 1]; // the arrays are dimension(2:47)
}
C2R_HD double band_f(const BandDataByRow &bd, int b, int n) { return bd.rows[b].f[n]; }

C2R_HD double dmax(double a, double b) { return a > b ? a : b; }
C2R_HD double dmin(double a, double b) { return a < b ? a : b; }

// ----------------------------------------------------------------------------------------
// Correctly rounded a/b for a divisor used many times (the shell volume `vol`, the table step
// `dlogtau`): with y = RN(1/b), q = RN(a*y), r = a - b*q (exact, one fma), the result
// RN(q + r*y) IS RN(a/b) -- Markstein's theorem (P. Markstein, IBM J. Res. Dev. 34, 1990; Muller et
// al., Handbook of Floating-Point Arithmetic, ch. "division with an fma") -- provided the
// significand of b is not all ones and nothing under/overflows.  Both provisos are checked, with
// the plain division as the fall-back, so the value is identical to the reference's `a/b` always.
// tests/test_math_host.py checks 2e7 random operands; 8e8 were run once (0 mismatches).
// 3 FP64 instructions instead of the ~12 (one quarter-rate) of the gfx950 division sequence.
struct Recip {
  double b, y;
  bool ok;     // div_recip's proviso on b
  bool ok_big; // ok, and b >= 1: then a quotient's own magnitude tells whether anything underflowed (div_by_vol)
};
C2R_HD Recip make_recip(double b) {
  Recip R;
  R.b = b;
  R.y = 1.0 / b;
  const uint64_t u = __builtin_bit_cast(uint64_t, b);
  const int e = (int)((u >> 52) & 0x7ff);
  R.ok = (u & 0x000FFFFFFFFFFFFFULL) != 0x000FFFFFFFFFFFFFULL && e > 1023 - 400 && e < 1023 + 400;
  R.ok_big = R.ok && e >= 1023 && (u >> 63) == 0;
  return R;
}
C2R_HD double div_recip(double a, const Recip &R) {
  const double aa = fabs(a);
  if (R.ok && ((aa > 0x1p-500 && aa < 0x1p500) || a == 0.0)) {
    const double q = a * R.y;
    const double r = __builtin_fma(-R.b, q, a);
    return __builtin_fma(r, R.y, q);
  }
  return a / R.b;
}

// ----------------------------------------------------------------------------------------
// cgsconstants.f90:140-266  ini_rec_colion_factors
C2R_HD void ini_rec_colion_factors(double T, RecCoef &rc) {
  double lambda = 2.0 * (temph0 / T);
  rc.arech0 = C2R_F(1.269e-13) * C2R_MATH_POW(lambda, 1.503) /
              C2R_MATH_POW(1.0 + C2R_MATH_POW(lambda / C2R_F(0.522), C2R_F(0.470)), C2R_F(1.923));
  rc.brech0 = C2R_F(2.753e-14) * C2R_MATH_POW(lambda, 1.500) /
              C2R_MATH_POW(1.0 + C2R_MATH_POW(lambda / C2R_F(2.740), C2R_F(0.407)), C2R_F(2.242));
  if (T < 9.e3) {
    lambda = 2.0 * (temph0 / T);
    rc.areche0 = 1.269e-13 * C2R_MATH_POW(lambda, 1.503) /
                 C2R_MATH_POW(1.0 + C2R_MATH_POW(lambda / C2R_F(0.522), C2R_F(0.470)), C2R_F(1.923));
    rc.breche0 = 2.753e-14 * C2R_MATH_POW(lambda, 1.500) /
                 C2R_MATH_POW(1.0 + C2R_MATH_POW(lambda / C2R_F(2.740), C2R_F(0.407)), C2R_F(2.242));
  } else {
    lambda = 2.0 * (temphe0 / T);
    double dielectronic =
        1.9e-3 * C2R_MATH_POW(T, -1.5) * C2R_MATH_EXP(-4.7e5 / T) * (1.0 + 0.3 * C2R_MATH_EXP(-9.4e4 / T));
    rc.areche0 = 3.000e-14 * C2R_MATH_POW(lambda, 0.654) + dielectronic;
    // the flang -O2 build of the reference evaluates x**0.750 as sqrt(x)*sqrt(sqrt(x))
    rc.breche0 = 1.260e-14 * (sqrt(lambda) * sqrt(sqrt(lambda))) + dielectronic;
  }
  rc.oreche0 = rc.areche0 - rc.breche0;
  lambda = 2.0 * (temphe1 / T);
  rc.breche1 = 5.5060e-14 * C2R_MATH_POW(lambda, 1.5) /
               C2R_MATH_POW(1.0 + C2R_MATH_POW(lambda / 2.740, 0.407), 2.242);
  rc.areche1 = C2R_F(2.538e-13) * C2R_MATH_POW(lambda, 1.503) /
               C2R_MATH_POW(1.0 + C2R_MATH_POW(lambda / 0.522, 0.470), 1.923);
  rc.treche1 = 3.4e-13 * C2R_MATH_POW(T / 1.0e4, -0.6);
  rc.v = 0.285 * C2R_MATH_POW(T / 1.0e4, 0.119);
  double sqrtt0 = sqrt(T);
  rc.colli_HI = colh0 * sqrtt0 * C2R_MATH_EXP(-temph0 / T);
  rc.colli_HeI = colhe0 * sqrtt0 * C2R_MATH_EXP(-temphe0 / T);
  rc.colli_HeII = colhe1 * sqrtt0 * C2R_MATH_EXP(-temphe1 / T);
}

// tped.f90:41-84
C2R_HD double temper2pressr(double temper, double ndens, double eldens) { return (ndens + eldens) * k_B * temper; }
C2R_HD double pressr2temper(double pressr, double ndens, double eldens) { return pressr / (k_B * (ndens + eldens)); }
C2R_HD double electrondens(double ndens, const double *xh, const double *xhe) {
  return ndens * (xh[1] * (1.0 - abu_he) + abu_c + abu_he * (xhe[1] + 2.0 * xhe[2]));
}
// doric.f90:358-372
C2R_HD double coldens(double path, double neufrac, double ndens, double abundance) {
  return neufrac * ndens * path * abundance;
}

// doric.f90:317-351
C2R_HD void prepare_doric_factors(double NH, double NHe0, double NHe1, double &yfrac, double &zfrac,
                                  double &y2afrac, double &y2bfrac) {
  double tau_H_heth = NH * sigma_H_heth;
  double tau_He_heth = NHe0 * sigma_HeI_at_ion_freq;
  double tau_H_heLya = NH * sigma_H_heLya;
  double tau_He_heLya = NHe0 * sigma_He_heLya;
  double tau_H_he2th = NH * sigma_H_he2;
  double tau_He_he2th = NHe0 * sigma_He_he2;
  double tau_He2_he2th = NHe1 * sigma_HeII_at_ion_freq;
  yfrac = tau_H_heth / (tau_H_heth + tau_He_heth);
  zfrac = tau_H_heLya / (tau_H_heLya + tau_He_heLya);
  y2afrac = tau_He2_he2th / (tau_He2_he2th + tau_He_he2th + tau_H_he2th);
  y2bfrac = tau_He_he2th / (tau_He2_he2th + tau_He_he2th + tau_H_he2th);
}

// doric.f90:35-313
C2R_HD void doric(double dt, double rhe, IonStates &ion, double phi_HI, double phi_HeI, double phi_HeII,
                  double yfrac, double zfrac, double y2afrac, double y2bfrac, const RecCoef &rc,
                  double clumping) {
  const double v = rc.v;
  const double pfrac = 0.96;
  const double heliumfraction = abu_he / (1.0 - abu_he);
  double ffrac = dmax(dmin(10.0 * ion.h[0], 1.0), 0.01);
  double wfrac = (1.425 - 0.737) + 0.737 * yfrac;

  double alpha_h_B = clumping * rc.brech0;
  double alpha_he_1 = clumping * rc.oreche0;
  double alpha_he_B = clumping * rc.breche0;
  double alpha_he_A = clumping * rc.areche0;
  double alpha_he2_B = clumping * rc.breche1;
  double alpha_he2_A = clumping * rc.areche1;
  double alpha_he2_2 = clumping * rc.treche1;
  double alpha_he2_1 = alpha_he2_A - alpha_he2_B;

  double aih0 = dmax(phi_HI + rhe * rc.colli_HI, 1.0e-200);
  double aihe0 = dmax(phi_HeI + rhe * rc.colli_HeI, 1.0e-200);
  double aihe1 = dmax(phi_HeII + rhe * rc.colli_HeII, 1.0e-200);

  double Lmat = -(aih0 + rhe * alpha_h_B);
  double Mmat = (yfrac * rhe * alpha_he_1 + pfrac * rhe * alpha_he_B) * heliumfraction;
  double Nmat = ((ffrac * zfrac * (1.0 - v) + v * wfrac) * alpha_he2_B + alpha_he2_2 +
                 (1.0 - y2afrac - y2bfrac) * alpha_he2_1) * heliumfraction * rhe;
  double Pmat = -aihe0 - aihe1 - rhe * (alpha_he_A - (1.0 - yfrac) * alpha_he_1);
  double Emat = -rhe * (alpha_he2_A - y2afrac * alpha_he2_1);
  double Qmat = -aihe0 + rhe * alpha_he2_B * (ffrac * (1.0 - zfrac) * (1.0 - v) + v * (1.425 - wfrac)) -
                Emat + alpha_he2_1 * y2bfrac * rhe;

  double Bcoef = Emat - Pmat;
  double Scoef = sqrt(Bcoef * Bcoef + 4.0 * aihe1 * Qmat);
  double QHEPcoef = 1.0 / (Qmat * aihe1 - Emat * Pmat);
  double BminusS = Bcoef - Scoef;
  double BplusS = Bcoef + Scoef;

  double lambda1 = Lmat;
  double lambda2 = 0.5 * (Emat + Pmat - Scoef);
  double lambda3 = 0.5 * (Emat + Pmat + Scoef);

  double rx = -1.0 / Lmat * (aih0 + (Mmat * Emat - Nmat * aihe1) * (aihe0 * QHEPcoef));
  double ry = aihe0 * (Emat * QHEPcoef);
  double rz = -aihe0 * (aihe1 * QHEPcoef);

  double twoaihe1 = 2.0 * aihe1;
  double eigv2x = -Nmat / (Lmat - lambda2) + (Mmat / twoaihe1) * BplusS / (Lmat - lambda2);
  double eigv3x = (-twoaihe1 * Nmat + Mmat * (BminusS)) / (twoaihe1 * (Lmat - lambda3));
  double eigv2y = (-BplusS) / (twoaihe1);
  double eigv3y = (-BminusS) / (twoaihe1);

  double Rcoef = twoaihe1 * (ry - ion.he_old[1]);
  double Tcoef = rz - ion.he_old[2];

  double coef2 = (Rcoef + (BminusS)*Tcoef) / (2.0 * Scoef);
  double coef3 = -(Rcoef + (BplusS)*Tcoef) / (2.0 * Scoef);
  double coef1 = -rx + (eigv3x - eigv2x) * (Rcoef / (2.0 * Scoef)) +
                 Tcoef * ((BplusS * eigv3x / (2.0 * Scoef) - BminusS * eigv2x / (2.0 * Scoef))) +
                 ion.h_old[1];

  double lam1dt = dt * lambda1, lam2dt = dt * lambda2, lam3dt = dt * lambda3;
  double elam1dt = C2R_MATH_EXP(lam1dt), elam2dt = C2R_MATH_EXP(lam2dt), elam3dt = C2R_MATH_EXP(lam3dt);

  ion.h[1] = coef1 * elam1dt + coef2 * elam2dt * eigv2x + coef3 * elam3dt * eigv3x + rx;
  ion.he[1] = coef2 * elam2dt * eigv2y + coef3 * elam3dt * eigv3y + ry;
  ion.he[2] = coef2 * elam2dt + coef3 * elam3dt + rz;
  ion.h[0] = 1.0 - ion.h[1];
  ion.he[0] = 1.0 - ion.he[1] - ion.he[2];

  if (ion.h[0] < epsilon) { ion.h[0] = epsilon; ion.h[1] = 1.0 - epsilon; }
  if (ion.h[1] < epsilon) { ion.h[1] = epsilon; ion.h[0] = 1.0 - epsilon; }
  if (ion.he[0] <= epsilon || ion.he[1] <= epsilon || ion.he[2] <= epsilon) {
    if (ion.he[0] < epsilon) ion.he[0] = epsilon;
    if (ion.he[1] < epsilon) ion.he[1] = epsilon;
    if (ion.he[2] < epsilon) ion.he[2] = epsilon;
    double normfac = ion.he[0] + ion.he[1] + ion.he[2];
    ion.he[0] = ion.he[0] / normfac;
    ion.he[1] = ion.he[1] / normfac;
    ion.he[2] = ion.he[2] / normfac;
  }

  const double small = C2R_F(1.0e-8);
  double avg_factor_1, avg_factor_2, avg_factor_3;
  if (fabs(lam1dt) < small) avg_factor_1 = coef1; else avg_factor_1 = coef1 * (elam1dt - 1.0) / lam1dt;
  if (fabs(lam2dt) < small) avg_factor_2 = coef2; else avg_factor_2 = coef2 * (elam2dt - 1.0) / lam2dt;
  if (fabs(lam3dt) < small) avg_factor_3 = coef3; else avg_factor_3 = coef3 * (elam3dt - 1.0) / lam3dt;

  ion.h_av[1] = rx + avg_factor_1 + eigv2x * avg_factor_2 + eigv3x * avg_factor_3;
  ion.he_av[1] = ry + eigv2y * avg_factor_2 + eigv3y * avg_factor_3;
  ion.he_av[2] = rz + avg_factor_2 + avg_factor_3;
  ion.h_av[0] = 1.0 - ion.h_av[1];
  ion.he_av[0] = 1.0 - ion.he_av[1] - ion.he_av[2];

  if (ion.h_av[1] < epsilon) { ion.h_av[1] = epsilon; ion.h_av[0] = 1.0 - epsilon; }
  if (ion.h_av[0] < epsilon) { ion.h_av[0] = epsilon; ion.h_av[1] = 1.0 - epsilon; }
  if (ion.he_av[0] <= epsilon || ion.he_av[1] <= epsilon || ion.he_av[2] <= epsilon) {
    if (ion.he_av[1] < epsilon) ion.he_av[1] = epsilon;
    if (ion.he_av[2] < epsilon) ion.he_av[2] = epsilon;
    if (ion.he_av[0] < epsilon) ion.he_av[0] = epsilon;
    double normfac = ion.he_av[0] + ion.he_av[1] + ion.he_av[2];
    ion.he_av[0] = ion.he_av[0] / normfac;
    ion.he_av[1] = ion.he_av[1] / normfac;
    ion.he_av[2] = ion.he_av[2] / normfac;
  }
}

// cooling_h.f90:40-71; cool = 5 x 801 linear curves (h0, h1, he0, he1, he2)
// `logtab`: the (invc, logc) table of the bit-exact log, wherever the caller keeps it (null: the global one)
C2R_HD double coolin(const double *cool, double mintemp, double dtemp, double nucldens, double eldens,
                     const double *xh, const double *xhe, double temp0, const double *logtab = nullptr) {
  double tpos = ((logtab ? C2R_MATH_LOG10PT(temp0, logtab) : C2R_MATH_LOG10P(temp0)) - mintemp) / dtemp + 1.0;
  int itpos = (int)tpos;
  itpos = itpos < 1 ? 1 : itpos;
  itpos = itpos > NCOOL - 1 ? NCOOL - 1 : itpos;
  double dtpos = tpos - (double)itpos;
  int itpos1 = itpos + 1 > NCOOL ? NCOOL : itpos + 1;
  int a = itpos - 1, b = itpos1 - 1;
  const double *h0 = cool, *h1 = cool + NCOOL, *he0 = cool + 2 * NCOOL, *he1 = cool + 3 * NCOOL,
               *he2 = cool + 4 * NCOOL;
  return nucldens * eldens *
         ((xh[0] * (h0[a] + (h0[b] - h0[a]) * dtpos) + xh[1] * (h1[a] + (h1[b] - h1[a]) * dtpos)) * (1.0 - abu_he) +
          (xhe[0] * (he0[a] + (he0[b] - he0[a]) * dtpos) + xhe[1] * (he1[a] + (he1[b] - he1[a]) * dtpos) +
           xhe[2] * (he2[a] + (he2[b] - he2[a]) * dtpos)) * abu_he);
}

// cosmology.f90:207-234
C2R_HD double cosmo_cool(double e_int, double zred, double H0, double Omega0) {
  double opz = 1.0 + zred;
  double dzdt = H0 * opz * sqrt(Omega0 * (opz * opz * opz) + 1.0 - Omega0);
  return e_int * 2.0 / (1.0 + zred) * dzdt;
}

struct CoolData {
  const double *cool;
  double mintemp, dtemp;
  double zred, H0, Omega0;
  const double *logtab; // see coolin; may be null
};

// thermal.f90:22-174 (cosmological = .true.)
// `work` / `budget` (optional): the caller's running count of sub-steps and its ceiling.  When the count
// passes a non-zero ceiling the routine returns true at once and its outputs are meaningless: the caller
// abandons the cell and redoes it from scratch later (k_chemistry's tiers).
C2R_HD bool thermal(const CoolData &cd, double dt, double &end_temper, double &avg_temper, double ndens_electron,
                    double ndens_atom, const IonStates &ion, double heating, int *work = nullptr, int budget = 0) {
  double internal_energy =
      temper2pressr(end_temper, ndens_atom, electrondens(ndens_atom, ion.h_old, ion.he_old)) / gamma1;
  double cosmo_cool_rate = cosmo_cool(internal_energy, cd.zred, cd.H0, cd.Omega0);
  if (end_temper > minitemp) {
    double cumulative_time = 0.0;
    int i_heating = 0;
    avg_temper = 0.0;
    double initial_temp = end_temper;
    const double eldens_av = electrondens(ndens_atom, ion.h_av, ion.he_av);
    for (;;) {
      i_heating++;
      if (work) {
        if (++(*work) > budget && budget) return true;
      }
      double cooling =
          coolin(cd.cool, cd.mintemp, cd.dtemp, ndens_atom, ndens_electron, ion.h_av, ion.he_av, end_temper, cd.logtab) +
          cosmo_cool_rate;
      double thermal_rate = dmax(1e-50, fabs(cooling - heating));
      double thermal_timescale = internal_energy / fabs(thermal_rate);
      double dt_thermal = relative_denergy * thermal_timescale;
      double dt_ODE = dmin(dt_thermal, dt - cumulative_time);
      internal_energy = internal_energy + dt_ODE * (heating - cooling);
      avg_temper = avg_temper + 0.5 * end_temper * dt_ODE;
      end_temper = pressr2temper(internal_energy * gamma1, ndens_atom, eldens_av);
      avg_temper = avg_temper + 0.5 * end_temper * dt_ODE;
      if (end_temper < minitemp) {
        internal_energy = temper2pressr(minitemp, ndens_atom, eldens_av); // thermal.f90:141, no /gamma1
        end_temper = minitemp;
      }
      cumulative_time = cumulative_time + dt_ODE;
      if (cumulative_time >= dt || fabs(cumulative_time - dt) < C2R_F(1e-6) * dt) break;
      if (i_heating > 10000) break;
    }
    if (dt > 0.0) avg_temper = avg_temper / dt; else avg_temper = initial_temp;
    end_temper = pressr2temper(internal_energy * gamma1, ndens_atom, electrondens(ndens_atom, ion.h, ion.he));
  }
  return false;
}

// true if the predicate holds in any lane of the wave (on the host: for this one evaluation)
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ bool any_lane(bool p) { return __any(p ? 1 : 0) != 0; }
#else
inline bool any_lane(bool p) { return p; }
#endif

// Diagnostic build only (-DC2R_RATES_COUNT, tools/bench_config4.py --lane-census): how well the lanes of a wave
// are used by the band loop.  [0] band bodies a wave ran, [1] lanes alive in them, [2] bands skipped by the
// whole wave, [3] source bodies a wave ran, [4] lanes inside the sub-box in them, [5] sources skipped by the wave;
// [6..8] the same triple for div_by_vol's doubt branch, [9..11] for its true divisions, [12..14] for the near-1 path
// of the log, [21..23] for optically thin bands (photo_lookuptable's second branch).
#if defined(C2R_RATES_COUNT)
extern __device__ unsigned long long c2r_rates_cnt[64 * 24];
#endif
#if defined(C2R_RATES_COUNT) && defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void count_lanes(int what, bool alive) {
  const unsigned long long here = __ballot(1), m = __ballot(alive ? 1 : 0);
  if ((int)(threadIdx.x & 63) == __ffsll((long long)here) - 1) {
    unsigned long long *c = c2r_rates_cnt + (blockIdx.x & 63) * 24;
    if (m) {
      atomicAdd(c + what, 1ull);
      atomicAdd(c + what + 1, (unsigned long long)__popcll(m));
    } else {
      atomicAdd(c + what + 2, 1ull);
    }
  }
}
#define C2R_COUNT_LANES(what, alive) count_lanes(what, alive)
#else
#define C2R_COUNT_LANES(what, alive) ((void)0)
#endif

// ----------------------------------------------------------------------------------------
// radiation_photoionrates.f90: table position of one optical depth (:282-306)
struct TauPos {
  int ipos;
  double residual;
};
#if defined(__HIP_DEVICE_COMPILE__)
// max / min of two numbers neither of which is a NaN: one instruction
C2R_HD double dmax_num(double a, double b) { return __builtin_fmax(a, b); }
C2R_HD double dmin_num(double a, double b) { return __builtin_fmin(a, b); }
// max(a, k) with the instruction spelled out: __builtin_fmax puts a canonicalising v_max_f64 a, a, a in front
// whenever it cannot see where `a` was computed (another basic block); the operands here are results of
// arithmetic, never signalling NaNs
__device__ __forceinline__ double dmax_const(double a, double k) {
  double d;
  asm("v_max_f64 %0, %1, %2" : "=v"(d) : "v"(a), "s"(k));
  return d;
}
#else
C2R_HD double dmax_num(double a, double b) { return a > b ? a : b; }
C2R_HD double dmin_num(double a, double b) { return a < b ? a : b; }
C2R_HD double dmax_const(double a, double k) { return a > k ? a : k; }
#endif
// the position that belongs to lt = log10(max(1e-20, tau))
C2R_HD TauPos table_position_of_log(double lt) {
  // (lt - minlogtau)/dlogtau, correctly rounded through the constant's reciprocal (div_recip);
  // the numerator lies in [0, 24.5], dlogtau = 0x1.89374bc6a7efap-7
  const double num = lt - minlogtau;
  const double rdl = 1.0 / dlogtau;
  const double q0 = num * rdl;
  const double quo = __builtin_fma(__builtin_fma(-dlogtau, q0, num), rdl, q0);
  // max(0, .) of the reference (:299) cannot bind: lt >= log10(1e-20) makes 1 + quo >= 1 - 1e-12
  const double odpos = dmin_num((double)NTAU, 1.0 + quo);
  TauPos p;
  p.ipos = (int)odpos;
#if defined(__HIP_DEVICE_COMPILE__)
  p.residual = __builtin_amdgcn_fract(odpos); // odpos - floor(odpos), exact; odpos >= 0
#else
  p.residual = odpos - (double)p.ipos;
#endif
  return p;
}
#if !defined(C2R_USE_PLATFORM_LIBM)
// `logtab`: the (invc, logc) table of the bit-exact log, wherever the caller keeps it (k_rates: LDS)
C2R_HD TauPos tau_table_position(double tau, const double *logtab = C2R_LOGTAB_DEFAULT) {
  const double x = dmax_num(tau, 1.0e-20); // positive, normal and finite by construction
  return table_position_of_log(gm::log10_norm(x, logtab));
}
// Two optical depths at once (the two faces of a cell): both table-path evaluations of the log run as one
// straight line -- their loads and dependent fma chains overlap --, and the polynomial path of __log_fma
// (arguments near 1: one in ten, spatially coherent) is entered only when some lane of the wave needs it.
// The table comes in two forms, told apart by its type: (invc, logc) pairs (const double *) or the 256 entries with
// the power of two folded in (const gm::LogEntry *; k_rates keeps that one in LDS).  (A third, with log10's two
// products of its own power of two tabulated per exponent as well -- 17 KB more LDS, three instructions fewer per
// logarithm -- was measured: 19.4 against 18.2 ms per launch; six more live registers cost the fifth wave per SIMD.)
C2R_HD double log_table_path_of(const gm::Log10Arg &a, const double *tab, const gm::LogPins *pins) { return gm::log_table_path(a, tab, pins); }
C2R_HD double log_table_path_of(const gm::Log10Arg &a, const gm::LogEntry *tab, const gm::LogPins *pins) { return gm::log_table_path4(a, tab, pins); }

template <class LT>
C2R_HD void tau_table_positions(double tau_a, double tau_b, const LT *logtab, TauPos &pa, TauPos &pb,
                                const gm::LogPins *pins = nullptr) {
  const gm::Log10Arg a = gm::log10_split(dmax_const(tau_a, 1.0e-20)), b = gm::log10_split(dmax_const(tau_b, 1.0e-20));
  double lga = log_table_path_of(a, logtab, pins), lgb = log_table_path_of(b, logtab, pins);
  const bool na = gm::log10_near1(a), nb = gm::log10_near1(b);
  C2R_COUNT_LANES(12, na || nb);
  C2R_COUNT_LANES(15, na && nb);
  C2R_COUNT_LANES(18, any_lane(na) && any_lane(nb));
  if (na || nb) {
    if (na) lga = gm::log_near1(gm::log10_arg_value(a));
    if (nb) lgb = gm::log_near1(gm::log10_arg_value(b));
  }
  pa = table_position_of_log(gm::log10_finish(a, lga));
  pb = table_position_of_log(gm::log10_finish(b, lgb));
}
#else
C2R_HD TauPos tau_table_position(double tau, const double * = nullptr) {
  return table_position_of_log(C2R_MATH_LOG10N(dmax(1.0e-20, tau)));
}
template <class LT>
C2R_HD void tau_table_positions(double tau_a, double tau_b, const LT *, TauPos &pa, TauPos &pb, const void * = nullptr) {
  pa = tau_table_position(tau_a);
  pb = tau_table_position(tau_b);
}
#endif
// :310-326; col points at row 0 of a column with pitch NTAUP whose row 2001 duplicates row 2000,
// so that ipos_p1 = min(NumTau, ipos+1) needs no clamp: (c[2001]-c[2000])*residual == 0 exactly.
C2R_HD double read_table(const double *col, const TauPos &p) {
  const double *e = col + (unsigned)p.ipos;
  double a = e[0], b = e[1];
  return a + (b - a) * p.residual;
}

// Heating tables as the rates kernels read them: INTERLEAVED by band.  The reference keeps one column of 0:NumTau
// per (band, species) -- 1 + 2 x 26 + 3 x 20 = 113 columns (radiation_sizes.f90:23) -- and heat_lookuptable reads,
// for a band with n absorbing species, n columns at the same one or two table positions: n (or 2n) gathers from
// lines 16 KB apart.  Here the n columns of a band are woven together, entry (row r, species k) of band b at
// first_col(b) * NTAUP + r * n + k: the values of one position are n neighbours, the two rows of an interpolation
// 2n consecutive doubles -- one or two cache lines per position instead of n or 2n.  Same numbers, same
// arithmetic (read_heat forms a + (b - a) * residual per species like read_table).  A band's block has the size
// of its n columns, so the whole table keeps its size and every band its offset.  The duplicated last row
// (NTAUP = NumTau + 2) is kept per species.
C2R_HD int heat_species(int b) { return b < NB1 ? 1 : (b < NB1 + NB2 ? 2 : 3); }
C2R_HD int heat_first_col(int b) { // 0-based heating column of (band b, HI)
  return b < NB1 ? b : (b < NB1 + NB2 ? 2 * (b + 1) - NB1 - 2 : 3 * (b + 1) - NB2 - 2 * NB1 - 3);
}
// columns of pitch NTAUP -> interleaved (host side, once per table set)
inline void heat_interleave(const double *cols, double *woven) {
  for (int b = 0; b < NFREQ; b++) {
    const int n = heat_species(b), c0 = heat_first_col(b);
    for (int k = 0; k < n; k++)
      for (int r = 0; r < NTAUP; r++) woven[(size_t)c0 * NTAUP + (size_t)r * n + k] = cols[(size_t)(c0 + k) * NTAUP + r];
  }
}
// the N species of a band at one table position; `band` points at the band's block of an interleaved table
template <int N>
C2R_HD void read_heat(const double *band, const TauPos &p, double (&t)[N]) {
  const double *e = band + (unsigned)p.ipos * (unsigned)N;
  double lo[N], hi[N];
#pragma unroll
  for (int k = 0; k < N; k++) { lo[k] = e[k]; hi[k] = e[N + k]; }
#pragma unroll
  for (int k = 0; k < N; k++) t[k] = lo[k] + (hi[k] - lo[k]) * p.residual;
}

// Beyond some optical depth the tables of a band hold exact zeros (the integrands of radiation_tables.f90:471
// are cut at tau*s(nu) >= 700): every rate of the band is then +0, and adding it changes no sum.  In neutral gas
// behind an ionisation front that is most bands of most cells.  Given the band's columns (pitch NTAUP, `ncols` of
// them: photo thick and thin, and its heating columns if there are any) this returns a depth X such that
// tau >= X puts the table position at or beyond the last non-zero entry of all of them: one table step above
// the exact boundary, so that no rounding of the log can matter.  Host side, once per table set.
inline double band_tau_zero(const double *const *cols, int ncols) {
  int z = 0; // first row from which all columns are zero
  for (int c = 0; c < ncols; c++)
    for (int r = NTAU + 1; r >= z; r--)
      if (cols[c][r] != 0.0) { z = r + 1; break; }
  if (z > NTAU) return (double)INFINITY;
  return pow(10.0, minlogtau + (double)z * dlogtau);
}

struct PhotoOut {
  double photo_HI, photo_HeI, photo_HeII; // cell rates (before division by neutral densities)
  double heat;
  double photo_out;                       // photons leaving the cell (all bands)
};

// Ricotti et al. 2002 secondary-ionisation parameters (radiation_photoionrates.f90:49-55, :558-564).
// They depend on the cell's ionised fraction only, so a kernel that loops over sources evaluates them
// once per cell (12 pow) and hands them to every photoion_rates call of that cell.
struct Ricotti {
  double y1R[3], y2R[3];
};
// The six numbers are used by the heating bands only, a few times per band; a kernel short of registers may keep them
// somewhere cheaper than twelve vector registers for the whole of its source loop (k_rates: LDS, one column per lane):
// RicottiParked points at y1R[0] of this lane, component n (y1R[0..2], y2R[0..2]) lies n * stride doubles further on.
#if defined(__HIP_DEVICE_COMPILE__)
// LDS, said so in the type: a volatile access through a generic pointer is a flat_load with system scope
typedef const volatile __attribute__((address_space(3))) double *ParkedPtr;
#else
typedef const volatile double *ParkedPtr;
#endif
struct RicottiParked {
  ParkedPtr p;
  int stride;
};
C2R_HD double ric_y1(const Ricotti &r, int i) { return r.y1R[i]; }
C2R_HD double ric_y2(const Ricotti &r, int i) { return r.y2R[i]; }
// (volatile: read where it is used -- hoisted out of the band loops the values would be back in registers)
C2R_HD double ric_y1(const RicottiParked &r, int i) { return r.p[i * r.stride]; }
C2R_HD double ric_y2(const RicottiParked &r, int i) { return r.p[(3 + i) * r.stride]; }
C2R_HD Ricotti ricotti_parameters(double i_state) {
  const double CR1[3] = {0.3908, 0.0554, 1.0}, bR1[3] = {0.4092, 0.4614, 0.2663},
               dR1[3] = {1.7592, 1.6660, 1.3163};
  const double CR2[3] = {0.6941, 0.0984, 3.9811}, aR2[3] = {0.2, 0.2, 0.4}, bR2[3] = {0.38, 0.38, 0.34};
  Ricotti R;
  for (int i = 0; i < 3; i++) {
    R.y1R[i] = CR1[i] * C2R_MATH_POW(1.0 - C2R_MATH_POW(i_state, bR1[i]), dR1[i]);
    double xeb = 1.0 - C2R_MATH_POW(i_state, bR2[i]);
    R.y2R[i] = CR2[i] * C2R_MATH_POW(i_state, aR2[i]) * xeb * xeb;
  }
  return R;
}

// What stays the same for every band of one (cell, source) pair.
struct CellSrc {
  double cin_HI, cin_HeI, cin_HeII, cout_HI, cout_HeI, cout_HeII;
  double cell_HI, cell_HeI, cell_HeII; // column of the cell itself
  double NFlux;
  Recip rvol;      // the shell volume vol_ph as divisor
  bool recip_safe; // the sums of scale_int2/3 lie where recip_nr is exact (see there)
  const gm::LogPins *pins; // see gm::LogPins; may be null
};

// RN(1/x) for the denominators of scale_int2 / scale_int3: the instruction sequence the compiler emits for 1.0/x
// (v_rcp_f64 seed, two Newton steps, the residual correction) without the operand scaling (v_div_scale) and the
// special-case patch-up (v_div_fixup) around it, which do nothing for 2^-600 < x < 2^600: 7 instructions instead
// of 11, same bits (tests/test_gpu_math.py compares it with the division on 4e6 operands).  The caller
// guarantees the range: CellSrc::recip_safe (cell columns within 2^+-300, cross sections within [2^-100, 2^-30],
// checked once per table set on the host).
C2R_HD double recip_nr(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  double y = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-x, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-x, y, 1.0);
  return __builtin_fma(e, y, y);
#else
  return 1.0 / x;
#endif
}
C2R_HD bool column_in_recip_range(double c) { return c >= 0x1p-300 && c <= 0x1p300; }

// a/vol for up to three numerators of one band at once: Markstein's sequence (div_recip) unconditionally, one
// comparison per quotient as the guard, and the plain divisions only where a guard fails for a non-zero
// numerator.  With the divisor in [1, 2^400] (Recip::ok_big) a quotient of magnitude >= 2^-900 proves that
// nothing under- or overflowed on the way; a zero numerator gives a zero quotient either way.
template <int N>
C2R_HD void div_by_vol(const Recip &R, const double (&a)[N], double (&d)[N]) {
  double q[N];
  bool doubt = !R.ok_big;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(C2R_DIV_GUARD_F64)
  // the guard on the high words: for q >= 0, q >= 2^-900 exactly when its high word, read as a signed integer, is >=
  // 0x07B00000; a negative quotient has a negative high word and lands in the doubt branch, whose own test -- the
  // one below, on the magnitudes -- then decides as it always did.  One v_min3_i32 and one compare for up to three
  // quotients instead of a 64-bit compare each.
  int m = 0x7fffffff;
#pragma unroll
  for (int n = 0; n < N; n++) {
    q[n] = a[n] * R.y;
    d[n] = __builtin_fma(__builtin_fma(-R.b, q[n], a[n]), R.y, q[n]);
    const int hi = __double2hiint(q[n]);
    m = hi < m ? hi : m;
  }
  doubt = doubt || m < 0x07B00000;
#else
#pragma unroll
  for (int n = 0; n < N; n++) {
    q[n] = a[n] * R.y;
    d[n] = __builtin_fma(__builtin_fma(-R.b, q[n], a[n]), R.y, q[n]);
    doubt = doubt || !(fabs(q[n]) >= 0x1p-900);
  }
#endif
  C2R_COUNT_LANES(6, doubt);
  if (doubt) {
    bool redo = !R.ok_big;
#pragma unroll
    for (int n = 0; n < N; n++) redo = redo || (!(fabs(q[n]) >= 0x1p-900) && a[n] != 0.0);
    C2R_COUNT_LANES(9, redo);
    if (redo) {
#pragma unroll
      for (int n = 0; n < N; n++) d[n] = a[n] / R.b;
    }
  }
}

// sums one SED's lookuptable calls carry from band to band
struct SedSums {
  double photo_HI, photo_HeI, photo_HeII, photo_out;
  double f_heat, f_ion_HI, f_ion_HeI, df_ion_HI, df_ion_HeI;
};

// What one band adds to ONE SED's sums (the lookuptable bodies proper), given everything of the band that does not
// depend on the SED: the cross sections, both table positions, the species split and the thick/thin decisions.
struct BandShared {
  double sHI, sHeI, sHeII;
  double dtau;
  bool thick, hthick;
  TauPos pin, pout;
  double sc_HI, sc_HeI, sc_HeII;
};
template <bool HEAT, int CLS, class RIC, class BD = BandData>
C2R_HD void band_sed(const BD &bd, const double *photo_thick, const double *photo_thin, const double *heat_thick,
                     const double *heat_thin, int b, const CellSrc &c, double NFlux, const BandShared &B, const RIC &ric,
                     SedSums &o) {
  const double sHI = B.sHI, sHeI = B.sHeI, sHeII = B.sHeII, dtau = B.dtau;
  const bool thick = B.thick, hthick = B.hthick;
  const TauPos pin = B.pin, pout = B.pout;
  const double sc_HI = B.sc_HI, sc_HeI = B.sc_HeI, sc_HeII = B.sc_HeII;
  (void)sHeI; (void)sHeII; (void)sc_HI; (void)sc_HeI; (void)sc_HeII; (void)hthick;
  // photo_lookuptable body
  C2R_COUNT_LANES(21, !thick); // [21..23] waves with an optically thin lane in this band, such lanes, waves without
  {
    const double *tk = photo_thick + (size_t)b * NTAUP;
    double phi_in, phi_out, phi_all;
    if (thick) {
      const double t_in = read_table(tk, pin), t_out = read_table(tk, pout);
      phi_in = NFlux * t_in;
      phi_out = NFlux * t_out;
      phi_all = phi_in - phi_out;
    } else {
      const double t_in = read_table(tk, pin), t_thin = read_table(photo_thin + (size_t)b * NTAUP, pin);
      phi_in = NFlux * t_in;
      phi_all = NFlux * dtau * t_thin;
      phi_out = phi_in - phi_all;
    }
    o.photo_out = o.photo_out + phi_out;
    if (CLS == 0) {
      const double a[1] = {phi_all};
      double d[1];
      div_by_vol<1>(c.rvol, a, d);
      o.photo_HI = o.photo_HI + d[0];
    } else if (CLS == 1) {
      const double a[2] = {sc_HI * phi_all, sc_HeI * phi_all};
      double d[2];
      div_by_vol<2>(c.rvol, a, d);
      o.photo_HI = o.photo_HI + d[0];
      o.photo_HeI = o.photo_HeI + d[1];
    } else {
      const double a[3] = {sc_HI * phi_all, sc_HeI * phi_all, sc_HeII * phi_all};
      double d[3];
      div_by_vol<3>(c.rvol, a, d);
      o.photo_HI = o.photo_HI + d[0];
      o.photo_HeI = o.photo_HeI + d[1];
      o.photo_HeII = o.photo_HeII + d[2];
    }
  }

  if (HEAT) {
    double df_heat;
    // heat_thick / heat_thin: interleaved tables (heat_interleave)
    if (CLS == 0) {
      const double *tk = heat_thick + (size_t)b * NTAUP;
      double a[1];
      if (hthick) {
        const double in_HI = NFlux * read_table(tk, pin), out_HI = NFlux * read_table(tk, pout);
        a[0] = in_HI - out_HI;
      } else {
        a[0] = NFlux * (c.cell_HI * sHI) * read_table(heat_thin + (size_t)b * NTAUP, pin);
      }
      double h[1];
      div_by_vol<1>(c.rvol, a, h);
      df_heat = h[0];
    } else if (CLS == 1) {
      const int cH = 2 * (b + 1) - NB1 - 1 - 1; // 0-based heating column of (band, HI)
      double a[2];
      if (hthick) {
        double t[2], u[2];
        read_heat<2>(heat_thick + (size_t)cH * NTAUP, pin, t);
        read_heat<2>(heat_thick + (size_t)cH * NTAUP, pout, u);
        const double in_HI = NFlux * t[0], in_HeI = NFlux * t[1];
        const double out_HI = NFlux * u[0], out_HeI = NFlux * u[1];
        a[0] = sc_HI * (in_HI - out_HI);
        a[1] = sc_HeI * (in_HeI - out_HeI);
      } else {
        double t[2];
        read_heat<2>(heat_thin + (size_t)cH * NTAUP, pin, t);
        a[0] = NFlux * (c.cell_HI * sHI) * t[0];
        a[1] = NFlux * (c.cell_HeI * sHeI) * t[1];
      }
      double h[2];
      div_by_vol<2>(c.rvol, a, h);
      const double h_HI = h[0], h_HeI = h[1];
      df_heat = h_HI + h_HeI;
const double fra_sum1 = band_f(bd, b, 0) * h_HI + band_f(bd, b, 1) * h_HeI;
      const double fra_sum2 = band_f(bd, b, 3) * h_HI + band_f(bd, b, 4) * h_HeI;
      const double fra_sum3 = band_f(bd, b, 6) * h_HI + band_f(bd, b, 7) * h_HeI;
      const double fra_sum4 = band_f(bd, b, 9) * h_HI + band_f(bd, b, 10) * h_HeI;
      o.df_ion_HeI = ric_y1(ric, 1) * fra_sum1 - ric_y2(ric, 1) * fra_sum2;
      o.df_ion_HI = ric_y1(ric, 0) * fra_sum1 - ric_y2(ric, 0) * fra_sum2;
      df_heat = df_heat - ric_y1(ric, 2) * fra_sum3 + ric_y2(ric, 2) * fra_sum4;
    } else {
      const int cH = 3 * (b + 1) - NB2 - NB1 * 2 - 2 - 1;
      double a[3];
      if (hthick) {
        double t[3], u[3];
        read_heat<3>(heat_thick + (size_t)cH * NTAUP, pin, t);
        read_heat<3>(heat_thick + (size_t)cH * NTAUP, pout, u);
        const double in_HI = NFlux * t[0], in_HeI = NFlux * t[1], in_HeII = NFlux * t[2];
        const double out_HI = NFlux * u[0], out_HeI = NFlux * u[1], out_HeII = NFlux * u[2];
        a[0] = sc_HI * (in_HI - out_HI);
        a[1] = sc_HeI * (in_HeI - out_HeI);
        a[2] = sc_HeII * (in_HeII - out_HeII);
      } else {
        double t[3];
        read_heat<3>(heat_thin + (size_t)cH * NTAUP, pin, t);
        a[0] = NFlux * (c.cell_HI * sHI) * t[0];
        a[1] = NFlux * (c.cell_HeI * sHeI) * t[1];
        a[2] = NFlux * (c.cell_HeII * sHeII) * t[2];
      }
      double h[3];
      div_by_vol<3>(c.rvol, a, h);
      const double h_HI = h[0], h_HeI = h[1], h_HeII = h[2];
      df_heat = h_HI + h_HeI + h_HeII;
const double fra_sum1 = band_f(bd, b, 0) * h_HI + band_f(bd, b, 1) * h_HeI + band_f(bd, b, 2) * h_HeII;
      const double fra_sum2 = band_f(bd, b, 3) * h_HI + band_f(bd, b, 4) * h_HeI + band_f(bd, b, 5) * h_HeII;
      const double fra_sum3 = band_f(bd, b, 6) * h_HI + band_f(bd, b, 7) * h_HeI + band_f(bd, b, 8) * h_HeII;
      const double fra_sum4 = band_f(bd, b, 9) * h_HI + band_f(bd, b, 10) * h_HeI + band_f(bd, b, 11) * h_HeII;
      o.df_ion_HeI = ric_y1(ric, 1) * fra_sum1 - ric_y2(ric, 1) * fra_sum2;
      o.df_ion_HI = ric_y1(ric, 0) * fra_sum1 - ric_y2(ric, 0) * fra_sum2;
      df_heat = df_heat - ric_y1(ric, 2) * fra_sum3 + ric_y2(ric, 2) * fra_sum4;
    }
    o.f_heat = o.f_heat + df_heat;
    o.f_ion_HI = o.f_ion_HI + o.df_ion_HI;
    o.f_ion_HeI = o.f_ion_HeI + o.df_ion_HeI;
  }
}

// The optical depths of band b at the two faces of the cell, and the band's cross sections (into B)
template <int CLS, class BD = BandData>
C2R_HD void band_depths(const BD &bd, int b, const CellSrc &c, BandShared &B, double &tau_in, double &tau_out) {
  const double sHI = band_sigma(bd, b, 0);
  double sHeI = 0.0, sHeII = 0.0;
  tau_in = c.cin_HI * sHI;
  tau_out = c.cout_HI * sHI;
  if (CLS >= 1) {
    sHeI = band_sigma(bd, b, 1);
    tau_in = tau_in + c.cin_HeI * sHeI;
    tau_out = tau_out + c.cout_HeI * sHeI;
  }
  if (CLS >= 2) {
    sHeII = band_sigma(bd, b, 2);
    tau_in = tau_in + c.cin_HeII * sHeII;
    tau_out = tau_out + c.cout_HeII * sHeII;
  }
  B.sHI = sHI; B.sHeI = sHeI; B.sHeII = sHeII;
}
// ... and the rest of what the band's SEDs share: thick or thin, both table positions, the species split
template <int CLS, class LT>
C2R_HD void band_positions(const LT *logtab, const CellSrc &c, double tau_in, double tau_out, BandShared &B) {
  const double dtau = tau_out - tau_in;
  B.dtau = dtau;
  B.thick = fabs(dtau) > tau_photo_limit;
  B.hthick = fabs(dtau) > tau_heat_limit;
  // both positions always: an optically thin band (no use for pout) is rare, and one straight line for the
  // two logs is worth more than skipping one of them now and then
  tau_table_positions(tau_in, tau_out, logtab, B.pin, B.pout, c.pins);
  // species split of this band (scale_int2 / scale_int3)
  double sc_HI = 1.0, sc_HeI = 0.0, sc_HeII = 0.0;
  if (CLS == 1) {
    const double tH = B.sHI * c.cell_HI, tHe = B.sHeI * c.cell_HeI;
    const double den = tH + tHe;
    const double forscaleing = c.recip_safe ? recip_nr(den) : 1.0 / den;
    sc_HI = tH * forscaleing;
    sc_HeI = tHe * forscaleing;
  } else if (CLS == 2) {
    const double tH = B.sHI * c.cell_HI, tHe = B.sHeI * c.cell_HeI, tHe2 = B.sHeII * c.cell_HeII;
    const double den = tH + tHe + tHe2;
    const double forscaleing = c.recip_safe ? recip_nr(den) : 1.0 / den;
    sc_HI = tH * forscaleing;
    sc_HeI = tHe * forscaleing;
    sc_HeII = tHe2 * forscaleing;
  }
  B.sc_HI = sc_HI; B.sc_HeI = sc_HeI; B.sc_HeII = sc_HeII;
}

// One frequency band of photo_lookuptable (radiation_photoionrates.f90:331-464) + heat_lookuptable (:470-779) +
// scale_int2/3 (:787-823).  CLS = 0: the band below the He I threshold (HI only), 1: bands NumBndin1+1 ..
// +NumBndin2 (HI and HeI), 2: the bands above the He II threshold (all three species).  The cross sections of
// species that cannot absorb in a band are exactly 0 (radiation_sizes.f90:382-383, :405; checked when the
// tables are set), so their terms -- x*0 + ... with finite x -- are left out: the sums keep their bits.
// `look_for_zero`: test whether the band is beyond the last non-zero table entry (band_tau_zero); returns
// whether it was.  Within a class the optical depth falls from band to band, so once no lane of a wave has
// found a band dead the caller stops asking (a missed skip costs time, never a bit).
template <bool HEAT, int CLS, class LT, class RIC, class BD = BandData>
C2R_HD bool band_rates(const BD &bd, const double *photo_thick, const double *photo_thin, const double *heat_thick,
                       const double *heat_thin, const LT *logtab, const double *tau_zero, bool look_for_zero, int b,
                       const CellSrc &c, const RIC &ric, SedSums &o) {
  BandShared B;
  double tau_in, tau_out;
  band_depths<CLS>(bd, b, c, B, tau_in, tau_out);
  if (look_for_zero && tau_in >= tau_zero[b]) {
    // every table entry this band would read is exactly 0 (tau_out >= tau_in): all its rates are +0 and no sum
    // changes; what the reference's band would leave behind in df_ion is +0 as well
    if (HEAT && CLS >= 1) o.df_ion_HI = o.df_ion_HeI = 0.0;
    return true;
  }
  band_positions<CLS, LT>(logtab, c, tau_in, tau_out, B);
  band_sed<HEAT, CLS>(bd, photo_thick, photo_thin, heat_thick, heat_thin, b, c, c.NFlux, B, ric, o);
  return false;
}

// The same band for TWO SEDs that cover it (the power-law and the quasar-like SED of the -DPL -DQUASARS builds share
// their band range): optical depths, logs, table positions and species split once, then each SED's look-ups and sums
// exactly as band_rates makes them -- every sum sees the same operands in the same order.  dead[k]: band_rates' return
// value for SED k.
template <bool HEAT, int CLS, class LT, class RIC, class BD = BandData>
C2R_HD void band_rates_pair(const BD &bd, const double *const (&photo_thick)[2], const double *const (&photo_thin)[2],
                            const double *const (&heat_thick)[2], const double *const (&heat_thin)[2], const LT *logtab,
                            const double *const (&tau_zero)[2], const bool (&look_for_zero)[2], int b, const CellSrc &c,
                            const double (&NFlux)[2], const RIC &ric, SedSums (&o)[2], bool (&dead)[2]) {
  BandShared B;
  double tau_in, tau_out;
  band_depths<CLS>(bd, b, c, B, tau_in, tau_out);
#pragma unroll
  for (int k = 0; k < 2; k++) {
    dead[k] = look_for_zero[k] && tau_in >= tau_zero[k][b];
    if (dead[k] && HEAT && CLS >= 1) o[k].df_ion_HI = o[k].df_ion_HeI = 0.0;
  }
  if (dead[0] && dead[1]) return;
  band_positions<CLS, LT>(logtab, c, tau_in, tau_out, B);
#pragma unroll
  for (int k = 0; k < 2; k++)
    if (!dead[k]) band_sed<HEAT, CLS>(bd, photo_thick[k], photo_thin[k], heat_thick[k], heat_thin[k], b, c, NFlux[k], B, ric, o[k]);
}

// what one photo_lookuptable + heat_lookuptable pair of calls returns for one SED (before the sums of
// radiation_photoionrates.f90:178-262 put them together)
struct SedAcc {
  double photo_HI, photo_HeI, photo_HeII, photo_out; // photo_lookuptable
  double f_heat, f_ion_HI, f_ion_HeI;                // heat_lookuptable
};

// radiation_photoionrates.f90:108-277 photoion_rates with its callees photo_lookuptable (:331-464),
// heat_lookuptable (:470-779), scale_int2/3 (:787-823) fused into one pass over the active bands
// [blo, bhi) (0-based), in three stretches by band class; every sum runs in band order as in the reference.
// HEAT selects the non-isothermal path.  `logtab`: see tau_table_position.
template <bool HEAT, class LT, class RIC, class BD = BandData>
C2R_HD void sed_rates(const BD &bd, const double *photo_thick, const double *photo_thin,
                      const double *heat_thick, const double *heat_thin, int blo, int bhi, double cin_HI,
                      double cout_HI, double cin_HeI, double cout_HeI, double cin_HeII, double cout_HeII, double vol,
                      double NFlux, const RIC &ric, SedAcc &out, const LT *logtab, const double *tau_zero,
                      const gm::LogPins *pins = nullptr) {
  out.photo_HI = out.photo_HeI = out.photo_HeII = 0.0;
  out.photo_out = 0.0;
  out.f_heat = out.f_ion_HI = out.f_ion_HeI = 0.0;
  if (!(NFlux > 0.0)) return;
  CellSrc c;
  c.cin_HI = cin_HI; c.cin_HeI = cin_HeI; c.cin_HeII = cin_HeII;
  c.cout_HI = cout_HI; c.cout_HeI = cout_HeI; c.cout_HeII = cout_HeII;
  c.cell_HI = cout_HI - cin_HI;
  c.cell_HeI = cout_HeI - cin_HeI;
  c.cell_HeII = cout_HeII - cin_HeII;
  c.NFlux = NFlux;
  c.rvol = make_recip(vol);
  c.pins = pins;
  c.recip_safe = column_in_recip_range(c.cell_HI) && column_in_recip_range(c.cell_HeI) && column_in_recip_range(c.cell_HeII);
  SedSums o = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  const int e0 = bhi < NB1 ? bhi : NB1, e1 = bhi < NB1 + NB2 ? bhi : NB1 + NB2;
  int b = blo;
  bool look = true;
  for (; b < e0; b++) {
    const bool dead = band_rates<HEAT, 0, LT>(bd, photo_thick, photo_thin, heat_thick, heat_thin, logtab, tau_zero, look, b, c, ric, o);
    C2R_COUNT_LANES(0, !dead);
    look = any_lane(dead);
  }
  look = true;
  for (; b < e1; b++) {
    const bool dead = band_rates<HEAT, 1, LT>(bd, photo_thick, photo_thin, heat_thick, heat_thin, logtab, tau_zero, look, b, c, ric, o);
    C2R_COUNT_LANES(0, !dead);
    look = any_lane(dead);
  }
  look = true;
  for (; b < bhi; b++) {
    const bool dead = band_rates<HEAT, 2, LT>(bd, photo_thick, photo_thin, heat_thick, heat_thin, logtab, tau_zero, look, b, c, ric, o);
    C2R_COUNT_LANES(0, !dead);
    look = any_lane(dead);
  }
  out.photo_HI = o.photo_HI;
  out.photo_HeI = o.photo_HeI;
  out.photo_HeII = o.photo_HeII;
  out.photo_out = o.photo_out;
  if (HEAT) {
    out.f_heat = o.f_heat;
    out.f_ion_HI = o.f_ion_HI;
    out.f_ion_HeI = o.f_ion_HeI;
  }
}


// sed_rates for two SEDs with the same band range at once (band_rates_pair): out[k] is bit for bit what
// sed_rates(tables of k, NFlux[k]) returns
template <bool HEAT, class LT, class RIC, class BD = BandData>
C2R_HD void sed_rates_pair(const BD &bd, const double *const (&photo_thick)[2], const double *const (&photo_thin)[2],
                           const double *const (&heat_thick)[2], const double *const (&heat_thin)[2], int blo, int bhi,
                           double cin_HI, double cout_HI, double cin_HeI, double cout_HeI, double cin_HeII, double cout_HeII,
                           double vol, const double (&NFlux)[2], const RIC &ric, SedAcc (&out)[2], const LT *logtab,
                           const double *const (&tau_zero)[2], const gm::LogPins *pins = nullptr) {
  CellSrc c;
  c.cin_HI = cin_HI; c.cin_HeI = cin_HeI; c.cin_HeII = cin_HeII;
  c.cout_HI = cout_HI; c.cout_HeI = cout_HeI; c.cout_HeII = cout_HeII;
  c.cell_HI = cout_HI - cin_HI;
  c.cell_HeI = cout_HeI - cin_HeI;
  c.cell_HeII = cout_HeII - cin_HeII;
  c.NFlux = 0.0; // per SED: handed to band_rates_pair
  c.rvol = make_recip(vol);
  c.pins = pins;
  c.recip_safe = column_in_recip_range(c.cell_HI) && column_in_recip_range(c.cell_HeI) && column_in_recip_range(c.cell_HeII);
  SedSums o[2] = {{0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0}};
  const int e0 = bhi < NB1 ? bhi : NB1, e1 = bhi < NB1 + NB2 ? bhi : NB1 + NB2;
  int b = blo;
  bool look[2] = {true, true}, dead[2];
  for (; b < e0; b++) {
    band_rates_pair<HEAT, 0, LT>(bd, photo_thick, photo_thin, heat_thick, heat_thin, logtab, tau_zero, look, b, c, NFlux, ric, o, dead);
    look[0] = any_lane(dead[0]); look[1] = any_lane(dead[1]);
  }
  look[0] = look[1] = true;
  for (; b < e1; b++) {
    band_rates_pair<HEAT, 1, LT>(bd, photo_thick, photo_thin, heat_thick, heat_thin, logtab, tau_zero, look, b, c, NFlux, ric, o, dead);
    look[0] = any_lane(dead[0]); look[1] = any_lane(dead[1]);
  }
  look[0] = look[1] = true;
  for (; b < bhi; b++) {
    band_rates_pair<HEAT, 2, LT>(bd, photo_thick, photo_thin, heat_thick, heat_thin, logtab, tau_zero, look, b, c, NFlux, ric, o, dead);
    look[0] = any_lane(dead[0]); look[1] = any_lane(dead[1]);
  }
#pragma unroll
  for (int k = 0; k < 2; k++) {
    out[k].photo_HI = o[k].photo_HI;
    out[k].photo_HeI = o[k].photo_HeI;
    out[k].photo_HeII = o[k].photo_HeII;
    out[k].photo_out = o[k].photo_out;
    out[k].f_heat = out[k].f_ion_HI = out[k].f_ion_HeI = 0.0;
    if (HEAT) {
      out[k].f_heat = o[k].f_heat;
      out[k].f_ion_HI = o[k].f_ion_HI;
      out[k].f_ion_HeI = o[k].f_ion_HeI;
    }
  }
}

// photoion_rates for a source with the black-body SED only
template <bool HEAT, class LT = double, class RIC = Ricotti, class BD = BandData>
C2R_HD void photoion_rates(const BD &bd, const double *photo_thick, const double *photo_thin,
                           const double *heat_thick, const double *heat_thin, double cin_HI, double cout_HI,
                           double cin_HeI, double cout_HeI, double cin_HeII, double cout_HeII, double vol,
                           double NFlux, const RIC &ric, PhotoOut &o, const LT *logtab = C2R_LOGTAB_DEFAULT,
                           const gm::LogPins *pins = nullptr) {
  SedAcc a;
  sed_rates<HEAT, LT>(bd, photo_thick, photo_thin, heat_thick, heat_thin, 0, bd.bb_upper, cin_HI, cout_HI, cin_HeI, cout_HeI,
                      cin_HeII, cout_HeII, vol, NFlux, ric, a, logtab, bd.tau_zero[0], pins);
  o.photo_HI = a.photo_HI;
  o.photo_HeI = a.photo_HeI;
  o.photo_HeII = a.photo_HeII;
  o.photo_out = a.photo_out;
  o.heat = 0.0;
  if (HEAT) {
    // phi = phi + heat_lookuptable(...) (:247-252): adds to photo_cell_HI / HeI and heat
    o.heat = a.f_heat;
    o.photo_HI = o.photo_HI + a.f_ion_HI / (ion_freq_HI * hplanck);
    o.photo_HeI = o.photo_HeI + a.f_ion_HeI / (ion_freq_HeI * hplanck);
  }
}

// ----------------------------------------------------------------------------------------
// The -DPL / -DQUASARS builds of the reference add up to two more SEDs per source (power law,
// quasar-like; radiation_photoionrates.f90:215-228, :256-271): the same band loop with that SED's
// tables, flux and band range [Minimum_FreqBnd, Maximum_FreqBnd], each lookuptable call starting
// its sums from zero, the results added to phi in the order photo BB, PL, QPL, heat BB, PL, QPL.
constexpr int NSED = 3;
struct SedSet {
  const double *photo_thick[NSED], *photo_thin[NSED], *heat_thick[NSED], *heat_thin[NSED];
  int lo[NSED], hi[NSED]; // 0-based first band, one past the last band; lo == hi: SED absent
};

template <bool HEAT, class LT = double, class RIC = Ricotti, class BD = BandData>
C2R_HD void photoion_rates_multi(const BD &bd, const SedSet &ss, double cin_HI, double cout_HI, double cin_HeI,
                                 double cout_HeI, double cin_HeII, double cout_HeII, double vol, const double *NFlux,
                                 const RIC &ric, PhotoOut &o, const LT *logtab = C2R_LOGTAB_DEFAULT,
                                 const gm::LogPins *pins = nullptr) {
  o.photo_HI = o.photo_HeI = o.photo_HeII = 0.0;
  o.heat = 0.0;
  o.photo_out = 0.0;
  // one band loop per SED over its own band range, like the reference's lookuptable calls (BB 1..33, PL and QPL
  // 38..47 in the nominal set-up).  The per-SED results live in NAMED variables, and the SED number -- uniform over
  // the wave -- only ever selects between them: an array indexed by the loop counter of a loop the compiler cannot
  // unroll (its trip count depends on `pair`) is a private segment, i.e. scratch memory, on the device
  // (rounds 1-3: 176-208 bytes per lane, a few scratch accesses per cell.source).
  const bool act0 = NFlux[0] > 0.0 && ss.hi[0] > ss.lo[0], act1 = NFlux[1] > 0.0 && ss.hi[1] > ss.lo[1],
             act2 = NFlux[2] > 0.0 && ss.hi[2] > ss.lo[2];
  // the two extra SEDs over the same bands (the nominal set-up: both 38..47): one band loop for both (band_rates_pair).
  // Isothermal kernels only (round 4): with heating the pair's second set of running sums costs the rates kernel its
  // fourth wave per SIMD -- measured on one box, 256^3, 128 sources per pass: every source with three SEDs 866 (pair,
  // 2 waves) / 905 (no pair, 2 waves) / 868 ms (no pair, 4 waves); a third of the sources with a power law and a fifth
  // with a quasar component 378 / 372 / 358 ms.  -DC2R_SED_PAIR_HEAT=1 brings it back, -DC2R_NO_SED_PAIR switches
  // the pair off everywhere (diagnostic builds).
#if defined(C2R_NO_SED_PAIR)
  const bool pair = false;
#else
#if !defined(C2R_SED_PAIR_HEAT)
#define C2R_SED_PAIR_HEAT 0
#endif
  const bool pair = (!HEAT || C2R_SED_PAIR_HEAT) && act1 && act2 && ss.lo[1] == ss.lo[2] && ss.hi[1] == ss.hi[2];
#endif
  // phi = phi + photo_lookuptable(B) [+ (P)] [+ (Q)] runs along with the loops: every sum still receives BB, PL, QPL in
  // that order.  The heat_lookuptable results are added after ALL the photo terms (:247-271), so f_ion of each SED
  // waits in f_HI* / f_HeI*; o.heat only ever receives f_heat, in SED order as well.
  double f_HI0 = 0.0, f_HI1 = 0.0, f_HI2 = 0.0, f_HeI0 = 0.0, f_HeI1 = 0.0, f_HeI2 = 0.0;
  const int nsingle = pair ? 1 : NSED;
  for (int s = 0; s < nsingle; s++) {
    const bool act = s == 0 ? act0 : (s == 1 ? act1 : act2);
    if (!act) continue;
    const double nf = s == 0 ? NFlux[0] : (s == 1 ? NFlux[1] : NFlux[2]);
    SedAcc a;
    sed_rates<HEAT, LT>(bd, ss.photo_thick[s], ss.photo_thin[s], ss.heat_thick[s], ss.heat_thin[s], ss.lo[s], ss.hi[s], cin_HI,
                        cout_HI, cin_HeI, cout_HeI, cin_HeII, cout_HeII, vol, nf, ric, a, logtab, bd.tau_zero[s], pins);
    o.photo_HI = o.photo_HI + a.photo_HI;
    o.photo_HeI = o.photo_HeI + a.photo_HeI;
    o.photo_HeII = o.photo_HeII + a.photo_HeII;
    o.photo_out = o.photo_out + a.photo_out;
    if (HEAT) {
      o.heat = o.heat + a.f_heat;
      if (s == 0) { f_HI0 = a.f_ion_HI; f_HeI0 = a.f_ion_HeI; }
      else if (s == 1) { f_HI1 = a.f_ion_HI; f_HeI1 = a.f_ion_HeI; }
      else { f_HI2 = a.f_ion_HI; f_HeI2 = a.f_ion_HeI; }
    }
  }
  if (pair) {
    const double *const pt[2] = {ss.photo_thick[1], ss.photo_thick[2]}, *const pn[2] = {ss.photo_thin[1], ss.photo_thin[2]};
    const double *const ht[2] = {ss.heat_thick[1], ss.heat_thick[2]}, *const hn[2] = {ss.heat_thin[1], ss.heat_thin[2]};
    const double *const tz[2] = {bd.tau_zero[1], bd.tau_zero[2]};
    const double nf[2] = {NFlux[1], NFlux[2]};
    SedAcc a2[2];
    sed_rates_pair<HEAT, LT>(bd, pt, pn, ht, hn, ss.lo[1], ss.hi[1], cin_HI, cout_HI, cin_HeI, cout_HeI, cin_HeII, cout_HeII, vol, nf,
                             ric, a2, logtab, tz, pins);
#pragma unroll
    for (int k = 0; k < 2; k++) {
      o.photo_HI = o.photo_HI + a2[k].photo_HI;
      o.photo_HeI = o.photo_HeI + a2[k].photo_HeI;
      o.photo_HeII = o.photo_HeII + a2[k].photo_HeII;
      o.photo_out = o.photo_out + a2[k].photo_out;
      if (HEAT) o.heat = o.heat + a2[k].f_heat;
    }
    if (HEAT) {
      f_HI1 = a2[0].f_ion_HI; f_HeI1 = a2[0].f_ion_HeI;
      f_HI2 = a2[1].f_ion_HI; f_HeI2 = a2[1].f_ion_HeI;
    }
  }
  if (HEAT) {
    // phi = phi + heat_lookuptable(B) [+ (P)] [+ (Q)] (:247-271): only for the SEDs that ran
    if (act0) {
      o.photo_HI = o.photo_HI + f_HI0 / (ion_freq_HI * hplanck);
      o.photo_HeI = o.photo_HeI + f_HeI0 / (ion_freq_HeI * hplanck);
    }
    if (act1) {
      o.photo_HI = o.photo_HI + f_HI1 / (ion_freq_HI * hplanck);
      o.photo_HeI = o.photo_HeI + f_HeI1 / (ion_freq_HeI * hplanck);
    }
    if (act2) {
      o.photo_HI = o.photo_HI + f_HI2 / (ion_freq_HI * hplanck);
      o.photo_HeI = o.photo_HeI + f_HeI2 / (ion_freq_HeI * hplanck);
    }
  }
}

// phi_photo_out of ONE band of SED `sed` (photo_lookuptable, radiation_photoionrates.f90:331-464): the term
// photo_out_only / photo_out_multi add up in band order.  Used on its own by the sampled boundary loss, where
// any subset of these non-negative terms, in any order, is a valid lower bound.
C2R_HD double photo_out_band(const BandData &bd, int sed, const double *photo_thick, const double *photo_thin, int b,
                             double cin_HI, double cout_HI, double cin_HeI, double cout_HeI, double cin_HeII,
                             double cout_HeII, double NFlux) {
  const double sHI = bd.sigma_HI[b], sHeI = bd.sigma_HeI[b], sHeII = bd.sigma_HeII[b];
  const double tau_in = cin_HI * sHI + cin_HeI * sHeI + cin_HeII * sHeII;
  if (tau_in >= bd.tau_zero[sed][b]) return 0.0; // phi_out = +0 exactly (band_tau_zero)
  const double tau_out = cout_HI * sHI + cout_HeI * sHeI + cout_HeII * sHeII;
  const double *tk = photo_thick + (size_t)b * NTAUP;
  if (fabs(tau_out - tau_in) > tau_photo_limit) {
    const TauPos pout = tau_table_position(tau_out);
    return NFlux * read_table(tk, pout);
  }
  const TauPos pin = tau_table_position(tau_in);
  const double phi_in = NFlux * read_table(tk, pin);
  const double phi_all = NFlux * (tau_out - tau_in) * read_table(photo_thin + (size_t)b * NTAUP, pin);
  return phi_in - phi_all;
}

C2R_HD double photo_out_multi(const BandData &bd, const SedSet &ss, double cin_HI, double cout_HI, double cin_HeI,
                              double cout_HeI, double cin_HeII, double cout_HeII, const double *NFlux) {
  double total = 0.0;
  for (int s = 0; s < NSED; s++) {
    if (!(NFlux[s] > 0.0 && ss.hi[s] > ss.lo[s])) continue;
    double photo_out = 0.0;
    for (int b = ss.lo[s]; b < ss.hi[s]; b++) {
      const double sHI = bd.sigma_HI[b], sHeI = bd.sigma_HeI[b], sHeII = bd.sigma_HeII[b];
      const double tau_in = cin_HI * sHI + cin_HeI * sHeI + cin_HeII * sHeII;
      if (tau_in >= bd.tau_zero[s][b]) continue; // phi_out = +0 exactly (band_tau_zero)
      const double tau_out = cout_HI * sHI + cout_HeI * sHeI + cout_HeII * sHeII;
      const double *tk = ss.photo_thick[s] + (size_t)b * NTAUP;
      double phi_out;
      if (fabs(tau_out - tau_in) > tau_photo_limit) {
        const TauPos pout = tau_table_position(tau_out);
        phi_out = NFlux[s] * read_table(tk, pout);
      } else {
        const TauPos pin = tau_table_position(tau_in);
        double phi_in = NFlux[s] * read_table(tk, pin);
        double phi_all = NFlux[s] * (tau_out - tau_in) * read_table(ss.photo_thin[s] + (size_t)b * NTAUP, pin);
        phi_out = phi_in - phi_all;
      }
      photo_out = photo_out + phi_out;
    }
    total = total + photo_out;
  }
  return total;
}

// photo_out only (the quantity evolve0D adds to the photon loss of boundary cells,
// evolve_point.F90:310-315): the band loop of photo_lookuptable reduced to phi_photo_out_all.
// An optically thick band needs only the table position of tau_out (phi_out = NFlux*T(tau_out));
// a thin one only that of tau_in (phi_out = phi_in - NFlux*dtau*Tthin(tau_in)): one log10 per band.
C2R_HD double photo_out_only(const BandData &bd, const double *photo_thick, const double *photo_thin,
                             double cin_HI, double cout_HI, double cin_HeI, double cout_HeI, double cin_HeII,
                             double cout_HeII, double NFlux) {
  double photo_out = 0.0;
  if (!(NFlux > 0.0)) return photo_out;
  const int nb = bd.bb_upper;
  for (int b = 0; b < nb; b++) {
    const double sHI = bd.sigma_HI[b], sHeI = bd.sigma_HeI[b], sHeII = bd.sigma_HeII[b];
    const double tau_in = cin_HI * sHI + cin_HeI * sHeI + cin_HeII * sHeII;
    if (tau_in >= bd.tau_zero[0][b]) continue; // phi_out = +0 exactly (band_tau_zero)
    const double tau_out = cout_HI * sHI + cout_HeI * sHeI + cout_HeII * sHeII;
    const double *tk = photo_thick + (size_t)b * NTAUP;
    double phi_out;
    if (fabs(tau_out - tau_in) > tau_photo_limit) {
      const TauPos pout = tau_table_position(tau_out);
      phi_out = NFlux * read_table(tk, pout);
    } else {
      const TauPos pin = tau_table_position(tau_in);
      double phi_in = NFlux * read_table(tk, pin);
      double phi_all = NFlux * (tau_out - tau_in) * read_table(photo_thin + (size_t)b * NTAUP, pin);
      phi_out = phi_in - phi_all;
    }
    photo_out = photo_out + phi_out;
  }
  return photo_out;
}

// ----------------------------------------------------------------------------------------
// files_for_3D/column_density.f90:28-376 -- geometry part of cinterp: which four upstream cells,
// their bilinear weights, the diagonal factor and the path length (in cell units) for the
// unwrapped offset (idel,jdel,kdel) = rtpos - srcpos of a source at mesh position (i0,j0,k0).
// The crossing point is formed in ABSOLUTE mesh coordinates as the reference does
// (xc = alam*di + real(i0), :113): the rounding of that sum depends on i0, so the weights do too.
struct ShortChar {
  int ci[4], cj[4], ck[4]; // unwrapped offsets (relative to the source) of the four corners c1..c4
  double s[4];             // s1..s4
  double diag;             // 1, sqrt2 or sqrt3 (single-precision values, :53-54)
  double path;
};
C2R_HD int isign1(int x) { return x >= 0 ? 1 : -1; } // sign(1,x)

C2R_HD void short_characteristic(int i0, int j0, int k0, int idel, int jdel, int kdel, ShortChar &sc) {
  const double sqrt3 = 1.73205077648162842, sqrt2 = 1.41421353816986084; // (double)sqrtf(3), (double)sqrtf(2)
  const int idela = idel < 0 ? -idel : idel, jdela = jdel < 0 ? -jdel : jdel, kdela = kdel < 0 ? -kdel : kdel;
  const int sgni = isign1(idel), sgnj = isign1(jdel), sgnk = isign1(kdel);
  const int im = idel - sgni, jm = jdel - sgnj, km = kdel - sgnk; // offsets of the cell closer to the source
  const double di = (double)idel, dj = (double)jdel, dk = (double)kdel;
  bool d2, d3;
  if (kdela >= jdela && kdela >= idela) { // :107 z-plane crossing
    double alam = ((double)km + sgnk * 0.5) / dk;
    double xc = alam * di + (double)i0, yc = alam * dj + (double)j0;
    double dx = 2.0 * fabs(xc - ((double)(i0 + im) + 0.5 * sgni));
    double dy = 2.0 * fabs(yc - ((double)(j0 + jm) + 0.5 * sgnj));
    sc.s[0] = (1. - dx) * (1. - dy);
    sc.s[1] = (1. - dy) * dx;
    sc.s[2] = (1. - dx) * dy;
    sc.s[3] = dx * dy;
    sc.ci[0] = im;   sc.cj[0] = jm;   sc.ck[0] = km;
    sc.ci[1] = idel; sc.cj[1] = jm;   sc.ck[1] = km;
    sc.ci[2] = im;   sc.cj[2] = jdel; sc.ck[2] = km;
    sc.ci[3] = idel; sc.cj[3] = jdel; sc.ck[3] = km;
    d2 = (kdela == 1 && (idela == 1 || jdela == 1));
    d3 = (idela == 1 && jdela == 1);
    sc.path = sqrt((di * di + dj * dj) / (dk * dk) + 1.0);
  } else if (jdela >= idela && jdela >= kdela) { // :199 y-plane crossing
    double alam = ((double)jm + sgnj * 0.5) / dj;
    double zc = alam * dk + (double)k0, xc = alam * di + (double)i0;
    double dz = 2.0 * fabs(zc - ((double)(k0 + km) + 0.5 * sgnk));
    double dx = 2.0 * fabs(xc - ((double)(i0 + im) + 0.5 * sgni));
    sc.s[0] = (1. - dx) * (1. - dz);
    sc.s[1] = (1. - dz) * dx;
    sc.s[2] = (1. - dx) * dz;
    sc.s[3] = dx * dz;
    sc.ci[0] = im;   sc.cj[0] = jm; sc.ck[0] = km;
    sc.ci[1] = idel; sc.cj[1] = jm; sc.ck[1] = km;
    sc.ci[2] = im;   sc.cj[2] = jm; sc.ck[2] = kdel;
    sc.ci[3] = idel; sc.cj[3] = jm; sc.ck[3] = kdel;
    d2 = (jdela == 1 && (idela == 1 || kdela == 1));
    d3 = (idela == 1 && kdela == 1);
    sc.path = sqrt((di * di + dk * dk) / (dj * dj) + 1.0);
  } else { // :275 x-plane crossing
    double alam = ((double)im + sgni * 0.5) / di;
    double zc = alam * dk + (double)k0, yc = alam * dj + (double)j0;
    double dz = 2.0 * fabs(zc - ((double)(k0 + km) + 0.5 * sgnk));
    double dy = 2.0 * fabs(yc - ((double)(j0 + jm) + 0.5 * sgnj));
    sc.s[0] = (1. - dz) * (1. - dy);
    sc.s[1] = (1. - dz) * dy;
    sc.s[2] = (1. - dy) * dz;
    sc.s[3] = dy * dz;
    sc.ci[0] = im; sc.cj[0] = jm;   sc.ck[0] = km;
    sc.ci[1] = im; sc.cj[1] = jdel; sc.ck[1] = km;
    sc.ci[2] = im; sc.cj[2] = jm;   sc.ck[2] = kdel;
    sc.ci[3] = im; sc.cj[3] = jdel; sc.ck[3] = kdel;
    d2 = (idela == 1 && (jdela == 1 || kdela == 1));
    d3 = (jdela == 1 && kdela == 1);
    sc.path = sqrt(1.0 + (dj * dj + dk * dk) / (di * di));
  }
  sc.diag = d2 ? (d3 ? sqrt3 : sqrt2) : 1.0;
}

// column_density.f90:351-376 weightf
C2R_HD double weightf(double cd, double sig) { return 1.0 / dmax(0.6, cd * sig); }

// weighted mean of the four corner columns of one species (:145-163) times the diagonal factor
C2R_HD double interp_column(const ShortChar &sc, double c1, double c2, double c3, double c4, double sig) {
  double w1 = sc.s[0] * weightf(c1, sig), w2 = sc.s[1] * weightf(c2, sig), w3 = sc.s[2] * weightf(c3, sig),
         w4 = sc.s[3] * weightf(c4, sig);
  double c = (c1 * w1 + c2 * w2 + c3 * w3 + c4 * w4) / (w1 + w2 + w3 + w4);
  if (sc.diag != 1.0) c = sc.diag * c;
  return c;
}

} // namespace c2r
