/* TEST INFRASTRUCTURE ONLY -- see c2ray_oracle.h.
 *
 * Plain-C, fp64, single-threaded restatement of the reference's hot path
 * (garrelt/C2-Ray3Dm1D_Helium), function by function, in the reference's own
 * evaluation order.  Each function cites the reference file:line it follows
 * (paths relative to /root/reference/code).  Compile with
 *     gcc -O2 -ffp-contract=off -fPIC -shared
 * (no fast-math, no FMA contraction: the flang -O2 build of the reference on
 * x86-64 uses neither).
 *
 * Numerics rule (SURVEY.md section 8a "numerics note"): Fortran literals
 * without _dp are REAL(4); a dp parameter initialised from one holds the
 * float-rounded value.  F(x) below reproduces exactly that.
 */
#include "c2ray_oracle.h"

#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdlib.h>
#include <string.h>

#define F(x) ((double)(x##f))

/* mathconstants.f90:21, abundances.f90:23-29, atomic.f90 */
static const double pi = F(3.141592654);
static const double abu_he = F(0.074);
static const double abu_c = F(7.1e-7);
/* cgsconstants.f90:26-103 */
static const double m_p = 1.672661e-24;
static const double hplanck = 6.6260755e-27;
static const double k_B = 1.381e-16;
static const double eth0 = F(13.598);
static const double ethe0 = F(24.587), ethe1 = F(54.416);
static const double ev2fr = F(0.241838e15);
/* c2ray_parameters.f90:26-89 */
static const double epsilon = 1.0e-20;
#ifdef ORC_CONVERGENCE_FRACTION /* e.g. -DORC_CONVERGENCE_FRACTION=1.0e-3f: the REAL(4) literal of another c2ray_parameters.f90 */
static const double convergence_fraction = (double)(ORC_CONVERGENCE_FRACTION);
#else
static const double convergence_fraction = F(2.5e-4);
#endif
static const double minimum_fractional_change = F(1.0e-2);
static const double minimum_fraction_of_atoms = F(1.0e-8);
static const double minitemp = F(1.0);
static const double relative_denergy = F(0.1);
#ifndef SUBBOXSIZE   /* c2ray_parameters.f90:51,56; -DSUBBOXSIZE=.. -DMAX_SUBBOX=..: the checker of a product build with other parameters */
#define SUBBOXSIZE 10
#endif
#ifndef MAX_SUBBOX
#define MAX_SUBBOX 1150
#endif
/* cgsphotoconstants.f90:25-50 */
static const double sigma_HI_at_ion_freq = F(6.346e-18);
static const double sigma_HeI_at_ion_freq = F(7.430e-18);
static const double sigma_HeII_at_ion_freq = F(1.589e-18);
static const double sigma_H_heth = 1.238e-18;
static const double sigma_H_heLya = 9.907e-22;
static const double sigma_He_heLya = 1.301e-20;
static const double sigma_He_he2 = 1.690780687052975e-18;
static const double sigma_H_he2 = 1.230695924714239e-19;

static double ev2k(void) { return (double)(1.0f / 8.617e-05f); } /* cgsconstants.f90:39, folded in single */
static double temph0(void) { return eth0 * ev2k(); }              /* :80 */
static double temphe(int i) { return (i == 0 ? ethe0 : ethe1) * ev2k(); } /* :95 */
static double colh0(void) { return F(1.3e-8) * F(0.83) * F(1.0) / (eth0 * eth0); } /* :86 */
static double colhe(int i) {                                       /* :101-103 */
  return i == 0 ? F(1.3e-8) * F(0.63) * F(2.0) / (ethe0 * ethe0)
                : F(1.3e-8) * F(1.30) * F(1.0) / (ethe1 * ethe1);
}
static double gamma1(void) { return 5.0 / 3.0 - 1.0; }             /* atomic.f90 */
static double ion_freq_HI(void) { return ev2fr * eth0; }           /* cgsphotoconstants.f90:31-35 */
static double ion_freq_HeI(void) { return ev2fr * ethe0; }
static double ion_freq_HeII(void) { return ev2fr * ethe1; }

int orc_constants(double *out, int n) {
  double c[] = {pi, abu_he, abu_c, (1.0 - abu_he) + 4.0 * abu_he, gamma1(), hplanck, k_B, m_p,
                temph0(), temphe(0), temphe(1), colh0(), colhe(0), colhe(1), ev2k(), ev2fr, eth0,
                ethe0, ethe1, sigma_HI_at_ion_freq, sigma_HeI_at_ion_freq, sigma_HeII_at_ion_freq,
                ion_freq_HI(), ion_freq_HeI(), ion_freq_HeII(), sigma_H_heth, sigma_H_heLya,
                sigma_He_heLya, sigma_He_he2, sigma_H_he2, epsilon, convergence_fraction,
                minimum_fractional_change, minimum_fraction_of_atoms, minitemp, relative_denergy,
                -20.0, (4.0 - (-20.0)) / (double)(float)ORC_NTAU};
  int m = (int)(sizeof(c) / sizeof(c[0]));
  for (int i = 0; i < m && i < n; i++) out[i] = c[i];
  return m;
}

static double dmax(double a, double b) { return a > b ? a : b; }
static double dmin(double a, double b) { return a < b ? a : b; }
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }
static int fmodulo(int a, int p) { int r = a % p; return r < 0 ? r + p : r; } /* Fortran MODULO */

/* ---------------------------------------------------------------------------------------------
 * cgsconstants.f90:140-266  ini_rec_colion_factors
 */
void orc_ini_rec_colion_factors(double T, orc_reccoef *rc) {
  double lambda;
  /* ini_hydrogen_recombination :156-175 (Hui & Gnedin fits; bare literals are REAL(4)) */
  lambda = 2.0 * (temph0() / T);
  rc->arech0 = F(1.269e-13) * pow(lambda, 1.503) / pow(1.0 + pow(lambda / F(0.522), F(0.470)), F(1.923));
  rc->brech0 = F(2.753e-14) * pow(lambda, 1.500) / pow(1.0 + pow(lambda / F(2.740), F(0.407)), F(2.242));
  /* ini_helium0_recombination :179-213 */
  if (T < 9.e3) {
    lambda = 2.0 * (temph0() / T);
    rc->areche0 = 1.269e-13 * pow(lambda, 1.503) / pow(1.0 + pow(lambda / F(0.522), F(0.470)), F(1.923));
    rc->breche0 = 2.753e-14 * pow(lambda, 1.500) / pow(1.0 + pow(lambda / F(2.740), F(0.407)), F(2.242));
  } else {
    lambda = 2.0 * (temphe(0) / T);
    double dielectronic = 1.9e-3 * pow(T, -1.5) * exp(-4.7e5 / T) * (1.0 + 0.3 * exp(-9.4e4 / T));
    rc->areche0 = 3.000e-14 * pow(lambda, 0.654) + dielectronic;
    /* flang -O2 strength-reduces x**0.75 (constant exponent) to sqrt(x)*sqrt(sqrt(x)) (MLIR math
     * algebraic simplification); the oracle build of the reference therefore evaluates it this way,
     * which differs from pow() by 1 ulp for some arguments (seen at T = 1e4 K). */
    rc->breche0 = 1.260e-14 * (sqrt(lambda) * sqrt(sqrt(lambda))) + dielectronic;
  }
  rc->oreche0 = rc->areche0 - rc->breche0;
  /* ini_helium1_recombination :217-240 */
  lambda = 2.0 * (temphe(1) / T);
  rc->breche1 = 5.5060e-14 * pow(lambda, 1.5) / pow(1.0 + pow(lambda / 2.740, 0.407), 2.242);
  rc->areche1 = F(2.538e-13) * pow(lambda, 1.503) / pow(1.0 + pow(lambda / 0.522, 0.470), 1.923);
  rc->treche1 = 3.4e-13 * pow(T / 1.0e4, -0.6);
  rc->v = 0.285 * pow(T / 1.0e4, 0.119);
  /* ini_hydrogen_helium_collisional_ionization :244-266 (Cox 1970) */
  double sqrtt0 = sqrt(T);
  rc->colli_HI = colh0() * sqrtt0 * exp(-temph0() / T);
  rc->colli_HeI = colhe(0) * sqrtt0 * exp(-temphe(0) / T);
  rc->colli_HeII = colhe(1) * sqrtt0 * exp(-temphe(1) / T);
}

/* tped.f90:41-84 */
static double temper2pressr(double temper, double ndens, double eldens) { return (ndens + eldens) * k_B * temper; }
static double pressr2temper(double pressr, double ndens, double eldens) { return pressr / (k_B * (ndens + eldens)); }
double orc_electrondens(double ndens, const double xh[2], const double xhe[3]) {
  return ndens * (xh[1] * (1.0 - abu_he) + abu_c + abu_he * (xhe[1] + 2.0 * xhe[2]));
}

/* doric.f90:358-372 */
static double coldens(double path, double neufrac, double ndens, double abundance) {
  return neufrac * ndens * path * abundance;
}

/* doric.f90:317-351 */
void orc_prepare_doric_factors(double NH, const double NHe[2], double *yfrac, double *zfrac,
                               double *y2afrac, double *y2bfrac) {
  double tau_H_heth = NH * sigma_H_heth;
  double tau_He_heth = NHe[0] * sigma_HeI_at_ion_freq;
  double tau_H_heLya = NH * sigma_H_heLya;
  double tau_He_heLya = NHe[0] * sigma_He_heLya;
  double tau_H_he2th = NH * sigma_H_he2;
  double tau_He_he2th = NHe[0] * sigma_He_he2;
  double tau_He2_he2th = NHe[1] * sigma_HeII_at_ion_freq;
  *yfrac = tau_H_heth / (tau_H_heth + tau_He_heth);
  *zfrac = tau_H_heLya / (tau_H_heLya + tau_He_heLya);
  *y2afrac = tau_He2_he2th / (tau_He2_he2th + tau_He_he2th + tau_H_he2th);
  *y2bfrac = tau_He_he2th / (tau_He2_he2th + tau_He_he2th + tau_H_he2th);
}

/* doric.f90:35-313 */
void orc_doric(double dt, double rhe, double rhh, orc_ionstates *ion, const orc_photrates *phi,
               double yfrac, double zfrac, double y2afrac, double y2bfrac, const orc_reccoef *rc,
               float clumping_f) {
  (void)rhh;
  const double clumping = (double)clumping_f;
  const double v = rc->v;
  double pfrac = 0.96;
  double heliumfraction = abu_he / (1.0 - abu_he);
  double ffrac = dmax(dmin(10.0 * ion->h[0], 1.0), 0.01);
  double wfrac = (1.425 - 0.737) + 0.737 * yfrac;

  double alpha_h_B = clumping * rc->brech0;
  double alpha_he_1 = clumping * rc->oreche0;
  double alpha_he_B = clumping * rc->breche0;
  double alpha_he_A = clumping * rc->areche0;
  double alpha_he2_B = clumping * rc->breche1;
  double alpha_he2_A = clumping * rc->areche1;
  double alpha_he2_2 = clumping * rc->treche1;
  double alpha_he2_1 = alpha_he2_A - alpha_he2_B;

  double aih0 = dmax(phi->photo_cell_HI + rhe * rc->colli_HI, 1.0e-200);
  double aihe0 = dmax(phi->photo_cell_HeI + rhe * rc->colli_HeI, 1.0e-200);
  double aihe1 = dmax(phi->photo_cell_HeII + rhe * rc->colli_HeII, 1.0e-200);

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

  double Rcoef = twoaihe1 * (ry - ion->he_old[1]);
  double Tcoef = rz - ion->he_old[2];

  double coef2 = (Rcoef + (BminusS)*Tcoef) / (2.0 * Scoef);
  double coef3 = -(Rcoef + (BplusS)*Tcoef) / (2.0 * Scoef);
  double coef1 = -rx + (eigv3x - eigv2x) * (Rcoef / (2.0 * Scoef)) +
                 Tcoef * ((BplusS * eigv3x / (2.0 * Scoef) - BminusS * eigv2x / (2.0 * Scoef))) +
                 ion->h_old[1];

  double lam1dt = dt * lambda1, lam2dt = dt * lambda2, lam3dt = dt * lambda3;
  double elam1dt = exp(lam1dt), elam2dt = exp(lam2dt), elam3dt = exp(lam3dt);

  ion->h[1] = coef1 * elam1dt + coef2 * elam2dt * eigv2x + coef3 * elam3dt * eigv3x + rx;
  ion->he[1] = coef2 * elam2dt * eigv2y + coef3 * elam3dt * eigv3y + ry;
  ion->he[2] = coef2 * elam2dt + coef3 * elam3dt + rz;
  ion->h[0] = 1.0 - ion->h[1];
  ion->he[0] = 1.0 - ion->he[1] - ion->he[2];

  if (ion->h[0] < epsilon) { ion->h[0] = epsilon; ion->h[1] = 1.0 - epsilon; }
  if (ion->h[1] < epsilon) { ion->h[1] = epsilon; ion->h[0] = 1.0 - epsilon; }
  if (ion->he[0] <= epsilon || ion->he[1] <= epsilon || ion->he[2] <= epsilon) {
    if (ion->he[0] < epsilon) ion->he[0] = epsilon;
    if (ion->he[1] < epsilon) ion->he[1] = epsilon;
    if (ion->he[2] < epsilon) ion->he[2] = epsilon;
    double normfac = ion->he[0] + ion->he[1] + ion->he[2];
    ion->he[0] = ion->he[0] / normfac;
    ion->he[1] = ion->he[1] / normfac;
    ion->he[2] = ion->he[2] / normfac;
  }

  const double small = F(1.0e-8); /* doric.f90:267 bare literal */
  double avg_factor_1, avg_factor_2, avg_factor_3;
  if (fabs(lam1dt) < small) avg_factor_1 = coef1; else avg_factor_1 = coef1 * (elam1dt - 1.0) / lam1dt;
  if (fabs(lam2dt) < small) avg_factor_2 = coef2; else avg_factor_2 = coef2 * (elam2dt - 1.0) / lam2dt;
  if (fabs(lam3dt) < small) avg_factor_3 = coef3; else avg_factor_3 = coef3 * (elam3dt - 1.0) / lam3dt;

  ion->h_av[1] = rx + avg_factor_1 + eigv2x * avg_factor_2 + eigv3x * avg_factor_3;
  ion->he_av[1] = ry + eigv2y * avg_factor_2 + eigv3y * avg_factor_3;
  ion->he_av[2] = rz + avg_factor_2 + avg_factor_3;
  ion->h_av[0] = 1.0 - ion->h_av[1];
  ion->he_av[0] = 1.0 - ion->he_av[1] - ion->he_av[2];

  if (ion->h_av[1] < epsilon) { ion->h_av[1] = epsilon; ion->h_av[0] = 1.0 - epsilon; }
  if (ion->h_av[0] < epsilon) { ion->h_av[0] = epsilon; ion->h_av[1] = 1.0 - epsilon; }
  if (ion->he_av[0] <= epsilon || ion->he_av[1] <= epsilon || ion->he_av[2] <= epsilon) {
    if (ion->he_av[1] < epsilon) ion->he_av[1] = epsilon;
    if (ion->he_av[2] < epsilon) ion->he_av[2] = epsilon;
    if (ion->he_av[0] < epsilon) ion->he_av[0] = epsilon;
    double normfac = ion->he_av[0] + ion->he_av[1] + ion->he_av[2];
    ion->he_av[0] = ion->he_av[0] / normfac;
    ion->he_av[1] = ion->he_av[1] / normfac;
    ion->he_av[2] = ion->he_av[2] / normfac;
  }
}

/* cooling_h.f90:40-71 */
double orc_coolin(const orc_tables *tb, double nucldens, double eldens, const double xh[2],
                  const double xhe[3], double temp0) {
  const double *h0 = tb->cool, *h1 = tb->cool + ORC_NCOOL, *he0 = tb->cool + 2 * ORC_NCOOL,
               *he1 = tb->cool + 3 * ORC_NCOOL, *he2 = tb->cool + 4 * ORC_NCOOL;
  double tpos = (log10(temp0) - tb->cool_mintemp) / tb->cool_dtemp + 1.0;
  int itpos = imin(ORC_NCOOL - 1, imax(1, (int)tpos));
  double dtpos = tpos - (double)(float)itpos;
  int itpos1 = imin(ORC_NCOOL, itpos + 1);
  int a = itpos - 1, b = itpos1 - 1; /* 0-based */
  return nucldens * eldens *
         ((xh[0] * (h0[a] + (h0[b] - h0[a]) * dtpos) + xh[1] * (h1[a] + (h1[b] - h1[a]) * dtpos)) * (1.0 - abu_he) +
          (xhe[0] * (he0[a] + (he0[b] - he0[a]) * dtpos) + xhe[1] * (he1[a] + (he1[b] - he1[a]) * dtpos) +
           xhe[2] * (he2[a] + (he2[b] - he2[a]) * dtpos)) * abu_he);
}

/* cosmology.f90:207-234 */
static double cosmo_cool(double e_int, double zred, double H0, double Omega0) {
  double opz = 1.0 + zred;
  double dzdt = H0 * opz * sqrt(Omega0 * (opz * opz * opz) + 1.0 - Omega0);
  return e_int * 2.0 / (1.0 + zred) * dzdt;
}

/* thermal.f90:22-174 (cosmological = .true., c2ray_parameters.f90:84) */
void orc_thermal(const orc_tables *tb, double dt, double *end_temper, double *avg_temper,
                 double ndens_electron, double ndens_atom, const orc_ionstates *ion,
                 const orc_photrates *phi, double zred, double H0, double Omega0) {
  double heating = phi->heat;
  double internal_energy =
      temper2pressr(*end_temper, ndens_atom, orc_electrondens(ndens_atom, ion->h_old, ion->he_old)) / gamma1();
  double cosmo_cool_rate = cosmo_cool(internal_energy, zred, H0, Omega0);
  if (*end_temper > minitemp) {
    double cumulative_time = 0.0;
    int i_heating = 0;
    *avg_temper = 0.0;
    double initial_temp = *end_temper;
    for (;;) {
      i_heating++;
      double cooling = orc_coolin(tb, ndens_atom, ndens_electron, ion->h_av, ion->he_av, *end_temper) + cosmo_cool_rate;
      double thermal_rate = dmax(1e-50, fabs(cooling - heating));
      double thermal_timescale = internal_energy / fabs(thermal_rate);
      double dt_thermal = relative_denergy * thermal_timescale;
      double dt_ODE = dmin(dt_thermal, dt - cumulative_time);
      internal_energy = internal_energy + dt_ODE * (heating - cooling);
      *avg_temper = *avg_temper + 0.5 * *end_temper * dt_ODE;
      *end_temper = pressr2temper(internal_energy * gamma1(), ndens_atom,
                                  orc_electrondens(ndens_atom, ion->h_av, ion->he_av));
      *avg_temper = *avg_temper + 0.5 * *end_temper * dt_ODE;
      if (*end_temper < minitemp) {
        /* thermal.f90:141 -- no division by gamma1 here, reproduced as is */
        internal_energy = temper2pressr(minitemp, ndens_atom, orc_electrondens(ndens_atom, ion->h_av, ion->he_av));
        *end_temper = minitemp;
      }
      cumulative_time = cumulative_time + dt_ODE;
      if (cumulative_time >= dt || fabs(cumulative_time - dt) < F(1e-6) * dt) break;
      if (i_heating > 10000) break;
    }
    if (dt > 0.0) *avg_temper = *avg_temper / dt; else *avg_temper = initial_temp;
    *end_temper = pressr2temper(internal_energy * gamma1(), ndens_atom,
                                orc_electrondens(ndens_atom, ion->h, ion->he));
  }
}

/* ---------------------------------------------------------------------------------------------
 * radiation_photoionrates.f90
 */
typedef struct {
  double tau[ORC_NFREQ], odpos[ORC_NFREQ], residual[ORC_NFREQ];
  int ipos[ORC_NFREQ], ipos_p1[ORC_NFREQ];
} tablepos;

/* :282-306 */
static void set_tau_table_positions(const double *tau, tablepos *p) {
  const double minlogtau = -20.0;
  const double dlogtau = (4.0 - (-20.0)) / (double)(float)ORC_NTAU; /* radiation_tables.f90:59-61 */
  for (int b = 0; b < ORC_NFREQ; b++) {
    p->tau[b] = log10(dmax(1.0e-20, tau[b]));
    p->odpos[b] = dmin((double)ORC_NTAU, dmax(0.0, 1.0 + (p->tau[b] - minlogtau) / dlogtau));
    p->ipos[b] = (int)p->odpos[b];
    p->residual[b] = p->odpos[b] - (double)p->ipos[b];
    p->ipos_p1[b] = imin(ORC_NTAU, p->ipos[b] + 1);
  }
}

/* :310-326; band b and column col are 1-based as in the reference */
static double read_table(const double *table, const tablepos *p, int b, int col) {
  const double *c = table + (size_t)(col - 1) * (ORC_NTAU + 1);
  return c[p->ipos[b - 1]] + (c[p->ipos_p1[b - 1]] - c[p->ipos[b - 1]]) * p->residual[b - 1];
}

#define NB1 1
#define NB2 26
#define NB3 20

/* :331-464; thick/thin = the SED's tables, bands lo..hi = its Minimum/Maximum_FreqBnd */
static void photo_lookuptable(const double *thick, const double *thin, int lo, int hi, const tablepos *pin,
                              const tablepos *pout,
                              const double *tau_in, const double *tau_out, double NFlux, double vol,
                              const double *sc_HI, const double *sc_HeI, const double *sc_HeII,
                              orc_photrates *r) {
  const double tau_photo_limit = F(1.0e-7);
  memset(r, 0, sizeof(*r));
  for (int b = lo; b <= hi; b++) {
    double phi_photo_in_all = NFlux * read_table(thick, pin, b, b);
    double phi_photo_out_all, phi_photo_all;
    r->photo_in = r->photo_in + phi_photo_in_all;
    if (fabs(tau_out[b - 1] - tau_in[b - 1]) > tau_photo_limit) {
      phi_photo_out_all = NFlux * read_table(thick, pout, b, b);
      phi_photo_all = phi_photo_in_all - phi_photo_out_all;
    } else {
      phi_photo_all = NFlux * (tau_out[b - 1] - tau_in[b - 1]) * read_table(thin, pin, b, b);
      phi_photo_out_all = phi_photo_in_all - phi_photo_all;
    }
    r->photo_out = r->photo_out + phi_photo_out_all;
    if (b <= NB1) {
      r->photo_cell_HI = r->photo_cell_HI + phi_photo_all / vol;
    } else if (b <= NB1 + NB2) {
      r->photo_cell_HI = r->photo_cell_HI + sc_HI[b - 1] * phi_photo_all / vol;
      r->photo_cell_HeI = r->photo_cell_HeI + sc_HeI[b - 1] * phi_photo_all / vol;
    } else {
      r->photo_cell_HI = r->photo_cell_HI + sc_HI[b - 1] * phi_photo_all / vol;
      r->photo_cell_HeI = r->photo_cell_HeI + sc_HeI[b - 1] * phi_photo_all / vol;
      r->photo_cell_HeII = r->photo_cell_HeII + sc_HeII[b - 1] * phi_photo_all / vol;
    }
  }
}

/* :470-779 */
static void heat_lookuptable(const orc_tables *tb, const double *thick, const double *thin, int lo, int hi,
                             const tablepos *pin, const tablepos *pout,
                             const double *tau_in, const double *tau_out, const double *tau_cell_HI,
                             const double *tau_cell_HeI, const double *tau_cell_HeII, double NFlux,
                             double vol, double i_state, const double *sc_HI, const double *sc_HeI,
                             const double *sc_HeII, orc_photrates *r) {
  static const double CR1[3] = {0.3908, 0.0554, 1.0}, bR1[3] = {0.4092, 0.4614, 0.2663},
                      dR1[3] = {1.7592, 1.6660, 1.3163};
  static const double CR2[3] = {0.6941, 0.0984, 3.9811}, aR2[3] = {0.2, 0.2, 0.4},
                      bR2[3] = {0.38, 0.38, 0.34};
  const double tau_heat_limit = F(1.0e-4);
  memset(r, 0, sizeof(*r));
  double f_heat = 0.0, f_ion_HI = 0.0, f_ion_HeI = 0.0;
  double fra_sum1 = 0.0, fra_sum2 = 0.0, fra_sum3 = 0.0, fra_sum4 = 0.0;
  double df_ion_HI = 0.0, df_ion_HeI = 0.0, df_heat = 0.0;
  double y1R[3], y2R[3];
  for (int i = 0; i < 3; i++) {
    y1R[i] = CR1[i] * pow(1.0 - pow(i_state, bR1[i]), dR1[i]);
    double xeb = 1.0 - pow(i_state, bR2[i]);
    y2R[i] = CR2[i] * pow(i_state, aR2[i]) * xeb * xeb;
  }
  for (int b = lo; b <= hi; b++) {
    double phi_heat_HI = 0.0, phi_heat_HeI = 0.0, phi_heat_HeII = 0.0;
    int optically_thick = fabs(tau_out[b - 1] - tau_in[b - 1]) > tau_heat_limit;
    if (b <= NB1) {
      double phi_heat_in_HI = NFlux * read_table(thick, pin, b, b);
      if (optically_thick) {
        double phi_heat_out_HI = NFlux * read_table(thick, pout, b, b);
        phi_heat_HI = (phi_heat_in_HI - phi_heat_out_HI) / vol;
      } else {
        phi_heat_HI = NFlux * tau_cell_HI[b - 1] * read_table(thin, pin, b, b);
        phi_heat_HI = phi_heat_HI / vol;
      }
      df_heat = phi_heat_HI;
    } else if (b <= NB1 + NB2) {
      int cH = 2 * b - NB1 - 1, cHe = 2 * b - NB1;
      double phi_heat_in_HI = NFlux * read_table(thick, pin, b, cH);
      double phi_heat_in_HeI = NFlux * read_table(thick, pin, b, cHe);
      if (optically_thick) {
        double phi_heat_out_HI = NFlux * read_table(thick, pout, b, cH);
        phi_heat_HI = sc_HI[b - 1] * (phi_heat_in_HI - phi_heat_out_HI) / vol;
        double phi_heat_out_HeI = NFlux * read_table(thick, pout, b, cHe);
        phi_heat_HeI = sc_HeI[b - 1] * (phi_heat_in_HeI - phi_heat_out_HeI) / vol;
      } else {
        phi_heat_HI = NFlux * tau_cell_HI[b - 1] * read_table(thin, pin, b, cH);
        phi_heat_HI = phi_heat_HI / vol;
        phi_heat_HeI = NFlux * tau_cell_HeI[b - 1] * read_table(thin, pin, b, cHe);
        phi_heat_HeI = phi_heat_HeI / vol;
      }
      df_heat = phi_heat_HI + phi_heat_HeI;
      int q = b - 2; /* f arrays are dimension(2:47) */
      fra_sum1 = tb->f1ion_HI[q] * phi_heat_HI + tb->f1ion_HeI[q] * phi_heat_HeI;
      fra_sum2 = tb->f2ion_HI[q] * phi_heat_HI + tb->f2ion_HeI[q] * phi_heat_HeI;
      fra_sum3 = tb->f1heat_HI[q] * phi_heat_HI + tb->f1heat_HeI[q] * phi_heat_HeI;
      fra_sum4 = tb->f2heat_HI[q] * phi_heat_HI + tb->f2heat_HeI[q] * phi_heat_HeI;
      df_ion_HeI = y1R[1] * fra_sum1 - y2R[1] * fra_sum2;
      df_ion_HI = y1R[0] * fra_sum1 - y2R[0] * fra_sum2;
      df_heat = df_heat - y1R[2] * fra_sum3 + y2R[2] * fra_sum4;
    } else {
      int cH = 3 * b - NB2 - NB1 * 2 - 2, cHe = cH + 1, cHe2 = cH + 2;
      double phi_heat_in_HI = NFlux * read_table(thick, pin, b, cH);
      double phi_heat_in_HeI = NFlux * read_table(thick, pin, b, cHe);
      double phi_heat_in_HeII = NFlux * read_table(thick, pin, b, cHe2);
      if (optically_thick) {
        double phi_heat_out_HI = NFlux * read_table(thick, pout, b, cH);
        phi_heat_HI = sc_HI[b - 1] * (phi_heat_in_HI - phi_heat_out_HI) / vol;
        double phi_heat_out_HeI = NFlux * read_table(thick, pout, b, cHe);
        phi_heat_HeI = sc_HeI[b - 1] * (phi_heat_in_HeI - phi_heat_out_HeI) / vol;
        double phi_heat_out_HeII = NFlux * read_table(thick, pout, b, cHe2);
        phi_heat_HeII = sc_HeII[b - 1] * (phi_heat_in_HeII - phi_heat_out_HeII) / vol;
      } else {
        phi_heat_HI = NFlux * tau_cell_HI[b - 1] * read_table(thin, pin, b, cH);
        phi_heat_HI = phi_heat_HI / vol;
        phi_heat_HeI = NFlux * tau_cell_HeI[b - 1] * read_table(thin, pin, b, cHe);
        phi_heat_HeI = phi_heat_HeI / vol;
        phi_heat_HeII = NFlux * tau_cell_HeII[b - 1] * read_table(thin, pin, b, cHe2);
        phi_heat_HeII = phi_heat_HeII / vol;
      }
      df_heat = phi_heat_HI + phi_heat_HeI + phi_heat_HeII;
      int q = b - 2;
      fra_sum1 = tb->f1ion_HI[q] * phi_heat_HI + tb->f1ion_HeI[q] * phi_heat_HeI + tb->f1ion_HeII[q] * phi_heat_HeII;
      fra_sum2 = tb->f2ion_HI[q] * phi_heat_HI + tb->f2ion_HeI[q] * phi_heat_HeI + tb->f2ion_HeII[q] * phi_heat_HeII;
      fra_sum3 = tb->f1heat_HI[q] * phi_heat_HI + tb->f1heat_HeI[q] * phi_heat_HeI + tb->f1heat_HeII[q] * phi_heat_HeII;
      fra_sum4 = tb->f2heat_HI[q] * phi_heat_HI + tb->f2heat_HeI[q] * phi_heat_HeI + tb->f2heat_HeII[q] * phi_heat_HeII;
      df_ion_HeI = y1R[1] * fra_sum1 - y2R[1] * fra_sum2;
      df_ion_HI = y1R[0] * fra_sum1 - y2R[0] * fra_sum2;
      df_heat = df_heat - y1R[2] * fra_sum3 + y2R[2] * fra_sum4;
    }
    f_heat = f_heat + df_heat;
    f_ion_HI = f_ion_HI + df_ion_HI;
    f_ion_HeI = f_ion_HeI + df_ion_HeI;
  }
  r->heat = f_heat;
  r->photo_cell_HI = f_ion_HI / (ion_freq_HI() * hplanck);
  r->photo_cell_HeI = f_ion_HeI / (ion_freq_HeI() * hplanck);
}

static void photrates_add(orc_photrates *a, const orc_photrates *b) { /* :827-854 */
  double *x = (double *)a;
  const double *y = (const double *)b;
  for (int i = 0; i < 21; i++) x[i] = x[i] + y[i];
}

/* :108-277; normflux[0..2] = NormFlux, NormFluxPL, NormFluxQPL of the source (PL / QPL only in the
 * -DPL -DQUASARS build: pass 0 and/or leave the tables NULL otherwise).  Order of the additions as in
 * the reference: photo BB, PL, QPL, then heat BB, PL, QPL. */
void orc_photoion_rates3(const orc_tables *tb, double colum_in_HI, double colum_out_HI,
                         double colum_in_HeI, double colum_out_HeI, double colum_in_HeII,
                         double colum_out_HeII, double vol, const double normflux[3], double i_state,
                         int isothermal, orc_photrates *out) {
  double tau_in_all[ORC_NFREQ], tau_out_all[ORC_NFREQ];
  double tau_cell_HI[ORC_NFREQ], tau_cell_HeI[ORC_NFREQ], tau_cell_HeII[ORC_NFREQ];
  double sc_HI[ORC_NFREQ], sc_HeI[ORC_NFREQ], sc_HeII[ORC_NFREQ];
  tablepos pin, pout;
  orc_photrates phi, tmp;
  memset(&phi, 0, sizeof(phi));
  double colum_cell_HI = colum_out_HI - colum_in_HI;
  double colum_cell_HeI = colum_out_HeI - colum_in_HeI;
  double colum_cell_HeII = colum_out_HeII - colum_in_HeII;
  for (int b = 0; b < ORC_NFREQ; b++)
    tau_in_all[b] = colum_in_HI * tb->sigma_HI[b] + colum_in_HeI * tb->sigma_HeI[b] + colum_in_HeII * tb->sigma_HeII[b];
  for (int b = 0; b < ORC_NFREQ; b++)
    tau_out_all[b] = colum_out_HI * tb->sigma_HI[b] + colum_out_HeI * tb->sigma_HeI[b] + colum_out_HeII * tb->sigma_HeII[b];
  set_tau_table_positions(tau_in_all, &pin);
  set_tau_table_positions(tau_out_all, &pout);
  for (int b = NB1; b < NB1 + NB2; b++) { /* scale_int2 :787-800 */
    double forscaleing = 1.0 / (tb->sigma_HI[b] * colum_cell_HI + tb->sigma_HeI[b] * colum_cell_HeI);
    sc_HI[b] = tb->sigma_HI[b] * colum_cell_HI * forscaleing;
    sc_HeI[b] = tb->sigma_HeI[b] * colum_cell_HeI * forscaleing;
  }
  for (int b = NB1 + NB2; b < ORC_NFREQ; b++) { /* scale_int3 :808-823 */
    double forscaleing = 1.0 / (tb->sigma_HI[b] * colum_cell_HI + tb->sigma_HeI[b] * colum_cell_HeI +
                                tb->sigma_HeII[b] * colum_cell_HeII);
    sc_HI[b] = colum_cell_HI * tb->sigma_HI[b] * forscaleing;
    sc_HeI[b] = colum_cell_HeI * tb->sigma_HeI[b] * forscaleing;
    sc_HeII[b] = colum_cell_HeII * tb->sigma_HeII[b] * forscaleing;
  }
  const double *pthick[3] = {tb->photo_thick, tb->pl_photo_thick, tb->qpl_photo_thick};
  const double *pthin[3] = {tb->photo_thin, tb->pl_photo_thin, tb->qpl_photo_thin};
  const double *hthick[3] = {tb->heat_thick, tb->pl_heat_thick, tb->qpl_heat_thick};
  const double *hthin[3] = {tb->heat_thin, tb->pl_heat_thin, tb->qpl_heat_thin};
  const int lo[3] = {1, tb->pl_lower, tb->qpl_lower}, hi[3] = {tb->bb_upper, tb->pl_upper, tb->qpl_upper};
  for (int sed = 0; sed < 3; sed++) {
    if (pthick[sed] && normflux[sed] > 0.0) {
      photo_lookuptable(pthick[sed], pthin[sed], lo[sed], hi[sed], &pin, &pout, tau_in_all, tau_out_all, normflux[sed],
                        vol, sc_HI, sc_HeI, sc_HeII, &tmp);
      photrates_add(&phi, &tmp);
    }
  }
  if (!isothermal) {
    for (int b = 0; b < ORC_NFREQ; b++) {
      tau_cell_HI[b] = colum_cell_HI * tb->sigma_HI[b];
      tau_cell_HeI[b] = colum_cell_HeI * tb->sigma_HeI[b];
      tau_cell_HeII[b] = colum_cell_HeII * tb->sigma_HeII[b];
    }
    for (int sed = 0; sed < 3; sed++) {
      if (hthick[sed] && normflux[sed] > 0.0) {
        heat_lookuptable(tb, hthick[sed], hthin[sed], lo[sed], hi[sed], &pin, &pout, tau_in_all, tau_out_all,
                         tau_cell_HI, tau_cell_HeI, tau_cell_HeII, normflux[sed], vol, i_state, sc_HI, sc_HeI, sc_HeII,
                         &tmp);
        photrates_add(&phi, &tmp);
      }
    }
  }
  *out = phi;
}

void orc_photoion_rates(const orc_tables *tb, double colum_in_HI, double colum_out_HI,
                        double colum_in_HeI, double colum_out_HeI, double colum_in_HeII,
                        double colum_out_HeII, double vol, double normflux, double i_state,
                        int isothermal, orc_photrates *out) {
  const double nf[3] = {normflux, 0.0, 0.0};
  orc_photoion_rates3(tb, colum_in_HI, colum_out_HI, colum_in_HeI, colum_out_HeI, colum_in_HeII, colum_out_HeII, vol,
                      nf, i_state, isothermal, out);
}

/* ---------------------------------------------------------------------------------------------
 * files_for_3D/column_density.f90:28-376
 */
static double weightf(double cd, int id) { /* :351-376 */
  double sig = id == 0 ? sigma_HI_at_ion_freq : (id == 1 ? sigma_HeI_at_ion_freq : sigma_HeII_at_ion_freq);
  return 1.0 / dmax(0.6, cd * sig);
}

#define IDX(i, j, k) ((size_t)((i)-1) + (size_t)n1 * ((size_t)((j)-1) + (size_t)n2 * (size_t)((k)-1)))

void orc_cinterp(const int mesh[3], const double *cH, const double *cHe, const int pos[3],
                 const int srcpos[3], double *cdensi, double *cdensihe0, double *cdensihe1,
                 double *path) {
  const int n1 = mesh[0], n2 = mesh[1], n3 = mesh[2];
  const size_t ncell = (size_t)n1 * n2 * n3;
  const double sqrt3 = (double)sqrtf(3.0f), sqrt2 = (double)sqrtf(2.0f); /* :53-54 */
  int i = pos[0], j = pos[1], k = pos[2], i0 = srcpos[0], j0 = srcpos[1], k0 = srcpos[2];
  int idel = i - i0, jdel = j - j0, kdel = k - k0;
  int idela = abs(idel), jdela = abs(jdel), kdela = abs(kdel);
  int sgni = idel >= 0 ? 1 : -1, sgnj = jdel >= 0 ? 1 : -1, sgnk = kdel >= 0 ? 1 : -1; /* sign(1,x) */
  int im = i - sgni, jm = j - sgnj, km = k - sgnk;
  double di = (double)idel, dj = (double)jdel, dk = (double)kdel;
  double s1, s2, s3, s4;
  size_t q1, q2, q3, q4;
  int diag3, diag2;

  if (kdela >= jdela && kdela >= idela) { /* :107 z-plane crossing */
    double alam = ((double)(km - k0) + sgnk * 0.5) / dk;
    double xc = alam * di + (double)i0;
    double yc = alam * dj + (double)j0;
    double dx = 2.0 * fabs(xc - ((double)im + 0.5 * sgni));
    double dy = 2.0 * fabs(yc - ((double)jm + 0.5 * sgnj));
    s1 = (1. - dx) * (1. - dy);
    s2 = (1. - dy) * dx;
    s3 = (1. - dx) * dy;
    s4 = dx * dy;
    int ip = fmodulo(i - 1, n1) + 1, imp = fmodulo(im - 1, n1) + 1;
    int jp = fmodulo(j - 1, n2) + 1, jmp = fmodulo(jm - 1, n2) + 1;
    int kmp = fmodulo(km - 1, n3) + 1;
    q1 = IDX(imp, jmp, kmp); q2 = IDX(ip, jmp, kmp); q3 = IDX(imp, jp, kmp); q4 = IDX(ip, jp, kmp);
    diag2 = (kdela == 1 && (idela == 1 || jdela == 1));
    diag3 = (idela == 1 && jdela == 1);
    *path = sqrt((di * di + dj * dj) / (dk * dk) + 1.0);
  } else if (jdela >= idela && jdela >= kdela) { /* :199 y-plane crossing */
    double alam = ((double)(jm - j0) + sgnj * 0.5) / dj;
    double zc = alam * dk + (double)k0;
    double xc = alam * di + (double)i0;
    double dz = 2.0 * fabs(zc - ((double)km + 0.5 * sgnk));
    double dx = 2.0 * fabs(xc - ((double)im + 0.5 * sgni));
    s1 = (1. - dx) * (1. - dz);
    s2 = (1. - dz) * dx;
    s3 = (1. - dx) * dz;
    s4 = dx * dz;
    int ip = fmodulo(i - 1, n1) + 1, imp = fmodulo(im - 1, n1) + 1;
    int jmp = fmodulo(jm - 1, n2) + 1;
    int kp = fmodulo(k - 1, n3) + 1, kmp = fmodulo(km - 1, n3) + 1;
    q1 = IDX(imp, jmp, kmp); q2 = IDX(ip, jmp, kmp); q3 = IDX(imp, jmp, kp); q4 = IDX(ip, jmp, kp);
    diag2 = (jdela == 1 && (idela == 1 || kdela == 1));
    diag3 = (idela == 1 && kdela == 1);
    *path = sqrt((di * di + dk * dk) / (dj * dj) + 1.0);
  } else { /* :275 x-plane crossing */
    double alam = ((double)(im - i0) + sgni * 0.5) / di;
    double zc = alam * dk + (double)k0;
    double yc = alam * dj + (double)j0;
    double dz = 2.0 * fabs(zc - ((double)km + 0.5 * sgnk));
    double dy = 2.0 * fabs(yc - ((double)jm + 0.5 * sgnj));
    s1 = (1. - dz) * (1. - dy);
    s2 = (1. - dz) * dy;
    s3 = (1. - dy) * dz;
    s4 = dy * dz;
    int imp = fmodulo(im - 1, n1) + 1;
    int jp = fmodulo(j - 1, n2) + 1, jmp = fmodulo(jm - 1, n2) + 1;
    int kp = fmodulo(k - 1, n3) + 1, kmp = fmodulo(km - 1, n3) + 1;
    q1 = IDX(imp, jmp, kmp); q2 = IDX(imp, jp, kmp); q3 = IDX(imp, jmp, kp); q4 = IDX(imp, jp, kp);
    diag2 = (idela == 1 && (jdela == 1 || kdela == 1));
    diag3 = (jdela == 1 && kdela == 1);
    *path = sqrt(1.0 + (dj * dj + dk * dk) / (di * di));
  }
  double out[3];
  for (int sp = 0; sp < 3; sp++) {
    const double *g = sp == 0 ? cH : cHe + (size_t)(sp - 1) * ncell;
    double c1 = g[q1], c2 = g[q2], c3 = g[q3], c4 = g[q4];
    double w1 = s1 * weightf(c1, sp), w2 = s2 * weightf(c2, sp), w3 = s3 * weightf(c3, sp), w4 = s4 * weightf(c4, sp);
    out[sp] = (c1 * w1 + c2 * w2 + c3 * w3 + c4 * w4) / (w1 + w2 + w3 + w4);
    if (diag2) out[sp] = (diag3 ? sqrt3 : sqrt2) * out[sp];
  }
  *cdensi = out[0];
  *cdensihe0 = out[1];
  *cdensihe1 = out[2];
}

/* ---------------------------------------------------------------------------------------------
 * files_for_3D/evolve_point.F90:79-319  evolve0D
 */
typedef struct {
  int last_l[3], last_r[3];
  double photon_loss_src_thread;
} sweep_ctx;

static void evolve0D(const orc_tables *tb, const orc_step *st, orc_state *s, const int rtpos[3], int ns,
                     sweep_ctx *cx) {
  const int n1 = st->mesh[0], n2 = st->mesh[1], n3 = st->mesh[2];
  const size_t ncell = (size_t)n1 * n2 * n3;
  const double max_coldensh = (double)2e29f; /* :91 bare literal 2e29 */
  const int *src = st->srcpos + 3 * (ns - 1);
  int pos[3];
  pos[0] = fmodulo(rtpos[0] - 1, n1) + 1;
  pos[1] = fmodulo(rtpos[1] - 1, n2) + 1;
  pos[2] = fmodulo(rtpos[2] - 1, n3) + 1;
  size_t q = IDX(pos[0], pos[1], pos[2]);
  if (s->coldensh_out[q] != 0.0) return;

  double h_av[2], he_av[3];
  for (int nx = 0; nx < 2; nx++) h_av[nx] = dmax(s->xh_av[q + nx * ncell], epsilon);
  for (int nx = 0; nx < 3; nx++) he_av[nx] = dmax(s->xhe_av[q + nx * ncell], epsilon);
  double ndens_p = st->ndens[q];
  double coldensh_in, coldenshe_in[2], path, vol_ph;
  if (rtpos[0] == src[0] && rtpos[1] == src[1] && rtpos[2] == src[2]) {
    coldensh_in = 0.0;
    coldenshe_in[0] = coldenshe_in[1] = 0.0;
    path = 0.5 * st->dr[0];
    vol_ph = st->dr[0] * st->dr[1] * st->dr[2];
  } else {
    orc_cinterp(st->mesh, s->coldensh_out, s->coldenshe_out, rtpos, src, &coldensh_in, &coldenshe_in[0],
                &coldenshe_in[1], &path);
    path = path * st->dr[0];
    double xs = st->dr[0] * (double)(float)(rtpos[0] - src[0]);
    double ys = st->dr[1] * (double)(float)(rtpos[1] - src[1]);
    double zs = st->dr[2] * (double)(float)(rtpos[2] - src[2]);
    double dist2 = xs * xs + ys * ys + zs * zs;
    vol_ph = 4.0 * pi * dist2 * path;
    if (st->use_lls) { /* :177-180 */
      const double coldensh_LLS = st->lls_grid ? (double)st->lls_grid[q] : st->coldensh_lls;
      coldensh_in = coldensh_in + coldensh_LLS * path / st->dr[0];
    }
  }
  s->coldensh_out[q] = coldensh_in + coldens(path, h_av[0], ndens_p, (1.0 - abu_he));
  s->coldenshe_out[q] = coldenshe_in[0] + coldens(path, he_av[0], ndens_p, abu_he);
  s->coldenshe_out[q + ncell] = coldenshe_in[1] + coldens(path, he_av[1], ndens_p, abu_he);

  orc_photrates phi;
  if (coldensh_in < max_coldensh) {
    const double nf[3] = {st->normflux[ns - 1], st->normflux_pl ? st->normflux_pl[ns - 1] : 0.0,
                          st->normflux_qpl ? st->normflux_qpl[ns - 1] : 0.0};
    orc_photoion_rates3(tb, coldensh_in, s->coldensh_out[q], coldenshe_in[0], s->coldenshe_out[q],
                        coldenshe_in[1], s->coldenshe_out[q + ncell], vol_ph, nf, h_av[1],
                        st->isothermal, &phi);
    phi.photo_cell_HI = phi.photo_cell_HI / (h_av[0] * ndens_p * (1.0 - abu_he));
    phi.photo_cell_HeI = phi.photo_cell_HeI / (he_av[0] * ndens_p * abu_he);
    phi.photo_cell_HeII = phi.photo_cell_HeII / (he_av[1] * ndens_p * abu_he);
  } else {
    memset(&phi, 0, sizeof(phi));
  }
  s->phih[q] = s->phih[q] + phi.photo_cell_HI;
  s->phihe[q] = s->phihe[q] + phi.photo_cell_HeI;
  s->phihe[q + ncell] = s->phihe[q + ncell] + phi.photo_cell_HeII;
  if (!st->isothermal) s->phiheat[q] = s->phiheat[q] + phi.heat;

  if (rtpos[0] == cx->last_l[0] || rtpos[1] == cx->last_l[1] || rtpos[2] == cx->last_l[2] ||
      rtpos[0] == cx->last_r[0] || rtpos[1] == cx->last_r[1] || rtpos[2] == cx->last_r[2]) {
    cx->photon_loss_src_thread = cx->photon_loss_src_thread + phi.photo_out * st->vol / vol_ph;
  }
}

/* files_for_3D/evolve_source.F90:244-284  evolve2D */
static void evolve2D(const orc_tables *tb, const orc_step *st, orc_state *s, int rtpos[3], int ns, sweep_ctx *cx) {
  const int *src = st->srcpos + 3 * (ns - 1);
  for (int j = src[1]; j <= cx->last_r[1]; j++) {
    rtpos[1] = j;
    for (int i = src[0]; i <= cx->last_r[0]; i++) { rtpos[0] = i; evolve0D(tb, st, s, rtpos, ns, cx); }
    for (int i = src[0] - 1; i >= cx->last_l[0]; i--) { rtpos[0] = i; evolve0D(tb, st, s, rtpos, ns, cx); }
  }
  for (int j = src[1] - 1; j >= cx->last_l[1]; j--) {
    rtpos[1] = j;
    for (int i = src[0]; i <= cx->last_r[0]; i++) { rtpos[0] = i; evolve0D(tb, st, s, rtpos, ns, cx); }
    for (int i = src[0] - 1; i >= cx->last_l[0]; i--) { rtpos[0] = i; evolve0D(tb, st, s, rtpos, ns, cx); }
  }
}

/* files_for_3D/evolve_source.F90:66-238  do_source (serial branch, periodic_bc = .true.) */
int orc_do_source(const orc_tables *tb, const orc_step *st, orc_state *s, int ns, double *loss_out) {
  const size_t ncell = (size_t)st->mesh[0] * st->mesh[1] * st->mesh[2];
  const int *src = st->srcpos + 3 * (ns - 1);
  int lastpos_r[3], lastpos_l[3];
  sweep_ctx cx;
  memset(s->coldensh_out, 0, ncell * sizeof(double));
  memset(s->coldenshe_out, 0, 2 * ncell * sizeof(double));
  for (int d = 0; d < 3; d++) {
    lastpos_r[d] = src[d] + imin(MAX_SUBBOX, st->mesh[d] / 2 - 1 + st->mesh[d] % 2);
    lastpos_l[d] = src[d] - imin(MAX_SUBBOX, st->mesh[d] / 2);
  }
  int nbox = 0;
  double total_source_flux = st->normflux[ns - 1] * st->s_star; /* evolve_source.F90:122-128 */
  if (st->normflux_pl) total_source_flux = total_source_flux + st->normflux_pl[ns - 1] * st->pl_s_star;
  if (st->normflux_qpl) total_source_flux = total_source_flux + st->normflux_qpl[ns - 1] * st->qpl_s_star;
  double photon_loss_src = total_source_flux;
  for (int d = 0; d < 3; d++) { cx.last_r[d] = src[d]; cx.last_l[d] = src[d]; }
  while (photon_loss_src > F(1e-10) * total_source_flux && cx.last_r[2] < lastpos_r[2] &&
         cx.last_l[2] > lastpos_l[2]) {
    nbox++;
    photon_loss_src = 0.0;
    cx.photon_loss_src_thread = 0.0;
    for (int d = 0; d < 3; d++) {
      cx.last_r[d] = imin(src[d] + SUBBOXSIZE * nbox, lastpos_r[d]);
      cx.last_l[d] = imax(src[d] - SUBBOXSIZE * nbox, lastpos_l[d]);
    }
    int rtpos[3];
    for (int k = src[2]; k <= cx.last_r[2]; k++) { rtpos[2] = k; evolve2D(tb, st, s, rtpos, ns, &cx); }
    for (int k = src[2] - 1; k >= cx.last_l[2]; k--) { rtpos[2] = k; evolve2D(tb, st, s, rtpos, ns, &cx); }
    photon_loss_src = cx.photon_loss_src_thread;
  }
  if (loss_out) *loss_out = photon_loss_src;
  return nbox;
}

/* evolve.F90:371-381 + 385-431 (no MPI: photon_loss_all = photon_loss) */
void orc_pass_all_sources(const orc_tables *tb, const orc_step *st, orc_state *s) {
  const size_t ncell = (size_t)st->mesh[0] * st->mesh[1] * st->mesh[2];
  memset(s->phih, 0, ncell * sizeof(double));
  memset(s->phihe, 0, 2 * ncell * sizeof(double));
  memset(s->phiheat, 0, ncell * sizeof(double));
  memset(s->photon_loss, 0, sizeof(s->photon_loss));
  s->sum_nbox = 0;
  for (int ns = 1; ns <= st->nsrc; ns++) {
    double loss;
    int nbox = orc_do_source(tb, st, s, ns, &loss);
    s->photon_loss[0] = s->photon_loss[0] + loss;
    s->sum_nbox += nbox;
  }
}

/* ---------------------------------------------------------------------------------------------
 * files_for_3D/evolve_point.F90:444-646 do_chemistry (local = .false.) and :325-440 evolve0D_global
 */
static void do_chemistry(const orc_tables *tb, const orc_step *st, orc_state *s, double dt, double ndens_p,
                         orc_ionstates *ion, const orc_photrates *phi, size_t q) {
  const size_t ncell = (size_t)st->mesh[0] * st->mesh[1] * st->mesh[2];
  double temper_inter, avg_temper, temper1, temper0, temper2;
  orc_reccoef rc = st->rc;
  if (st->isothermal) {
    temper_inter = avg_temper = temper1 = st->temper_val;
  } else {
    temper_inter = (double)s->temperature[q];
    avg_temper = (double)s->temperature[q + ncell];
    temper1 = (double)s->temperature[q + 2 * ncell];
  }
  (void)temper_inter;
  temper0 = temper1;
  const double path = 1.0;
  /* :483-484 clumping_point when type_of_clumping == 5 */
  const float clumping = st->clumping_grid ? st->clumping_grid[q] : st->clumping;
  int nit = 0;
  for (;;) {
    nit++;
    temper2 = temper1;
    double yh0_av_old = ion->h_av[0];
    double yhe0_av_old = ion->he_av[0];
    double yhe2_av_old = ion->he_av[2];
    double de = orc_electrondens(ndens_p, ion->h_av, ion->he_av);
    if (!st->isothermal) orc_ini_rec_colion_factors(avg_temper, &rc);

    double coldensh_cell = coldens(path, ion->h[0], ndens_p, (1.0 - abu_he));
    double coldenshe_cell[2];
    coldenshe_cell[0] = coldens(path, ion->he[0], ndens_p, abu_he);
    coldenshe_cell[1] = coldens(path, ion->he[1], ndens_p, abu_he);
    double yfrac, zfrac, y2afrac, y2bfrac;
    orc_prepare_doric_factors(coldensh_cell, coldenshe_cell, &yfrac, &zfrac, &y2afrac, &y2bfrac);
    orc_doric(dt, de, ndens_p, ion, phi, yfrac, zfrac, y2afrac, y2bfrac, &rc, clumping);
    de = orc_electrondens(ndens_p, ion->h_av, ion->he_av);

    coldensh_cell = coldens(path, ion->h[0], ndens_p, (1.0 - abu_he));
    coldenshe_cell[0] = coldens(path, ion->he[0], ndens_p, abu_he);
    coldenshe_cell[1] = coldens(path, ion->he[1], ndens_p, abu_he);
    orc_prepare_doric_factors(coldensh_cell, coldenshe_cell, &yfrac, &zfrac, &y2afrac, &y2bfrac);

    double ionh0old = ion->h[0], ionh1old = ion->h[1];
    double ionhe0old = ion->he[0], ionhe1old = ion->he[1], ionhe2old = ion->he[2];
    double oldhav = ion->h_av[0], oldhe0av = ion->he_av[0], oldhe1av = ion->he_av[1];

    orc_doric(dt, de, ndens_p, ion, phi, yfrac, zfrac, y2afrac, y2bfrac, &rc, clumping);

    ion->h[0] = (ion->h[0] + ionh0old) / 2.0;
    ion->h[1] = (ion->h[1] + ionh1old) / 2.0;
    ion->he[0] = (ion->he[0] + ionhe0old) / 2.0;
    ion->he[1] = (ion->he[1] + ionhe1old) / 2.0;
    ion->he[2] = (ion->he[2] + ionhe2old) / 2.0;
    ion->h_av[0] = (ion->h_av[0] + oldhav) / 2.0;
    ion->he_av[0] = (ion->he_av[0] + oldhe0av) / 2.0;
    ion->he_av[1] = (ion->he_av[1] + oldhe1av) / 2.0;

    de = orc_electrondens(ndens_p, ion->h_av, ion->he_av);
    temper1 = temper0;
    if (!st->isothermal)
      orc_thermal(tb, dt, &temper1, &avg_temper, de, ndens_p, ion, phi, st->zred, st->H0, st->Omega0);

    if ((fabs((ion->h_av[0] - yh0_av_old) / ion->h_av[0]) < minimum_fractional_change ||
         ion->h_av[0] < minimum_fraction_of_atoms) &&
        (fabs((ion->he_av[0] - yhe0_av_old) / ion->he_av[0]) < minimum_fractional_change ||
         ion->he_av[0] < minimum_fraction_of_atoms) &&
        (fabs((ion->he_av[2] - yhe2_av_old) / ion->he_av[2]) < minimum_fractional_change ||
         ion->he_av[2] < minimum_fraction_of_atoms) &&
        fabs((temper1 - temper2) / temper1) < minimum_fractional_change)
      break;
    if (nit > 400) break;
  }
  if (!st->isothermal) { /* set_temperature_point, mat_ini_test.F90:491-502 */
    s->temperature[q] = (float)temper1;
    s->temperature[q + ncell] = (float)avg_temper;
  }
}

static void evolve0D_global(const orc_tables *tb, const orc_step *st, orc_state *s, double dt, size_t q,
                            int *conv_flag) {
  const size_t ncell = (size_t)st->mesh[0] * st->mesh[1] * st->mesh[2];
  orc_ionstates ion;
  orc_photrates phi;
  memset(&phi, 0, sizeof(phi));
  for (int nx = 0; nx < 2; nx++) {
    ion.h[nx] = dmax(epsilon, s->xh_intermed[q + nx * ncell]);
    ion.h_old[nx] = dmax(epsilon, s->xh[q + nx * ncell]);
    ion.h_av[nx] = dmax(epsilon, s->xh_av[q + nx * ncell]);
  }
  for (int nx = 0; nx < 3; nx++) {
    ion.he[nx] = dmax(epsilon, s->xhe_intermed[q + nx * ncell]);
    ion.he_old[nx] = dmax(epsilon, s->xhe[q + nx * ncell]);
    ion.he_av[nx] = dmax(epsilon, s->xhe_av[q + nx * ncell]);
  }
  double ndens_p = st->ndens[q];
  double temp_av_old = st->isothermal ? st->temper_val : (double)s->temperature[q + ncell];
  phi.photo_cell_HI = s->phih[q];
  phi.photo_cell_HeI = s->phihe[q];
  phi.photo_cell_HeII = s->phihe[q + ncell];
  if (!st->isothermal) phi.heat = s->phiheat[q];

  do_chemistry(tb, st, s, dt, ndens_p, &ion, &phi, q);

  double yh0_av_old = s->xh_av[q];
  double yhe0_av_old = s->xhe_av[q];
  double yhe2_av_old = s->xhe_av[q + 2 * ncell];
  double temp_av_new = st->isothermal ? st->temper_val : (double)s->temperature[q + ncell];
  const double mfc = minimum_fractional_change, mfa = minimum_fraction_of_atoms;
  if ((fabs(ion.h_av[0] - yh0_av_old) > mfc && fabs((ion.h_av[0] - yh0_av_old) / ion.h_av[0]) > mfc &&
       ion.h_av[0] > mfa) ||
      (fabs(ion.he_av[0] - yhe0_av_old) > mfc && fabs((ion.he_av[0] - yhe0_av_old) / ion.he_av[0]) > mfc &&
       ion.he_av[0] > mfa) ||
      (fabs(ion.he_av[2] - yhe2_av_old) > mfc && fabs((ion.he_av[2] - yhe2_av_old) / ion.he_av[2]) > mfc &&
       ion.he_av[2] > mfa) ||
      (fabs((temp_av_old - temp_av_new) / temp_av_new) > 1.0e-1 && fabs(temp_av_new - temp_av_old) > 100.0))
    (*conv_flag)++;
  for (int nx = 0; nx < 2; nx++) {
    s->xh_intermed[q + nx * ncell] = ion.h[nx];
    s->xh_av[q + nx * ncell] = ion.h_av[nx];
  }
  for (int nx = 0; nx < 3; nx++) {
    s->xhe_intermed[q + nx * ncell] = ion.he[nx];
    s->xhe_av[q + nx * ncell] = ion.he_av[nx];
  }
}

int orc_global_pass(const orc_tables *tb, const orc_step *st, orc_state *s, double dt) {
  const size_t ncell = (size_t)st->mesh[0] * st->mesh[1] * st->mesh[2];
  int conv_flag = 0;
  for (size_t q = 0; q < ncell; q++) evolve0D_global(tb, st, s, dt, q, &conv_flag);
  return conv_flag;
}

/* the same over `nthreads` OpenMP threads: cells are independent, the count is an integer sum */
int orc_global_pass_threads(const orc_tables *tb, const orc_step *st, orc_state *s, double dt, int nthreads) {
  const long long ncell = (long long)st->mesh[0] * st->mesh[1] * st->mesh[2];
  int conv_flag = 0;
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for schedule(dynamic, 4096) num_threads(nthreads) reduction(+ : conv_flag)
  for (long long q = 0; q < ncell; q++) {
    int c = 0;
    evolve0D_global(tb, st, s, dt, (size_t)q, &c);
    conv_flag += c;
  }
  return conv_flag;
}

/* files_for_3D/evolve.F90:78-229, restart == 0 */
int orc_evolve3d(const orc_tables *tb, const orc_step *st, orc_state *s, double dt, int max_iter) {
  const size_t ncell = (size_t)st->mesh[0] * st->mesh[1] * st->mesh[2];
  memcpy(s->xh_av, s->xh, 2 * ncell * sizeof(double));
  memcpy(s->xh_intermed, s->xh, 2 * ncell * sizeof(double));
  memcpy(s->xhe_av, s->xhe, 3 * ncell * sizeof(double));
  memcpy(s->xhe_intermed, s->xhe, 3 * ncell * sizeof(double));
  int niter = 0;
  int conv_flag = (int)ncell;
  /* :147 -- REAL(4) product convergence_fraction(dp)*mesh... is dp*int */
  int conv_criterion = imin((int)(convergence_fraction * st->mesh[0] * st->mesh[1] * st->mesh[2]), st->nsrc);
  for (;;) {
    if (conv_flag < conv_criterion && niter > 1) {
      memcpy(s->xh, s->xh_intermed, 2 * ncell * sizeof(double));
      memcpy(s->xhe, s->xhe_intermed, 3 * ncell * sizeof(double));
      if (!st->isothermal) /* set_final_temperature_point */
        memcpy(s->temperature + 2 * ncell, s->temperature, ncell * sizeof(float));
      break;
    } else if (niter > 500) {
      break;
    }
    if (max_iter > 0 && niter >= max_iter) break; /* test hook: stop early without the final copy */
    niter++;
    if (st->nsrc > 0) {
      orc_pass_all_sources(tb, st, s);
    } else {
      memset(s->phih, 0, ncell * sizeof(double));
      memset(s->phihe, 0, 2 * ncell * sizeof(double));
      memset(s->phiheat, 0, ncell * sizeof(double));
      memset(s->photon_loss, 0, sizeof(s->photon_loss));
    }
    conv_flag = orc_global_pass(tb, st, s, dt);
    if (niter <= 512) s->conv_flags[niter - 1] = conv_flag;
  }
  s->niter = niter;
  return niter;
}

/* ---------------------------------------------------------------------------------------------
 * do_source in L-infinity shell order, the cells of a shell in parallel (OpenMP).
 *
 * Not how the reference orders its sweep (evolve2D / the 8-way OpenMP decomposition of
 * evolve_source.F90:158-189), but the order the HIP kernels use: every upstream corner cinterp reads lies
 * in a smaller shell around the source, or in the same shell with an interpolation weight that is
 * exactly 0.  Columns and rates therefore come out bit for bit as in the serial sweep (each cell is
 * written once, from inputs that are complete); only the photon loss, a sum over boundary cells, is
 * added up in a different order.  Used (a) as a CPU proof of that statement (tests/) and (b) as the
 * all-cores CPU baseline of bench.py.  (A same-shell corner may be read while another thread writes it;
 * its weight is 0 and the value finite either way.)
 */
typedef struct { int p[3]; } cellpos;

int orc_do_source_shells(const orc_tables *tb, const orc_step *st, orc_state *s, int ns, double *loss_out, int nthreads) {
  const size_t ncell = (size_t)st->mesh[0] * st->mesh[1] * st->mesh[2];
  const int *src = st->srcpos + 3 * (ns - 1);
  int lastpos_r[3], lastpos_l[3];
  memset(s->coldensh_out, 0, ncell * sizeof(double));
  memset(s->coldenshe_out, 0, 2 * ncell * sizeof(double));
  int smax_all = 0;
  for (int d = 0; d < 3; d++) {
    lastpos_r[d] = src[d] + imin(MAX_SUBBOX, st->mesh[d] / 2 - 1 + st->mesh[d] % 2);
    lastpos_l[d] = src[d] - imin(MAX_SUBBOX, st->mesh[d] / 2);
    smax_all = imax(smax_all, st->mesh[d] / 2);
  }
  cellpos *cells = malloc(sizeof(cellpos) * (size_t)(24 * (size_t)smax_all * smax_all + 2));
  if (nthreads < 1) nthreads = 1;
  double *tloss = malloc(sizeof(double) * (size_t)nthreads);
  int nbox = 0, s_done = -1;
  double total_source_flux = st->normflux[ns - 1] * st->s_star;
  if (st->normflux_pl) total_source_flux = total_source_flux + st->normflux_pl[ns - 1] * st->pl_s_star;
  if (st->normflux_qpl) total_source_flux = total_source_flux + st->normflux_qpl[ns - 1] * st->qpl_s_star;
  double photon_loss_src = total_source_flux;
  int last_r[3], last_l[3];
  for (int d = 0; d < 3; d++) { last_r[d] = src[d]; last_l[d] = src[d]; }
  while (photon_loss_src > F(1e-10) * total_source_flux && last_r[2] < lastpos_r[2] && last_l[2] > lastpos_l[2]) {
    nbox++;
    photon_loss_src = 0.0;
    int smax = 0;
    for (int d = 0; d < 3; d++) {
      last_r[d] = imin(src[d] + SUBBOXSIZE * nbox, lastpos_r[d]);
      last_l[d] = imax(src[d] - SUBBOXSIZE * nbox, lastpos_l[d]);
      smax = imax(smax, imax(last_r[d] - src[d], src[d] - last_l[d]));
    }
    for (int sh = s_done + 1; sh <= smax; sh++) {
      /* the cells of shell sh inside the box */
      int n = 0;
      for (int dk = imax(-sh, last_l[2] - src[2]); dk <= imin(sh, last_r[2] - src[2]); dk++)
        for (int dj = imax(-sh, last_l[1] - src[1]); dj <= imin(sh, last_r[1] - src[1]); dj++) {
          const int face = (dk == sh || dk == -sh || dj == sh || dj == -sh);
          const int ilo = imax(-sh, last_l[0] - src[0]), ihi = imin(sh, last_r[0] - src[0]);
          if (face) {
            for (int di = ilo; di <= ihi; di++) { cells[n].p[0] = src[0] + di; cells[n].p[1] = src[1] + dj; cells[n].p[2] = src[2] + dk; n++; }
          } else {
            if (-sh >= ilo) { cells[n].p[0] = src[0] - sh; cells[n].p[1] = src[1] + dj; cells[n].p[2] = src[2] + dk; n++; }
            if (sh <= ihi && sh != 0) { cells[n].p[0] = src[0] + sh; cells[n].p[1] = src[1] + dj; cells[n].p[2] = src[2] + dk; n++; }
          }
        }
      for (int t = 0; t < nthreads; t++) tloss[t] = 0.0;
#pragma omp parallel num_threads(nthreads)
      {
        sweep_ctx cx;
        for (int d = 0; d < 3; d++) { cx.last_r[d] = last_r[d]; cx.last_l[d] = last_l[d]; }
        cx.photon_loss_src_thread = 0.0;
#pragma omp for schedule(static)
        for (int c = 0; c < n; c++) evolve0D(tb, st, s, cells[c].p, ns, &cx);
        int me = 0;
#ifdef _OPENMP
        me = omp_get_thread_num();
#endif
        tloss[me] = cx.photon_loss_src_thread;
      }
      for (int t = 0; t < nthreads; t++) photon_loss_src = photon_loss_src + tloss[t];
    }
    s_done = smax;
  }
  free(cells);
  free(tloss);
  if (loss_out) *loss_out = photon_loss_src;
  return nbox;
}

/* set_rates_to_zero + all sources through orc_do_source_shells */
void orc_pass_all_sources_shells(const orc_tables *tb, const orc_step *st, orc_state *s, int nthreads) {
  const size_t ncell = (size_t)st->mesh[0] * st->mesh[1] * st->mesh[2];
  memset(s->phih, 0, ncell * sizeof(double));
  memset(s->phihe, 0, 2 * ncell * sizeof(double));
  memset(s->phiheat, 0, ncell * sizeof(double));
  memset(s->photon_loss, 0, sizeof(s->photon_loss));
  s->sum_nbox = 0;
  for (int ns = 1; ns <= st->nsrc; ns++) {
    double loss = 0.0;
    int nbox = orc_do_source_shells(tb, st, s, ns, &loss, nthreads);
    s->photon_loss[0] = s->photon_loss[0] + loss;
    s->sum_nbox += nbox;
  }
}

/* ---------------------------------------------------------------------------------------------
 * radiation_tables.f90:172-422 spec_integration for one SED, from what spectrum_parms /
 * setup_scalingfactors / romberg_initialisation / normalize_seds leave behind (orc_sed_setup).
 * Integrands: fill_photo_integrands :462-536, fill_heating_integrands_* :540-783; integration:
 * Vector_Romberg, romberg.f90:158-188 (serial sum over the 513 frequencies, weight delta_freq).
 * Tables are (0:NumTau, ncol), tau index fastest, like the reference's.
 */
static double sed_photo_integrand(const orc_sed_setup *S, double freq, double csfd, double tau, int thin) {
  /* :471 the cut-off that avoids underflow of exp(-tau*...) ; 700.0 is a REAL(4) literal (exact) */
  if (!(tau * csfd < 700.0)) return 0.0;
  if (S->sed == 0) {
    if (!(freq * S->h_over_kT < 700.0)) return 0.0; /* :474 */
    double t = 4.0 * S->pi * S->R_star2 * S->two_pi_over_c_square * freq * freq;
    if (thin) t = t * csfd;
    t = t * exp(-tau * csfd);
    return t / (exp(freq * S->h_over_kT) - 1.0);
  }
  double t = S->pl_scaling * pow(freq, -S->pl_index); /* :491-510 */
  if (thin) t = t * csfd;
  return t * exp(-tau * csfd);
}

void orc_build_tables(const orc_sed_setup *S, int heat, double *photo_thick, double *photo_thin,
                      double *heat_thick, double *heat_thin) {
  const int nf = S->nfreq, nt = ORC_NTAU + 1;
  double *freq = malloc((size_t)(nf + 1) * sizeof(double)), *csfd = malloc((size_t)(nf + 1) * sizeof(double));
  for (int b = 1; b <= NB1 + NB2 + NB3; b++) {
    const double fmin = S->freq_min[b - 1], df = S->delta_freq[b - 1];
    for (int i = 0; i <= nf; i++) { /* set_frequency_array :438, set_cross_section_freq_dependence :449 */
      freq[i] = fmin + df * (double)(float)i;
      csfd[i] = pow(freq[i] / fmin, -S->xsec_index[b - 1]);
    }
    /* heating columns of this band and their threshold frequencies (:300-310, :343-390) */
    int hcol[3], nh;
    const double f0[3] = {S->ion_freq_HI, S->ion_freq_HeI, S->ion_freq_HeII};
    if (b <= NB1) { nh = 1; hcol[0] = 1; }
    else if (b <= NB1 + NB2) { nh = 2; hcol[0] = b * 2 - NB1 - 1; hcol[1] = b * 2 - NB1; }
    else { nh = 3; hcol[0] = b * 3 - NB2 - NB1 * 2 - 2; hcol[1] = hcol[0] + 1; hcol[2] = hcol[0] + 2; }
    for (int it = 0; it < nt; it++) {
      const double tau = S->tau[it];
      for (int thin = 0; thin < 2; thin++) {
        double itg = 0.0, itgh[3] = {0.0, 0.0, 0.0};
        for (int x = 0; x <= nf; x++) {
          const double f = sed_photo_integrand(S, freq[x], csfd[x], tau, thin);
          itg = itg + f * df * S->romw[x];
          if (heat)
            for (int k = 0; k < nh; k++) {
              const double fh = S->hplanck * (freq[x] - f0[k]) * f;
              itgh[k] = itgh[k] + fh * df * S->romw[x];
            }
        }
        (thin ? photo_thin : photo_thick)[(size_t)(b - 1) * nt + it] = itg;
        if (heat)
          for (int k = 0; k < nh; k++) (thin ? heat_thin : heat_thick)[(size_t)(hcol[k] - 1) * nt + it] = itgh[k];
      }
    }
  }
  free(freq);
  free(csfd);
}
