// TEST-ONLY: the product's device functions (c2-ray3dm1d_helium_amd/csrc/c2ray_device.hpp, c2ray_shell.hpp) ONE ROUTINE AT
// A TIME on the GPU, on arrays of inputs -- the device-side twin of tests/host_harness.cpp.  tests/test_gpu_functions.py feeds it
// the vectors the REFERENCE's compiled routines produced (tests/golden/funcvec.npz: photoion_rates x800, doric x300,
// thermal x200, ini_rec_colion_factors x40; oracle/probe/evolve_tap.f90) and the reference's own column grids for cinterp,
// so that a failing whole-call test can be bisected to a routine on the device.  Nothing in the product links this file.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -o _gpu_func_probe.so gpu_func_probe.hip
// `mode` of the photo-ionisation probes: how the kernel under test reaches the bit-exact log's table --
//   0: the table in global memory (k_evolve0d, k_chemistry's first tier)
//   1: gm::LogEntry table in LDS + two pinned constants (k_rates<isothermal>)
//   2: gm::LogEntry table in LDS, no pins (k_rates<heating>)
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "../c2-ray3dm1d_helium_amd/csrc/c2ray_device.hpp"
#include "../c2-ray3dm1d_helium_amd/csrc/c2ray_shell.hpp"

using namespace c2r;

namespace {

constexpr int PB = 256; // threads per block: the LDS log table has 256 entries, one per thread (as in k_rates)

struct Dev {
  BandDataByRow h_bd{};
  BandDataByRow *bd = nullptr;
  double *pthick = nullptr, *pthin = nullptr, *hthick = nullptr, *hthin = nullptr, *hthick_il = nullptr, *hthin_il = nullptr;
  double *cool = nullptr;
  double mintemp = 1.0, dtemp = 0.01;
  double *s_pt[2] = {nullptr, nullptr}, *s_pn[2] = {nullptr, nullptr}, *s_ht_il[2] = {nullptr, nullptr}, *s_hn_il[2] = {nullptr, nullptr};
  int s_lo[2] = {0, 0}, s_hi[2] = {0, 0};
} D;

void pitch(const double *src, int ncol, std::vector<double> &dst) {
  dst.assign((size_t)ncol * NTAUP, 0.0);
  for (int c = 0; c < ncol; c++) {
    std::memcpy(&dst[(size_t)c * NTAUP], src + (size_t)c * (NTAU + 1), sizeof(double) * (NTAU + 1));
    dst[(size_t)c * NTAUP + NTAU + 1] = src[(size_t)c * (NTAU + 1) + NTAU];
  }
}
void set_tau_zero(int sed, const std::vector<double> &pt, const std::vector<double> &pn, const std::vector<double> &ht,
                  const std::vector<double> &hn) {
  for (int b = 0; b < NFREQ; b++) {
    const double *cols[8];
    int n = 0;
    cols[n++] = &pt[(size_t)b * NTAUP];
    cols[n++] = &pn[(size_t)b * NTAUP];
    const int nh = heat_species(b), c0 = heat_first_col(b);
    for (int k = 0; k < nh; k++) {
      cols[n++] = &ht[(size_t)(c0 + k) * NTAUP];
      cols[n++] = &hn[(size_t)(c0 + k) * NTAUP];
    }
    D.h_bd.tau_zero[sed][b] = band_tau_zero(cols, n);
  }
}
int up(double **d, const std::vector<double> &h) {
  if (*d) (void)hipFree(*d);
  if (hipMalloc(d, sizeof(double) * h.size()) != hipSuccess) return 1;
  return hipMemcpy(*d, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice) != hipSuccess;
}
int up_bd() {
  if (!D.bd && hipMalloc(&D.bd, sizeof(BandDataByRow)) != hipSuccess) return 1;
  return hipMemcpy(D.bd, &D.h_bd, sizeof(BandDataByRow), hipMemcpyHostToDevice) != hipSuccess;
}

// device arrays for one call: inputs copied in, outputs copied back, everything freed on the way out
struct Arg {
  void *d = nullptr;
  void *h = nullptr;
  size_t bytes = 0;
  bool out = false;
};
struct Call {
  std::vector<Arg> a;
  bool bad = false;
  template <class T>
  T *in(const T *h, size_t n) { return static_cast<T *>(add(const_cast<T *>(h), sizeof(T) * n, false, true)); }
  template <class T>
  T *out(T *h, size_t n) { return static_cast<T *>(add(h, sizeof(T) * n, true, false)); }
  template <class T>
  T *inout(T *h, size_t n) { return static_cast<T *>(add(h, sizeof(T) * n, true, true)); }
  void *add(void *h, size_t bytes, bool out, bool copy_in) {
    Arg x;
    x.h = h; x.bytes = bytes; x.out = out;
    if (hipMalloc(&x.d, bytes ? bytes : 8) != hipSuccess) { bad = true; return nullptr; }
    if (copy_in && bytes && hipMemcpy(x.d, h, bytes, hipMemcpyHostToDevice) != hipSuccess) bad = true;
    a.push_back(x);
    return x.d;
  }
  int finish() {
    if (hipDeviceSynchronize() != hipSuccess) bad = true;
    if (hipGetLastError() != hipSuccess) bad = true;
    for (Arg &x : a) {
      if (x.out && !bad && x.bytes && hipMemcpy(x.h, x.d, x.bytes, hipMemcpyDeviceToHost) != hipSuccess) bad = true;
      (void)hipFree(x.d);
    }
    return bad ? 1 : 0;
  }
};
dim3 grid(int n) { return dim3((unsigned)((n + PB - 1) / PB)); }

__global__ void k_reccoef(int n, const double *T, double *out12) {
  const int i = blockIdx.x * PB + threadIdx.x;
  if (i >= n) return;
  RecCoef rc;
  ini_rec_colion_factors(T[i], rc);
  const double *p = reinterpret_cast<const double *>(&rc);
  for (int k = 0; k < 12; k++) out12[12 * i + k] = p[k];
}

template <bool HEAT>
__global__ void __launch_bounds__(PB) k_photoion(int n, int mode, const BandData *bd, const double *pthick, const double *pthin,
                                                 const double *hthick, const double *hthin, const double *cin6, const double *vol,
                                                 const double *nflux, const double *istate, double *out5) {
  __shared__ gm::LogEntry s_logtab[256];
  s_logtab[threadIdx.x] = gm::make_log_entry((int)threadIdx.x);
  __syncthreads();
  const int i = blockIdx.x * PB + threadIdx.x;
  if (i >= n) return;
  const double *c = cin6 + 6 * i;
  Ricotti ric = {};
  if (HEAT) ric = ricotti_parameters(istate[i]);
  PhotoOut o;
  if (mode == 0) {
    photoion_rates<HEAT>(*bd, pthick, pthin, hthick, hthin, c[0], c[1], c[2], c[3], c[4], c[5], vol[i], nflux[i], ric, o);
  } else {
    gm::LogPins pins_ = {0.0, 0.0};
    const gm::LogPins *pins = nullptr;
    if (mode == 1) {
      pins_ = gm::pin_log_constants();
      pins = &pins_;
    }
    photoion_rates<HEAT, gm::LogEntry>(*bd, pthick, pthin, hthick, hthin, c[0], c[1], c[2], c[3], c[4], c[5], vol[i], nflux[i], ric,
                                       o, s_logtab, pins);
  }
  double *r = out5 + 5 * i;
  r[0] = o.photo_HI; r[1] = o.photo_HeI; r[2] = o.photo_HeII; r[3] = o.heat; r[4] = o.photo_out;
}

template <bool HEAT, class BD>
__global__ void __launch_bounds__(PB) k_photoion_multi(int n, int mode, const BD *bd, SedSet ss, const double *cin6, const double *vol,
                                                       const double *nflux3, const double *istate, double *out5) {
  __shared__ gm::LogEntry s_logtab[256];
  s_logtab[threadIdx.x] = gm::make_log_entry((int)threadIdx.x);
  __syncthreads();
  const int i = blockIdx.x * PB + threadIdx.x;
  if (i >= n) return;
  const double *c = cin6 + 6 * i;
  Ricotti ric = {};
  if (HEAT) ric = ricotti_parameters(istate[i]);
  PhotoOut o;
  const double nf[NSED] = {nflux3[3 * i], nflux3[3 * i + 1], nflux3[3 * i + 2]};
  if (mode == 0) photoion_rates_multi<HEAT>(*bd, ss, c[0], c[1], c[2], c[3], c[4], c[5], vol[i], nf, ric, o);
  else photoion_rates_multi<HEAT, gm::LogEntry>(*bd, ss, c[0], c[1], c[2], c[3], c[4], c[5], vol[i], nf, ric, o, s_logtab);
  double *r = out5 + 5 * i;
  r[0] = o.photo_HI; r[1] = o.photo_HeI; r[2] = o.photo_HeII; r[3] = o.heat; r[4] = o.photo_out;
}

__global__ void k_photo_out(int n, const BandData *bd, SedSet ss, int multi, const double *cin6, const double *nflux3, double *out) {
  const int i = blockIdx.x * PB + threadIdx.x;
  if (i >= n) return;
  const double *c = cin6 + 6 * i;
  const double nf[NSED] = {nflux3[3 * i], nflux3[3 * i + 1], nflux3[3 * i + 2]};
  out[i] = multi ? photo_out_multi(*bd, ss, c[0], c[1], c[2], c[3], c[4], c[5], nf)
                 : photo_out_only(*bd, ss.photo_thick[0], ss.photo_thin[0], c[0], c[1], c[2], c[3], c[4], c[5], nf[0]);
}

__global__ void k_doric(int n, const double *dt, const double *de, double *ion15, const double *phi3, const double *fr4,
                        const double *rc12, const double *clumping) {
  const int i = blockIdx.x * PB + threadIdx.x;
  if (i >= n) return;
  IonStates ion;
  RecCoef rc;
  double *pi_ = reinterpret_cast<double *>(&ion), *pr = reinterpret_cast<double *>(&rc);
  for (int k = 0; k < 15; k++) pi_[k] = ion15[15 * i + k];
  for (int k = 0; k < 12; k++) pr[k] = rc12[12 * i + k];
  doric(dt[i], de[i], ion, phi3[3 * i], phi3[3 * i + 1], phi3[3 * i + 2], fr4[4 * i], fr4[4 * i + 1], fr4[4 * i + 2], fr4[4 * i + 3],
        rc, clumping[i]);
  for (int k = 0; k < 15; k++) ion15[15 * i + k] = pi_[k];
}

__global__ void k_doric_factors(int n, const double *N3, double *out4) {
  const int i = blockIdx.x * PB + threadIdx.x;
  if (i >= n) return;
  prepare_doric_factors(N3[3 * i], N3[3 * i + 1], N3[3 * i + 2], out4[4 * i], out4[4 * i + 1], out4[4 * i + 2], out4[4 * i + 3]);
}

// lds != 0: cooling curves and the log's table in LDS, as k_chemistry<true, true> (the repacked heating tiers) holds them
__global__ void __launch_bounds__(PB) k_thermal(int n, int lds, CoolData cd, const double *dt, double *tend, double *tavg,
                                                const double *de, const double *nd, const double *ion15, const double *heat) {
  __shared__ double s_cool[5 * NCOOL];
  __shared__ double s_log[256];
  if (lds) {
    for (int k = (int)threadIdx.x; k < 5 * NCOOL; k += PB) s_cool[k] = cd.cool[k];
    s_log[threadIdx.x] = gm::log_table()[threadIdx.x];
    __syncthreads();
    cd.cool = s_cool;
    cd.logtab = s_log;
  }
  const int i = blockIdx.x * PB + threadIdx.x;
  if (i >= n) return;
  IonStates ion;
  double *pi_ = reinterpret_cast<double *>(&ion);
  for (int k = 0; k < 15; k++) pi_[k] = ion15[15 * i + k];
  double te = tend[i], ta = -1.0;
  thermal(cd, dt[i], te, ta, de[i], nd[i], ion, heat[i]);
  tend[i] = te;
  tavg[i] = ta;
}

// cinterp (column_density.f90:28-345) for every offset of a mesh around one source, from mesh-ordered column grids:
// short_characteristic + interp_column as k_sweep_shell runs them (shells 0 and 1, and every shell under
// C2R_SWEEP_GENERIC) and, for shells >= 2, the per-shell form of k_sweep_shell_fast (shell_decode_fast,
// shell_short_characteristic, interp_column_fast) -- whose corners are positions in the shell-ordered arrays, read here
// through the inverse map shell_decode.  out: 4 doubles per offset from the general path, 4 from the fast path
// (copies of the general ones where the fast path does not apply).
__global__ void k_cinterp(int n1, int n2, int n3, const double *cH, const double *cHe, int i0, int j0, int k0, const ShellGeom *geom,
                          int smax_fast, double *out8) {
  const size_t nc = (size_t)n1 * n2 * n3;
  const size_t t = (size_t)blockIdx.x * PB + threadIdx.x;
  if (t >= nc) return;
  const int lo1 = -(n1 / 2), lo2 = -(n2 / 2), lo3 = -(n3 / 2);
  const int di = lo1 + (int)(t % n1), dj = lo2 + (int)((t / n1) % n2), dk = lo3 + (int)(t / ((size_t)n1 * n2));
  double *o = out8 + 8 * t;
  for (int k = 0; k < 8; k++) o[k] = 0.0;
  if (di == 0 && dj == 0 && dk == 0) return;
  auto cell = [&](int ci, int cj, int ck) {
    const int i = ((i0 - 1 + ci) % n1 + n1) % n1, j = ((j0 - 1 + cj) % n2 + n2) % n2, k = ((k0 - 1 + ck) % n3 + n3) % n3;
    return (size_t)i + (size_t)n1 * ((size_t)j + (size_t)n2 * k);
  };
  ShortChar s4;
  short_characteristic(i0, j0, k0, di, dj, dk, s4);
  size_t qc[4];
  for (int c = 0; c < 4; c++) qc[c] = cell(s4.ci[c], s4.cj[c], s4.ck[c]);
  o[0] = interp_column(s4, cH[qc[0]], cH[qc[1]], cH[qc[2]], cH[qc[3]], sigma_HI_at_ion_freq);
  o[1] = interp_column(s4, cHe[qc[0]], cHe[qc[1]], cHe[qc[2]], cHe[qc[3]], sigma_HeI_at_ion_freq);
  o[2] = interp_column(s4, cHe[nc + qc[0]], cHe[nc + qc[1]], cHe[nc + qc[2]], cHe[nc + qc[3]], sigma_HeII_at_ion_freq);
  o[3] = s4.path;
  for (int k = 0; k < 4; k++) o[4 + k] = o[k];
  const int ia = di < 0 ? -di : di, ja = dj < 0 ? -dj : dj, ka = dk < 0 ? -dk : dk;
  const int s = ia > ja ? (ia > ka ? ia : ka) : (ja > ka ? ja : ka);
  if (s < 2 || s > smax_fast) return;
  const ShellGeom G = geom[s];
  const int tt = (int)((long long)shell_position(di, dj, dk) - G.off);
  int fi, fj, fk;
  const int face = shell_decode_fast(G, tt, fi, fj, fk);
  if (fi != di || fj != dj || fk != dk) { o[4] = o[5] = o[6] = o[7] = -1.0; return; } // the thread -> cell map is off
  ShellCorners sc;
  shell_short_characteristic(G, face, i0, j0, k0, di, dj, dk, sc);
  size_t qf[4];
  for (int c = 0; c < 4; c++) {
    int ci, cj, ck;
    shell_decode(s - 1, (int)((long long)sc.p[c] - G.offp), ci, cj, ck);
    qf[c] = cell(ci, cj, ck);
  }
  o[4] = interp_column_fast(sc.s, cH[qf[0]], cH[qf[1]], cH[qf[2]], cH[qf[3]], sigma_HI_at_ion_freq);
  o[5] = interp_column_fast(sc.s, cHe[qf[0]], cHe[qf[1]], cHe[qf[2]], cHe[qf[3]], sigma_HeI_at_ion_freq);
  o[6] = interp_column_fast(sc.s, cHe[nc + qf[0]], cHe[nc + qf[1]], cHe[nc + qf[2]], cHe[nc + qf[3]], sigma_HeII_at_ion_freq);
  o[7] = sc.path;
}

SedSet make_sedset() {
  SedSet ss;
  ss.photo_thick[0] = D.pthick; ss.photo_thin[0] = D.pthin;
  ss.heat_thick[0] = D.hthick_il; ss.heat_thin[0] = D.hthin_il;
  ss.lo[0] = 0; ss.hi[0] = D.h_bd.bb_upper;
  for (int k = 0; k < 2; k++) {
    ss.photo_thick[k + 1] = D.s_pt[k]; ss.photo_thin[k + 1] = D.s_pn[k];
    ss.heat_thick[k + 1] = D.s_ht_il[k]; ss.heat_thin[k + 1] = D.s_hn_il[k];
    ss.lo[k + 1] = D.s_lo[k]; ss.hi[k + 1] = D.s_hi[k];
  }
  return ss;
}

} // namespace

extern "C" {

int fp_set_tables(const double *pthick, const double *pthin, const double *hthick, const double *hthin, const double *sHI,
                  const double *sHeI, const double *sHeII, const double *const f[12], int bb_upper, const double *cool,
                  double mintemp, double dtemp) {
  std::memset(&D.h_bd, 0, sizeof D.h_bd);
  std::vector<double> pt, pn, ht, hn;
  pitch(pthick, NFREQ, pt);
  pitch(pthin, NFREQ, pn);
  pitch(hthick, NHEAT, ht);
  pitch(hthin, NHEAT, hn);
  std::memcpy(D.h_bd.sigma_HI, sHI, sizeof D.h_bd.sigma_HI);
  std::memcpy(D.h_bd.sigma_HeI, sHeI, sizeof D.h_bd.sigma_HeI);
  std::memcpy(D.h_bd.sigma_HeII, sHeII, sizeof D.h_bd.sigma_HeII);
  double *dst[12] = {D.h_bd.f1ion_HI, D.h_bd.f1ion_HeI, D.h_bd.f1ion_HeII, D.h_bd.f2ion_HI, D.h_bd.f2ion_HeI, D.h_bd.f2ion_HeII,
                     D.h_bd.f1heat_HI, D.h_bd.f1heat_HeI, D.h_bd.f1heat_HeII, D.h_bd.f2heat_HI, D.h_bd.f2heat_HeI, D.h_bd.f2heat_HeII};
  for (int i = 0; i < 12; i++) std::memcpy(dst[i], f[i], sizeof(double) * (NFREQ - 1));
  D.h_bd.bb_upper = bb_upper;
  band_rows_fill(D.h_bd);
  set_tau_zero(0, pt, pn, ht, hn);
  for (int s = 1; s < 3; s++)
    for (int b = 0; b < NFREQ; b++) D.h_bd.tau_zero[s][b] = (double)INFINITY;
  std::vector<double> ht_il(ht.size()), hn_il(hn.size());
  heat_interleave(ht.data(), ht_il.data());
  heat_interleave(hn.data(), hn_il.data());
  std::vector<double> cl(cool, cool + 5 * NCOOL);
  D.mintemp = mintemp;
  D.dtemp = dtemp;
  return up(&D.pthick, pt) || up(&D.pthin, pn) || up(&D.hthick, ht) || up(&D.hthin, hn) || up(&D.hthick_il, ht_il) ||
         up(&D.hthin_il, hn_il) || up(&D.cool, cl) || up_bd();
}

int fp_set_sed(int sed, const double *pthick, const double *pthin, const double *hthick, const double *hthin, int lower, int upper) {
  const int k = sed - 1;
  std::vector<double> pt, pn, ht, hn;
  pitch(pthick, NFREQ, pt);
  pitch(pthin, NFREQ, pn);
  pitch(hthick, NHEAT, ht);
  pitch(hthin, NHEAT, hn);
  D.s_lo[k] = lower - 1;
  D.s_hi[k] = upper;
  set_tau_zero(sed, pt, pn, ht, hn);
  std::vector<double> ht_il(ht.size()), hn_il(hn.size());
  heat_interleave(ht.data(), ht_il.data());
  heat_interleave(hn.data(), hn_il.data());
  return up(&D.s_pt[k], pt) || up(&D.s_pn[k], pn) || up(&D.s_ht_il[k], ht_il) || up(&D.s_hn_il[k], hn_il) || up_bd();
}

int fp_reccoef(int n, const double *T, double *out12) {
  Call c;
  const double *dT = c.in(T, n);
  double *dout = c.out(out12, 12 * (size_t)n);
  if (!c.bad) hipLaunchKernelGGL(k_reccoef, grid(n), dim3(PB), 0, 0, n, dT, dout);
  return c.finish();
}

// out5 = photo_HI, photo_HeI, photo_HeII, heat, photo_out per vector
int fp_photoion(int n, const double *cin6, const double *vol, const double *nflux, const double *istate, int heat, int mode, double *out5) {
  Call c;
  const double *a = c.in(cin6, 6 * (size_t)n), *v = c.in(vol, n), *f = c.in(nflux, n), *s = c.in(istate, n);
  double *o = c.out(out5, 5 * (size_t)n);
  if (!c.bad) {
    if (heat) hipLaunchKernelGGL(k_photoion<true>, grid(n), dim3(PB), 0, 0, n, mode, D.bd, D.pthick, D.pthin, D.hthick_il, D.hthin_il, a, v, f, s, o);
    else hipLaunchKernelGGL(k_photoion<false>, grid(n), dim3(PB), 0, 0, n, mode, D.bd, D.pthick, D.pthin, D.hthick, D.hthin, a, v, f, s, o);
  }
  return c.finish();
}

// rows != 0: cross sections and secondary-ionisation factors read band by band (BandDataByRow, the three-SED heating kernel)
int fp_photoion_multi(int n, const double *cin6, const double *vol, const double *nflux3, const double *istate, int heat, int rows,
                      int mode, double *out5) {
  Call c;
  const double *a = c.in(cin6, 6 * (size_t)n), *v = c.in(vol, n), *f = c.in(nflux3, 3 * (size_t)n), *s = c.in(istate, n);
  double *o = c.out(out5, 5 * (size_t)n);
  const SedSet ss = make_sedset();
  if (!c.bad) {
    const BandData *bd = D.bd;
    const BandDataByRow *bdr = D.bd;
    if (heat && rows) hipLaunchKernelGGL((k_photoion_multi<true, BandDataByRow>), grid(n), dim3(PB), 0, 0, n, mode, bdr, ss, a, v, f, s, o);
    else if (heat) hipLaunchKernelGGL((k_photoion_multi<true, BandData>), grid(n), dim3(PB), 0, 0, n, mode, bd, ss, a, v, f, s, o);
    else if (rows) hipLaunchKernelGGL((k_photoion_multi<false, BandDataByRow>), grid(n), dim3(PB), 0, 0, n, mode, bdr, ss, a, v, f, s, o);
    else hipLaunchKernelGGL((k_photoion_multi<false, BandData>), grid(n), dim3(PB), 0, 0, n, mode, bd, ss, a, v, f, s, o);
  }
  return c.finish();
}

// photo_out_only (multi == 0: nflux3[3 i] is the black-body flux) / photo_out_multi: what the sub-box loop's loss kernels evaluate
int fp_photo_out(int n, int multi, const double *cin6, const double *nflux3, double *out) {
  Call c;
  const double *a = c.in(cin6, 6 * (size_t)n), *f = c.in(nflux3, 3 * (size_t)n);
  double *o = c.out(out, n);
  const SedSet ss = make_sedset();
  if (!c.bad) hipLaunchKernelGGL(k_photo_out, grid(n), dim3(PB), 0, 0, n, static_cast<const BandData *>(D.bd), ss, multi, a, f, o);
  return c.finish();
}

int fp_doric(int n, const double *dt, const double *de, double *ion15, const double *phi3, const double *fr4, const double *rc12,
             const double *clumping) {
  Call c;
  const double *a = c.in(dt, n), *b = c.in(de, n), *p = c.in(phi3, 3 * (size_t)n), *f = c.in(fr4, 4 * (size_t)n),
               *r = c.in(rc12, 12 * (size_t)n), *cl = c.in(clumping, n);
  double *ion = c.inout(ion15, 15 * (size_t)n);
  if (!c.bad) hipLaunchKernelGGL(k_doric, grid(n), dim3(PB), 0, 0, n, a, b, ion, p, f, r, cl);
  return c.finish();
}

int fp_prepare_doric_factors(int n, const double *N3, double *out4) {
  Call c;
  const double *a = c.in(N3, 3 * (size_t)n);
  double *o = c.out(out4, 4 * (size_t)n);
  if (!c.bad) hipLaunchKernelGGL(k_doric_factors, grid(n), dim3(PB), 0, 0, n, a, o);
  return c.finish();
}

int fp_thermal(int n, int lds, const double *dt, double *tend, double *tavg, const double *de, const double *nd, const double *ion15,
               const double *heat, double zred, double H0, double Omega0) {
  Call c;
  const double *a = c.in(dt, n), *e = c.in(de, n), *d = c.in(nd, n), *ion = c.in(ion15, 15 * (size_t)n), *h = c.in(heat, n);
  double *te = c.inout(tend, n), *ta = c.out(tavg, n);
  CoolData cd{D.cool, D.mintemp, D.dtemp, zred, H0, Omega0, nullptr};
  if (!c.bad) hipLaunchKernelGGL(k_thermal, grid(n), dim3(PB), 0, 0, n, lds, cd, a, te, ta, e, d, ion, h);
  return c.finish();
}

// out8[8 * t], t = (di - lo1) + n1 * ((dj - lo2) + n2 * (dk - lo3)) with lo = -(n / 2): see k_cinterp
int fp_cinterp_all(const int *mesh, const double *cH, const double *cHe, const int *src, double *out8) {
  const size_t nc = (size_t)mesh[0] * mesh[1] * mesh[2];
  const int smax = std::max(mesh[0], std::max(mesh[1], mesh[2])) / 2 + 1;
  std::vector<ShellGeom> geom;
  for (int s = 0; s <= smax; s++) geom.push_back(shell_geometry(s));
  Call c;
  const double *a = c.in(cH, nc), *b = c.in(cHe, 2 * nc);
  const ShellGeom *g = c.in(geom.data(), geom.size());
  double *o = c.out(out8, 8 * nc);
  if (!c.bad) hipLaunchKernelGGL(k_cinterp, grid((int)nc), dim3(PB), 0, 0, mesh[0], mesh[1], mesh[2], a, b, src[0], src[1], src[2], g, smax, o);
  return c.finish();
}

} // extern "C"
