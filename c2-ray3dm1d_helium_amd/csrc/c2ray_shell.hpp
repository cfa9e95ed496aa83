// L-infinity shells around a source: the order the column sweep visits cells in, the layout of a source's column
// block, and the per-shell geometry of cinterp (files_for_3D/column_density.f90:28-345) for one whole shell.
//
// Shell s = all cells at L-infinity distance s from the source: 24 s^2 + 2 cells (1 for s = 0).  Within a shell: the
// two k-faces (|dk| = s; (2s+1)^2 cells each, i fastest), then the two j-faces (|dj| = s, |dk| < s; (2s-1)(2s+1)
// cells each, i fastest), then the two i-faces (|di| = s, |dj| < s, |dk| < s; (2s-1)^2 cells each, j fastest).  A
// shell-ordered array holds shell s at [(2s-1)^3, (2s+1)^3).
//
// Host-compilable (tests/host_harness.cpp runs the same functions on the CPU against the generic
// short_characteristic / shell_position pair); the product only uses them in device code and to fill ShellGeom.
#pragma once

#include <stdint.h>

#include "c2ray_device.hpp"

namespace c2r {

// number of cells of the L-infinity shell s
C2R_HD long long shell_count(int s) { return s == 0 ? 1 : 24LL * s * s + 2; }

// first entry of shell s in a shell-ordered array: the (2s-1)^3 cells of all smaller shells come first
C2R_HD long long shell_offset(int s) { return s == 0 ? 0 : (long long)(2 * s - 1) * (2 * s - 1) * (2 * s - 1); }

// t in [0, shell_count(s)) -> offset (di,dj,dk) with max(|di|,|dj|,|dk|) == s; i runs fastest on
// the k- and j-faces so that consecutive lanes touch consecutive memory there.
C2R_HD void shell_decode(int s, int t, int &di, int &dj, int &dk) {
  if (s == 0) { di = dj = dk = 0; return; }
  const int w = 2 * s + 1, v = 2 * s - 1;
  const int A = w * w, B = v * w, C = v * v;
  if (t < 2 * A) {
    dk = t < A ? s : -s;
    if (t >= A) t -= A;
    dj = t / w - s;
    di = t % w - s;
  } else if (t < 2 * A + 2 * B) {
    t -= 2 * A;
    dj = t < B ? s : -s;
    if (t >= B) t -= B;
    dk = t / w - (s - 1);
    di = t % w - s;
  } else {
    t -= 2 * A + 2 * B;
    di = t < C ? s : -s;
    if (t >= C) t -= C;
    dk = t / v - (s - 1);
    dj = t % v - (s - 1);
  }
}

// inverse of shell_decode: position of the cell at offset (di,dj,dk) in a shell-ordered array
C2R_HD size_t shell_position(int di, int dj, int dk) {
  const int ia = di < 0 ? -di : di, ja = dj < 0 ? -dj : dj, ka = dk < 0 ? -dk : dk;
  const int s = ia > ja ? (ia > ka ? ia : ka) : (ja > ka ? ja : ka);
  if (s == 0) return 0;
  const int w = 2 * s + 1, v = 2 * s - 1;
  const int A = w * w, B = v * w, C = v * v;
  int t;
  if (ka == s) t = (dk > 0 ? 0 : A) + (dj + s) * w + (di + s);
  else if (ja == s) t = 2 * A + (dj > 0 ? 0 : B) + (dk + s - 1) * w + (di + s);
  else t = 2 * A + 2 * B + (di > 0 ? 0 : C) + (dk + s - 1) * v + (dj + s - 1);
  return (size_t)shell_offset(s) + (size_t)t;
}

// ----------------------------------------------------------------------------------------------------------------
// Everything about shell s >= 2 that is the same for all its cells, worked out once on the host (k_sweep_shell gets
// it as a kernel argument):
//   * the thread -> cell map needs t / w and t / v: by multiplication with a magic number (Granlund & Montgomery);
//   * cinterp's alam = (km - k0 + sgnk/2) / dk is (s - 1/2) / s for every cell of the shell, whatever the crossing
//     plane and the signs (IEEE division is odd in both operands): one division per launch instead of one per cell;
//   * the path length sqrt(1 + (a^2 + b^2) / n^2) divides by n^2 = s^2: Markstein's sequence with the reciprocal
//     of s^2 (div_recip; operands are small integers, nothing can under- or overflow, s^2 has no all-ones significand);
//   * the four corners cinterp interpolates from lie in shell s - 1 -- or, on the edges of a face, in shell s itself
//     with a bilinear weight that is exactly 0.0 ((s - 1/2) / s * s == s - 1/2 in double for every s, checked on the host
//     up to 4096; tests/test_device_functions_host.py): those are read from the nearest cell of shell s - 1 instead
//     (any finite value gives the same bits), so that every corner's position follows from the face formulas of
//     ONE known shell, without the general inverse map.
constexpr int SHELL_FAST_MAX = 640; // (2s+1)^3 < 2^31: positions within a source's shell-ordered arrays fit 32 bits
struct ShellGeom {
  int s, w, v;        // the shell, 2s+1, 2s-1
  int A, B, C;        // w*w, v*w, v*v: cells of one k-, j-, i-face
  uint32_t mw, mv;    // t / w == mulhi(t, mw) >> kw for 0 <= t < 2^27; likewise v
  int kw, kv;
  int sp;             // s - 1, the shell of the corners
  int wp, vp;         // 2s-1, 2s-3
  int Ap, Bp, Cp;     // its face sizes
  long long off, offp; // shell_offset(s), shell_offset(s-1)
  double alam;        // (s - 0.5) / s
  double n2, rn2;     // s*s, RN(1 / (s*s))
};

// q = t / d for 0 <= t < 2^27 as mulhi(t, m) >> k (round-up method: p = max(32, 27 + ceil(log2 d)), m = ceil(2^p / d))
inline void magic_division(uint32_t d, uint32_t &m, int &k) {
  int L = 0;
  while ((1u << L) < d) L++;
  const int p = 27 + L > 32 ? 27 + L : 32;
  const uint64_t one = (uint64_t)1 << p; // p <= 27 + 13
  m = (uint32_t)((one + d - 1) / d);
  k = p - 32;
}

inline ShellGeom shell_geometry(int s) {
  ShellGeom G;
  G.s = s; G.w = 2 * s + 1; G.v = 2 * s - 1;
  G.A = G.w * G.w; G.B = G.v * G.w; G.C = G.v * G.v;
  magic_division((uint32_t)G.w, G.mw, G.kw);
  magic_division((uint32_t)(G.v > 0 ? G.v : 1), G.mv, G.kv);
  G.sp = s - 1; G.wp = 2 * s - 1; G.vp = 2 * s - 3;
  G.Ap = G.wp * G.wp; G.Bp = G.vp * G.wp; G.Cp = G.vp * G.vp;
  G.off = shell_offset(s); G.offp = shell_offset(s > 0 ? s - 1 : 0);
  G.alam = ((double)s - 0.5) / (double)s;
  G.n2 = (double)s * (double)s;
  G.rn2 = 1.0 / G.n2;
  return G;
}

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ uint32_t mulhi_u32(uint32_t a, uint32_t b) { return __umulhi(a, b); }
#else
inline uint32_t mulhi_u32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
#endif

// shell_decode for the shell of G; also tells the face: 0 = k-face (z-plane crossing, column_density.f90:107),
// 1 = j-face (y-plane crossing, :199), 2 = i-face (x-plane crossing, :275)
C2R_HD int shell_decode_fast(const ShellGeom &G, int t, int &di, int &dj, int &dk) {
  // One straight line with selects on VALUES: written as three branches that assign di, dj, dk in different roles,
  // the compiler merged the assignments into stores through a selected pointer -- the three offsets went to scratch
  // memory and came back with a full wait each, in every thread of the sweep.
  const int s = G.s;
  const int face = t < 2 * G.A ? 0 : (t < 2 * G.A + 2 * G.B ? 1 : 2);
  const int first = face == 0 ? 0 : (face == 1 ? 2 * G.A : 2 * G.A + 2 * G.B); // first cell of the pair of faces
  const int size = face == 0 ? G.A : (face == 1 ? G.B : G.C);                  // cells of one face
  const int width = face == 2 ? G.v : G.w;                                     // cells of one row
  const uint32_t magic = face == 2 ? G.mv : G.mw;
  const int shift = face == 2 ? G.kv : G.kw;
  const int tf = t - first;
  const bool neg = tf >= size;
  const uint32_t u = (uint32_t)(neg ? tf - size : tf);
  const uint32_t r = mulhi_u32(u, magic) >> shift;          // the row
  const int col = (int)(u - r * (uint32_t)width);           // the place in the row
  const int a = col - (face == 2 ? s - 1 : s);              // the coordinate that runs fastest
  const int b = (int)r - (face == 0 ? s : s - 1);           // the one that counts the rows
  const int c = neg ? -s : s;                               // the one that is fixed on the face
  di = face == 2 ? c : a;
  dj = face == 0 ? b : (face == 1 ? c : a);
  dk = face == 0 ? c : b;
  return face;
}

// position within shell s-1 (without its shell_offset) of a cell known to lie in that shell: max(|i|,|j|,|k|) == s-1
C2R_HD int position_in_previous_shell(const ShellGeom &G, int i, int j, int k) {
  const int sp = G.sp;
  const int ja = j < 0 ? -j : j, ka = k < 0 ? -k : k;
  const int tk = (k > 0 ? 0 : G.Ap) + (j + sp) * G.wp + (i + sp);
  const int tj = 2 * G.Ap + (j > 0 ? 0 : G.Bp) + (k + sp - 1) * G.wp + (i + sp);
  const int ti = 2 * G.Ap + 2 * G.Bp + (i > 0 ? 0 : G.Cp) + (k + sp - 1) * G.vp + (j + sp - 1);
  return ka == sp ? tk : (ja == sp ? tj : ti);
}

// cinterp's geometry for a cell of shell G.s >= 2 on face `face` (shell_decode_fast): positions of the four corners
// c1..c4 in the shell-ordered arrays, their bilinear weights s1..s4 and the path length in cell units.  The same
// operations on the same values as short_characteristic (c2ray_device.hpp) -- the diagonal factors of
// column_density.f90:174-184 only occur in shell 1 --, with the per-shell constants of ShellGeom.
struct ShellCorners {
  uint32_t p[4]; // positions in the shell-ordered arrays: 32 bits hold them up to shell 645 (SHELL_FAST_MAX)
  double s[4];
  double path;
};
C2R_HD void shell_short_characteristic(const ShellGeom &G, int face, int i0, int j0, int k0, int di, int dj, int dk,
                                       ShellCorners &sc) {
  const int sp = G.sp;
  // the two in-plane axes (a, b) of the crossing plane and the source's coordinates along them:
  // k-face: (x, y); j-face: (x, z); i-face: (y, z)   (column_density.f90:113-131, :205-223, :281-299)
  const int da = face == 2 ? dj : di, db = face == 0 ? dj : dk;
  const int a0 = face == 2 ? j0 : i0, b0 = face == 0 ? j0 : k0;
  const int sga = da >= 0 ? 1 : -1, sgb = db >= 0 ? 1 : -1;
  const int am = da - sga, bm = db - sgb; // the cell closer to the source along each in-plane axis
  const double fa = (double)da, fb = (double)db;
  const double ac = G.alam * fa + (double)a0, bc = G.alam * fb + (double)b0;
  const double ea = 2.0 * fabs(ac - ((double)(a0 + am) + 0.5 * sga));
  const double eb = 2.0 * fabs(bc - ((double)(b0 + bm) + 0.5 * sgb));
  sc.s[0] = (1. - ea) * (1. - eb);
  sc.s[1] = (1. - eb) * ea;
  sc.s[2] = (1. - ea) * eb;
  sc.s[3] = ea * eb;
  // sqrt((a^2 + b^2) / n^2 + 1), n^2 = s^2: Markstein's division by the per-shell constant
  const double num = fa * fa + fb * fb;
  const double q0 = num * G.rn2;
  const double quo = __builtin_fma(__builtin_fma(-G.n2, q0, num), G.rn2, q0);
  sc.path = sqrt(quo + 1.0);
  // corners (am|da, bm|db) in the plane one step closer to the source; a coordinate beyond shell s-1 (the cell sits
  // on an edge of its face) belongs to a corner of weight exactly 0: take the nearest cell of shell s-1 instead
  const int ca = da > sp ? sp : (da < -sp ? -sp : da), cb = db > sp ? sp : (db < -sp ? -sp : db);
  const uint32_t offp = (uint32_t)G.offp;
  if (face == 0) {
    const int km = dk > 0 ? sp : -sp;
    const int base = (km > 0 ? 0 : G.Ap) + sp;
    const int ra = base + (bm + sp) * G.wp, rb = base + (cb + sp) * G.wp;
    sc.p[0] = offp + (uint32_t)(ra + am);
    sc.p[1] = offp + (uint32_t)(ra + ca);
    sc.p[2] = offp + (uint32_t)(rb + am);
    sc.p[3] = offp + (uint32_t)(rb + ca);
  } else if (face == 1) {
    const int jm = dj > 0 ? sp : -sp;
    // a row (fixed k') of the plane j' = jm lies in a k-face of shell s-1 if |k'| == s-1, else in its j-face
    const int rk_base = (jm + sp) * G.wp + sp, rj_base = 2 * G.Ap + (jm > 0 ? 0 : G.Bp) + sp + (sp - 1) * G.wp;
    const int bma = bm < 0 ? -bm : bm, cba = cb < 0 ? -cb : cb;
    const int ra = bma == sp ? (bm > 0 ? 0 : G.Ap) + rk_base : rj_base + bm * G.wp;
    const int rb = cba == sp ? (cb > 0 ? 0 : G.Ap) + rk_base : rj_base + cb * G.wp;
    sc.p[0] = offp + (uint32_t)(ra + am);
    sc.p[1] = offp + (uint32_t)(ra + ca);
    sc.p[2] = offp + (uint32_t)(rb + am);
    sc.p[3] = offp + (uint32_t)(rb + ca);
  } else {
    const int im = di > 0 ? sp : -sp;
    sc.p[0] = offp + (uint32_t)position_in_previous_shell(G, im, am, bm);
    sc.p[1] = offp + (uint32_t)position_in_previous_shell(G, im, ca, bm);
    sc.p[2] = offp + (uint32_t)position_in_previous_shell(G, im, am, cb);
    sc.p[3] = offp + (uint32_t)position_in_previous_shell(G, im, ca, cb);
  }
}

// 1 / max(0.6, cd * sig) (weightf, column_density.f90:351-376): the argument of the reciprocal lies in
// [0.6, 1.2e291] for every finite column, where recip_nr (the division's own instruction sequence without operand
// scaling) is exact
// (the maximum as __builtin_fmax -- one v_max_f64; `a > b ? a : b` is a compare and two selects, twelve times per
// cell; the product of a finite column and a cross section is never a NaN --: 28 instructions per cell less, 1 % of the
// sweep.  The same instruction through inline asm (dmax_const) made the sweep 3 % SLOWER: the compiler no longer
// scheduled around it.  A square root without operand scaling for the path length, seven instructions less: neutral.)
C2R_HD double weightf_fast(double cd, double sig) { return recip_nr(dmax_num(cd * sig, 0.6)); }

// weighted mean of the four corner columns of one species (column_density.f90:145-163) with those weights
C2R_HD double interp_column_fast(const double (&s)[4], double c1, double c2, double c3, double c4, double sig) {
  const double w1 = s[0] * weightf_fast(c1, sig), w2 = s[1] * weightf_fast(c2, sig), w3 = s[2] * weightf_fast(c3, sig),
               w4 = s[3] * weightf_fast(c4, sig);
  return (c1 * w1 + c2 * w2 + c3 * w3 + c4 * w4) / (w1 + w2 + w3 + w4);
}

} // namespace c2r
