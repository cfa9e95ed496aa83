// MI355X (gfx950) implementation of C2-Ray's evolve3D hot path behind the C ABI of
// include/c2ray_hip.h.  See DESIGN.md for the data layout and the kernel list.
//
//   column sweep   k_sweep_shell   one launch per L-infinity shell around the sources of a batch:
//                                  short-characteristics interpolation of the three incoming
//                                  columns (cinterp) + the cell's own columns (evolve0D, first half)
//   rates          k_rates         all cells x all sources of the batch, no dependencies:
//                                  photoion_rates + accumulation into the rate grids in source order
//                                  (evolve0D, second half)
//   boundary loss  k_loss_probe_rounds  a lower bound of the photon loss through a sub-box surface from a sample of the
//                                  stored columns (decides "this source goes on" for several rounds in one launch);
//                  k_loss, k_loss_finish   the full, block-ordered sum where the bound does not decide or the loss is kept;
//                  k_loss_stored   the kept loss of a final round, left behind by k_rates in the unused N_in(HI) slots
//   chemistry      k_chemistry     evolve0D_global / do_chemistry / doric / thermal per cell
//   statistics     k_state_sums, k_total_rates, k_stat_finish   the grid sums of photonstatistics.f90
//   tables         k_build_tables  spec_integration: the photo-ionisation / heating tables of one SED
//
// Compile: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "../../include/c2ray_hip.h"
#include "c2ray_device.hpp"
#include "c2ray_shell.hpp"

using namespace c2r;

namespace {

constexpr int BATCH_MAX = 4096; // sources in flight per batch (device-resident records, no kernel-argument limit)
constexpr int BLOCK = 256;

struct Grid {
  int n1, n2, n3;
  int l1, l2, l3; // left extent mesh/2 (evolve_source.F90:105)
  size_t ncell;
  size_t colsize; // entries of one shell-ordered column array that holds the whole mesh: (2*smax+1)^3
  int smax;       // largest L-infinity shell index, max(l1,l2,l3)
};

// One source of the batch in flight, in device memory (read through the scalar cache: uniform per block).
struct SrcDev {
  int i0, j0, k0;      // 1-based mesh position (srcpos)
  int lo[3], hi[3];    // FINAL sub-box as offsets last_l - srcpos, last_r - srcpos (set before the rates launch)
  double nflux;        // NormFlux(ns)
  double nflux_sed[2]; // NormFluxPL(ns), NormFluxQPL(ns) (-DPL / -DQUASARS builds), else 0
  double *cols;        // this source's column block in the scratch arena
  size_t cz;           // entries per column array of that block: (2*cap+1)^3, cap = shells the block can hold
  int loss_lo;         // >= 0: the rates launch leaves, for every surface cell of the final sub-box in shells >= loss_lo,
                       // the photons that leave the box through it in the cell's N_in(HI) slot (k_loss_stored adds
                       // them up); -1: no such request (set before the rates launch)
};
// the sub-box of the round in flight: the same for every active source of a batch (all are in the same round)
struct Box {
  int lo[3], hi[3];
};

struct StepScalars {
  double dr1, dr2, dr3, vol;
  double cellvol;       // dr1 * dr2 * dr3 (vol_ph of a source's own cell, evolve_point.F90:203), formed on the host: a product of
                        // uniform numbers has no scalar instruction, and hoisted out of k_rates' source loop it costs two vector registers
  double clumping;
  double temper_val;
  RecCoef rc;
  CoolData cd;
  int use_lls;          // c2ray_parameters: use_LLS
  double coldensh_lls;  // material: coldensh_LLS (type_of_LLS = 1)
};

__device__ __forceinline__ int wrap0(int x, int n) { // 0-based periodic index of x in [-n, 2n)
  // of x, x + n and x - n exactly one lies in [0, n); read as unsigned numbers it is the smallest of the three
  // (v_add, v_sub, v_min3_u32 instead of two compares, two selects and the arithmetic)
  const unsigned u = (unsigned)x, m = (unsigned)n;
  const unsigned a = u + m, b = u - m;
  const unsigned lo = a < b ? a : b;
  return (int)(u < lo ? u : lo);
}

__device__ __forceinline__ double block_sum(double x, double *sh) {
  // fixed-shape tree: wave shuffle, then the 4 wave sums in order -> deterministic
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) sh[w] = x;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) {
    for (int i = 0; i < BLOCK / 64; i++) r += sh[i];
  }
  return r; // valid in thread 0
}

// path length of cinterp for an offset, without the corner work (column_density.f90:194,269,341)
__device__ __forceinline__ double sc_path(int idel, int jdel, int kdel) {
  const int ia = idel < 0 ? -idel : idel, ja = jdel < 0 ? -jdel : jdel, ka = kdel < 0 ? -kdel : kdel;
  const double di = (double)idel, dj = (double)jdel, dk = (double)kdel;
  if (ka >= ja && ka >= ia) return sqrt((di * di + dj * dj) / (dk * dk) + 1.0);
  if (ja >= ia && ja >= ka) return sqrt((di * di + dk * dk) / (dj * dj) + 1.0);
  return sqrt(1.0 + (dj * dj + dk * dk) / (di * di));
}

// A pointer that comes out of a record in memory (SrcDev::cols) is a generic pointer to the compiler, which then uses
// flat loads and stores; the column blocks are device memory.
#ifdef C2R_FLAT_COLS
typedef double global_double;
#else
typedef __attribute__((address_space(1))) double global_double;
#endif

// Where the six columns of the cell at shell position p live in a source's block of 6 cz doubles: the three incoming
// columns of a cell side by side (24 bytes), then all outgoing ones likewise.  The rates kernel, whose 4 x 4 x 4 cubes
// read rows of four cells, then uses 96 of every 128 bytes it touches instead of 32 (-0.3..0.5 ms per launch at
// 256^3 x 8); the sweep's stores become three strided ones per triple (+0.05..0.1 ms per pass).
#ifndef C2R_COLS_AOS
#define C2R_COLS_AOS 1
#endif
// 0: six arrays; 1: both triples side by side; 2: only the incoming ones; 3: only the outgoing ones
__host__ __device__ inline size_t col_in(size_t p, int k, size_t cz) {
  return (C2R_COLS_AOS == 1 || C2R_COLS_AOS == 2) ? 3 * p + (size_t)k : p + (size_t)k * cz;
}
__host__ __device__ inline size_t col_out(size_t p, int k, size_t cz) {
  return (C2R_COLS_AOS == 1 || C2R_COLS_AOS == 3) ? 3 * cz + 3 * p + (size_t)k : p + (size_t)(3 + k) * cz;
}

// The sweep's stores of a cell's six columns.  C2R_SWEEP_NT: 1 = the incoming columns, which only the rates launch
// reads, long after, as non-temporal stores: they then do not sit dirty in the L2s until the end of the launch, when
// every XCD writes its cache back before the next shell may start (a launch per shell: up to 32 MB each time).
// 2 = the outgoing ones as well.  Measured at 256^3 x 8, same box, ms per pass (sweep | rates): isothermal 0: 3.98-4.03
// | 18.5, 1: 3.82 | 18.5, 2: 3.78-3.83 | 18.4; with heating 0: 3.86 | 27.9, 1: 3.63 | 28.0-28.1, 2: 3.69 | 28.3 -- the
// rates launch reads non-temporally written columns a little slower, so only the incoming ones are stored that way.
// (Non-temporal LOADS of the columns in the rates kernel: +0.4 ms per launch, neighbouring cubes share their lines.)
#ifndef C2R_SWEEP_NT
#define C2R_SWEEP_NT 1
#endif
__device__ __forceinline__ void store_columns(global_double *cs, size_t p, size_t cz, double cin_HI, double cin_HeI, double cin_HeII,
                                              double cout_HI, double cout_HeI, double cout_HeII) {
#if C2R_SWEEP_NT >= 1
  __builtin_nontemporal_store(cin_HI, &cs[col_in(p, 0, cz)]);
  __builtin_nontemporal_store(cin_HeI, &cs[col_in(p, 1, cz)]);
  __builtin_nontemporal_store(cin_HeII, &cs[col_in(p, 2, cz)]);
#else
  cs[col_in(p, 0, cz)] = cin_HI;
  cs[col_in(p, 1, cz)] = cin_HeI;
  cs[col_in(p, 2, cz)] = cin_HeII;
#endif
#if C2R_SWEEP_NT >= 2
  __builtin_nontemporal_store(cout_HI, &cs[col_out(p, 0, cz)]);
  __builtin_nontemporal_store(cout_HeI, &cs[col_out(p, 1, cz)]);
  __builtin_nontemporal_store(cout_HeII, &cs[col_out(p, 2, cz)]);
#else
  cs[col_out(p, 0, cz)] = cout_HI;
  cs[col_out(p, 1, cz)] = cout_HeI;
  cs[col_out(p, 2, cz)] = cout_HeII;
#endif
}

// ---------------------------------------------------------------------------------------------
// What the sweep needs of a cell's state, once per pass instead of once per cell.source: the three products
// neufrac * ndens of coldens (doric.f90:358-372: neufrac*ndens*path*abundance is evaluated from the left, so the
// first product does not depend on the source) for HI, HeI, HeII with the fractions clamped at epsilon
// (evolve_point.F90:132-136) -- 24 bytes per cell.source instead of 32 --, in mesh order (`packed`, index
// i + n1*(j + n2*k)) and, for the i-faces of a shell (fixed i, consecutive j), in (j,i,k) order (`packedT`, index
// j + n2*(i + n1*k)); three arrays of ncell each.  32x32 tiles of a k-plane through LDS.
__global__ void __launch_bounds__(BLOCK)
k_pack_state(Grid g, const double *__restrict__ ndens, const double *__restrict__ xh_av, const double *__restrict__ xhe_av,
             double *__restrict__ packed, double *__restrict__ packedT) {
  __shared__ double tile[3][32][33];
  const size_t nc = g.ncell;
  const int k = blockIdx.z;
  const size_t plane = (size_t)k * g.n1 * g.n2;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8
  const int i0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
  for (int r = ty; r < 32; r += 8) {
    const int i = i0 + tx, j = j0 + r;
    if (i < g.n1 && j < g.n2) {
      const size_t q = plane + (size_t)i + (size_t)g.n1 * j;
      const double nd = ndens[q];
      const double u0 = dmax(xh_av[q], epsilon) * nd, u1 = dmax(xhe_av[q], epsilon) * nd, u2 = dmax(xhe_av[q + nc], epsilon) * nd;
      packed[q] = u0; packed[q + nc] = u1; packed[q + 2 * nc] = u2;
      tile[0][r][tx] = u0; tile[1][r][tx] = u1; tile[2][r][tx] = u2;
    }
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int j = j0 + tx, i = i0 + r;
    if (i < g.n1 && j < g.n2) {
      const size_t qT = plane + (size_t)j + (size_t)g.n2 * i;
      packedT[qT] = tile[0][tx][r]; packedT[qT + nc] = tile[1][tx][r]; packedT[qT + 2 * nc] = tile[2][tx][r];
    }
  }
}

// packed -> packedT alone, when the global pass has left `packed` up to date (k_chemistry writes it with the
// fractions it stores): 24 bytes read and written per cell instead of 32 and 48
__global__ void __launch_bounds__(BLOCK)
k_transpose_packed(Grid g, const double *__restrict__ packed, double *__restrict__ packedT) {
  __shared__ double tile[32][33];
  const int a = blockIdx.z / g.n3, k = blockIdx.z % g.n3;
  const double *src = packed + (size_t)a * g.ncell + (size_t)k * g.n1 * g.n2;
  double *dst = packedT + (size_t)a * g.ncell + (size_t)k * g.n1 * g.n2;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // 32 x 8
  const int i0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
  for (int r = ty; r < 32; r += 8) {
    const int i = i0 + tx, j = j0 + r;
    if (i < g.n1 && j < g.n2) tile[r][tx] = src[(size_t)i + (size_t)g.n1 * j];
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int j = j0 + tx, i = i0 + r;
    if (i < g.n1 && j < g.n2) dst[(size_t)j + (size_t)g.n2 * i] = tile[tx][r];
  }
}

// ---------------------------------------------------------------------------------------------
// The photon loss through the surface of the current sub-box (evolve_point.F90:310-315) from the columns the
// sweep has stored, for all shells [s_lo, s_hi] of the round in ONE launch.  A block takes 256 consecutive cells
// of one shell (the sweep's own thread <-> cell map); its partial sum goes to partial[source][block number
// within the round], and k_loss_finish adds the partials in that order: the same bits on every run and for
// every batch composition.
// `sample` > 1: only every sample-th block is evaluated (blockIdx.x counts those).  Every term is >= 0, so this
// is a lower bound: enough to show that the loss of a round that cannot be a source's last for geometric
// reasons exceeds the threshold of evolve_source.F90:136.  Anything a sample does not decide is redone with
// sample = 1, and only such full sums are ever kept.
// (Inside k_sweep_shell this work sat in a sixteenth of the waves, all of them on two of the eight XCDs: the
// last shell of every round took twice as long as its neighbours.  Here every lane of every wave has a cell.)
__global__ void __launch_bounds__(BLOCK)
k_loss(Grid g, const SrcDev *__restrict__ src, const int *__restrict__ list, int multi, int s_lo, int s_hi, Box box,
       StepScalars sc, const BandData *__restrict__ bd, SedSet ss,
       const int *__restrict__ block_base, double *__restrict__ loss_partial, int pitch, int sample, int first_block) {
  // first_block: the blocks before it hold no surface cell (the host has set their partials to zero and does not
  // launch them); block numbers, and with them the order of the sum, stay what they are
  __shared__ double sh[BLOCK / 64];
  const SrcDev &S = src[list[blockIdx.y]];
  const int bx = first_block + (int)blockIdx.x;
  const int B = block_base[s_lo] + bx * sample;
  int lo = s_lo, hi = s_hi; // largest shell with block_base[shell] <= B
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (block_base[mid] <= B) lo = mid; else hi = mid - 1;
  }
  const int shell = lo;
  const long long cnt = shell_count(shell);
  const long long t = (long long)(B - block_base[shell]) * BLOCK + threadIdx.x;
  double loss = 0.0;
  if (t < cnt) {
    int di, dj, dk;
    shell_decode(shell, (int)t, di, dj, dk);
    const bool inside = di >= box.lo[0] && di <= box.hi[0] && dj >= box.lo[1] && dj <= box.hi[1] && dk >= box.lo[2] &&
                        dk <= box.hi[2];
    const bool boundary = di == box.lo[0] || dj == box.lo[1] || dk == box.lo[2] || di == box.hi[0] || dj == box.hi[1] ||
                          dk == box.hi[2];
    if (inside && boundary) {
      const size_t cz = S.cz;
      const size_t p = (size_t)shell_offset(shell) + (size_t)t;
      const global_double *cs = (const global_double *)S.cols;
      const double cin_HI = cs[col_in(p, 0, cz)], cin_HeI = cs[col_in(p, 1, cz)], cin_HeII = cs[col_in(p, 2, cz)];
      const double cout_HI = cs[col_out(p, 0, cz)], cout_HeI = cs[col_out(p, 1, cz)], cout_HeII = cs[col_out(p, 2, cz)];
      if (cin_HI < max_coldensh) {
        double vol_ph;
        if (shell == 0) {
          vol_ph = sc.dr1 * sc.dr2 * sc.dr3;
        } else {
          const double path = sc_path(di, dj, dk) * sc.dr1;
          const double xs = sc.dr1 * (double)di, ys = sc.dr2 * (double)dj, zs = sc.dr3 * (double)dk;
          const double dist2 = xs * xs + ys * ys + zs * zs;
          vol_ph = 4.0 * pi * dist2 * path;
        }
        double po;
        if (multi) {
          const double nf[NSED] = {S.nflux, S.nflux_sed[0], S.nflux_sed[1]};
          po = photo_out_multi(*bd, ss, cin_HI, cout_HI, cin_HeI, cout_HeI, cin_HeII, cout_HeII, nf);
        } else {
          po = photo_out_only(*bd, ss.photo_thick[0], ss.photo_thin[0], cin_HI, cout_HI, cin_HeI, cout_HeI, cin_HeII,
                              cout_HeII, S.nflux);
        }
        loss = po * sc.vol / vol_ph;
      }
    }
  }
  const double bs = block_sum(loss, sh);
  if (threadIdx.x == 0) loss_partial[(size_t)blockIdx.y * pitch + bx] = bs;
}

// ---------------------------------------------------------------------------------------------
// The same sum from values the rates kernel has left behind.  The loss that is KEPT for a source whose last round
// ended for geometric reasons (the box cannot grow any further, evolve_source.F90:136-139) decides nothing, and
// its terms -- photo_out * vol / vol_ph of the surface cells, evolve_point.F90:310-315 -- are quantities k_rates
// computes anyway for every cell.source (the reference takes them from the same photoion_rates call, phi%photo_out).
// So k_rates stores them (SrcDev::loss_lo) in the cell's N_in(HI) slot of the column block, which nobody reads
// after it, and this kernel adds them up with k_loss's thread <-> cell map, block partials and block order: the
// same bits as k_loss gives, without evaluating a single band a second time.
__global__ void __launch_bounds__(BLOCK)
k_loss_stored(Grid g, const SrcDev *__restrict__ src, const int *__restrict__ list, int s_lo, int s_hi, Box box,
              const int *__restrict__ block_base, double *__restrict__ loss_partial, int pitch, int first_block) {
  __shared__ double sh[BLOCK / 64];
  const SrcDev &S = src[list[blockIdx.y]];
  const int bx = first_block + (int)blockIdx.x;
  const int B = block_base[s_lo] + bx;
  int lo = s_lo, hi = s_hi; // largest shell with block_base[shell] <= B
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (block_base[mid] <= B) lo = mid; else hi = mid - 1;
  }
  const int shell = lo;
  const long long cnt = shell_count(shell);
  const long long t = (long long)(B - block_base[shell]) * BLOCK + threadIdx.x;
  double loss = 0.0;
  if (t < cnt) {
    int di, dj, dk;
    shell_decode(shell, (int)t, di, dj, dk);
    const bool inside = di >= box.lo[0] && di <= box.hi[0] && dj >= box.lo[1] && dj <= box.hi[1] && dk >= box.lo[2] &&
                        dk <= box.hi[2];
    const bool boundary = di == box.lo[0] || dj == box.lo[1] || dk == box.lo[2] || di == box.hi[0] || dj == box.hi[1] ||
                          dk == box.hi[2];
    if (inside && boundary) {
      const size_t p = (size_t)shell_offset(shell) + (size_t)t;
      loss = ((const global_double *)S.cols)[col_in(p, 0, S.cz)];
    }
  }
  const double bs = block_sum(loss, sh);
  if (threadIdx.x == 0) loss_partial[(size_t)blockIdx.y * pitch + bx] = bs;
}


// ---------------------------------------------------------------------------------------------
// Column sweep of one shell for every active source of the batch (blockIdx.y counts `active`).
// evolve0D, files_for_3D/evolve_point.F90:114-168 and :237-244, with cinterp (column_density.f90:28-345).
// A source's column block in the arena: 6 cz doubles = N_in(HI,HeI,HeII) of every cell, then N_out(HI,HeI,HeII)
// (col_in / col_out), the cells in SHELL ORDER (shell_position of the cell's offset from its source): thread t of
// shell s owns entry shell_offset(s)+t, so the stores of a wave cover one contiguous range and -- because the
// corners of consecutive cells are consecutive cells of the previous shell on every face, the i-faces included --
// so do the corner loads.  (In mesh order the i-faces of a shell are one cell per 128-byte line.)
#ifndef C2R_SWEEP_WAVES
#define C2R_SWEEP_WAVES 1
#endif
// what every cell of a sweep launch is given
struct SweepArgs {
  Grid g;
  Box box;
  StepScalars sc;
  const double *ndens, *xh_av, *xhe_av;
  const double *packed, *packedT; // neufrac * ndens of the three species in mesh and in (j,i,k) order (k_pack_state,
                                  // k_chemistry, k_transpose_packed); null while they are being made
  const float *lls_grid;
};

// neufrac * ndens of the three species for the cell at mesh position (i,j,k), 0-based (k_pack_state), or from the
// grids while the packed copies of this pass are still being made (the inner shells); iface: the cell lies on an
// i-face of its shell, where consecutive lanes have consecutive j -> the (j,i,k)-ordered copy
__device__ __forceinline__ void sweep_cell_state(const SweepArgs &A, int i, int j, int k, bool iface, double &u_HI, double &u_HeI,
                                                 double &u_HeII) {
  const Grid &g = A.g;
  const size_t nc = g.ncell;
  if (A.packed) {
    const bool tr = iface && A.packedT; // (the inner shells run before the (j,i,k)-ordered copy exists: strided reads)
    const double *P = tr ? A.packedT : A.packed;
    // a mesh has fewer than 2^31 cells (c2r_create): the cell number in 32 bits, one widening for the address
    const unsigned q = tr ? (unsigned)j + (unsigned)g.n2 * ((unsigned)i + (unsigned)g.n1 * (unsigned)k)
                          : (unsigned)i + (unsigned)g.n1 * ((unsigned)j + (unsigned)g.n2 * (unsigned)k);
    u_HI = P[q];
    u_HeI = P[q + nc];
    u_HeII = P[q + 2 * nc];
  } else {
    const size_t q = (size_t)i + (size_t)g.n1 * ((size_t)j + (size_t)g.n2 * (size_t)k);
    const double nd = A.ndens[q];
    u_HI = dmax(A.xh_av[q], epsilon) * nd;
    u_HeI = dmax(A.xhe_av[q], epsilon) * nd;
    u_HeII = dmax(A.xhe_av[q + nc], epsilon) * nd;
  }
}

// evolve0D's column part (files_for_3D/evolve_point.F90:114-168, :237-244) for cell t of shell `shell` around source
// S, any shell: cinterp through the general short_characteristic / shell_position
__device__ __forceinline__ void sweep_cell(const SweepArgs &A, const SrcDev &S, int shell, int t) {
  const Grid &g = A.g;
  const StepScalars &sc = A.sc;
  int di, dj, dk;
  shell_decode(shell, t, di, dj, dk);
  const bool inside = di >= A.box.lo[0] && di <= A.box.hi[0] && dj >= A.box.lo[1] && dj <= A.box.hi[1] && dk >= A.box.lo[2] &&
                      dk <= A.box.hi[2];
  if (!inside) return;
  const size_t cz = S.cz;
  const size_t p = (size_t)shell_offset(shell) + (size_t)t;
  global_double *cs = (global_double *)S.cols;
  const int i = wrap0(S.i0 - 1 + di, g.n1), j = wrap0(S.j0 - 1 + dj, g.n2), k = wrap0(S.k0 - 1 + dk, g.n3);
  const int w_ = 2 * shell + 1;
  double u_HI, u_HeI, u_HeII;
  sweep_cell_state(A, i, j, k, shell > 0 && t >= 2 * w_ * w_ + 2 * (w_ - 2) * w_, u_HI, u_HeI, u_HeII);
  double cin_HI, cin_HeI, cin_HeII, path;
  if (shell == 0) {
    cin_HI = cin_HeI = cin_HeII = 0.0;
    path = 0.5 * sc.dr1;
  } else {
    ShortChar s4;
    short_characteristic(S.i0, S.j0, S.k0, di, dj, dk, s4);
    size_t qc[4];
#pragma unroll
    for (int c = 0; c < 4; c++) qc[c] = shell_position(s4.ci[c], s4.cj[c], s4.ck[c]);
    cin_HI = interp_column(s4, cs[col_out(qc[0], 0, cz)], cs[col_out(qc[1], 0, cz)], cs[col_out(qc[2], 0, cz)],
                           cs[col_out(qc[3], 0, cz)], sigma_HI_at_ion_freq);
    cin_HeI = interp_column(s4, cs[col_out(qc[0], 1, cz)], cs[col_out(qc[1], 1, cz)], cs[col_out(qc[2], 1, cz)],
                            cs[col_out(qc[3], 1, cz)], sigma_HeI_at_ion_freq);
    cin_HeII = interp_column(s4, cs[col_out(qc[0], 2, cz)], cs[col_out(qc[1], 2, cz)], cs[col_out(qc[2], 2, cz)],
                             cs[col_out(qc[3], 2, cz)], sigma_HeII_at_ion_freq);
    path = s4.path * sc.dr1;
    if (sc.use_lls) {
      // Lyman-limit-system fog on the incoming HI column (evolve_point.F90:177-180); the per-cell
      // grid is LLS_point of type_of_LLS = 2 (REAL(4), mat_ini_cubep3m.F90:859-870)
      const double coldensh_LLS =
          A.lls_grid ? (double)A.lls_grid[(size_t)i + (size_t)g.n1 * ((size_t)j + (size_t)g.n2 * (size_t)k)] : sc.coldensh_lls;
      cin_HI = cin_HI + coldensh_LLS * path / sc.dr1;
    }
  }
  // coldens (doric.f90:358-372): neufrac * ndens * path * abundance, from the left
  const double cout_HI = cin_HI + u_HI * path * (1.0 - abu_he);
  const double cout_HeI = cin_HeI + u_HeI * path * abu_he;
  const double cout_HeII = cin_HeII + u_HeII * path * abu_he;
  store_columns(cs, p, cz, cin_HI, cin_HeI, cin_HeII, cout_HI, cout_HeI, cout_HeII);
}

// The same for a shell s >= 2, with everything that is common to the cells of one shell taken out of the cell's
// work (ShellGeom, c2ray_shell.hpp): no integer division in the thread -> cell map, cinterp's alam as a per-launch
// constant, the path's division by s^2 as three fma, the 12 reciprocals of weightf without the division's operand
// scaling, and the corners' positions from the face formulas of shell s-1 instead of four general inverse maps
// (corners of weight exactly 0 on the edges of a face are read from the nearest cell of shell s-1).  Bit for bit
// the columns of sweep_cell (tests/test_gpu_parity.py::test_fast_sweep_equals_the_general_sweep runs that one for every
// shell, C2R_SWEEP_GENERIC=1, in a process of its own and compares the columns).
__device__ __forceinline__ void sweep_cell_fast(const SweepArgs &A, const SrcDev &S, const ShellGeom &G, int t) {
  const Grid &g = A.g;
  const StepScalars &sc = A.sc;
  int di, dj, dk;
  const int face = shell_decode_fast(G, t, di, dj, dk);
  const bool inside = di >= A.box.lo[0] && di <= A.box.hi[0] && dj >= A.box.lo[1] && dj <= A.box.hi[1] && dk >= A.box.lo[2] &&
                      dk <= A.box.hi[2];
  if (!inside) return;
  const size_t cz = S.cz;
  const size_t p = (size_t)G.off + (size_t)t;
  global_double *cs = (global_double *)S.cols;
  const int i = wrap0(S.i0 - 1 + di, g.n1), j = wrap0(S.j0 - 1 + dj, g.n2), k = wrap0(S.k0 - 1 + dk, g.n3);
  double u_HI, u_HeI, u_HeII;
  sweep_cell_state(A, i, j, k, face == 2, u_HI, u_HeI, u_HeII);
  ShellCorners c4;
  shell_short_characteristic(G, face, S.i0, S.j0, S.k0, di, dj, dk, c4);
  double cin_HI = interp_column_fast(c4.s, cs[col_out((size_t)c4.p[0], 0, cz)], cs[col_out((size_t)c4.p[1], 0, cz)],
                                     cs[col_out((size_t)c4.p[2], 0, cz)], cs[col_out((size_t)c4.p[3], 0, cz)], sigma_HI_at_ion_freq);
  const double cin_HeI = interp_column_fast(c4.s, cs[col_out((size_t)c4.p[0], 1, cz)], cs[col_out((size_t)c4.p[1], 1, cz)],
                                            cs[col_out((size_t)c4.p[2], 1, cz)], cs[col_out((size_t)c4.p[3], 1, cz)], sigma_HeI_at_ion_freq);
  const double cin_HeII = interp_column_fast(c4.s, cs[col_out((size_t)c4.p[0], 2, cz)], cs[col_out((size_t)c4.p[1], 2, cz)],
                                             cs[col_out((size_t)c4.p[2], 2, cz)], cs[col_out((size_t)c4.p[3], 2, cz)], sigma_HeII_at_ion_freq);
  const double path = c4.path * sc.dr1;
  if (sc.use_lls) {
    // Lyman-limit-system fog on the incoming HI column (evolve_point.F90:177-180)
    const double coldensh_LLS =
        A.lls_grid ? (double)A.lls_grid[(size_t)i + (size_t)g.n1 * ((size_t)j + (size_t)g.n2 * (size_t)k)] : sc.coldensh_lls;
    cin_HI = cin_HI + coldensh_LLS * path / sc.dr1;
  }
  // coldens (doric.f90:358-372): neufrac * ndens * path * abundance, from the left
  const double cout_HI = cin_HI + u_HI * path * (1.0 - abu_he);
  const double cout_HeI = cin_HeI + u_HeI * path * abu_he;
  const double cout_HeII = cin_HeII + u_HeII * path * abu_he;
  store_columns(cs, p, cz, cin_HI, cin_HeI, cin_HeII, cout_HI, cout_HeI, cout_HeII);
}

// Which 256 cells of a shell of `cnt` cells a block takes (-1: none).  Blocks are dealt round-robin over the 8 XCDs,
// each with its own L2: with block b on cells [256 b, 256 b + 256) the two rows of the previous shell that a row of
// cells reads are fetched by two or three XCDs.  Instead the blocks of one XCD (equal b mod 8) take one contiguous
// eighth of the shell, so that a previous-shell line is fetched by one L2 (the host rounds launches of 64 blocks or
// more up to a multiple of 8; the surplus blocks leave at once).
__device__ __forceinline__ int sweep_block_cell(int cnt) {
  const int nblk_shell = (cnt + BLOCK - 1) / BLOCK;
  int vb = (int)blockIdx.x;
  if ((gridDim.x & 7u) == 0) {
    const int chunk = (int)gridDim.x >> 3;
    vb = ((int)blockIdx.x & 7) * chunk + ((int)blockIdx.x >> 3);
  }
  if (vb >= nblk_shell) return -1;
  const int t = vb * BLOCK + (int)threadIdx.x;
  return t < cnt ? t : -1;
}

__global__ void __launch_bounds__(BLOCK, C2R_SWEEP_WAVES)
k_sweep_shell(SweepArgs A, const SrcDev *__restrict__ src, const int *__restrict__ active, int shell) {
  const int t = sweep_block_cell((int)shell_count(shell));
  if (t >= 0) sweep_cell(A, src[active[blockIdx.y]], shell, t);
}

__global__ void __launch_bounds__(BLOCK, C2R_SWEEP_WAVES)
k_sweep_shell_fast(SweepArgs A, const SrcDev *__restrict__ src, const int *__restrict__ active, ShellGeom G) {
  const int t = sweep_block_cell(24 * G.s * G.s + 2);
  if (t >= 0) sweep_cell_fast(A, src[active[blockIdx.y]], G, t);
}

// photon_loss_src_thread(tn) += ... (evolve_point.F90:312): sum the block partials of all shells of
// one sub-box round, per source (blockIdx.x).  Fixed shape (256 strided serial sums, then the
// block tree) => the same bits on every run.
__global__ void __launch_bounds__(BLOCK)
k_loss_finish(const double *__restrict__ loss_partial, int pitch, int count, double *__restrict__ loss_acc) {
  __shared__ double sh[BLOCK / 64];
  const double *p = loss_partial + (size_t)blockIdx.x * pitch;
  double a = 0.0;
  for (int i = threadIdx.x; i < count; i += BLOCK) a += p[i];
  const double tot = block_sum(a, sh);
  if (threadIdx.x == 0) loss_acc[blockIdx.x] = tot;
}

// A quick LOWER BOUND of that loss, to decide "this source goes on" without waiting for the full sum: of every
// `sample`-th block of the round's shells, 8 cells (every 32nd), each cell's bands spread over 32 lanes -- a thread
// evaluates one or two bands, so the launch lasts microseconds where a thread of k_loss walks all bands of
// its cell.  Every term is a non-negative photon count, so any subset, in any order, bounds the loss from below;
// a bound that clears the threshold of evolve_source.F90:136 with a factor 2 to spare is decisive, anything else
// is replaced by the full sum of k_loss.  No bit of any result depends on this kernel.
// The probes of several rounds of a batch in ONE launch (blockIdx.z counts the rounds): the rounds that were swept on
// trust are probed together, and two launches (this one and k_loss_finish_rounds) instead of two per round keep the
// probe's footprint on the device at a few microseconds.
struct ProbeRound {
  int s_lo, s_hi;   // shells of the round
  Box box;          // its sub-box
  int list_off;     // its active list in the batch's list buffer
  int nact;         // sources in that list
  int nblk;         // sampled blocks per source
  int partial_off;  // first partial sum of the round (nblk per source)
  int acc_off;      // first result of the round (one per source)
};
__global__ void __launch_bounds__(BLOCK)
k_loss_probe_rounds(Grid g, const SrcDev *__restrict__ src, const int *__restrict__ lists, const ProbeRound *__restrict__ rounds,
                    int multi, StepScalars sc, const BandData *__restrict__ bd, SedSet ss, const int *__restrict__ block_base,
                    double *__restrict__ loss_partial, int sample) {
  __shared__ double sh[BLOCK / 64];
  const ProbeRound &R = rounds[blockIdx.z];
  if ((int)blockIdx.x >= R.nblk || (int)blockIdx.y >= R.nact) return; // uniform per block
  const SrcDev &S = src[lists[R.list_off + blockIdx.y]];
  const int s_lo = R.s_lo, s_hi = R.s_hi;
  // every sample-th block counted from the END of the round (the surface of a box is mostly its outermost shell)
  const int B = block_base[s_hi + 1] - 1 - (int)blockIdx.x * sample;
  int lo = s_lo, hi = s_hi; // largest shell with block_base[shell] <= B
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (block_base[mid] <= B) lo = mid; else hi = mid - 1;
  }
  const int shell = lo;
  const long long cnt = shell_count(shell);
  const int slot = threadIdx.x & 31;
  const long long t = (long long)(B - block_base[shell]) * BLOCK + (threadIdx.x >> 5) * 32;
  double loss = 0.0;
  if (t < cnt) {
    int di, dj, dk;
    shell_decode(shell, (int)t, di, dj, dk);
    const bool inside = di >= R.box.lo[0] && di <= R.box.hi[0] && dj >= R.box.lo[1] && dj <= R.box.hi[1] && dk >= R.box.lo[2] &&
                        dk <= R.box.hi[2];
    const bool boundary = di == R.box.lo[0] || dj == R.box.lo[1] || dk == R.box.lo[2] || di == R.box.hi[0] || dj == R.box.hi[1] ||
                          dk == R.box.hi[2];
    if (inside && boundary) {
      const size_t cz = S.cz;
      const size_t p = (size_t)shell_offset(shell) + (size_t)t;
      const global_double *cs = (const global_double *)S.cols;
      const double cin_HI = cs[col_in(p, 0, cz)], cin_HeI = cs[col_in(p, 1, cz)], cin_HeII = cs[col_in(p, 2, cz)];
      const double cout_HI = cs[col_out(p, 0, cz)], cout_HeI = cs[col_out(p, 1, cz)], cout_HeII = cs[col_out(p, 2, cz)];
      if (cin_HI < max_coldensh) {
        double vol_ph;
        if (shell == 0) {
          vol_ph = sc.dr1 * sc.dr2 * sc.dr3;
        } else {
          const double path = sc_path(di, dj, dk) * sc.dr1;
          const double xs = sc.dr1 * (double)di, ys = sc.dr2 * (double)dj, zs = sc.dr3 * (double)dk;
          vol_ph = 4.0 * pi * (xs * xs + ys * ys + zs * zs) * path;
        }
        const double nf[NSED] = {S.nflux, S.nflux_sed[0], S.nflux_sed[1]};
        double po = 0.0;
        for (int sd = 0; sd < (multi ? NSED : 1); sd++) {
          if (!(nf[sd] > 0.0)) continue;
          for (int b = ss.lo[sd] + slot; b < ss.hi[sd]; b += 32)
            po += photo_out_band(*bd, sd, ss.photo_thick[sd], ss.photo_thin[sd], b, cin_HI, cout_HI, cin_HeI, cout_HeI, cin_HeII,
                                 cout_HeII, nf[sd]);
        }
        loss = po * sc.vol / vol_ph;
      }
    }
  }
  const double bs = block_sum(loss, sh);
  if (threadIdx.x == 0) loss_partial[(size_t)R.partial_off + (size_t)blockIdx.y * R.nblk + blockIdx.x] = bs;
}
// the partial sums of (round blockIdx.y, source blockIdx.x) added up, fixed shape as k_loss_finish
__global__ void __launch_bounds__(BLOCK)
k_loss_finish_rounds(const ProbeRound *__restrict__ rounds, const double *__restrict__ loss_partial, double *__restrict__ loss_acc) {
  __shared__ double sh[BLOCK / 64];
  const ProbeRound &R = rounds[blockIdx.y];
  if ((int)blockIdx.x >= R.nact) return;
  const double *p = loss_partial + (size_t)R.partial_off + (size_t)blockIdx.x * R.nblk;
  double a = 0.0;
  for (int i = threadIdx.x; i < R.nblk; i += BLOCK) a += p[i];
  const double tot = block_sum(a, sh);
  if (threadIdx.x == 0) loss_acc[R.acc_off + blockIdx.x] = tot;
}

// columns of one slot from shell order back to mesh order (diagnostic download only)
__global__ void __launch_bounds__(BLOCK)
k_col_to_grid(Grid g, SrcDev S, const double *__restrict__ cs, double *__restrict__ out) {
  const size_t q = (size_t)blockIdx.x * BLOCK + threadIdx.x;
  if (q >= g.ncell) return;
  const int i = (int)(q % g.n1), j = (int)((q / g.n1) % g.n2), k = (int)(q / ((size_t)g.n1 * g.n2));
  const int di = wrap0(i + 1 - S.i0 + g.l1, g.n1) - g.l1, dj = wrap0(j + 1 - S.j0 + g.l2, g.n2) - g.l2,
            dk = wrap0(k + 1 - S.k0 + g.l3, g.n3) - g.l3;
  const bool inside = di >= S.lo[0] && di <= S.hi[0] && dj >= S.lo[1] && dj <= S.hi[1] && dk >= S.lo[2] && dk <= S.hi[2];
  const size_t p = shell_position(di, dj, dk);
  for (int c = 0; c < 3; c++) out[q + c * g.ncell] = inside ? cs[col_out(p, c, S.cz)] : 0.0;
}

// ---------------------------------------------------------------------------------------------
// Rates of every cell from every source of the batch, accumulated in source order
// (evolve_point.F90:246-306 with photoion_rates, radiation_photoionrates.f90:108-277).
// rates layout: [phih | phihe0 | phihe1 | phiheat] each ncell.
#ifndef C2R_RATES_WAVES_ISO
#define C2R_RATES_WAVES_ISO 4
#endif
// Waves per SIMD the register allocation aims at.  The kernel is bound by instruction issue at any of these
// occupancies, so the setting only steers how many copies, spills and waits the compiler makes; measured per variant
// (isothermal: 20.1 / 19.8 / 19.8 ms at 5 / 4 / 3 waves -- 91 registers either way, so five waves still run;
// heating, one SED: 31.5 / 29.4 / 29.5 ms at 4 / 3 / 2 waves; heating, three SEDs: 443 / 466 / 425 ms in round 3, when
// the kernel still kept its per-SED results in scratch memory; round 4, without scratch and without the SED pair:
// 4 waves -- 128 registers, 4 of them spilled -- 358 against 372 ms at 2 waves, see photoion_rates_multi).
#ifndef C2R_RATES_WAVES_HEAT
#define C2R_RATES_WAVES_HEAT 3
#endif
#ifndef C2R_RATES_WAVES_HEAT_MULTI
#define C2R_RATES_WAVES_HEAT_MULTI 4
#endif
#ifndef C2R_RATES_XCD_CHUNK
#define C2R_RATES_XCD_CHUNK 128
#endif
#ifndef C2R_RATES_BAND_ROWS
#define C2R_RATES_BAND_ROWS 1
#endif
#ifndef C2R_RATES_PARK_RICOTTI
#define C2R_RATES_PARK_RICOTTI 0
#endif
#ifndef C2R_RATES_PARK_SUMS
#define C2R_RATES_PARK_SUMS 2
#endif
template <bool HEAT, bool MULTI>
__global__ void __launch_bounds__(BLOCK, MULTI ? (HEAT ? C2R_RATES_WAVES_HEAT_MULTI : 4) : (HEAT ? C2R_RATES_WAVES_HEAT : C2R_RATES_WAVES_ISO))
k_rates(Grid g, const SrcDev *__restrict__ src, int nsrc, StepScalars sc, const double *__restrict__ ndens,
        const double *__restrict__ xh_av, const double *__restrict__ xhe_av,
        const BandDataByRow *__restrict__ bdr, SedSet ss, double *__restrict__ rates, const int *__restrict__ tiles,
        const int *__restrict__ tile_ptr, const int *__restrict__ tile_src, int tile_base, int fresh) {
  const size_t nc = g.ncell;
  // One block = a tile of 8 x 8 x 4 cells, one wave = a 4 x 4 x 4 cube of it.  Neighbouring cells see
  // similar optical depths: the lanes of a cube mostly take the same branch of the bit-exact log (its
  // near-1 path is 10 % of all arguments, so a wave of 64 unrelated cells nearly always runs both) and
  // gather from few table lines -- 31.2 -> 26.9 ms per launch at 256^3 x 8 sources against 64
  // consecutive i.  With sub-boxes much smaller than the mesh a cube also keeps ~(w/(w+3))^3 of its lanes
  // busy for a box of width w instead of w/(w+63).  Loads are 16 segments of 32 B; the kernel is ALU-bound.
  // `tiles`, when given, lists the tiles that intersect a sub-box of the batch (built on the host): the
  // launch then holds only blocks with work, which keeps enough heavy waves resident per SIMD.  With it come
  // `tile_ptr` / `tile_src`: for each listed tile the sources (positions in `src`, ascending = source order)
  // whose sub-box reaches into it, so that a batch of hundreds of faint sources costs a cell only the sources
  // near it.  Without lists every cell walks all nsrc sources of the batch (few sources, boxes that fill the mesh).
  // the (invc, logc) table of the bit-exact log (2 KB) in LDS: two gathers per band iteration that no longer
  // queue behind the photo-table gathers in the vector memory path
#if !defined(C2R_NO_LOGTAB4)
  // ... with the log's power of two folded in (gm::LogEntry, 8 KB)
  __shared__ gm::LogEntry s_logtab[256];
  s_logtab[threadIdx.x] = gm::make_log_entry((int)threadIdx.x);
#else
  __shared__ double s_logtab[256];
  s_logtab[threadIdx.x] = gm::log_table()[threadIdx.x];
#endif
  __syncthreads();
  // the band data as uploaded (a BandDataByRow: the plain arrays and, behind them, the same numbers band by band); a kernel
  // chooses its reading of them by the TYPE it hands down -- the base for the array form (an upcast, not a reinterpretation)
  const BandData *const bd = bdr;
  // two polynomial constants of the log held in vector registers for the whole kernel (gm::LogPins): -0.25 ms per
  // launch in the isothermal kernel; the heating kernels, which have no registers to spare, lose 2.7 ms with them
  gm::LogPins pins_ = {0.0, 0.0};
  const gm::LogPins *pins = nullptr;
  if (!HEAT) {
    pins_ = gm::pin_log_constants();
    pins = &pins_;
  }
  const int ti = (g.n1 + 7) >> 3, tj = (g.n2 + 7) >> 3;
  // Workgroups are dealt to the 8 XCDs round-robin, and each XCD has its own L2.  A cube reads its columns as 32-byte
  // rows of shell faces, so the other half of every cache line belongs to the neighbouring cube: give each XCD a
  // contiguous run of tiles (a slab of the mesh), so that the neighbour's request finds the line in the same L2.
  int vb = (int)blockIdx.x;
#if C2R_RATES_XCD_CHUNK > 0
  {
    constexpr int C = C2R_RATES_XCD_CHUNK;
    const int full = (int)(gridDim.x / (8 * C)) * (8 * C);
    if (vb < full) {
      const int r = vb >> 3, xcd = vb & 7;
      vb = (r / C) * (8 * C) + xcd * C + (r % C);
    }
  }
#endif
  const int tile = tiles ? tiles[tile_base + vb] : tile_base + vb;
  const int bi = tile % ti, bj = (tile / ti) % tj, bk = tile / (ti * tj);
  const int lane = threadIdx.x & 63;
  const int w_ = threadIdx.x >> 6;
  const int i = bi * 8 + (w_ & 1) * 4 + (lane & 3), j = bj * 8 + (w_ >> 1) * 4 + ((lane >> 2) & 3), k = bk * 4 + (lane >> 4);
  if (i >= g.n1 || j >= g.n2 || k >= g.n3) return;
  // The cell's own quantities are needed once per source, after its band loops: only what those divisions use stays
  // in registers across the loops -- the three denominators h0 * nd * (1 - abu_he), ... (evaluated from the left, as
  // evolve_point.F90:288-296 does per source), not the four factors, and not the cell number, which is formed again
  // for the stores at the end.
  double den_HI, den_HeI, den_HeII, h1;
  double a_HI = 0.0, a_HeI = 0.0, a_HeII = 0.0, a_heat = 0.0;
  {
    const size_t q = (size_t)i + (size_t)g.n1 * ((size_t)j + (size_t)g.n2 * (size_t)k);
    const double nd = ndens[q];
    const double h0 = dmax(xh_av[q], epsilon);
    h1 = dmax(xh_av[q + nc], epsilon);
    const double he0 = dmax(xhe_av[q], epsilon), he1 = dmax(xhe_av[q + nc], epsilon);
    den_HI = h0 * nd * (1.0 - abu_he);
    den_HeI = he0 * nd * abu_he;
    den_HeII = he1 * nd * abu_he;
    // fresh: the first launch after set_rates_to_zero when the launch covers every cell -- the grids then need not
    // be zeroed first (4 x 8 bytes per cell written and read again: 1.3 ms per iteration at 256^3); 0 + x == x
    if (!fresh) {
      a_HI = rates[q];
      a_HeI = rates[q + nc];
      a_HeII = rates[q + 2 * nc];
      if (HEAT) a_heat = rates[q + 3 * nc];
    }
  }
  // secondary-ionisation parameters of this cell, i_state = h_av(1) (evolve_point.F90:255): once per cell,
  // not once per source
  // Values of a cell that the band loops do not touch can wait in LDS, one column per lane, where the register
  // allocation would otherwise spill them to scratch memory (bit 0 of the macros: the one-SED heating kernel, bit 1: the
  // three-SED heating kernel).  C2R_RATES_PARK_SUMS: the cell's denominators and running sums, touched once per source
  // (default: the three-SED kernel, which then fits four waves per SIMD without a private segment).
  // C2R_RATES_PARK_RICOTTI: the secondary-ionisation parameters too (default: nowhere -- six LDS reads per heating band
  // cost more than the registers: 369 against 357 ms per pass on one box).
  constexpr bool PARK = HEAT && (((C2R_RATES_PARK_SUMS) >> (MULTI ? 1 : 0)) & 1) != 0;
  constexpr bool PARK_RIC = HEAT && (((C2R_RATES_PARK_RICOTTI) >> (MULTI ? 1 : 0)) & 1) != 0;
  __shared__ double s_ric[PARK_RIC ? 6 * BLOCK : 1];
  // the three denominators of the cell and its four running sums, touched once per source: [den_HI, den_HeI, den_HeII,
  // a_HI, a_HeI, a_HeII, a_heat] x BLOCK
  __shared__ double s_den[PARK ? 7 * BLOCK : 1];
  Ricotti ric_ = {};
  if (HEAT) ric_ = ricotti_parameters(h1);
  if (PARK_RIC) {
#pragma unroll
    for (int n = 0; n < 3; n++) {
      s_ric[n * BLOCK + threadIdx.x] = ric_.y1R[n];
      s_ric[(3 + n) * BLOCK + threadIdx.x] = ric_.y2R[n];
    }
  }
  if (PARK) {
    s_den[threadIdx.x] = den_HI;
    s_den[BLOCK + threadIdx.x] = den_HeI;
    s_den[2 * BLOCK + threadIdx.x] = den_HeII;
    s_den[3 * BLOCK + threadIdx.x] = a_HI;
    s_den[4 * BLOCK + threadIdx.x] = a_HeI;
    s_den[5 * BLOCK + threadIdx.x] = a_HeII;
    s_den[6 * BLOCK + threadIdx.x] = a_heat;
  }
  const RicottiParked ric_parked = {(ParkedPtr)&s_ric[PARK_RIC ? threadIdx.x : 0], BLOCK};
  // (each lane reads back only what it wrote itself: no barrier)
  const auto &ric = [&]() -> const std::conditional_t<PARK_RIC, RicottiParked, Ricotti> & {
    if constexpr (PARK_RIC) return ric_parked; else return ric_;
  }();
  bool touched = false;
  const int slot = tile_base + vb;
  const int e0 = tile_ptr ? tile_ptr[slot] : 0, e1 = tile_ptr ? tile_ptr[slot + 1] : nsrc;
  for (int e = e0; e < e1; e++) {
    const SrcDev &S = src[tile_ptr ? tile_src[e] : e];
    // unwrapped offset rtpos - srcpos in [-mesh/2, mesh - mesh/2 - 1]
    int di = i + 1 - S.i0, dj = j + 1 - S.j0, dk = k + 1 - S.k0;
    di = wrap0(di + g.l1, g.n1) - g.l1;
    dj = wrap0(dj + g.l2, g.n2) - g.l2;
    dk = wrap0(dk + g.l3, g.n3) - g.l3;
    // Cells outside the source's last sub-box were never traced (evolve_source.F90:136-144): no
    // contribution.  (The reference's own marker is coldensh_out == 0, evolve_point.F90:120; every cell
    // of the box is traced exactly once, so "inside the box" is the same set and needs no zeroing.)
    const bool outside = di < S.lo[0] || di > S.hi[0] || dj < S.lo[1] || dj > S.hi[1] || dk < S.lo[2] || dk > S.hi[2];
    C2R_COUNT_LANES(3, !outside);
    if (outside) continue;
    touched = true;
    const size_t cz = S.cz;
    const size_t p = shell_position(di, dj, dk);
    global_double *cs = (global_double *)S.cols;
    const double cout_HI = cs[col_out(p, 0, cz)];
    const double cin_HI = cs[col_in(p, 0, cz)], cin_HeI = cs[col_in(p, 1, cz)], cin_HeII = cs[col_in(p, 2, cz)];
    const double cout_HeI = cs[col_out(p, 1, cz)], cout_HeII = cs[col_out(p, 2, cz)];
#if defined(C2R_RATES_DUMMY_TRAFFIC)
    // EXPERIMENT (never in the product): the column sweep's compulsory traffic -- 96 bytes per cell.source, half read, half
    // written -- moved by this kernel on top of its own, to see what a fused sweep + rates kernel could hope to hide
    // behind the band loops.  Reads come from another part of the source's block, the writes go back there unchanged.
    {
      const size_t p2 = (p + cz / 2) % cz;
      double d0 = cs[col_in(p2, 0, cz)], d1 = cs[col_in(p2, 1, cz)], d2 = cs[col_in(p2, 2, cz)];
      double d3 = cs[col_out(p2, 0, cz)], d4 = cs[col_out(p2, 1, cz)], d5 = cs[col_out(p2, 2, cz)];
      asm volatile("" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5)); // opaque: the stores are not "the value just loaded"
      cs[col_in(p2, 0, cz)] = d0; cs[col_in(p2, 1, cz)] = d1; cs[col_in(p2, 2, cz)] = d2;
      cs[col_out(p2, 0, cz)] = d3; cs[col_out(p2, 1, cz)] = d4; cs[col_out(p2, 2, cz)] = d5;
    }
#endif
    double vol_ph;
    if (di == 0 && dj == 0 && dk == 0) {
      vol_ph = sc.cellvol;
    } else {
      const double path = sc_path(di, dj, dk) * sc.dr1;
      const double xs = sc.dr1 * (double)di, ys = sc.dr2 * (double)dj, zs = sc.dr3 * (double)dk;
      const double dist2 = xs * xs + ys * ys + zs * zs;
      vol_ph = 4.0 * pi * dist2 * path;
    }
    // photoion_rates and the sums of this cell; with_loss: also return photo_out, which the kernel otherwise never
    // forms (one addition per band, registers that stay alive through the band loop, scalar registers short
    // enough already: 4 % of the launch when every wave pays it) -- two copies of the code, chosen per wave below
    auto rates_of_source = [&](auto with_loss) -> double {
      double photo_out = 0.0;
      if (cin_HI < max_coldensh) {
        PhotoOut o;
        if (MULTI) {
          const double nf[NSED] = {S.nflux, S.nflux_sed[0], S.nflux_sed[1]};
          if constexpr (HEAT && (C2R_RATES_BAND_ROWS)) // this kernel reads cross sections and factors band by band (BandDataByRow)
            photoion_rates_multi<HEAT>(*bdr, ss, cin_HI, cout_HI, cin_HeI, cout_HeI, cin_HeII, cout_HeII,
                                       vol_ph, nf, ric, o, &s_logtab[0], pins);
          else
            photoion_rates_multi<HEAT>(*bd, ss, cin_HI, cout_HI, cin_HeI, cout_HeI, cin_HeII, cout_HeII, vol_ph, nf, ric, o, &s_logtab[0], pins);
        } else {
          photoion_rates<HEAT>(*bd, ss.photo_thick[0], ss.photo_thin[0], ss.heat_thick[0], ss.heat_thin[0], cin_HI, cout_HI,
                               cin_HeI, cout_HeI, cin_HeII, cout_HeII, vol_ph, S.nflux, ric, o, &s_logtab[0], pins);
        }
        if (PARK) { // (volatile: read here, not hoisted back into registers)
          volatile __attribute__((address_space(3))) double *dn = (volatile __attribute__((address_space(3))) double *)&s_den[threadIdx.x];
          dn[3 * BLOCK] = dn[3 * BLOCK] + o.photo_HI / dn[0];
          dn[4 * BLOCK] = dn[4 * BLOCK] + o.photo_HeI / dn[BLOCK];
          dn[5 * BLOCK] = dn[5 * BLOCK] + o.photo_HeII / dn[2 * BLOCK];
          dn[6 * BLOCK] = dn[6 * BLOCK] + o.heat;
        } else {
          a_HI = a_HI + o.photo_HI / den_HI;
          a_HeI = a_HeI + o.photo_HeI / den_HeI;
          a_HeII = a_HeII + o.photo_HeII / den_HeII;
          if (HEAT) a_heat = a_heat + o.heat;
        }
        if (decltype(with_loss)::value) photo_out = o.photo_out;
      } else {
        // rates are zero: x + 0.0 == x
      }
      return photo_out;
    };
    // evolve_point.F90:310-315: a cell on the surface of the (final) sub-box loses photo_out * vol / vol_ph photons
    // through it.  For a source whose last round ended for geometric reasons (SrcDev::loss_lo >= 0) that loss is
    // the one that is kept: leave it in the cell's N_in(HI) slot, which nobody reads any more, for k_loss_stored.
    // Only the shells of the final round count; the others were done -- and counted, for a loss that is not kept
    // -- in earlier rounds.
    // (Isothermal kernels only: the heating kernels, three times the code and short of registers as they are, lose
    // 10 % of their launch to the second copy -- 34.0 against 31.2 ms -- where the isothermal one gains; heating runs
    // evaluate the kept loss with k_loss beside the rates launch, as rounds 1 and 2 did.)
    bool surface = false;
    if (!HEAT && S.loss_lo >= 0) { // uniform
      const int ia = di < 0 ? -di : di, ja = dj < 0 ? -dj : dj, ka = dk < 0 ? -dk : dk;
      const int shell = ia > ja ? (ia > ka ? ia : ka) : (ja > ka ? ja : ka);
      surface = (di == S.lo[0] || dj == S.lo[1] || dk == S.lo[2] || di == S.hi[0] || dj == S.hi[1] || dk == S.hi[2]) &&
                shell >= S.loss_lo;
    }
    if (!HEAT && __any(surface ? 1 : 0)) {
      const double photo_out = rates_of_source(std::true_type{});
      if (surface) cs[col_in(p, 0, cz)] = photo_out * sc.vol / vol_ph;
    } else {
      (void)rates_of_source(std::false_type{});
    }
  }
  if (touched || fresh) {
    const size_t q = (size_t)i + (size_t)g.n1 * ((size_t)j + (size_t)g.n2 * (size_t)k);
    if (PARK) {
      a_HI = s_den[3 * BLOCK + threadIdx.x];
      a_HeI = s_den[4 * BLOCK + threadIdx.x];
      a_HeII = s_den[5 * BLOCK + threadIdx.x];
      a_heat = s_den[6 * BLOCK + threadIdx.x];
    }
    rates[q] = a_HI;
    rates[q + nc] = a_HeI;
    rates[q + 2 * nc] = a_HeII;
    if (HEAT) rates[q + 3 * nc] = a_heat;
  }
}

// ---------------------------------------------------------------------------------------------
// evolve0D_global + do_chemistry (files_for_3D/evolve_point.F90:325-440, :444-646), one cell per lane.
constexpr int CHEM_HIST = 24; // buckets of the histogram of thermal sub-steps per cell (powers of two)
// The counters of a pass -- cells not converged, the histogram, the two longest chains -- are one atomic per wave each,
// and atomics of device scope on ONE address are served one after the other, some 6 ns apiece: with the 262 144
// single-wave blocks of 256^3 every such counter cost 1.6 ms of a 5 ms heating launch (measured: 6.7 ms with three of
// them, 5.1 with two, 3.6 with none).  So the waves spread them over CHEM_CTL_COPIES copies, a page apart, chosen by
// block index; k_chem_ctl_reduce folds the copies into the counters the host reads when the pass is over.
constexpr int CHEM_CTL_COPIES = 64, CHEM_CTL_STRIDE = 1024;            // copies; ints between two of them
constexpr int CHEM_CTL_MAXWORK = CHEM_HIST, CHEM_CTL_MAXNIT = CHEM_HIST + 1, CHEM_CTL_NOTCONV = CHEM_HIST + 2, CHEM_CTL_SLOTS = CHEM_HIST + 3;
#ifndef C2R_CHEM_BLOCK
#define C2R_CHEM_BLOCK 64
#endif
#ifndef C2R_CHEM_WAVES
#define C2R_CHEM_WAVES 2
#endif
#ifndef C2R_CHEM_XCD_CHUNK
#define C2R_CHEM_XCD_CHUNK 64
#endif
#ifndef C2R_CHEM_CUBES
#define C2R_CHEM_CUBES 1
#endif
// LDSTAB (heating only): the five cooling curves (32 KB) and the log's table (2 KB) in LDS.  A thermal sub-step is one
// long chain of dependent operations, two of whose links are memory round trips -- the table of log10(T), then ten
// cooling-curve entries -- and with two waves per SIMD nothing hides them: from LDS they cost a sixth.  Filling 34 KB
// per block only pays for cells known to sub-cycle, so the first launch of a pass, where most cells are done after
// a handful of sub-steps, reads the tables from global memory and the repacked tiers after it use this variant;
// blocks are two waves there, so that eight waves per compute unit still fit beside 4 x 34 KB.
constexpr int CHEM_BLOCK_LDS = 128;
#if defined(C2R_CHEM_NIT_HIST)
__device__ unsigned long long c2r_chem_nit[64 * 136];
#endif
template <bool HEAT, bool LDSTAB = false>
__global__ void __launch_bounds__(LDSTAB ? CHEM_BLOCK_LDS : C2R_CHEM_BLOCK, C2R_CHEM_WAVES)
k_chemistry(Grid g, StepScalars sc, double dt, const double *__restrict__ ndens, const double *__restrict__ xh,
            const double *__restrict__ xhe, double *__restrict__ xh_av, double *__restrict__ xhe_av,
            double *__restrict__ xh_int, double *__restrict__ xhe_int, float *__restrict__ temperature,
            const double *__restrict__ rates, int *__restrict__ ctl, double *__restrict__ rc_last,
            const float *__restrict__ clumping_grid, size_t q_first, size_t q_end, const int *__restrict__ list,
            int budget, int *__restrict__ deferred, int *__restrict__ ndeferred, double *__restrict__ packed, int cubes) {
  // Cells are taken from the range [q_first, q_end) or, when `list` is given, from list[q_first .. q_end).
  // cubes != 0 (a range of whole k-planes, mesh sizes multiples of 4): a wave takes a 4 x 4 x 4 cube of the range
  // instead of 64 consecutive cells.  A wave lasts as long as its slowest lane, and the cells that need more
  // do_chemistry iterations than their neighbours lie on surfaces (ionisation fronts, the edges of sub-boxes): a row of
  // 64 cells crosses such a surface in one or two cells, a cube in sixteen -- an eighth as many waves are held up
  // (profiles/r04_chem_nit.json: the first iteration of a time step in ionised gas has a mean of 2.2 iterations per cell
  // and of 3.0 per row-shaped wave).  Cubes of one XCD are neighbours (runs of C2R_CHEM_XCD_CHUNK): the other three
  // quarters of every 128-byte line a cube touches belong to the next cubes along i.
  // Heating runs: budget > 0 drops a cell whose thermal sub-cycling passes `budget` steps -- nothing of it is
  // stored -- and appends it to `deferred`, to be redone from scratch by a launch that holds only such cells
  // (c2r_global_pass_finish).  ctl: the spread counters (see CHEM_CTL_COPIES), among them the cells per power of
  // two of sub-steps.
  const size_t nc = g.ncell;
  int *const my_ctl = ctl + (size_t)(blockIdx.x & (CHEM_CTL_COPIES - 1)) * CHEM_CTL_STRIDE;
  __shared__ double s_cool[LDSTAB ? 5 * NCOOL : 1];
  __shared__ double s_log[LDSTAB ? 256 : 1];
  if (LDSTAB) {
    for (int n = (int)threadIdx.x; n < 5 * NCOOL; n += CHEM_BLOCK_LDS) s_cool[n] = sc.cd.cool[n];
    for (int n = (int)threadIdx.x; n < 256; n += CHEM_BLOCK_LDS) s_log[n] = gm::log_table()[n];
    __syncthreads();
    sc.cd.cool = s_cool;
    sc.cd.logtab = s_log;
  } else {
    sc.cd.logtab = nullptr; // known at compile time: coolin's choice of table folds away
  }
  size_t idx = q_first + (size_t)blockIdx.x * (LDSTAB ? CHEM_BLOCK_LDS : C2R_CHEM_BLOCK) + threadIdx.x;
  size_t q_cube = 0;
  if (!LDSTAB && cubes) {
    constexpr int WPB = C2R_CHEM_BLOCK / 64; // waves per block
    int vb = (int)blockIdx.x;
    {
      constexpr int C = C2R_CHEM_XCD_CHUNK;
      const int full = (int)(gridDim.x / (8 * C)) * (8 * C);
      if (vb < full) {
        const int r = vb >> 3, xcd = vb & 7;
        vb = (r / C) * (8 * C) + xcd * C + (r % C);
      }
    }
    const int wv = vb * WPB + (int)(threadIdx.x >> 6), lane_ = (int)(threadIdx.x & 63);
    // cubes == 1: 4 x 4 x 4 cells; cubes == 2: 8 x 4 x 2 (rows of 64 bytes)
    const int si = cubes == 2 ? 3 : 2, sk = cubes == 2 ? 1 : 2; // log2 of the extent along i and k (along j: 4)
    const int ci_n = g.n1 >> si, cj_n = g.n2 >> 2;
    const int ci = wv % ci_n, cj = (wv / ci_n) % cj_n, ck = wv / (ci_n * cj_n);
    const size_t plane = (size_t)g.n1 * g.n2;
    const size_t k0 = q_first / plane;
    const size_t i = (size_t)((ci << si) + (lane_ & ((1 << si) - 1))), j = (size_t)(4 * cj + ((lane_ >> si) & 3)),
                 k = k0 + (size_t)((ck << sk) + (lane_ >> (si + 2)));
    q_cube = i + (size_t)g.n1 * (j + (size_t)g.n2 * k);
    idx = q_cube < q_end ? q_first : q_end; // in range (the launch covers whole cubes), or beyond it
  }
  int notconv = 0;
  int bucket = -1;
  int work_done = 0, nit_done = 0;
  bool dropped = false;
  int q_dropped = 0;
  if (idx < q_end) {
    const size_t q = list ? (size_t)list[idx] : ((!LDSTAB && cubes) ? q_cube : idx);
    int work = 0;
    // clumping_point for type_of_clumping = 5 (evolve_point.F90:483-484; REAL(4) grid)
    const double clumping = clumping_grid ? (double)clumping_grid[q] : sc.clumping;
    IonStates ion;
    ion.h[0] = dmax(epsilon, xh_int[q]);       ion.h[1] = dmax(epsilon, xh_int[q + nc]);
    ion.h_old[0] = dmax(epsilon, xh[q]);       ion.h_old[1] = dmax(epsilon, xh[q + nc]);
    const double yh0_av_old = xh_av[q];
    ion.h_av[0] = dmax(epsilon, yh0_av_old);   ion.h_av[1] = dmax(epsilon, xh_av[q + nc]);
    ion.he[0] = dmax(epsilon, xhe_int[q]);     ion.he[1] = dmax(epsilon, xhe_int[q + nc]);
    ion.he[2] = dmax(epsilon, xhe_int[q + 2 * nc]);
    ion.he_old[0] = dmax(epsilon, xhe[q]);     ion.he_old[1] = dmax(epsilon, xhe[q + nc]);
    ion.he_old[2] = dmax(epsilon, xhe[q + 2 * nc]);
    const double yhe0_av_old = xhe_av[q], yhe2_av_old = xhe_av[q + 2 * nc];
    ion.he_av[0] = dmax(epsilon, yhe0_av_old); ion.he_av[1] = dmax(epsilon, xhe_av[q + nc]);
    ion.he_av[2] = dmax(epsilon, yhe2_av_old);
    const double ndens_p = ndens[q];
    const double phi_HI = rates[q], phi_HeI = rates[q + nc], phi_HeII = rates[q + 2 * nc];
    const double heat = HEAT ? rates[q + 3 * nc] : 0.0;

    // get_temperature_point (mat_ini_test.F90:469-487)
    double avg_temper, temper1, temp_av_old;
    if (HEAT) {
      avg_temper = (double)temperature[q + nc];
      temper1 = (double)temperature[q + 2 * nc];
    } else {
      avg_temper = sc.temper_val;
      temper1 = sc.temper_val;
    }
    temp_av_old = avg_temper;
    const double temper0 = temper1;
    RecCoef rc = sc.rc;
    const double path = 1.0;
    int nit = 0;
    for (;;) {
      nit++;
      const double temper2 = temper1;
      const double h0_old = ion.h_av[0], he0_old = ion.he_av[0], he2_old = ion.he_av[2];
      double de = electrondens(ndens_p, ion.h_av, ion.he_av);
      if (HEAT) ini_rec_colion_factors(avg_temper, rc);

      double yfrac, zfrac, y2afrac, y2bfrac;
      prepare_doric_factors(coldens(path, ion.h[0], ndens_p, (1.0 - abu_he)), coldens(path, ion.he[0], ndens_p, abu_he),
                            coldens(path, ion.he[1], ndens_p, abu_he), yfrac, zfrac, y2afrac, y2bfrac);
      doric(dt, de, ion, phi_HI, phi_HeI, phi_HeII, yfrac, zfrac, y2afrac, y2bfrac, rc, clumping);
      de = electrondens(ndens_p, ion.h_av, ion.he_av);
      prepare_doric_factors(coldens(path, ion.h[0], ndens_p, (1.0 - abu_he)), coldens(path, ion.he[0], ndens_p, abu_he),
                            coldens(path, ion.he[1], ndens_p, abu_he), yfrac, zfrac, y2afrac, y2bfrac);
      const double ionh0old = ion.h[0], ionh1old = ion.h[1];
      const double ionhe0old = ion.he[0], ionhe1old = ion.he[1], ionhe2old = ion.he[2];
      const double oldhav = ion.h_av[0], oldhe0av = ion.he_av[0], oldhe1av = ion.he_av[1];
      doric(dt, de, ion, phi_HI, phi_HeI, phi_HeII, yfrac, zfrac, y2afrac, y2bfrac, rc, clumping);
      ion.h[0] = (ion.h[0] + ionh0old) / 2.0;
      ion.h[1] = (ion.h[1] + ionh1old) / 2.0;
      ion.he[0] = (ion.he[0] + ionhe0old) / 2.0;
      ion.he[1] = (ion.he[1] + ionhe1old) / 2.0;
      ion.he[2] = (ion.he[2] + ionhe2old) / 2.0;
      ion.h_av[0] = (ion.h_av[0] + oldhav) / 2.0;
      ion.he_av[0] = (ion.he_av[0] + oldhe0av) / 2.0;
      ion.he_av[1] = (ion.he_av[1] + oldhe1av) / 2.0;
      de = electrondens(ndens_p, ion.h_av, ion.he_av);
      temper1 = temper0;
      if (HEAT) {
        if (thermal(sc.cd, dt, temper1, avg_temper, de, ndens_p, ion, heat, &work, budget)) {
          dropped = true;
          break;
        }
      }

      const double mfc = minimum_fractional_change, mfa = minimum_fraction_of_atoms;
      if ((fabs((ion.h_av[0] - h0_old) / ion.h_av[0]) < mfc || ion.h_av[0] < mfa) &&
          (fabs((ion.he_av[0] - he0_old) / ion.he_av[0]) < mfc || ion.he_av[0] < mfa) &&
          (fabs((ion.he_av[2] - he2_old) / ion.he_av[2]) < mfc || ion.he_av[2] < mfa) &&
          fabs((temper1 - temper2) / temper1) < mfc)
        break;
      if (nit > 400) break;
    }
    if (dropped) {
      q_dropped = (int)q;
    } else {
    if (HEAT) bucket = work > 0 ? 32 - __clz(work) : 0;
    work_done = work;
    nit_done = nit;
    // The reference keeps the coefficients in module-global variables (cgsconstants.f90:106-133): after
    // the global pass they hold what the LAST cell (mesh,mesh,mesh) computed last, and
    // photonstatistics:total_rates then uses those for every cell.  Export them for the host.
    if (HEAT && q == nc - 1) {
      const double v12[12] = {rc.arech0, rc.brech0, rc.areche0, rc.breche0, rc.oreche0, rc.areche1,
                              rc.breche1, rc.treche1, rc.colli_HI, rc.colli_HeI, rc.colli_HeII, rc.v};
      for (int n = 0; n < 12; n++) rc_last[n] = v12[n];
    }
    double temp_av_new = temp_av_old;
    if (HEAT) { // set_temperature_point (mat_ini_test.F90:491-502): stored as REAL(4)
      const float t0 = (float)temper1, t1 = (float)avg_temper;
      temperature[q] = t0;
      temperature[q + nc] = t1;
      temp_av_new = (double)t1;
    }
    const double mfc = minimum_fractional_change, mfa = minimum_fraction_of_atoms;
    if ((fabs(ion.h_av[0] - yh0_av_old) > mfc && fabs((ion.h_av[0] - yh0_av_old) / ion.h_av[0]) > mfc &&
         ion.h_av[0] > mfa) ||
        (fabs(ion.he_av[0] - yhe0_av_old) > mfc && fabs((ion.he_av[0] - yhe0_av_old) / ion.he_av[0]) > mfc &&
         ion.he_av[0] > mfa) ||
        (fabs(ion.he_av[2] - yhe2_av_old) > mfc && fabs((ion.he_av[2] - yhe2_av_old) / ion.he_av[2]) > mfc &&
         ion.he_av[2] > mfa) ||
        (fabs((temp_av_old - temp_av_new) / temp_av_new) > 1.0e-1 && fabs(temp_av_new - temp_av_old) > 100.0))
      notconv = 1;
    xh_int[q] = ion.h[0];       xh_int[q + nc] = ion.h[1];
    xh_av[q] = ion.h_av[0];     xh_av[q + nc] = ion.h_av[1];
    xhe_int[q] = ion.he[0];     xhe_int[q + nc] = ion.he[1];     xhe_int[q + 2 * nc] = ion.he[2];
    xhe_av[q] = ion.he_av[0];   xhe_av[q + nc] = ion.he_av[1];   xhe_av[q + 2 * nc] = ion.he_av[2];
    // what the next pass's column sweep reads of this cell (k_pack_state): neufrac * ndens, fractions clamped at epsilon
    packed[q] = dmax(ion.h_av[0], epsilon) * ndens_p;
    packed[q + nc] = dmax(ion.he_av[0], epsilon) * ndens_p;
    packed[q + 2 * nc] = dmax(ion.he_av[1], epsilon) * ndens_p;
    } // not dropped
  }
  const int lane = threadIdx.x & 63;
  if (HEAT) {
    // the dropped cells of the wave take consecutive places in the list: one atomic per wave
    const unsigned long long dm = __ballot(dropped);
    if (dm) {
      const int leader = __ffsll((long long)dm) - 1;
      int base = 0;
      if (lane == leader) base = atomicAdd(ndeferred, (int)__popcll(dm));
      base = __shfl(base, leader, 64);
      if (dropped) deferred[base + (int)__popcll(dm & ((1ull << lane) - 1ull))] = q_dropped;
    }
    // the longest chain of the launch: most thermal sub-steps and most do_chemistry iterations of one cell
    // (diagnostics, C2R_CHEM_LOG).  A plain look at the copy first: it only ever grows, so a stale value costs one
    // atomic too many and never one too few.
    int wmax = bucket >= 0 ? work_done : 0, nmax = bucket >= 0 ? nit_done : 0;
    for (int off = 32; off > 0; off >>= 1) {
      wmax = max(wmax, __shfl_xor(wmax, off, 64));
      nmax = max(nmax, __shfl_xor(nmax, off, 64));
    }
    if (lane == 0) {
      if (wmax > *(volatile int *)&my_ctl[CHEM_CTL_MAXWORK]) atomicMax(&my_ctl[CHEM_CTL_MAXWORK], wmax);
      if (nmax > *(volatile int *)&my_ctl[CHEM_CTL_MAXNIT]) atomicMax(&my_ctl[CHEM_CTL_MAXNIT], nmax);
    }
    // the histogram: one atomic per wave and distinct bucket
    unsigned long long live = __ballot(bucket >= 0);
    while (live) {
      const int leader = __ffsll((long long)live) - 1;
      const int b = __shfl(bucket, leader, 64);
      const unsigned long long same = __ballot(bucket == b);
      if (lane == leader) atomicAdd(&my_ctl[b], (int)__popcll(same));
      live &= ~same;
    }
  }
  // conv_flag = conv_flag + 1 (evolve_point.F90:423): integer count, order-independent
  const unsigned long long m = __ballot(notconv);
  if (lane == 0 && m) atomicAdd(&my_ctl[CHEM_CTL_NOTCONV], (int)__popcll(m));
#if defined(C2R_CHEM_NIT_HIST)
  // diagnostic build (tools/chem_nit.py): do_chemistry iterations per cell [0..63], and per wave the largest of its
  // lanes [64..127] and the sum over its lanes [128] -- a wave lasts as long as its slowest lane
  {
    unsigned long long *hc = c2r_chem_nit + (blockIdx.x & 63) * 136;
    const int nb = nit_done < 63 ? nit_done : 63;
    int nmx = nit_done;
    long long nsum = nit_done;
    for (int off = 32; off > 0; off >>= 1) {
      nmx = max(nmx, __shfl_xor(nmx, off, 64));
      nsum += __shfl_xor(nsum, off, 64);
    }
    if (idx < q_end) atomicAdd(hc + nb, 1ull);
    if (lane == 0) {
      atomicAdd(hc + 64 + (nmx < 63 ? nmx : 63), 1ull);
      atomicAdd(hc + 128, (unsigned long long)nsum);
    }
  }
#endif
}

// The spread counters of k_chemistry, folded (and zeroed for the next pass): the cells not converged are added to
// conv_flag, the histogram and the two maxima to hist (heating; null otherwise).
__global__ void k_chem_ctl_reduce(int *__restrict__ ctl, int *__restrict__ conv_flag, int *__restrict__ hist) {
  const int t = threadIdx.x;
  if (t >= CHEM_CTL_SLOTS) return;
  const bool is_max = t == CHEM_CTL_MAXWORK || t == CHEM_CTL_MAXNIT;
  int acc = 0;
  for (int k = 0; k < CHEM_CTL_COPIES; k++) {
    int *p = ctl + (size_t)k * CHEM_CTL_STRIDE + t;
    const int v = *p;
    *p = 0;
    acc = is_max ? max(acc, v) : acc + v;
  }
  if (t == CHEM_CTL_NOTCONV) *conv_flag += acc;
  else if (hist) hist[t] = is_max ? max(hist[t], acc) : hist[t] + acc;
}

// ---------------------------------------------------------------------------------------------
// spec_integration (radiation_tables.f90:172-422) for one SED: every entry of the photo and heating
// tables is a 513-point Romberg-weighted sum over frequency of an integrand that holds one exp(-tau*s(nu)).
// One thread = one (tau, band, thick|thin): it walks the frequencies in order (Vector_Romberg is a serial
// sum, romberg.f90:180-186) and carries the photo sum and the up-to-three heating sums of the band.
// Per-frequency factors that do not depend on tau (SED shape, Planck denominator, h(nu - nu_ion)) come
// from the host, computed with the same bit-exact exp/pow (build_sed_vectors below).
// vec layout per band b, frequency x (index b*513 + x): csfd, A (numerator prefix), D (denominator),
// H0, H1, H2 (heating factors) -- six arrays of 47*513.  out: device table layout, pitch NTAUP.
constexpr int SEDV = NFREQ * 513;
__global__ void __launch_bounds__(BLOCK)
k_build_tables(const double *__restrict__ vec, const double *__restrict__ tau_tab, const double *__restrict__ romw,
               const double *__restrict__ delta_freq, int heat, double *__restrict__ photo_thick,
               double *__restrict__ photo_thin, double *__restrict__ heat_thick, double *__restrict__ heat_thin) {
  const int it = blockIdx.x * BLOCK + threadIdx.x;
  const int b = blockIdx.y >> 1, thin = blockIdx.y & 1; // band (0-based), table kind
  if (it > NTAU) return;
  const double tau = tau_tab[it], df = delta_freq[b];
  const double *csfd = vec + (size_t)b * 513, *A = csfd + SEDV, *D = A + SEDV;
  const double *H0 = D + SEDV, *H1 = H0 + SEDV, *H2 = H1 + SEDV;
  const int nh = b < NB1 ? 1 : (b < NB1 + NB2 ? 2 : 3);
  double itg = 0.0, h0 = 0.0, h1 = 0.0, h2 = 0.0;
  for (int x = 0; x <= 512; x++) {
    const double s_ = csfd[x];
    double f = 0.0;
    if (tau * s_ < 700.0) { // :471
      double t = A[x];
      if (thin) t = t * s_;
      t = t * C2R_MATH_EXP(-tau * s_);
      f = t / D[x];
    }
    const double w = romw[x];
    itg = itg + f * df * w;
    if (heat) {
      h0 = h0 + (H0[x] * f) * df * w;
      if (nh > 1) h1 = h1 + (H1[x] * f) * df * w;
      if (nh > 2) h2 = h2 + (H2[x] * f) * df * w;
    }
  }
  double *pt = (thin ? photo_thin : photo_thick) + (size_t)b * NTAUP;
  pt[it] = itg;
  if (it == NTAU) pt[NTAU + 1] = itg; // the duplicated last row (read_table)
  if (heat) {
    // heating columns of the band (radiation_tables.f90:300-310, 343-390), 0-based
    const int c0 = b < NB1 ? 0 : (b < NB1 + NB2 ? 2 * (b + 1) - NB1 - 2 : 3 * (b + 1) - NB2 - 2 * NB1 - 3);
    double *ht = (thin ? heat_thin : heat_thick) + (size_t)c0 * NTAUP;
    const double hv[3] = {h0, h1, h2};
    for (int k = 0; k < nh; k++) {
      ht[(size_t)k * NTAUP + it] = hv[k];
      if (it == NTAU) ht[(size_t)k * NTAUP + NTAU + 1] = hv[k];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// photonstatistics.f90: the grid sums of state_before/state_after (:117-144, :208-234) and of
// total_rates (:150-203).  Fixed launch shape (STAT_BLOCKS x 256, grid-stride, block tree, then one
// finishing block) => the same bits on every run; the order differs from the reference's serial
// loops, so the sums agree with it to rounding (~1e-14 relative), not bit for bit.
constexpr int STAT_BLOCKS = 1024;

template <int NV>
__device__ __forceinline__ void stat_block_reduce(double (&v)[NV], double *partial) {
  __shared__ double sh[NV][BLOCK / 64];
  for (int n = 0; n < NV; n++) {
    double x = v[n];
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    if ((threadIdx.x & 63) == 0) sh[n][threadIdx.x >> 6] = x;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    double r = 0.0;
    for (int i = 0; i < BLOCK / 64; i++) r += sh[threadIdx.x][i];
    partial[(size_t)blockIdx.x * NV + threadIdx.x] = r;
  }
}

__global__ void __launch_bounds__(BLOCK)
k_state_sums(size_t nc, const double *__restrict__ ndens, const double *__restrict__ xh,
             const double *__restrict__ xhe, double *__restrict__ partial) {
  double v[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (size_t q = (size_t)blockIdx.x * BLOCK + threadIdx.x; q < nc; q += (size_t)gridDim.x * BLOCK) {
    const double nd = ndens ? ndens[q] : 1.0; // ndens == nullptr: plain sums of the fractions
    v[0] += nd * xh[q];
    v[1] += nd * xh[q + nc];
    v[2] += nd * xhe[q];
    v[3] += nd * xhe[q + nc];
    v[4] += nd * xhe[q + 2 * nc];
  }
  stat_block_reduce<5>(v, partial);
}

__global__ void __launch_bounds__(BLOCK)
k_total_rates(size_t nc, RecCoef rc, double clumping_scalar, const float *__restrict__ clumping_grid,
              const double *__restrict__ ndens, const double *__restrict__ xh, const double *__restrict__ xhe,
              double *__restrict__ partial) {
  double v[3] = {0.0, 0.0, 0.0};
  for (size_t q = (size_t)blockIdx.x * BLOCK + threadIdx.x; q < nc; q += (size_t)gridDim.x * BLOCK) {
    const double clumping = clumping_grid ? (double)clumping_grid[q] : clumping_scalar; // photonstatistics.f90:175-177
    const double nd = ndens[q];
    const double yh[2] = {xh[q], xh[q + nc]};
    const double yhe[3] = {xhe[q], xhe[q + nc], xhe[q + 2 * nc]};
    const double de = electrondens(nd, yh, yhe);
    v[0] += nd * (yh[1] * rc.brech0 * (1.0 - abu_he) + yhe[1] * rc.breche0 * abu_he * 0.04) * de * clumping;
    v[1] += nd * de * (yh[0] * rc.colli_HI + yhe[0] * rc.colli_HeI + yhe[1] * rc.colli_HeII);
    v[2] += nd * abu_he * clumping * (yhe[2] * 1.121 * rc.breche1 + yhe[1] * rc.breche0 * 0.96) * abu_he * de;
  }
  stat_block_reduce<3>(v, partial);
}

// minval(xh_av(:,:,:,0)), minval(xhe_av(:,:,:,0)) of the log line at evolve.F90:463-466 (a minimum is exact in any order)
__global__ void __launch_bounds__(BLOCK)
k_state_min(size_t nc, const double *__restrict__ xh, const double *__restrict__ xhe, double *__restrict__ partial) {
  __shared__ double sh[2][BLOCK / 64];
  double a = (double)INFINITY, b = (double)INFINITY;
  for (size_t q = (size_t)blockIdx.x * BLOCK + threadIdx.x; q < nc; q += (size_t)gridDim.x * BLOCK) {
    a = dmin(a, xh[q]);
    b = dmin(b, xhe[q]);
  }
  for (int off = 32; off > 0; off >>= 1) {
    a = dmin(a, __shfl_down(a, off, 64));
    b = dmin(b, __shfl_down(b, off, 64));
  }
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = a; sh[1][threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x < 2) {
    double r = sh[threadIdx.x][0];
    for (int i = 1; i < BLOCK / 64; i++) r = dmin(r, sh[threadIdx.x][i]);
    partial[(size_t)blockIdx.x * 2 + threadIdx.x] = r;
  }
}
__global__ void __launch_bounds__(BLOCK)
k_min_finish(const double *__restrict__ partial, int nblocks, double *__restrict__ out) {
  __shared__ double sh[2][BLOCK / 64];
  double a = (double)INFINITY, b = (double)INFINITY;
  for (int i = threadIdx.x; i < nblocks; i += BLOCK) {
    a = dmin(a, partial[(size_t)i * 2]);
    b = dmin(b, partial[(size_t)i * 2 + 1]);
  }
  for (int off = 32; off > 0; off >>= 1) {
    a = dmin(a, __shfl_down(a, off, 64));
    b = dmin(b, __shfl_down(b, off, 64));
  }
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = a; sh[1][threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x < 2) {
    double r = sh[threadIdx.x][0];
    for (int i = 1; i < BLOCK / 64; i++) r = dmin(r, sh[threadIdx.x][i]);
    out[threadIdx.x] = r;
  }
}

template <int NV>
__global__ void __launch_bounds__(BLOCK)
k_stat_finish(const double *__restrict__ partial, int nblocks, double *__restrict__ out) {
  __shared__ double sh[NV][BLOCK / 64];
  for (int n = 0; n < NV; n++) {
    double x = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += BLOCK) x += partial[(size_t)i * NV + n];
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    if ((threadIdx.x & 63) == 0) sh[n][threadIdx.x >> 6] = x;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    double r = 0.0;
    for (int i = 0; i < BLOCK / 64; i++) r += sh[threadIdx.x][i];
    out[threadIdx.x] = r;
  }
}

// evolve0D for one cell and one source (c2r_evolve0d): the column part (sweep_cell) and the rates part (as k_rates
// for one source) in one thread, the loss through the cell if the caller says it lies on its sub-box's surface.
template <bool HEAT, bool MULTI>
__global__ void __launch_bounds__(64)
k_evolve0d(SweepArgs A, SrcDev S, int di, int dj, int dk, const BandData *__restrict__ bd, SedSet ss, double *__restrict__ rates,
           double *__restrict__ loss_out) {
  if (threadIdx.x != 0) return;
  const Grid &g = A.g;
  const size_t nc = g.ncell;
  const int ia = di < 0 ? -di : di, ja = dj < 0 ? -dj : dj, ka = dk < 0 ? -dk : dk;
  const int shell = ia > ja ? (ia > ka ? ia : ka) : (ja > ka ? ja : ka);
  const size_t p = shell_position(di, dj, dk);
  sweep_cell(A, S, shell, (int)(p - (size_t)shell_offset(shell)));
  __threadfence_block();
  const global_double *cs = (const global_double *)S.cols;
  const size_t cz = S.cz;
  const double cin_HI = cs[col_in(p, 0, cz)], cin_HeI = cs[col_in(p, 1, cz)], cin_HeII = cs[col_in(p, 2, cz)];
  const double cout_HI = cs[col_out(p, 0, cz)], cout_HeI = cs[col_out(p, 1, cz)], cout_HeII = cs[col_out(p, 2, cz)];
  const int i = wrap0(S.i0 - 1 + di, g.n1), j = wrap0(S.j0 - 1 + dj, g.n2), k = wrap0(S.k0 - 1 + dk, g.n3);
  const size_t q = (size_t)i + (size_t)g.n1 * ((size_t)j + (size_t)g.n2 * (size_t)k);
  const double nd = A.ndens[q];
  const double h0 = dmax(A.xh_av[q], epsilon), h1 = dmax(A.xh_av[q + nc], epsilon);
  const double he0 = dmax(A.xhe_av[q], epsilon), he1 = dmax(A.xhe_av[q + nc], epsilon);
  double vol_ph;
  if (shell == 0) {
    vol_ph = A.sc.dr1 * A.sc.dr2 * A.sc.dr3;
  } else {
    const double path = sc_path(di, dj, dk) * A.sc.dr1;
    const double xs = A.sc.dr1 * (double)di, ys = A.sc.dr2 * (double)dj, zs = A.sc.dr3 * (double)dk;
    const double dist2 = xs * xs + ys * ys + zs * zs;
    vol_ph = 4.0 * pi * dist2 * path;
  }
  double photo_out = 0.0;
  if (cin_HI < max_coldensh) {
    Ricotti ric = {};
    if (HEAT) ric = ricotti_parameters(h1);
    PhotoOut o;
    if (MULTI) {
      const double nf[NSED] = {S.nflux, S.nflux_sed[0], S.nflux_sed[1]};
      photoion_rates_multi<HEAT>(*bd, ss, cin_HI, cout_HI, cin_HeI, cout_HeI, cin_HeII, cout_HeII, vol_ph, nf, ric, o);
    } else {
      photoion_rates<HEAT>(*bd, ss.photo_thick[0], ss.photo_thin[0], ss.heat_thick[0], ss.heat_thin[0], cin_HI, cout_HI, cin_HeI,
                           cout_HeI, cin_HeII, cout_HeII, vol_ph, S.nflux, ric, o);
    }
    rates[q] = rates[q] + o.photo_HI / (h0 * nd * (1.0 - abu_he));
    rates[q + nc] = rates[q + nc] + o.photo_HeI / (he0 * nd * abu_he);
    rates[q + 2 * nc] = rates[q + 2 * nc] + o.photo_HeII / (he1 * nd * abu_he);
    if (HEAT) rates[q + 3 * nc] = rates[q + 3 * nc] + o.heat;
    photo_out = o.photo_out;
  }
  if (loss_out) *loss_out = photo_out * A.sc.vol / vol_ph;
}

// Everything the reference's loop reduces over the grid after a global pass, in one sweep of the arrays
// (c2r_iteration): v[0..4] = k_state_sums with ndens over (xh_intermed, xhe_intermed), v[5..9] = the same without
// ndens (the means), v[10..12] = k_total_rates over (xh_av, xhe_av), and the minima of xh_av(0), xhe_av(0).  Every sum
// runs over the cells in the order of the single-purpose kernels (same grid, same stride, same block tree), so the
// numbers are theirs bit for bit.  rc_dev: the coefficients the chemistry pass left behind (non-isothermal runs).
constexpr int ITER_NV = 13;
__global__ void __launch_bounds__(BLOCK)
k_iter_stats(size_t nc, RecCoef rc_host, const double *__restrict__ rc_dev, double clumping_scalar,
             const float *__restrict__ clumping_grid, const double *__restrict__ ndens, const double *__restrict__ xh_int,
             const double *__restrict__ xhe_int, const double *__restrict__ xh_av, const double *__restrict__ xhe_av,
             double *__restrict__ partial, double *__restrict__ partial_min) {
  RecCoef rc = rc_host;
  if (rc_dev) {
    rc.arech0 = rc_dev[0]; rc.brech0 = rc_dev[1]; rc.areche0 = rc_dev[2]; rc.breche0 = rc_dev[3]; rc.oreche0 = rc_dev[4];
    rc.areche1 = rc_dev[5]; rc.breche1 = rc_dev[6]; rc.treche1 = rc_dev[7]; rc.colli_HI = rc_dev[8]; rc.colli_HeI = rc_dev[9];
    rc.colli_HeII = rc_dev[10]; rc.v = rc_dev[11];
  }
  double v[ITER_NV];
  for (int n = 0; n < ITER_NV; n++) v[n] = 0.0;
  double a = (double)INFINITY, b = (double)INFINITY;
  for (size_t q = (size_t)blockIdx.x * BLOCK + threadIdx.x; q < nc; q += (size_t)gridDim.x * BLOCK) {
    const double nd = ndens[q];
    const double i0 = xh_int[q], i1 = xh_int[q + nc], j0 = xhe_int[q], j1 = xhe_int[q + nc], j2 = xhe_int[q + 2 * nc];
    v[0] += nd * i0; v[1] += nd * i1; v[2] += nd * j0; v[3] += nd * j1; v[4] += nd * j2;
    v[5] += 1.0 * i0; v[6] += 1.0 * i1; v[7] += 1.0 * j0; v[8] += 1.0 * j1; v[9] += 1.0 * j2;
    const double clumping = clumping_grid ? (double)clumping_grid[q] : clumping_scalar; // photonstatistics.f90:175-177
    const double yh[2] = {xh_av[q], xh_av[q + nc]};
    const double yhe[3] = {xhe_av[q], xhe_av[q + nc], xhe_av[q + 2 * nc]};
    const double de = electrondens(nd, yh, yhe);
    v[10] += nd * (yh[1] * rc.brech0 * (1.0 - abu_he) + yhe[1] * rc.breche0 * abu_he * 0.04) * de * clumping;
    v[11] += nd * de * (yh[0] * rc.colli_HI + yhe[0] * rc.colli_HeI + yhe[1] * rc.colli_HeII);
    v[12] += nd * abu_he * clumping * (yhe[2] * 1.121 * rc.breche1 + yhe[1] * rc.breche0 * 0.96) * abu_he * de;
    a = dmin(a, yh[0]);
    b = dmin(b, yhe[0]);
  }
  stat_block_reduce<ITER_NV>(v, partial);
  __shared__ double shm[2][BLOCK / 64];
  for (int off = 32; off > 0; off >>= 1) {
    a = dmin(a, __shfl_down(a, off, 64));
    b = dmin(b, __shfl_down(b, off, 64));
  }
  if ((threadIdx.x & 63) == 0) { shm[0][threadIdx.x >> 6] = a; shm[1][threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x < 2) {
    double r = shm[threadIdx.x][0];
    for (int i = 1; i < BLOCK / 64; i++) r = dmin(r, shm[threadIdx.x][i]);
    partial_min[(size_t)blockIdx.x * 2 + threadIdx.x] = r;
  }
}
// out[0..12] the sums, out[13..14] the minima, out[15..26] the coefficients used
__global__ void __launch_bounds__(BLOCK)
k_iter_stats_finish(const double *__restrict__ partial, const double *__restrict__ partial_min, int nblocks, RecCoef rc_host,
                    const double *__restrict__ rc_dev, double *__restrict__ out) {
  __shared__ double sh[ITER_NV][BLOCK / 64];
  __shared__ double shm[2][BLOCK / 64];
  for (int n = 0; n < ITER_NV; n++) {
    double x = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += BLOCK) x += partial[(size_t)i * ITER_NV + n];
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    if ((threadIdx.x & 63) == 0) sh[n][threadIdx.x >> 6] = x;
  }
  double a = (double)INFINITY, b = (double)INFINITY;
  for (int i = threadIdx.x; i < nblocks; i += BLOCK) {
    a = dmin(a, partial_min[(size_t)i * 2]);
    b = dmin(b, partial_min[(size_t)i * 2 + 1]);
  }
  for (int off = 32; off > 0; off >>= 1) {
    a = dmin(a, __shfl_down(a, off, 64));
    b = dmin(b, __shfl_down(b, off, 64));
  }
  if ((threadIdx.x & 63) == 0) { shm[0][threadIdx.x >> 6] = a; shm[1][threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x < ITER_NV) {
    double r = 0.0;
    for (int i = 0; i < BLOCK / 64; i++) r += sh[threadIdx.x][i];
    out[threadIdx.x] = r;
  } else if (threadIdx.x < ITER_NV + 2) {
    const int m = threadIdx.x - ITER_NV;
    double r = shm[m][0];
    for (int i = 1; i < BLOCK / 64; i++) r = dmin(r, shm[m][i]);
    out[threadIdx.x] = r;
  } else if (threadIdx.x < ITER_NV + 2 + 12) {
    const int m = threadIdx.x - ITER_NV - 2;
    const double h[12] = {rc_host.arech0, rc_host.brech0, rc_host.areche0, rc_host.breche0, rc_host.oreche0, rc_host.areche1,
                          rc_host.breche1, rc_host.treche1, rc_host.colli_HI, rc_host.colli_HeI, rc_host.colli_HeII, rc_host.v};
    out[threadIdx.x] = rc_dev ? rc_dev[m] : h[m];
  }
}

// cosmo_evol's ndens(:,:,:) = ndens(:,:,:) / zfactor3 (cosmology.f90:193) on the device copy: one IEEE division per cell,
// correctly rounded on gfx950 as on the host (c2r_scale_ndens)
__global__ void __launch_bounds__(BLOCK) k_divide_by(double *__restrict__ a, size_t n, double d) {
  for (size_t i = (size_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * BLOCK) a[i] = a[i] / d;
}

} // namespace

// =============================================================================================
// host side
struct c2r_ctx {
  int device = 0;
  std::vector<c2r_ctx *> replicas; // further devices of a context made by c2r_create_multi (each a plain context)
  // sum over ranks (c2ray_comm.inc): kind 0 none, 1 RCCL, 2 in-process sum for replicas that share a device
  int comm_kind = 0, comm_rank = 0, comm_nranks = 1;
  void *comm = nullptr;            // ncclComm_t
  bool comm_broken = false;        // the communicator was aborted after an error inside a collective phase (c2ray_comm.inc)
  std::string comm_broken_why;
  long long slab_passes = 0;       // slab-wise passes this context has run (C2R_FAULT_INJECT counts them)
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_comm_a = nullptr, ev_comm_b = nullptr;
  std::vector<hipEvent_t> ev_sum;  // slab s of the rate grids is summed over the ranks
  hipStream_t stream = nullptr;
  Grid g{};
  std::string err;

  double *d_photo_thick = nullptr, *d_photo_thin = nullptr, *d_heat_thick = nullptr, *d_heat_thin = nullptr;
  // the heating tables as the kernels read them, interleaved by band (c2ray_device.hpp heat_interleave): [sed][thick, thin];
  // the column-wise copies above (and d_sed_tab[][2..3]) stay for c2r_download_tables and band_tau_zero
  double *d_heat_woven[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
  BandDataByRow *d_bands = nullptr;
  BandDataByRow h_bands{};         // host copy (tau_zero is refreshed whenever a table set changes); d_bands is its upload: the
                                   // other kernels take it as its base BandData, the three-SED heating kernel as what it is
  bool have_tables = false, have_heat_tables = false, have_bands = false, have_fvec = false;
  int bb_upper = 0;
  double *d_cool = nullptr;
  bool have_cool = false;
  double cool_mintemp = 1.0, cool_dtemp = 0.01;

  double *d_ndens = nullptr;
  StepScalars sc{};
  double zred = 0, H0 = 0, Omega0 = 0;
  int isothermal = 1;
  bool have_step = false;
  float *d_lls = nullptr, *d_clump = nullptr; // LLS_grid (type_of_LLS = 2), clumping_grid (type_of_clumping = 5)
  bool lls_on_grid = false, clumping_on_grid = false;

  int nsrc = 0;
  std::vector<int> srcpos;
  std::vector<double> normflux;
  double s_star = 0;
  // -DPL / -DQUASARS: power-law (0) and quasar-like (1) SEDs
  double *d_sed_tab[2][4] = {{nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr}};
  int sed_lo[2] = {0, 0}, sed_hi[2] = {0, 0}; // 0-based [lo, hi)
  bool have_sed[2] = {false, false}, have_sed_heat[2] = {false, false}, have_sed_limits[2] = {false, false};
  std::vector<double> normflux_sed[2];
  double s_star_sed[2] = {0, 0};

  double *d_xh = nullptr, *d_xhe = nullptr, *d_xh_av = nullptr, *d_xhe_av = nullptr, *d_xh_int = nullptr,
         *d_xhe_int = nullptr;
  float *d_temp = nullptr;
  bool have_state = false;

  double *d_rates = nullptr, *d_rates_own = nullptr;
  bool rates_zero_pending = false; // set_rates_to_zero has been asked for but not yet carried out (see flush_rates_zero)
  bool phiheat_dirty = false;      // a heating run has written phiheat since it was last zeroed
  size_t rates_count = 0;

  int batch = 256;                 // most sources per batch (c2r_set_batch); the scratch arena may allow fewer
  // Column scratch: per ping-pong set a list of segments; a source's block is cut from the current segment of
  // its set (shell-ordered arrays are prefixes of one another, so a block that turns out too small moves to a
  // deeper one by six copies).  A set that runs out of room gets another segment -- nothing that exists moves, no
  // sweep starts over.  Segments are zeroed when allocated and only ever hold finite columns afterwards.
  struct Segment {
    double *p = nullptr;
    size_t n = 0; // doubles
  };
  std::vector<Segment> segs[2];
  size_t seg_cur[2] = {0, 0}, seg_used[2] = {0, 0}; // bump allocation within a batch
  std::vector<Segment> vacated[2];                  // blocks whose source moved to a deeper one in this batch: reused first
  size_t arena_total = 0;                          // doubles in all segments
  std::vector<int> prev_nbox;      // per source: sub-boxes of the last pass (0: unknown), sizes the next block
  std::vector<int> prev_grow;      // per source: by how many sub-boxes its box grew from the pass before last to the last one
  int last_first = 0, last_stride = 0; // the share of the sources the last pass swept: first, first + stride, ... (arena_prepare plans the next step's scratch for it)
  bool arena_reserved = false;     // C2R_ARENA_RESERVE_GB has been looked at (first c2r_begin_step)
  bool in_pass = false;            // pass_list is running (arena statistics: allocations that land inside an iteration)
  long long arena_stats[5] = {0, 0, 0, 0, 0}; // segments allocated, of them inside a pass, doubles allocated, block moves, batch restarts
  SrcDev *d_src[2] = {nullptr, nullptr}, *h_src[2] = {nullptr, nullptr}; // source records of the two sets (h: pinned)
  int *d_list[2] = {nullptr, nullptr}, *h_list[2] = {nullptr, nullptr};  // active lists of a batch's rounds, one after another
  size_t list_cap = 0;             // ints per set
  double *d_loss_partial = nullptr, *d_loss_acc = nullptr;
  size_t loss_partial_cap = 0;     // doubles
  // the sampled losses of the rounds of a batch (one entry per round and active source), read back when the
  // sub-box loop needs them
  double *d_probe_partial = nullptr, *d_probe_acc = nullptr, *h_probe = nullptr;
  size_t probe_partial_cap = 0, probe_acc_cap = 0;
  void *d_probe_rounds = nullptr, *h_probe_rounds = nullptr; // ProbeRound[probe_rounds_cap], device and pinned host
  size_t probe_rounds_cap = 0;
  hipEvent_t ev_probe = nullptr;  // the probes launched last, and the copy of their results, are through
  hipEvent_t ev_round = nullptr;  // the shells those probes look at have been queued
  hipStream_t stream_probe = nullptr;           // probes run beside the next rounds' shells
  // losses the rates launch of a batch leaves behind (k_loss_stored), per ping-pong set
  double *d_final_partial[2] = {nullptr, nullptr}, *d_final_acc[2] = {nullptr, nullptr}, *h_final[2] = {nullptr, nullptr};
  size_t final_partial_cap[2] = {0, 0};
  // per batch of the open pass, in order: what photon_loss and sum_nbox get from each of its sources
  // (evolve_source.F90:233-236); a loss still on its way from the device is a slot of h_final[set]
  struct BatchTail {
    int set = 0;
    bool resolved = false;
    std::vector<double> loss;
    std::vector<int> slot; // -1: `loss` holds the value
    std::vector<int> nbox;
  };
  std::vector<BatchTail> tails;
  double *h_tail = nullptr;       // pinned: photon_loss(1:47), sum_nbox on their way to the reduction buffer
  // rates launches: listed tiles, and for each the sources that reach it (CSR), per set (h: pinned)
  int *d_tiles[2] = {nullptr, nullptr}, *h_tiles[2] = {nullptr, nullptr};
  int *d_tptr[2] = {nullptr, nullptr}, *h_tptr[2] = {nullptr, nullptr};
  int *d_tsrc[2] = {nullptr, nullptr}, *h_tsrc[2] = {nullptr, nullptr};
  size_t tsrc_cap[2] = {0, 0};
  size_t ntiles = 0;
  std::vector<int> tile_count;     // scratch of the CSR construction
  int *d_block_base = nullptr;     // device copy of block_base
  int blocks_total = 0;            // blocks of all shells 0..smax
  std::vector<int> block_base;     // first block of shell s in a partial-sum row
  std::vector<ShellGeom> shell_geom; // per-shell constants of k_sweep_shell_fast
  double *d_colgrid = nullptr;     // 3 ncell, diagnostic download
  double *d_stateT = nullptr;      // 6 ncell: k_pack_state's products in mesh order (3 ncell), then in (j,i,k) order
  bool packed_valid = false;       // the mesh-ordered half matches xh_av / xhe_av (a complete global pass wrote it)
  bool transposed_valid = false;   // ... and the (j,i,k)-ordered half too (that pass queued k_transpose_packed behind itself)
  size_t chem_cells = 0;           // cells the open global pass has been asked to do so far
  double *d_rc_last = nullptr;     // 12: coefficients left by the last cell of the chemistry pass
  double *d_stat = nullptr;        // STAT_BLOCKS*5 partials + 8 results
  double *h_stat = nullptr;        // pinned, 8
  double *d_iter = nullptr;        // k_iter_stats: STAT_BLOCKS*(ITER_NV+2) partials + 32 results
  double *h_iter = nullptr;        // pinned, 32
  int chem_first = 16;              // first rung of the next heating pass's ladder of sub-step ceilings (CHEM_FIRST_CEILING)
  // c2r_evolve0d: the column block of the source being traced cell by cell, and which source that is
  double *d_point_cols = nullptr;
  int point_ns = 0, point_niter = 0;
  double *d_point_loss = nullptr, *h_point_loss = nullptr;
  bool want_iter_stats = false;    // the global pass being closed is followed by k_iter_stats (c2r_iteration)
  double *h_loss = nullptr; // pinned, BATCH_MAX
  int *d_conv = nullptr;
  int *h_conv = nullptr;    // pinned
  long long last_conv = -1;  // non-converged cells of the last global pass (-1: a step has just begun); shapes the next pass's waves
  int last_src = 0;
  double *last_cols = nullptr;      // column block of the last source swept (c2r_download_columns)
  size_t last_cz = 0;
  int last_lo[3] = {0, 0, 0}, last_hi[3] = {0, 0, 0};

  double photon_loss[C2R_NFREQ] = {0};
  int sum_nbox = 0;

  // second stream for the rates kernels: the (latency-bound) column sweep of batch n+1 runs beside
  // the (ALU-bound) rates kernel of batch n; two scratch sets ping-pong between them
  hipStream_t stream2 = nullptr;
  hipStream_t stream3 = nullptr;    // takes every other slab of a cut rates launch (kernel tails overlap)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipEvent_t ev_transposed = nullptr; // the (j,i,k)-ordered state copies of this pass are complete
  hipEvent_t ev_chem_done = nullptr;  // the launches of a global pass are through (its transposition waits for this)
  hipEvent_t ev_sweep_done[2] = {nullptr, nullptr}, ev_rates_done[2] = {nullptr, nullptr};
  bool set_busy[2] = {false, false};
  std::vector<hipEvent_t> ev_pool; // timing events, grown on demand
  size_t ev_used = 0;

  int chem_pieces = 0;              // pieces of the open global pass (c2r_global_pass_cells)
  double chem_dt = 0.0;
  int *d_defer[2] = {nullptr, nullptr}; // cells dropped by a tier of the heating global pass (ping-pong lists)
  int *d_chemspread = nullptr;      // the per-wave counters of k_chemistry, CHEM_CTL_COPIES copies (k_chem_ctl_reduce)
  int *d_chemctl = nullptr;         // {count of list 0, count of list 1, histogram[CHEM_HIST]}
  // slab-wise hand-over of the rate grids (c2r_pass_sources_begin / _wait_slab / _end)
  bool pass_open = false;
  int pass_slabs = 0;
  std::vector<int> slab_k;          // k-plane boundaries of the slabs
  std::vector<hipEvent_t> ev_slab;
  std::vector<hipEvent_t> pass_tev;

  bool timing = false;
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  c2r_timing tm{};
};

static std::string g_create_error;

static int fail(c2r_ctx *c, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf; else g_create_error = buf;
  return 1;
}
#define HIPCHK(c, call)                                                                             \
  do {                                                                                              \
    hipError_t e_ = (call);                                                                         \
    if (e_ != hipSuccess) return fail(c, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// hipStreamSynchronize -- with a deadline where the stream may be waiting for OTHER RANKS (an RCCL communicator of more
// than one rank: the chemistry of a slab is queued behind the sum of that slab, the sum waits for every rank).  A rank
// that died, or returned from its pass with an error, never issues its share of the sum, and a plain synchronisation
// would then wait for ever.  C2R_COMM_TIMEOUT_S: seconds to wait (default 1800; 0: for ever).  The caller turns the
// error into an abort of the communicator (with_comm_abort), which is what ends the waiting kernels.
static int sync_stream(c2r_ctx *c, hipStream_t s, const char *what) {
  double limit = 0.0;
  if (c->comm_kind == 1 && c->comm_nranks > 1) {
    const char *e = getenv("C2R_COMM_TIMEOUT_S");
    limit = e ? atof(e) : 1800.0;
  }
  if (limit <= 0.0) {
    HIPCHK(c, hipStreamSynchronize(s));
    return 0;
  }
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spin = 0;; spin++) {
    const hipError_t e = hipStreamQuery(s);
    if (e == hipSuccess) return 0;
    if (e != hipErrorNotReady) return fail(c, "hipStreamQuery failed while waiting for %s: %s", what, hipGetErrorString(e));
    if (spin < 4096) std::this_thread::yield();
    else std::this_thread::sleep_for(std::chrono::microseconds(50));
    const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (waited > limit)
      return fail(c, "rank %d of %d waited %.0f s for %s: a peer rank has not issued its share of the sum over the ranks "
                  "(C2R_COMM_TIMEOUT_S)", c->comm_rank, c->comm_nranks, waited, what);
  }
}

// a context made by c2r_create_multi drives further devices through `replicas` (plain one-device contexts)
template <class F>
static int for_replicas(c2r_ctx *c, F f) {
  if (!c) return 0;
  for (c2r_ctx *r : c->replicas)
    if (int e = f(r)) {
      c->err = "device " + std::to_string(r->device) + ": " + r->err;
      return e;
    }
  return 0;
}

extern "C" const char *c2r_last_error(const c2r_ctx *c) { return c ? c->err.c_str() : "null context"; }
extern "C" const char *c2r_create_error(void) { return g_create_error.c_str(); }

// Zero a device range in pieces of at most 1 GiB.  (One hipMemset over more than 16 GiB -- a batch of 16
// scratch slots at 256^3, or any batch at 512^3 -- left the tail of the range untouched on ROCm 7.2;
// found because results then depended on the batch size.)
static hipError_t zero_device(void *ptr, size_t bytes, hipStream_t stream) {
  const size_t piece = (size_t)1 << 30;
  char *p = static_cast<char *>(ptr);
  for (size_t off = 0; off < bytes; off += piece) {
    hipError_t e = hipMemsetAsync(p + off, 0, std::min(piece, bytes - off), stream);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

static size_t block_doubles(int cap) { // doubles of a column block that holds shells 0..cap
  const size_t w = (size_t)(2 * cap + 1);
  return 6 * w * w * w;
}

// `n` doubles for a column block of set `set`, or nullptr when the device has no room for another segment (the
// caller then shrinks its batch).  Blocks are cut from the set's segments one after another; a new segment is at
// least half of what the set has so far (and 2 GB), so that their number stays small: hipMalloc costs ~25 ms
// per GB on this system, which is why nothing is allocated ahead of need.  `hint`: what the caller knows it will ask
// for in all (a new segment is made that large if the device has the room).
static double *arena_alloc(c2r_ctx *c, int set, size_t n, size_t hint = 0) {
  // a place some source of this batch has moved out of (its stream order protects it: whatever read or copied the old
  // block was queued on the sweep stream before anything the new owner will queue): the smallest that fits
  {
    std::vector<c2r_ctx::Segment> &v = c->vacated[set];
    size_t best = v.size();
    for (size_t i = 0; i < v.size(); i++)
      if (v[i].n >= n && (best == v.size() || v[i].n < v[best].n)) best = i;
    if (best < v.size()) {
      double *p = v[best].p;
      v.erase(v.begin() + (long)best);
      return p;
    }
  }
  for (;;) {
    if (c->seg_cur[set] < c->segs[set].size()) {
      c2r_ctx::Segment &sg = c->segs[set][c->seg_cur[set]];
      if (c->seg_used[set] + n <= sg.n) {
        double *p = sg.p + c->seg_used[set];
        c->seg_used[set] += n;
        return p;
      }
      c->seg_cur[set]++; // the rest of this segment stays unused in this batch
      c->seg_used[set] = 0;
      continue;
    }
    size_t have = 0;
    for (const c2r_ctx::Segment &sg : c->segs[set]) have += sg.n;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return nullptr;
    // a set may hold 45 % of what the device could give to column blocks: the two sets alternate, and a set that
    // took everything would leave the other with batches of a few sources
    size_t budget = (size_t)(0.45 * ((double)(free_b / sizeof(double)) + (double)c->arena_total));
    // C2R_ARENA_BUDGET_MB: a smaller budget per set, so that tests reach the release / shrink paths on small meshes
    if (const char *e = getenv("C2R_ARENA_BUDGET_MB")) budget = std::min(budget, (size_t)(atof(e) * 1.0e6 / sizeof(double)));
    if (have >= budget) return nullptr;
    const size_t room = std::min((size_t)(0.9 * (double)free_b) / sizeof(double), budget - have);
    if (n > room) return nullptr;
    // C2R_ARENA_MIN_SEGMENT_MB (tests): a smaller floor than 2 GB, so that small meshes reach the growth paths
    size_t floor_doubles = (size_t)1 << 28;
    if (const char *e = getenv("C2R_ARENA_MIN_SEGMENT_MB")) floor_doubles = std::max<size_t>(1024, (size_t)(atof(e) * 1.0e6 / sizeof(double)));
    // ... and a ceiling on the geometric growth: on this system a device allocation is free up to some tens of GB and
    // costs ~20 ms per GB beyond (0.97 s for a 45 GB segment, round 5), inside whatever iteration needs it
    size_t ceil_doubles = (size_t)2 << 30; // 16 GB
    if (const char *e = getenv("C2R_ARENA_MAX_SEGMENT_GB")) ceil_doubles = std::max<size_t>(floor_doubles, (size_t)(atof(e) * 1.0e9 / sizeof(double)));
    const size_t want = std::min(room, std::max(std::max(n, hint), std::max(std::min(have / 2, ceil_doubles), floor_doubles)));
    c2r_ctx::Segment sg;
    const auto t0 = std::chrono::steady_clock::now();
    if (hipMalloc(&sg.p, sizeof(double) * want) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    sg.n = want;
    // zeroed on the sweep stream: whatever uses the block is queued behind it there
    if (zero_device(sg.p, sizeof(double) * want, c->stream) != hipSuccess) { (void)hipFree(sg.p); return nullptr; }
    if (getenv("C2R_ARENA_LOG"))
      fprintf(stderr, "c2ray_hip: column scratch, set %d: segment %zu of %.2f GB (%.0f ms), %.2f GB in all\n", set,
              c->segs[set].size(), want * 8e-9, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(),
              (c->arena_total + want) * 8e-9);
    c->segs[set].push_back(sg);
    c->arena_total += want;
    c->arena_stats[0]++;
    if (c->in_pass) c->arena_stats[1]++;
    c->arena_stats[2] += (long long)want;
  }
}

// Give a set's segments back to the device.  Only when nothing in flight uses them: the caller has waited for the
// set's last rates launch and has not yet started a sweep on it.
static void arena_release(c2r_ctx *c, int set) {
  (void)hipStreamSynchronize(c->stream); // segments are zeroed on the sweep stream
  for (c2r_ctx::Segment &sg : c->segs[set]) {
    (void)hipFree(sg.p);
    c->arena_total -= sg.n;
    if (c->last_cols >= sg.p && c->last_cols < sg.p + sg.n) c->last_cols = nullptr;
  }
  c->segs[set].clear();
  c->vacated[set].clear();
  c->seg_cur[set] = c->seg_used[set] = 0;
  if (getenv("C2R_ARENA_LOG")) fprintf(stderr, "c2ray_hip: column scratch, set %d released, %.2f GB left in all\n", set, c->arena_total * 8e-9);
}

template <class T>
static int ensure_pair(c2r_ctx *c, T **d, T **h, size_t *cap, size_t need) {
  if (*cap >= need) return 0;
  if (*d) HIPCHK(c, hipFree(*d));
  if (h && *h) HIPCHK(c, hipHostFree(*h));
  *d = nullptr;
  if (h) *h = nullptr;
  const size_t want = need + need / 2 + 1024;
  HIPCHK(c, hipMalloc(d, sizeof(T) * want));
  if (h) HIPCHK(c, hipHostMalloc(h, sizeof(T) * want));
  *cap = want;
  return 0;
}

extern "C" int c2r_create(c2r_ctx **out, int device, const int mesh[3]) {
  if (!out || !mesh) return fail(nullptr, "c2r_create: null argument");
  if (mesh[0] < 2 || mesh[1] < 2 || mesh[2] < 2 || mesh[0] > 4096 || mesh[1] > 4096 || mesh[2] > 4096)
    return fail(nullptr, "c2r_create: mesh %d x %d x %d out of range [2,4096]", mesh[0], mesh[1], mesh[2]);
  // cell numbers travel as 32-bit integers through the tile lists and the chemistry's lists of deferred cells
  if ((double)mesh[0] * mesh[1] * mesh[2] >= 2147483648.0)
    return fail(nullptr, "c2r_create: mesh %d x %d x %d has 2^31 cells or more (the largest cube is 1290^3)", mesh[0], mesh[1], mesh[2]);
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0)
    return fail(nullptr, "c2r_create: no HIP device available (%s); this library has no CPU path",
                e == hipSuccess ? "device count 0" : hipGetErrorString(e));
  if (device < 0 || device >= ndev) return fail(nullptr, "c2r_create: device %d not in [0,%d)", device, ndev);
  c2r_ctx *c = new c2r_ctx;
  c->device = device;
  c->g.n1 = mesh[0]; c->g.n2 = mesh[1]; c->g.n3 = mesh[2];
  c->g.l1 = mesh[0] / 2; c->g.l2 = mesh[1] / 2; c->g.l3 = mesh[2] / 2;
  c->g.ncell = (size_t)mesh[0] * mesh[1] * mesh[2];
  c->g.smax = std::max(c->g.l1, std::max(c->g.l2, c->g.l3));
  c->g.colsize = (size_t)(2 * c->g.smax + 1) * (2 * c->g.smax + 1) * (2 * c->g.smax + 1);
  const size_t nc = c->g.ncell;
#define CR(call)                                                                                    \
  do {                                                                                              \
    hipError_t e_ = (call);                                                                         \
    if (e_ != hipSuccess) {                                                                         \
      fail(nullptr, "%s failed: %s", #call, hipGetErrorString(e_));                                 \
      c2r_destroy(c);                                                                               \
      return 1;                                                                                     \
    }                                                                                               \
  } while (0)
  CR(hipSetDevice(device));
  {
    int lo = 0, hi = 0; // numerically lower = higher priority
    CR(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CR(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi));
    CR(hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, lo));
    CR(hipStreamCreateWithPriority(&c->stream3, hipStreamNonBlocking, lo));
    {
      // C2R_PROBE_PRIO (diagnostic): 0 = the sweep stream's (high) priority, 1 = default priority, 2 = low
      const char *e = getenv("C2R_PROBE_PRIO");
      const int mode = e ? atoi(e) : 0;
      if (mode == 1) CR(hipStreamCreateWithFlags(&c->stream_probe, hipStreamNonBlocking));
      else CR(hipStreamCreateWithPriority(&c->stream_probe, hipStreamNonBlocking, mode == 2 ? lo : hi));
    }
    CR(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    CR(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
    CR(hipEventCreateWithFlags(&c->ev_transposed, hipEventDisableTiming));
    CR(hipEventCreateWithFlags(&c->ev_chem_done, hipEventDisableTiming));
  }
  for (int i = 0; i < 2; i++) {
    CR(hipEventCreateWithFlags(&c->ev_sweep_done[i], hipEventDisableTiming));
    CR(hipEventCreateWithFlags(&c->ev_rates_done[i], hipEventDisableTiming));
  }
  CR(hipMalloc(&c->d_ndens, sizeof(double) * nc));
  CR(hipMalloc(&c->d_xh, sizeof(double) * 2 * nc));
  CR(hipMalloc(&c->d_xhe, sizeof(double) * 3 * nc));
  CR(hipMalloc(&c->d_xh_av, sizeof(double) * 2 * nc));
  CR(hipMalloc(&c->d_xhe_av, sizeof(double) * 3 * nc));
  CR(hipMalloc(&c->d_xh_int, sizeof(double) * 2 * nc));
  CR(hipMalloc(&c->d_xhe_int, sizeof(double) * 3 * nc));
  CR(hipMalloc(&c->d_temp, sizeof(float) * 3 * nc));
  CR(hipMalloc(&c->d_stateT, sizeof(double) * 6 * nc));
  CR(hipMalloc(&c->d_rc_last, sizeof(double) * 12));
  CR(hipMemset(c->d_rc_last, 0, sizeof(double) * 12));
  CR(hipMalloc(&c->d_stat, sizeof(double) * (STAT_BLOCKS * 5 + 8)));
  CR(hipHostMalloc(&c->h_stat, sizeof(double) * 8));
  CR(hipMalloc(&c->d_iter, sizeof(double) * (STAT_BLOCKS * (ITER_NV + 2) + 32)));
  CR(hipHostMalloc(&c->h_iter, sizeof(double) * 32));
  c->rates_count = 4 * nc + C2R_NFREQ + 1;
  CR(hipMalloc(&c->d_rates_own, sizeof(double) * c->rates_count));
  c->d_rates = c->d_rates_own;
  CR(zero_device(c->d_rates, sizeof(double) * c->rates_count, nullptr));
  // the spread counters of the chemistry launches (CHEM_CTL_COPIES): here, not in the first global pass -- an allocation
  // is a device-wide synchronisation, and an outer iteration has exactly one, at its end
  CR(hipMalloc(&c->d_chemspread, sizeof(int) * (size_t)CHEM_CTL_COPIES * CHEM_CTL_STRIDE));
  CR(hipMemset(c->d_chemspread, 0, sizeof(int) * (size_t)CHEM_CTL_COPIES * CHEM_CTL_STRIDE)); // k_chem_ctl_reduce leaves it zeroed
  CR(hipDeviceSynchronize()); // phih_grid = 0 for initial output (evolve_data.F90:77,80)
  // block bookkeeping of the shells 0..smax
  c->block_base.assign(c->g.smax + 2, 0);
  for (int s = 0; s <= c->g.smax; s++)
    c->block_base[s + 1] = c->block_base[s] + (int)((shell_count(s) + BLOCK - 1) / BLOCK);
  c->blocks_total = c->block_base[c->g.smax + 1];
  for (int s = 0; s <= c->g.smax; s++) c->shell_geom.push_back(shell_geometry(s));
  CR(hipMalloc(&c->d_block_base, sizeof(int) * c->block_base.size()));
  CR(hipMemcpy(c->d_block_base, c->block_base.data(), sizeof(int) * c->block_base.size(), hipMemcpyHostToDevice));
  c->ntiles = (size_t)((mesh[0] + 7) / 8) * ((mesh[1] + 7) / 8) * ((mesh[2] + 3) / 4);
  for (int k = 0; k < 2; k++) {
    CR(hipMalloc(&c->d_tiles[k], sizeof(int) * c->ntiles));
    CR(hipHostMalloc(&c->h_tiles[k], sizeof(int) * c->ntiles));
    CR(hipMalloc(&c->d_tptr[k], sizeof(int) * (c->ntiles + 1)));
    CR(hipHostMalloc(&c->h_tptr[k], sizeof(int) * (c->ntiles + 1)));
    CR(hipMalloc(&c->d_src[k], sizeof(SrcDev) * BATCH_MAX));
    CR(hipHostMalloc(&c->h_src[k], sizeof(SrcDev) * BATCH_MAX));
    // active lists of all rounds of a batch, one after another (two per round at most, plus the final losses)
    c->list_cap = (size_t)(2 * (c->g.smax / SUBBOXSIZE + 2) + 4) * BATCH_MAX;
    CR(hipMalloc(&c->d_list[k], sizeof(int) * c->list_cap));
    CR(hipHostMalloc(&c->h_list[k], sizeof(int) * c->list_cap));
  }
  CR(hipMalloc(&c->d_loss_acc, sizeof(double) * BATCH_MAX));
  CR(hipHostMalloc(&c->h_loss, sizeof(double) * BATCH_MAX));
  CR(hipEventCreateWithFlags(&c->ev_probe, hipEventDisableTiming));
  CR(hipEventCreateWithFlags(&c->ev_round, hipEventDisableTiming));
  for (int k = 0; k < 2; k++) {
    CR(hipMalloc(&c->d_final_acc[k], sizeof(double) * BATCH_MAX));
    CR(hipHostMalloc(&c->h_final[k], sizeof(double) * BATCH_MAX));
  }
  CR(hipHostMalloc(&c->h_tail, sizeof(double) * (C2R_NFREQ + 1)));
  CR(hipMalloc(&c->d_conv, sizeof(int)));
  CR(hipHostMalloc(&c->h_conv, sizeof(int)));
  CR(hipMalloc(&c->d_bands, sizeof(BandDataByRow)));
  for (auto &ev : c->ev) CR(hipEventCreate(&ev));
#undef CR
  *out = c;
  return 0;
}

extern "C" void c2r_destroy(c2r_ctx *c) {
  if (!c) return;
  (void)c2r_comm_destroy(c);
  for (c2r_ctx *r : c->replicas) c2r_destroy(r);
  c->replicas.clear();
  (void)hipSetDevice(c->device);
  if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
  if (c->ev_comm_a) (void)hipEventDestroy(c->ev_comm_a);
  if (c->ev_comm_b) (void)hipEventDestroy(c->ev_comm_b);
  for (auto &ev : c->ev_sum) (void)hipEventDestroy(ev);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (auto &row : c->d_sed_tab)
    for (double *p : row)
      if (p) (void)hipFree(p);
  void *ptrs[] = {c->d_photo_thick, c->d_photo_thin, c->d_heat_thick, c->d_heat_thin, c->d_bands, c->d_cool,
                  c->d_ndens, c->d_xh, c->d_xhe, c->d_xh_av, c->d_xhe_av, c->d_xh_int, c->d_xhe_int, c->d_temp,
                  c->d_rates_own, c->d_stateT, c->d_loss_partial, c->d_loss_acc, c->d_conv, c->d_colgrid, c->d_rc_last, c->d_stat, c->d_lls, c->d_clump, c->d_block_base, c->d_defer[0], c->d_defer[1], c->d_chemctl, c->d_chemspread};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  for (auto &w : c->d_heat_woven)
    for (double *p : w)
      if (p) (void)hipFree(p);
  for (auto &list : c->segs)
    for (auto &sg : list)
      if (sg.p) (void)hipFree(sg.p);
  if (c->h_loss) (void)hipHostFree(c->h_loss);
  if (c->h_probe) (void)hipHostFree(c->h_probe);
  if (c->h_tail) (void)hipHostFree(c->h_tail);
  if (c->d_probe_partial) (void)hipFree(c->d_probe_partial);
  if (c->d_probe_rounds) (void)hipFree(c->d_probe_rounds);
  if (c->h_probe_rounds) (void)hipHostFree(c->h_probe_rounds);
  if (c->ev_probe) (void)hipEventDestroy(c->ev_probe);
  if (c->ev_round) (void)hipEventDestroy(c->ev_round);
  for (int k = 0; k < 2; k++) {
    if (c->d_final_partial[k]) (void)hipFree(c->d_final_partial[k]);
    if (c->d_final_acc[k]) (void)hipFree(c->d_final_acc[k]);
    if (c->h_final[k]) (void)hipHostFree(c->h_final[k]);
  }
  if (c->d_probe_acc) (void)hipFree(c->d_probe_acc);
  for (int k = 0; k < 2; k++) {
    void *dev[] = {c->d_tiles[k], c->d_tptr[k], c->d_tsrc[k], c->d_src[k], c->d_list[k]};
    void *host[] = {c->h_tiles[k], c->h_tptr[k], c->h_tsrc[k], c->h_src[k], c->h_list[k]};
    for (void *q : dev)
      if (q) (void)hipFree(q);
    for (void *q : host)
      if (q) (void)hipHostFree(q);
  }
  if (c->h_conv) (void)hipHostFree(c->h_conv);
  if (c->h_stat) (void)hipHostFree(c->h_stat);
  if (c->h_iter) (void)hipHostFree(c->h_iter);
  if (c->h_point_loss) (void)hipHostFree(c->h_point_loss);
  if (c->d_point_loss) (void)hipFree(c->d_point_loss);
  if (c->d_point_cols) (void)hipFree(c->d_point_cols);
  if (c->d_iter) (void)hipFree(c->d_iter);
  for (auto &ev : c->ev)
    if (ev) (void)hipEventDestroy(ev);
  for (auto &ev : c->ev_pool) (void)hipEventDestroy(ev);
  for (auto &ev : c->ev_slab) (void)hipEventDestroy(ev);
  for (int i = 0; i < 2; i++) {
    if (c->ev_sweep_done[i]) (void)hipEventDestroy(c->ev_sweep_done[i]);
    if (c->ev_rates_done[i]) (void)hipEventDestroy(c->ev_rates_done[i]);
  }
  if (c->stream2) { (void)hipStreamSynchronize(c->stream2); (void)hipStreamDestroy(c->stream2); }
  if (c->stream3) { (void)hipStreamSynchronize(c->stream3); (void)hipStreamDestroy(c->stream3); }
  if (c->stream_probe) { (void)hipStreamSynchronize(c->stream_probe); (void)hipStreamDestroy(c->stream_probe); }
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->ev_transposed) (void)hipEventDestroy(c->ev_transposed);
  if (c->ev_chem_done) (void)hipEventDestroy(c->ev_chem_done);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

// host (0:NumTau, ncol) -> device columns of pitch NTAUP with row 2001 = row 2000
static int upload_table(c2r_ctx *c, const double *src, int ncol, double **dst) {
  std::vector<double> tmp((size_t)ncol * NTAUP);
  for (int col = 0; col < ncol; col++) {
    const double *s = src + (size_t)col * (NTAU + 1);
    double *d = tmp.data() + (size_t)col * NTAUP;
    std::memcpy(d, s, sizeof(double) * (NTAU + 1));
    d[NTAU + 1] = s[NTAU];
  }
  if (!*dst) HIPCHK(c, hipMalloc(dst, sizeof(double) * tmp.size()));
  HIPCHK(c, hipMemcpy(*dst, tmp.data(), sizeof(double) * tmp.size(), hipMemcpyHostToDevice));
  return 0;
}

// BandData::tau_zero of SED `sed` (0 black body, 1 power law, 2 quasar) from the tables now on the device
// (uploaded or built there), and the band data to the device again.
static int refresh_tau_zero(c2r_ctx *c, int sed) {
  const double *tab[4] = {sed == 0 ? c->d_photo_thick : c->d_sed_tab[sed - 1][0], sed == 0 ? c->d_photo_thin : c->d_sed_tab[sed - 1][1],
                          sed == 0 ? c->d_heat_thick : c->d_sed_tab[sed - 1][2], sed == 0 ? c->d_heat_thin : c->d_sed_tab[sed - 1][3]};
  const bool have = sed == 0 ? c->have_tables : c->have_sed[sed - 1];
  const bool heat = sed == 0 ? c->have_heat_tables : c->have_sed_heat[sed - 1];
  for (int b = 0; b < NFREQ; b++) c->h_bands.tau_zero[sed][b] = (double)INFINITY;
  if (have) {
    std::vector<double> h[4];
    const int ncol[4] = {NFREQ, NFREQ, NHEAT, NHEAT};
    for (int t = 0; t < (heat ? 4 : 2); t++) {
      h[t].resize((size_t)ncol[t] * NTAUP);
      HIPCHK(c, hipMemcpy(h[t].data(), tab[t], sizeof(double) * h[t].size(), hipMemcpyDeviceToHost));
    }
    for (int b = 0; b < NFREQ; b++) {
      const double *cols[8];
      int n = 0;
      cols[n++] = &h[0][(size_t)b * NTAUP];
      cols[n++] = &h[1][(size_t)b * NTAUP];
      if (heat) {
        const int nh = b < NB1 ? 1 : (b < NB1 + NB2 ? 2 : 3);
        const int c0 = b < NB1 ? 0 : (b < NB1 + NB2 ? 2 * (b + 1) - NB1 - 2 : 3 * (b + 1) - NB2 - 2 * NB1 - 3);
        for (int k = 0; k < nh; k++) {
          cols[n++] = &h[2][(size_t)(c0 + k) * NTAUP];
          cols[n++] = &h[3][(size_t)(c0 + k) * NTAUP];
        }
      }
      c->h_bands.tau_zero[sed][b] = band_tau_zero(cols, n);
    }
    if (heat) { // the copies the kernels read (same numbers, woven by band)
      std::vector<double> woven((size_t)NHEAT * NTAUP);
      for (int t = 0; t < 2; t++) {
        heat_interleave(h[2 + t].data(), woven.data());
        if (!c->d_heat_woven[sed][t]) HIPCHK(c, hipMalloc(&c->d_heat_woven[sed][t], sizeof(double) * woven.size()));
        HIPCHK(c, hipMemcpy(c->d_heat_woven[sed][t], woven.data(), sizeof(double) * woven.size(), hipMemcpyHostToDevice));
      }
    }
  }
  if (c->have_bands) {
    band_rows_fill(c->h_bands);
    HIPCHK(c, hipMemcpy(c->d_bands, &c->h_bands, sizeof(BandDataByRow), hipMemcpyHostToDevice));
  }
  return 0;
}

static int set_tables_one(c2r_ctx *c, const double *photo_thick, const double *photo_thin,
                              const double *heat_thick, const double *heat_thin, const double *sigma_HI,
                              const double *sigma_HeI, const double *sigma_HeII, const double *const fvec[12],
                              int bb_upper) {
  if (!c) return 1;
  if (!sigma_HI || !sigma_HeI || !sigma_HeII) return fail(c, "c2r_set_tables: the cross sections are required");
  if ((photo_thick == nullptr) != (photo_thin == nullptr) || (heat_thick == nullptr) != (heat_thin == nullptr))
    return fail(c, "c2r_set_tables: thick and thin tables come in pairs");
  if (bb_upper < 1 || bb_upper > NFREQ) return fail(c, "c2r_set_tables: bb_upper %d not in [1,%d]", bb_upper, NFREQ);
  // The band loop of the rates kernels leaves out the terms of species that cannot absorb in a band and takes
  // the reciprocals of scale_int2/3 without operand scaling: both rest on the band layout of
  // radiation_sizes.f90:375-545 (sigma_HeI = 0 below the He I threshold, sigma_HeII = 0 below the He II
  // threshold, every other cross section an ordinary number), checked here once.
  for (int b = 0; b < NFREQ; b++) {
    const double lo = 0x1p-100, hi = 0x1p-30;
    const bool he1 = b >= NB1, he2 = b >= NB1 + NB2;
    const bool ok = sigma_HI[b] >= lo && sigma_HI[b] <= hi &&
                    (he1 ? (sigma_HeI[b] >= lo && sigma_HeI[b] <= hi) : sigma_HeI[b] == 0.0) &&
                    (he2 ? (sigma_HeII[b] >= lo && sigma_HeII[b] <= hi) : sigma_HeII[b] == 0.0);
    if (!ok)
      return fail(c, "c2r_set_tables: band %d has cross sections (%g, %g, %g): not the band layout of radiation_sizes.f90 "
                  "(He I / He II cross sections exactly 0 below their thresholds, all others within [2^-100, 2^-30])",
                  b + 1, sigma_HI[b], sigma_HeI[b], sigma_HeII[b]);
  }
  HIPCHK(c, hipSetDevice(c->device));
  // without tables this call only sets the band vectors; c2r_build_tables makes the tables on the device
  c->have_tables = false;
  if (photo_thick) {
    if (upload_table(c, photo_thick, NFREQ, &c->d_photo_thick)) return 1;
    if (upload_table(c, photo_thin, NFREQ, &c->d_photo_thin)) return 1;
    c->have_tables = true;
  }
  c->have_heat_tables = false;
  c->have_fvec = false;
  BandData bd;
  std::memset(&bd, 0, sizeof bd);
  std::memcpy(bd.sigma_HI, sigma_HI, sizeof bd.sigma_HI);
  std::memcpy(bd.sigma_HeI, sigma_HeI, sizeof bd.sigma_HeI);
  std::memcpy(bd.sigma_HeII, sigma_HeII, sizeof bd.sigma_HeII);
  bd.bb_upper = bb_upper;
  if (heat_thick && !fvec) return fail(c, "c2r_set_tables: heat tables given without the secondary-ionisation vectors");
  if (fvec) {
    for (int i = 0; i < 12; i++)
      if (!fvec[i]) return fail(c, "c2r_set_tables: fvec[%d] is NULL", i);
    if (heat_thick) {
      if (upload_table(c, heat_thick, NHEAT, &c->d_heat_thick)) return 1;
      if (upload_table(c, heat_thin, NHEAT, &c->d_heat_thin)) return 1;
    }
    double *dst[12] = {bd.f1ion_HI, bd.f1ion_HeI, bd.f1ion_HeII, bd.f2ion_HI, bd.f2ion_HeI, bd.f2ion_HeII,
                       bd.f1heat_HI, bd.f1heat_HeI, bd.f1heat_HeII, bd.f2heat_HI, bd.f2heat_HeI, bd.f2heat_HeII};
    for (int i = 0; i < 12; i++) std::memcpy(dst[i], fvec[i], sizeof(double) * (NFREQ - 1));
    c->have_fvec = true;
    c->have_heat_tables = heat_thick != nullptr;
  }
  for (int sd = 0; sd < 3; sd++) // the other SEDs keep theirs
    std::memcpy(bd.tau_zero[sd], c->h_bands.tau_zero[sd], sizeof bd.tau_zero[sd]);
  if (!c->have_bands)
    for (int sd = 0; sd < 3; sd++)
      for (int b = 0; b < NFREQ; b++) bd.tau_zero[sd][b] = (double)INFINITY;
  static_cast<BandData &>(c->h_bands) = bd;
  c->bb_upper = bb_upper;
  c->have_bands = true;
  return refresh_tau_zero(c, 0); // uploads the band data too
}

static int set_cooling_one(c2r_ctx *c, const double *cool, double mintemp, double dtemp) {
  if (!c) return 1;
  if (!cool || !(dtemp > 0.0)) return fail(c, "c2r_set_cooling: bad arguments");
  HIPCHK(c, hipSetDevice(c->device));
  if (!c->d_cool) HIPCHK(c, hipMalloc(&c->d_cool, sizeof(double) * 5 * NCOOL));
  HIPCHK(c, hipMemcpy(c->d_cool, cool, sizeof(double) * 5 * NCOOL, hipMemcpyHostToDevice));
  c->cool_mintemp = mintemp;
  c->cool_dtemp = dtemp;
  c->have_cool = true;
  return 0;
}

// The lists of the heating tiers (cells dropped by a launch, redone by the next) and their counters: made when a step
// is declared non-isothermal (c2r_set_step), zeroed on the stream their first user follows.
static int alloc_heating_lists(c2r_ctx *c) {
  if (c->isothermal || (c->d_defer[0] && c->d_defer[1] && c->d_chemctl)) return 0;
  // all three or none: an allocation that fails half-way (out of memory) gives back what it got, so that the next
  // c2r_set_step tries again instead of finding "the lists exist" with a null pointer among them (round-4 ADVICE)
  hipError_t e = hipSuccess;
  for (int k = 0; k < 2 && e == hipSuccess; k++)
    if (!c->d_defer[k]) e = hipMalloc(&c->d_defer[k], sizeof(int) * c->g.ncell);
  if (e == hipSuccess && !c->d_chemctl) e = hipMalloc(&c->d_chemctl, sizeof(int) * (4 + CHEM_HIST)); // two list counts, then the histogram
  if (e == hipSuccess) e = hipMemsetAsync(c->d_chemctl, 0, sizeof(int) * (4 + CHEM_HIST), c->stream);
  if (e != hipSuccess) {
    for (int k = 0; k < 2; k++) {
      if (c->d_defer[k]) (void)hipFree(c->d_defer[k]);
      c->d_defer[k] = nullptr;
    }
    if (c->d_chemctl) (void)hipFree(c->d_chemctl);
    c->d_chemctl = nullptr;
    (void)hipGetLastError();
    return fail(c, "c2r_set_step: the lists of the heating global pass could not be allocated: %s", hipGetErrorString(e));
  }
  return 0;
}

static int set_step_one(c2r_ctx *c, const double *ndens, const double dr[3], double vol, float clumping,
                            double zred, double H0, double Omega0, int isothermal, double temper_val,
                            const double reccoef[12]) {
  if (!c) return 1;
  if (!dr || !reccoef) return fail(c, "c2r_set_step: null argument");
  // ndens == nullptr (c2r_set_step_scalars): the density on the device stays as it is
  if (!ndens && !c->have_step) return fail(c, "c2r_set_step_scalars: no density on the device yet (c2r_set_step comes first)");
  HIPCHK(c, hipSetDevice(c->device));
  if (ndens) HIPCHK(c, hipMemcpyAsync(c->d_ndens, ndens, sizeof(double) * c->g.ncell, hipMemcpyHostToDevice, c->stream));
  c->isothermal = isothermal ? 1 : 0;
  if (alloc_heating_lists(c)) return 1; // a non-isothermal step: the tiers' lists exist before any pass starts
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->sc.dr1 = dr[0]; c->sc.dr2 = dr[1]; c->sc.dr3 = dr[2]; c->sc.vol = vol;
  c->sc.clumping = (double)clumping;
  c->sc.temper_val = temper_val;
  std::memcpy(&c->sc.rc, reccoef, sizeof(double) * 12);
  c->zred = zred; c->H0 = H0; c->Omega0 = Omega0;
  c->have_step = true;
  c->packed_valid = c->transposed_valid = false; // ndens may have changed
  return 0;
}

static int set_sources_one(c2r_ctx *c, int nsrc, const int *srcpos, const double *normflux, double s_star) {
  if (!c) return 1;
  if (nsrc < 0 || (nsrc > 0 && (!srcpos || !normflux))) return fail(c, "c2r_set_sources: bad arguments");
  for (int s = 0; s < nsrc; s++) {
    const int *p = srcpos + 3 * s;
    if (p[0] < 1 || p[0] > c->g.n1 || p[1] < 1 || p[1] > c->g.n2 || p[2] < 1 || p[2] > c->g.n3)
      return fail(c, "c2r_set_sources: source %d at (%d,%d,%d) outside the mesh", s + 1, p[0], p[1], p[2]);
  }
  // What the last pass learnt about each source -- how many sub-boxes it needed: the size of its column block and the
  // rounds that are swept without waiting for a loss probe -- stays valid across time steps as long as the source list
  // is the same (the reference's driver passes the same list at every evolve3D call of a redshift slice); a wrong
  // guess costs time, never a bit.  Forgetting it made every evolve3D call after the first regrow its column blocks
  // and probe every round again: 2.7 ms per iteration over the 8-9 iterations of such a call (round 3's
  // dropin timing: 26.0 against 23.1 ms).
  const bool same_list = nsrc == c->nsrc && c->srcpos.size() == 3 * (size_t)nsrc &&
                         std::equal(srcpos, srcpos + 3 * (size_t)nsrc, c->srcpos.begin());
  if (!same_list) {
    // another list (the next redshift slice: sources appear, disappear, change places in the list): what is known about a
    // cell's source goes with the cell.  Sources that share a cell share the larger count.
    std::unordered_map<long long, int> known;
    if (c->prev_nbox.size() == (size_t)c->nsrc && c->srcpos.size() == 3 * (size_t)c->nsrc)
      for (int s = 0; s < c->nsrc; s++) {
        const int *p = &c->srcpos[3 * (size_t)s];
        int &v = known[((long long)p[2] * 8192 + p[1]) * 8192 + p[0]];
        v = std::max(v, c->prev_nbox[(size_t)s]);
      }
    std::vector<int> carried((size_t)nsrc, 0);
    for (int s = 0; s < nsrc; s++) {
      const int *p = srcpos + 3 * (size_t)s;
      const auto it = known.find(((long long)p[2] * 8192 + p[1]) * 8192 + p[0]);
      if (it != known.end()) carried[(size_t)s] = it->second;
    }
    c->prev_nbox.swap(carried);
  }
  c->nsrc = nsrc;
  if (c->prev_nbox.size() != (size_t)nsrc) c->prev_nbox.assign((size_t)nsrc, 0);
  c->srcpos.assign(srcpos, srcpos + 3 * (size_t)nsrc);
  c->normflux.assign(normflux, normflux + nsrc);
  c->s_star = s_star;
  c->normflux_sed[0].clear();
  c->normflux_sed[1].clear();
  return 0;
}

static int set_sed_tables_one(c2r_ctx *c, int sed, const double *photo_thick, const double *photo_thin,
                                  const double *heat_thick, const double *heat_thin, int lower, int upper) {
  if (!c) return 1;
  if (sed < 1 || sed > 2) return fail(c, "c2r_set_sed_tables: sed = %d, expected 1 (power law) or 2 (quasar)", sed);
  if (lower < 1 || upper > NFREQ || lower > upper)
    return fail(c, "c2r_set_sed_tables: band range %d..%d not inside 1..%d", lower, upper, NFREQ);
  HIPCHK(c, hipSetDevice(c->device));
  const int k = sed - 1;
  c->have_sed[k] = false;
  c->have_sed_heat[k] = false;
  c->sed_lo[k] = lower - 1;
  c->sed_hi[k] = upper;
  c->have_sed_limits[k] = true;
  if (!photo_thick && !photo_thin) return 0; // band range only: c2r_build_tables makes the tables
  if (!photo_thick || !photo_thin) return fail(c, "c2r_set_sed_tables: thick and thin tables come in pairs");
  if (upload_table(c, photo_thick, NFREQ, &c->d_sed_tab[k][0])) return 1;
  if (upload_table(c, photo_thin, NFREQ, &c->d_sed_tab[k][1])) return 1;
  if (heat_thick && heat_thin) {
    if (upload_table(c, heat_thick, NHEAT, &c->d_sed_tab[k][2])) return 1;
    if (upload_table(c, heat_thin, NHEAT, &c->d_sed_tab[k][3])) return 1;
    c->have_sed_heat[k] = true;
  }
  c->sed_lo[k] = lower - 1;
  c->sed_hi[k] = upper;
  c->have_sed[k] = true;
  return refresh_tau_zero(c, sed);
}

static int set_sources_sed_one(c2r_ctx *c, int sed, const double *normflux, double s_star) {
  if (!c) return 1;
  if (sed < 1 || sed > 2) return fail(c, "c2r_set_sources_sed: sed = %d, expected 1 or 2", sed);
  const int k = sed - 1;
  if (!normflux) { c->normflux_sed[k].clear(); return 0; }
  if (!c->have_sed[k]) return fail(c, "c2r_set_sources_sed: c2r_set_sed_tables(%d) has not been called", sed);
  c->normflux_sed[k].assign(normflux, normflux + c->nsrc);
  c->s_star_sed[k] = s_star;
  return 0;
}

// The per-frequency vectors of k_build_tables, with the host versions of the bit-exact exp / pow:
// set_frequency_array, set_cross_section_freq_dependence and the tau-independent factors of
// fill_photo_integrands / fill_heating_integrands_* (radiation_tables.f90:432-783).
static void build_sed_vectors(const c2r_sed_setup &S, std::vector<double> &v) {
  v.assign((size_t)6 * SEDV, 0.0);
  double *csfd = v.data(), *A = csfd + SEDV, *D = A + SEDV, *H0 = D + SEDV, *H1 = H0 + SEDV, *H2 = H1 + SEDV;
  for (int b = 0; b < NFREQ; b++) {
    const double fmin = S.freq_min[b], df = S.delta_freq[b];
    for (int x = 0; x <= 512; x++) {
      const size_t q = (size_t)b * 513 + x;
      const double freq = fmin + df * (double)(float)x;
      csfd[q] = C2R_MATH_POW(freq / fmin, -S.xsec_index[b]);
      if (S.sed == 0) {
        if (freq * S.h_over_kT < 700.0) { // :474; otherwise the integrand is 0: A = 0, D = 1
          A[q] = 4.0 * S.pi * S.R_star2 * S.two_pi_over_c_square * freq * freq;
          D[q] = C2R_MATH_EXP(freq * S.h_over_kT) - 1.0;
        } else {
          A[q] = 0.0;
          D[q] = 1.0;
        }
      } else {
        A[q] = S.pl_scaling * C2R_MATH_POW(freq, -S.pl_index);
        D[q] = 1.0; // x / 1.0 == x
      }
      H0[q] = S.hplanck * (freq - S.ion_freq_HI);
      H1[q] = S.hplanck * (freq - S.ion_freq_HeI);
      H2[q] = S.hplanck * (freq - S.ion_freq_HeII);
    }
  }
}

static int build_tables_one(c2r_ctx *c, const c2r_sed_setup *S, int with_heat) {
  if (!c || !S) return 1;
  if (S->nfreq != 512) return fail(c, "c2r_build_tables: nfreq = %d, the band set-up is for NumFreq = 512", S->nfreq);
  if (S->sed < 0 || S->sed > 2) return fail(c, "c2r_build_tables: sed = %d, expected 0 (black body), 1 (power law), 2 (quasar)", S->sed);
  if (!S->freq_min || !S->delta_freq || !S->xsec_index || !S->tau || !S->romw) return fail(c, "c2r_build_tables: null vector");
  if (S->sed == 0 && !c->have_bands) return fail(c, "c2r_build_tables: c2r_set_tables (band vectors) has not been called");
  if (S->sed > 0 && !c->have_sed_limits[S->sed - 1]) return fail(c, "c2r_build_tables: c2r_set_sed_tables(%d) (band range) has not been called", S->sed);
  if (S->sed == 0 && with_heat && !c->have_fvec) return fail(c, "c2r_build_tables: heating tables need the secondary-ionisation vectors of c2r_set_tables");
  HIPCHK(c, hipSetDevice(c->device));
  std::vector<double> v;
  build_sed_vectors(*S, v);
  struct DevBuf { // freed on every way out, the error returns included
    double *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
  } buf_v, buf_small;
  HIPCHK(c, hipMalloc(&buf_v.p, sizeof(double) * v.size()));
  HIPCHK(c, hipMalloc(&buf_small.p, sizeof(double) * (NTAU + 1 + 513 + NFREQ)));
  double *d_v = buf_v.p, *d_small = buf_small.p;
  HIPCHK(c, hipMemcpy(d_v, v.data(), sizeof(double) * v.size(), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(d_small, S->tau, sizeof(double) * (NTAU + 1), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(d_small + NTAU + 1, S->romw, sizeof(double) * 513, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(d_small + NTAU + 1 + 513, S->delta_freq, sizeof(double) * NFREQ, hipMemcpyHostToDevice));
  double **tab = S->sed == 0 ? nullptr : c->d_sed_tab[S->sed - 1];
  double **pt = S->sed == 0 ? &c->d_photo_thick : &tab[0], **pn = S->sed == 0 ? &c->d_photo_thin : &tab[1];
  double **ht = S->sed == 0 ? &c->d_heat_thick : &tab[2], **hn = S->sed == 0 ? &c->d_heat_thin : &tab[3];
  if (!*pt) HIPCHK(c, hipMalloc(pt, sizeof(double) * (size_t)NFREQ * NTAUP));
  if (!*pn) HIPCHK(c, hipMalloc(pn, sizeof(double) * (size_t)NFREQ * NTAUP));
  if (with_heat) {
    if (!*ht) HIPCHK(c, hipMalloc(ht, sizeof(double) * (size_t)NHEAT * NTAUP));
    if (!*hn) HIPCHK(c, hipMalloc(hn, sizeof(double) * (size_t)NHEAT * NTAUP));
  }
  hipLaunchKernelGGL(k_build_tables, dim3((NTAU + 1 + BLOCK - 1) / BLOCK, 2 * NFREQ), dim3(BLOCK), 0, c->stream, d_v,
                     d_small, d_small + NTAU + 1, d_small + NTAU + 1 + 513, with_heat ? 1 : 0, *pt, *pn,
                     with_heat ? *ht : nullptr, with_heat ? *hn : nullptr);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (S->sed == 0) {
    c->have_tables = true;
    c->have_heat_tables = with_heat != 0;
  } else {
    c->have_sed[S->sed - 1] = true;
    c->have_sed_heat[S->sed - 1] = with_heat != 0;
  }
  return refresh_tau_zero(c, S->sed);
}

extern "C" int c2r_download_tables(c2r_ctx *c, int sed, double *photo_thick, double *photo_thin, double *heat_thick,
                                   double *heat_thin) {
  if (!c) return 1;
  if (sed < 0 || sed > 2) return fail(c, "c2r_download_tables: sed = %d", sed);
  HIPCHK(c, hipSetDevice(c->device));
  const bool have = sed == 0 ? c->have_tables : c->have_sed[sed - 1];
  const bool have_heat = sed == 0 ? c->have_heat_tables : c->have_sed_heat[sed - 1];
  if (!have) return fail(c, "c2r_download_tables: SED %d has no tables", sed);
  if ((heat_thick || heat_thin) && !have_heat) return fail(c, "c2r_download_tables: SED %d has no heating tables", sed);
  const double *src[4] = {sed == 0 ? c->d_photo_thick : c->d_sed_tab[sed - 1][0], sed == 0 ? c->d_photo_thin : c->d_sed_tab[sed - 1][1],
                          sed == 0 ? c->d_heat_thick : c->d_sed_tab[sed - 1][2], sed == 0 ? c->d_heat_thin : c->d_sed_tab[sed - 1][3]};
  double *dst[4] = {photo_thick, photo_thin, heat_thick, heat_thin};
  const int ncol[4] = {NFREQ, NFREQ, NHEAT, NHEAT};
  for (int t = 0; t < 4; t++) {
    if (!dst[t]) continue;
    std::vector<double> tmp((size_t)ncol[t] * NTAUP);
    HIPCHK(c, hipMemcpy(tmp.data(), src[t], sizeof(double) * tmp.size(), hipMemcpyDeviceToHost));
    for (int col = 0; col < ncol[t]; col++)
      std::memcpy(dst[t] + (size_t)col * (NTAU + 1), tmp.data() + (size_t)col * NTAUP, sizeof(double) * (NTAU + 1));
  }
  return 0;
}

static int set_lls_one(c2r_ctx *c, int use_lls, double coldensh_lls, const float *lls_grid) {
  if (!c) return 1;
  HIPCHK(c, hipSetDevice(c->device));
  c->sc.use_lls = use_lls ? 1 : 0;
  c->sc.coldensh_lls = coldensh_lls;
  c->lls_on_grid = false;
  if (use_lls && lls_grid) {
    if (!c->d_lls) HIPCHK(c, hipMalloc(&c->d_lls, sizeof(float) * c->g.ncell));
    HIPCHK(c, hipMemcpyAsync(c->d_lls, lls_grid, sizeof(float) * c->g.ncell, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->lls_on_grid = true;
  }
  return 0;
}

static int set_clumping_grid_one(c2r_ctx *c, const float *clumping_grid) {
  if (!c) return 1;
  HIPCHK(c, hipSetDevice(c->device));
  c->clumping_on_grid = false;
  if (clumping_grid) {
    if (!c->d_clump) HIPCHK(c, hipMalloc(&c->d_clump, sizeof(float) * c->g.ncell));
    HIPCHK(c, hipMemcpyAsync(c->d_clump, clumping_grid, sizeof(float) * c->g.ncell, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->clumping_on_grid = true;
  }
  return 0;
}

static int upload_state_one(c2r_ctx *c, const double *xh, const double *xhe, const float *temperature) {
  if (!c) return 1;
  if (!xh || !xhe) return fail(c, "c2r_upload_state: null argument");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t nc = c->g.ncell;
  HIPCHK(c, hipMemcpyAsync(c->d_xh, xh, sizeof(double) * 2 * nc, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_xhe, xhe, sizeof(double) * 3 * nc, hipMemcpyHostToDevice, c->stream));
  if (temperature)
    HIPCHK(c, hipMemcpyAsync(c->d_temp, temperature, sizeof(float) * 3 * nc, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->have_state = true;
  return 0;
}

extern "C" int c2r_download_state(c2r_ctx *c, double *xh, double *xhe, float *temperature) {
  if (!c) return 1;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t nc = c->g.ncell;
  if (xh) HIPCHK(c, hipMemcpyAsync(xh, c->d_xh, sizeof(double) * 2 * nc, hipMemcpyDeviceToHost, c->stream));
  if (xhe) HIPCHK(c, hipMemcpyAsync(xhe, c->d_xhe, sizeof(double) * 3 * nc, hipMemcpyDeviceToHost, c->stream));
  if (temperature)
    HIPCHK(c, hipMemcpyAsync(temperature, c->d_temp, sizeof(float) * 3 * nc, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

static int check_ready(c2r_ctx *c, const char *who) {
  if (!c->have_tables) return fail(c, "%s: c2r_set_tables has not been called", who);
  if (!c->have_step) return fail(c, "%s: c2r_set_step has not been called", who);
  if (!c->have_state) return fail(c, "%s: c2r_upload_state has not been called", who);
  if (!c->isothermal) {
    if (!c->have_heat_tables) return fail(c, "%s: non-isothermal run needs the heating tables", who);
    if (!c->have_cool) return fail(c, "%s: non-isothermal run needs c2r_set_cooling", who);
    for (int k = 0; k < 2; k++)
      if (c->have_sed[k] && !c->normflux_sed[k].empty() && !c->have_sed_heat[k])
        return fail(c, "%s: non-isothermal run needs the heating tables of SED %d", who, k + 1);
  }
  return 0;
}

static SedSet sedset(c2r_ctx *c, bool *multi) {
  SedSet ss;
  const BandData *unused = nullptr;
  (void)unused;
  ss.photo_thick[0] = c->d_photo_thick; ss.photo_thin[0] = c->d_photo_thin;
  // heating tables: the interleaved copies (made with tau_zero whenever a SED's tables are set or built)
  ss.heat_thick[0] = c->have_heat_tables ? c->d_heat_woven[0][0] : nullptr;
  ss.heat_thin[0] = c->have_heat_tables ? c->d_heat_woven[0][1] : nullptr;
  ss.lo[0] = 0; ss.hi[0] = c->bb_upper;
  *multi = false;
  for (int k = 0; k < 2; k++) {
    const bool on = c->have_sed[k] && !c->normflux_sed[k].empty();
    ss.photo_thick[k + 1] = on ? c->d_sed_tab[k][0] : nullptr;
    ss.photo_thin[k + 1] = on ? c->d_sed_tab[k][1] : nullptr;
    ss.heat_thick[k + 1] = on && c->have_sed_heat[k] ? c->d_heat_woven[k + 1][0] : nullptr;
    ss.heat_thin[k + 1] = on && c->have_sed_heat[k] ? c->d_heat_woven[k + 1][1] : nullptr;
    ss.lo[k + 1] = on ? c->sed_lo[k] : 0;
    ss.hi[k + 1] = on ? c->sed_hi[k] : 0;
    *multi = *multi || on;
  }
  return ss;
}

static StepScalars scalars(c2r_ctx *c) {
  StepScalars s = c->sc;
  s.cellvol = s.dr1 * s.dr2 * s.dr3;
  s.cd.cool = c->d_cool;
  s.cd.mintemp = c->cool_mintemp;
  s.cd.dtemp = c->cool_dtemp;
  s.cd.zred = c->zred;
  s.cd.H0 = c->H0;
  s.cd.Omega0 = c->Omega0;
  s.cd.logtab = nullptr;
  return s;
}

// Column scratch for the passes of the step that begins, sized from what the LAST pass learnt about every source (sub-box
// counts carried from time step to time step, and with a source's cell from source list to source list): for each ping-pong set
// the blocks of its largest batch, in ONE segment.  A device allocation costs ~25 ms per GB here and synchronises the device;
// made now, it does not land inside an outer iteration (round-4 VERDICT: 0.3-0.8 s of such allocations, and a batch that
// started over, inside iterations 6 and 7 of profiles/r04_config4_call.json).  What cannot be known -- a first time step from a
// neutral start, a source that brightens -- still grows inside a pass, as before.
// Shells a source's column block is made for: the sub-boxes it needed in the last pass plus one round -- plus as many as its
// box grew by between the last two passes when it is growing faster than that (while the ionisation fronts of a first
// time step break out, boxes jump several rounds per iteration: a block sized for one more round then moved to one twice
// as deep in the middle of the sweep, its old place lay idle until the batch was over, and a batch that ran out of room
// that way started over); four rounds when nothing is known; never more than the mesh.
static int predicted_shells(const c2r_ctx *c, int ns) {
  const int prev = c->prev_nbox[(size_t)ns - 1];
  const int grow = c->prev_grow.size() == c->prev_nbox.size() ? c->prev_grow[(size_t)ns - 1] : 0;
  const int cap = std::min(c->g.smax, SUBBOXSIZE * (prev > 0 ? prev + std::max(1, 2 * grow) : 4));
  // within a fifth of the mesh limit: the limit (a block less than twice as large, and never a move)
  return 5 * cap >= 4 * c->g.smax ? c->g.smax : cap;
}

static int arena_prepare(c2r_ctx *c) {
  // C2R_ARENA_RESERVE_GB="a[,b]": column scratch of set 0 (and set 1) made when the FIRST step begins, for hosts that know
  // what their source lists will need.  A device allocation beyond the first tens of GB costs ~24 ms per GB on this system
  // and stalls every launch of the process while it lasts, whichever thread makes it (tools/micro/malloc_overlap.hip,
  // profiles/r05_malloc_overlap.txt: 0.97 s for 40 GB, the kernel stream of another thread idle for all of it): it can be
  // moved, not hidden -- this moves it in front of the first iteration.
  if (!c->arena_reserved) {
    c->arena_reserved = true;
    if (const char *e = getenv("C2R_ARENA_RESERVE_GB")) {
      double gb[2] = {0.0, 0.0};
      const int n = sscanf(e, "%lf,%lf", &gb[0], &gb[1]);
      for (int set = 0; set < 2 && set < std::max(n, 0); set++) {
        const size_t want = (size_t)(gb[set] * 1.0e9 / sizeof(double));
        size_t have = 0;
        for (const c2r_ctx::Segment &sg : c->segs[set]) have += sg.n;
        if (want > have) {
          c->seg_cur[set] = c->segs[set].size(); // a new segment, whatever the existing ones have free
          c->seg_used[set] = 0;
          (void)arena_alloc(c, set, want - have, want - have);
          c->seg_cur[set] = c->seg_used[set] = 0;
        }
      }
    }
  }
  if (c->last_stride < 1 || c->prev_nbox.size() != (size_t)c->nsrc) return 0;
  std::vector<int> mine; // this context's share, as the last pass had it (do_grid_static: first = 1 + rank, stride = ranks)
  for (int ns = c->last_first; ns <= c->nsrc; ns += c->last_stride) mine.push_back(ns);
  const int limit = std::min(c->batch, BATCH_MAX);
  size_t need[2] = {0, 0};
  int bi = 0;
  for (size_t b0 = 0; b0 < mine.size(); b0 += (size_t)limit, bi++) {
    size_t sum = 0, spare = 0;
    for (size_t b = b0; b < std::min(mine.size(), b0 + (size_t)limit); b++) {
      const int cap = predicted_shells(c, mine[b]); // pass_list's predicted_cap
      sum += block_doubles(cap);
      // ... and room for ONE source of the batch to outgrow that (its block then moves to one twice as deep)
      if (cap < c->g.smax) spare = std::max(spare, block_doubles(std::min(c->g.smax, 2 * cap)));
    }
    need[bi & 1] = std::max(need[bi & 1], sum + spare);
  }
  for (int set = 0; set < 2; set++) {
    if (need[set] == 0) continue;
    std::vector<c2r_ctx::Segment> &sg = c->segs[set];
    size_t largest = 0;
    for (size_t i = 1; i < sg.size(); i++)
      if (sg[i].n > sg[largest].n) largest = i;
    if (!sg.empty() && sg[largest].n >= need[set]) {
      std::swap(sg[0], sg[largest]); // blocks are cut from the segments in order: the one that holds a whole batch first
      continue;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream2)); // nothing of an earlier pass may still read the segments given back
    HIPCHK(c, hipStreamSynchronize(c->stream3));
    arena_release(c, set);
    c->seg_cur[set] = c->seg_used[set] = 0;
    (void)arena_alloc(c, set, need[set], need[set]); // no room for all of it: the pass will cut its batches, as before
    c->seg_cur[set] = c->seg_used[set] = 0;
  }
  return 0;
}

static int begin_step_one(c2r_ctx *c) {
  if (!c) return 1;
  if (!c->have_state) return fail(c, "c2r_begin_step: c2r_upload_state has not been called");
  HIPCHK(c, hipSetDevice(c->device));
  if (arena_prepare(c)) return 1;
  c->packed_valid = c->transposed_valid = false; // xh_av, xhe_av are overwritten below
  HIPCHK(c, hipSetDevice(c->device));
  const size_t nc = c->g.ncell;
  HIPCHK(c, hipMemcpyAsync(c->d_xh_av, c->d_xh, sizeof(double) * 2 * nc, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_xh_int, c->d_xh, sizeof(double) * 2 * nc, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_xhe_av, c->d_xhe, sizeof(double) * 3 * nc, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_xhe_int, c->d_xhe, sizeof(double) * 3 * nc, hipMemcpyDeviceToDevice, c->stream));
  c->last_conv = -1;
  return 0;
}

static int end_step_one(c2r_ctx *c) {
  if (!c) return 1;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t nc = c->g.ncell;
  HIPCHK(c, hipMemcpyAsync(c->d_xh, c->d_xh_int, sizeof(double) * 2 * nc, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_xhe, c->d_xhe_int, sizeof(double) * 3 * nc, hipMemcpyDeviceToDevice, c->stream));
  if (!c->isothermal) // set_final_temperature_point (mat_ini_test.F90:505-515)
    HIPCHK(c, hipMemcpyAsync(c->d_temp + 2 * nc, c->d_temp, sizeof(float) * nc, hipMemcpyDeviceToDevice, c->stream));
  return 0;
}

static int set_rates_to_zero_one(c2r_ctx *c) {
  if (!c) return 1;
  HIPCHK(c, hipSetDevice(c->device));
  // Carried out lazily: when the pass that follows starts with a rates launch that covers every cell, that launch
  // writes the grids instead of adding to them and nothing has to be zeroed (flush_rates_zero otherwise).
  c->rates_zero_pending = true;
  std::memset(c->photon_loss, 0, sizeof c->photon_loss);
  c->sum_nbox = 0;
  return 0;
}

// the zeroing a c2r_set_rates_to_zero left pending, before anything reads or partly writes the rate grids
static int flush_rates_zero(c2r_ctx *c) {
  if (!c->rates_zero_pending) return 0;
  c->rates_zero_pending = false;
  c->phiheat_dirty = false;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, zero_device(c->d_rates, sizeof(double) * c->rates_count, c->stream));
  return 0;
}

static int set_batch_one(c2r_ctx *c, int nbatch) {
  if (!c) return 1;
  if (nbatch < 1 || nbatch > BATCH_MAX) return fail(c, "c2r_set_batch: %d not in [1,%d]", nbatch, BATCH_MAX);
  c->batch = nbatch;
  return 0;
}

// host-side sub-box bookkeeping of one source (do_source, evolve_source.F90:96-144, 233-236)
struct SrcRun {
  int known_before = 0;   // prev_nbox of the source when the batch was first put together
  int ns;                 // 1-based source number
  int nbox = 0;
  double total_flux = 0, loss = 0;
  bool active = true;
  bool final_loss_due = false; // the last round ended for geometric reasons: its full loss is evaluated after the sweep
  int cap = 0;            // shells the column block can hold
  int smax_prev = -1;     // largest shell already swept
};

// extent of the mesh as seen from a source, periodic_bc = .true. (evolve_source.F90:103-105)
struct Reach {
  int l[3], r[3];
};
static Reach mesh_reach(const Grid &g) {
  const int mesh[3] = {g.n1, g.n2, g.n3};
  Reach R;
  for (int d = 0; d < 3; d++) {
    R.r[d] = std::min(MAX_SUBBOX, mesh[d] / 2 - 1 + mesh[d] % 2);
    R.l[d] = -std::min(MAX_SUBBOX, mesh[d] / 2);
  }
  return R;
}
// the sub-box after `nbox` rounds (evolve_source.F90:141-144)
static Box round_box(const Reach &R, int nbox) {
  Box b;
  for (int d = 0; d < 3; d++) {
    b.hi[d] = std::min(SUBBOXSIZE * nbox, R.r[d]);
    b.lo[d] = std::max(-SUBBOXSIZE * nbox, R.l[d]);
  }
  return b;
}
static int box_smax(const Box &b) {
  int m = 0;
  for (int d = 0; d < 3; d++) m = std::max(m, std::max(b.hi[d], -b.lo[d]));
  return m;
}
static long long box_cells(const Box &b) {
  long long v = 1;
  for (int d = 0; d < 3; d++) v *= (b.hi[d] - b.lo[d] + 1);
  return v;
}
// the while-test of evolve_source.F90:136-139 can still pass after this box (z extent short of the mesh)
static bool box_can_grow(const Reach &R, const Box &b) { return b.hi[2] < R.r[2] && b.lo[2] > R.l[2]; }

static int pool_event(c2r_ctx *c, hipEvent_t *out) {
  if (c->ev_used == c->ev_pool.size()) {
    hipEvent_t e;
    HIPCHK(c, hipEventCreate(&e));
    c->ev_pool.push_back(e);
  }
  *out = c->ev_pool[c->ev_used++];
  return 0;
}


// Photon loss of the round's box for the sources h_list[set][list_off .. +n) (positions in the batch), full
// (sample = 1) or sampled, into c->h_loss[0..n).  Synchronises the sweep stream.
static int boundary_loss(c2r_ctx *c, int set, size_t list_off, int n, int s_lo, int s_hi, const Box &box, int sample,
                         const StepScalars &sc, const SedSet &ss, bool multi) {
  const int count = c->block_base[s_hi + 1] - c->block_base[s_lo];
  const int nblk = (count + sample - 1) / sample;
  const size_t need = (size_t)nblk * n;
  if (c->loss_partial_cap < need) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (ensure_pair<double>(c, &c->d_loss_partial, (double **)nullptr, &c->loss_partial_cap, need)) return 1;
  }
  // a surface cell has one coordinate on a face of the box, so its shell is at least the nearest face's distance:
  // the shells before that hold none, their blocks are not launched and their partial sums are set to zero
  int first_block = 0;
  if (sample == 1) {
    int s_first = 1 << 30;
    for (int d = 0; d < 3; d++) s_first = std::min(s_first, std::min(std::abs(box.lo[d]), std::abs(box.hi[d])));
    s_first = std::min(std::max(s_first, s_lo), s_hi);
    first_block = c->block_base[s_first] - c->block_base[s_lo];
    if (first_block > 0) HIPCHK(c, zero_device(c->d_loss_partial, sizeof(double) * need, c->stream));
  }
  hipLaunchKernelGGL(k_loss, dim3(nblk - first_block, n), dim3(BLOCK), 0, c->stream, c->g, c->d_src[set], c->d_list[set] + list_off,
                     multi ? 1 : 0, s_lo, s_hi, box, sc, c->d_bands, ss, c->d_block_base, c->d_loss_partial, nblk,
                     sample, first_block);
  hipLaunchKernelGGL(k_loss_finish, dim3(n), dim3(BLOCK), 0, c->stream, c->d_loss_partial, nblk, nblk, c->d_loss_acc);
  c->tm.sweep_launches += 2;
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(c->h_loss, c->d_loss_acc, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

// shells below this one read the i-faces' state from the mesh-ordered grids (see pass_list)
#ifndef C2R_TRANSPOSED_FROM_SHELL
#define C2R_TRANSPOSED_FROM_SHELL 32
#endif
constexpr int TRANSPOSED_FROM_SHELL = C2R_TRANSPOSED_FROM_SHELL;

constexpr int PROBE_SAMPLE = 16; // the probe looks at 8 cells of every 16th block of the round's shells

// One round of a batch's sub-box loop, as the probes and the while-test see it.
struct RoundRec {
  int round = 0, nact = 0, s_lo = 0, s_hi = 0;
  size_t off = 0;     // its active list in h_list / d_list
  size_t acc_off = 0; // its results in d_probe_acc / h_probe
  Box box{};
};

// Queue the probes of the rounds recs[from .. to) (lower bounds of their boundary losses, k_loss_probe_rounds) on the probe
// stream, behind everything queued on the sweep stream so far, and the copy of all their results to h_probe; ev_probe
// marks the end.  Does not wait.  One event on the sweep stream for all of them: the rounds of a batch whose
// sources are all expected to go on (prev_nbox) are probed together, late, and the sweep stream sees no probe
// kernel, no copy and no other event between its shells.  Whoever then changes what a probe reads -- the SrcDev
// entries, when a block moves -- waits for ev_probe first.
static int launch_probes(c2r_ctx *c, int set, const std::vector<RoundRec> &recs, size_t from, size_t to, const StepScalars &sc,
                         const SedSet &ss, bool multi) {
  if (from >= to) return 0;
  size_t need = 0;
  for (size_t i = from; i < to; i++) {
    const int count = c->block_base[recs[i].s_hi + 1] - c->block_base[recs[i].s_lo];
    need += (size_t)((count + PROBE_SAMPLE - 1) / PROBE_SAMPLE) * (size_t)recs[i].nact;
  }
  const size_t acc_need = recs[to - 1].acc_off + (size_t)recs[to - 1].nact;
  HIPCHK(c, hipEventSynchronize(c->ev_probe)); // the descriptors of an earlier launch have left the pinned buffer
  if (c->probe_acc_cap < acc_need) return fail(c, "internal: probe results of %zu rounds do not fit (%zu > %zu)", to, acc_need, c->probe_acc_cap);
  if (c->probe_partial_cap < need) {
    HIPCHK(c, hipStreamSynchronize(c->stream_probe)); // earlier probes may still use the old buffer
    if (ensure_pair<double>(c, &c->d_probe_partial, (double **)nullptr, &c->probe_partial_cap, need)) return 1;
  }
#ifdef C2R_PROBE_INLINE
  hipStream_t st = c->stream;
#else
  hipStream_t st = c->stream_probe;
  HIPCHK(c, hipEventRecord(c->ev_round, c->stream));
  HIPCHK(c, hipStreamWaitEvent(st, c->ev_round, 0));
#endif
  // descriptors of the rounds, then one probe launch and one finishing launch for all of them
  const size_t nr = to - from;
  if (c->probe_rounds_cap < nr) {
    HIPCHK(c, hipStreamSynchronize(c->stream_probe));
    if (c->d_probe_rounds) HIPCHK(c, hipFree(c->d_probe_rounds));
    if (c->h_probe_rounds) HIPCHK(c, hipHostFree(c->h_probe_rounds));
    c->d_probe_rounds = c->h_probe_rounds = nullptr;
    const size_t want = nr + 64;
    HIPCHK(c, hipMalloc(&c->d_probe_rounds, sizeof(ProbeRound) * want));
    HIPCHK(c, hipHostMalloc(&c->h_probe_rounds, sizeof(ProbeRound) * want));
    c->probe_rounds_cap = want;
  }
  ProbeRound *hr = static_cast<ProbeRound *>(c->h_probe_rounds);
  size_t used = 0;
  int max_nblk = 0, max_nact = 0;
  for (size_t i = from; i < to; i++) {
    const RoundRec &R = recs[i];
    const int count = c->block_base[R.s_hi + 1] - c->block_base[R.s_lo];
    ProbeRound &P = hr[i - from];
    P.s_lo = R.s_lo; P.s_hi = R.s_hi; P.box = R.box;
    P.list_off = (int)R.off; P.nact = R.nact;
    P.nblk = (count + PROBE_SAMPLE - 1) / PROBE_SAMPLE;
    P.partial_off = (int)used; P.acc_off = (int)R.acc_off;
    used += (size_t)P.nblk * (size_t)R.nact;
    max_nblk = std::max(max_nblk, P.nblk);
    max_nact = std::max(max_nact, P.nact);
  }
  HIPCHK(c, hipMemcpyAsync(c->d_probe_rounds, hr, sizeof(ProbeRound) * nr, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_loss_probe_rounds, dim3(max_nblk, max_nact, (unsigned)nr), dim3(BLOCK), 0, st, c->g, c->d_src[set], c->d_list[set],
                     static_cast<const ProbeRound *>(c->d_probe_rounds), multi ? 1 : 0, sc, c->d_bands, ss, c->d_block_base,
                     c->d_probe_partial, PROBE_SAMPLE);
  hipLaunchKernelGGL(k_loss_finish_rounds, dim3(max_nact, (unsigned)nr), dim3(BLOCK), 0, st,
                     static_cast<const ProbeRound *>(c->d_probe_rounds), c->d_probe_partial, c->d_probe_acc);
  c->tm.sweep_launches += 2;
  HIPCHK(c, hipGetLastError());
  const size_t a0 = recs[from].acc_off;
  HIPCHK(c, hipMemcpyAsync(c->h_probe + a0, c->d_probe_acc + a0, sizeof(double) * (acc_need - a0), hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipEventRecord(c->ev_probe, st));
  return 0;
}

// do_source (evolve_source.F90:66-238) for every source number in `mine`, up to c->batch at a time.
// Per batch: the column sweep (dependent shell launches, host test of the sub-box loop) on the
// high-priority stream, then ONE rates launch for the whole batch on the second stream; the next
// batch's sweep overlaps it (scratch sets ping-pong).  Rates launches are ordered on their stream,
// so the accumulation over sources keeps the reference's order.
static int pass_finish(c2r_ctx *c);
static int flush_rates_zero(c2r_ctx *c);

// the losses of set `set`'s batches that were still on their way from the device (the caller has waited for the
// set's last rates launch and what follows it)
static void resolve_tails(c2r_ctx *c, int set) {
  for (c2r_ctx::BatchTail &bt : c->tails) {
    if (bt.resolved || bt.set != set) continue;
    for (size_t b = 0; b < bt.loss.size(); b++)
      if (bt.slot[b] >= 0) bt.loss[b] = c->h_final[set][bt.slot[b]];
    bt.resolved = true;
  }
}

// nslab > 0: the caller wants to consume the rate grids slab by slab (z ranges) while later slabs are
// still being computed: the rates launch of the LAST batch is cut into nslab launches, an event is
// recorded after each, and the final synchronisation is left to pass_finish.
static int pass_list(c2r_ctx *c, const std::vector<int> &mine, int nslab = 0) {
  HIPCHK(c, hipSetDevice(c->device));
  if (c->pass_open) return fail(c, "previous c2r_pass_sources_begin was not closed by c2r_pass_sources_end");
  struct InPass {
    c2r_ctx *c;
    explicit InPass(c2r_ctx *c_) : c(c_) { c->in_pass = true; }
    ~InPass() { c->in_pass = false; }
  } in_pass_guard(c);
  const Grid g = c->g;
  const size_t nc = g.ncell;
  const StepScalars sc = scalars(c);
  bool multi = false;
  const SedSet ss = sedset(c, &multi);
  const int mesh[3] = {g.n1, g.n2, g.n3};
  const Reach reach = mesh_reach(g);
  c->tm.sweep_ms = c->tm.rates_ms = 0.0;
  c->tm.sweep_launches = c->tm.rates_launches = 0;
  c->tm.cells_swept = 0;
  c->ev_used = 0;
  // every pass is closed by pass_finish, which adds the batches' kept losses into photon_loss / sum_nbox and empties
  // this list; entries still here belong to a pass that ended in an error and must not be added to this one
  c->tails.clear();
  if (c->prev_nbox.size() != (size_t)c->nsrc) c->prev_nbox.assign((size_t)c->nsrc, 0);
  std::vector<hipEvent_t> tev; // per batch: sweep start, sweep end, rates start, rates end
  // a rank without sources writes nothing: a pending zeroing of the rate grids has to happen for real
  if (mine.empty() && flush_rates_zero(c)) return 1;
  // everything queued earlier on the main stream (state upload, zeroing of the rates) must be
  // visible to the rates stream
  HIPCHK(c, hipEventRecord(c->ev_sweep_done[0], c->stream));
  HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_sweep_done[0], 0));

  {
    // the products neufrac * ndens the sweep reads (k_pack_state; xh_av, xhe_av change every iteration), in mesh
    // order and in (j,i,k) order: made on the third stream while the inner shells, which do without them, are
    // already on their way
    HIPCHK(c, hipStreamWaitEvent(c->stream3, c->ev_sweep_done[0], 0));
    if (c->packed_valid && c->transposed_valid) {
      // nothing to make: the global pass transposed its own products (ev_transposed is on record since then)
    } else if (c->packed_valid)
      hipLaunchKernelGGL(k_transpose_packed, dim3((g.n1 + 31) / 32, (g.n2 + 31) / 32, 3 * g.n3), dim3(BLOCK), 0, c->stream3, g,
                         c->d_stateT, c->d_stateT + 3 * nc);
    else
      hipLaunchKernelGGL(k_pack_state, dim3((g.n1 + 31) / 32, (g.n2 + 31) / 32, g.n3), dim3(BLOCK), 0, c->stream3, g, c->d_ndens, c->d_xh_av,
                         c->d_xhe_av, c->d_stateT, c->d_stateT + 3 * nc);
    HIPCHK(c, hipGetLastError());
    if (!(c->packed_valid && c->transposed_valid)) HIPCHK(c, hipEventRecord(c->ev_transposed, c->stream3));
  }
  bool transposed_seen = false; // the sweep stream has waited for ev_transposed
  const bool packed_early = c->packed_valid; // the mesh-ordered products need not wait for anything
  // slabs: tile layers (4 planes each) [slab_layer[s], slab_layer[s+1])
  const int nt3_all = (g.n3 + 3) / 4;
  const int ns_eff = nslab > 0 ? std::min(nslab, nt3_all) : 0;
  c->slab_k.assign(1, 0);
  for (int sidx = 1; sidx <= ns_eff; sidx++) c->slab_k.push_back(std::min(g.n3, 4 * (int)((long long)nt3_all * sidx / ns_eff)));
  while (c->ev_slab.size() < (size_t)ns_eff) {
    hipEvent_t e;
    HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    c->ev_slab.push_back(e);
  }
  // column blocks: what a source needed in the last pass plus one round, or four rounds when nothing is known;
  // never less than one round, never more than the mesh
  if (c->prev_grow.size() != (size_t)c->nsrc) c->prev_grow.assign((size_t)c->nsrc, 0);
  auto predicted_cap = [&](int ns) { return predicted_shells(c, ns); };
  int bi = 0;
  int batch_limit = std::min(c->batch, BATCH_MAX);
  for (size_t b0 = 0; b0 < mine.size(); bi++) {
    int nb = (int)std::min<size_t>((size_t)batch_limit, mine.size() - b0);
    const int set = bi & 1;
    // this set's scratch may still be read by the rates launch of two batches ago, and its pinned lists may
    // still feed copies queued then
    if (c->set_busy[set]) {
      HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_rates_done[set], 0));
      HIPCHK(c, hipEventSynchronize(c->ev_rates_done[set]));
      resolve_tails(c, set); // h_final[set] is about to be written again
    }
    std::vector<SrcRun> run;
    std::vector<int> known0((size_t)nb); // what the last pass knew about the batch's sources (a batch that starts over raises prev_nbox)
    for (int b = 0; b < nb; b++) known0[(size_t)b] = c->prev_nbox[(size_t)mine[b0 + b] - 1];
    int restarts = 0;
  restart_batch:
    run.assign((size_t)nb, SrcRun());
    for (int b = 0; b < nb; b++) run[(size_t)b].known_before = known0[(size_t)b];
    {
      // Blocks of this batch.  Segments are kept from batch to batch and pass to pass (allocation is slow), so what
      // the set holds may be cut for other block sizes than this batch needs (many small segments of a pass with
      // small boxes).  When the blocks do not fit although less than half of what the set holds is in use, its
      // segments are given back and made anew in one piece; otherwise the batch is what did fit.
      for (bool released = false;;) {
        c->seg_cur[set] = 0;
        c->seg_used[set] = 0;
        c->vacated[set].clear();
        SrcDev *hs = c->h_src[set];
        size_t total = 0;
        for (int b = 0; b < nb; b++) total += block_doubles(predicted_cap(mine[b0 + b]));
        int placed = 0;
        size_t placed_doubles = 0;
        for (; placed < nb; placed++) {
          SrcRun &r = run[placed];
          r.ns = mine[b0 + placed];
          r.cap = predicted_cap(r.ns);
          const size_t w = (size_t)(2 * r.cap + 1);
          hs[placed].cz = w * w * w;
          hs[placed].cols = arena_alloc(c, set, 6 * hs[placed].cz, total);
          if (!hs[placed].cols) break;
          total -= 6 * hs[placed].cz;
          placed_doubles += 6 * hs[placed].cz;
        }
        if (placed == nb) break;
        size_t have = 0;
        for (const c2r_ctx::Segment &sg : c->segs[set]) have += sg.n;
        if (!released && have > 0 && placed_doubles < have / 2) {
          released = true;
          arena_release(c, set);
          continue;
        }
        if (placed == 0) return fail(c, "column scratch: one source of this mesh does not fit in device memory");
        // A batch cut to exactly what fits has no room for a source that outgrows its block (the deeper block is needed
        // while the old one is still in use) and starts over at the first such move -- at 512^3, where a block at the mesh
        // limit is 6.4 GB and a set holds some twenty, 615 times in six passes (round 5).  So the cut leaves room for a third
        // of the sources that can still grow to move once; sources at the mesh limit need none.
        int keep = placed;
        for (;;) {
          size_t used = 0, reserve = 0;
          for (int b = 0; b < keep; b++) {
            used += 6 * hs[b].cz;
            if (run[b].cap < g.smax) reserve += block_doubles(std::min(g.smax, 2 * run[b].cap));
          }
          if (keep <= 1 || used + reserve / 3 <= placed_doubles) break;
          keep--;
        }
        if (getenv("C2R_ARENA_LOG")) fprintf(stderr, "c2ray_hip: column scratch, set %d: batch of %d cut to %d sources (%d fit)\n", set, nb, keep, placed);
        nb = keep; // the same calls place the same blocks again
        run.resize((size_t)nb);
      }
      SrcDev *hs = c->h_src[set];
      for (int b = 0; b < nb; b++) {
        SrcRun &r = run[b];
        r.total_flux = c->normflux[r.ns - 1] * c->s_star; // evolve_source.F90:122-128
        for (int k = 0; k < 2; k++)
          if (multi && !c->normflux_sed[k].empty())
            r.total_flux = r.total_flux + c->normflux_sed[k][r.ns - 1] * c->s_star_sed[k];
        r.loss = r.total_flux;
        SrcDev &S = hs[b];
        const int *p = &c->srcpos[3 * (size_t)(r.ns - 1)];
        S.i0 = p[0]; S.j0 = p[1]; S.k0 = p[2];
        for (int d = 0; d < 3; d++) { S.lo[d] = 0; S.hi[d] = 0; }
        S.nflux = c->normflux[r.ns - 1];
        for (int k = 0; k < 2; k++) S.nflux_sed[k] = c->normflux_sed[k].empty() ? 0.0 : c->normflux_sed[k][r.ns - 1];
      }
      HIPCHK(c, hipMemcpyAsync(c->d_src[set], hs, sizeof(SrcDev) * nb, hipMemcpyHostToDevice, c->stream));
    }
    // coldensh_out = 0 for every new source (evolve_source.F90:94-95) serves two purposes in the
    // reference: the "already done" marker (replaced here by the shell order: every cell is visited
    // once) and finite values for corners whose interpolation weight is exactly 0.  The arena is
    // zeroed once at allocation and only ever holds finite columns afterwards, so 0*w stays 0.
    hipEvent_t e_s0 = nullptr, e_s1 = nullptr, e_r0 = nullptr, e_r1 = nullptr;
    if (c->timing) {
      if (pool_event(c, &e_s0) || pool_event(c, &e_s1) || pool_event(c, &e_r0) || pool_event(c, &e_r1)) return 1;
    }
    bool sweep_started = false; // e_s0 goes in front of the first shell launch, behind the uploads of records and lists
    size_t list_used = 0;
    long long batch_cells = 0;
    int *hl = c->h_list[set];
    // The sub-box loop (evolve_source.F90:136-144).  After the shells of round r the loss through the box surface
    // decides who sweeps round r+1; a quick lower bound of it (k_loss_probe_rounds) nearly always does.  When every active
    // source went beyond round r in the previous pass (prev_nbox), round r+1 is launched at once for all of them, on
    // trust, and the probe of round r is not even queued yet: the probes of all such rounds go out together, on
    // their own stream, when a decision is really needed -- before a round that some source did not reach last time,
    // or before the round that is everybody's last for geometric reasons -- so that the sweep stream carries shells
    // and nothing else (a probe per round cost it an event, and 13-24 us of gap at each of the 12 round boundaries
    // of a 256^3 sweep).  A source that turns out to have stopped at r keeps round r as its last -- its shells
    // of later rounds are never looked at (the rates launch only sees the final box), so a wrong guess costs time,
    // never a bit.
    std::vector<RoundRec> recs;       // the rounds swept so far
    size_t settled = 0, launched = 0; // recs[0, settled): while-test applied; recs[settled, launched): probes in flight
    {
      const size_t acc_need = (size_t)nb * (size_t)(g.smax / SUBBOXSIZE + 3);
      if (c->probe_acc_cap < acc_need) {
        HIPCHK(c, hipStreamSynchronize(c->stream_probe));
        if (ensure_pair<double>(c, &c->d_probe_acc, &c->h_probe, &c->probe_acc_cap, acc_need)) return 1;
      }
    }
    auto launch = [&]() -> int {
      if (launch_probes(c, set, recs, launched, recs.size(), sc, ss, multi)) return 1;
      launched = recs.size();
      return 0;
    };
    // read the probes of all rounds not yet decided, replace what a probe leaves undecided by the full sum, apply
    // the while-test round by round: sources that stop get active = false and that round as their last
    auto settle = [&]() -> int {
      if (settled == recs.size()) return 0;
      if (launch()) return 1;
      HIPCHK(c, hipEventSynchronize(c->ev_probe));
      for (; settled < recs.size(); settled++) {
        const RoundRec &P = recs[settled];
        const double *probe = c->h_probe + P.acc_off;
        const int *lst = hl + P.off;
        std::vector<int> undecided;
        for (int a = 0; a < P.nact; a++) {
          SrcRun &r = run[lst[a]];
          if (!r.active) continue; // stopped at an earlier round: swept this one on trust, to no effect
          r.loss = probe[a];
          if (!(r.loss > 2.0 * (C2R_F(1e-10) * r.total_flux))) undecided.push_back(lst[a]);
        }
        if (!undecided.empty()) {
          int *ul = hl + list_used;
          std::copy(undecided.begin(), undecided.end(), ul);
          HIPCHK(c, hipMemcpyAsync(c->d_list[set] + list_used, ul, sizeof(int) * undecided.size(), hipMemcpyHostToDevice, c->stream));
          if (boundary_loss(c, set, list_used, (int)undecided.size(), P.s_lo, P.s_hi, P.box, 1, sc, ss, multi)) return 1;
          list_used += undecided.size();
          for (size_t j = 0; j < undecided.size(); j++) run[undecided[j]].loss = c->h_loss[j];
        }
        for (int a = 0; a < P.nact; a++) {
          SrcRun &r = run[lst[a]];
          if (!r.active) continue;
          if (!(r.loss > C2R_F(1e-10) * r.total_flux)) { // evolve_source.F90:136: the box does not grow any more
            r.active = false;
            r.nbox = P.round;
          }
        }
      }
      return 0;
    };
    // every source still active reached at least round `round` in the previous pass
    auto all_reached = [&](int round) {
      bool yes = true;
      for (int b = 0; b < nb && yes; b++)
        if (run[b].active) yes = c->prev_nbox[(size_t)run[b].ns - 1] >= round;
      return yes;
    };
    size_t cur_off = 0; // the active list of the last round launched
    int cur_nact = 0;
    size_t acc_used = 0;
    for (int round = 1;; round++) {
      const Box box = round_box(reach, round);
      const int s_hi = box_smax(box);
      // may this round start before the losses of the rounds before it are known?
      const bool ahead = round > 1 && cur_nact > 0 && all_reached(round);
      if (!ahead && settle()) return 1;
      // who sweeps this round: the while-test of evolve_source.F90:136-139 (its loss part taken on trust when ahead)
      int nact = 0;
      int *act = hl + list_used;
      int s_lo = 1 << 30;
      for (int b = 0; b < nb; b++) {
        SrcRun &r = run[b];
        if (!r.active) continue;
        if (!box_can_grow(reach, round_box(reach, r.nbox)) || !(ahead || r.loss > C2R_F(1e-10) * r.total_flux)) {
          r.active = false;
          continue;
        }
        r.nbox = round;
        act[nact++] = b;
        s_lo = std::min(s_lo, r.smax_prev + 1);
      }
      if (nact == 0) {
        if (settle()) return 1;
        break;
      }
      if (s_hi > g.smax) return fail(c, "internal: shell %d beyond smax %d", s_hi, g.smax);
      // blocks too small for this round move to larger ones (the shells stored so far are a prefix of every array)
      for (int a = 0; a < nact; a++) {
        SrcRun &r = run[act[a]];
        if (r.cap >= s_hi) continue;
        const int ncap = std::min(g.smax, std::max(s_hi, 2 * r.cap));
        const size_t nd = block_doubles(ncap);
        double *ncols = arena_alloc(c, set, nd);
        if (!ncols) {
          // No room left on the device in the middle of a sweep.  Nothing of this batch has reached the rate
          // grids yet, so the batch starts over with what this attempt has taught (sources that have stopped get
          // the block they needed, those still growing two rounds more than they have come to) and, since even
          // that did not fit, with fewer sources.
          HIPCHK(c, hipStreamSynchronize(c->stream));
          HIPCHK(c, hipStreamSynchronize(c->stream_probe));
          // (round 5: a source still growing when the room ran out is given the mesh limit, not two rounds more -- at 512^3
          // the fronts of a whole batch race there together, and "two more" made every batch start over twice)
          const int limit_rounds = (g.smax + SUBBOXSIZE - 1) / SUBBOXSIZE;
          for (int b = 0; b < nb; b++) {
            int &pn = c->prev_nbox[(size_t)run[b].ns - 1];
            pn = std::max(pn, run[b].active ? limit_rounds : run[b].nbox);
          }
          if (nb == 1) return fail(c, "column scratch: one source of this mesh does not fit in device memory");
          if (getenv("C2R_ARENA_LOG")) fprintf(stderr, "c2ray_hip: column scratch, set %d: no room to grow in round %d, batch of %d starts over\n", set, round, nb);
          // the first time with the same sources: blocks made for what is known now need no moves, and what a move leaves
          // behind is most of what filled the set; a batch that runs out of room again is halved
          if (restarts++ > 0) nb = (nb + 1) / 2;
          c->arena_stats[4]++;
          goto restart_batch;
        }
        // probes in flight read this source's SrcDev entry: they must be through before the entry changes
        if (launched > settled) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_probe, 0));
        SrcDev &S = c->h_src[set][act[a]];
        const size_t wn = (size_t)(2 * ncap + 1), ncz = wn * wn * wn;
        const size_t wp = (size_t)(2 * r.smax_prev + 1), have = r.smax_prev >= 0 ? wp * wp * wp : 0;
        // the shells stored so far are a prefix of each array (or of each array of triples)
        for (int half = 0; half < 2 && have > 0; half++) {
          const bool triples = C2R_COLS_AOS == 1 || C2R_COLS_AOS == (half ? 3 : 2);
          for (int k = 0; k < (triples ? 1 : 3); k++) {
            const size_t at = (size_t)(3 * half + k);
            HIPCHK(c, hipMemcpyAsync(ncols + at * ncz, S.cols + at * S.cz, sizeof(double) * (triples ? 3 : 1) * have,
                                     hipMemcpyDeviceToDevice, c->stream));
          }
        }
        if (S.cz > 0) c->vacated[set].push_back(c2r_ctx::Segment{S.cols, 6 * S.cz}); // free for the batch's later moves
        S.cols = ncols;
        S.cz = ncz;
        r.cap = ncap;
        c->arena_stats[3]++;
        HIPCHK(c, hipMemcpyAsync(c->d_src[set] + act[a], &S, sizeof(SrcDev), hipMemcpyHostToDevice, c->stream));
      }
      // the same sources as in the last round: the list is on the device already
      size_t act_off = list_used;
      if (cur_nact == nact && round > 1 && std::equal(act, act + nact, hl + cur_off)) {
        act_off = cur_off;
      } else {
        HIPCHK(c, hipMemcpyAsync(c->d_list[set] + list_used, act, sizeof(int) * nact, hipMemcpyHostToDevice, c->stream));
        list_used += (size_t)nact;
      }
      cur_off = act_off;
      cur_nact = nact;
      // C2R_SWEEP_GENERIC=1 (diagnostic): every shell through the general per-cell code, one launch per shell
      static const bool generic_sweep = getenv("C2R_SWEEP_GENERIC") && atoi(getenv("C2R_SWEEP_GENERIC")) > 0;
      SweepArgs SA;
      SA.g = g; SA.box = box; SA.sc = sc;
      SA.ndens = c->d_ndens; SA.xh_av = c->d_xh_av; SA.xhe_av = c->d_xhe_av;
      SA.lls_grid = c->lls_on_grid ? c->d_lls : nullptr;
      if (c->timing && !sweep_started) {
        HIPCHK(c, hipEventRecord(e_s0, c->stream));
        sweep_started = true;
      }
      for (int s = s_lo; s <= s_hi; s++) {
        // (a large batch waits at once: its many small faces would pay more for strided reads than the wait costs)
        if ((s >= TRANSPOSED_FROM_SHELL || nb > 16) && !transposed_seen) {
          HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_transposed, 0));
          transposed_seen = true;
        }
        SA.packed = SA.packedT = nullptr;
        if (transposed_seen || packed_early) SA.packed = c->d_stateT;
        if (transposed_seen) SA.packedT = c->d_stateT + 3 * nc;
        const int nblk = c->block_base[s + 1] - c->block_base[s];
        // from 64 blocks on: a multiple of 8 blocks, one contiguous eighth of the shell per XCD (see the kernel)
        const int nlaunch = nblk >= 64 ? ((nblk + 7) & ~7) : nblk;
        if (s >= 2 && s <= SHELL_FAST_MAX && !generic_sweep)
          hipLaunchKernelGGL(k_sweep_shell_fast, dim3(nlaunch, nact), dim3(BLOCK), 0, c->stream, SA, c->d_src[set],
                             c->d_list[set] + act_off, c->shell_geom[(size_t)s]);
        else
          hipLaunchKernelGGL(k_sweep_shell, dim3(nlaunch, nact), dim3(BLOCK), 0, c->stream, SA, c->d_src[set], c->d_list[set] + act_off, s);
        c->tm.sweep_launches++;
      }
      HIPCHK(c, hipGetLastError());
      for (int a = 0; a < nact; a++) run[hl[act_off + a]].smax_prev = s_hi;
      if (!box_can_grow(reach, box)) {
        // The while-test after this round fails whatever the loss: the round is every active source's last, and
        // its loss (the one that is kept, evolve_source.F90:233) comes out of the rates launch (SrcDev::loss_lo).
        // The rounds before it are decided now; their probes were queued before this round's shells.
        if (settle()) return 1;
        for (int a = 0; a < nact; a++) {
          SrcRun &r = run[hl[act_off + a]];
          if (!r.active) continue; // stopped a round earlier after all
          r.final_loss_due = true;
          r.active = false;
        }
        break;
      }
      // The loss of this round decides whether a source goes on.
      RoundRec R;
      R.round = round; R.nact = nact; R.s_lo = s_lo; R.s_hi = s_hi; R.off = act_off; R.acc_off = acc_used; R.box = box;
      acc_used += (size_t)nact;
      recs.push_back(R);
      // Its probe goes out now if the next round needs the answer before it can start (somebody may stop here) or if
      // the next round is the last for geometric reasons (the probes then run beside its shells, and the answers
      // are there when the rates launch has to be put together); otherwise it waits for company.
      const bool next_is_last = !box_can_grow(reach, round_box(reach, round + 1));
      // C2R_PROBE_EACH_ROUND=1 (diagnostic): queue every round's probe right behind its shells
      static const bool each_round = getenv("C2R_PROBE_EACH_ROUND") && atoi(getenv("C2R_PROBE_EACH_ROUND")) > 0;
      if ((each_round || !all_reached(round + 1) || next_is_last) && launch()) return 1;
    }
    for (int b = 0; b < nb; b++) batch_cells += run[b].nbox > 0 ? box_cells(round_box(reach, run[b].nbox)) : 0;
    // final sub-boxes for the rates launch
    bool any_final = false;
    // C2R_FINAL_LOSS_KERNEL=1 (diagnostic): evaluate the kept losses of final rounds with k_loss, beside the rates
    // launch, as rounds 1 and 2 of this library did, instead of taking them from the rates launch
    static const bool legacy_env = getenv("C2R_FINAL_LOSS_KERNEL") && atoi(getenv("C2R_FINAL_LOSS_KERNEL")) > 0;
    const bool legacy_final_loss = legacy_env || !c->isothermal; // heating kernels do not keep photo_out (see k_rates)
    for (int b = 0; b < nb; b++) {
      SrcDev &S = c->h_src[set][b];
      const Box fb = round_box(reach, run[b].nbox);
      for (int d = 0; d < 3; d++) { S.lo[d] = fb.lo[d]; S.hi[d] = fb.hi[d]; }
      // a source whose while-test failed before the first sub-box (a mesh only two cells deep) traced nothing:
      // an empty box, so that no cell passes the in-box test
      if (run[b].nbox == 0) { S.lo[0] = 1; S.hi[0] = 0; }
      // the kept loss of a round that was the last for geometric reasons comes out of the rates launch: the
      // surface cells in the shells of that round
      S.loss_lo = -1;
      if (run[b].final_loss_due && !legacy_final_loss) {
        any_final = true;
        S.loss_lo = run[b].nbox > 1 ? box_smax(round_box(reach, run[b].nbox - 1)) + 1 : 0;
      }
      {
        int &pn = c->prev_nbox[(size_t)run[b].ns - 1];
        // (first_try_nbox: what was known before this batch -- a batch that started over has raised prev_nbox meanwhile)
        c->prev_grow[(size_t)run[b].ns - 1] = run[b].known_before > 0 ? std::max(0, run[b].nbox - run[b].known_before) : 0;
        pn = run[b].nbox;
      }
    }
    if (c->timing) { // the sweep's span: from the first shell launch to the last, record uploads on either side left out
      if (!sweep_started) HIPCHK(c, hipEventRecord(e_s0, c->stream));
      HIPCHK(c, hipEventRecord(e_s1, c->stream));
    }
    HIPCHK(c, hipMemcpyAsync(c->d_src[set], c->h_src[set], sizeof(SrcDev) * nb, hipMemcpyHostToDevice, c->stream));
    if (!transposed_seen) {
      // small boxes only: nobody needed the copies, but whatever follows this sweep (the global pass rewrites the
      // state) has to come after the kernel that reads it
      HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_transposed, 0));
      transposed_seen = true;
    }
    HIPCHK(c, hipEventRecord(c->ev_sweep_done[set], c->stream));

    // rates of the whole batch, in source order, on the second stream
    HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_sweep_done[set], 0));
    if (c->timing) HIPCHK(c, hipEventRecord(e_r0, c->stream2));
    // tiles (8 x 8 x 4 cells) that intersect the final sub-box of some source of the batch, and per tile the
    // sources that do
    const int nt1 = (g.n1 + 7) / 8, nt2 = (g.n2 + 7) / 8, nt3 = (g.n3 + 3) / 4;
    int nblk = nt1 * nt2 * nt3;
    const int *d_tiles = nullptr, *d_tptr = nullptr, *d_tsrc = nullptr;
    {
      bool full = false;
      for (int b = 0; b < nb && !full; b++) {
        const Box fb = round_box(reach, run[b].nbox);
        full = run[b].nbox > 0;
        for (int d = 0; d < 3; d++) full = full && (fb.hi[d] - fb.lo[d] + 1 >= mesh[d]);
      }
      // few sources one of which fills the mesh: every cell simply walks all of them
      if (!(full && nb <= 16)) {
        std::vector<int> &cnt = c->tile_count;
        cnt.assign((size_t)nblk, 0);
        std::vector<int> cov[3];
        const int tsz[3] = {8, 8, 4}, ntd[3] = {nt1, nt2, nt3};
        auto covered = [&](int b) {
          const Box fb = round_box(reach, run[b].nbox);
          const int *p = &c->srcpos[3 * (size_t)(run[b].ns - 1)];
          for (int d = 0; d < 3; d++) {
            cov[d].clear();
            int last = -1;
            std::vector<unsigned char> seen((size_t)ntd[d], 0);
            for (int o = fb.lo[d]; o <= fb.hi[d]; o++) {
              int x = (p[d] - 1 + o) % mesh[d];
              if (x < 0) x += mesh[d];
              const int t = x / tsz[d];
              if (t != last && !seen[t]) { seen[t] = 1; cov[d].push_back(t); }
              last = t;
            }
          }
        };
        size_t total = 0;
        for (int b = 0; b < nb; b++) {
          if (run[b].nbox == 0) continue;
          covered(b);
          for (int tk : cov[2])
            for (int tj_ : cov[1]) {
              int *row = &cnt[((size_t)tk * nt2 + tj_) * nt1];
              for (int ti_ : cov[0]) row[ti_]++;
            }
          total += cov[0].size() * cov[1].size() * cov[2].size();
        }
        if (ensure_pair<int>(c, &c->d_tsrc[set], &c->h_tsrc[set], &c->tsrc_cap[set], total + 1)) return 1;
        int *list = c->h_tiles[set], *tp = c->h_tptr[set], *tsrc = c->h_tsrc[set];
        int ntl = 0;
        size_t acc = 0;
        for (int t = 0; t < nblk; t++) {
          if (!cnt[t]) { cnt[t] = -1; continue; }
          list[ntl] = t;
          tp[ntl] = (int)acc;
          acc += (size_t)cnt[t];
          cnt[t] = ntl++; // from count to position in the list
        }
        if (ntl == 0) { list[0] = 0; tp[0] = 0; ntl = 1; }
        tp[ntl] = (int)acc;
        std::vector<int> fill(tp, tp + ntl);
        for (int b = 0; b < nb; b++) { // ascending b: every tile's sources end up in source order
          if (run[b].nbox == 0) continue;
          covered(b);
          for (int tk : cov[2])
            for (int tj_ : cov[1]) {
              const int *row = &cnt[((size_t)tk * nt2 + tj_) * nt1];
              for (int ti_ : cov[0]) tsrc[fill[row[ti_]]++] = b;
            }
        }
        HIPCHK(c, hipMemcpyAsync(c->d_tiles[set], list, sizeof(int) * (size_t)ntl, hipMemcpyHostToDevice, c->stream2));
        HIPCHK(c, hipMemcpyAsync(c->d_tptr[set], tp, sizeof(int) * (size_t)(ntl + 1), hipMemcpyHostToDevice, c->stream2));
        if (acc > 0) HIPCHK(c, hipMemcpyAsync(c->d_tsrc[set], tsrc, sizeof(int) * acc, hipMemcpyHostToDevice, c->stream2));
        d_tiles = c->d_tiles[set];
        d_tptr = c->d_tptr[set];
        d_tsrc = c->d_tsrc[set];
        nblk = ntl;
      }
    }
    // a pending set_rates_to_zero: this launch writes the grids if it covers every cell, else they are zeroed now
    // (on the rates stream, which this batch's launches follow)
    int fresh = 0;
    if (c->rates_zero_pending) {
      if (!d_tiles) {
        fresh = 1;
        c->rates_zero_pending = false;
        if (c->isothermal && c->phiheat_dirty) { // phiheat of an earlier heating step: not written by this launch
          HIPCHK(c, zero_device(c->d_rates + 3 * nc, sizeof(double) * nc, c->stream2));
          c->phiheat_dirty = false;
        }
      } else {
        c->rates_zero_pending = false;
        c->phiheat_dirty = false;
        // (the four grids only: the tail of the buffer is written at the end of this pass, on the other stream)
        HIPCHK(c, zero_device(c->d_rates, sizeof(double) * 4 * nc, c->stream2));
      }
    }
    if (!c->isothermal) c->phiheat_dirty = true;
#define C2R_LAUNCH_RATES(H, M)                                                                               \
  hipLaunchKernelGGL((k_rates<H, M>), dim3(cnt_), dim3(BLOCK), 0, st_, g, c->d_src[set], nb, sc, c->d_ndens, c->d_xh_av, \
                     c->d_xhe_av, c->d_bands, ss, c->d_rates, d_tiles, d_tptr, d_tsrc, base_, fresh)
    const bool last_batch = b0 + nb >= mine.size();
    const int pieces = (last_batch && ns_eff > 0) ? ns_eff : 1;
    const int per_layer = nt1 * nt2;
    // slabs alternate between two streams so that the thin tail of one launch overlaps the start of the
    // next; the third stream is forked from (and joined back into) the rates stream, which carries the
    // dependencies on the sweep, on earlier batches and on the tile-list copy
    if (pieces > 1) {
      HIPCHK(c, hipEventRecord(c->ev_fork, c->stream2));
      HIPCHK(c, hipStreamWaitEvent(c->stream3, c->ev_fork, 0));
    }
    for (int piece = 0; piece < pieces; piece++) {
      hipStream_t st_ = (pieces > 1 && (piece & 1)) ? c->stream3 : c->stream2;
      // tiles of the layers [l0, l1): a contiguous range of tile ids, hence of the (sorted) list too
      const int l0 = pieces == 1 ? 0 : c->slab_k[piece] / 4;
      const int l1 = pieces == 1 ? nt3 : (c->slab_k[piece + 1] + 3) / 4;
      int base_, cnt_;
      if (d_tiles) {
        const int *list = c->h_tiles[set];
        base_ = (int)(std::lower_bound(list, list + nblk, l0 * per_layer) - list);
        cnt_ = (int)(std::lower_bound(list, list + nblk, l1 * per_layer) - list) - base_;
      } else {
        base_ = l0 * per_layer;
        cnt_ = (l1 - l0) * per_layer;
      }
      if (cnt_ > 0) {
        if (c->isothermal) {
          if (multi) C2R_LAUNCH_RATES(false, true); else C2R_LAUNCH_RATES(false, false);
        } else {
          if (multi) C2R_LAUNCH_RATES(true, true); else C2R_LAUNCH_RATES(true, false);
        }
      }
      if (last_batch && ns_eff > 0) HIPCHK(c, hipEventRecord(c->ev_slab[piece], st_));
    }
    if (pieces > 1) {
      HIPCHK(c, hipEventRecord(c->ev_join, c->stream3));
      HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_join, 0));
    }
#undef C2R_LAUNCH_RATES
    HIPCHK(c, hipGetLastError());
    c->tm.rates_launches++;
    if (c->timing) {
      HIPCHK(c, hipEventRecord(e_r1, c->stream2));
      tev.push_back(e_s0); tev.push_back(e_s1); tev.push_back(e_r0); tev.push_back(e_r1);
    }
    // The losses that are kept but decided nothing (rounds that were a source's last for geometric reasons): the
    // rates launch has left their terms in the column blocks; add them up behind it, on its stream, one launch per
    // final round (the same for every source of a mesh, so one launch), and send the sums to the host.
    c2r_ctx::BatchTail bt;
    bt.set = set;
    bt.loss.resize((size_t)nb);
    bt.slot.assign((size_t)nb, -1);
    bt.nbox.resize((size_t)nb);
    for (int b = 0; b < nb; b++) {
      bt.loss[(size_t)b] = run[b].loss;
      bt.nbox[(size_t)b] = run[b].nbox;
    }
    if (any_final) {
      int nslot = 0;
      for (;;) {
        int fr = -1;
        for (int b = 0; b < nb; b++)
          if (run[b].final_loss_due) { fr = run[b].nbox; break; }
        if (fr < 0) break;
        int *fl = hl + list_used;
        int nf = 0;
        for (int b = 0; b < nb; b++)
          if (run[b].final_loss_due && run[b].nbox == fr) {
            bt.slot[(size_t)b] = nslot + nf;
            fl[nf++] = b;
            run[b].final_loss_due = false;
          }
        HIPCHK(c, hipMemcpyAsync(c->d_list[set] + list_used, fl, sizeof(int) * nf, hipMemcpyHostToDevice, c->stream2));
        const Box fb = round_box(reach, fr);
        const int f_lo = fr > 1 ? box_smax(round_box(reach, fr - 1)) + 1 : 0, f_hi = box_smax(fb);
        const int nblk_l = c->block_base[f_hi + 1] - c->block_base[f_lo];
        const size_t need = (size_t)nblk_l * (size_t)nf;
        if (c->final_partial_cap[set] < need) {
          HIPCHK(c, hipStreamSynchronize(c->stream2));
          if (ensure_pair<double>(c, &c->d_final_partial[set], (double **)nullptr, &c->final_partial_cap[set], need)) return 1;
        }
        // a surface cell has one coordinate on a face of the box, so its shell is at least the nearest face's
        // distance: the shells before that hold none, their blocks are not launched and their partial sums are zero
        int s_first = 1 << 30;
        for (int d = 0; d < 3; d++) s_first = std::min(s_first, std::min(std::abs(fb.lo[d]), std::abs(fb.hi[d])));
        s_first = std::min(std::max(s_first, f_lo), f_hi);
        const int first_block = c->block_base[s_first] - c->block_base[f_lo];
        if (first_block > 0) HIPCHK(c, zero_device(c->d_final_partial[set], sizeof(double) * need, c->stream2));
        hipLaunchKernelGGL(k_loss_stored, dim3(nblk_l - first_block, nf), dim3(BLOCK), 0, c->stream2, c->g, c->d_src[set],
                           c->d_list[set] + list_used, f_lo, f_hi, fb, c->d_block_base, c->d_final_partial[set], nblk_l, first_block);
        hipLaunchKernelGGL(k_loss_finish, dim3(nf), dim3(BLOCK), 0, c->stream2, c->d_final_partial[set], nblk_l, nblk_l,
                           c->d_final_acc[set] + nslot);
        HIPCHK(c, hipGetLastError());
        list_used += (size_t)nf;
        nslot += nf;
      }
      HIPCHK(c, hipMemcpyAsync(c->h_final[set], c->d_final_acc[set], sizeof(double) * nslot, hipMemcpyDeviceToHost, c->stream2));
    } else {
      if (legacy_final_loss)
        for (;;) {
          int fr = -1;
          for (int b = 0; b < nb; b++)
            if (run[b].final_loss_due) { fr = run[b].nbox; break; }
          if (fr < 0) break;
          int *fl = hl + list_used;
          int nf = 0;
          for (int b = 0; b < nb; b++)
            if (run[b].final_loss_due && run[b].nbox == fr) fl[nf++] = b;
          HIPCHK(c, hipMemcpyAsync(c->d_list[set] + list_used, fl, sizeof(int) * nf, hipMemcpyHostToDevice, c->stream));
          const Box fb = round_box(reach, fr);
          const int f_lo = fr > 1 ? box_smax(round_box(reach, fr - 1)) + 1 : 0;
          if (boundary_loss(c, set, list_used, nf, f_lo, box_smax(fb), fb, 1, sc, ss, multi)) return 1;
          for (int j = 0; j < nf; j++) {
            bt.loss[(size_t)fl[j]] = c->h_loss[j];
            run[fl[j]].final_loss_due = false;
          }
          list_used += (size_t)nf;
        }
      bt.resolved = true;
    }
    c->tails.push_back(std::move(bt));
    HIPCHK(c, hipEventRecord(c->ev_rates_done[set], c->stream2));
    c->set_busy[set] = true;
    c->tm.cells_swept += batch_cells;
    c->last_src = run[nb - 1].ns;
    c->last_cols = c->h_src[set][nb - 1].cols;
    c->last_cz = c->h_src[set][nb - 1].cz;
    {
      const SrcDev &S = c->h_src[set][nb - 1];
      for (int d = 0; d < 3; d++) { c->last_lo[d] = S.lo[d]; c->last_hi[d] = S.hi[d]; }
    }
    b0 += (size_t)nb;
  }
  if (!transposed_seen) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_transposed, 0));
  // a rank without sources of its own still owes the caller its slab events
  if (mine.empty())
    for (int sidx = 0; sidx < ns_eff; sidx++) HIPCHK(c, hipEventRecord(c->ev_slab[sidx], c->stream2));
  c->pass_tev = tev;
  c->pass_open = true;
  c->pass_slabs = ns_eff;
  if (nslab > 0) return 0;
  return pass_finish(c);
}

static int pass_finish(c2r_ctx *c) {
  if (!c->pass_open) return 0;
  if (sync_stream(c, c->stream2, "the pass (rates stream)")) return 1;
  if (sync_stream(c, c->stream, "the pass")) return 1;
  c->set_busy[0] = c->set_busy[1] = false;
  // photon_loss(1) += photon_loss_src ; sum_nbox += nbox  (evolve_source.F90:233-236): batch by batch, source by
  // source, now that the losses the rates launches left behind have arrived
  resolve_tails(c, 0);
  resolve_tails(c, 1);
  for (const c2r_ctx::BatchTail &bt : c->tails)
    for (size_t b = 0; b < bt.loss.size(); b++) {
      c->photon_loss[0] = c->photon_loss[0] + bt.loss[b];
      c->sum_nbox += bt.nbox[b];
    }
  c->tails.clear();
  // tail of the reduction buffer: photon_loss(1:47), sum_nbox
  std::memcpy(c->h_tail, c->photon_loss, sizeof c->photon_loss);
  c->h_tail[C2R_NFREQ] = (double)c->sum_nbox;
  HIPCHK(c, hipMemcpyAsync(c->d_rates + 4 * c->g.ncell, c->h_tail, sizeof(double) * (C2R_NFREQ + 1), hipMemcpyHostToDevice, c->stream));
  if (sync_stream(c, c->stream, "the pass")) return 1;
  const std::vector<hipEvent_t> &tev = c->pass_tev;
  for (size_t i = 0; i + 3 < tev.size(); i += 4) {
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, tev[i], tev[i + 1]));
    c->tm.sweep_ms += ms;
    HIPCHK(c, hipEventElapsedTime(&ms, tev[i + 2], tev[i + 3]));
    c->tm.rates_ms += ms;
  }
  c->pass_open = false;
  return 0;
}

static int pass_sources_one(c2r_ctx *c, int first, int stride) {
  if (check_ready(c, "c2r_pass_sources")) return 1;
  if (first < 1 || stride < 1) return fail(c, "c2r_pass_sources: first=%d stride=%d", first, stride);
  std::vector<int> mine;
  for (int ns = first; ns <= c->nsrc; ns += stride) mine.push_back(ns);
  c->last_first = first;
  c->last_stride = stride;
  return pass_list(c, mine);
}

extern "C" int c2r_pass_sources_begin(c2r_ctx *c, int first, int stride, int nslab) {
  if (!c) return 1;
  if (!c->replicas.empty()) return fail(c, "c2r_pass_sources_begin: not available on a multi-device context");
  if (check_ready(c, "c2r_pass_sources_begin")) return 1;
  if (first < 1 || stride < 1 || nslab < 1) return fail(c, "c2r_pass_sources_begin: first=%d stride=%d nslab=%d", first, stride, nslab);
  std::vector<int> mine;
  for (int ns = first; ns <= c->nsrc; ns += stride) mine.push_back(ns);
  return pass_list(c, mine, nslab);
}

extern "C" int c2r_pass_slab_count(c2r_ctx *c) { return c && c->pass_open ? c->pass_slabs : 0; }

extern "C" int c2r_pass_wait_slab(c2r_ctx *c, int slab, size_t *first_cell, size_t *ncells) {
  if (!c) return 1;
  if (!c->pass_open || slab < 0 || slab >= c->pass_slabs) return fail(c, "c2r_pass_wait_slab: slab %d of %d (pass %s)", slab, c->pass_slabs, c->pass_open ? "open" : "not open");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipEventSynchronize(c->ev_slab[slab]));
  const size_t plane = (size_t)c->g.n1 * c->g.n2;
  if (first_cell) *first_cell = plane * (size_t)c->slab_k[slab];
  if (ncells) *ncells = plane * (size_t)(c->slab_k[slab + 1] - c->slab_k[slab]);
  return 0;
}

extern "C" int c2r_pass_sources_end(c2r_ctx *c) {
  if (!c) return 1;
  HIPCHK(c, hipSetDevice(c->device));
  return pass_finish(c);
}

extern "C" int c2r_do_source(c2r_ctx *c, int ns) {
  if (!c) return 1;
  if (!c->replicas.empty()) return fail(c, "c2r_do_source: not available on a multi-device context");
  if (check_ready(c, "c2r_do_source")) return 1;
  if (ns < 1 || ns > c->nsrc) return fail(c, "c2r_do_source: source %d not in [1,%d]", ns, c->nsrc);
  return pass_list(c, std::vector<int>{ns});
}

// The global pass over a range of cells, queued behind `after_event` (an event of another stream, e.g. the
// one on which the caller's sum over ranks of these cells completes; may be null).  A range starting at
// cell 0 opens a pass (zeroes the non-converged count); c2r_global_pass_finish closes it.
// Heating global pass in tiers.  The number of sub-steps thermal() takes for a cell spans four orders of magnitude
// (cells at an ionisation front sub-cycle the energy equation hundreds to thousands of times, over hundreds of
// do_chemistry iterations, next to neighbours that need two), a wave lasts as long as its slowest lane, and a sub-step
// is a chain of dependent operations some 2 700 cycles long (the log10 of coolin, ten cooling-curve look-ups, six
// divisions) with two waves per SIMD to hide it: a plain launch leaves nine tenths of the lane-time idle.  So every
// launch gives its cells a ceiling of sub-steps; a cell that needs more stores nothing, is put on a list and is redone
// from scratch, densely packed with its likes, by the next launch under a ceiling four times higher: 16, 64, 256,
// 1 024, then none -- and none earlier once the cells that are left fit the device all at once (two waves per SIMD:
// 131 072 cells): such a launch lasts as long as its longest chain whatever the others do.  The ladder starts at the
// lowest rung under which at least half of the PREVIOUS pass's cells finished (a smoothly ionised box whose cells all
// need some tens of sub-steps would otherwise do 16 of them in vain for every cell).  Within a tier the cells
// then differ by a factor of four at most, and what the dropped cells did in vain is a third of their useful work
// at worst.  (Rungs at 4 096 and 16 384 were measured too: 1-2 ms slower at 256^3, where they only add their own
// chains, and 0.5-1 s slower on 512^3 with 10^4 sources, where two thirds of the cells beyond 1 024 lie beyond 4 096
// as well and do 4 096 sub-steps in vain.)  (Rounds 1 and 2 chose ONE first ceiling from
// the previous pass's histogram, twice its 90th percentile, and went on in steps of eight: measured on the 128 faint
// sources of a configs[3] rank with heating, C2R_CHEM_LOG, that first launch took 90-100 ms of a 125 ms pass under a
// ceiling of 512 while two thirds of its cells needed fewer than 64 sub-steps.)  The last tier is the longest chain
// of the mesh -- some 14 000 sub-steps, 16 ms, however few cells are left -- and nothing on the device shortens that.
// Results do not depend on any of this: a dropped cell stores nothing and is recomputed from the same inputs.
constexpr int CHEM_FIRST_CEILING = 16, CHEM_CEILING_STEP = 4, CHEM_LAST_CEILING = 1024, CHEM_TIERS = 6, CHEM_RESIDENT_CELLS = 131072;

static int launch_chemistry(c2r_ctx *c, hipStream_t st, double dt, size_t first, size_t count, const int *list, int budget,
                            int *deferred, int *ndeferred) {
  if (count == 0) return 0;
  const Grid g = c->g;
  const StepScalars sc = scalars(c);
  // C2R_CHEM_LDS=0 (diagnostic): tables from global memory in every tier
  static const bool lds_tiers = !(getenv("C2R_CHEM_LDS") && atoi(getenv("C2R_CHEM_LDS")) == 0);
  const bool lds = !c->isothermal && list != nullptr && lds_tiers; // the repacked tiers of a heating pass
  const int bs = lds ? CHEM_BLOCK_LDS : C2R_CHEM_BLOCK;
  const int nblk = (int)((count + bs - 1) / bs);
  // a range of whole k-planes on a mesh whose sizes are multiples of 4: waves take cubes (see the kernel)
  const size_t plane = (size_t)g.n1 * g.n2;
  // ... while the pass before left more than 1/64 of the cells unconverged (or a step has just begun): once nearly
  // every cell is done after one iteration there is nothing to gain, and rows of 64 cells read their 17 grids in
  // longer runs.  Same box, ms per pass over the 8 iterations of a time step in ionised gas, rows | 4x4x4 | 8x4x2 cells:
  // 2.80 2.53 2.28 1.80 1.54 1.44 1.17 1.02 | 2.04 1.92 1.84 1.62 1.53 1.37 1.19 1.15 | 2.04 1.87 1.80 1.55 1.45 1.29 1.08 1.06.
  // C2R_CHEM_CUBES=0 / 1 / 2 (environment): rows always / 4 x 4 x 4 always / 8 x 4 x 2 always; default: 8 x 4 x 2 by the rule.
  static const int cubes_env = getenv("C2R_CHEM_CUBES") ? atoi(getenv("C2R_CHEM_CUBES")) : -1;
  const bool busy = c->last_conv < 0 || c->last_conv * 64 > (long long)g.ncell;
  const int cubes_want = cubes_env >= 0 ? cubes_env : (C2R_CHEM_CUBES && busy ? 2 : 0);
  const int cubes = cubes_want && !list && g.n1 % 8 == 0 && g.n2 % 4 == 0 && first % plane == 0 && count % (4 * plane) == 0 ? cubes_want : 0;
  if (c->isothermal)
    hipLaunchKernelGGL(k_chemistry<false>, dim3(nblk), dim3(bs), 0, st, g, sc, dt, c->d_ndens, c->d_xh,
                       c->d_xhe, c->d_xh_av, c->d_xhe_av, c->d_xh_int, c->d_xhe_int, c->d_temp, c->d_rates, c->d_chemspread,
                       c->d_rc_last, c->clumping_on_grid ? c->d_clump : nullptr, first, first + count, list, 0,
                       (int *)nullptr, (int *)nullptr, c->d_stateT, cubes);
  else if (lds)
    hipLaunchKernelGGL((k_chemistry<true, true>), dim3(nblk), dim3(bs), 0, st, g, sc, dt, c->d_ndens, c->d_xh,
                       c->d_xhe, c->d_xh_av, c->d_xhe_av, c->d_xh_int, c->d_xhe_int, c->d_temp, c->d_rates, c->d_chemspread,
                       c->d_rc_last, c->clumping_on_grid ? c->d_clump : nullptr, first, first + count, list, budget, deferred,
                       ndeferred, c->d_stateT, 0);
  else
    hipLaunchKernelGGL(k_chemistry<true>, dim3(nblk), dim3(bs), 0, st, g, sc, dt, c->d_ndens, c->d_xh,
                       c->d_xhe, c->d_xh_av, c->d_xhe_av, c->d_xh_int, c->d_xhe_int, c->d_temp, c->d_rates, c->d_chemspread,
                       c->d_rc_last, c->clumping_on_grid ? c->d_clump : nullptr, first, first + count, list, budget, deferred,
                       ndeferred, c->d_stateT, cubes);
  HIPCHK(c, hipGetLastError());
  c->tm.chem_launches++;
  return 0;
}

// the spread counters of the launches so far, folded into d_conv and (heating) the histogram behind d_chemctl + 2
static int fold_chemistry_counters(c2r_ctx *c, hipStream_t st) {
  hipLaunchKernelGGL(k_chem_ctl_reduce, dim3(1), dim3(64), 0, st, c->d_chemspread, c->d_conv,
                     c->isothermal ? (int *)nullptr : c->d_chemctl + 2);
  HIPCHK(c, hipGetLastError());
  return 0;
}

// Everything a global pass needs exists before it starts: the spread counters since c2r_create, the heating lists since
// the c2r_set_step that declared the step non-isothermal (round 3 allocated both here, inside the first pass).
static int ensure_chemistry_buffers(c2r_ctx *c) {
  if (!c->d_chemspread) return fail(c, "the chemistry counters are missing (c2r_create did not finish?)");
  if (!c->isothermal && !(c->d_defer[0] && c->d_defer[1] && c->d_chemctl))
    return fail(c, "a non-isothermal pass before c2r_set_step declared the step non-isothermal (or its lists could not be allocated)");
  return 0;
}

// evolve0D_global(dt,pos,conv_flag) (files_for_3D/evolve_point.F90:325-440) for ONE cell, as the reference's
// global_pass calls it (evolve.F90:477-484): applies the collected rates to the cell at 1-based mesh position
// pos and adds 1 to conv_flag if the cell has not converged.  A whole pass is c2r_global_pass; this entry point
// exists for hosts written against the per-cell interface (one launch and one synchronisation per call).
extern "C" int c2r_evolve0d_global(c2r_ctx *c, double dt, const int pos[3], int *conv_flag) {
  if (!c || !pos) return 1;
  if (check_ready(c, "c2r_evolve0d_global")) return 1;
  const Grid g = c->g;
  if (pos[0] < 1 || pos[0] > g.n1 || pos[1] < 1 || pos[1] > g.n2 || pos[2] < 1 || pos[2] > g.n3)
    return fail(c, "c2r_evolve0d_global: position (%d,%d,%d) outside the mesh", pos[0], pos[1], pos[2]);
  HIPCHK(c, hipSetDevice(c->device));
  if (flush_rates_zero(c)) return 1;
  if (ensure_chemistry_buffers(c)) return 1;
  const size_t q = (size_t)(pos[0] - 1) + (size_t)g.n1 * ((size_t)(pos[1] - 1) + (size_t)g.n2 * (size_t)(pos[2] - 1));
  HIPCHK(c, hipMemsetAsync(c->d_conv, 0, sizeof(int), c->stream));
  c->transposed_valid = false; // the launch keeps the mesh-ordered products of its cell up to date, not their (j,i,k) copy
  if (launch_chemistry(c, c->stream, dt, q, 1, nullptr, 0, c->d_defer[0], c->d_chemctl)) return 1;
  if (fold_chemistry_counters(c, c->stream)) return 1;
  HIPCHK(c, hipMemcpyAsync(c->h_conv, c->d_conv, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (conv_flag) *conv_flag += *c->h_conv;
  for (c2r_ctx *r : c->replicas) // the other devices of a multi-device context keep the same state
    if (c2r_evolve0d_global(r, dt, pos, nullptr)) { c->err = r->err; return 1; }
  return 0;
}

// evolve0D(dt,rtpos,ns,niter) for one cell (see the header): one small launch per call, in the caller's order
extern "C" int c2r_evolve0d(c2r_ctx *c, const int rtpos[3], int ns, int niter, int on_surface, double *loss) {
  if (!c || !rtpos) return 1;
  if (!c->replicas.empty()) return fail(c, "c2r_evolve0d: not available on a multi-device context");
  if (check_ready(c, "c2r_evolve0d")) return 1;
  if (ns < 1 || ns > c->nsrc) return fail(c, "c2r_evolve0d: source %d not in [1,%d]", ns, c->nsrc);
  if (c->pass_open) return fail(c, "c2r_evolve0d: a pass opened by c2r_pass_sources_begin is still open");
  HIPCHK(c, hipSetDevice(c->device));
  const Grid g = c->g;
  const int *sp = &c->srcpos[3 * (size_t)(ns - 1)];
  const int di = rtpos[0] - sp[0], dj = rtpos[1] - sp[1], dk = rtpos[2] - sp[2];
  const Reach R = mesh_reach(g);
  if (di < R.l[0] || di > R.r[0] || dj < R.l[1] || dj > R.r[1] || dk < R.l[2] || dk > R.r[2])
    return fail(c, "c2r_evolve0d: cell (%d,%d,%d) is beyond the reach of source %d at (%d,%d,%d)", rtpos[0], rtpos[1], rtpos[2], ns,
                sp[0], sp[1], sp[2]);
  if (flush_rates_zero(c)) return 1;
  const size_t w = (size_t)(2 * g.smax + 1), cz = w * w * w;
  if (!c->d_point_cols) {
    // a block that holds the whole mesh, zeroed once: only ever finite columns afterwards (see the arena)
    HIPCHK(c, hipMalloc(&c->d_point_cols, sizeof(double) * 6 * cz));
    HIPCHK(c, zero_device(c->d_point_cols, sizeof(double) * 6 * cz, c->stream));
    HIPCHK(c, hipMalloc(&c->d_point_loss, sizeof(double)));
    HIPCHK(c, hipHostMalloc(&c->h_point_loss, sizeof(double)));
  }
  c->point_ns = ns;
  c->point_niter = niter;
  const StepScalars sc = scalars(c);
  bool multi = false;
  const SedSet ss = sedset(c, &multi);
  SweepArgs SA;
  SA.g = g; SA.sc = sc;
  for (int d = 0; d < 3; d++) { SA.box.lo[d] = R.l[d]; SA.box.hi[d] = R.r[d]; }
  SA.ndens = c->d_ndens; SA.xh_av = c->d_xh_av; SA.xhe_av = c->d_xhe_av;
  SA.packed = SA.packedT = nullptr;
  SA.lls_grid = c->lls_on_grid ? c->d_lls : nullptr;
  SrcDev S{};
  S.i0 = sp[0]; S.j0 = sp[1]; S.k0 = sp[2];
  S.nflux = c->normflux[(size_t)ns - 1];
  for (int k = 0; k < 2; k++) S.nflux_sed[k] = c->normflux_sed[k].empty() ? 0.0 : c->normflux_sed[k][(size_t)ns - 1];
  S.cols = c->d_point_cols;
  S.cz = cz;
  S.loss_lo = -1;
  double *dl = on_surface ? c->d_point_loss : nullptr;
  if (!c->isothermal) c->phiheat_dirty = true;
#define C2R_LAUNCH_POINT(H, M) \
  hipLaunchKernelGGL((k_evolve0d<H, M>), dim3(1), dim3(64), 0, c->stream, SA, S, di, dj, dk, c->d_bands, ss, c->d_rates, dl)
  if (c->isothermal) {
    if (multi) C2R_LAUNCH_POINT(false, true); else C2R_LAUNCH_POINT(false, false);
  } else {
    if (multi) C2R_LAUNCH_POINT(true, true); else C2R_LAUNCH_POINT(true, false);
  }
#undef C2R_LAUNCH_POINT
  HIPCHK(c, hipGetLastError());
  if (on_surface) {
    HIPCHK(c, hipMemcpyAsync(c->h_point_loss, c->d_point_loss, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (loss) *loss = *c->h_point_loss;
  }
  return 0;
}

extern "C" int c2r_global_pass_cells(c2r_ctx *c, double dt, size_t first_cell, size_t ncells, void *after_event) {
  if (!c) return 1;
  if (check_ready(c, "c2r_global_pass_cells")) return 1;
  HIPCHK(c, hipSetDevice(c->device));
  if (flush_rates_zero(c)) return 1;
  const Grid g = c->g;
  if (first_cell > g.ncell || ncells > g.ncell - first_cell) return fail(c, "c2r_global_pass_cells: range outside the mesh");
  const bool heat = !c->isothermal;
  if (ensure_chemistry_buffers(c)) return 1;
  if (first_cell == 0) {
    HIPCHK(c, hipMemsetAsync(c->d_conv, 0, sizeof(int), c->stream));
    if (heat) HIPCHK(c, hipMemsetAsync(c->d_chemctl, 0, sizeof(int) * (4 + CHEM_HIST), c->stream));
    if (c->timing) HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
    // the second stream takes every other piece: it must see the zeroed counters (and everything before)
    HIPCHK(c, hipEventRecord(c->ev_sweep_done[0], c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_sweep_done[0], 0));
    c->tm.chem_launches = 0;
    c->chem_pieces = 0;
    c->packed_valid = c->transposed_valid = false;
    c->chem_cells = 0;
    c->chem_dt = dt;
  }
  c->chem_cells += ncells;
  // Pieces alternate between the two streams: every launch ends in a long thin tail; on alternating streams
  // the next piece fills the chip while the previous one drains.
  hipStream_t st = (c->chem_pieces++ & 1) ? c->stream2 : c->stream;
  if (after_event) HIPCHK(c, hipStreamWaitEvent(st, static_cast<hipEvent_t>(after_event), 0));
  return launch_chemistry(c, st, dt, first_cell, ncells, nullptr, heat ? c->chem_first : 0, c->d_defer[0], c->d_chemctl);
}

extern "C" int c2r_global_pass_finish(c2r_ctx *c, int *conv_flag) {
  if (!c) return 1;
  HIPCHK(c, hipSetDevice(c->device));
  // join the second stream
  HIPCHK(c, hipEventRecord(c->ev_rates_done[0], c->stream2));
  HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_rates_done[0], 0));
  if (!c->isothermal) {
    // later tiers: the cells dropped by the previous launch(es), densely packed
    // C2R_CHEM_LOG=1 (diagnostic): where the time of a heating pass goes, tier by tier, on stderr
    static const bool chem_log = getenv("C2R_CHEM_LOG") && atoi(getenv("C2R_CHEM_LOG")) > 0;
    auto t_tier = std::chrono::steady_clock::now();
    int cur = 0;
    long long ceiling = c->chem_first;
    for (int t = 1; t < CHEM_TIERS; t++) {
      int n = 0;
      HIPCHK(c, hipMemcpyAsync(&n, c->d_chemctl + cur, sizeof(int), hipMemcpyDeviceToHost, c->stream));
      if (sync_stream(c, c->stream, "a tier of the global pass")) return 1;
      if (chem_log) {
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "c2ray_hip: heating pass, tier %d (ceiling %lld sub-steps) done after %.1f ms: %d cells deferred\n", t - 1,
                ceiling, std::chrono::duration<double, std::milli>(now - t_tier).count(), n);
        t_tier = now;
      }
      if (n <= 0) break;
      const int nxt = cur ^ 1;
      ceiling *= CHEM_CEILING_STEP;
      HIPCHK(c, hipMemsetAsync(c->d_chemctl + nxt, 0, sizeof(int), c->stream));
      const bool last = t == CHEM_TIERS - 1 || n <= CHEM_RESIDENT_CELLS || ceiling > CHEM_LAST_CEILING;
      if (launch_chemistry(c, c->stream, c->chem_dt, 0, (size_t)n, c->d_defer[cur], last ? 0 : (int)ceiling, c->d_defer[nxt],
                           c->d_chemctl + nxt))
        return 1;
      cur = nxt;
      if (last) break;
    }
    if (fold_chemistry_counters(c, c->stream)) return 1;
    // where the next pass starts its ladder: the lowest rung under which at least half of this pass's cells finished
    // (a pass whose cells all need some tens of sub-steps would otherwise do 16 of them in vain for every cell)
    int hist[CHEM_HIST + 2];
    HIPCHK(c, hipMemcpyAsync(hist, c->d_chemctl + 2, sizeof hist, hipMemcpyDeviceToHost, c->stream));
    if (sync_stream(c, c->stream, "the global pass")) return 1;
    {
      long long total = 0, acc = 0;
      for (int b = 0; b < CHEM_HIST; b++) total += hist[b];
      int first = CHEM_FIRST_CEILING, b = 0;
      for (long long rung = CHEM_FIRST_CEILING; rung <= CHEM_LAST_CEILING; rung *= CHEM_CEILING_STEP) {
        for (; b < CHEM_HIST && (1LL << b) <= rung; b++) acc += hist[b]; // bucket b holds [2^(b-1), 2^b)
        first = (int)rung;
        if (2 * acc >= total) break;
      }
      c->chem_first = total > 0 ? first : CHEM_FIRST_CEILING;
    }
    if (chem_log) {
      fprintf(stderr, "c2ray_hip: heating pass, last tier done after %.1f ms; longest chain: %d thermal sub-steps, %d do_chemistry iterations; cells per power of two of sub-steps:",
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_tier).count(), hist[CHEM_HIST], hist[CHEM_HIST + 1]);
      for (int b = 0; b < CHEM_HIST; b++) fprintf(stderr, " %d", hist[b]);
      fprintf(stderr, "\n");
    }
  }
  if (c->isothermal && fold_chemistry_counters(c, c->stream)) return 1;
  if (c->timing) HIPCHK(c, hipEventRecord(c->ev[4], c->stream));
  // every cell's products for the next column sweep are up to date if the pieces of this pass covered the mesh
  // (pieces that overlap or leave gaps, a caller's business, leave the flag down: the next pass then packs anew)
  // C2R_PACK_AT_PASS_START=1 (diagnostic): never rely on the global pass's copy
  static const bool pack_always = getenv("C2R_PACK_AT_PASS_START") && atoi(getenv("C2R_PACK_AT_PASS_START")) > 0;
  const bool packed_now = c->chem_cells == c->g.ncell && !pack_always;
  // C2R_TRANSPOSE_AT_PASS_START=1 (diagnostic): the transposition at the start of the next pass, as before round 3
  static const bool transpose_late = getenv("C2R_TRANSPOSE_AT_PASS_START") && atoi(getenv("C2R_TRANSPOSE_AT_PASS_START")) > 0;
  if (packed_now && !transpose_late) {
    // ... and their (j,i,k)-ordered copy is made right away, on the third stream, while the host reads this pass's
    // results: at the start of the next pass it would share the device with the sweep's innermost shells (measured: 9-14
    // us each beside it, 5-7 us alone)
    HIPCHK(c, hipEventRecord(c->ev_chem_done, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->stream3, c->ev_chem_done, 0));
    hipLaunchKernelGGL(k_transpose_packed, dim3((c->g.n1 + 31) / 32, (c->g.n2 + 31) / 32, 3 * c->g.n3), dim3(BLOCK), 0, c->stream3, c->g,
                       c->d_stateT, c->d_stateT + 3 * c->g.ncell);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev_transposed, c->stream3));
  }
  if (c->want_iter_stats) {
    // c2r_iteration: the grid reductions the host loop wants after this pass, behind it in the same queue
    double *partial = c->d_iter, *partial_min = c->d_iter + (size_t)STAT_BLOCKS * ITER_NV, *res = partial_min + (size_t)STAT_BLOCKS * 2;
    const double *rc_dev = c->isothermal ? nullptr : c->d_rc_last;
    hipLaunchKernelGGL(k_iter_stats, dim3(STAT_BLOCKS), dim3(BLOCK), 0, c->stream, c->g.ncell, c->sc.rc, rc_dev, c->sc.clumping,
                       c->clumping_on_grid ? c->d_clump : nullptr, c->d_ndens, c->d_xh_int, c->d_xhe_int, c->d_xh_av, c->d_xhe_av,
                       partial, partial_min);
    hipLaunchKernelGGL(k_iter_stats_finish, dim3(1), dim3(BLOCK), 0, c->stream, partial, partial_min, STAT_BLOCKS, c->sc.rc, rc_dev, res);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->h_iter, res, sizeof(double) * (ITER_NV + 2 + 12), hipMemcpyDeviceToHost, c->stream));
    c->want_iter_stats = false;
  }
  HIPCHK(c, hipMemcpyAsync(c->h_conv, c->d_conv, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  if (sync_stream(c, c->stream, "the global pass")) return 1;
  if (c->timing) {
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev[3], c->ev[4]));
    c->tm.chem_ms = ms;
  }
  if (conv_flag) *conv_flag = *c->h_conv;
  c->last_conv = *c->h_conv;
  c->packed_valid = packed_now;
  c->transposed_valid = packed_now && !transpose_late;
  return 0;
}

static int global_pass_one(c2r_ctx *c, double dt, int *conv_flag) {
  if (c2r_global_pass_cells(c, dt, 0, c->g.ncell, nullptr)) return 1;
  return c2r_global_pass_finish(c, conv_flag);
}

extern "C" int c2r_download_rates(c2r_ctx *c, double *phih, double *phihe, double *phiheat, double *photon_loss47,
                                  int *sum_nbox) {
  if (!c) return 1;
  HIPCHK(c, hipSetDevice(c->device));
  if (flush_rates_zero(c)) return 1;
  const size_t nc = c->g.ncell;
  if (phih) HIPCHK(c, hipMemcpyAsync(phih, c->d_rates, sizeof(double) * nc, hipMemcpyDeviceToHost, c->stream));
  if (phihe) HIPCHK(c, hipMemcpyAsync(phihe, c->d_rates + nc, sizeof(double) * 2 * nc, hipMemcpyDeviceToHost, c->stream));
  if (phiheat) HIPCHK(c, hipMemcpyAsync(phiheat, c->d_rates + 3 * nc, sizeof(double) * nc, hipMemcpyDeviceToHost, c->stream));
  double tail[C2R_NFREQ + 1];
  HIPCHK(c, hipMemcpyAsync(tail, c->d_rates + 4 * nc, sizeof tail, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (photon_loss47) std::memcpy(photon_loss47, tail, sizeof(double) * C2R_NFREQ);
  if (sum_nbox) *sum_nbox = (int)(tail[C2R_NFREQ] + 0.5);
  return 0;
}

// the same for hosts that cannot pass a null pointer for an array (Fortran): bit 0 phih, bit 1 phihe, bit 2 phiheat
extern "C" int c2r_download_rates_sel(c2r_ctx *c, int which, double *phih, double *phihe, double *phiheat, double *photon_loss47,
                                      int *sum_nbox) {
  return c2r_download_rates(c, (which & 1) ? phih : nullptr, (which & 2) ? phihe : nullptr, (which & 4) ? phiheat : nullptr,
                            photon_loss47, sum_nbox);
}

extern "C" int c2r_get_loss(c2r_ctx *c, double *photon_loss47, int *sum_nbox) {
  if (!c) return 1;
  if (photon_loss47) std::memcpy(photon_loss47, c->photon_loss, sizeof c->photon_loss);
  if (sum_nbox) *sum_nbox = c->sum_nbox;
  return 0;
}

extern "C" int c2r_download_iter_state(c2r_ctx *c, double *xh_av, double *xhe_av, double *xh_intermed,
                                       double *xhe_intermed) {
  if (!c) return 1;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t nc = c->g.ncell;
  if (xh_av) HIPCHK(c, hipMemcpyAsync(xh_av, c->d_xh_av, sizeof(double) * 2 * nc, hipMemcpyDeviceToHost, c->stream));
  if (xhe_av) HIPCHK(c, hipMemcpyAsync(xhe_av, c->d_xhe_av, sizeof(double) * 3 * nc, hipMemcpyDeviceToHost, c->stream));
  if (xh_intermed)
    HIPCHK(c, hipMemcpyAsync(xh_intermed, c->d_xh_int, sizeof(double) * 2 * nc, hipMemcpyDeviceToHost, c->stream));
  if (xhe_intermed)
    HIPCHK(c, hipMemcpyAsync(xhe_intermed, c->d_xhe_int, sizeof(double) * 3 * nc, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

static int upload_rates_one(c2r_ctx *c, const double *phih, const double *phihe, const double *phiheat) {
  if (!c) return 1;
  HIPCHK(c, hipSetDevice(c->device));
  if (flush_rates_zero(c)) return 1;
  if (phiheat) c->phiheat_dirty = true;
  const size_t nc = c->g.ncell;
  if (phih) HIPCHK(c, hipMemcpyAsync(c->d_rates, phih, sizeof(double) * nc, hipMemcpyHostToDevice, c->stream));
  if (phihe) HIPCHK(c, hipMemcpyAsync(c->d_rates + nc, phihe, sizeof(double) * 2 * nc, hipMemcpyHostToDevice, c->stream));
  if (phiheat) HIPCHK(c, hipMemcpyAsync(c->d_rates + 3 * nc, phiheat, sizeof(double) * nc, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

static int upload_iter_state_one(c2r_ctx *c, const double *xh_av, const double *xhe_av, const double *xh_intermed,
                                     const double *xhe_intermed) {
  if (!c) return 1;
  c->packed_valid = c->transposed_valid = false;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t nc = c->g.ncell;
  if (xh_av) HIPCHK(c, hipMemcpyAsync(c->d_xh_av, xh_av, sizeof(double) * 2 * nc, hipMemcpyHostToDevice, c->stream));
  if (xhe_av) HIPCHK(c, hipMemcpyAsync(c->d_xhe_av, xhe_av, sizeof(double) * 3 * nc, hipMemcpyHostToDevice, c->stream));
  if (xh_intermed)
    HIPCHK(c, hipMemcpyAsync(c->d_xh_int, xh_intermed, sizeof(double) * 2 * nc, hipMemcpyHostToDevice, c->stream));
  if (xhe_intermed)
    HIPCHK(c, hipMemcpyAsync(c->d_xhe_int, xhe_intermed, sizeof(double) * 3 * nc, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int c2r_download_columns(c2r_ctx *c, double *coldensh_out, double *coldenshe_out) {
  if (!c) return 1;
  if (!c->last_cols || c->last_src < 1) return fail(c, "c2r_download_columns: no source has been swept yet");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t nc = c->g.ncell;
  if (!c->d_colgrid) HIPCHK(c, hipMalloc(&c->d_colgrid, sizeof(double) * 3 * nc));
  const int *p = &c->srcpos[3 * (size_t)(c->last_src - 1)];
  const int nblk = (int)((nc + BLOCK - 1) / BLOCK);
  SrcDev S{};
  S.i0 = p[0]; S.j0 = p[1]; S.k0 = p[2];
  for (int d = 0; d < 3; d++) { S.lo[d] = c->last_lo[d]; S.hi[d] = c->last_hi[d]; }
  S.cz = c->last_cz;
  hipLaunchKernelGGL(k_col_to_grid, dim3(nblk), dim3(BLOCK), 0, c->stream, c->g, S, c->last_cols, c->d_colgrid);
  HIPCHK(c, hipGetLastError());
  if (coldensh_out) HIPCHK(c, hipMemcpyAsync(coldensh_out, c->d_colgrid, sizeof(double) * nc, hipMemcpyDeviceToHost, c->stream));
  if (coldenshe_out)
    HIPCHK(c, hipMemcpyAsync(coldenshe_out, c->d_colgrid + nc, sizeof(double) * 2 * nc, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int c2r_state_sums(c2r_ctx *c, int which, double out5[5]) {
  if (!c || !out5) return 1;
  if (!c->have_state || !c->have_step) return fail(c, "c2r_state_sums: state / step not set");
  if (which < 0 || which > 2) return fail(c, "c2r_state_sums: which = %d not in {0,1,2}", which);
  HIPCHK(c, hipSetDevice(c->device));
  const double *xh = which == 0 ? c->d_xh : (which == 1 ? c->d_xh_int : c->d_xh_av);
  const double *xhe = which == 0 ? c->d_xhe : (which == 1 ? c->d_xhe_int : c->d_xhe_av);
  hipLaunchKernelGGL(k_state_sums, dim3(STAT_BLOCKS), dim3(BLOCK), 0, c->stream, c->g.ncell, c->d_ndens, xh, xhe, c->d_stat);
  hipLaunchKernelGGL(k_stat_finish<5>, dim3(1), dim3(BLOCK), 0, c->stream, c->d_stat, STAT_BLOCKS, c->d_stat + STAT_BLOCKS * 5);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(c->h_stat, c->d_stat + STAT_BLOCKS * 5, sizeof(double) * 5, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  // h0 = sum*vol*(1-abu_he) ... (photonstatistics.f90:139-143)
  out5[0] = c->h_stat[0] * c->sc.vol * (1.0 - abu_he);
  out5[1] = c->h_stat[1] * c->sc.vol * (1.0 - abu_he);
  out5[2] = c->h_stat[2] * c->sc.vol * abu_he;
  out5[3] = c->h_stat[3] * c->sc.vol * abu_he;
  out5[4] = c->h_stat[4] * c->sc.vol * abu_he;
  return 0;
}

extern "C" int c2r_fraction_means(c2r_ctx *c, int which, double out5[5]) {
  if (!c || !out5) return 1;
  if (!c->have_state) return fail(c, "c2r_fraction_means: state not set");
  if (which < 0 || which > 2) return fail(c, "c2r_fraction_means: which = %d not in {0,1,2}", which);
  HIPCHK(c, hipSetDevice(c->device));
  const double *xh = which == 0 ? c->d_xh : (which == 1 ? c->d_xh_int : c->d_xh_av);
  const double *xhe = which == 0 ? c->d_xhe : (which == 1 ? c->d_xhe_int : c->d_xhe_av);
  hipLaunchKernelGGL(k_state_sums, dim3(STAT_BLOCKS), dim3(BLOCK), 0, c->stream, c->g.ncell, (const double *)nullptr, xh, xhe,
                     c->d_stat);
  hipLaunchKernelGGL(k_stat_finish<5>, dim3(1), dim3(BLOCK), 0, c->stream, c->d_stat, STAT_BLOCKS, c->d_stat + STAT_BLOCKS * 5);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(c->h_stat, c->d_stat + STAT_BLOCKS * 5, sizeof(double) * 5, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int n = 0; n < 5; n++) out5[n] = c->h_stat[n] / (double)c->g.ncell;
  return 0;
}

extern "C" int c2r_fraction_minima(c2r_ctx *c, int which, double out2[2]) {
  if (!c || !out2) return 1;
  if (!c->have_state) return fail(c, "c2r_fraction_minima: state not set");
  if (which < 0 || which > 2) return fail(c, "c2r_fraction_minima: which = %d not in {0,1,2}", which);
  HIPCHK(c, hipSetDevice(c->device));
  const double *xh = which == 0 ? c->d_xh : (which == 1 ? c->d_xh_int : c->d_xh_av);
  const double *xhe = which == 0 ? c->d_xhe : (which == 1 ? c->d_xhe_int : c->d_xhe_av);
  hipLaunchKernelGGL(k_state_min, dim3(STAT_BLOCKS), dim3(BLOCK), 0, c->stream, c->g.ncell, xh, xhe, c->d_stat);
  hipLaunchKernelGGL(k_min_finish, dim3(1), dim3(BLOCK), 0, c->stream, c->d_stat, STAT_BLOCKS, c->d_stat + STAT_BLOCKS * 5);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(c->h_stat, c->d_stat + STAT_BLOCKS * 5, sizeof(double) * 2, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  out2[0] = c->h_stat[0];
  out2[1] = c->h_stat[1];
  return 0;
}

// The numerical and algorithmic parameters compiled into the device code, in the order of the header's list.
extern "C" int c2r_get_constants(double *out, int capacity) {
  const double v[] = {(double)SUBBOXSIZE, (double)MAX_SUBBOX, abu_he, abu_c, epsilon, convergence_fraction,
                      minimum_fractional_change, minimum_fraction_of_atoms, relative_denergy, minitemp,
                      (double)NTAU, minlogtau, 4.0 /* maxlogtau */, (double)NFREQ, (double)NHEAT};
  const int n = (int)(sizeof v / sizeof v[0]);
  for (int i = 0; i < n && i < capacity; i++) out[i] = v[i];
  return n;
}

extern "C" int c2r_total_rates(c2r_ctx *c, double dt, const double reccoef[12], double out3[3]) {
  if (!c || !reccoef || !out3) return 1;
  if (!c->have_state || !c->have_step) return fail(c, "c2r_total_rates: state / step not set");
  HIPCHK(c, hipSetDevice(c->device));
  RecCoef rc;
  std::memcpy(&rc, reccoef, sizeof rc);
  hipLaunchKernelGGL(k_total_rates, dim3(STAT_BLOCKS), dim3(BLOCK), 0, c->stream, c->g.ncell, rc, c->sc.clumping,
                     c->clumping_on_grid ? c->d_clump : nullptr, c->d_ndens, c->d_xh_av, c->d_xhe_av, c->d_stat);
  hipLaunchKernelGGL(k_stat_finish<3>, dim3(1), dim3(BLOCK), 0, c->stream, c->d_stat, STAT_BLOCKS, c->d_stat + STAT_BLOCKS * 5);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(c->h_stat, c->d_stat + STAT_BLOCKS * 5, sizeof(double) * 3, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  for (int n = 0; n < 3; n++) out3[n] = c->h_stat[n] * c->sc.vol * dt; // photonstatistics.f90:199-201
  return 0;
}

extern "C" int c2r_get_reccoef(c2r_ctx *c, double out12[12]) {
  if (!c || !out12) return 1;
  if (c->isothermal) {
    std::memcpy(out12, &c->sc.rc, sizeof(double) * 12);
    return 0;
  }
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(out12, c->d_rc_last, sizeof(double) * 12, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int c2r_arena_stats(const c2r_ctx *c, long long out[6]) {
  if (!c || !out) return 1;
  for (int i = 0; i < 5; i++) out[i] = c->arena_stats[i];
  out[5] = (long long)c->arena_total;
  return 0;
}
extern "C" size_t c2r_rates_count(const c2r_ctx *c) { return c ? c->rates_count : 0; }
extern "C" void *c2r_rates_device_ptr(c2r_ctx *c) {
  if (!c) return nullptr;
  (void)flush_rates_zero(c); // whoever asks for the address may read it
  return (void *)c->d_rates;
}
extern "C" int c2r_set_rates_buffer(c2r_ctx *c, void *device_ptr, size_t count) {
  if (!c) return 1;
  c->phiheat_dirty = true; // contents of the other buffer unknown
  if (!device_ptr) { c->d_rates = c->d_rates_own; return 0; }
  if (count < c->rates_count) return fail(c, "c2r_set_rates_buffer: %zu doubles given, %zu needed", count, c->rates_count);
  c->d_rates = (double *)device_ptr;
  return 0;
}
static int synchronize_one(c2r_ctx *c) {
  if (!c) return 1;
  HIPCHK(c, hipSetDevice(c->device));
  if (flush_rates_zero(c)) return 1;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return 0;
}
static int enable_timing_one(c2r_ctx *c, int on) {
  if (!c) return 1;
  c->timing = on != 0;
  return 0;
}
extern "C" int c2r_get_timing(c2r_ctx *c, c2r_timing *out) {
  if (!c || !out) return 1;
  *out = c->tm;
  return 0;
}

// ---------------------------------------------------------------------------------------------
// state-setting entry points: the same call on every device of a multi-device context
extern "C" int c2r_set_tables(c2r_ctx *c, const double *photo_thick, const double *photo_thin, const double *heat_thick, const double *heat_thin, const double *sigma_HI, const double *sigma_HeI, const double *sigma_HeII, const double *const fvec[12], int bb_upper) {
  if (int e_ = set_tables_one(c, photo_thick, photo_thin, heat_thick, heat_thin, sigma_HI, sigma_HeI, sigma_HeII, fvec, bb_upper)) return e_;
  return for_replicas(c, [&](c2r_ctx *r) { return set_tables_one(r, photo_thick, photo_thin, heat_thick, heat_thin, sigma_HI, sigma_HeI, sigma_HeII, fvec, bb_upper); });
}

extern "C" int c2r_set_cooling(c2r_ctx *c, const double *cool, double mintemp, double dtemp) {
  if (int e_ = set_cooling_one(c, cool, mintemp, dtemp)) return e_;
  return for_replicas(c, [&](c2r_ctx *r) { return set_cooling_one(r, cool, mintemp, dtemp); });
}

extern "C" int c2r_set_step(c2r_ctx *c, const double *ndens, const double dr[3], double vol, float clumping, double zred, double H0, double Omega0, int isothermal, double temper_val, const double reccoef[12]) {
  if (int e_ = set_step_one(c, ndens, dr, vol, clumping, zred, H0, Omega0, isothermal, temper_val, reccoef)) return e_;
  return for_replicas(c, [&](c2r_ctx *r) { return set_step_one(r, ndens, dr, vol, clumping, zred, H0, Omega0, isothermal, temper_val, reccoef); });
}

extern "C" int c2r_set_step_scalars(c2r_ctx *c, const double dr[3], double vol, float clumping, double zred, double H0, double Omega0, int isothermal, double temper_val, const double reccoef[12]) {
  if (set_step_one(c, nullptr, dr, vol, clumping, zred, H0, Omega0, isothermal, temper_val, reccoef)) return 1;
  return for_replicas(c, [&](c2r_ctx *r) { return set_step_one(r, nullptr, dr, vol, clumping, zred, H0, Omega0, isothermal, temper_val, reccoef); });
}
static int scale_ndens_one(c2r_ctx *c, double divisor) {
  if (!c->have_step) return fail(c, "c2r_scale_ndens: no density on the device yet");
  if (!(divisor > 0.0) || !std::isfinite(divisor)) return fail(c, "c2r_scale_ndens: divisor %g", divisor);
  HIPCHK(c, hipSetDevice(c->device));
  const int nblk = (int)std::min<size_t>(65535, (c->g.ncell + BLOCK - 1) / BLOCK);
  hipLaunchKernelGGL(k_divide_by, dim3(nblk), dim3(BLOCK), 0, c->stream, c->d_ndens, c->g.ncell, divisor);
  HIPCHK(c, hipGetLastError());
  c->packed_valid = c->transposed_valid = false; // the products neufrac * ndens of the sweep are stale
  return 0;
}
extern "C" int c2r_scale_ndens(c2r_ctx *c, double divisor) {
  if (!c) return 1;
  if (scale_ndens_one(c, divisor)) return 1;
  return for_replicas(c, [&](c2r_ctx *r) { return scale_ndens_one(r, divisor); });
}
extern "C" int c2r_set_sources(c2r_ctx *c, int nsrc, const int *srcpos, const double *normflux, double s_star) {
  if (int e_ = set_sources_one(c, nsrc, srcpos, normflux, s_star)) return e_;
  return for_replicas(c, [&](c2r_ctx *r) { return set_sources_one(r, nsrc, srcpos, normflux, s_star); });
}

extern "C" int c2r_set_sed_tables(c2r_ctx *c, int sed, const double *photo_thick, const double *photo_thin, const double *heat_thick, const double *heat_thin, int lower, int upper) {
  if (int e_ = set_sed_tables_one(c, sed, photo_thick, photo_thin, heat_thick, heat_thin, lower, upper)) return e_;
  return for_replicas(c, [&](c2r_ctx *r) { return set_sed_tables_one(r, sed, photo_thick, photo_thin, heat_thick, heat_thin, lower, upper); });
}

extern "C" int c2r_set_sources_sed(c2r_ctx *c, int sed, const double *normflux, double s_star) {
  if (int e_ = set_sources_sed_one(c, sed, normflux, s_star)) return e_;
  return for_replicas(c, [&](c2r_ctx *r) { return set_sources_sed_one(r, sed, normflux, s_star); });
}

extern "C" int c2r_build_tables(c2r_ctx *c, const c2r_sed_setup *S, int with_heat) {
  if (int e_ = build_tables_one(c, S, with_heat)) return e_;
  return for_replicas(c, [&](c2r_ctx *r) { return build_tables_one(r, S, with_heat); });
}

extern "C" int c2r_set_lls(c2r_ctx *c, int use_lls, double coldensh_lls, const float *lls_grid) {
  if (int e_ = set_lls_one(c, use_lls, coldensh_lls, lls_grid)) return e_;
  return for_replicas(c, [&](c2r_ctx *r) { return set_lls_one(r, use_lls, coldensh_lls, lls_grid); });
}

extern "C" int c2r_set_clumping_grid(c2r_ctx *c, const float *clumping_grid) {
  if (int e_ = set_clumping_grid_one(c, clumping_grid)) return e_;
  return for_replicas(c, [&](c2r_ctx *r) { return set_clumping_grid_one(r, clumping_grid); });
}

extern "C" int c2r_upload_state(c2r_ctx *c, const double *xh, const double *xhe, const float *temperature) {
  if (int e_ = upload_state_one(c, xh, xhe, temperature)) return e_;
  return for_replicas(c, [&](c2r_ctx *r) { return upload_state_one(r, xh, xhe, temperature); });
}

extern "C" int c2r_begin_step(c2r_ctx *c) {
  if (int e_ = begin_step_one(c)) return e_;
  return for_replicas(c, [&](c2r_ctx *r) { return begin_step_one(r); });
}

extern "C" int c2r_end_step(c2r_ctx *c) {
  if (int e_ = end_step_one(c)) return e_;
  return for_replicas(c, [&](c2r_ctx *r) { return end_step_one(r); });
}

extern "C" int c2r_set_rates_to_zero(c2r_ctx *c) {
  if (int e_ = set_rates_to_zero_one(c)) return e_;
  return for_replicas(c, [&](c2r_ctx *r) { return set_rates_to_zero_one(r); });
}

extern "C" int c2r_set_batch(c2r_ctx *c, int nbatch) {
  if (int e_ = set_batch_one(c, nbatch)) return e_;
  return for_replicas(c, [&](c2r_ctx *r) { return set_batch_one(r, nbatch); });
}

extern "C" int c2r_upload_rates(c2r_ctx *c, const double *phih, const double *phihe, const double *phiheat) {
  if (int e_ = upload_rates_one(c, phih, phihe, phiheat)) return e_;
  return for_replicas(c, [&](c2r_ctx *r) { return upload_rates_one(r, phih, phihe, phiheat); });
}

extern "C" int c2r_upload_iter_state(c2r_ctx *c, const double *xh_av, const double *xhe_av, const double *xh_intermed, const double *xhe_intermed) {
  if (int e_ = upload_iter_state_one(c, xh_av, xhe_av, xh_intermed, xhe_intermed)) return e_;
  return for_replicas(c, [&](c2r_ctx *r) { return upload_iter_state_one(r, xh_av, xhe_av, xh_intermed, xhe_intermed); });
}

extern "C" int c2r_enable_timing(c2r_ctx *c, int on) {
  if (int e_ = enable_timing_one(c, on)) return e_;
  return for_replicas(c, [&](c2r_ctx *r) { return enable_timing_one(r, on); });
}

extern "C" int c2r_synchronize(c2r_ctx *c) {
  if (int e_ = synchronize_one(c)) return e_;
  return for_replicas(c, [&](c2r_ctx *r) { return synchronize_one(r); });
}

#ifdef C2R_CHEM_NIT_HIST
// diagnostic build only: histogram of do_chemistry iterations (k_chemistry), read and optionally reset
extern "C" int c2r_debug_chem_nit(unsigned long long out[136], int reset) {
  static unsigned long long h[64 * 136];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(c2r_chem_nit), sizeof(h)) != hipSuccess) return 1;
  for (int k = 0; k < 136; k++) {
    out[k] = 0;
    for (int s = 0; s < 64; s++) out[k] += h[s * 136 + k];
  }
  if (reset) {
    memset(h, 0, sizeof(h));
    if (hipMemcpyToSymbol(HIP_SYMBOL(c2r_chem_nit), h, sizeof(h)) != hipSuccess) return 1;
  }
  return 0;
}
#endif
#ifdef C2R_RATES_COUNT
// diagnostic build only: see count_lanes (c2ray_device.hpp)
__device__ unsigned long long c2r::c2r_rates_cnt[64 * 24];
extern "C" int c2r_debug_rates_counters(unsigned long long out[24], int reset) {
  unsigned long long h[64 * 24];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(c2r::c2r_rates_cnt), sizeof(h)) != hipSuccess) return 1;
  for (int k = 0; k < 24; k++) {
    out[k] = 0;
    for (int s = 0; s < 64; s++) out[k] += h[s * 24 + k];
  }
  if (reset) {
    memset(h, 0, sizeof(h));
    if (hipMemcpyToSymbol(HIP_SYMBOL(c2r::c2r_rates_cnt), h, sizeof(h)) != hipSuccess) return 1;
  }
  return 0;
}
#endif
#include "c2ray_comm.inc"
