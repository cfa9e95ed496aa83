/* C ABI of the MI355X (gfx950) implementation of C2-Ray's evolve3D hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  Each entry
 * point names the reference interface it replaces (paths relative to the reference's
 * code/ directory).  Host arrays are caller-owned contiguous Fortran arrays, i fastest,
 * components slowest: xh(N1,N2,N3,0:1), xhe(N1,N2,N3,0:2), temperature_grid(N1,N2,N3,0:2) real(4).
 * The library owns device memory only.  All calls return 0 on success; on failure they return a
 * non-zero status and c2r_last_error() describes it (the reference has no error convention: its
 * Fortran shim logs the text to unit logf and stops).
 *
 * Threading: one host thread per context; one context per GPU (one rank per GPU).
 */
#ifndef C2RAY_HIP_H
#define C2RAY_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct c2r_ctx c2r_ctx;

#define C2R_NFREQ 47  /* radiation_sizes.f90:22 NumFreqBnd */
#define C2R_NHEAT 113 /* radiation_sizes.f90:23 NumheatBin */
#define C2R_NTAU 2000 /* radiation_sizes.f90:18 NumTau */
#define C2R_NCOOL 801 /* cooling_h.f90:25 temppoints */

/* evolve_ini (files_for_3D/evolve_data.F90:74-97): allocate the work arrays, here on `device`.
 * mesh = sizes.f90:31. */
int c2r_create(c2r_ctx **out, int device, const int mesh[3]);
void c2r_destroy(c2r_ctx *ctx);
const char *c2r_last_error(const c2r_ctx *ctx);
/* error text of a failed c2r_create (no context exists yet) */
const char *c2r_create_error(void);

/* What rad_ini leaves behind (radiation_tables.f90:141-168, radiation_sizes.f90:62-688):
 * bb_photo_thick/thin_table(0:NumTau,1:NumFreqBnd), bb_heat_thick/thin_table(0:NumTau,1:NumheatBin)
 * (tau index fastest; heat tables may be NULL for isothermal-only use), sigma_HI/HeI/HeII(1:47),
 * the twelve secondary-ionisation vectors f1ion_HI .. f2heat_HeII, each (2:47) = 46 doubles, in the
 * order f1ion_{HI,HeI,HeII}, f2ion_{..}, f1heat_{..}, f2heat_{..} (may be NULL with the heat tables),
 * and bb_FreqBnd_UpperLimit (radiation_tables.f90:193-199). */
/* photo_thick/photo_thin and heat_thick/heat_thin may be NULL: the call then only sets the band vectors and
 * c2r_build_tables makes the tables on the device. */
int c2r_set_tables(c2r_ctx *ctx, const double *photo_thick, const double *photo_thin,
                   const double *heat_thick, const double *heat_thin, const double *sigma_HI,
                   const double *sigma_HeI, const double *sigma_HeII, const double *const fvec[12],
                   int bb_upper);

/* What setup_cool leaves behind (cooling_h.f90:76-171): five linear cooling curves of 801 points
 * (H0, H1, He0, He1, He2), log10 T of the first row and the step. */
int c2r_set_cooling(c2r_ctx *ctx, const double *cool5x801, double mintemp, double dtemp);

/* Host state read at every evolve3D call because the driver rescales it each step
 * (cosmology.f90:159-202): material:ndens, grid:dr,vol, material:clumping (real(4)), cosmology:zred,
 * cosmology_parameters:H0,Omega0, material:isothermal,temper_val, and the twelve module-global
 * coefficients of cgsconstants.f90:106-133 in the order arech0, brech0, areche0, breche0, oreche0,
 * areche1, breche1, treche1, colli_HI, colli_HeI, colli_HeII, v (used as they stand when isothermal;
 * re-evaluated per cell from the temperature otherwise, evolve_point.F90:543). */
int c2r_set_step(c2r_ctx *ctx, const double *ndens, const double dr[3], double vol, float clumping,
                 double zred, double H0, double Omega0, int isothermal, double temper_val,
                 const double reccoef[12]);
/* The same without the density: material:ndens on the device stays as the last c2r_set_step (or c2r_scale_ndens) left it.
 * For hosts that know ndens has not changed since (a run without cosmological expansion), or has only been rescaled: */
int c2r_set_step_scalars(c2r_ctx *ctx, const double dr[3], double vol, float clumping, double zred, double H0, double Omega0,
                         int isothermal, double temper_val, const double reccoef[12]);
/* cosmo_evol's  ndens(:,:,:) = ndens(:,:,:) / zfactor3  (cosmology.f90:193) applied to the device copy: one correctly
 * rounded IEEE division per cell, the same bits as the host's.  The Fortran drop-in uses the two calls when
 * C2RAY_HIP_KEEP_STATE is set and a sample of the host array confirms that this is all that happened to it. */
int c2r_scale_ndens(c2r_ctx *ctx, double divisor);


/* Table construction on the device: spec_integration (radiation_tables.f90:172-422) for one SED from what
 * spectrum_parms, setup_scalingfactors (radiation_sizes.f90:62-688), romberg_initialisation(NumFreq)
 * (romberg.f90:24-92) and normalize_seds leave behind.  All of it is public module data of the reference:
 *   freq_min, delta_freq (47)          radiation_sizes
 *   xsec_index (47)                    cross_section_HI_powerlaw_index(1), ..HeI..(2:27), ..HeII..(28:47), the
 *                                      index spec_integration passes per band (radiation_tables.f90:278,315,349)
 *   tau (0:NumTau)                     radiation_tables:tau, romw = romberg:romw(0:NumFreq, 9)
 *   R_star2, h_over_kT                 radiation_sed_parameters (black body); pl_scaling/pl_index or
 *                                      qpl_scaling/qpl_index for sed = 1 / 2
 *   two_pi_over_c_square, hplanck (cgsconstants), pi (mathconstants), ion_freq_* (cgsphotoconstants)
 * The band vectors (c2r_set_tables with NULL tables) or the band range (c2r_set_sed_tables with NULL tables)
 * must have been given before.  Results are bit-identical to the reference's host-built tables.
 * c2r_download_tables returns tables in the reference's layout (0:NumTau, ncol); pointers may be NULL. */
typedef struct c2r_sed_setup {
  int nfreq;                /* NumFreq = 512 */
  int sed;                  /* 0 black body, 1 power law, 2 quasar-like power law */
  const double *freq_min, *delta_freq, *xsec_index, *tau, *romw;
  double R_star2, h_over_kT, two_pi_over_c_square, hplanck, pi;
  double ion_freq_HI, ion_freq_HeI, ion_freq_HeII;
  double pl_scaling, pl_index;
} c2r_sed_setup;
int c2r_build_tables(c2r_ctx *ctx, const c2r_sed_setup *setup, int with_heat);
int c2r_download_tables(c2r_ctx *ctx, int sed, double *photo_thick, double *photo_thin, double *heat_thick,
                        double *heat_thin);

/* Lyman-limit systems (c2ray_parameters.f90:72-78 use_LLS, type_of_LLS): a fog of unresolved absorbers added
 * to the incoming HI column of every cell but the source's, coldensh_in += coldensh_LLS*path/dr(1)
 * (evolve_point.F90:177-180).  use_lls = 0 switches it off (the reference's default); lls_grid == NULL is
 * type_of_LLS = 1 with material:coldensh_LLS as set by set_LLS each step (mat_ini_test.F90:640-662);
 * lls_grid != NULL is type_of_LLS = 2: material's REAL(4) LLS_grid(mesh) read per cell by LLS_point
 * (mat_ini_cubep3m.F90:859-870).  Stays in force until the next call.  LLS_loss stays 0 as in the
 * reference (it multiplies photo_in_HI, which photoion_rates never sets). */
int c2r_set_lls(c2r_ctx *ctx, int use_lls, double coldensh_lls, const float *lls_grid);

/* Position-dependent clumping, type_of_clumping = 5: material's REAL(4) clumping_grid(mesh), read per cell
 * by clumping_point in do_chemistry (evolve_point.F90:483-484) and in total_rates
 * (photonstatistics.f90:175-177).  NULL returns to the scalar material:clumping of c2r_set_step. */
int c2r_set_clumping_grid(c2r_ctx *ctx, const float *clumping_grid);

/* sourceprops: NumSrc, srcpos(3,NumSrc) (1-based mesh coordinates), NormFlux(1:NumSrc), and
 * radiation_sed_parameters:S_star (sourceprops_test.F90:38-40). */
int c2r_set_sources(c2r_ctx *ctx, int nsrc, const int *srcpos, const double *normflux, double s_star);

/* The -DPL / -DQUASARS builds of the reference give every source up to two more SEDs
 * (radiation_photoionrates.f90:215-228, 256-271).  sed = 1: power law (pl_*), sed = 2: quasar-like (qpl_*).
 *   c2r_set_sed_tables   pl_/qpl_photo_thick/thin_table, pl_/qpl_heat_thick/thin_table (heat may be NULL
 *                        for isothermal-only use) and pl_/qpl_FreqBnd_LowerLimit..UpperLimit (1-based,
 *                        inclusive; radiation_tables.f90:207-256)
 *   c2r_set_sources_sed  NormFluxPL / NormFluxQPL(1:NumSrc) and pl_S_star / qpl_S_star; call after
 *                        c2r_set_sources (which clears them); NULL switches the SED off again */
int c2r_set_sed_tables(c2r_ctx *ctx, int sed, const double *photo_thick, const double *photo_thin,
                       const double *heat_thick, const double *heat_thin, int lower, int upper);
int c2r_set_sources_sed(c2r_ctx *ctx, int sed, const double *normflux, double s_star);

/* material:xh, xhe, temperature_grid (temperature may be NULL when isothermal) */
int c2r_upload_state(c2r_ctx *ctx, const double *xh, const double *xhe, const float *temperature);
int c2r_download_state(c2r_ctx *ctx, double *xh, double *xhe, float *temperature);

/* evolve3D(time,dt,restart) with restart == 0 (files_for_3D/evolve.F90:78-229): the whole
 * convergence loop on the device.  niter_out: number of outer iterations; conv_flags_out (may be
 * NULL, capacity cap): non-converged count after each global pass (evolve.F90:488). */
int c2r_evolve3d(c2r_ctx *ctx, double dt, int *niter_out, int *conv_flags_out, int cap);

/* The pieces of evolve3D, for hosts that drive the loop themselves (multi-rank runs reduce the
 * rate grids between c2r_pass_sources and c2r_global_pass):
 *   c2r_begin_step      evolve.F90:131-136  xh_av = xh_intermed = xh, xhe_* likewise
 *   c2r_set_rates_to_zero evolve.F90:371-381 (carried out by the next pass's first rates launch, or by whoever reads
 *                       the grids first: nothing observable differs)
 *   c2r_pass_sources    do_grid_static (master_slave.F90:74-96): do_source for ns = first,
 *                       first+stride, ... <= NumSrc (1-based) -- evolve_source.F90:66-238
 *   c2r_global_pass     evolve.F90:435-501 loop: evolve0D_global for every cell
 *   c2r_end_step        evolve.F90:164-166  xh = xh_intermed, xhe = xhe_intermed,
 *                       set_final_temperature_point */
int c2r_begin_step(c2r_ctx *ctx);
int c2r_set_rates_to_zero(c2r_ctx *ctx);
int c2r_pass_sources(c2r_ctx *ctx, int first, int stride);
/* The same pass, handing the rate grids over in nslab slabs of k-planes while later slabs are still being
 * computed, so that a multi-rank host can start the sum over ranks (mpi_accumulate_grid_quantities,
 * evolve.F90:505-548) of slab s while the device works on slabs s+1..:
 *   c2r_pass_sources_begin  queues the whole pass and returns without waiting for the device
 *   c2r_pass_slab_count     slabs of the open pass (<= nslab: at least one 4-plane tile layer each)
 *   c2r_pass_wait_slab      blocks until phih/phihe/phiheat of slab `slab` hold this rank's final sums;
 *                           the slab is cells [first_cell, first_cell + ncells) of every component grid
 *   c2r_pass_sources_end    waits for the rest (photon_loss, sum_nbox, timing); must close every begin */
int c2r_pass_sources_begin(c2r_ctx *ctx, int first, int stride, int nslab);
int c2r_pass_slab_count(c2r_ctx *ctx);
int c2r_pass_wait_slab(c2r_ctx *ctx, int slab, size_t *first_cell, size_t *ncells);
int c2r_pass_sources_end(c2r_ctx *ctx);
/* do_source(dt,ns1,niter) (evolve_source.F90:66-238) for the single source ns1 (1-based): trace it
 * and add its contribution to the rate grids, photon_loss(1) and sum_nbox. */
int c2r_do_source(c2r_ctx *ctx, int ns);
int c2r_global_pass(c2r_ctx *ctx, double dt, int *conv_flag);
/* The same pass in pieces, for a multi-rank host that applies the rates of a slab of cells as soon as their
 * sum over ranks is complete: c2r_global_pass_cells queues evolve0D_global for cells [first_cell, first_cell +
 * ncells) on the library's stream, after `after_event` (a hipEvent_t recorded on the stream that finishes the
 * sum, or NULL); the piece starting at cell 0 opens a pass.  c2r_global_pass_finish waits for all pieces and
 * returns the non-converged count of the pass. */
int c2r_global_pass_cells(c2r_ctx *ctx, double dt, size_t first_cell, size_t ncells, void *after_event);
int c2r_global_pass_finish(c2r_ctx *ctx, int *conv_flag);
int c2r_end_step(c2r_ctx *ctx);

/* evolve_data: phih_grid, phihe_grid(:,:,:,0:1), phiheat; photonstatistics: photon_loss(1:47)
 * (as summed over this rank's sources, i.e. photon_loss_all before the division by mesh^3 of
 * evolve.F90:457); evolve_source: sum_nbox.  Any pointer may be NULL. */
int c2r_download_rates(c2r_ctx *ctx, double *phih, double *phihe, double *phiheat,
                       double *photon_loss47, int *sum_nbox);
/* The same with the grids chosen by a mask (bit 0 phih, bit 1 phihe, bit 2 phiheat) instead of by null pointers, for
 * hosts whose language cannot pass a null array: the Fortran drop-in leaves phiheat on the device in isothermal runs
 * (the grid is zero there and on the host, evolve_data.F90:80 -- 134 MB of PCIe per evolve3D call at 256^3). */
int c2r_download_rates_sel(c2r_ctx *ctx, int which, double *phih, double *phihe, double *phiheat,
                           double *photon_loss47, int *sum_nbox);
/* photon_loss(1:47) and sum_nbox of this rank's sources since the last c2r_set_rates_to_zero, from
 * the host-side bookkeeping (no device copy). */
int c2r_get_loss(c2r_ctx *ctx, double *photon_loss47, int *sum_nbox);
/* evolve_data: xh_av, xhe_av, xh_intermed, xhe_intermed -- the iteration-dump content
 * (write_iteration_dump, evolve.F90:233-275). */
int c2r_download_iter_state(c2r_ctx *ctx, double *xh_av, double *xhe_av, double *xh_intermed,
                            double *xhe_intermed);
/* start_from_dump (evolve.F90:279-367) reloads exactly these arrays before calling global_pass:
 * phih_grid, phihe_grid, [phiheat], xh_av, xhe_av, xh_intermed, xhe_intermed.  Any pointer may be
 * NULL (left as is). */
int c2r_upload_rates(c2r_ctx *ctx, const double *phih, const double *phihe, const double *phiheat);
int c2r_upload_iter_state(c2r_ctx *ctx, const double *xh_av, const double *xhe_av,
                          const double *xh_intermed, const double *xhe_intermed);
/* evolve_data: coldensh_out, coldenshe_out(:,:,:,0:1) of the source swept last (diagnostic). */
int c2r_download_columns(c2r_ctx *ctx, double *coldensh_out, double *coldenshe_out);

/* Photon statistics on the device (files_for_3D/photonstatistics.f90) -- the grid reductions the
 * reference runs on the host before and after every evolve3D call:
 *   c2r_state_sums   state_before (:117-144) / state_after (:208-234): number of H0, H+, He0, He+, He++
 *                    (sum of ndens*x times vol*(1-abu_he) or vol*abu_he) of which = 0: xh,xhe;
 *                    1: xh_intermed,xhe_intermed; 2: xh_av,xhe_av
 *   c2r_total_rates  total_rates (:150-203) on the time-averaged fractions xh_av,xhe_av with the given
 *                    twelve coefficients: out = totrec, totcollisions, recomions (already x vol x dt)
 *   c2r_get_reccoef  the module-global coefficients of cgsconstants.f90:106-133 as the reference's
 *                    global pass leaves them (isothermal: unchanged; otherwise what the last cell
 *                    (mesh,mesh,mesh) computed last, evolve_point.F90:543) -- total_rates uses these.
 * The sums are deterministic but associate differently from the reference's serial loops: they agree
 * with it to rounding, not bit for bit. */
int c2r_state_sums(c2r_ctx *ctx, int which, double out5[5]);
/* sum(x(:,:,:,n))/mesh^3 for the five fractions of `which` -- the "Intermediate result for mean
 * ... ionization fraction" lines of global_pass (evolve.F90:489-494) */
int c2r_fraction_means(c2r_ctx *ctx, int which, double out5[5]);
int c2r_total_rates(c2r_ctx *ctx, double dt, const double reccoef[12], double out3[3]);
int c2r_get_reccoef(c2r_ctx *ctx, double out12[12]);

/* The buffer that mpi_accumulate_grid_quantities (evolve.F90:505-548) sums over ranks, as ONE
 * contiguous device array of c2r_rates_count() doubles:
 *   [ phih_grid | phihe_grid(0) | phihe_grid(1) | phiheat | photon_loss(1:47) | sum_nbox ]
 * so that a single all-reduce(SUM, fp64) replaces the reference's four grid all-reduces and two
 * small ones.  c2r_rates_device_ptr returns its device address (after c2r_synchronize the data is
 * complete); c2r_set_rates_buffer makes the library use a caller-allocated device buffer instead
 * (e.g. a torch tensor that torch.distributed/RCCL reduces in place). */
size_t c2r_rates_count(const c2r_ctx *ctx);
void *c2r_rates_device_ptr(c2r_ctx *ctx);
int c2r_set_rates_buffer(c2r_ctx *ctx, void *device_ptr, size_t count);
int c2r_synchronize(c2r_ctx *ctx);

/* evolve0D_global(dt,pos,conv_flag) (files_for_3D/evolve_point.F90:325-440) for the ONE cell at 1-based mesh
 * position pos, as the reference's global_pass calls it cell by cell (evolve.F90:477-484): applies the
 * collected rates, adds 1 to *conv_flag when the cell has not converged.  c2r_global_pass is the whole pass. */
int c2r_evolve0d_global(c2r_ctx *ctx, double dt, const int pos[3], int *conv_flag);

/* minval(xh(:,:,:,0)), minval(xhe(:,:,:,0)) of the grids `which` selects as in c2r_fraction_means: the
 * "min xh_av / min xhe_av" log lines of global_pass (evolve.F90:463-466) want which = 2. */
int c2r_fraction_minima(c2r_ctx *ctx, int which, double out2[2]);

/* The numerical and algorithmic parameters that are COMPILED INTO the device code, for a host to compare with
 * the modules it was built with (c2ray_parameters.f90, abundances.f90, radiation_sizes.f90) before the first
 * step -- linking, say, c2ray_parameters_TEST4.f90 (subboxsize = mesh(1)) must stop the run, not silently
 * trace other sub-boxes.  out[0..]: subboxsize, max_subbox, abu_he, abu_c, epsilon, convergence_fraction,
 * minimum_fractional_change, minimum_fraction_of_atoms, relative_denergy, minitemp, NumTau, minlogtau,
 * maxlogtau, NumFreqBnd, NumheatBin.  Returns how many there are (15). */
int c2r_get_constants(double *out, int capacity);

/* Most sources swept by one batch of launches (1..4096; default 256).  Their column blocks come out of a
 * scratch arena (6 shell-ordered arrays per source, sized by the sub-boxes the source needed in the last
 * pass); a batch is cut short when the arena cannot hold it. */
int c2r_set_batch(c2r_ctx *ctx, int nbatch);

/* ---- several GPUs: sources over ranks and the sum over ranks ----------------------------------------
 * The reference's MPI strategy (master_slave.F90:74-96 do_grid_static, evolve.F90:505-548
 * mpi_accumulate_grid_quantities): every rank holds the full grid, rank r sweeps sources r+1, r+1+npr, ...,
 * six MPI_ALLREDUCE calls sum phih_grid, phihe_grid, phiheat, photon_loss and sum_nbox, every rank runs the
 * global pass.  Here a rank is a GPU and the six sums are one fp64 ncclAllReduce (RCCL over xGMI) of the
 * contiguous buffer above, device to device.
 *
 * One process per GPU (an MPI rank, a torch.distributed.run rank): c2r_create, then c2r_comm_init with the
 * 128-byte id that ONE rank obtained from c2r_comm_unique_id and the launcher passed to the others (MPI_Bcast,
 * a file, ...).  One process for several GPUs (the reference's no_mpi build): c2r_create_multi returns a
 * context that applies every state-setting call to all its devices, runs passes on one host thread per device
 * and reads results from the first; c2r_comm_init_local gives it its communicators (ncclCommInitAll).  The two
 * compose: c2r_comm_init on a multi-device context makes its devices ranks first_rank, first_rank+1, ...
 * RCCL is loaded on first use; single-GPU runs never touch it. */
int c2r_device_count(void);
int c2r_create_multi(c2r_ctx **out, int ndev, const int *devices, const int mesh[3]);
int c2r_num_devices(const c2r_ctx *ctx);
/* 0 when librccl can be loaded and reports the major version this library was compiled against (what
 * c2r_comm_unique_id / c2r_comm_init need); otherwise non-zero with the reason in c2r_create_error().  Lets every
 * rank of a launcher agree BEFORE the collective ncclCommInitRank whether the communicator can be made at all. */
int c2r_comm_available(void);
int c2r_comm_unique_id(char id[128]);
int c2r_comm_init(c2r_ctx *ctx, int first_rank, int nranks, const char id[128]);
int c2r_comm_init_local(c2r_ctx *ctx);
int c2r_comm_destroy(c2r_ctx *ctx);
int c2r_comm_rank(const c2r_ctx *ctx);
int c2r_comm_nranks(const c2r_ctx *ctx);
/* What carries the sum over ranks (MPI_ALLREDUCE in evolve.F90:505-548): 0 nothing (one rank, no communicator),
 * 1 RCCL (ncclAllReduce), 2 the in-process sum of replicas that share a device (rehearsal mode of
 * c2r_comm_init_local).  A harness reports c2r_comm_nranks as "ranks RCCL saw" only when this is 1. */
int c2r_comm_kind(const c2r_ctx *ctx);
/* Path of the library whose ncclAllReduce carries the sums (librccl.so.1 of the loader's path, or the file named by
 * the environment variable C2R_RCCL_LIBRARY), "" when none could be loaded (non-zero return, reason in
 * c2r_create_error()).  A harness prints it next to its result: a sum carried by anything but RCCL is not an RCCL result. */
int c2r_comm_library(char *out, int capacity);
/* Errors while a communicator is in use (the reference has none: its ranks log, go on and meet again at the next
 * MPI_ALLREDUCE, evolve.F90:177-181,523-538; its drop-in host ends the job with MPI_ABORT).  c2r_allreduce_rates,
 * c2r_pass_allreduce_chemistry, c2r_iteration and c2r_evolve3d ABORT the context's RCCL communicators (ncclCommAbort)
 * before they return an error: nothing of this process is left waiting in a sum that cannot complete, every later
 * collective call on the context fails at once, and the host is expected to exit non-zero so that its launcher ends the
 * other ranks.  A rank whose peer never issues its share of a sum does not wait for ever either: after
 * C2R_COMM_TIMEOUT_S seconds (environment; default 1800, 0 = no limit) its wait ends with an error and the same abort. */
/* mpi_accumulate_grid_quantities (evolve.F90:505-548) after c2r_pass_sources: the whole buffer in one
 * all-reduce; afterwards c2r_get_loss / c2r_download_rates return the summed photon_loss and sum_nbox_all.
 * A no-op on a single rank without communicator.  On a multi-device context c2r_pass_sources(first, stride)
 * gives device i the sources first + i*stride, first + i*stride + stride*ndev, ... */
int c2r_allreduce_rates(c2r_ctx *ctx);
/* One outer iteration's pass_all_sources + mpi_accumulate_grid_quantities + global_pass
 * (evolve.F90:185-217) with all three overlapped: the rates launch of the last batch is cut into nslab slabs
 * of k-planes, slab s is summed over the ranks while slab s+1 is computed, and its chemistry runs as soon as
 * its sum is complete.  conv_flag: non-converged cells (evolve.F90:488). */
int c2r_pass_allreduce_chemistry(c2r_ctx *ctx, int first, int stride, int nslab, double dt, int *conv_flag);

/* evolve0D(dt,rtpos,ns,niter) (files_for_3D/evolve_point.F90:79-319) for ONE cell, for hosts that drive the sweep
 * themselves, cell by cell, as the reference's do_source does through evolve2D / evolve1D_axis / evolve2D_plane /
 * evolve3D_quadrant (files_for_3D/evolve_source.F90:244-608): the incoming columns of the cell at mesh position rtpos
 * (1-based, not wrapped: rtpos - srcpos(:,ns) is the offset from the source) by short characteristics from the cells
 * of source ns done before it, its own columns, its photo-ionisation (and heating) rates added to the rate grids.
 * The caller keeps the order of the reference's sweeps (a cell after the cells it interpolates from) and the
 * reference's "already done" test (coldensh_out(pos) == 0): every cell of a source is given once.  A new (ns, niter)
 * pair starts a new source.  on_surface != 0: the cell lies on the surface of the caller's current sub-box
 * (evolve_point.F90:310-315); *loss then receives phi%photo_out * vol / vol_ph, for which the call waits for the
 * device -- all other calls only queue work.  One launch per cell: an interface for the reference's own loops, tests
 * and small meshes; c2r_do_source / c2r_pass_sources trace a source as a whole. */
int c2r_evolve0d(c2r_ctx *ctx, const int rtpos[3], int ns, int niter, int on_surface, double *loss);

/* One outer iteration of evolve3D after set_rates_to_zero (files_for_3D/evolve.F90:185-217): pass_all_sources for the
 * sources first, first + stride, ... (:385-431), mpi_accumulate_grid_quantities (:505-548; a no-op without a
 * communicator) and global_pass (:435-501) -- c2r_pass_allreduce_chemistry -- followed, in the same queue and
 * with ONE host synchronisation for all of it, by every grid reduction the reference's loop prints or feeds to
 * calculate_photon_statistics after a global pass (:463-466, :487-499; photonstatistics.f90:117-234).  The numbers
 * are those of the single-purpose entry points (same kernels' summation order):
 *   means_intermed  = c2r_fraction_means(ctx, 1, .)      sums_intermed = c2r_state_sums(ctx, 1, .)
 *   total_rates     = c2r_total_rates(ctx, dt, reccoef, .) with reccoef = c2r_get_reccoef(ctx, .)
 *   minima_av       = c2r_fraction_minima(ctx, 2, .) as the NEXT iteration's log lines "min xh_av" / "min xhe_av"
 *                     will want them (xh_av does not change between a global pass and the next one)
 *   photon_loss, sum_nbox = c2r_get_loss (summed over the ranks).
 * A host driver that follows the reference's loop line by line makes six calls with six synchronisations for this;
 * the Fortran drop-in (fortran/evolve.F90) uses this one unless an iteration dump is due. */
typedef struct {
  int conv_flag;
  int sum_nbox;
  double photon_loss[C2R_NFREQ];
  double means_intermed[5];
  double sums_intermed[5];
  double total_rates[3];
  double minima_av[2];
  double reccoef[12];
} c2r_iteration_report;
int c2r_iteration(c2r_ctx *ctx, int first, int stride, int nslab, double dt, c2r_iteration_report *report);

/* Timing of the last c2r_pass_sources / c2r_global_pass on the context's stream, measured with
 * HIP events on that stream: milliseconds spent in the column sweep launches, the rates kernel
 * and the chemistry kernel, and the number of launches of each. */
typedef struct {
  double sweep_ms, rates_ms, chem_ms;
  int sweep_launches, rates_launches, chem_launches;
  long long cells_swept; /* cell x source pairs actually traced by the last c2r_pass_sources */
} c2r_timing;
int c2r_get_timing(c2r_ctx *ctx, c2r_timing *out);
/* The column scratch (the reference's coldensh_out / coldenshe_out per source in flight, evolve_source.F90:94-95) since the
 * context was made: out[0] device segments allocated, out[1] of them INSIDE a pass (the others by c2r_begin_step, which sizes
 * the scratch of a time step from the sub-box counts the step before ended with), out[2] doubles allocated in all, out[3]
 * column blocks moved to deeper ones in the middle of a sweep, out[4] batches that started over for lack of room,
 * out[5] doubles held now. */
int c2r_arena_stats(const c2r_ctx *ctx, long long out[6]);
/* the same for device `idev` (0 .. c2r_num_devices-1) of a context made by c2r_create_multi */
int c2r_get_timing_device(c2r_ctx *ctx, int idev, c2r_timing *out);
int c2r_enable_timing(c2r_ctx *ctx, int on);

#ifdef __cplusplus
}
#endif
#endif
