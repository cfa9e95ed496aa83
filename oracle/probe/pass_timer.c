/* TEST / MEASUREMENT INFRASTRUCTURE ONLY -- a clock around the reference's own pass_all_sources.
 *
 * Linked into the reference build with -Wl,--wrap=_QMmaster_slave_processingPdo_grid (oracle/ref_build.sh N omp timer):
 * evolve3D's call of do_grid (files_for_3D/evolve.F90:409, master_slave.F90:53) goes through this wrapper, which calls
 * the unmodified routine and prints, unbuffered, when each pass started and how long it took.  The time from one entry
 * to the next is one outer iteration of evolve3D (pass_all_sources + global_pass + the loop's bookkeeping).  The
 * reference's own Timings.log has the same stamps in tenths of a second but is only flushed when the run ends, which at
 * 256^3 is half an hour away; with C2R_REF_STOP_AFTER=n the run ends at the entry of pass n+1.
 * Nothing of the reference is modified or copied; the product never links this file. */
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

void __real__QMmaster_slave_processingPdo_grid(double *dt, int *niter);

/* Round 5: the non-converged count of every global pass as well, printed when the next pass begins.  global_pass itself is
 * called from within its own object (no undefined reference for the linker to wrap), but the per-cell routine it calls for
 * every cell, evolve0D_global(dt,pos,conv_flag) of module evolve_point (files_for_3D/evolve.F90:477-484,
 * evolve_point.F90:325), is reached across objects: with -Wl,--wrap=_QMevolve_pointPevolve0d_global its running count is seen
 * after every call; the value after the last cell is the count of the pass. */
void __real__QMevolve_pointPevolve0d_global(double *dt, int *pos, int *conv_flag);
static int last_conv_flag = -1;
/* ... and the sub-boxes all sources of the pass needed (module variable sum_nbox of evolve_source, evolve_source.F90:236) */
extern int _QMevolve_sourceEsum_nbox;
void __wrap__QMevolve_pointPevolve0d_global(double *dt, int *pos, int *conv_flag) {
  __real__QMevolve_pointPevolve0d_global(dt, pos, conv_flag);
  last_conv_flag = *conv_flag;
}

static double now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void __wrap__QMmaster_slave_processingPdo_grid(double *dt, int *niter) {
  static double last_entry = 0.0;
  static int passes = 0;
  const double t0 = now();
  if (passes > 0) fprintf(stderr, "pass_timer: iteration %d took %.3f s (entry to entry), non-converged cells %d\n", passes, t0 - last_entry, last_conv_flag);
  const char *stop = getenv("C2R_REF_STOP_AFTER");
  if (stop && passes >= atoi(stop)) {
    fprintf(stderr, "pass_timer: stopping after %d iterations (C2R_REF_STOP_AFTER)\n", passes);
    fflush(stderr);
    exit(0);
  }
  last_entry = t0;
  __real__QMmaster_slave_processingPdo_grid(dt, niter);
  passes++;
  fprintf(stderr, "pass_timer: pass %d (niter %d) took %.3f s, sum_nbox %d\n", passes, *niter, now() - t0, _QMevolve_sourceEsum_nbox);
  fflush(stderr);
}
