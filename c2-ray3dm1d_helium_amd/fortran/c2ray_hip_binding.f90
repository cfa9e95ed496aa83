!> iso_c_binding view of include/c2ray_hip.h -- the only place where the Fortran host meets C.
!! Every interface below binds one entry point of the C ABI; argument order and meaning are those
!! of the header.  Arrays are passed by address (assumed-size dummies), scalars by value.
module c2ray_hip

  use, intrinsic :: iso_c_binding

  implicit none

  public

  !> c2r_timing of include/c2ray_hip.h
  type, bind(C) :: c2r_timing
     real(c_double) :: sweep_ms, rates_ms, chem_ms
     integer(c_int) :: sweep_launches, rates_launches, chem_launches
     integer(c_long_long) :: cells_swept
  end type c2r_timing

  !> c2r_iteration_report of include/c2ray_hip.h: what evolve3D's loop reports about one outer iteration
  type, bind(C) :: c2r_iteration_report
     integer(c_int) :: conv_flag
     integer(c_int) :: sum_nbox
     real(c_double) :: photon_loss(47)
     real(c_double) :: means_intermed(5)
     real(c_double) :: sums_intermed(5)
     real(c_double) :: total_rates(3)
     real(c_double) :: minima_av(2)
     real(c_double) :: reccoef(12)
  end type c2r_iteration_report

  !> c2r_sed_setup of include/c2ray_hip.h: what spec_integration starts from for one SED
  type, bind(C) :: c2r_sed_setup
     integer(c_int) :: nfreq, sed
     type(c_ptr) :: freq_min, delta_freq, xsec_index, tau, romw
     real(c_double) :: R_star2, h_over_kT, two_pi_over_c_square, hplanck, pi
     real(c_double) :: ion_freq_HI, ion_freq_HeI, ion_freq_HeII
     real(c_double) :: pl_scaling, pl_index
  end type c2r_sed_setup

  interface

     integer(c_int) function c2r_create(ctx, device, mesh) bind(C, name="c2r_create")
       import :: c_int, c_ptr
       type(c_ptr), intent(out) :: ctx
       integer(c_int), value :: device
       integer(c_int), intent(in) :: mesh(3)
     end function c2r_create

     subroutine c2r_destroy(ctx) bind(C, name="c2r_destroy")
       import :: c_ptr
       type(c_ptr), value :: ctx
     end subroutine c2r_destroy

     type(c_ptr) function c2r_last_error(ctx) bind(C, name="c2r_last_error")
       import :: c_ptr
       type(c_ptr), value :: ctx
     end function c2r_last_error

     type(c_ptr) function c2r_create_error() bind(C, name="c2r_create_error")
       import :: c_ptr
     end function c2r_create_error

     integer(c_int) function c2r_set_tables(ctx, photo_thick, photo_thin, heat_thick, heat_thin, &
          sigma_HI, sigma_HeI, sigma_HeII, fvec, bb_upper) bind(C, name="c2r_set_tables")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       type(c_ptr), value :: photo_thick, photo_thin ! c_loc of the tables, or c_null_ptr (c2r_build_tables)
       type(c_ptr), value :: heat_thick, heat_thin   ! c_loc of the tables, or c_null_ptr
       real(c_double), intent(in) :: sigma_HI(*), sigma_HeI(*), sigma_HeII(*)
       type(c_ptr), intent(in) :: fvec(12)           ! c_loc of the twelve f vectors
       integer(c_int), value :: bb_upper
     end function c2r_set_tables

     integer(c_int) function c2r_set_cooling(ctx, cool, mintemp, dtemp) bind(C, name="c2r_set_cooling")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: cool(*)
       real(c_double), value :: mintemp, dtemp
     end function c2r_set_cooling

     integer(c_int) function c2r_set_step(ctx, ndens, dr, vol, clumping, zred, H0, Omega0, &
          isothermal, temper_val, reccoef) bind(C, name="c2r_set_step")
       import :: c_int, c_ptr, c_double, c_float
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: ndens(*), dr(3)
       real(c_double), value :: vol
       real(c_float), value :: clumping
       real(c_double), value :: zred, H0, Omega0
       integer(c_int), value :: isothermal
       real(c_double), value :: temper_val
       real(c_double), intent(in) :: reccoef(12)
     end function c2r_set_step

     integer(c_int) function c2r_set_step_scalars(ctx, dr, vol, clumping, zred, H0, Omega0, &
          isothermal, temper_val, reccoef) bind(C, name="c2r_set_step_scalars")
       import :: c_int, c_ptr, c_double, c_float
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: dr(3)
       real(c_double), value :: vol
       real(c_float), value :: clumping
       real(c_double), value :: zred, H0, Omega0
       integer(c_int), value :: isothermal
       real(c_double), value :: temper_val
       real(c_double), intent(in) :: reccoef(12)
     end function c2r_set_step_scalars

     integer(c_int) function c2r_arena_stats(ctx, out) bind(C, name="c2r_arena_stats")
       import :: c_int, c_ptr, c_long_long
       type(c_ptr), value :: ctx
       integer(c_long_long), intent(out) :: out(6)
     end function c2r_arena_stats

     integer(c_int) function c2r_scale_ndens(ctx, divisor) bind(C, name="c2r_scale_ndens")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), value :: divisor
     end function c2r_scale_ndens

     integer(c_int) function c2r_set_sources(ctx, nsrc, srcpos, normflux, s_star) &
          bind(C, name="c2r_set_sources")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: nsrc
       integer(c_int), intent(in) :: srcpos(*)
       real(c_double), intent(in) :: normflux(*)
       real(c_double), value :: s_star
     end function c2r_set_sources

     integer(c_int) function c2r_set_sed_tables(ctx, sed, photo_thick, photo_thin, heat_thick, heat_thin, &
          lower, upper) bind(C, name="c2r_set_sed_tables")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: sed
       type(c_ptr), value :: photo_thick, photo_thin ! c_loc of the tables, or c_null_ptr (c2r_build_tables)
       type(c_ptr), value :: heat_thick, heat_thin
       integer(c_int), value :: lower, upper
     end function c2r_set_sed_tables

     integer(c_int) function c2r_set_sources_sed(ctx, sed, normflux, s_star) bind(C, name="c2r_set_sources_sed")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: sed
       real(c_double), intent(in) :: normflux(*)
       real(c_double), value :: s_star
     end function c2r_set_sources_sed

     integer(c_int) function c2r_build_tables(ctx, setup, with_heat) bind(C, name="c2r_build_tables")
       import :: c_int, c_ptr, c2r_sed_setup
       type(c_ptr), value :: ctx
       type(c2r_sed_setup), intent(in) :: setup
       integer(c_int), value :: with_heat
     end function c2r_build_tables

     integer(c_int) function c2r_download_tables(ctx, sed, photo_thick, photo_thin, heat_thick, heat_thin) &
          bind(C, name="c2r_download_tables")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: sed
       type(c_ptr), value :: photo_thick, photo_thin, heat_thick, heat_thin  ! (0:NumTau, ncol) or c_null_ptr
     end function c2r_download_tables

     integer(c_int) function c2r_set_lls(ctx, use_lls, coldensh_lls, lls_grid) bind(C, name="c2r_set_lls")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: use_lls
       real(c_double), value :: coldensh_lls
       type(c_ptr), value :: lls_grid            ! real(c_float) (mesh) or c_null_ptr
     end function c2r_set_lls

     integer(c_int) function c2r_set_clumping_grid(ctx, clumping_grid) bind(C, name="c2r_set_clumping_grid")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       type(c_ptr), value :: clumping_grid       ! real(c_float) (mesh) or c_null_ptr
     end function c2r_set_clumping_grid

     integer(c_int) function c2r_upload_state(ctx, xh, xhe, temperature) bind(C, name="c2r_upload_state")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: xh(*), xhe(*)
       type(c_ptr), value :: temperature            ! c_loc(temperature_grid) or c_null_ptr
     end function c2r_upload_state

     integer(c_int) function c2r_download_state(ctx, xh, xhe, temperature) bind(C, name="c2r_download_state")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: xh(*), xhe(*)
       type(c_ptr), value :: temperature
     end function c2r_download_state

     integer(c_int) function c2r_evolve3d(ctx, dt, niter, conv_flags, cap) bind(C, name="c2r_evolve3d")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), value :: dt
       integer(c_int), intent(out) :: niter
       integer(c_int), intent(out) :: conv_flags(*)
       integer(c_int), value :: cap
     end function c2r_evolve3d

     integer(c_int) function c2r_begin_step(ctx) bind(C, name="c2r_begin_step")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function c2r_begin_step

     integer(c_int) function c2r_set_rates_to_zero(ctx) bind(C, name="c2r_set_rates_to_zero")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function c2r_set_rates_to_zero

     integer(c_int) function c2r_pass_sources(ctx, first, stride) bind(C, name="c2r_pass_sources")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: first, stride
     end function c2r_pass_sources

     integer(c_int) function c2r_pass_sources_begin(ctx, first, stride, nslab) bind(C, name="c2r_pass_sources_begin")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: first, stride, nslab
     end function c2r_pass_sources_begin

     integer(c_int) function c2r_pass_slab_count(ctx) bind(C, name="c2r_pass_slab_count")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function c2r_pass_slab_count

     integer(c_int) function c2r_pass_wait_slab(ctx, slab, first_cell, ncells) bind(C, name="c2r_pass_wait_slab")
       import :: c_int, c_ptr, c_size_t
       type(c_ptr), value :: ctx
       integer(c_int), value :: slab
       integer(c_size_t), intent(out) :: first_cell, ncells
     end function c2r_pass_wait_slab

     integer(c_int) function c2r_pass_sources_end(ctx) bind(C, name="c2r_pass_sources_end")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function c2r_pass_sources_end

     integer(c_int) function c2r_do_source(ctx, ns) bind(C, name="c2r_do_source")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: ns
     end function c2r_do_source

     integer(c_int) function c2r_global_pass_cells(ctx, dt, first_cell, ncells, after_event) &
          bind(C, name="c2r_global_pass_cells")
       import :: c_int, c_ptr, c_double, c_size_t
       type(c_ptr), value :: ctx
       real(c_double), value :: dt
       integer(c_size_t), value :: first_cell, ncells
       type(c_ptr), value :: after_event          ! hipEvent_t or c_null_ptr
     end function c2r_global_pass_cells

     integer(c_int) function c2r_global_pass_finish(ctx, conv_flag) bind(C, name="c2r_global_pass_finish")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), intent(out) :: conv_flag
     end function c2r_global_pass_finish

     integer(c_int) function c2r_global_pass(ctx, dt, conv_flag) bind(C, name="c2r_global_pass")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), value :: dt
       integer(c_int), intent(out) :: conv_flag
     end function c2r_global_pass

     integer(c_int) function c2r_end_step(ctx) bind(C, name="c2r_end_step")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function c2r_end_step

     integer(c_int) function c2r_download_rates(ctx, phih, phihe, phiheat, photon_loss, sum_nbox) &
          bind(C, name="c2r_download_rates")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: phih(*), phihe(*), phiheat(*), photon_loss(*)
       integer(c_int), intent(out) :: sum_nbox
     end function c2r_download_rates

     integer(c_int) function c2r_download_rates_sel(ctx, which, phih, phihe, phiheat, photon_loss, sum_nbox) &
          bind(C, name="c2r_download_rates_sel")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: which
       real(c_double), intent(out) :: phih(*), phihe(*), phiheat(*), photon_loss(*)
       integer(c_int), intent(out) :: sum_nbox
     end function c2r_download_rates_sel

     integer(c_int) function c2r_get_loss(ctx, photon_loss, sum_nbox) bind(C, name="c2r_get_loss")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: photon_loss(*)
       integer(c_int), intent(out) :: sum_nbox
     end function c2r_get_loss

     integer(c_int) function c2r_download_iter_state(ctx, xh_av, xhe_av, xh_intermed, xhe_intermed) &
          bind(C, name="c2r_download_iter_state")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: xh_av(*), xhe_av(*), xh_intermed(*), xhe_intermed(*)
     end function c2r_download_iter_state

     integer(c_int) function c2r_upload_rates(ctx, phih, phihe, phiheat) bind(C, name="c2r_upload_rates")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: phih(*), phihe(*), phiheat(*)
     end function c2r_upload_rates

     integer(c_int) function c2r_upload_iter_state(ctx, xh_av, xhe_av, xh_intermed, xhe_intermed) &
          bind(C, name="c2r_upload_iter_state")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: xh_av(*), xhe_av(*), xh_intermed(*), xhe_intermed(*)
     end function c2r_upload_iter_state

     integer(c_int) function c2r_download_columns(ctx, coldensh_out, coldenshe_out) &
          bind(C, name="c2r_download_columns")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: coldensh_out(*), coldenshe_out(*)
     end function c2r_download_columns

     integer(c_int) function c2r_state_sums(ctx, which, out5) bind(C, name="c2r_state_sums")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: which
       real(c_double), intent(out) :: out5(5)
     end function c2r_state_sums

     integer(c_int) function c2r_fraction_means(ctx, which, out5) bind(C, name="c2r_fraction_means")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: which
       real(c_double), intent(out) :: out5(5)
     end function c2r_fraction_means

     integer(c_int) function c2r_evolve0d_global(ctx, dt, pos, conv_flag) bind(C, name="c2r_evolve0d_global")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), value :: dt
       integer(c_int), intent(in) :: pos(3)
       integer(c_int), intent(inout) :: conv_flag
     end function c2r_evolve0d_global

     integer(c_int) function c2r_fraction_minima(ctx, which, out2) bind(C, name="c2r_fraction_minima")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: which
       real(c_double), intent(out) :: out2(2)
     end function c2r_fraction_minima

     integer(c_int) function c2r_get_constants(out, capacity) bind(C, name="c2r_get_constants")
       import :: c_int, c_double
       real(c_double), intent(out) :: out(*)
       integer(c_int), value :: capacity
     end function c2r_get_constants

     ! ---- several GPUs: sources over ranks and the sum over ranks (RCCL inside the library) ----
     integer(c_int) function c2r_device_count() bind(C, name="c2r_device_count")
       import :: c_int
     end function c2r_device_count

     integer(c_int) function c2r_create_multi(ctx, ndev, devices, mesh) bind(C, name="c2r_create_multi")
       import :: c_int, c_ptr
       type(c_ptr), intent(out) :: ctx
       integer(c_int), value :: ndev
       integer(c_int), intent(in) :: devices(*)
       integer(c_int), intent(in) :: mesh(3)
     end function c2r_create_multi

     integer(c_int) function c2r_num_devices(ctx) bind(C, name="c2r_num_devices")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function c2r_num_devices

     integer(c_int) function c2r_comm_unique_id(id) bind(C, name="c2r_comm_unique_id")
       import :: c_int, c_char
       character(kind=c_char), intent(out) :: id(128)
     end function c2r_comm_unique_id

     integer(c_int) function c2r_comm_init(ctx, first_rank, nranks, id) bind(C, name="c2r_comm_init")
       import :: c_int, c_ptr, c_char
       type(c_ptr), value :: ctx
       integer(c_int), value :: first_rank, nranks
       character(kind=c_char), intent(in) :: id(128)
     end function c2r_comm_init

     integer(c_int) function c2r_comm_init_local(ctx) bind(C, name="c2r_comm_init_local")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function c2r_comm_init_local

     integer(c_int) function c2r_comm_destroy(ctx) bind(C, name="c2r_comm_destroy")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function c2r_comm_destroy

     integer(c_int) function c2r_comm_nranks(ctx) bind(C, name="c2r_comm_nranks")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function c2r_comm_nranks

     integer(c_int) function c2r_comm_available() bind(C, name="c2r_comm_available")
       import :: c_int
     end function c2r_comm_available

     integer(c_int) function c2r_comm_rank(ctx) bind(C, name="c2r_comm_rank")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function c2r_comm_rank

     integer(c_int) function c2r_comm_kind(ctx) bind(C, name="c2r_comm_kind")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function c2r_comm_kind

     integer(c_int) function c2r_comm_library(out, capacity) bind(C, name="c2r_comm_library")
       import :: c_int, c_char
       character(kind=c_char), dimension(*) :: out
       integer(c_int), value :: capacity
     end function c2r_comm_library

     integer(c_int) function c2r_allreduce_rates(ctx) bind(C, name="c2r_allreduce_rates")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function c2r_allreduce_rates

     integer(c_int) function c2r_pass_allreduce_chemistry(ctx, first, stride, nslab, dt, conv_flag) &
          bind(C, name="c2r_pass_allreduce_chemistry")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       integer(c_int), value :: first, stride, nslab
       real(c_double), value :: dt
       integer(c_int), intent(out) :: conv_flag
     end function c2r_pass_allreduce_chemistry

     integer(c_int) function c2r_evolve0d(ctx, rtpos, ns, niter, on_surface, loss) bind(C, name="c2r_evolve0d")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       integer(c_int), intent(in) :: rtpos(3)
       integer(c_int), value :: ns, niter, on_surface
       real(c_double), intent(out) :: loss
     end function c2r_evolve0d

     integer(c_int) function c2r_iteration(ctx, first, stride, nslab, dt, report) bind(C, name="c2r_iteration")
       import :: c_int, c_ptr, c_double, c2r_iteration_report
       type(c_ptr), value :: ctx
       integer(c_int), value :: first, stride, nslab
       real(c_double), value :: dt
       type(c2r_iteration_report), intent(out) :: report
     end function c2r_iteration

     integer(c_int) function c2r_total_rates(ctx, dt, reccoef, out3) bind(C, name="c2r_total_rates")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), value :: dt
       real(c_double), intent(in) :: reccoef(12)
       real(c_double), intent(out) :: out3(3)
     end function c2r_total_rates

     integer(c_int) function c2r_get_reccoef(ctx, out12) bind(C, name="c2r_get_reccoef")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), intent(out) :: out12(12)
     end function c2r_get_reccoef

     integer(c_size_t) function c2r_rates_count(ctx) bind(C, name="c2r_rates_count")
       import :: c_size_t, c_ptr
       type(c_ptr), value :: ctx
     end function c2r_rates_count

     type(c_ptr) function c2r_rates_device_ptr(ctx) bind(C, name="c2r_rates_device_ptr")
       import :: c_ptr
       type(c_ptr), value :: ctx
     end function c2r_rates_device_ptr

     integer(c_int) function c2r_set_rates_buffer(ctx, device_ptr, count) bind(C, name="c2r_set_rates_buffer")
       import :: c_int, c_ptr, c_size_t
       type(c_ptr), value :: ctx, device_ptr
       integer(c_size_t), value :: count
     end function c2r_set_rates_buffer

     integer(c_int) function c2r_synchronize(ctx) bind(C, name="c2r_synchronize")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function c2r_synchronize

     integer(c_int) function c2r_set_batch(ctx, nbatch) bind(C, name="c2r_set_batch")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: nbatch
     end function c2r_set_batch

     integer(c_int) function c2r_enable_timing(ctx, on) bind(C, name="c2r_enable_timing")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: on
     end function c2r_enable_timing

     integer(c_int) function c2r_get_timing_device(ctx, idev, tm) bind(C, name="c2r_get_timing_device")
       import :: c_int, c_ptr, c2r_timing
       type(c_ptr), value :: ctx
       integer(c_int), value :: idev
       type(c2r_timing), intent(out) :: tm
     end function c2r_get_timing_device

     integer(c_int) function c2r_get_timing(ctx, tm) bind(C, name="c2r_get_timing")
       import :: c_int, c_ptr, c2r_timing
       type(c_ptr), value :: ctx
       type(c2r_timing), intent(out) :: tm
     end function c2r_get_timing

  end interface

contains

  !> Text of the last error of a context (or of a failed c2r_create when ctx is null)
  function c2r_error_text(ctx) result(text)
    type(c_ptr), intent(in) :: ctx
    character(len=:), allocatable :: text
    type(c_ptr) :: p
    character(kind=c_char), pointer :: s(:)
    integer :: n

    if (c_associated(ctx)) then
       p = c2r_last_error(ctx)
    else
       p = c2r_create_error()
    endif
    text = ""
    if (.not. c_associated(p)) return
    call c_f_pointer(p, s, (/ 512 /))
    n = 0
    do while (n < 512)
       if (s(n+1) == c_null_char) exit
       n = n + 1
    enddo
    allocate(character(len=n) :: text)
    if (n > 0) text = transfer(s(1:n), text)
  end function c2r_error_text

end module c2ray_hip
