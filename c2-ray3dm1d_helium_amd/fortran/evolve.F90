!> Drop-in replacement of the reference's module evolve (files_for_3D/evolve.F90): same module
!! name, same public symbol `evolve3D(time,dt,restart)` with the same meaning.  The convergence loop
!! {zero rates -> all sources -> (sum over ranks) -> global chemistry pass} is driven from here, as in
!! the reference, but every grid operation runs on the GPU through the C ABI of
!! include/c2ray_hip.h (module c2ray_hip).  Everything around it -- the driver, source lists,
!! material set-up, cosmology, output routines -- is the reference's own code, untouched: this file
!! only marshals the module data that evolve3D reads and writes (SURVEY.md section 8b).
!!
!! Kept from the reference's evolve3D: the log lines ("Number of non-converged points", mean
!! fractions, "Average number of subboxes"), the Timings.log stamps, the photon statistics line per
!! global pass and per call (unit 90), the iteration dumps every 15 minutes of wall time
!! (iterdump1.bin / iterdump2.bin, same record order) and the restart from such a dump.
module evolve

  use precision, only: dp
  use clocks, only: timestamp_wallclock
  use file_admin, only: logf, timefile, iterdump, dump_dir
  use my_mpi ! rank, npr, MPI_COMM_NEW
  use sizes, only: Ndim, mesh
  use grid, only: dr, vol
  use material, only: ndens, xh, xhe, temperature_grid, isothermal, temper_val, clumping
  use sourceprops, only: NumSrc, srcpos, NormFlux
  use radiation_sed_parameters, only: S_star
  use radiation_sizes, only: NumFreqBnd, sigma_HI, sigma_HeI, sigma_HeII
  use radiation_sizes, only: f1ion_HI, f1ion_HeI, f1ion_HeII, f2ion_HI, f2ion_HeI, f2ion_HeII
  use radiation_sizes, only: f1heat_HI, f1heat_HeI, f1heat_HeII, f2heat_HI, f2heat_HeI, f2heat_HeII
  use radiation_tables, only: bb_photo_thick_table, bb_photo_thin_table
  use radiation_tables, only: bb_heat_thick_table, bb_heat_thin_table, bb_FreqBnd_UpperLimit
#ifdef PL
  use sourceprops, only: NormFluxPL
  use radiation_sed_parameters, only: pl_S_star
  use radiation_tables, only: pl_photo_thick_table, pl_photo_thin_table
  use radiation_tables, only: pl_heat_thick_table, pl_heat_thin_table
  use radiation_tables, only: pl_FreqBnd_LowerLimit, pl_FreqBnd_UpperLimit
#endif
#ifdef QUASARS
  use sourceprops, only: NormFluxQPL
  use radiation_sed_parameters, only: qpl_S_star
  use radiation_tables, only: qpl_photo_thick_table, qpl_photo_thin_table
  use radiation_tables, only: qpl_heat_thick_table, qpl_heat_thin_table
  use radiation_tables, only: qpl_FreqBnd_LowerLimit, qpl_FreqBnd_UpperLimit
#endif
  use cosmology, only: zred
  use cosmology_parameters, only: H0, Omega0
  use c2ray_parameters, only: convergence_fraction
  use c2ray_parameters, only: use_LLS, type_of_LLS, type_of_clumping
  use material, only: coldensh_LLS, LLS_point, clumping_point
  use cgsconstants, only: arech0, brech0, areche0, breche0, oreche0, areche1, breche1, treche1
  use cgsconstants, only: colli_HI, colli_HeI, colli_HeII, v
  use photonstatistics, only: photon_loss, LLS_loss
  use photonstatistics, only: totrec, totcollisions, recomions, dh0, dhe0, dhe2, total_ion
  use photonstatistics, only: report_photonstatistics, update_grandtotal_photonstatistics
  use evolve_data, only: phih_grid, phihe_grid, phiheat
  use evolve_data, only: xh_av, xhe_av, xh_intermed, xhe_intermed
  use evolve_data, only: photon_loss_all, hip_ctx
  use evolve_source, only: sum_nbox, sum_nbox_all
#ifdef C2RAY_REFERENCE_DO_GRID
  ! the reference's own master_slave.F90, unmodified: its do_source calls land in our evolve_source
  use master_slave_processing, only: do_grid
#endif
#ifdef C2RAY_GLOBAL_PASS_BY_CELL
  use evolve_point, only: evolve0D_global
#endif
  use, intrinsic :: iso_c_binding
  use c2ray_hip

  implicit none

  save

  private

  public :: evolve3D

  logical :: tables_uploaded = .false.
  !> number of H0, H+, He0, He+, He++ at the start of the time step (state_before)
  real(kind=dp) :: n_before(5)
  !> alternating dump file counter
  integer :: ndump = 0
  !> One outer iteration as one library call (c2r_iteration: pass, sum over the ranks, global pass and all the
  !! grid reductions of the log and of calculate_photon_statistics, overlapped where a communicator exists, one
  !! synchronisation).  C2RAY_HIP_STEPWISE=1 keeps the reference's call-by-call order for every iteration; an
  !! iteration that writes an iteration dump uses it anyway (the dump holds the state between pass and global pass).
  logical :: fused_iterations = .true.
  logical :: fused_settings_read = .false.
  logical :: loop_timing = .false.
  !> evolve_data's phiheat still holds the zeros it was allocated with (see download_rates)
  logical :: phiheat_host_is_zero = .true.
  integer :: allreduce_slabs = 4
  !> C2RAY_HIP_KEEP_STATE (opt-in, INTEGRATION.md): what the device already holds is not sent again, what no routine of
  !! the reference reads on the host is not brought back.  1: xh, xhe, temperature_grid are uploaded only when a sample of
  !! the host arrays differs from what this module downloaded at the end of the previous call; ndens only when it is neither
  !! unchanged nor the previous array divided by cosmo_evol's zfactor**3 (the division is then made on the device copy);
  !! phihe_grid stays on the device (only the iteration dump reads it, and fetches it itself).  2: phih_grid and phiheat
  !! stay there as well -- for runs without output stream 3, their only reader (output.F90:311-379).
  integer :: keep_state = 0
  integer,parameter :: nfp = 8192 !< samples of a fingerprint
  logical :: fp_valid = .false. !< the fingerprints below describe host arrays that equal the device's
  integer(kind=8) :: fp_xh(nfp), fp_xhe(nfp), fp_ndens(nfp)
  integer(kind=4) :: fp_temp(nfp)
  real(kind=dp) :: fp_ndens_value(nfp) !< the sampled densities themselves (for the zfactor**3 test)
  real(kind=dp) :: prev_zred = 0.0_dp, prev_dr(3) = 0.0_dp, prev_vol = 0.0_dp
  logical :: prev_isothermal = .true.
  !> what c2r_iteration reported about the iteration in flight, for global_pass to log
  type(c2r_iteration_report) :: report
  logical :: report_valid = .false.
  !> minima of xh_av(0), xhe_av(0) as left by the last global pass (the next iteration's "min xh_av" lines)
  real(kind=dp) :: minima_next(2)
  logical :: minima_next_valid = .false.

#ifdef MPI
  integer :: mympierror
#endif

contains

  ! ===========================================================================

  !> Evolve the entire grid over a time step dt
  subroutine evolve3D (time,dt,restart)

    real(kind=dp),intent(in) :: time !< time (unused, as in the reference)
    real(kind=dp),intent(in) :: dt !< time step
    integer,intent(in) :: restart !< restart flag

    integer :: niter
    integer :: conv_flag
    integer :: conv_criterion
    integer(kind=8) :: wallclock0, wallclock1, wallclock2, countspersec
    integer(kind=8) :: loopclock1, loopclock2, callclock2
    type(c2r_timing) :: kernel_times
    logical :: dump_due, fused
    integer(c_int) :: iso
    real(kind=dp) :: reccoef(12)
    real(kind=dp) :: n_after(5)
    type(c_ptr) :: tptr

    call system_clock (wallclock1)
    wallclock0 = wallclock1

    if (.not. tables_uploaded) call upload_tables ()
    if (.not. fused_settings_read) call read_fused_settings ()
    minima_next_valid = .false.
    report_valid = .false.

    ! --- host state that changes between calls (cosmo_evol rescales dr, vol, ndens each step)
    iso = 0
    if (isothermal) iso = 1
    reccoef = (/ arech0, brech0, areche0, breche0, oreche0, areche1, breche1, treche1, &
         colli_HI, colli_HeI, colli_HeII, v /)
    call send_step (iso, reccoef)
    call pass_subgrid_fields ()
    if (NumSrc > 0) then
       call check (c2r_set_sources (hip_ctx, int(NumSrc,c_int), srcpos, NormFlux(1:NumSrc), S_star), &
            "c2r_set_sources")
    else
       call check (c2r_set_sources (hip_ctx, 0_c_int, (/ 0_c_int /), (/ 0.0_dp /), S_star), &
            "c2r_set_sources")
    endif
#ifdef PL
    if (NumSrc > 0) call check (c2r_set_sources_sed (hip_ctx, 1_c_int, NormFluxPL(1:NumSrc), pl_S_star), &
         "c2r_set_sources_sed")
#endif
#ifdef QUASARS
    if (NumSrc > 0) call check (c2r_set_sources_sed (hip_ctx, 2_c_int, NormFluxQPL(1:NumSrc), qpl_S_star), &
         "c2r_set_sources_sed")
#endif
    tptr = c_null_ptr
    if (.not.isothermal) tptr = c_loc_real4 (temperature_grid)
    if (state_is_on_device (restart)) then
       if (rank == 0) write(logf,*) "evolve3D: xh, xhe, temperature_grid kept on the device (C2RAY_HIP_KEEP_STATE)"
    else
       call check (c2r_upload_state (hip_ctx, xh, xhe, tptr), "c2r_upload_state")
    endif

    ! Initial state (for photon statistics): state_before (photonstatistics.f90:117-144)
    call check (c2r_state_sums (hip_ctx, 0_c_int, n_before), "c2r_state_sums")

    if (restart == 0) then
       ! xh_av = xh_intermed = xh, xhe_av = xhe_intermed = xhe
       call check (c2r_begin_step (hip_ctx), "c2r_begin_step")
       niter=0
       conv_flag=mesh(1)*mesh(2)*mesh(3)
    else
       call start_from_dump (restart,niter)
       call global_pass (conv_flag,dt)
    endif

    conv_criterion=min(int(convergence_fraction*mesh(1)*mesh(2)*mesh(3)),NumSrc)

    if (rank == 0) write(timefile,"(A,F8.1)") &
         "Time before starting iteration: ", timestamp_wallclock ()
    call system_clock (loopclock1,countspersec)

    do
       if (conv_flag < conv_criterion .and. niter > 1) then
          ! xh = xh_intermed, xhe = xhe_intermed, set_final_temperature_point
          call check (c2r_end_step (hip_ctx), "c2r_end_step")
          if (rank == 0) then
             write(logf,*) "Multiple sources convergence reached"
             write(logf,*) "Test 1 values: ",conv_flag, conv_criterion
          endif
          exit
       else
          if (niter > 500) then
             if (rank == 0) write(logf,*) 'Multiple sources not converging'
             exit
          endif
       endif

       niter=niter+1

       call check (c2r_set_rates_to_zero (hip_ctx), "c2r_set_rates_to_zero")
       photon_loss(:)=0.0
       LLS_loss = 0.0

       if (NumSrc > 0) then
          ! Is an iteration dump due (every 15 minutes of wall time, evolve.F90:193-203)?  Asked before the pass,
          ! not after it: the dump holds the state between pass and global pass, so such an iteration goes call by call.
          call system_clock (wallclock2,countspersec)
          dump_due = wallclock2-wallclock1 > 15*60*countspersec .or. wallclock2-wallclock1 < 0
#ifdef MPI
          call MPI_BCAST (dump_due,1,MPI_LOGICAL,0,MPI_COMM_NEW,mympierror)
#endif
          fused = fused_iterations .and. .not.dump_due
          call pass_all_sources (niter,dt,fused)

          if (rank == 0) then
             write(logf,*) "Average number of subboxes: ", real(sum_nbox_all)/real(NumSrc)
             call system_clock (wallclock2,countspersec)
             write(logf,*) "Time and limit are: ", wallclock2-wallclock1, 15.0*60.0*countspersec
             if (dump_due) then
                call write_iteration_dump (niter)
                wallclock1=wallclock2
             endif
          endif
       endif

       call global_pass (conv_flag,dt)

       if (rank == 0) write(timefile,"(A,I3,A,F8.1)") &
            "Time after iteration ",niter," : ", timestamp_wallclock ()
       if (rank == 0 .and. loop_timing) then
          call check (c2r_get_timing (hip_ctx, kernel_times), "c2r_get_timing")
          write(timefile,"(A,I4,3(A,F9.3),A)") "evolve3D kernels, iteration ",niter,": sweep ",kernel_times%sweep_ms, &
               " ms, rates ",kernel_times%rates_ms," ms, chemistry ",kernel_times%chem_ms," ms"
       endif
    enddo

    ! C2RAY_HIP_TIMING=1: the loop's wall time with the clock's own resolution (Timings.log has tenths of a second)
    call system_clock (loopclock2)
    if (rank == 0 .and. loop_timing) write(timefile,"(A,I5,A,F12.6,A)") "evolve3D loop: ",niter, &
         " iterations in ", real(loopclock2-loopclock1,dp)/real(countspersec,dp), " s"

    ! --- results back to the modules that own them
    call check (c2r_download_state (hip_ctx, xh, xhe, tptr), "c2r_download_state")
    call download_rates ()
    if (keep_state > 0) call remember_host_state ()

    ! Calculate photon statistics: calculate_photon_statistics (dt,xh,xh_av,xhe,xhe_av)
    call check (c2r_state_sums (hip_ctx, 0_c_int, n_after), "c2r_state_sums")
    call photon_statistics (dt,n_after)
    call report_photonstatistics (dt)
    call update_grandtotal_photonstatistics (dt)

    ! ... and where a call's time goes besides the loop: host arrays to the device before it, results back after it
    call system_clock (callclock2)
    if (rank == 0 .and. loop_timing) write(timefile,"(A,3(F10.6,A))") "evolve3D call: set-up ", &
         real(loopclock1-wallclock0,dp)/real(countspersec,dp), " s, loop ", &
         real(loopclock2-loopclock1,dp)/real(countspersec,dp), " s, results ", &
         real(callclock2-loopclock2,dp)/real(countspersec,dp), " s"

  end subroutine evolve3D

  ! ===========================================================================

  !> c2r_set_step: the scalars of the step and material's ndens -- which, with C2RAY_HIP_KEEP_STATE, is sent only when the
  !! device's copy cannot be brought to the same bits otherwise
  subroutine send_step (iso, reccoef)

    integer(c_int),intent(in) :: iso
    real(kind=dp),intent(in) :: reccoef(12)

    integer(kind=8) :: fp_now(nfp)
    real(kind=dp) :: now_value(nfp), zfactor, zfactor3
    integer(kind=8) :: n
    logical :: same, scaled
    integer :: k

    n = int(mesh(1),8)*int(mesh(2),8)*int(mesh(3),8)
    same = .false.
    scaled = .false.
    if (keep_state > 0 .and. fp_valid) then
       call sample8 (ndens, n, fp_now, now_value)
       same = all(fp_now == fp_ndens)
       if (.not.same .and. prev_zred > 0.0_dp) then
          ! redshift_evol's zfactor (cosmology.f90:149, a private variable) from the same operands, cosmo_evol's
          ! zfactor3 (:177), and the three things it did with them (:184-193) -- all to the bit, or the array is sent
          zfactor=(1.0+prev_zred)/(1.+zred)
          zfactor3=zfactor*zfactor*zfactor
          scaled = all(transfer(prev_dr(:)*zfactor,1_8,3) == transfer(dr(:),1_8,3)) .and. &
               transfer(prev_vol*zfactor3,1_8) == transfer(vol,1_8)
          if (scaled) then
             do k=1,nfp
                if (transfer(fp_ndens_value(k)/zfactor3,1_8) /= fp_now(k)) scaled = .false.
             enddo
          endif
       endif
    endif
    if (same .or. scaled) then
       if (scaled) call check (c2r_scale_ndens (hip_ctx, zfactor3), "c2r_scale_ndens")
       call check (c2r_set_step_scalars (hip_ctx, dr, vol, real(clumping,c_float), zred, H0, Omega0, &
            iso, temper_val, reccoef), "c2r_set_step_scalars")
       if (rank == 0 .and. scaled) write(logf,*) "evolve3D: ndens rescaled on the device (C2RAY_HIP_KEEP_STATE)"
       if (rank == 0 .and. same) write(logf,*) "evolve3D: ndens kept on the device (C2RAY_HIP_KEEP_STATE)"
    else
       call check (c2r_set_step (hip_ctx, ndens, dr, vol, real(clumping,c_float), zred, H0, Omega0, &
            iso, temper_val, reccoef), "c2r_set_step")
    endif

  end subroutine send_step

  ! ===========================================================================

  !> Is what the device holds since the end of the previous evolve3D call still what material's arrays hold?
  function state_is_on_device (restart) result(kept)

    integer,intent(in) :: restart
    logical :: kept

    integer(kind=8) :: fp8(nfp)
    integer(kind=4) :: fp4(nfp)
    real(kind=dp) :: values(nfp)
    integer(kind=8) :: n

    kept = .false.
    if (keep_state == 0 .or. .not.fp_valid .or. restart /= 0) return
    if (isothermal .neqv. prev_isothermal) return
    n = int(mesh(1),8)*int(mesh(2),8)*int(mesh(3),8)
    call sample8 (xh, 2*n, fp8, values)
    if (any(fp8 /= fp_xh)) return
    call sample8 (xhe, 3*n, fp8, values)
    if (any(fp8 /= fp_xhe)) return
    if (.not.isothermal) then
       call sample4 (temperature_grid, 3*n, fp4)
       if (any(fp4 /= fp_temp)) return
    endif
    kept = .true.

  end function state_is_on_device

  ! ===========================================================================

  !> after the results of a call have reached the host: what they look like, and the scalars cosmo_evol will change
  subroutine remember_host_state ()

    integer(kind=8) :: n
    real(kind=dp) :: values(nfp)

    n = int(mesh(1),8)*int(mesh(2),8)*int(mesh(3),8)
    call sample8 (xh, 2*n, fp_xh, values)
    call sample8 (xhe, 3*n, fp_xhe, values)
    if (.not.isothermal) call sample4 (temperature_grid, 3*n, fp_temp)
    call sample8 (ndens, n, fp_ndens, fp_ndens_value)
    prev_zred = zred
    prev_dr(:) = dr(:)
    prev_vol = vol
    prev_isothermal = isothermal
    fp_valid = .true.

  end subroutine remember_host_state

  ! ===========================================================================

  !> bit patterns of nfp elements of a real(dp) array taken as flat: evenly spread, the last one included
  subroutine sample8 (a, n, bits, values)

    real(kind=dp),intent(in) :: a(*)
    integer(kind=8),intent(in) :: n
    integer(kind=8),intent(out) :: bits(nfp)
    real(kind=dp),intent(out) :: values(nfp)

    integer :: k
    integer(kind=8) :: idx

    do k=1,nfp
       idx = 1_8 + ((n-1_8)*int(k-1,8))/int(nfp-1,8)
       values(k) = a(idx)
       bits(k) = transfer(a(idx),1_8)
    enddo

  end subroutine sample8

  subroutine sample4 (a, n, bits)

    real(kind=4),intent(in) :: a(*)
    integer(kind=8),intent(in) :: n
    integer(kind=4),intent(out) :: bits(nfp)

    integer :: k
    integer(kind=8) :: idx

    do k=1,nfp
       idx = 1_8 + ((n-1_8)*int(k-1,8))/int(nfp-1,8)
       bits(k) = transfer(a(idx),1_4)
    enddo

  end subroutine sample4

  ! ===========================================================================

  !> Ray trace the whole grid for all sources of this rank and sum over the ranks
  subroutine pass_subgrid_fields ()

    ! use_LLS (evolve_point.F90:177-180) and type_of_clumping == 5 (evolve_point.F90:483-484).
    ! The grids behind LLS_point / clumping_point are private to module material, so they are read
    ! through those public accessors, which is what the reference's evolve0D / do_chemistry do per cell.
    real(c_float),dimension(:,:,:),allocatable,target,save :: field
    real :: clumping_saved
    integer :: i,j,k

    if (use_LLS) then
       if (type_of_LLS == 2) then
          if (.not.allocated(field)) allocate(field(mesh(1),mesh(2),mesh(3)))
          do k=1,mesh(3)
             do j=1,mesh(2)
                do i=1,mesh(1)
                   call LLS_point (i,j,k)
                   field(i,j,k)=real(coldensh_LLS,c_float)
                enddo
             enddo
          enddo
          call check (c2r_set_lls (hip_ctx, 1_c_int, coldensh_LLS, c_loc(field)), "c2r_set_lls")
       else
          call check (c2r_set_lls (hip_ctx, 1_c_int, coldensh_LLS, c_null_ptr), "c2r_set_lls")
       endif
    endif
    if (type_of_clumping == 5) then
       if (.not.allocated(field)) allocate(field(mesh(1),mesh(2),mesh(3)))
       clumping_saved=clumping
       do k=1,mesh(3)
          do j=1,mesh(2)
             do i=1,mesh(1)
                call clumping_point (i,j,k)
                field(i,j,k)=real(clumping,c_float)
             enddo
          enddo
       enddo
       clumping=clumping_saved
       call check (c2r_set_clumping_grid (hip_ctx, c_loc(field)), "c2r_set_clumping_grid")
    endif

  end subroutine pass_subgrid_fields

  !----------------------------------------------------------------------------

  subroutine pass_all_sources (niter,dt,fused)

    integer,intent(in) :: niter
    real(kind=dp),intent(in) :: dt
    logical,intent(in) :: fused !< the whole iteration in one library call (see fused_iterations)

    integer(c_int) :: nbox
    real(kind=dp) :: tail(NumFreqBnd)

    if (rank == 0) write(logf,*) 'Doing all sources '

#if !defined(C2RAY_REFERENCE_DO_GRID) && !defined(C2RAY_GLOBAL_PASS_BY_CELL)
    if (fused) then
       ! the "min xh_av" lines of this iteration's global pass show the fractions as they are now
       if (.not.minima_next_valid) then
          call check (c2r_fraction_minima (hip_ctx, 2_c_int, minima_next), "c2r_fraction_minima")
          minima_next_valid = .true.
       endif
       ! pass_all_sources, mpi_accumulate_grid_quantities and the global pass with its statistics: the sum over
       ! the ranks overlaps the pass before it and the chemistry after it, slab by slab; global_pass below
       ! only writes the log lines
       call check (c2r_iteration (hip_ctx, int(1+rank,c_int), int(npr,c_int), int(allreduce_slabs,c_int), dt, report), &
            "c2r_iteration")
       report_valid = .true.
       photon_loss_all(:)=report%photon_loss(:)
       sum_nbox=report%sum_nbox
       sum_nbox_all=report%sum_nbox
       return
    endif
#endif

    ! static distribution of the sources over the ranks: ns = 1+rank, NumSrc, npr
    ! (do_grid_static, master_slave.F90:74-96)
#ifdef C2RAY_REFERENCE_DO_GRID
#ifdef C2RAY_REFERENCE_DO_SOURCE
    sum_nbox=0
#endif
    call do_grid (dt,niter)
#else
    call check (c2r_pass_sources (hip_ctx, int(1+rank,c_int), int(npr,c_int)), "c2r_pass_sources")
#endif

    ! mpi_accumulate_grid_quantities (evolve.F90:505-548): one RCCL all-reduce of the rate grids, photon_loss
    ! and sum_nbox, device to device, over the GPUs of all ranks (nothing to do for one GPU); afterwards
    ! c2r_get_loss returns the sums
    call check (c2r_allreduce_rates (hip_ctx), "c2r_allreduce_rates")
#ifdef C2RAY_REFERENCE_DO_SOURCE
    ! the reference's own do_source (on this library's evolve0D) has kept the books on the host
    ! (evolve_source.F90:233-236), as in the reference's pass_all_sources without MPI (evolve.F90:423-426)
    photon_loss_all(:)=photon_loss(:)
    sum_nbox_all=sum_nbox
#else
    call check (c2r_get_loss (hip_ctx, tail, nbox), "c2r_get_loss")
    photon_loss_all(:)=tail(:)
    sum_nbox=nbox
    sum_nbox_all=nbox
#endif

  end subroutine pass_all_sources

  ! ===========================================================================

  !> Apply the rates: one chemistry pass over the whole grid, report on convergence
  subroutine global_pass (conv_flag,dt)

    integer,intent(out) :: conv_flag
    real(kind=dp),intent(in) :: dt

    integer(c_int) :: cf
    real(kind=dp) :: means(5), n_now(5), minima(2)
#ifdef C2RAY_GLOBAL_PASS_BY_CELL
    integer :: i,j,k
#endif

    ! mean photon loss per cell (evolve.F90:457)
    photon_loss(:)=photon_loss_all(:)/(real(mesh(1))*real(mesh(2))*real(mesh(3)))

    ! Report minimum value of xh_av(0) to check for zeros (evolve.F90:463-466): of the fractions the global
    ! pass is about to replace, i.e. as the previous global pass (or the start of the step) left them
    if (rank == 0) then
       if (minima_next_valid) then
          minima=minima_next
       else
          call check (c2r_fraction_minima (hip_ctx, 2_c_int, minima), "c2r_fraction_minima")
       endif
       write(logf,*) "min xh_av: ",minima(1)
       write(logf,*) "min xhe_av: ",minima(2)
    endif
    minima_next_valid = .false.

    if (rank == 0) write(logf,*) 'Doing global '
    if (report_valid) then
       ! the iteration has been done (c2r_iteration in pass_all_sources): its numbers, in the reference's order
       report_valid = .false.
       conv_flag=report%conv_flag
       if (rank == 0) then
          write(logf,*) "Number of non-converged points: ",conv_flag
          write(logf,*) "Intermediate result for mean H ionization fraction: ", report%means_intermed(2)
          write(logf,*) "Intermediate result for mean He(+,++) ionization fraction: ", report%means_intermed(4), &
               report%means_intermed(5)
       endif
       minima_next=report%minima_av
       minima_next_valid = .true.
       call photon_statistics_from (report%reccoef,report%total_rates,report%sums_intermed)
       call report_photonstatistics (dt)
       return
    endif
#ifdef C2RAY_GLOBAL_PASS_BY_CELL
    ! the reference's own loop (evolve.F90:477-484) over the per-cell interface of module evolve_point
    conv_flag=0
    do k=1,mesh(3)
       do j=1,mesh(2)
          do i=1,mesh(1)
             call evolve0D_global (dt,(/ i,j,k /),conv_flag)
          enddo
       enddo
    enddo
#else
    call check (c2r_global_pass (hip_ctx, dt, cf), "c2r_global_pass")
    conv_flag=cf
#endif

    if (rank == 0) then
       call check (c2r_fraction_means (hip_ctx, 1_c_int, means), "c2r_fraction_means")
       write(logf,*) "Number of non-converged points: ",conv_flag
       write(logf,*) "Intermediate result for mean H ionization fraction: ", means(2)
       write(logf,*) "Intermediate result for mean He(+,++) ionization fraction: ", means(4), means(5)
    endif

    ! photon conservation: calculate_photon_statistics (dt,xh_intermed,xh_av,xhe_intermed,xhe_av)
    call check (c2r_state_sums (hip_ctx, 1_c_int, n_now), "c2r_state_sums")
    call photon_statistics (dt,n_now)
    call report_photonstatistics (dt)

  end subroutine global_pass

  ! ===========================================================================

  !> total_rates + total_ionizations of photonstatistics.f90 with the grid sums from the device
  subroutine photon_statistics (dt,n_after)

    real(kind=dp),intent(in) :: dt
    real(kind=dp),intent(in) :: n_after(5)

    real(kind=dp) :: reccoef(12), rates3(3)

    ! The reference's global pass leaves the temperature-dependent coefficients of the last cell in the
    ! module variables of cgsconstants (evolve_point.F90:543); total_rates uses them
    call check (c2r_get_reccoef (hip_ctx, reccoef), "c2r_get_reccoef")
    if (.not.isothermal) then
       arech0=reccoef(1); brech0=reccoef(2); areche0=reccoef(3); breche0=reccoef(4)
       oreche0=reccoef(5); areche1=reccoef(6); breche1=reccoef(7); treche1=reccoef(8)
       colli_HI=reccoef(9); colli_HeI=reccoef(10); colli_HeII=reccoef(11); v=reccoef(12)
    endif
    call check (c2r_total_rates (hip_ctx, dt, reccoef, rates3), "c2r_total_rates")
    totrec=rates3(1)
    totcollisions=rates3(2)
    recomions=rates3(3)
    dh0=n_before(1)-n_after(1)
    dhe0=n_before(3)-n_after(3)
    dhe2=n_after(5)-n_before(5)
    total_ion=dh0+dhe0+dhe2

  end subroutine photon_statistics

  ! ===========================================================================

  !> the same from numbers c2r_iteration has already reduced on the device
  subroutine photon_statistics_from (reccoef,rates3,n_after)

    real(kind=dp),intent(in) :: reccoef(12), rates3(3), n_after(5)

    if (.not.isothermal) then
       arech0=reccoef(1); brech0=reccoef(2); areche0=reccoef(3); breche0=reccoef(4)
       oreche0=reccoef(5); areche1=reccoef(6); breche1=reccoef(7); treche1=reccoef(8)
       colli_HI=reccoef(9); colli_HeI=reccoef(10); colli_HeII=reccoef(11); v=reccoef(12)
    endif
    totrec=rates3(1)
    totcollisions=rates3(2)
    recomions=rates3(3)
    dh0=n_before(1)-n_after(1)
    dhe0=n_before(3)-n_after(3)
    dhe2=n_after(5)-n_before(5)
    total_ion=dh0+dhe0+dhe2

  end subroutine photon_statistics_from

  ! ===========================================================================

  !> C2RAY_HIP_STEPWISE, C2RAY_HIP_TIMING, C2R_ALLREDUCE_SLABS
  subroutine read_fused_settings ()

    character(len=32) :: text
    integer :: length, status, n

    fused_settings_read = .true.
    call get_environment_variable ("C2RAY_HIP_STEPWISE", text, length, status)
    if (status == 0 .and. length > 0) then
       if (text(1:1) /= "0") fused_iterations = .false.
    endif
    call get_environment_variable ("C2RAY_HIP_TIMING", text, length, status)
    if (status == 0 .and. length > 0) then
       if (text(1:1) /= "0") loop_timing = .true.
    endif
    ! with the loop's clock also the kernels of every iteration (HIP events on the library's streams)
    if (loop_timing) call check (c2r_enable_timing (hip_ctx, 1_c_int), "c2r_enable_timing")
    call get_environment_variable ("C2RAY_HIP_KEEP_STATE", text, length, status)
    if (status == 0 .and. length > 0) then
       read(text(1:length),*,iostat=status) n
       if (status == 0 .and. n >= 0 .and. n <= 2) keep_state = n
    endif
    if (rank == 0 .and. keep_state > 0) write(logf,*) "evolve3D: C2RAY_HIP_KEEP_STATE = ", keep_state
    call get_environment_variable ("C2R_ALLREDUCE_SLABS", text, length, status)
    if (status == 0 .and. length > 0) then
       read(text(1:length),*,iostat=status) n
       if (status == 0 .and. n > 0) allreduce_slabs = n
    endif
#if defined(C2RAY_REFERENCE_DO_GRID) || defined(C2RAY_GLOBAL_PASS_BY_CELL)
    fused_iterations = .false.
#endif
    if (rank == 0) then
       if (fused_iterations) then
          write(logf,*) "evolve3D: one library call per outer iteration (c2r_iteration)"
       else
          write(logf,*) "evolve3D: outer iterations call by call, as the reference's loop"
       endif
    endif

  end subroutine read_fused_settings

  ! ===========================================================================

  !> phih_grid, phihe_grid, phiheat, photon_loss_all, sum_nbox to their host mirrors
  subroutine download_rates (everything)

    logical,intent(in),optional :: everything !< whatever C2RAY_HIP_KEEP_STATE says (the iteration dump holds all of them)
    integer(c_int) :: nbox, which
    logical :: all_of_them
#ifdef C2RAY_REFERENCE_DO_SOURCE
    real(kind=dp) :: tail(NumFreqBnd)
#endif

    ! phiheat stays zero in an isothermal run, on the device and in evolve_data's array (zeroed when allocated,
    ! written by nobody else): no need to move 8 bytes per cell of zeros at every call
    which = 7
    if (isothermal .and. phiheat_host_is_zero) which = 3
    if (.not.isothermal) phiheat_host_is_zero = .false.
    ! C2RAY_HIP_KEEP_STATE: phihe_grid has no reader on the host but the iteration dump (which fetches it itself,
    ! write_iteration_dump); with level 2 neither have phih_grid and phiheat (output stream 3 is off, says the host)
    all_of_them = .false.
    if (present(everything)) all_of_them = everything
    if (keep_state >= 1 .and. .not.all_of_them) which = iand(which, 5)
    if (keep_state >= 2 .and. .not.all_of_them) which = 0
#ifdef C2RAY_REFERENCE_DO_SOURCE
    call check (c2r_download_rates_sel (hip_ctx, which, phih_grid, phihe_grid, phiheat, tail, nbox), "c2r_download_rates")
#else
    call check (c2r_download_rates_sel (hip_ctx, which, phih_grid, phihe_grid, phiheat, photon_loss_all, nbox), &
         "c2r_download_rates")
    sum_nbox=nbox
    sum_nbox_all=nbox
#endif

  end subroutine download_rates

  ! ===========================================================================

  !> Same content and record order as the reference's dump (evolve.F90:233-275), so either code can
  !! restart from the other's file
  subroutine write_iteration_dump (niter)

    integer,intent(in) :: niter
    character(len=20) :: iterfile
    type(c_ptr) :: tptr

    write(timefile,"(A,F8.1)") "Time before writing iterdump: ", timestamp_wallclock ()

    call download_rates (everything=.true.)
    call check (c2r_download_iter_state (hip_ctx, xh_av, xhe_av, xh_intermed, xhe_intermed), &
         "c2r_download_iter_state")
    if (.not.isothermal) then
       tptr = c_loc_real4 (temperature_grid)
       ! (xh, xhe on the device are still the start-of-step values the host holds)
       call check (c2r_download_state (hip_ctx, xh, xhe, tptr), "c2r_download_state")
    endif

    ndump=ndump+1
    if (mod(ndump,2) == 0) then
       iterfile="iterdump2.bin"
    else
       iterfile="iterdump1.bin"
    endif
    open(unit=iterdump,file=trim(adjustl(dump_dir))//iterfile,form="unformatted",status="unknown")
    write(iterdump) niter
    write(iterdump) photon_loss_all
    write(iterdump) phih_grid
    write(iterdump) xh_av
    write(iterdump) xh_intermed
    write(iterdump) phihe_grid
    write(iterdump) xhe_av
    write(iterdump) xhe_intermed
    if (.not.isothermal) then
       write(iterdump) phiheat
       write(iterdump) temperature_grid
    endif
    close(iterdump)

    write(timefile,"(A,F8.1)") "Time after writing iterdump: ", timestamp_wallclock ()

  end subroutine write_iteration_dump

  ! ===========================================================================

  !> Reload the iteration state of a dump (evolve.F90:279-367) and put it on the device
  subroutine start_from_dump (restart,niter)

    integer,intent(in) :: restart
    integer,intent(out) :: niter
    character(len=20) :: iterfile
    type(c_ptr) :: tptr

    niter=0
    if (rank == 0) then
       write(timefile,"(A,F8.1)") "Time before reading iterdump: ", timestamp_wallclock ()
       select case (restart)
       case (1)
          iterfile="iterdump1.bin"
       case (2)
          iterfile="iterdump2.bin"
       case default
          iterfile="iterdump.bin"
       end select
       open(unit=iterdump,file=trim(adjustl(dump_dir))//iterfile,form="unformatted",status="old")
       read(iterdump) niter
       read(iterdump) photon_loss_all
       read(iterdump) phih_grid
       read(iterdump) xh_av
       read(iterdump) xh_intermed
       read(iterdump) phihe_grid
       read(iterdump) xhe_av
       read(iterdump) xhe_intermed
       if (.not.isothermal) then
          read(iterdump) phiheat
          phiheat_host_is_zero = .false.
          read(iterdump) temperature_grid
       endif
       close(iterdump)
       write(logf,*) "Read iteration ",niter," from dump file"
       write(logf,*) 'photon loss counter: ',photon_loss_all
    endif
#ifdef MPI
    call MPI_BCAST (niter,1,MPI_INTEGER,0,MPI_COMM_NEW,mympierror)
    call MPI_BCAST (photon_loss_all,NumFreqBnd,MPI_DOUBLE_PRECISION,0,MPI_COMM_NEW,mympierror)
    call MPI_BCAST (phih_grid,size(phih_grid),MPI_DOUBLE_PRECISION,0,MPI_COMM_NEW,mympierror)
    call MPI_BCAST (phihe_grid,size(phihe_grid),MPI_DOUBLE_PRECISION,0,MPI_COMM_NEW,mympierror)
    call MPI_BCAST (xh_av,size(xh_av),MPI_DOUBLE_PRECISION,0,MPI_COMM_NEW,mympierror)
    call MPI_BCAST (xhe_av,size(xhe_av),MPI_DOUBLE_PRECISION,0,MPI_COMM_NEW,mympierror)
    call MPI_BCAST (xh_intermed,size(xh_intermed),MPI_DOUBLE_PRECISION,0,MPI_COMM_NEW,mympierror)
    call MPI_BCAST (xhe_intermed,size(xhe_intermed),MPI_DOUBLE_PRECISION,0,MPI_COMM_NEW,mympierror)
    if (.not.isothermal) then
       call MPI_BCAST (phiheat,size(phiheat),MPI_DOUBLE_PRECISION,0,MPI_COMM_NEW,mympierror)
       call MPI_BCAST (temperature_grid,size(temperature_grid),MPI_REAL,0,MPI_COMM_NEW,mympierror)
    endif
#endif
    call check (c2r_upload_rates (hip_ctx, phih_grid, phihe_grid, phiheat), "c2r_upload_rates")
    call check (c2r_upload_iter_state (hip_ctx, xh_av, xhe_av, xh_intermed, xhe_intermed), &
         "c2r_upload_iter_state")
    if (.not.isothermal) then
       tptr = c_loc_real4 (temperature_grid)
       call check (c2r_upload_state (hip_ctx, xh, xhe, tptr), "c2r_upload_state")
    endif
    if (rank == 0) write(timefile,"(A,F8.1)") "Time after reading iterdump: ", timestamp_wallclock ()

  end subroutine start_from_dump

  ! ===========================================================================

  !> rad_ini's tables and vectors, and the cooling curves, go to the device once
  subroutine upload_tables ()

    type(c_ptr) :: fvec(12)
    type(c_ptr) :: pt, pn, ht, hn
    real(kind=dp) :: cool(801,5)
    real(kind=dp) :: mintemp, dtemp
    logical :: device_tables, heat
    character(len=8) :: value
    integer :: length, status

    ! C2RAY_HIP_BUILD_TABLES=1: the photo-ionisation and heating tables are integrated on the device
    ! (c2r_build_tables) from the band set-up instead of being copied from rad_ini's host arrays
    call get_environment_variable("C2RAY_HIP_BUILD_TABLES", value, length, status)
    device_tables = (status == 0 .and. length > 0 .and. value(1:1) == "1")

    heat = allocated(bb_heat_thick_table) .and. allocated(f1ion_HI)
    pt = c_null_ptr
    pn = c_null_ptr
    ht = c_null_ptr
    hn = c_null_ptr
    fvec(:) = c_null_ptr
    if (.not.device_tables) then
       pt = c_loc_2d (bb_photo_thick_table)
       pn = c_loc_2d (bb_photo_thin_table)
    endif
    if (heat) then
       if (.not.device_tables) then
          ht = c_loc_2d (bb_heat_thick_table)
          hn = c_loc_2d (bb_heat_thin_table)
       endif
       fvec = (/ c_loc_1d(f1ion_HI), c_loc_1d(f1ion_HeI), c_loc_1d(f1ion_HeII), &
            c_loc_1d(f2ion_HI), c_loc_1d(f2ion_HeI), c_loc_1d(f2ion_HeII), &
            c_loc_1d(f1heat_HI), c_loc_1d(f1heat_HeI), c_loc_1d(f1heat_HeII), &
            c_loc_1d(f2heat_HI), c_loc_1d(f2heat_HeI), c_loc_1d(f2heat_HeII) /)
    endif
    call check (c2r_set_tables (hip_ctx, pt, pn, ht, hn, &
         sigma_HI, sigma_HeI, sigma_HeII, fvec, int(bb_FreqBnd_UpperLimit,c_int)), "c2r_set_tables")
    if (device_tables) call build_tables_on_device (0, heat)

#ifdef PL
    pt = c_null_ptr
    pn = c_null_ptr
    ht = c_null_ptr
    hn = c_null_ptr
    if (.not.device_tables) then
       pt = c_loc_2d (pl_photo_thick_table)
       pn = c_loc_2d (pl_photo_thin_table)
       if (allocated(pl_heat_thick_table)) then
          ht = c_loc_2d (pl_heat_thick_table)
          hn = c_loc_2d (pl_heat_thin_table)
       endif
    endif
    call check (c2r_set_sed_tables (hip_ctx, 1_c_int, pt, pn, ht, hn, &
         int(pl_FreqBnd_LowerLimit,c_int), int(pl_FreqBnd_UpperLimit,c_int)), "c2r_set_sed_tables")
    if (device_tables) call build_tables_on_device (1, heat)
#endif
#ifdef QUASARS
    pt = c_null_ptr
    pn = c_null_ptr
    ht = c_null_ptr
    hn = c_null_ptr
    if (.not.device_tables) then
       pt = c_loc_2d (qpl_photo_thick_table)
       pn = c_loc_2d (qpl_photo_thin_table)
       if (allocated(qpl_heat_thick_table)) then
          ht = c_loc_2d (qpl_heat_thick_table)
          hn = c_loc_2d (qpl_heat_thin_table)
       endif
    endif
    call check (c2r_set_sed_tables (hip_ctx, 2_c_int, pt, pn, ht, hn, &
         int(qpl_FreqBnd_LowerLimit,c_int), int(qpl_FreqBnd_UpperLimit,c_int)), "c2r_set_sed_tables")
    if (device_tables) call build_tables_on_device (2, heat)
#endif

    if (.not.isothermal) then
       ! the cooling curves are private to the reference's radiative_cooling module, so the same
       ! five files are read here the way setup_cool reads them (cooling_h.f90:76-171)
       call read_cooling_tables (cool, mintemp, dtemp)
       call check (c2r_set_cooling (hip_ctx, cool, mintemp, dtemp), "c2r_set_cooling")
    endif
    tables_uploaded = .true.

  end subroutine upload_tables

  ! ===========================================================================

  !> spec_integration (radiation_tables.f90:172-422) on the device for SED `sed` (0 BB, 1 PL, 2 QPL), from
  !! the public module data that spectrum_parms, setup_scalingfactors, romberg_initialisation and
  !! normalize_seds have set up
  subroutine build_tables_on_device (sed, heat)

    use mathconstants, only: pi
    use cgsconstants, only: hplanck, two_pi_over_c_square
    use cgsphotoconstants, only: ion_freq_HI, ion_freq_HeI, ion_freq_HeII
    use radiation_sizes, only: NumFreq, NumBndin1, NumBndin2, freq_min, delta_freq
    use radiation_sizes, only: cross_section_HI_powerlaw_index, cross_section_HeI_powerlaw_index
    use radiation_sizes, only: cross_section_HeII_powerlaw_index
    use radiation_sed_parameters, only: R_star2, h_over_kT
#ifdef PL
    use radiation_sed_parameters, only: pl_scaling, pl_index
#endif
#ifdef QUASARS
    use radiation_sed_parameters, only: qpl_scaling, qpl_index
#endif
    use radiation_tables, only: tau
    use romberg, only: romw

    integer,intent(in) :: sed
    logical,intent(in) :: heat

    type(c2r_sed_setup) :: setup
    real(kind=dp),dimension(NumFreqBnd),target,save :: xsec_index
    real(kind=dp),dimension(:),allocatable,target,save :: tau_copy, romw_copy
    integer :: b
    integer(c_int) :: with_heat

    ! the index spec_integration passes per band (radiation_tables.f90:278,315,349)
    do b=1,NumFreqBnd
       if (b <= NumBndin1) then
          xsec_index(b)=cross_section_HI_powerlaw_index(b)
       elseif (b <= NumBndin1+NumBndin2) then
          xsec_index(b)=cross_section_HeI_powerlaw_index(b)
       else
          xsec_index(b)=cross_section_HeII_powerlaw_index(b)
       endif
    enddo
    if (.not.allocated(tau_copy)) allocate(tau_copy(size(tau)),romw_copy(NumFreq+1))
    tau_copy(:)=tau(:)
    romw_copy(:)=romw(0:NumFreq,nint(log(real(NumFreq,dp))/log(2.0)))

    setup%nfreq=NumFreq
    setup%sed=sed
    setup%freq_min=c_loc_1d(freq_min)
    setup%delta_freq=c_loc_1d(delta_freq)
    setup%xsec_index=c_loc(xsec_index)
    setup%tau=c_loc(tau_copy)
    setup%romw=c_loc(romw_copy)
    setup%R_star2=R_star2
    setup%h_over_kT=h_over_kT
    setup%two_pi_over_c_square=two_pi_over_c_square
    setup%hplanck=hplanck
    setup%pi=pi
    setup%ion_freq_HI=ion_freq_HI
    setup%ion_freq_HeI=ion_freq_HeI
    setup%ion_freq_HeII=ion_freq_HeII
    setup%pl_scaling=1.0_dp
    setup%pl_index=1.0_dp
#ifdef PL
    if (sed == 1) then
       setup%pl_scaling=pl_scaling
       setup%pl_index=pl_index
    endif
#endif
#ifdef QUASARS
    if (sed == 2) then
       setup%pl_scaling=qpl_scaling
       setup%pl_index=qpl_index
    endif
#endif
    with_heat=0
    if (heat) with_heat=1
    call check (c2r_build_tables (hip_ctx, setup, with_heat), "c2r_build_tables")
    if (rank == 0) write(logf,"(A,I2)") "c2ray_hip: photo-ionisation and heating tables built on the device for SED ", sed

  end subroutine build_tables_on_device

  !----------------------------------------------------------------------------

  subroutine read_cooling_tables (cool, mintemp, dtemp)

    real(kind=dp),intent(out) :: cool(801,5)
    real(kind=dp),intent(out) :: mintemp, dtemp
    character(len=40),parameter :: files(5) = (/ &
         "../tables/H0-cool.tab                   ", &
         "../tables/H1-cool-B.tab                 ", &
         "../tables/He0-cool_new.tab              ", &
         "../tables/He1-cool_new_nocollion.tab    ", &
         "../tables/He2-cool.tab                  " /)
    real(kind=dp) :: temp(801)
    integer :: n, itemp, element, ion, nchck, u

    do n=1,5
       open(newunit=u,file=trim(files(n)),status='old')
       read(u,*) element,ion,nchck
       do itemp=1,801
          read(u,*) temp(itemp),cool(itemp,n)
       enddo
       close(u)
       if (n == 1) then
          mintemp=temp(1)
          dtemp=temp(2)-temp(1)
       endif
    enddo
    do n=1,5
       do itemp=1,801
          cool(itemp,n)=10.0d0**cool(itemp,n)
       enddo
    enddo

  end subroutine read_cooling_tables

  ! ===========================================================================


  ! ===========================================================================

  !> The reference has no error convention (it logs and continues, or stops on unusable input):
  !! a failing device call is logged to logf and stops the run.
  subroutine check (ierr, what)

    integer(c_int),intent(in) :: ierr
    character(len=*),intent(in) :: what

    if (ierr /= 0) then
       write(logf,*) "c2ray_hip: ", what, " failed: ", c2r_error_text (hip_ctx)
       write(*,*) "c2ray_hip: ", what, " failed: ", c2r_error_text (hip_ctx)
       flush(logf)
#ifdef MPI
       ! the other ranks would wait in their next collective for ever
       call MPI_ABORT (MPI_COMM_NEW, 1, mympierror)
#endif
       stop 1
    endif

  end subroutine check

  function c_loc_1d (a) result(p)
    real(kind=dp),dimension(:),allocatable,target,intent(in) :: a
    type(c_ptr) :: p
    p = c_loc (a)
  end function c_loc_1d

  function c_loc_2d (a) result(p)
    real(kind=dp),dimension(:,:),allocatable,target,intent(in) :: a
    type(c_ptr) :: p
    p = c_loc (a)
  end function c_loc_2d

  function c_loc_real4 (a) result(p)
    real(kind=4),dimension(:,:,:,:),allocatable,target,intent(in) :: a
    type(c_ptr) :: p
    p = c_loc (a)
  end function c_loc_real4

end module evolve
