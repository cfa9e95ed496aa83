!> Drop-in replacement of the reference's module evolve (files_for_3D/evolve.F90): same module
!! name, same public symbol `evolve3D(time,dt,restart)` with the same meaning; the convergence loop
!! {zero rates -> all sources -> global chemistry pass} runs on the GPU through the C ABI of
!! include/c2ray_hip.h (module c2ray_hip).  Everything around it -- the driver, source lists,
!! material set-up, cosmology, output routines, photon statistics -- is the reference's own code,
!! untouched: this file only marshals the module data that evolve3D reads and writes
!! (SURVEY.md section 8b).
module evolve

  use precision, only: dp
  use clocks, only: timestamp_wallclock
  use file_admin, only: logf, timefile
  use my_mpi ! rank, npr
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
  use cosmology, only: zred
  use cosmology_parameters, only: H0, Omega0
  use cgsconstants, only: arech0, brech0, areche0, breche0, oreche0, areche1, breche1, treche1
  use cgsconstants, only: colli_HI, colli_HeI, colli_HeII, v
  use photonstatistics, only: photon_loss, LLS_loss
  use photonstatistics, only: totrec, totcollisions, recomions, dh0, dhe0, dhe2, total_ion
  use photonstatistics, only: report_photonstatistics, update_grandtotal_photonstatistics
  use evolve_data, only: phih_grid, phihe_grid, phiheat
  use evolve_data, only: xh_av, xhe_av, xh_intermed, xhe_intermed
  use evolve_data, only: photon_loss_all, hip_ctx
  use, intrinsic :: iso_c_binding
  use c2ray_hip

  implicit none

  save

  private

  public :: evolve3D

  !> sum of all nboxes (the reference keeps these in module evolve_source)
  integer,public :: sum_nbox
  integer,public :: sum_nbox_all

  logical :: tables_uploaded = .false.

contains

  ! ===========================================================================

  !> Evolve the entire grid over a time step dt
  subroutine evolve3D (time,dt,restart)

    real(kind=dp),intent(in) :: time !< time (unused, as in the reference)
    real(kind=dp),intent(in) :: dt !< time step
    integer,intent(in) :: restart !< restart flag

    integer(c_int) :: niter
    integer(c_int) :: conv_flags(512)
    integer(c_int) :: nbox
    integer :: n
    integer(c_int) :: iso
    real(kind=dp) :: reccoef(12)
    real(kind=dp) :: before(5), after(5), rates3(3)
    type(c_ptr) :: tptr

    if (restart /= 0) then
       write(logf,*) "evolve3D (HIP): restart from an iteration dump is not available in this build"
       stop 1
    endif
    if (npr > 1) then
       write(logf,*) "evolve3D (HIP): multi-rank Fortran hosts must drive c2r_pass_sources / an ", &
            "all-reduce / c2r_global_pass themselves (see INTEGRATION.md); npr = ", npr
       stop 1
    endif

    if (.not. tables_uploaded) call upload_tables ()

    ! --- host state that changes between calls (cosmo_evol rescales dr, vol, ndens each step)
    iso = 0
    if (isothermal) iso = 1
    reccoef = (/ arech0, brech0, areche0, breche0, oreche0, areche1, breche1, treche1, &
         colli_HI, colli_HeI, colli_HeII, v /)
    call check (c2r_set_step (hip_ctx, ndens, dr, vol, real(clumping,c_float), zred, H0, Omega0, &
         iso, temper_val, reccoef), "c2r_set_step")
    if (NumSrc > 0) then
       call check (c2r_set_sources (hip_ctx, int(NumSrc,c_int), srcpos, NormFlux(1:NumSrc), S_star), &
            "c2r_set_sources")
    else
       call check (c2r_set_sources (hip_ctx, 0_c_int, (/ 0_c_int /), (/ 0.0_dp /), S_star), &
            "c2r_set_sources")
    endif
    tptr = c_null_ptr
    if (.not.isothermal) tptr = c_loc_real4 (temperature_grid)
    call check (c2r_upload_state (hip_ctx, xh, xhe, tptr), "c2r_upload_state")

    ! Initial state (for photon statistics): state_before (photonstatistics.f90:117-144), on the device
    call check (c2r_state_sums (hip_ctx, 0_c_int, before), "c2r_state_sums")

    if (rank == 0) write(timefile,"(A,F8.1)") &
         "Time before starting iteration: ", timestamp_wallclock ()

    ! --- the convergence loop of evolve.F90:154-222, on the device
    call check (c2r_evolve3d (hip_ctx, dt, niter, conv_flags, 512_c_int), "c2r_evolve3d")

    ! --- results back to the modules that own them
    call check (c2r_download_state (hip_ctx, xh, xhe, tptr), "c2r_download_state")
    call check (c2r_download_rates (hip_ctx, phih_grid, phihe_grid, phiheat, photon_loss_all, nbox), &
         "c2r_download_rates")
    ! xh_av, xhe_av, xh_intermed, xhe_intermed stay on the device: nothing outside the evolve chain
    ! reads them (c2r_download_iter_state fetches them when a dump is wanted)
    sum_nbox = nbox
    sum_nbox_all = nbox
    ! global_pass leaves the mean loss per cell in photonstatistics:photon_loss (evolve.F90:457)
    photon_loss(:)=photon_loss_all(:)/(real(mesh(1))*real(mesh(2))*real(mesh(3)))
    LLS_loss = 0.0

    if (rank == 0) then
       do n=1,niter
          write(logf,*) "Number of non-converged points: ",conv_flags(n)
       enddo
       if (NumSrc > 0) write(logf,*) "Average number of subboxes: ", &
            real(sum_nbox_all)/real(NumSrc)
       if (niter <= 500) then
          write(logf,*) "Multiple sources convergence reached"
       else
          write(logf,*) 'Multiple sources not converging'
       endif
       write(timefile,"(A,I3,A,F8.1)") &
            "Time after iteration ",niter," : ", timestamp_wallclock ()
    endif

    ! The reference's global pass leaves the temperature-dependent coefficients of the last cell in the
    ! module variables of cgsconstants (evolve_point.F90:543); photon statistics use them
    if (.not.isothermal) then
       call check (c2r_get_reccoef (hip_ctx, reccoef), "c2r_get_reccoef")
       arech0=reccoef(1); brech0=reccoef(2); areche0=reccoef(3); breche0=reccoef(4)
       oreche0=reccoef(5); areche1=reccoef(6); breche1=reccoef(7); treche1=reccoef(8)
       colli_HI=reccoef(9); colli_HeI=reccoef(10); colli_HeII=reccoef(11); v=reccoef(12)
    endif

    ! Calculate photon statistics: calculate_photon_statistics (dt,xh,xh_av,xhe,xhe_av) of
    ! evolve.F90:225 = state_after(xh,xhe) + total_rates(dt,xh_av,xhe_av) + total_ionizations, with
    ! the grid reductions done on the device
    call check (c2r_state_sums (hip_ctx, 0_c_int, after), "c2r_state_sums")
    call check (c2r_total_rates (hip_ctx, dt, reccoef, rates3), "c2r_total_rates")
    totrec=rates3(1)
    totcollisions=rates3(2)
    recomions=rates3(3)
    dh0=before(1)-after(1)
    dhe0=before(3)-after(3)
    dhe2=after(5)-before(5)
    total_ion=dh0+dhe0+dhe2
    call report_photonstatistics (dt)
    call update_grandtotal_photonstatistics (dt)

  end subroutine evolve3D

  ! ===========================================================================

  !> rad_ini's tables and vectors, and the cooling curves, go to the device once
  subroutine upload_tables ()

    type(c_ptr) :: fvec(12)
    type(c_ptr) :: ht, hn
    real(kind=dp) :: cool(801,5)
    real(kind=dp) :: mintemp, dtemp

    if (allocated(bb_heat_thick_table) .and. allocated(f1ion_HI)) then
       ht = c_loc_2d (bb_heat_thick_table)
       hn = c_loc_2d (bb_heat_thin_table)
       fvec = (/ c_loc_1d(f1ion_HI), c_loc_1d(f1ion_HeI), c_loc_1d(f1ion_HeII), &
            c_loc_1d(f2ion_HI), c_loc_1d(f2ion_HeI), c_loc_1d(f2ion_HeII), &
            c_loc_1d(f1heat_HI), c_loc_1d(f1heat_HeI), c_loc_1d(f1heat_HeII), &
            c_loc_1d(f2heat_HI), c_loc_1d(f2heat_HeI), c_loc_1d(f2heat_HeII) /)
    else
       ht = c_null_ptr
       hn = c_null_ptr
       fvec(:) = c_null_ptr
    endif
    call check (c2r_set_tables (hip_ctx, bb_photo_thick_table, bb_photo_thin_table, ht, hn, &
         sigma_HI, sigma_HeI, sigma_HeII, fvec, int(bb_FreqBnd_UpperLimit,c_int)), "c2r_set_tables")

    if (.not.isothermal) then
       ! the cooling curves are private to the reference's radiative_cooling module, so the same
       ! five files are read here the way setup_cool reads them (cooling_h.f90:76-171)
       call read_cooling_tables (cool, mintemp, dtemp)
       call check (c2r_set_cooling (hip_ctx, cool, mintemp, dtemp), "c2r_set_cooling")
    endif
    tables_uploaded = .true.

  end subroutine upload_tables

  ! ===========================================================================

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

  !> The reference has no error convention (it logs and continues, or stops on unusable input):
  !! a failing device call is logged to logf and stops the run.
  subroutine check (ierr, what)

    integer(c_int),intent(in) :: ierr
    character(len=*),intent(in) :: what

    if (ierr /= 0) then
       write(logf,*) "c2ray_hip: ", what, " failed: ", c2r_error_text (hip_ctx)
       write(*,*) "c2ray_hip: ", what, " failed: ", c2r_error_text (hip_ctx)
       flush(logf)
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
