!> Drop-in replacement of the reference's module evolve_point (files_for_3D/evolve_point.F90): same module
!! name, same public symbols `evolve0D(dt,rtpos,ns,niter)` and `evolve0D_global(dt,pos,conv_flag)`.
!!
!! In the reference these two are the per-cell bodies of the ray-tracing sweep and of the global pass; on the
!! device a source is traced as a whole (module evolve_source, do_source) and the global pass is one launch
!! (module evolve, global_pass), so neither module calls into here.  The symbols stay for code written against
!! the per-cell interface:
!!   * evolve0D_global applies the collected rates to ONE cell on the device (c2r_evolve0d_global) with the
!!     reference's meaning of conv_flag -- a host loop over all cells, as in the reference's global_pass
!!     (evolve.F90:477-484), gives the same grids as the one-launch pass (one launch per cell: for tests and
!!     small meshes);
!!   * evolve0D has no cell-by-cell counterpart: the column of a cell needs its upstream cells of the same
!!     source in the order the sweep visits them, which only do_source knows.  Calling it is a programming
!!     error that is logged and stops the run.
module evolve_point

  use precision, only: dp
  use file_admin, only: logf
  use sizes, only: Ndim
  use evolve_data, only: hip_ctx
  use, intrinsic :: iso_c_binding
  use c2ray_hip, only: c2r_evolve0d_global, c2r_error_text

  implicit none

  save

  private

  public :: evolve0D, evolve0D_global

contains

  !> Photo-ionisation rate of one cell due to one source (files_for_3D/evolve_point.F90:79): not available cell
  !! by cell, see the module header
  subroutine evolve0D (dt,rtpos,ns,niter)

    real(kind=dp),intent(in) :: dt !< time step
    integer,dimension(Ndim),intent(in) :: rtpos !< cell position (for RT)
    integer,intent(in) :: ns !< source number
    integer,intent(in) :: niter !< global iteration number

    write(logf,*) "c2ray_hip: evolve0D called for cell ", rtpos, " source ", ns, &
         ": the device traces a source as a whole; call do_source (module evolve_source)"
    write(*,*) "c2ray_hip: evolve0D is not available cell by cell; call do_source (module evolve_source)"
    flush(logf)
    stop 1

  end subroutine evolve0D

  !> Evolution of the ionisation state of one cell from the collected rates of all sources
  !! (files_for_3D/evolve_point.F90:325)
  subroutine evolve0D_global (dt,pos,conv_flag)

    real(kind=dp),intent(in) :: dt !< time step
    integer,dimension(Ndim),intent(in) :: pos !< position on mesh
    integer,intent(inout) :: conv_flag !< convergence counter

    integer(c_int) :: cpos(3), cf, ierr

    cpos(:) = int(pos(:), c_int)
    cf = int(conv_flag, c_int)
    ierr = c2r_evolve0d_global (hip_ctx, dt, cpos, cf)
    if (ierr /= 0) then
       write(logf,*) "c2ray_hip error in evolve0D_global: ", c2r_error_text(hip_ctx)
       write(*,*) "c2ray_hip error in evolve0D_global: ", c2r_error_text(hip_ctx)
       flush(logf)
       stop 1
    endif
    conv_flag = cf

  end subroutine evolve0D_global

end module evolve_point
