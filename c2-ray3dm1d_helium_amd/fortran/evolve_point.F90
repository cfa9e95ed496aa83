!> Drop-in replacement of the reference's module evolve_point (files_for_3D/evolve_point.F90): same module
!! name, same public symbols `evolve0D(dt,rtpos,ns,niter)` and `evolve0D_global(dt,pos,conv_flag)`.
!!
!! In the reference these two are the per-cell bodies of the ray-tracing sweep and of the global pass; on the
!! device a source is traced as a whole (module evolve_source, do_source) and the global pass is one launch
!! (module evolve, global_pass), so neither of this library's own modules calls into here.  The symbols are for
!! code written against the per-cell interface, and do on the device, one cell per call, what the reference does:
!!   * evolve0D_global applies the collected rates to ONE cell (c2r_evolve0d_global) with the reference's
!!     meaning of conv_flag -- a host loop over all cells, as in the reference's global_pass
!!     (evolve.F90:477-484), gives the same grids as the one-launch pass;
!!   * evolve0D traces ONE cell for one source (c2r_evolve0d): incoming columns by short characteristics from
!!     the cells of that source done before, the cell's own columns, its rates added to the rate grids, its
!!     share of the photon loss when it lies on the surface of the current sub-box.  The caller brings the
!!     order: the reference's own do_source, linked unmodified on top of this module, visits the cells through
!!     evolve2D etc. (evolve_source.F90:244-608) exactly as it does on the CPU (build variant C2Ray_3D_hip_bypoint,
!!     oracle/ref_build.sh; C2RAY_HIP_POINT_INTERFACE=1 allocates the host marks it needs).
!! One launch per cell: interface parity, tests and small meshes -- do_source is the production path.
module evolve_point

  use precision, only: dp
  use file_admin, only: logf
  use sizes, only: Ndim, mesh
  use evolve_data, only: hip_ctx
  use evolve_data, only: coldensh_out, photon_loss_src_thread, last_l, last_r, tn
  use, intrinsic :: iso_c_binding
  use c2ray_hip, only: c2r_evolve0d_global, c2r_evolve0d, c2r_error_text

  implicit none

  save

  private

  public :: evolve0D, evolve0D_global

contains

  !> Photo-ionisation rates of one cell due to one source (files_for_3D/evolve_point.F90:79)
  subroutine evolve0D (dt,rtpos,ns,niter)

    real(kind=dp),intent(in) :: dt !< time step (unused, as in the reference's live branch)
    integer,dimension(Ndim),intent(in) :: rtpos !< cell position (for RT)
    integer,intent(in) :: ns !< source number
    integer,intent(in) :: niter !< global iteration number

    integer :: idim
    integer,dimension(Ndim) :: pos
    integer(c_int) :: crt(3), surface, ierr
    real(kind=dp) :: loss

    if (.not.allocated(coldensh_out)) then
       write(logf,*) "c2ray_hip: evolve0D needs C2RAY_HIP_POINT_INTERFACE=1 (the host marks of the cell-by-cell interface)"
       write(*,*) "c2ray_hip: evolve0D needs C2RAY_HIP_POINT_INTERFACE=1 in the environment"
       flush(logf)
       stop 1
    endif

    ! Map pos to mesh pos, assuming a periodic mesh
    do idim=1,Ndim
       pos(idim)=modulo(rtpos(idim)-1,mesh(idim))+1
    enddo

    ! If coldensh_out is zero, we have not done this point yet, so do it (evolve_point.F90:118-120).  The
    ! columns stay on the device; the host array carries the mark.
    if (coldensh_out(pos(1),pos(2),pos(3)) == 0.0) then
       surface = 0
       if (any(rtpos(:) == last_l(:)) .or. any(rtpos(:) == last_r(:))) surface = 1
       crt(:) = int(rtpos(:), c_int)
       loss = 0.0_dp
       ierr = c2r_evolve0d (hip_ctx, crt, int(ns,c_int), int(niter,c_int), surface, loss)
       if (ierr /= 0) then
          write(logf,*) "c2ray_hip error in evolve0D: ", c2r_error_text(hip_ctx)
          write(*,*) "c2ray_hip error in evolve0D: ", c2r_error_text(hip_ctx)
          flush(logf)
          stop 1
       endif
       coldensh_out(pos(1),pos(2),pos(3)) = 1.0_dp
       ! Photon statistics: register number of photons leaving the grid (evolve_point.F90:310-315)
       if (surface == 1) photon_loss_src_thread(tn)=photon_loss_src_thread(tn) + loss
    endif

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
