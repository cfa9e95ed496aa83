!> Drop-in replacement of the reference's module evolve_source (files_for_3D/evolve_source.F90): same
!! module name, same public symbols -- `do_source(dt,ns1,niter)`, `sum_nbox`, `sum_nbox_all` -- so that
!! code written against them keeps working: the reference's own master_slave.F90 (do_grid_static and
!! the master-slave source queue for more than ten ranks, master_slave.F90:53-326) links against this
!! module unchanged and hands its sources to the GPU one at a time.
!!
!! do_source traces one source on the device (sub-box loop, column sweep, rates) and adds its
!! contribution to the device-resident rate grids, photon loss and sub-box count -- what
!! evolve_source.F90:66-238 does with evolve0D on the host.  evolve3D's own pass (module evolve) does
!! not go through here by default: it hands the whole source list of the rank to c2r_pass_sources,
!! which batches sources per launch.  Both give the same bits (the sum over sources keeps its order).
module evolve_source

  use precision, only: dp
  use file_admin, only: logf
  use evolve_data, only: hip_ctx
  use, intrinsic :: iso_c_binding
  use c2ray_hip, only: c2r_do_source, c2r_error_text

  implicit none

  save

  private

  public :: do_source

  !> sum of all nboxes (on one processor)
  integer,public :: sum_nbox
  !> sum of all nboxes (on all processors)
  integer,public :: sum_nbox_all

contains

  !> Does the ray-tracing for one source (files_for_3D/evolve_source.F90:66)
  subroutine do_source (dt,ns1,niter)

    real(kind=dp),intent(in) :: dt !< time step; the reference passes it on to evolve0D, which ignores it
    integer,intent(in) :: ns1      !< number of the source being done
    integer,intent(in) :: niter    !< iteration counter (only selects a dead branch in the reference)

    integer(c_int) :: ierr

    ierr = c2r_do_source (hip_ctx, int(ns1,c_int))
    if (ierr /= 0) then
       write(logf,*) "c2ray_hip error in do_source: ", c2r_error_text(hip_ctx)
       write(*,*) "c2ray_hip error in do_source: ", c2r_error_text(hip_ctx)
       flush(logf)
       stop 1
    endif

  end subroutine do_source

end module evolve_source
