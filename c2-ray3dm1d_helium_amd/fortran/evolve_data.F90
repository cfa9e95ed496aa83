!> Drop-in replacement of the reference's module evolve_data (files_for_3D/evolve_data.F90): the
!! same module name and the same public symbols the rest of the reference uses
!! (C2Ray.F90:59 `use evolve_data, only: evolve_ini`; output.F90:26 `use evolve_data, only:
!! phih_grid, phiheat`), but the work arrays live on the GPU.  The host arrays declared here are
!! the mirrors evolve3D fills after the device loop (output routines and photon statistics read
!! them); coldensh_out etc. are scratch of the reference's CPU sweep and have no host mirror.
module evolve_data

  use, intrinsic :: iso_c_binding, only: c_ptr, c_null_ptr, c_int
  use c2ray_hip, only: c2r_create, c2r_error_text
  use file_admin, only: logf
  use my_mpi                                  ! rank
  use precision, only: dp
  use radiation_sizes, only: NumFreqBnd
  use sizes, only: Ndim, mesh

  implicit none
  save

  !> the sweep wraps around the mesh edges (the only mode the reference supports here)
  logical, parameter :: periodic_bc = .true.

  ! host mirrors of device arrays, public under the reference's names
  !> photo-ionisation rates summed over all sources: H, and He (components 0:1)
  real(kind=dp), allocatable :: phih_grid(:,:,:), phihe_grid(:,:,:,:)
  !> heating rate summed over all sources
  real(kind=dp), allocatable :: phiheat(:,:,:)
  !> ionisation fractions averaged over the time step: H (0:1), He (0:2)
  real(kind=dp), allocatable :: xh_av(:,:,:,:), xhe_av(:,:,:,:)
  !> ionisation fractions at the end of the time step, current iterate: H (0:1), He (0:2)
  real(kind=dp), allocatable :: xh_intermed(:,:,:,:), xhe_intermed(:,:,:,:)
  !> photons that left the mesh, per frequency band, summed over sources (and ranks)
  real(kind=dp) :: photon_loss_all(NumFreqBnd)

  !> The device context (include/c2ray_hip.h); one per rank = one per GPU
  type(c_ptr) :: hip_ctx = c_null_ptr
  !> GPU used by this rank
  integer :: hip_device = 0

contains

  !> evolve_ini of the reference (files_for_3D/evolve_data.F90:74): host mirrors here, work arrays on
  !! the device
  subroutine evolve_ini ()

    integer(c_int) :: ierr
    integer(c_int) :: cmesh(3)
    character(len=16) :: value
    integer :: length, status
    integer :: n1, n2, n3

    n1 = mesh(1)
    n2 = mesh(2)
    n3 = mesh(3)
    ! the rate grids are written out before the first evolve3D call (output.F90), so they start at zero
    allocate(phih_grid(n1,n2,n3), phiheat(n1,n2,n3), phihe_grid(n1,n2,n3,0:1))
    phih_grid(:,:,:) = 0.0_dp
    phiheat(:,:,:) = 0.0_dp
    phihe_grid(:,:,:,:) = 0.0_dp
    allocate(xh_av(n1,n2,n3,0:1), xh_intermed(n1,n2,n3,0:1))
    allocate(xhe_av(n1,n2,n3,0:2), xhe_intermed(n1,n2,n3,0:2))
    photon_loss_all(:) = 0.0_dp

    ! one rank per GPU: rank r of a node uses device r unless C2RAY_HIP_DEVICE says otherwise
    hip_device = rank
    call get_environment_variable("C2RAY_HIP_DEVICE", value, length, status)
    if (status == 0 .and. length > 0) read(value(1:length),*) hip_device

    cmesh(:) = mesh(:)
    ierr = c2r_create(hip_ctx, int(hip_device, c_int), cmesh)
    if (ierr /= 0) then
       write(logf,*) "c2ray_hip: ", c2r_error_text(c_null_ptr)
       write(*,*) "c2ray_hip: ", c2r_error_text(c_null_ptr)
       stop 1
    endif
    if (rank == 0) write(logf,"(A,I3)") "evolve_ini: evolve3D runs on HIP device ", hip_device

  end subroutine evolve_ini

end module evolve_data
