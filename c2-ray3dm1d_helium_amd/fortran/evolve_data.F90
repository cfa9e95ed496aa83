!> Drop-in replacement of the reference's module evolve_data (files_for_3D/evolve_data.F90): the
!! same module name and the same public symbols the rest of the reference uses
!! (C2Ray.F90:59 `use evolve_data, only: evolve_ini`; output.F90:26 `use evolve_data, only:
!! phih_grid, phiheat`), but the work arrays live on the GPU.  The host arrays declared here are
!! the mirrors evolve3D fills after the device loop (output routines and photon statistics read
!! them); coldensh_out etc. are scratch of the reference's CPU sweep and have no host mirror.
module evolve_data

  use precision, only: dp
  use my_mpi
  use sizes, only: mesh, Ndim
  use radiation_sizes, only: NumFreqBnd
  use file_admin, only: logf
  use, intrinsic :: iso_c_binding, only: c_ptr, c_null_ptr, c_int
  use c2ray_hip, only: c2r_create, c2r_error_text

  implicit none

  save

  !> Periodic boundary conditions, has to be true for this version
  logical,parameter :: periodic_bc = .true.

  !> H Photo-ionization rate on the entire grid
  real(kind=dp),dimension(:,:,:),allocatable :: phih_grid
  !> He Photo-ionization rate on the entire grid
  real(kind=dp),dimension(:,:,:,:),allocatable :: phihe_grid
  !> Heating  rate on the entire grid
  real(kind=dp),dimension(:,:,:),allocatable :: phiheat
  !> Time-averaged H ionization fraction
  real(kind=dp),dimension(:,:,:,:),allocatable :: xh_av
  !> Time-averaged He ionization fraction
  real(kind=dp),dimension(:,:,:,:),allocatable :: xhe_av
  !> Intermediate result for H ionization fraction
  real(kind=dp),dimension(:,:,:,:),allocatable :: xh_intermed
  !> Intermediate result for He ionization fraction
  real(kind=dp),dimension(:,:,:,:),allocatable :: xhe_intermed
  !> Photon loss from the grid
  real(kind=dp) :: photon_loss_all(1:NumFreqBnd)

  !> The device context (include/c2ray_hip.h); one per rank = one per GPU
  type(c_ptr) :: hip_ctx = c_null_ptr
  !> GPU used by this rank
  integer :: hip_device = 0

contains

  !> Allocate the arrays needed for evolve: host mirrors here, work arrays on the device
  subroutine evolve_ini ()

    integer(c_int) :: ierr
    integer(c_int) :: cmesh(3)
    character(len=16) :: value
    integer :: length, status

    allocate(phih_grid(mesh(1),mesh(2),mesh(3)))
    phih_grid=0.0 ! Needs value for initial output
    allocate(phihe_grid(mesh(1),mesh(2),mesh(3),0:1))
    phihe_grid=0.0
    allocate(phiheat(mesh(1),mesh(2),mesh(3)))
    phiheat=0.0 ! Needs value for initial output
    allocate(xh_av(mesh(1),mesh(2),mesh(3),0:1))
    allocate(xhe_av(mesh(1),mesh(2),mesh(3),0:2))
    allocate(xh_intermed(mesh(1),mesh(2),mesh(3),0:1))
    allocate(xhe_intermed(mesh(1),mesh(2),mesh(3),0:2))
    photon_loss_all(:)=0.0

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
