!> Drop-in replacement of the reference's module evolve_data (files_for_3D/evolve_data.F90): the
!! same module name and the same public symbols the rest of the reference uses
!! (C2Ray.F90:59 `use evolve_data, only: evolve_ini`; output.F90:26 `use evolve_data, only:
!! phih_grid, phiheat`), but the work arrays live on the GPU.  The host arrays declared here are
!! the mirrors evolve3D fills after the device loop (output routines and photon statistics read
!! them); coldensh_out etc. are scratch of the reference's CPU sweep and have no host mirror.
module evolve_data

  use, intrinsic :: iso_c_binding, only: c_ptr, c_null_ptr, c_int, c_char, c_double, c_null_char
  use c2ray_hip, only: c2r_create, c2r_create_multi, c2r_error_text, c2r_device_count, c2r_get_constants
  use c2ray_hip, only: c2r_comm_unique_id, c2r_comm_init, c2r_comm_init_local, c2r_comm_kind, c2r_comm_library
  use file_admin, only: logf
  use my_mpi                                  ! rank, npr, MPI_COMM_NEW
  use precision, only: dp
  use radiation_sizes, only: NumFreqBnd, NumheatBin, NumTau
  use sizes, only: Ndim, mesh
  use abundances, only: abu_he, abu_c
  use c2ray_parameters, only: subboxsize, max_subbox, epsilon, convergence_fraction
  use c2ray_parameters, only: minimum_fractional_change, minimum_fraction_of_atoms
  use c2ray_parameters, only: relative_denergy, minitemp, add_photon_losses

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

  ! What the reference's own do_source (files_for_3D/evolve_source.F90, linked unmodified on top of this library's
  ! evolve0D: the cell-by-cell interface, INTEGRATION.md) keeps in this module.  The columns themselves live on the
  ! device; coldensh_out only carries the reference's "this cell has been done for this source" mark (do_source
  ! zeroes it per source, evolve0D tests it).  Allocated when C2RAY_HIP_POINT_INTERFACE=1.
  integer :: tn=1 !< thread number
  real(kind=dp),dimension(:,:,:),allocatable :: coldensh_out
  real(kind=dp),dimension(:,:,:,:),allocatable :: coldenshe_out
  real(kind=dp),dimension(:),allocatable :: photon_loss_src_thread
  integer,dimension(Ndim) :: last_l !< mesh position of left end point for RT
  integer,dimension(Ndim) :: last_r !< mesh position of right end point for RT

  !> The device context (include/c2ray_hip.h): the GPU(s) of this rank
  type(c_ptr) :: hip_ctx = c_null_ptr
  !> first GPU used by this rank, and how many it drives (C2RAY_HIP_NGPU > 1: one process, several GPUs)
  integer :: hip_device = 0
  integer :: hip_ndevices = 1

contains

  !> evolve_ini of the reference (files_for_3D/evolve_data.F90:74): host mirrors here, work arrays on
  !! the device
  subroutine evolve_ini ()

    integer(c_int) :: ierr
    integer(c_int) :: cmesh(3)
    integer(c_int), allocatable :: devices(:)
    character(len=32) :: value
    integer :: length, status
    integer :: n1, n2, n3, ndev_node, local_rank, offset, i

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

    call get_environment_variable ("C2RAY_HIP_POINT_INTERFACE", value, length, status)
    if (status == 0 .and. length > 0) then
       if (value(1:1) /= "0") then
          allocate(coldensh_out(n1,n2,n3), coldenshe_out(n1,n2,n3,0:1), photon_loss_src_thread(1))
          coldensh_out(:,:,:) = 0.0_dp
          coldenshe_out(:,:,:,:) = 0.0_dp
          photon_loss_src_thread(:) = 0.0_dp
       endif
    endif

    call check_compiled_constants ()

    ! Which GPU(s).  A rank drives C2RAY_HIP_NGPU devices (default 1) starting at
    ! (node-local rank * C2RAY_HIP_NGPU + C2RAY_HIP_DEVICE) modulo the devices of the node; the node-local
    ! rank is what the MPI launcher exports (Open MPI, MVAPICH, Slurm), else the global rank.
    ndev_node = c2r_device_count ()
    local_rank = rank
    call env_integer ("OMPI_COMM_WORLD_LOCAL_RANK", local_rank)
    call env_integer ("MV2_COMM_WORLD_LOCAL_RANK", local_rank)
    call env_integer ("SLURM_LOCALID", local_rank)
    offset = 0
    call env_integer ("C2RAY_HIP_DEVICE", offset)
    hip_ndevices = 1
    call env_integer ("C2RAY_HIP_NGPU", hip_ndevices)
    hip_ndevices = max(1, hip_ndevices)
    allocate(devices(hip_ndevices))
    do i = 1, hip_ndevices
       devices(i) = int(modulo(local_rank*hip_ndevices + offset + i - 1, max(1, ndev_node)), c_int)
    enddo
    ! C2RAY_HIP_SAME_DEVICE=1: all on one GPU (a rehearsal of the multi-device path on a one-GPU box)
    i = 0
    call env_integer ("C2RAY_HIP_SAME_DEVICE", i)
    if (i /= 0) devices(:) = int(modulo(offset, max(1, ndev_node)), c_int)
    hip_device = devices(1)

    cmesh(:) = mesh(:)
    if (hip_ndevices > 1) then
       ierr = c2r_create_multi(hip_ctx, int(hip_ndevices, c_int), devices, cmesh)
    else
       ierr = c2r_create(hip_ctx, devices(1), cmesh)
    endif
    if (ierr /= 0) call stop_with (c2r_error_text(c_null_ptr))
    if (rank == 0) write(logf,"(A,I3,A,I3)") "evolve_ini: evolve3D runs on HIP device ", hip_device, &
         ", devices per rank: ", hip_ndevices

    call setup_communicator ()

  end subroutine evolve_ini

  !> The sum over ranks of mpi_accumulate_grid_quantities (files_for_3D/evolve.F90:505-548) is an RCCL
  !! all-reduce inside the library.  All it needs from the host's own parallel layer is to carry 128 bytes from
  !! rank 0 to the others once.
  subroutine setup_communicator ()

    character(kind=c_char) :: id(128), libpath(1024)
    integer(c_int) :: ierr
    integer :: force, nchar
#ifdef MPI
    integer :: mympierror
#endif

    if (npr > 1) then
#ifdef MPI
       if (rank == 0) then
          ierr = c2r_comm_unique_id (id)
          if (ierr /= 0) call stop_with (c2r_error_text(c_null_ptr))
       endif
       call MPI_BCAST (id, 128, MPI_CHARACTER, 0, MPI_COMM_NEW, mympierror)
       ierr = c2r_comm_init (hip_ctx, int(rank*hip_ndevices, c_int), int(npr*hip_ndevices, c_int), id)
       if (ierr /= 0) call stop_with (c2r_error_text(hip_ctx))
#else
       call stop_with ("npr > 1 in a build without MPI")
#endif
    elseif (hip_ndevices > 1) then
       ierr = c2r_comm_init_local (hip_ctx)
       if (ierr /= 0) call stop_with (c2r_error_text(hip_ctx))
       if (rank == 0) then
          if (c2r_comm_kind (hip_ctx) == 1) then
             libpath = c_null_char
             ierr = c2r_comm_library (libpath, int(size(libpath), c_int))
             nchar = 0
             do while (nchar < size(libpath))
                if (libpath(nchar+1) == c_null_char) exit
                nchar = nchar + 1
             enddo
             write(logf,"(A)") " evolve_ini: the sum over the devices is an ncclAllReduce of"
             write(logf,"(2A)") "   ", transfer(libpath(1:nchar), repeat(" ", nchar))
          else
             write(logf,"(A)") " evolve_ini: one device for all: their sum is made by the library itself"
          endif
       endif
    else
       ! C2RAY_HIP_FORCE_COMM=1: a communicator of one rank, so that a one-GPU run goes through the very calls
       ! (ncclCommInitRank, ncclAllReduce) a multi-rank run makes
       force = 0
       call env_integer ("C2RAY_HIP_FORCE_COMM", force)
       if (force /= 0) then
          ierr = c2r_comm_unique_id (id)
          if (ierr /= 0) call stop_with (c2r_error_text(c_null_ptr))
          ierr = c2r_comm_init (hip_ctx, 0_c_int, 1_c_int, id)
          if (ierr /= 0) call stop_with (c2r_error_text(hip_ctx))
          if (rank == 0) write(logf,*) "evolve_ini: RCCL communicator of one rank (C2RAY_HIP_FORCE_COMM)"
       endif
    endif

  end subroutine setup_communicator

  !> The device code has the numerical parameters of the reference compiled in (c2r_get_constants); a host
  !! built with other values in c2ray_parameters.f90 / abundances.f90 / radiation_sizes.f90 must not run.
  subroutine check_compiled_constants ()

    real(c_double) :: c(32)
    real(kind=dp) :: host(15)
    character(len=28), parameter :: names(15) = (/ "subboxsize                  ", &
         "max_subbox                  ", "abu_he                      ", "abu_c                       ", &
         "epsilon                     ", "convergence_fraction        ", "minimum_fractional_change   ", &
         "minimum_fraction_of_atoms   ", "relative_denergy            ", "minitemp                    ", &
         "NumTau                      ", "minlogtau                   ", "maxlogtau                   ", &
         "NumFreqBnd                  ", "NumheatBin                  " /)
    integer :: n, i

    n = c2r_get_constants (c, 32_c_int)
    host = (/ real(subboxsize,dp), real(max_subbox,dp), real(abu_he,dp), real(abu_c,dp), real(epsilon,dp), &
         real(convergence_fraction,dp), real(minimum_fractional_change,dp), real(minimum_fraction_of_atoms,dp), &
         real(relative_denergy,dp), real(minitemp,dp), real(NumTau,dp), -20.0_dp, 4.0_dp, &
         real(NumFreqBnd,dp), real(NumheatBin,dp) /)
    do i = 1, min(n, 15)
       if (c(i) /= host(i)) then
          write(logf,*) "c2ray_hip: ", trim(names(i)), " of this build is ", host(i), &
               " but the device library was compiled with ", c(i)
          call stop_with ("parameter "//trim(names(i))//" differs from the value compiled into libc2ray_hip")
       endif
    enddo
    if (add_photon_losses) call stop_with ("add_photon_losses = .true. (distribute_photon_losses, "// &
         "evolve_point.F90:546) is not implemented on the device")

  end subroutine check_compiled_constants

  subroutine env_integer (name, value)
    character(len=*), intent(in) :: name
    integer, intent(inout) :: value
    character(len=32) :: text
    integer :: length, status, v, ios
    call get_environment_variable(name, text, length, status)
    if (status == 0 .and. length > 0) then
       read(text(1:length),*,iostat=ios) v
       if (ios == 0) value = v
    endif
  end subroutine env_integer

  !> log the text and end the run on every rank
  subroutine stop_with (text)
    character(len=*), intent(in) :: text
#ifdef MPI
    integer :: mympierror
#endif
    write(logf,*) "c2ray_hip: ", text
    write(*,*) "c2ray_hip: ", text
    flush(logf)
#ifdef MPI
    call MPI_ABORT (MPI_COMM_NEW, 1, mympierror)
#endif
    stop 1
  end subroutine stop_with

end module evolve_data
