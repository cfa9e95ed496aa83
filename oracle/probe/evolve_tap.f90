! TEST INFRASTRUCTURE ONLY (oracle/).  Not part of the product.
!
! A tap on the reference's own hot-path boundary.  The unmodified reference
! driver (files_for_3D/C2Ray.F90:335 `call evolve3D(sim_time,actual_dt,iter_restart)`)
! is linked with `-Wl,--wrap=_QMevolvePevolve3d`, so that its call lands here;
! we dump every host array the path reads (SURVEY.md section 8b), call the real
! evolve3D (files_for_3D/evolve.F90:78), and dump everything it wrote.
! The dumps are the golden vectors of tests/golden/ (see oracle/make_golden.py).
!
! File format (stream, native endian): a sequence of records
!    name(len=16)  kind(int32: 1=int32, 2=float32, 3=float64)  count(int64)  payload
subroutine evolve3d_tap(time, dt, restart) bind(C, name="__wrap__QMevolvePevolve3d")
  use precision, only: dp
  use sizes, only: mesh
  use grid, only: dr, vol
  use material, only: ndens, xh, xhe, temperature_grid, isothermal, temper_val, clumping
  use material, only: coldensh_LLS
  use c2ray_parameters, only: use_LLS
  use sourceprops, only: NumSrc, srcpos, NormFlux
  use radiation_sed_parameters, only: S_star
#ifdef PL
  use radiation_sed_parameters, only: pl_S_star
  use sourceprops, only: NormFluxPL
#endif
#ifdef QUASARS
  use radiation_sed_parameters, only: qpl_S_star
  use sourceprops, only: NormFluxQPL
#endif
  use cosmology, only: zred
  use cosmology_parameters, only: H0, Omega0
  use cgsconstants, only: arech0, brech0, areche0, breche0, oreche0, areche1, breche1, &
       treche1, colli_HI, colli_HeI, colli_HeII, v
  use evolve_data, only: phih_grid, phihe_grid, phiheat, xh_av, xhe_av, xh_intermed, &
       xhe_intermed, coldensh_out, coldenshe_out, photon_loss_all
  use evolve_source, only: sum_nbox_all
  use photonstatistics, only: photon_loss
  implicit none
  real(kind=dp), intent(in) :: time, dt
  integer, intent(in) :: restart
  interface
     subroutine real_evolve3d(time, dt, restart) bind(C, name="__real__QMevolvePevolve3d")
       import :: dp
       real(kind=dp), intent(in) :: time, dt
       integer, intent(in) :: restart
     end subroutine real_evolve3d
  end interface
  integer, save :: ncall = 0
  integer :: u, iso
  character(len=64) :: fname

  ncall = ncall + 1
  if (ncall == 1) call dump_tables_and_vectors()

  write(fname, "(A,I4.4,A)") "results/tap_", ncall, "_in.bin"
  open(newunit=u, file=trim(fname), access="stream", form="unformatted", status="replace")
  call put_i(u, "mesh", mesh, 3)
  call put_d(u, "dt", (/dt/), 1)
  call put_d(u, "zred", (/zred/), 1)
  call put_d(u, "H0", (/H0/), 1)
  call put_d(u, "Omega0", (/Omega0/), 1)
  call put_d(u, "dr", dr, 3)
  call put_d(u, "vol", (/vol/), 1)
  call put_i(u, "NumSrc", (/NumSrc/), 1)
  call put_i(u, "srcpos", srcpos, 3*NumSrc)
  call put_d(u, "NormFlux", NormFlux(1:NumSrc), NumSrc)
  call put_d(u, "S_star", (/S_star/), 1)
#ifdef PL
  call put_d(u, "NormFluxPL", NormFluxPL(1:NumSrc), NumSrc)
  call put_d(u, "pl_S_star", (/pl_S_star/), 1)
#endif
#ifdef QUASARS
  call put_d(u, "NormFluxQPL", NormFluxQPL(1:NumSrc), NumSrc)
  call put_d(u, "qpl_S_star", (/qpl_S_star/), 1)
#endif
  iso = 0
  if (isothermal) iso = 1
  call put_i(u, "isothermal", (/iso/), 1)
  call put_d(u, "temper_val", (/temper_val/), 1)
  call put_f(u, "clumping", (/clumping/), 1)
  if (use_LLS) call put_d(u, "coldensh_LLS", (/coldensh_LLS/), 1)
  call put_d(u, "reccoef", (/arech0, brech0, areche0, breche0, oreche0, areche1, breche1, &
       treche1, colli_HI, colli_HeI, colli_HeII, v/), 12)
  call put_d(u, "ndens", ndens, size(ndens))
  call put_d(u, "xh", xh, size(xh))
  call put_d(u, "xhe", xhe, size(xhe))
  if (.not. isothermal) call put_f(u, "temperature", temperature_grid, size(temperature_grid))
  close(u)

  call real_evolve3d(time, dt, restart)

  write(fname, "(A,I4.4,A)") "results/tap_", ncall, "_out.bin"
  open(newunit=u, file=trim(fname), access="stream", form="unformatted", status="replace")
  call put_d(u, "xh", xh, size(xh))
  call put_d(u, "xhe", xhe, size(xhe))
  if (.not. isothermal) call put_f(u, "temperature", temperature_grid, size(temperature_grid))
  call put_d(u, "phih_grid", phih_grid, size(phih_grid))
  call put_d(u, "phihe_grid", phihe_grid, size(phihe_grid))
  call put_d(u, "phiheat", phiheat, size(phiheat))
  call put_d(u, "xh_av", xh_av, size(xh_av))
  call put_d(u, "xhe_av", xhe_av, size(xhe_av))
  call put_d(u, "xh_intermed", xh_intermed, size(xh_intermed))
  call put_d(u, "xhe_intermed", xhe_intermed, size(xhe_intermed))
  call put_d(u, "coldensh_out", coldensh_out, size(coldensh_out))
  call put_d(u, "coldenshe_out", coldenshe_out, size(coldenshe_out))
  call put_d(u, "photon_loss", photon_loss, size(photon_loss))
  call put_d(u, "photon_loss_all", photon_loss_all, size(photon_loss_all))
  call put_i(u, "sum_nbox_all", (/sum_nbox_all/), 1)
  call put_d(u, "reccoef", (/arech0, brech0, areche0, breche0, oreche0, areche1, breche1, &
       treche1, colli_HI, colli_HeI, colli_HeII, v/), 12)
  close(u)

contains

  subroutine put_hdr(u, name, kind, n)
    integer, intent(in) :: u, kind, n
    character(len=*), intent(in) :: name
    character(len=16) :: nm
    nm = name
    write(u) nm, int(kind, 4), int(n, 8)
  end subroutine put_hdr

  subroutine put_i(u, name, a, n)
    integer, intent(in) :: u, n
    character(len=*), intent(in) :: name
    integer, intent(in) :: a(*)
    call put_hdr(u, name, 1, n)
    write(u) a(1:n)
  end subroutine put_i

  subroutine put_f(u, name, a, n)
    integer, intent(in) :: u, n
    character(len=*), intent(in) :: name
    real(kind=4), intent(in) :: a(*)
    call put_hdr(u, name, 2, n)
    write(u) a(1:n)
  end subroutine put_f

  subroutine put_d(u, name, a, n)
    integer, intent(in) :: u, n
    character(len=*), intent(in) :: name
    real(kind=dp), intent(in) :: a(*)
    call put_hdr(u, name, 3, n)
    write(u) a(1:n)
  end subroutine put_d

  ! ---------------------------------------------------------------------------
  ! Constants, tables and function-level input/output vectors, produced by
  ! *calling the reference's own compiled routines*.
  subroutine dump_tables_and_vectors()
    use mathconstants, only: pi
    use abundances, only: abu_he, abu_c, mu
    use atomic, only: gamma1
    use cgsconstants, only: hplanck, k_B, m_p, temph0, temphe, colh0, colhe, ev2k, ev2fr, &
         eth0, ethe, ini_rec_colion_factors
    use cgsphotoconstants, only: sigma_HI_at_ion_freq, sigma_HeI_at_ion_freq, &
         sigma_HeII_at_ion_freq, ion_freq_HI, ion_freq_HeI, ion_freq_HeII, &
         sigma_H_heth, sigma_H_heLya, sigma_He_heLya, sigma_He_he2, sigma_H_he2
    use c2ray_parameters, only: epsilon, convergence_fraction, minimum_fractional_change, &
         minimum_fraction_of_atoms, minitemp, relative_denergy, subboxsize, max_subbox
    use radiation_sizes, only: NumFreqBnd, NumheatBin, NumTau, sigma_HI, sigma_HeI, sigma_HeII, &
         f1ion_HI, f1ion_HeI, f1ion_HeII, f2ion_HI, f2ion_HeI, f2ion_HeII, &
         f1heat_HI, f1heat_HeI, f1heat_HeII, f2heat_HI, f2heat_HeI, f2heat_HeII, &
         freq_min, freq_max, delta_freq, &
         cross_section_HI_powerlaw_index, cross_section_HeI_powerlaw_index, &
         cross_section_HeII_powerlaw_index
    use radiation_tables, only: bb_photo_thick_table, bb_photo_thin_table, &
         bb_heat_thick_table, bb_heat_thin_table, bb_FreqBnd_UpperLimit, minlogtau, dlogtau
    use radiation_sed_parameters, only: T_eff, R_star, L_star, R_star2, h_over_kT
    use radiation_tables, only: tau
    use radiation_sizes, only: NumFreq
    use romberg, only: romw
    use cgsconstants, only: two_pi_over_c_square
#ifdef PL
    use radiation_sed_parameters, only: pl_scaling, pl_index
#endif
#ifdef QUASARS
    use radiation_sed_parameters, only: qpl_scaling, qpl_index
#endif
#ifdef PL
    use radiation_tables, only: pl_photo_thick_table, pl_photo_thin_table, pl_heat_thick_table, &
         pl_heat_thin_table, pl_FreqBnd_UpperLimit, pl_FreqBnd_LowerLimit
#endif
#ifdef QUASARS
    use radiation_tables, only: qpl_photo_thick_table, qpl_photo_thin_table, qpl_heat_thick_table, &
         qpl_heat_thin_table, qpl_FreqBnd_UpperLimit, qpl_FreqBnd_LowerLimit
#endif
    use radiation_photoionrates, only: photrates, photoion_rates
    use doric_module, only: doric, prepare_doric_factors
    use thermalevolution, only: thermal
    use radiative_cooling, only: coolin
    use tped, only: electrondens
    use material, only: ionstates
    use column_density, only: cinterp

    integer :: u, i, n, k, ncase
    real(kind=dp) :: r(12), x, t, dtl, de, nd, yf, zf, y2a, y2b, NH, NHe(0:1)
    real(kind=dp) :: tend, tavg, cin(6), vph, ist
    real(kind=dp), allocatable :: buf(:)
    type(photrates) :: phi
    type(ionstates) :: ion
    logical :: iso_save
    real :: clump_save
    real(kind=dp) :: rc_save(12)
    integer :: seed_state
    real(kind=dp) :: u1, u2, u3, u4, u5, u6, u7, u8

    open(newunit=u, file="results/tables.bin", access="stream", form="unformatted", status="replace")
    call put_d(u, "consts", (/pi, abu_he, abu_c, mu, gamma1, hplanck, k_B, m_p, temph0, &
         temphe(0), temphe(1), colh0, colhe(0), colhe(1), ev2k, ev2fr, eth0, ethe(0), ethe(1), &
         sigma_HI_at_ion_freq, sigma_HeI_at_ion_freq, sigma_HeII_at_ion_freq, &
         ion_freq_HI, ion_freq_HeI, ion_freq_HeII, sigma_H_heth, sigma_H_heLya, sigma_He_heLya, &
         sigma_He_he2, sigma_H_he2, epsilon, convergence_fraction, minimum_fractional_change, &
         minimum_fraction_of_atoms, minitemp, relative_denergy, minlogtau, dlogtau, &
         T_eff, R_star, L_star, S_star, H0, Omega0/), 44)
    call put_i(u, "ints", (/NumFreqBnd, NumheatBin, NumTau, bb_FreqBnd_UpperLimit, subboxsize, &
         max_subbox/), 6)
    call put_d(u, "sigma_HI", sigma_HI, size(sigma_HI))
    call put_d(u, "sigma_HeI", sigma_HeI, size(sigma_HeI))
    call put_d(u, "sigma_HeII", sigma_HeII, size(sigma_HeII))
    call put_d(u, "freq_min", freq_min, size(freq_min))
    call put_d(u, "freq_max", freq_max, size(freq_max))
    call put_d(u, "delta_freq", delta_freq, size(delta_freq))
    call put_d(u, "pl_index_HI", cross_section_HI_powerlaw_index, size(cross_section_HI_powerlaw_index))
    call put_d(u, "pl_index_HeI", cross_section_HeI_powerlaw_index, size(cross_section_HeI_powerlaw_index))
    call put_d(u, "pl_index_HeII", cross_section_HeII_powerlaw_index, size(cross_section_HeII_powerlaw_index))
    if (allocated(f1ion_HI)) then
       ! all twelve are dimension(NumBndin1+1:NumFreqBnd) = (2:47)
       call put_d(u, "f1ion_HI", f1ion_HI, size(f1ion_HI))
       call put_d(u, "f1ion_HeI", f1ion_HeI, size(f1ion_HeI))
       call put_d(u, "f1ion_HeII", f1ion_HeII, size(f1ion_HeII))
       call put_d(u, "f2ion_HI", f2ion_HI, size(f2ion_HI))
       call put_d(u, "f2ion_HeI", f2ion_HeI, size(f2ion_HeI))
       call put_d(u, "f2ion_HeII", f2ion_HeII, size(f2ion_HeII))
       call put_d(u, "f1heat_HI", f1heat_HI, size(f1heat_HI))
       call put_d(u, "f1heat_HeI", f1heat_HeI, size(f1heat_HeI))
       call put_d(u, "f1heat_HeII", f1heat_HeII, size(f1heat_HeII))
       call put_d(u, "f2heat_HI", f2heat_HI, size(f2heat_HI))
       call put_d(u, "f2heat_HeI", f2heat_HeI, size(f2heat_HeI))
       call put_d(u, "f2heat_HeII", f2heat_HeII, size(f2heat_HeII))
    endif
    ! what spec_integration (radiation_tables.f90:172-422) starts from, besides the band vectors above
    call put_d(u, "sed_setup", (/R_star2, h_over_kT, two_pi_over_c_square/), 3)
    call put_d(u, "tau", tau, size(tau))
    call put_d(u, "romw9", romw(0:NumFreq,9), NumFreq+1)
#ifdef PL
    call put_d(u, "pl_setup", (/pl_scaling, pl_index/), 2)
#endif
#ifdef QUASARS
    call put_d(u, "qpl_setup", (/qpl_scaling, qpl_index/), 2)
#endif
    call put_d(u, "photo_thick", bb_photo_thick_table, size(bb_photo_thick_table))
    call put_d(u, "photo_thin", bb_photo_thin_table, size(bb_photo_thin_table))
    if (allocated(bb_heat_thick_table)) then
       call put_d(u, "heat_thick", bb_heat_thick_table, size(bb_heat_thick_table))
       call put_d(u, "heat_thin", bb_heat_thin_table, size(bb_heat_thin_table))
    endif
#ifdef PL
    call put_i(u, "pl_limits", (/pl_FreqBnd_LowerLimit, pl_FreqBnd_UpperLimit/), 2)
    call put_d(u, "pl_photo_thick", pl_photo_thick_table, size(pl_photo_thick_table))
    call put_d(u, "pl_photo_thin", pl_photo_thin_table, size(pl_photo_thin_table))
    if (allocated(pl_heat_thick_table)) then
       call put_d(u, "pl_heat_thick", pl_heat_thick_table, size(pl_heat_thick_table))
       call put_d(u, "pl_heat_thin", pl_heat_thin_table, size(pl_heat_thin_table))
    endif
#endif
#ifdef QUASARS
    call put_i(u, "qpl_limits", (/qpl_FreqBnd_LowerLimit, qpl_FreqBnd_UpperLimit/), 2)
    call put_d(u, "qpl_photo_thick", qpl_photo_thick_table, size(qpl_photo_thick_table))
    call put_d(u, "qpl_photo_thin", qpl_photo_thin_table, size(qpl_photo_thin_table))
    if (allocated(qpl_heat_thick_table)) then
       call put_d(u, "qpl_heat_thick", qpl_heat_thick_table, size(qpl_heat_thick_table))
       call put_d(u, "qpl_heat_thin", qpl_heat_thin_table, size(qpl_heat_thin_table))
    endif
#endif
    if (.not. isothermal) then
       ! cooling curve sampled through the public interface: coolin with unit densities
       ! and one species switched on recovers each table value*(abundance) exactly
       allocate(buf(5*801))
       do i = 1, 801
          t = 10.0_dp**(1.0_dp + 0.01_dp*real(i-1, dp))
          buf(i)        = coolin(1.0_dp, 1.0_dp, (/1.0_dp, 0.0_dp/), (/0.0_dp, 0.0_dp, 0.0_dp/), t)
          buf(801+i)    = coolin(1.0_dp, 1.0_dp, (/0.0_dp, 1.0_dp/), (/0.0_dp, 0.0_dp, 0.0_dp/), t)
          buf(2*801+i)  = coolin(1.0_dp, 1.0_dp, (/0.0_dp, 0.0_dp/), (/1.0_dp, 0.0_dp, 0.0_dp/), t)
          buf(3*801+i)  = coolin(1.0_dp, 1.0_dp, (/0.0_dp, 0.0_dp/), (/0.0_dp, 1.0_dp, 0.0_dp/), t)
          buf(4*801+i)  = coolin(1.0_dp, 1.0_dp, (/0.0_dp, 0.0_dp/), (/0.0_dp, 0.0_dp, 1.0_dp/), t)
       enddo
       call put_d(u, "coolin_probe", buf, 5*801)
       deallocate(buf)
    endif
    close(u)

    ! ---- function-level vectors ------------------------------------------------
    open(newunit=u, file="results/funcvec.bin", access="stream", form="unformatted", status="replace")
    rc_save = (/arech0, brech0, areche0, breche0, oreche0, areche1, breche1, &
         treche1, colli_HI, colli_HeI, colli_HeII, v/)

    ! ini_rec_colion_factors: T -> 12 coefficients (cgsconstants.f90:140)
    ncase = 40
    allocate(buf(13*ncase))
    do i = 1, ncase
       t = 10.0_dp**(1.5_dp + 5.0_dp*real(i-1, dp)/real(ncase-1, dp))
       if (i == 7) t = 8999.0_dp
       if (i == 8) t = 9000.0_dp
       if (i == 9) t = 1.0e4_dp
       call ini_rec_colion_factors(t)
       buf(13*(i-1)+1) = t
       buf(13*(i-1)+2:13*i) = (/arech0, brech0, areche0, breche0, oreche0, areche1, breche1, &
            treche1, colli_HI, colli_HeI, colli_HeII, v/)
    enddo
    call put_d(u, "reccoef_T", buf, 13*ncase)
    deallocate(buf)

    ! photoion_rates: 9 inputs -> 21 outputs, isothermal and with heating
    seed_state = 12345
    ncase = 400
    allocate(buf(ncase*(9+21)))
    iso_save = isothermal
    do k = 0, 1
       isothermal = (k == 0)
       if (k == 1 .and. .not. allocated(bb_heat_thick_table)) exit
       do i = 1, ncase
          u1 = lcg(seed_state); u2 = lcg(seed_state); u3 = lcg(seed_state); u4 = lcg(seed_state)
          u5 = lcg(seed_state); u6 = lcg(seed_state); u7 = lcg(seed_state); u8 = lcg(seed_state)
          ! incoming columns: log-uniform 1e12..1e23 (a few exactly zero = source cell)
          cin(1) = 10.0_dp**(12.0_dp + 11.0_dp*u1)
          cin(3) = 10.0_dp**(11.0_dp + 11.0_dp*u2)
          cin(5) = 10.0_dp**(9.0_dp + 12.0_dp*u3)
          if (mod(i, 23) == 0) then
             cin(1) = 0.0_dp; cin(3) = 0.0_dp; cin(5) = 0.0_dp
          endif
          ! cell columns: log-uniform 1e8..1e21 (thin and thick cells)
          cin(2) = cin(1) + 10.0_dp**(8.0_dp + 13.0_dp*u4)
          cin(4) = cin(3) + 10.0_dp**(7.0_dp + 13.0_dp*u5)
          cin(6) = cin(5) + 10.0_dp**(3.0_dp + 16.0_dp*u6)
          vph = 10.0_dp**(60.0_dp + 12.0_dp*u7)
          ist = 10.0_dp**(-6.0_dp*u8)
          if (mod(i, 17) == 0) ist = 1.0e-20_dp
          phi = photoion_rates(cin(1), cin(2), cin(3), cin(4), cin(5), cin(6), vph, 1, ist)
          n = (i-1)*30
          buf(n+1:n+6) = cin
          buf(n+7) = vph
          buf(n+8) = ist
          buf(n+9) = NormFlux(1)
          buf(n+10:n+30) = (/phi%photo_cell_HI, phi%photo_cell_HeI, phi%photo_cell_HeII, &
               phi%heat_cell_HI, phi%heat_cell_HeI, phi%heat_cell_HeII, &
               phi%photo_in_HI, phi%photo_in_HeI, phi%photo_in_HeII, &
               phi%heat_in_HI, phi%heat_in_HeI, phi%heat_in_HeII, &
               phi%photo_out_HI, phi%photo_out_HeI, phi%photo_out_HeII, &
               phi%heat_out_HI, phi%heat_out_HeI, phi%heat_out_HeII, &
               phi%heat, phi%photo_in, phi%photo_out/)
       enddo
       if (k == 0) call put_d(u, "photoion_iso", buf, ncase*30)
       if (k == 1) call put_d(u, "photoion_heat", buf, ncase*30)
    enddo
    isothermal = iso_save
    deallocate(buf)

    ! prepare_doric_factors + doric (doric.f90:35,317): inputs -> ion(15) out
    ncase = 300
    allocate(buf(ncase*(4+3+12+1+15+4+15)))
    clump_save = clumping
    do i = 1, ncase
       u1 = lcg(seed_state); u2 = lcg(seed_state); u3 = lcg(seed_state); u4 = lcg(seed_state)
       u5 = lcg(seed_state); u6 = lcg(seed_state); u7 = lcg(seed_state); u8 = lcg(seed_state)
       t = 10.0_dp**(2.0_dp + 3.0_dp*u1)
       call ini_rec_colion_factors(t)
       dtl = 10.0_dp**(10.0_dp + 5.0_dp*u2)
       nd = 10.0_dp**(-5.0_dp + 4.0_dp*u3)
       phi%photo_cell_HI = 10.0_dp**(-20.0_dp + 10.0_dp*u4)
       phi%photo_cell_HeI = 10.0_dp**(-21.0_dp + 10.0_dp*u5)
       phi%photo_cell_HeII = 10.0_dp**(-23.0_dp + 11.0_dp*u6)
       if (mod(i, 11) == 0) then
          phi%photo_cell_HI = 0.0_dp; phi%photo_cell_HeI = 0.0_dp; phi%photo_cell_HeII = 0.0_dp
       endif
       x = 10.0_dp**(-8.0_dp*u7)
       if (mod(i, 13) == 0) x = 1.0e-20_dp
       ion%h_old = (/1.0_dp - x, x/)
       ion%he_old = (/1.0_dp - x - 0.1_dp*x*u8, x, 0.1_dp*x*u8/)
       if (mod(i, 13) == 0) ion%he_old = (/1.0_dp - 2.0e-20_dp, 1.0e-20_dp, 1.0e-20_dp/)
       ion%h = ion%h_old; ion%he = ion%he_old
       ion%h_av = ion%h_old; ion%he_av = ion%he_old
       clumping = 1.0
       if (mod(i, 7) == 0) clumping = 3.5
       de = electrondens(nd, ion%h_av, ion%he_av)
       NH = ion%h(0)*nd*1.0_dp*(1.0_dp - abu_he)
       NHe(0) = ion%he(0)*nd*1.0_dp*abu_he
       NHe(1) = ion%he(1)*nd*1.0_dp*abu_he
       call prepare_doric_factors(NH, NHe, yf, zf, y2a, y2b)
       n = (i-1)*54
       buf(n+1:n+4) = (/dtl, de, nd, real(clumping, dp)/)
       buf(n+5:n+7) = (/phi%photo_cell_HI, phi%photo_cell_HeI, phi%photo_cell_HeII/)
       buf(n+8:n+19) = (/arech0, brech0, areche0, breche0, oreche0, areche1, breche1, &
            treche1, colli_HI, colli_HeI, colli_HeII, v/)
       buf(n+20) = t
       buf(n+21:n+35) = (/ion%h, ion%he, ion%h_av, ion%he_av, ion%h_old, ion%he_old/)
       buf(n+36:n+39) = (/yf, zf, y2a, y2b/)
       call doric(dtl, de, nd, ion, phi, yf, zf, y2a, y2b)
       buf(n+40:n+54) = (/ion%h, ion%he, ion%h_av, ion%he_av, ion%h_old, ion%he_old/)
    enddo
    call put_d(u, "doric", buf, ncase*54)
    deallocate(buf)
    clumping = clump_save

    ! thermal (thermal.f90:22) -- needs the cooling tables (non-isothermal runs only)
    if (.not. isothermal) then
       ncase = 200
       allocate(buf(ncase*(6+15+4)))
       do i = 1, ncase
          u1 = lcg(seed_state); u2 = lcg(seed_state); u3 = lcg(seed_state); u4 = lcg(seed_state)
          u5 = lcg(seed_state); u6 = lcg(seed_state)
          tend = 10.0_dp**(1.0_dp + 4.5_dp*u1)
          dtl = 10.0_dp**(11.0_dp + 4.0_dp*u2)
          nd = 10.0_dp**(-5.0_dp + 4.0_dp*u3)
          phi%heat = 10.0_dp**(-32.0_dp + 10.0_dp*u4)
          if (mod(i, 9) == 0) phi%heat = 0.0_dp
          x = 10.0_dp**(-6.0_dp*u5)
          ion%h_old = (/1.0_dp - x, x/)
          ion%he_old = (/1.0_dp - x, 0.7_dp*x, 0.3_dp*x/)
          x = min(1.0_dp - 1.0e-9_dp, x*(1.0_dp + 5.0_dp*u6))
          ion%h_av = (/1.0_dp - x, x/)
          ion%he_av = (/1.0_dp - x, 0.6_dp*x, 0.4_dp*x/)
          x = min(1.0_dp - 1.0e-9_dp, x*1.2_dp)
          ion%h = (/1.0_dp - x, x/)
          ion%he = (/1.0_dp - x, 0.5_dp*x, 0.5_dp*x/)
          de = electrondens(nd, ion%h_av, ion%he_av)
          n = (i-1)*25
          buf(n+1:n+6) = (/dtl, tend, de, nd, phi%heat, zred/)
          buf(n+7:n+21) = (/ion%h, ion%he, ion%h_av, ion%he_av, ion%h_old, ion%he_old/)
          tavg = -1.0_dp
          call thermal(dtl, tend, tavg, de, nd, ion, phi)
          buf(n+22:n+25) = (/tend, tavg, 0.0_dp, 0.0_dp/)
       enddo
       call put_d(u, "thermal", buf, ncase*25)
       deallocate(buf)
    endif

    ! restore the module-global coefficients we disturbed
    arech0 = rc_save(1); brech0 = rc_save(2); areche0 = rc_save(3); breche0 = rc_save(4)
    oreche0 = rc_save(5); areche1 = rc_save(6); breche1 = rc_save(7); treche1 = rc_save(8)
    colli_HI = rc_save(9); colli_HeI = rc_save(10); colli_HeII = rc_save(11); v = rc_save(12)
    close(u)
  end subroutine dump_tables_and_vectors

  ! tiny deterministic generator so the vectors do not depend on the compiler's RNG
  function lcg(state) result(r)
    integer, intent(inout) :: state
    real(kind=dp) :: r
    integer(kind=8) :: s
    s = int(state, 8)
    s = mod(s*1103515245_8 + 12345_8, 2147483648_8)
    state = int(s, 4)
    r = real(s, dp)/2147483648.0_dp
  end function lcg

end subroutine evolve3d_tap
