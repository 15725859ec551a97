!> Drop-in replacement of the reference's `evolve` module (evolve.F90) for C2-Ray, backed by the
!! MI355X HIP library (include/c2ray_hip.h) through ISO_C_BINDING.
!!
!! Link position: in the reference's makefile_core:31 replace
!!     EVOLVE = evolve_data.o column_density.o evolve_point.o evolve_source.o master_slave.o evolve.o
!! by
!!     EVOLVE = evolve_data.o evolve_hip.o            (and add  -L<dir> -lc2ray_hip  to the link line)
!! `evolve_data` (the arrays C2Ray.F90:61 and output.F90:31 use) stays the reference's own file.
!!
!! The public surface is the one the driver uses: `evolve3D(time,dt,restart)` (evolve.F90:83, called
!! from C2Ray.F90:379), plus `sum_nbox`/`sum_nbox_all` (evolve_source.F90:45-46).  All inputs are
!! taken from the same module-global arrays the reference routine reads, by `use` association; the
!! HIP side never keeps a host pointer after the call returns.
module evolve

  use, intrinsic :: iso_c_binding
  use precision, only: dp
  use my_mpi, only: rank
  use file_admin, only: logf, timefile
  use clocks, only: timestamp_wallclock
  use sizes, only: mesh
  use grid, only: dr, vol
  use density_module, only: ndens
  use ionfractions_module, only: xh
  use temperature_module, only: temper_val
  use clumping_module, only: clumping
  use lls_module, only: coldensh_LLS
  use sourceprops, only: NumSrc, srcpos, NormFlux_stellar
  use radiation_sizes, only: NumTau
  use radiation_tables, only: stellar_photo_thick_table, stellar_photo_thin_table, minlogtau, dlogtau
  use radiation_sed_parameters, only: S_star
  use cgsphotoconstants, only: sigma_HI_at_ion_freq
  use cgsconstants, only: bh00, albpow, colh0, temph0
  use mathconstants, only: pi
  use abundances, only: abu_c
  use c2ray_parameters, only: epsilon, convergence_fraction, minimum_fractional_change, &
       minimum_fraction_of_atoms, loss_fraction, subboxsize, max_subbox
  use photonstatistics, only: photon_loss, LLS_loss, state_before, calculate_photon_statistics, &
       report_photonstatistics, update_grandtotal_photonstatistics
  use evolve_data, only: phih_grid, xh_av, xh_intermed, photon_loss_all

  implicit none

  save

  private

  integer, parameter :: C2R_MAX_ITER_LOG = 128

  !> mirror of struct c2r_params (include/c2ray_hip.h)
  type, bind(C) :: c2r_params
     integer(c_int32_t) :: mesh(3), device, subboxsize, max_subbox, numtau, max_outer_iter, &
          max_chem_iter, deterministic_rates
     real(c_double) :: epsilon, convergence_fraction, minimum_fractional_change, &
          minimum_fraction_of_atoms, loss_fraction, max_coldensh, tau_photo_limit, sigma_HI, &
          minlogtau, dlogtau, weight_floor, sqrt2, sqrt3, pi, abu_c, bh00, albpow, colh0, temph0, S_star
     integer(c_size_t) :: scratch_bytes
  end type c2r_params

  !> mirror of struct c2r_report
  type, bind(C) :: c2r_report
     integer(c_int32_t) :: niter, converged
     integer(c_int64_t) :: conv_flag, conv_criterion, sum_nbox_all, visited
     real(c_double) :: photon_loss_all, seconds_sweep, seconds_chem
     integer(c_int32_t) :: chem_not_converged, reserved0
     integer(c_int64_t) :: it_conv_flag(C2R_MAX_ITER_LOG), it_sum_nbox(C2R_MAX_ITER_LOG)
     real(c_double) :: it_rel_change_xh1(C2R_MAX_ITER_LOG), it_rel_change_xh0(C2R_MAX_ITER_LOG), &
          it_sum_xh1(C2R_MAX_ITER_LOG)
     real(c_double) :: h0_before, h1_before, h0_after, h1_after, totrec, totcollisions, dh0, &
          total_ion, totalsrc, photcons, it_photcons(C2R_MAX_ITER_LOG)
  end type c2r_report

  interface
     integer(c_int) function c2r_default_params(p) bind(C, name="c2r_default_params")
       import :: c_int, c2r_params
       type(c2r_params), intent(out) :: p
     end function c2r_default_params
     integer(c_int) function c2r_create(ctx, p) bind(C, name="c2r_create")
       import :: c_int, c_ptr, c2r_params
       type(c_ptr), intent(out) :: ctx
       type(c2r_params), intent(in) :: p
     end function c2r_create
     subroutine c2r_destroy(ctx) bind(C, name="c2r_destroy")
       import :: c_ptr
       type(c_ptr), value :: ctx
     end subroutine c2r_destroy
     function c2r_last_error(ctx) bind(C, name="c2r_last_error") result(msg)
       import :: c_ptr
       type(c_ptr), value :: ctx
       type(c_ptr) :: msg
     end function c2r_last_error
     integer(c_int) function c2r_set_tables(ctx, thick, thin, n) bind(C, name="c2r_set_tables")
       import :: c_int, c_ptr, c_double, c_int32_t
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: thick(*), thin(*)
       integer(c_int32_t), value :: n
     end function c2r_set_tables
     integer(c_int) function c2r_set_step(ctx, dr, vol, coldensh_LLS, clumping, temper) &
          bind(C, name="c2r_set_step")
       import :: c_int, c_ptr, c_double, c_float
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: dr(3)
       real(c_double), value :: vol, coldensh_LLS, temper
       real(c_float), value :: clumping
     end function c2r_set_step
     integer(c_int) function c2r_set_sources(ctx, srcpos, normflux, nsrc) bind(C, name="c2r_set_sources")
       import :: c_int, c_ptr, c_double, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), intent(in) :: srcpos(3,*)
       real(c_double), intent(in) :: normflux(*)
       integer(c_int32_t), value :: nsrc
     end function c2r_set_sources
     integer(c_int) function c2r_evolve3d(ctx, dt, ndens, xh, xh_av, xh_intermed, phih_grid, rep) &
          bind(C, name="c2r_evolve3d")
       import :: c_int, c_ptr, c_double, c_float, c2r_report
       type(c_ptr), value :: ctx
       real(c_double), value :: dt
       real(c_float), intent(in) :: ndens(*)
       real(c_double), intent(inout) :: xh(*), xh_av(*), xh_intermed(*), phih_grid(*)
       type(c2r_report), intent(out) :: rep
     end function c2r_evolve3d
  end interface

  type(c_ptr) :: ctx = c_null_ptr
  integer, public :: sum_nbox      !< sum of all nboxes (this process)      evolve_source.F90:45
  integer, public :: sum_nbox_all  !< sum of all nboxes (all processes)     evolve_source.F90:46
  type(c2r_report), public :: last_report

  public :: evolve3D, evolve_hip_end

contains

  subroutine check(rc, what)
    integer(c_int), intent(in) :: rc
    character(len=*), intent(in) :: what
    character(kind=c_char), pointer :: cmsg(:)
    type(c_ptr) :: p
    integer :: i
    if (rc == 0) return
    write(logf,*) "c2ray_hip: ", what, " failed with status ", rc
    if (c_associated(ctx)) then
       p = c2r_last_error(ctx)
       if (c_associated(p)) then
          call c_f_pointer(p, cmsg, (/ 512 /))
          do i = 1, 512
             if (cmsg(i) == c_null_char) exit
          enddo
          write(logf,*) "c2ray_hip: ", cmsg(1:i-1)
       endif
    endif
    flush(logf)
    stop "c2ray_hip failure"
  end subroutine check

  !> evolve_ini counterpart for the device side: created lazily on the first evolve3D
  subroutine evolve_hip_ini()
    type(c2r_params) :: p
    call check(c2r_default_params(p), "c2r_default_params")
    p%mesh = mesh
    p%device = 0
    ! the parameters this build of the driver was compiled with
    p%subboxsize = subboxsize; p%max_subbox = max_subbox; p%numtau = NumTau
    p%epsilon = epsilon; p%convergence_fraction = convergence_fraction
    p%minimum_fractional_change = minimum_fractional_change
    p%minimum_fraction_of_atoms = minimum_fraction_of_atoms
    p%loss_fraction = loss_fraction
    p%sigma_HI = sigma_HI_at_ion_freq
    p%minlogtau = minlogtau; p%dlogtau = dlogtau
    p%pi = pi; p%abu_c = abu_c
    p%bh00 = bh00; p%albpow = albpow; p%colh0 = colh0; p%temph0 = temph0
    p%S_star = S_star
    call check(c2r_create(ctx, p), "c2r_create")
    call check(c2r_set_tables(ctx, stellar_photo_thick_table(:,1), stellar_photo_thin_table(:,1), &
         int(NumTau+1, c_int32_t)), "c2r_set_tables")
  end subroutine evolve_hip_ini

  subroutine evolve_hip_end()
    if (c_associated(ctx)) call c2r_destroy(ctx)
    ctx = c_null_ptr
  end subroutine evolve_hip_end

  !> Evolve the entire grid over a time step dt (same contract as evolve.F90:83)
  subroutine evolve3D (time,dt,restart)

    real(kind=dp),intent(in) :: time !< time
    real(kind=dp),intent(in) :: dt !< time step
    integer,intent(in) :: restart !< restart flag (iteration dumps are not supported here)

    integer :: k
    real(kind=dp) :: ncell

    if (restart /= 0) then
       write(logf,*) "c2ray_hip: restart from iteration dump is not supported by the HIP evolve module"
       stop "c2ray_hip: restart /= 0"
    endif
    if (.not. c_associated(ctx)) call evolve_hip_ini()

    call state_before (xh)                                            ! evolve.F90:136

    call check(c2r_set_step(ctx, dr, vol, coldensh_LLS, real(clumping, c_float), &
         real(temper_val, c_double)), "c2r_set_step")
    call check(c2r_set_sources(ctx, srcpos, NormFlux_stellar(1:NumSrc), int(NumSrc, c_int32_t)), &
         "c2r_set_sources")

    if (rank == 0) write(timefile,"(A,F8.1)") "Time before starting iteration: ", timestamp_wallclock ()

    call check(c2r_evolve3d(ctx, dt, ndens, xh, xh_av, xh_intermed, phih_grid, last_report), &
         "c2r_evolve3d")

    ! what the reference logs per outer iteration (evolve.F90:205-210, 249-251, 559-566)
    ncell = real(mesh(1),dp)*real(mesh(2),dp)*real(mesh(3),dp)
    if (rank == 0) then
       do k = 1, min(last_report%niter, C2R_MAX_ITER_LOG)
          write(logf,*) "Average number of subboxes: ", real(last_report%it_sum_nbox(k))/real(NumSrc)
          write(logf,*) "Number of non-converged points: ", last_report%it_conv_flag(k)
          write(logf,*) "Intermediate result for mean H ionization fraction: ", &
               last_report%it_sum_xh1(k)/ncell
          write(logf,*) "Convergence tests: "
          write(logf,*) "   Test 1 values: ", last_report%it_conv_flag(k), last_report%conv_criterion
          write(logf,*) "   Test 2 values: ", last_report%it_rel_change_xh1(k), &
               last_report%it_rel_change_xh0(k), convergence_fraction
       enddo
       if (last_report%converged /= 0) then
          write(logf,*) "Multiple sources convergence reached"
       else
          write(logf,*) "Multiple sources not converging"
       endif
       write(timefile,"(A,I3,A,F8.1)") "Time after iteration ", last_report%niter, " : ", &
            timestamp_wallclock ()
    endif

    sum_nbox = int(last_report%sum_nbox_all)
    sum_nbox_all = sum_nbox
    photon_loss_all(1) = last_report%photon_loss_all
    photon_loss(:) = photon_loss_all(:)/ncell                         ! evolve.F90:519
    LLS_loss = 0.0_dp                                                 ! identically zero, see DESIGN.md

    call calculate_photon_statistics (dt,xh,xh_av)                    ! evolve.F90:277-279
    call report_photonstatistics (dt)
    call update_grandtotal_photonstatistics (dt)

  end subroutine evolve3D

end module evolve
