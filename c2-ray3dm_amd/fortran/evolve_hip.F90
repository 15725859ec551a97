!> Drop-in replacement of the reference's `evolve` and `evolve_source` modules (evolve.F90,
!! evolve_source.F90) for C2-Ray, backed by the MI355X HIP library (include/c2ray_hip.h) through
!! ISO_C_BINDING.
!!
!! Link position: in the reference's makefile_core:31 replace
!!     EVOLVE = evolve_data.o column_density.o evolve_point.o evolve_source.o master_slave.o evolve.o
!! by
!!     EVOLVE = evolve_data.o evolve_hip.o            (and add  -L<dir> -lc2ray_hip  to the link line)
!! and drop photonstatistics.o from the program's object list (makefile_core:40 and the other *_SET lines): this file
!! brings its own module `photonstatistics` -- same public variables and routines, fed by the sums the device has
!! already taken (c2r_report) instead of three serial N^3 host loops per time step.  (-DC2R_REFERENCE_PHOTONSTATISTICS
!! keeps the reference's photonstatistics.o in the link and the host loops with it.)
!! `evolve_data` (the arrays C2Ray.F90:61 and output.F90:31 use) stays the reference's own file.
!!
!! Modules:
!!   c2ray_hip        the bind(C) mirror of include/c2ray_hip.h, the context, per-step state transfer
!!   photonstatistics the reference's module surface (photonstatistics.F90:39-55, :71-293), device-fed
!!   evolve_source    `do_source(dt,ns1,niter)`, `sum_nbox`, `sum_nbox_all`   (evolve_source.F90:45-58)
!!   master_slave_processing `do_grid(dt,niter)`                              (master_slave.F90:53)
!!   evolve_point     `evolve0D`, `evolve0D_global`, `evolve0D_global_all`    (evolve_point.F90:83, :305)
!!   evolve           `evolve3D(time,dt,restart)`                              (evolve.F90:83, C2Ray.F90:379)
!! All inputs are taken from the same module-global arrays the reference routines read, by `use`
!! association; the HIP side never keeps a host pointer after a call returns.
module c2ray_hip

  use, intrinsic :: iso_c_binding
  use precision, only: dp
  use file_admin, only: logf
  use sizes, only: mesh
  use grid, only: dr, vol
  use temperature_module, only: temper_val, temperature_grid
  use clumping_module, only: clumping, clumping_grid
  use lls_module, only: coldensh_LLS, LLS_grid, R_max_LLS
  use sourceprops, only: NumSrc, srcpos, NormFlux_stellar, NormFlux_xray
  use radiation_sizes, only: NumTau
  use radiation_tables, only: stellar_photo_thick_table, stellar_photo_thin_table, minlogtau, dlogtau, &
       stellar_heat_thick_table, stellar_heat_thin_table, xray_photo_thick_table, xray_photo_thin_table, &
       xray_heat_thick_table, xray_heat_thin_table
  use sed_parameters, only: use_xray_SED
  use radiation_sed_parameters, only: S_star
  use cgsphotoconstants, only: sigma_HI_at_ion_freq
  use cgsconstants, only: bh00, albpow, colh0, temph0, k_B
  use atomic, only: gamma1
  use cosmology_parameters, only: H0, Omega0
  use cosmology, only: zred
  use mathconstants, only: pi
  use abundances, only: abu_c
  use c2ray_parameters, only: epsilon, convergence_fraction, minimum_fractional_change, &
       minimum_fraction_of_atoms, loss_fraction, subboxsize, max_subbox, use_LLS, type_of_LLS, &
       type_of_clumping, isothermal, cosmological, minitemp, relative_denergy
#ifdef MPI
  ! -DMPI builds of the driver (mpi.F90:83-160): one rank per GPU; sources are distributed over the ranks inside the
  ! library (master_slave.F90:74-96 / :124-330) and Gamma is summed with RCCL over xGMI (evolve.F90:577-616)
  use my_mpi, only: rank, npr, MPI_COMM_NEW
#endif

  implicit none
#ifdef MPI
  include 'mpif.h'
#endif

  save

  public

  integer, parameter :: C2R_MAX_ITER_LOG = 128

  !> mirror of struct c2r_params (include/c2ray_hip.h)
  type, bind(C) :: c2r_params
     integer(c_int32_t) :: mesh(3), device, subboxsize, max_subbox, numtau, max_outer_iter, &
          max_chem_iter, deterministic_rates, sweep_mode, allfrac
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
     integer(c_int32_t) :: chem_not_converged, timing_split
     integer(c_int64_t) :: it_conv_flag(C2R_MAX_ITER_LOG), it_sum_nbox(C2R_MAX_ITER_LOG)
     real(c_double) :: it_rel_change_xh1(C2R_MAX_ITER_LOG), it_rel_change_xh0(C2R_MAX_ITER_LOG), &
          it_sum_xh1(C2R_MAX_ITER_LOG)
     real(c_double) :: h0_before, h1_before, h0_after, h1_after, totrec, totcollisions, dh0, &
          total_ion, totalsrc, photcons, it_photcons(C2R_MAX_ITER_LOG)
     real(c_double) :: seconds_upload, seconds_download, seconds_total
  end type c2r_report

  !> mirror of struct c2r_thermal_params (non-isothermal builds of the driver: c2ray_parameters.f90:28)
  type, bind(C) :: c2r_thermal_params
     real(c_double) :: tau_heat_limit, k_B, gamma1, minitemp, relative_denergy, thermal_rate_floor, &
          thermal_time_tol, temp_conv_rel, temp_conv_abs, H0, Omega0, cool_mintemp, cool_dtemp
     integer(c_int32_t) :: cool_points, thermal_max_steps, cosmological, reserved0
  end type c2r_thermal_params

  interface
     integer(c_int) function c2r_default_thermal(t) bind(C, name="c2r_default_thermal")
       import :: c_int, c2r_thermal_params
       type(c2r_thermal_params), intent(out) :: t
     end function c2r_default_thermal
     integer(c_int) function c2r_set_thermal(ctx, t, heat_thick, heat_thin, n, cie_cool) bind(C, name="c2r_set_thermal")
       import :: c_int, c_ptr, c_double, c_int32_t, c2r_thermal_params
       type(c_ptr), value :: ctx
       type(c2r_thermal_params), intent(in) :: t
       real(c_double), intent(in) :: heat_thick(*), heat_thin(*), cie_cool(*)
       integer(c_int32_t), value :: n
     end function c2r_set_thermal
     integer(c_int) function c2r_set_redshift(ctx, zred) bind(C, name="c2r_set_redshift")
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: ctx
       real(c_double), value :: zred
     end function c2r_set_redshift
     !> temperature_grid is an array of type(temperature_states): three default reals per cell, passed by address
     integer(c_int) function c2r_evolve3d_thermal(ctx, dt, restart_niter, photon_loss_all, ndens, xh, xh_av, &
          xh_intermed, phih_grid, phiheat_grid, temperature_grid, rep) bind(C, name="c2r_evolve3d_thermal")
       import :: c_int, c_ptr, c_double, c_float, c_int32_t, c2r_report
       type(c_ptr), value :: ctx
       real(c_double), value :: dt, photon_loss_all
       integer(c_int32_t), value :: restart_niter
       real(c_float), intent(in) :: ndens(*)
       real(c_double), intent(inout) :: xh(*), phih_grid(*), phiheat_grid(*)
       type(c_ptr), value :: xh_av, xh_intermed              ! c_loc of the array (download; upload on restart) or c_null_ptr
       type(*), dimension(*), intent(inout) :: temperature_grid
       type(c2r_report), intent(out) :: rep
     end function c2r_evolve3d_thermal
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
     !> xh_av, xh_intermed, phih_grid: c_loc of the array to have it downloaded after the step, c_null_ptr to leave it alone
     integer(c_int) function c2r_set_xray_tables(ctx, thick, thin, n) bind(C, name="c2r_set_xray_tables")
       import :: c_int, c_ptr, c_double, c_int32_t
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: thick(*), thin(*)
       integer(c_int32_t), value :: n
     end function c2r_set_xray_tables
     integer(c_int) function c2r_set_xray_heat_tables(ctx, heat_thick, heat_thin, n) bind(C, name="c2r_set_xray_heat_tables")
       import :: c_int, c_ptr, c_double, c_int32_t
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: heat_thick(*), heat_thin(*)
       integer(c_int32_t), value :: n
     end function c2r_set_xray_heat_tables
     integer(c_int) function c2r_set_xray_sources(ctx, normflux_xray, nsrc) bind(C, name="c2r_set_xray_sources")
       import :: c_int, c_ptr, c_double, c_int32_t
       type(c_ptr), value :: ctx
       real(c_double), intent(in) :: normflux_xray(*)
       integer(c_int32_t), value :: nsrc
     end function c2r_set_xray_sources
     integer(c_int) function c2r_evolve3d(ctx, dt, ndens, xh, xh_av, xh_intermed, phih_grid, rep) &
          bind(C, name="c2r_evolve3d")
       import :: c_int, c_ptr, c_double, c_float, c2r_report
       type(c_ptr), value :: ctx
       real(c_double), value :: dt
       real(c_float), intent(in) :: ndens(*)
       real(c_double), intent(inout) :: xh(*)
       type(c_ptr), value :: xh_av, xh_intermed, phih_grid
       type(c2r_report), intent(out) :: rep
     end function c2r_evolve3d
     integer(c_int) function c2r_evolve3d_restart(ctx, dt, niter, photon_loss_all, ndens, xh, xh_av, &
          xh_intermed, phih_grid, rep) bind(C, name="c2r_evolve3d_restart")
       import :: c_int, c_ptr, c_double, c_float, c_int32_t, c2r_report
       type(c_ptr), value :: ctx
       real(c_double), value :: dt, photon_loss_all
       integer(c_int32_t), value :: niter
       real(c_float), intent(in) :: ndens(*)
       real(c_double), intent(inout) :: xh(*), xh_av(*), xh_intermed(*), phih_grid(*)
       type(c2r_report), intent(out) :: rep
     end function c2r_evolve3d_restart
     integer(c_int) function c2r_set_lls(ctx, lls_type, lls_grid, R_max_LLS) bind(C, name="c2r_set_lls")
       import :: c_int, c_ptr, c_double, c_float, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: lls_type
       type(c_ptr), value :: lls_grid            ! const float* or NULL
       real(c_double), value :: R_max_LLS
     end function c2r_set_lls
     integer(c_int) function c2r_set_clumping_grid(ctx, clump_grid) bind(C, name="c2r_set_clumping_grid")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       type(c_ptr), value :: clump_grid          ! const float* or NULL
     end function c2r_set_clumping_grid
     integer(c_int) function c2r_do_source_host(ctx, ns, ndens, xh_av, phih_grid, coldensh_out, &
          photon_loss_src, nbox) bind(C, name="c2r_do_source_host")
       import :: c_int, c_ptr, c_double, c_float, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: ns           ! 1..NumSrc, as ns1 of do_source
       real(c_float), intent(in) :: ndens(*)
       real(c_double), intent(in) :: xh_av(*)
       real(c_double), intent(inout) :: phih_grid(*)
       real(c_double), intent(out) :: coldensh_out(*)
       real(c_double), intent(out) :: photon_loss_src
       integer(c_int32_t), intent(out) :: nbox
     end function c2r_do_source_host
     integer(c_int) function c2r_evolve0d_host(ctx, ns, rtpos, last_l, last_r, ndens, xh_av, coldensh_out, phih_grid, &
          phiheat_grid, photon_loss_src) bind(C, name="c2r_evolve0d_host")
       import :: c_int, c_ptr, c_double, c_float, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: ns           ! 1..NumSrc
       integer(c_int32_t), intent(in) :: rtpos(3), last_l(3), last_r(3)
       real(c_float), intent(in) :: ndens(*)
       real(c_double), intent(in) :: xh_av(*)
       real(c_double), intent(inout) :: coldensh_out(*), phih_grid(*)
       type(c_ptr), value :: phiheat_grid        ! double* (non-isothermal builds) or NULL
       real(c_double), intent(inout) :: photon_loss_src
     end function c2r_evolve0d_host
     integer(c_int) function c2r_global_pass_cell_host(ctx, dt, pos, ndens, xh, xh_av, xh_intermed, phih_grid, phiheat_grid, &
          temperature_grid, conv_flag) bind(C, name="c2r_global_pass_cell_host")
       import :: c_int, c_ptr, c_double, c_float, c_int32_t
       type(c_ptr), value :: ctx
       real(c_double), value :: dt
       integer(c_int32_t), intent(in) :: pos(3)
       real(c_float), intent(in) :: ndens(*)
       real(c_double), intent(in) :: xh(*), phih_grid(*)
       real(c_double), intent(inout) :: xh_av(*), xh_intermed(*)
       type(c_ptr), value :: phiheat_grid        ! const double* or NULL
       type(c_ptr), value :: temperature_grid    ! float* (3 per cell) or NULL
       integer(c_int32_t), intent(inout) :: conv_flag
     end function c2r_global_pass_cell_host
     integer(c_int) function c2r_do_grid_host(ctx, ndens, xh_av, phih_grid, phiheat_grid, photon_loss, sum_nbox) &
          bind(C, name="c2r_do_grid_host")
       import :: c_int, c_ptr, c_double, c_float, c_int64_t
       type(c_ptr), value :: ctx
       real(c_float), intent(in) :: ndens(*)
       real(c_double), intent(in) :: xh_av(*)
       real(c_double), intent(inout) :: phih_grid(*)
       type(c_ptr), value :: phiheat_grid        ! double* (non-isothermal builds) or NULL
       real(c_double), intent(out) :: photon_loss
       integer(c_int64_t), intent(out) :: sum_nbox
     end function c2r_do_grid_host
     integer(c_int) function c2r_global_pass_host(ctx, dt, ndens, xh, xh_av, xh_intermed, phih_grid, conv_flag) &
          bind(C, name="c2r_global_pass_host")
       import :: c_int, c_ptr, c_double, c_float, c_int64_t
       type(c_ptr), value :: ctx
       real(c_double), value :: dt
       real(c_float), intent(in) :: ndens(*)
       real(c_double), intent(in) :: xh(*), phih_grid(*)
       real(c_double), intent(inout) :: xh_av(*), xh_intermed(*)
       integer(c_int64_t), intent(out) :: conv_flag
     end function c2r_global_pass_host
     integer(c_int) function c2r_upload(ctx, which, host) bind(C, name="c2r_upload")
       import :: c_int, c_ptr, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: which
       type(*), dimension(*), intent(in) :: host
     end function c2r_upload
     integer(c_int) function c2r_set_iteration_hook(ctx, fn, user) bind(C, name="c2r_set_iteration_hook")
       import :: c_int, c_ptr, c_funptr
       type(c_ptr), value :: ctx
       type(c_funptr), value :: fn
       type(c_ptr), value :: user
     end function c2r_set_iteration_hook
     integer(c_int) function c2r_set_balance(ctx, on) bind(C, name="c2r_set_balance")
       import :: c_int, c_ptr, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: on
     end function c2r_set_balance
     integer(c_int) function c2r_get_device(ctx, device) bind(C, name="c2r_get_device")
       import :: c_int, c_ptr, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), intent(out) :: device
     end function c2r_get_device
     !> c2r_allreduce_fn of include/c2ray_hip.h as a C function pointer (c_funloc of a bind(C) function, or
     !! the RCCL binding's own callback installed by c2r_rccl_attach)
     integer(c_int) function c2r_set_rank(ctx, rank, nranks, fn, user) bind(C, name="c2r_set_rank")
       import :: c_int, c_ptr, c_funptr, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: rank, nranks
       type(c_funptr), value :: fn
       type(c_ptr), value :: user
     end function c2r_set_rank
#ifdef MPI
     ! libc2ray_rccl.so (include/c2ray_rccl.h): link with -lc2ray_rccl
     integer(c_int) function c2r_rccl_unique_id(id) bind(C, name="c2r_rccl_unique_id")
       import :: c_int, c_char
       character(kind=c_char), intent(out) :: id(128)          ! C2R_RCCL_ID_BYTES
     end function c2r_rccl_unique_id
     integer(c_int) function c2r_rccl_attach(ctx, id, rank, nranks) bind(C, name="c2r_rccl_attach")
       import :: c_int, c_ptr, c_char, c_int32_t
       type(c_ptr), value :: ctx
       character(kind=c_char), intent(in) :: id(128)
       integer(c_int32_t), value :: rank, nranks
     end function c2r_rccl_attach
     integer(c_int) function c2r_rccl_slab_chemistry(ctx, on) bind(C, name="c2r_rccl_slab_chemistry")
       import :: c_int, c_ptr, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: on
     end function c2r_rccl_slab_chemistry
     integer(c_int) function c2r_rccl_detach(ctx) bind(C, name="c2r_rccl_detach")
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
     end function c2r_rccl_detach
#endif
     integer(c_int) function c2r_download(ctx, which, host) bind(C, name="c2r_download")
       import :: c_int, c_ptr, c_double, c_int32_t
       type(c_ptr), value :: ctx
       integer(c_int32_t), value :: which
       type(*), dimension(*), intent(inout) :: host       ! f64 N^3 (arrays 1-5) or temperature_grid (array 6)
     end function c2r_download
  end interface

  type(c_ptr) :: ctx = c_null_ptr

  !> evolve3D also copies xh_av and xh_intermed back to the driver's arrays after every step.  Off by default: they are
  !! work arrays of the replaced modules (nothing else in the reference reads them: output.F90 takes xh and phih_grid;
  !! an iteration dump downloads them itself), 16 bytes per cell and step over PCIe.  Environment: C2R_SHIM_SYNC_WORK_ARRAYS=1
  logical :: sync_work_arrays = .false.

contains

  !> address of one of the driver's f64 mesh arrays (for the C entries that take an optional array as a pointer)
  function dp_address(g) result(p)
    real(kind=dp), dimension(:,:,:), allocatable, target, intent(in) :: g
    type(c_ptr) :: p
    p = c_null_ptr
    if (allocated(g)) p = c_loc(g)
  end function dp_address

#ifdef ALLFRAC
  !> ... and of the (mesh,0:1) ionization-fraction arrays of a -DALLFRAC build (the library is told: c2r_params%allfrac)
  function xfrac_address(g) result(p)
    real(kind=dp), dimension(:,:,:,:), allocatable, target, intent(in) :: g
    type(c_ptr) :: p
    p = c_null_ptr
    if (allocated(g)) p = c_loc(g)
  end function xfrac_address
#else
  function xfrac_address(g) result(p)
    real(kind=dp), dimension(:,:,:), allocatable, target, intent(in) :: g
    type(c_ptr) :: p
    p = dp_address(g)
  end function xfrac_address
#endif

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
    error stop "c2ray_hip failure"
  end subroutine check

  !> evolve_ini counterpart for the device side: created lazily on the first evolve3D
  subroutine evolve_hip_ini()
    type(c2r_params) :: p
    integer(c_int32_t) :: dev
    character(len=16) :: envval
    integer :: envstat
#ifdef MPI
    character(kind=c_char) :: uid(128)
    integer :: mympierror
#endif
    call check(c2r_default_params(p), "c2r_default_params")
    p%mesh = mesh
#ifdef MPI
    p%device = -1_c_int32_t            ! C2R_DEVICE_AUTO: this rank's LOCAL rank (from the launcher) modulo the node's GPUs
#else
    p%device = 0
#endif
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
    ! Sweep arithmetic: the reference has no such parameter, so this drop-in takes it from its own run-time switch, the
    ! environment variable C2R_SWEEP_MODE (unset or 1: C2R_SWEEP_FAST, the library default -- xh within 1e-9 of the Fortran where
    ! the task asks for 1e-5; 0: C2R_SWEEP_EXACT, column densities bit-identical to the Fortran, ~20 % slower).  The library
    ! reads no environment: what is set here is what runs, and it is logged below.
    call get_environment_variable("C2R_SWEEP_MODE", envval, status=envstat)
    if (envstat == 0 .and. len_trim(envval) > 0) p%sweep_mode = merge(0_c_int32_t, 1_c_int32_t, trim(envval) == "0")
#ifdef ALLFRAC
    ! this build stores both fractions (ionfractions_module.F90:19-50): every xh / xh_av / xh_intermed handed to the library is the
    ! whole (mesh,0:1) array, and the library keeps the stored neutral fraction instead of deriving it
    p%allfrac = 1_c_int32_t
#endif
    call get_environment_variable("C2R_SHIM_SYNC_WORK_ARRAYS", envval, status=envstat)
    sync_work_arrays = envstat == 0 .and. len_trim(envval) > 0 .and. trim(envval) /= "0"
    call check(c2r_create(ctx, p), "c2r_create")
    call check(c2r_set_tables(ctx, stellar_photo_thick_table(:,1), stellar_photo_thin_table(:,1), &
         int(NumTau+1, c_int32_t)), "c2r_set_tables")
    call check(c2r_get_device(ctx, dev), "c2r_get_device")
    write(logf,*) "c2ray_hip: evolve hot path on HIP device ", dev
    write(logf,*) "c2ray_hip: sweep mode ", merge("fast (C2R_SWEEP_FAST)  ", "exact (C2R_SWEEP_EXACT)", p%sweep_mode == 1)
    if (.not.isothermal) call thermal_hip_ini()
    ! builds with use_xray_SED=.true. (sed_parameters.f90:56): the second source type of photoion_rates
    ! (radiation_photoionrates.F90:133-137) -- its two tables now, NormFlux_xray with every source list (hip_step_state)
    if (use_xray_SED) then
       call check(c2r_set_xray_tables(ctx, xray_photo_thick_table(:,1), xray_photo_thin_table(:,1), &
            int(NumTau+1, c_int32_t)), "c2r_set_xray_tables")
       if (.not.isothermal) call check(c2r_set_xray_heat_tables(ctx, xray_heat_thick_table(:,1), xray_heat_thin_table(:,1), &
            int(NumTau+1, c_int32_t)), "c2r_set_xray_heat_tables")
       write(logf,*) "c2ray_hip: X-ray source type on the device (photo_lookuptable 'P')"
    endif
#ifdef MPI
    ! Join the RCCL communicator: rank 0 makes the 128-byte token, one broadcast next to those of mpi.F90 hands it
    ! out, every rank attaches on its own device.  From here on c2r_evolve3d sweeps this rank's share of the sources
    ! (static 1+rank,NumSrc,npr for the first pass, master_slave.F90:85; cost-balanced afterwards: the role of
    ! do_grid_master/do_grid_slave, :124-330) and sums phih_grid, photon_loss and sum_nbox over the ranks
    ! (mpi_accumulate_grid_quantities, evolve.F90:577-616) on the GPUs.
    if (npr > 1) then
       if (rank == 0) call check(c2r_rccl_unique_id(uid), "c2r_rccl_unique_id")
       call MPI_BCAST(uid, 128, MPI_BYTE, 0, MPI_COMM_NEW, mympierror)
       call check(c2r_rccl_attach(ctx, uid, int(rank, c_int32_t), int(npr, c_int32_t)), "c2r_rccl_attach")
       call check(c2r_set_balance(ctx, 1_c_int32_t), "c2r_set_balance")
       ! non-isothermal builds: the global pass is heavy (thermal.f90 per chemistry iteration), so it runs on z-slabs
       ! (reduce-scatter of the rates, all-gather of xh_av / xh_intermed / temperature_grid) instead of replicated on every
       ! rank behind an all-reduce (evolve.F90:548-555, :599-609); the isothermal pass costs less than the extra slab it moves
       if (.not.isothermal) call check(c2r_rccl_slab_chemistry(ctx, 1_c_int32_t), "c2r_rccl_slab_chemistry")
    endif
#endif
  end subroutine evolve_hip_ini

  !> Non-isothermal builds (c2ray_parameters.f90:28): hand the heating tables (rad_ini filled them,
  !! radiation_tables.F90:521-543), the cooling curve and the thermal parameters to the library.  The cooling curve is
  !! private to radiative_cooling, so the table file is read here exactly as setup_cool reads it (cooling.f90:64-87).
  subroutine thermal_hip_ini()
    integer, parameter :: temppoints = 61                              ! cooling.f90:26
    real(kind=dp) :: temp(temppoints), cie_cool(temppoints)
    type(c2r_thermal_params) :: t
    integer :: itemp
    open(unit=22, file='tables/corocool.tab', status='old')
    do itemp = 1, temppoints
       read(22,*) temp(itemp), cie_cool(itemp)
    enddo
    close(22)
    do itemp = 1, temppoints
       cie_cool(itemp) = 10.0d0**cie_cool(itemp)
    enddo
    call check(c2r_default_thermal(t), "c2r_default_thermal")
    t%k_B = k_B; t%gamma1 = gamma1; t%minitemp = minitemp; t%relative_denergy = relative_denergy
    t%H0 = H0; t%Omega0 = Omega0
    t%cool_mintemp = temp(1); t%cool_dtemp = temp(2) - temp(1); t%cool_points = temppoints
    t%cosmological = merge(1_c_int32_t, 0_c_int32_t, cosmological)
    call check(c2r_set_thermal(ctx, t, stellar_heat_thick_table(:,1), stellar_heat_thin_table(:,1), &
         int(NumTau+1, c_int32_t), cie_cool), "c2r_set_thermal")
    write(logf,*) "c2ray_hip: heating and cooling on the device (thermal.f90)"
  end subroutine thermal_hip_ini

  subroutine evolve_hip_end()
    integer(c_int) :: rc
    if (c_associated(ctx)) then
#ifdef MPI
       if (npr > 1) rc = c2r_rccl_detach(ctx)
#endif
       call c2r_destroy(ctx)
    endif
    ctx = c_null_ptr
  end subroutine evolve_hip_end

  !> What changes between calls and lives in host module variables of the driver: cell size,
  !! LLS column / grid, clumping, temperature, source list.  Cheap (a few scalars; grids only when
  !! the compile-time model uses them).
  subroutine hip_step_state()
    type(c_ptr) :: gridp
    if (.not. c_associated(ctx)) call evolve_hip_ini()
    call check(c2r_set_step(ctx, dr, vol, coldensh_LLS, real(clumping, c_float), &
         real(temper_val, c_double)), "c2r_set_step")
    if (use_LLS .and. type_of_LLS == 2) then                          ! LLS_point, LLS.F90:199
       call check(c2r_set_lls(ctx, 2_c_int32_t, grid_address(LLS_grid), 0.0_c_double), "c2r_set_lls")
    elseif (use_LLS .and. type_of_LLS == 3) then                      ! R_max_LLS, LLS.F90:186
       call check(c2r_set_lls(ctx, 3_c_int32_t, c_null_ptr, R_max_LLS), "c2r_set_lls")
    endif
    if (type_of_clumping >= 3 .and. type_of_clumping <= 5) then       ! clumping_point, evolve_point.F90:442, clumping_module.F90:106
       gridp = grid_address(clumping_grid)
       call check(c2r_set_clumping_grid(ctx, gridp), "c2r_set_clumping_grid")
    endif
    call check(c2r_set_sources(ctx, srcpos, NormFlux_stellar(1:NumSrc), int(NumSrc, c_int32_t)), &
         "c2r_set_sources")
    if (use_xray_SED .and. allocated(NormFlux_xray)) &
         call check(c2r_set_xray_sources(ctx, NormFlux_xray, int(NumSrc, c_int32_t)), "c2r_set_xray_sources")
    if (.not.isothermal) call check(c2r_set_redshift(ctx, zred), "c2r_set_redshift")   ! cosmo_cool, cosmology.F90:198
  end subroutine hip_step_state

  function grid_address(g) result(p)
    real, dimension(:,:,:), allocatable, target, intent(in) :: g
    type(c_ptr) :: p
    p = c_null_ptr
    if (allocated(g)) p = c_loc(g)
  end function grid_address

end module c2ray_hip

#ifndef C2R_REFERENCE_PHOTONSTATISTICS
! =============================================================================================

!> `photonstatistics` of the reference (photonstatistics.F90) -- same module name, same public variables (output.F90:33-36
!! reads totrec, totcollisions, dh0, total_ion, grtotal_ion, grtotal_src, LLS_loss, photon_loss) and the same public
!! routines -- but the four mesh sums behind them (neutrals / ions before and after the step, recombinations and collisional
!! ionizations during it: state_before :104, state_after :190, total_rates :137) are the ones the DEVICE has taken inside
!! c2r_evolve3d (k_photon_sums; the chemistry kernel for the per-iteration ones) and returned in c2r_report:
!! `photon_statistics_from_device`.  The reference walks the mesh three times per time step for them, serially, with a
!! power and an exponential per cell -- of order 0.1 s at 128^3 where the whole step takes milliseconds on the GPU.
!! The host-array routines remain for callers outside evolve3D; they run one fused loop over the mesh instead of three.
module photonstatistics

  use precision, only: dp
  use my_mpi, only: rank
  use file_admin, only: logf
  use cgsconstants, only: albpow, bh00, colh0, temph0
  use cgsphotoconstants, only: sigh => sigma_HI_at_ion_freq
  use sizes, only: mesh
  use grid, only: vol
  use density_module, only: ndens
  use temperature_module, only: temperature_states_dbl, get_temperature_point
  use clumping_module, only: clumping, clumping_point
  use abundances, only: abu_c
  use sourceprops, only: NormFlux_stellar, NumSrc
  use radiation_sed_parameters, only: S_star
  use radiation_sizes, only: NumFreqBnd
  use c2ray_parameters, only: type_of_clumping

  implicit none

  logical,parameter :: do_photonstatistics=.true.   !< photonstatistics.F90:39
  real(kind=dp) :: totrec          !< recombinations during the step
  real(kind=dp) :: totcollisions   !< collisional ionizations during the step
  real(kind=dp) :: dh0             !< change in the number of neutral H atoms
  real(kind=dp) :: total_ion       !< ionizing photons used
  real(kind=dp) :: LLS_loss        !< photons lost in LLSs (identically zero on this path, see DESIGN.md)
  real(kind=dp) :: grtotal_ion     !< grand total of ionizing photons used
  real(kind=dp) :: grtotal_src     !< grand total of ionizing photons produced
  real(kind=dp) :: photon_loss(NumFreqBnd)   !< photons leaving the grid (per cell after evolve3D)

  real(kind=dp),private :: h0_before, h0_after, h1_before, h1_after

contains

  subroutine initialize_photonstatistics ()
    grtotal_ion=0.0
    grtotal_src=0.0
  end subroutine initialize_photonstatistics

  !> The step's statistics as c2r_report holds them (h0/h1 before and after, totrec, totcollisions: already times vol
  !! and dt), followed by total_ionizations (photonstatistics.F90:222-228)
  subroutine photon_statistics_from_device (h0b, h1b, h0a, h1a, rec, col)
    real(kind=dp),intent(in) :: h0b, h1b, h0a, h1a, rec, col
    h0_before = h0b; h1_before = h1b; h0_after = h0a; h1_after = h1a
    totrec = rec; totcollisions = col
    call total_ionizations ()
  end subroutine photon_statistics_from_device

  !> One walk over the mesh for any subset of the sums (host arrays; callers outside evolve3D).  Same expressions and the
  !! same left-to-right order per sum as photonstatistics.F90:113-127, :153-180.
  subroutine mesh_sums (x_state, x_rates, h0, h1, rec, col)
#ifdef ALLFRAC
    real(kind=dp),dimension(mesh(1),mesh(2),mesh(3),0:1),intent(in),optional :: x_state, x_rates
#else
    real(kind=dp),dimension(mesh(1),mesh(2),mesh(3)),intent(in),optional :: x_state, x_rates
#endif
    real(kind=dp),intent(out) :: h0, h1, rec, col
    integer :: i, j, k
    real(kind=dp) :: nd, y0, y1, de
    type(temperature_states_dbl) :: t
    h0 = 0.0; h1 = 0.0; rec = 0.0; col = 0.0
    do k = 1, mesh(3)
       do j = 1, mesh(2)
          do i = 1, mesh(1)
             nd = ndens(i,j,k)
             if (present(x_state)) then
#ifdef ALLFRAC
                h0 = h0 + ndens(i,j,k)*x_state(i,j,k,0)
                h1 = h1 + ndens(i,j,k)*x_state(i,j,k,1)
#else
                h0 = h0 + ndens(i,j,k)*(1.0_dp - x_state(i,j,k))
                h1 = h1 + ndens(i,j,k)*x_state(i,j,k)
#endif
             endif
             if (present(x_rates)) then
#ifdef ALLFRAC
                y1 = x_rates(i,j,k,1); y0 = x_rates(i,j,k,0)
#else
                y1 = x_rates(i,j,k); y0 = 1.0_dp - y1
#endif
                de = nd*(y1 + abu_c)                                   ! tped.f90:81
                call get_temperature_point (i,j,k,t)
                if (type_of_clumping >= 3 .and. type_of_clumping <= 5) call clumping_point (i,j,k)
                rec = rec + nd*y1*de*clumping*bh00*(t%average/1e4)**albpow
                col = col + nd*y0*de*colh0*sqrt(t%average)*exp(-temph0/t%average)
             endif
          enddo
       enddo
    enddo
  end subroutine mesh_sums

  subroutine state_before (xh_l)
#ifdef ALLFRAC
    real(kind=dp),dimension(mesh(1),mesh(2),mesh(3),0:1),intent(in) :: xh_l
#else
    real(kind=dp),dimension(mesh(1),mesh(2),mesh(3)),intent(in) :: xh_l
#endif
    real(kind=dp) :: r, c
    call mesh_sums (x_state=xh_l, h0=h0_before, h1=h1_before, rec=r, col=c)
    h0_before = h0_before*vol; h1_before = h1_before*vol
  end subroutine state_before

  subroutine state_after (xh_l)
#ifdef ALLFRAC
    real(kind=dp),dimension(mesh(1),mesh(2),mesh(3),0:1),intent(in) :: xh_l
#else
    real(kind=dp),dimension(mesh(1),mesh(2),mesh(3)),intent(in) :: xh_l
#endif
    real(kind=dp) :: r, c
    call mesh_sums (x_state=xh_l, h0=h0_after, h1=h1_after, rec=r, col=c)
    h0_after = h0_after*vol; h1_after = h1_after*vol
  end subroutine state_after

  subroutine total_rates (dt,xh_l)
    real(kind=dp),intent(in) :: dt
#ifdef ALLFRAC
    real(kind=dp),dimension(mesh(1),mesh(2),mesh(3),0:1),intent(in) :: xh_l
#else
    real(kind=dp),dimension(mesh(1),mesh(2),mesh(3)),intent(in) :: xh_l
#endif
    real(kind=dp) :: a, b
    call mesh_sums (x_rates=xh_l, h0=a, h1=b, rec=totrec, col=totcollisions)
    totrec = totrec*vol*dt; totcollisions = totcollisions*vol*dt
  end subroutine total_rates

  !> photonstatistics.F90:82-99 for host arrays: neutrals after the step from xh_l, rates from xh_r, one walk over the mesh
  subroutine calculate_photon_statistics (dt,xh_l,xh_r)
    real(kind=dp),intent(in) :: dt
#ifdef ALLFRAC
    real(kind=dp),dimension(mesh(1),mesh(2),mesh(3),0:1),intent(in) :: xh_l, xh_r
#else
    real(kind=dp),dimension(mesh(1),mesh(2),mesh(3)),intent(in) :: xh_l, xh_r
#endif
    call mesh_sums (xh_l, xh_r, h0_after, h1_after, totrec, totcollisions)
    h0_after = h0_after*vol; h1_after = h1_after*vol
    totrec = totrec*vol*dt; totcollisions = totcollisions*vol*dt
    call total_ionizations ()
  end subroutine calculate_photon_statistics

  subroutine total_ionizations ()
    dh0 = (h0_before - h0_after)
    total_ion = totrec + dh0
  end subroutine total_ionizations

  !> photonstatistics.F90:233-249 (nothing on the HIP path calls it: LLS_loss stays zero, as in the reference, whose
  !! evolve0D passes a photo_in_HI that nothing sets)
  subroutine total_LLS_loss (phi_out,coldensh_LLS)
    real(kind=dp),intent(in) :: phi_out, coldensh_LLS
    LLS_loss = LLS_loss + phi_out*(1.0 - exp(-sigh*coldensh_LLS))
  end subroutine total_LLS_loss

  !> The conservation line of the log (photonstatistics.F90:254-281): same columns, same formats
  subroutine report_photonstatistics (dt)
    real(kind=dp),intent(in) :: dt
    real(kind=dp) :: totalsrc, lost_cells, lost_lls
    lost_cells = sum(photon_loss)*dt*real(mesh(1))*real(mesh(2))*real(mesh(3))
    lost_lls = LLS_loss*dt
    totalsrc = sum(NormFlux_stellar(1:NumSrc))*S_star*dt
    if (rank == 0) then
       write(logf,"(8(1pe10.3))") total_ion, totalsrc, (total_ion + LLS_loss - totcollisions)/totalsrc, dh0/total_ion, &
            totrec/total_ion, lost_lls/totalsrc, lost_cells/totalsrc, totcollisions/total_ion
       write(logf,*) h1_before,h1_after
    endif
  end subroutine report_photonstatistics

  subroutine update_grandtotal_photonstatistics (dt)
    real(kind=dp),intent(in) :: dt
    grtotal_src = grtotal_src + sum(NormFlux_stellar(1:NumSrc))*S_star*dt
    grtotal_ion = grtotal_ion + total_ion - totcollisions
  end subroutine update_grandtotal_photonstatistics

end module photonstatistics
#endif

! =============================================================================================

!> `evolve_source` of the reference (evolve_source.F90): ray tracing for ONE source on the GPU,
!! with the reference's side effects on the host module arrays.
module evolve_source

  use, intrinsic :: iso_c_binding
  use precision, only: dp
  use density_module, only: ndens
  use photonstatistics, only: photon_loss
  use evolve_data, only: phih_grid, phiheat_grid, xh_av, coldensh_out
  use c2ray_parameters, only: isothermal
  use c2ray_hip, only: ctx, check, hip_step_state, c2r_do_source_host, c2r_download

  implicit none

  save

  private

  integer, public :: sum_nbox      !< sum of all nboxes (this process)      evolve_source.F90:45
  integer, public :: sum_nbox_all  !< sum of all nboxes (all processes)     evolve_source.F90:46

  public :: do_source

contains

  !> Ray tracing over the entire 3D grid for source ns1 (same contract as evolve_source.F90:58):
  !! adds the source's photo-ionization rates into phih_grid, leaves its outgoing column densities
  !! in coldensh_out, adds its escaping photons to photon_loss(1) and its sub-box count to sum_nbox.
  subroutine do_source(dt,ns1,niter)

    real(kind=dp),intent(in) :: dt !< time step (unused by the transfer, as in the reference)
    integer,intent(in) :: ns1 !< number of the source being done
    integer,intent(in) :: niter !< iteration counter

    real(c_double) :: photon_loss_src
    integer(c_int32_t) :: nbox
    real(kind=dp), allocatable, save :: heat_src(:,:,:)

    call hip_step_state()
    call check(c2r_do_source_host(ctx, int(ns1, c_int32_t), ndens, xh_av, phih_grid, coldensh_out, &
         photon_loss_src, nbox), "c2r_do_source_host")
    if (.not.isothermal) then
       ! the source's heating rates (evolve_point.F90:285-286) are on the device after the call: add them on the host
       if (.not.allocated(heat_src)) allocate(heat_src(size(phiheat_grid,1), size(phiheat_grid,2), size(phiheat_grid,3)))
       call check(c2r_download(ctx, 5_c_int32_t, heat_src), "c2r_download")
       phiheat_grid = phiheat_grid + heat_src
    endif
    photon_loss(1) = photon_loss(1) + photon_loss_src                 ! evolve_source.F90:216
    sum_nbox = sum_nbox + nbox                                        ! evolve_source.F90:219

  end subroutine do_source

end module evolve_source

! =============================================================================================

!> `master_slave_processing` of the reference (master_slave.F90): the ray tracing of all sources.  The reference hands the
!! sources to do_source one by one -- statically (do ns1=1+rank,NumSrc,npr, :85) or through its master/worker scheduler
!! (:124-330); here the whole share of this process is ONE pass on the GPU (all sources of a shell in one launch), the
!! shares being static for the first pass and cost-balanced afterwards when the -DMPI block has attached the ranks.
module master_slave_processing

  use, intrinsic :: iso_c_binding
  use precision, only: dp
  use density_module, only: ndens
  use photonstatistics, only: photon_loss
  use evolve_data, only: phih_grid, phiheat_grid, xh_av
  use c2ray_parameters, only: isothermal
  use evolve_source, only: sum_nbox
  use c2ray_hip, only: ctx, check, hip_step_state, c2r_do_grid_host

  implicit none

  private

  public :: do_grid

contains

  !> Ray trace the whole grid for all sources of this process (same contract as master_slave.F90:53): adds their rates
  !! into phih_grid (phiheat_grid), their escaping photons to photon_loss(1), their sub-box counts to sum_nbox.
  !! coldensh_out is NOT left behind (the reference leaves the LAST source's: nothing reads it after do_grid).
  subroutine do_grid (dt,niter)

    real(kind=dp),intent(in) :: dt  !< time step (unused by the transfer, as in the reference)
    integer,intent(in) :: niter !< iteration counter

    real(c_double) :: loss
    integer(c_int64_t) :: nbox

    call hip_step_state()
    if (isothermal) then
       call check(c2r_do_grid_host(ctx, ndens, xh_av, phih_grid, c_null_ptr, loss, nbox), "c2r_do_grid_host")
    else
       call check(c2r_do_grid_host(ctx, ndens, xh_av, phih_grid, heat_address(phiheat_grid), loss, nbox), "c2r_do_grid_host")
    endif
    photon_loss(1) = photon_loss(1) + loss                            ! evolve_source.F90:216, summed over the sources
    sum_nbox = sum_nbox + int(nbox)                                   ! evolve_source.F90:219

  end subroutine do_grid

  function heat_address(g) result(p)
    real(kind=dp), dimension(:,:,:), allocatable, target, intent(in) :: g
    type(c_ptr) :: p
    p = c_null_ptr
    if (allocated(g)) p = c_loc(g)
  end function heat_address

end module master_slave_processing

! =============================================================================================

!> `evolve_point` of the reference (evolve_point.F90).  Its two public routines work on ONE cell: evolve0D(dt,rtpos,ns,niter)
!! for one (cell, source) pair inside do_source's sweep, evolve0D_global(dt,pos,conv_flag) for one cell inside
!! global_pass's triple loop.  Both are exported with the reference's signatures and side effects on the module arrays
!! (c2r_evolve0d_host, c2r_global_pass_cell_host: one kernel launch per call -- a launch costs what the reference spends on
!! thirty cells, so they are slow by construction and meant for hosts that walk the cells themselves and for tests).  The
!! product path is the same work in bulk: the per-(cell, source) work inside do_source / do_grid above (kernels
!! k_sweep_shell*), the per-cell chemistry for the WHOLE mesh at once -- the loop global_pass (evolve.F90:499-555) runs
!! around evolve0D_global -- as evolve0D_global_all.  The module flag local_chemistry (evolve_point.F90:73) is kept:
!! pass_all_sources resets it and nothing on this path sets it (the local variant of do_chemistry is not used by
!! C2-Ray3Dm's evolve3D).
module evolve_point

  use, intrinsic :: iso_c_binding
  use precision, only: dp
  use density_module, only: ndens
  use ionfractions_module, only: xh
  use temperature_module, only: temperature_grid, temperature_states
  use evolve_data, only: phih_grid, phiheat_grid, xh_av, xh_intermed, coldensh_out, last_l, last_r, &
       photon_loss_src_thread, tn
  use c2ray_parameters, only: isothermal
  use c2ray_hip, only: ctx, check, hip_step_state, c2r_global_pass_host, c2r_upload, c2r_download, &
       c2r_evolve0d_host, c2r_global_pass_cell_host

  implicit none

  save

  private

  !> Flag to know whether the do_chemistry routine was called with local option (evolve_point.F90:73)
  logical,public ::local_chemistry=.false.

  public :: evolve0D, evolve0D_global, evolve0D_global_all

contains

  !> evolve0D (evolve_point.F90:83-299): ray tracing for ONE cell at the unwrapped mesh position rtpos for source ns -- the
  !! call the reference's sweep routines make cell by cell (evolve_source.F90:227-591).  Same contract: does nothing when
  !! coldensh_out(pos) is set; otherwise sets it, adds the source's rate to phih_grid(pos) (and phiheat_grid(pos)) and, on
  !! the surface of the current sub-box (evolve_data's last_l, last_r), the escaping photons to photon_loss_src_thread(tn).
  !! One launch per cell on the GPU: slow by construction -- do_source and do_grid are the product path (they trace a whole
  !! source, or all of them, per call); this entry is for hosts that walk the cells themselves and for tests.
  subroutine evolve0D(dt,rtpos,ns,niter)

    real(kind=dp),intent(in) :: dt !< time step (unused by the transfer, as in the reference)
    integer,dimension(3),intent(in) :: rtpos !< cell position (for RT)
    integer,intent(in) :: ns !< source number
    integer,intent(in) :: niter !< global iteration number

    integer(c_int32_t) :: rt(3), ll(3), lr(3)
    type(c_ptr) :: heat

    call hip_step_state()
    rt = int(rtpos, c_int32_t); ll = int(last_l, c_int32_t); lr = int(last_r, c_int32_t)
    heat = c_null_ptr
    if (.not.isothermal) heat = heat_address0(phiheat_grid)
    call check(c2r_evolve0d_host(ctx, int(ns, c_int32_t), rt, ll, lr, ndens, xh_av, coldensh_out, phih_grid, heat, &
         photon_loss_src_thread(tn)), "c2r_evolve0d_host")

  end subroutine evolve0D

  !> evolve0D_global (evolve_point.F90:305-406) for ONE cell: do_chemistry with the collected rates and the global
  !! convergence test; updates xh_av(pos), xh_intermed(pos) (temperature_grid(pos)) and increments conv_flag when the
  !! cell has not converged.  One launch per cell: evolve0D_global_all (or evolve3D) is the product path.
  subroutine evolve0D_global(dt,pos,conv_flag)

    real(kind=dp),intent(in) :: dt !< time step
    integer,dimension(3),intent(in) :: pos !< position on mesh
    integer,intent(inout) :: conv_flag !< convergence counter

    integer(c_int32_t) :: p3(3), cf
    type(c_ptr) :: heat, temp

    call hip_step_state()
    p3 = int(pos, c_int32_t); cf = int(conv_flag, c_int32_t)
    heat = c_null_ptr; temp = c_null_ptr
    if (.not.isothermal) then
       heat = heat_address0(phiheat_grid); temp = temper_address0(temperature_grid)
    endif
    call check(c2r_global_pass_cell_host(ctx, dt, p3, ndens, xh, xh_av, xh_intermed, phih_grid, heat, temp, cf), &
         "c2r_global_pass_cell_host")
    conv_flag = int(cf)

  end subroutine evolve0D_global

  function heat_address0(g) result(p)
    real(kind=dp), dimension(:,:,:), allocatable, target, intent(in) :: g
    type(c_ptr) :: p
    p = c_null_ptr
    if (allocated(g)) p = c_loc(g)
  end function heat_address0

  function temper_address0(g) result(p)
    type(temperature_states), dimension(:,:,:), allocatable, target, intent(in) :: g
    type(c_ptr) :: p
    p = c_null_ptr
    if (allocated(g)) p = c_loc(g)
  end function temper_address0

  !> evolve0D_global (evolve_point.F90:305-406: do_chemistry with the collected rates, the global convergence test)
  !! for every cell of the mesh: reads xh, xh_av, phih_grid (phiheat_grid, temperature_grid), updates xh_av, xh_intermed
  !! (temperature_grid) and ADDS the number of non-converged cells to conv_flag, as the reference's loop does.
  subroutine evolve0D_global_all(dt,conv_flag)

    real(kind=dp),intent(in) :: dt !< time step
    integer,intent(inout) :: conv_flag !< convergence counter

    integer(c_int64_t) :: nonconv

    call hip_step_state()
    if (.not.isothermal) then
       call check(c2r_upload(ctx, 5_c_int32_t, phiheat_grid), "c2r_upload")
       call check(c2r_upload(ctx, 6_c_int32_t, temperature_grid), "c2r_upload")
    endif
    call check(c2r_global_pass_host(ctx, dt, ndens, xh, xh_av, xh_intermed, phih_grid, nonconv), "c2r_global_pass_host")
    if (.not.isothermal) call check(c2r_download(ctx, 6_c_int32_t, temperature_grid), "c2r_download")
    conv_flag = conv_flag + int(nonconv)

  end subroutine evolve0D_global_all

end module evolve_point

! =============================================================================================

module evolve

  use, intrinsic :: iso_c_binding
  use precision, only: dp
  use my_mpi, only: rank
  use file_admin, only: logf, timefile, iterdump, dump_dir
  use clocks, only: timestamp_wallclock, iterdump_minutes
  use sizes, only: mesh
  use density_module, only: ndens
  use ionfractions_module, only: xh
  use sourceprops, only: NumSrc
  use c2ray_parameters, only: convergence_fraction
  use temperature_module, only: temperature_grid
#ifdef C2R_REFERENCE_PHOTONSTATISTICS
  use photonstatistics, only: photon_loss, LLS_loss, state_before, calculate_photon_statistics, &
       report_photonstatistics, update_grandtotal_photonstatistics
#else
  use photonstatistics, only: photon_loss, LLS_loss, photon_statistics_from_device, &
       report_photonstatistics, update_grandtotal_photonstatistics
#endif
  use evolve_data, only: phih_grid, phiheat_grid, xh_av, xh_intermed, photon_loss_all
  use evolve_source, only: sum_nbox, sum_nbox_all
  use c2ray_hip

  implicit none

  save

  private

  type(c2r_report), public :: last_report
  real :: wallclock_last_dump = 0.0

  public :: evolve3D, evolve_hip_end

contains

  !> Evolve the entire grid over a time step dt (same contract as evolve.F90:83)
  subroutine evolve3D (time,dt,restart)

    real(kind=dp),intent(in) :: time !< time
    real(kind=dp),intent(in) :: dt !< time step
    integer,intent(in) :: restart !< restart flag (iteration dumps are not supported here)

    integer :: k, niter0
    real(kind=dp) :: ncell
    integer(kind=8) :: c_begin, c_state, c_call, c_end, c_rate
    real(kind=dp) :: t_total, t_setup, t_call, t_iter
    type(c_ptr) :: p_av, p_int

    call system_clock(c_begin, c_rate)
#ifdef C2R_REFERENCE_PHOTONSTATISTICS
    call state_before (xh)                                            ! evolve.F90:136 (on the device otherwise: c2r_report%h0_before)
#endif

    call hip_step_state()
    call check(c2r_set_iteration_hook(ctx, c_funloc(iteration_hook), c_null_ptr), &
         "c2r_set_iteration_hook")
    wallclock_last_dump = timestamp_wallclock ()

    if (rank == 0) write(timefile,"(A,F8.1)") "Time before starting iteration: ", timestamp_wallclock ()
    call system_clock(c_state)

    ! what comes back after the step: xh and phih_grid (output.F90 writes them); the work arrays only on request
    p_av = c_null_ptr; p_int = c_null_ptr
    if (sync_work_arrays) then
       p_av = xfrac_address(xh_av); p_int = xfrac_address(xh_intermed)
    endif
#ifdef C2R_REFERENCE_PHOTONSTATISTICS
    p_av = xfrac_address(xh_av)                                       ! calculate_photon_statistics(dt,xh,xh_av) below reads it
#endif

    niter0 = 0
    if (.not.isothermal) then
       ! heating and cooling: also phiheat_grid and temperature_grid (evolve_point.F90:285, :553; evolve.F90:220)
       if (restart /= 0) then
          call start_from_dump(restart, niter0)
          p_av = xfrac_address(xh_av); p_int = xfrac_address(xh_intermed)   ! (uploaded from the dump)
       endif
       call check(c2r_evolve3d_thermal(ctx, dt, int(merge(niter0, -1, restart /= 0), c_int32_t), photon_loss_all(1), &
            ndens, xh, p_av, p_int, phih_grid, phiheat_grid, temperature_grid, last_report), &
            "c2r_evolve3d_thermal")
    elseif (restart == 0) then
       call check(c2r_evolve3d(ctx, dt, ndens, xh, p_av, p_int, dp_address(phih_grid), last_report), &
            "c2r_evolve3d")
    else
       call start_from_dump(restart, niter0)                          ! evolve.F90:153-157
       call check(c2r_evolve3d_restart(ctx, dt, int(niter0, c_int32_t), photon_loss_all(1), ndens, &
            xh, xh_av, xh_intermed, phih_grid, last_report), "c2r_evolve3d_restart")
    endif
    call system_clock(c_call)

    ! what the reference logs per outer iteration (evolve.F90:205-210, 249-251, 559-566)
    ncell = real(mesh(1),dp)*real(mesh(2),dp)*real(mesh(3),dp)
    if (rank == 0) then
       ! (after a restart the first entry is the global pass that start_from_dump is followed by)
       do k = max(1, niter0), min(last_report%niter, C2R_MAX_ITER_LOG)
          if (k > niter0) write(logf,*) "Average number of subboxes: ", &
               real(last_report%it_sum_nbox(k))/real(NumSrc)
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

    sum_nbox = int(last_report%sum_nbox_all)                          ! last iteration, as evolve.F90:523
    sum_nbox_all = sum_nbox
    photon_loss_all(1) = last_report%photon_loss_all
    photon_loss(:) = photon_loss_all(:)/ncell                         ! evolve.F90:519
    LLS_loss = 0.0_dp                                                 ! identically zero, see DESIGN.md

#ifdef C2R_REFERENCE_PHOTONSTATISTICS
    call calculate_photon_statistics (dt,xh,xh_av)                    ! evolve.F90:277-279: three host loops over the mesh
#else
    ! evolve.F90:277-279: the same numbers, summed on the device while the step ran
    call photon_statistics_from_device (last_report%h0_before, last_report%h1_before, last_report%h0_after, &
         last_report%h1_after, last_report%totrec, last_report%totcollisions)
#endif
    call report_photonstatistics (dt)
    call update_grandtotal_photonstatistics (dt)

    ! where the wall time of this step went (seconds): the whole call; the shim's set-up before the library call (step
    ! scalars, source list, grids of the non-default switches); host-to-device copies; the outer iterations; device-to-host
    ! copies; the rest of the library call (sums, waits, page-locking on first use); logging and statistics after it
    call system_clock(c_end)
    t_total = real(c_end - c_begin, dp)/real(c_rate, dp)
    t_setup = real(c_state - c_begin, dp)/real(c_rate, dp)
    t_call = real(c_call - c_state, dp)/real(c_rate, dp)
    t_iter = last_report%seconds_sweep + last_report%seconds_chem
    if (rank == 0) write(logf,"(A,I4,7(1x,es10.3))") "c2ray_hip: evolve3D seconds [iterations|total set-up upload iterations download library-rest host-rest]:", &
         last_report%niter, t_total, t_setup, last_report%seconds_upload, t_iter, last_report%seconds_download, &
         t_call - last_report%seconds_upload - t_iter - last_report%seconds_download, &
         real(c_end - c_call, dp)/real(c_rate, dp)

  end subroutine evolve3D

  !> Called by the library after every outer iteration: the reference's dump decision
  !! (evolve.F90:271-275: write an iteration dump when iterdump_minutes of wall clock have passed).
  function iteration_hook(user, niter, loss_all) bind(C) result(rc)
    type(c_ptr), value :: user
    integer(c_int32_t), value :: niter
    real(c_double), value :: loss_all
    integer(c_int) :: rc
    real :: now
    rc = 0
    now = timestamp_wallclock ()
    if (rank == 0 .and. now - wallclock_last_dump > 60.0*iterdump_minutes) then
       wallclock_last_dump = now
       photon_loss_all(1) = loss_all
       rc = c2r_download(ctx, 4_c_int32_t, phih_grid)
#ifdef ALLFRAC
       ! (c2r_download moves ONE N^3 array: arrays 8 / 9 are the stored neutral halves, 2 / 3 the ionized ones)
       if (rc == 0) rc = c2r_download(ctx, 8_c_int32_t, xh_av(:,:,:,0))
       if (rc == 0) rc = c2r_download(ctx, 2_c_int32_t, xh_av(:,:,:,1))
       if (rc == 0) rc = c2r_download(ctx, 9_c_int32_t, xh_intermed(:,:,:,0))
       if (rc == 0) rc = c2r_download(ctx, 3_c_int32_t, xh_intermed(:,:,:,1))
#else
       if (rc == 0) rc = c2r_download(ctx, 2_c_int32_t, xh_av)
       if (rc == 0) rc = c2r_download(ctx, 3_c_int32_t, xh_intermed)
#endif
       if (.not.isothermal) then
          if (rc == 0) rc = c2r_download(ctx, 5_c_int32_t, phiheat_grid)
          if (rc == 0) rc = c2r_download(ctx, 6_c_int32_t, temperature_grid)
       endif
       if (rc == 0) call write_iteration_dump(int(niter))
    endif
  end function iteration_hook

  !> Name of the dump file a restart flag / a running dump number selects: the reference alternates
  !! iterdump1.bin and iterdump2.bin while it runs and reads iterdump.bin for restart=3 (evolve.F90:296-300, 346-353).
  function dump_path(which) result(path)
    integer, intent(in) :: which        !< 1, 2: the alternating files; anything else: iterdump.bin
    character(len=512) :: path
    character(len=16) :: leaf
    leaf = "iterdump.bin"
    if (which == 1) leaf = "iterdump1.bin"
    if (which == 2) leaf = "iterdump2.bin"
    path = trim(adjustl(dump_dir))//trim(leaf)
  end function dump_path

  !> One pass over the unformatted records of an iteration dump, in the reference's order
  !! (evolve.F90:303-317 write, :358-375 read): niter | photon_loss_all | phih_grid | xh_av | xh_intermed, and in
  !! non-isothermal builds | phiheat_grid | temperature_grid.
  subroutine dump_records (path, writing, niter)
    character(len=*), intent(in) :: path
    logical, intent(in) :: writing
    integer, intent(inout) :: niter
    if (writing) then
       open(unit=iterdump, file=trim(path), form="unformatted", status="unknown")
       write(iterdump) niter
       write(iterdump) photon_loss_all
       write(iterdump) phih_grid
       write(iterdump) xh_av
       write(iterdump) xh_intermed
       if (.not.isothermal) then
          write(iterdump) phiheat_grid
          write(iterdump) temperature_grid
       endif
    else
       open(unit=iterdump, file=trim(path), form="unformatted", status="old")
       read(iterdump) niter
       read(iterdump) photon_loss_all
       read(iterdump) phih_grid
       read(iterdump) xh_av
       read(iterdump) xh_intermed
       if (.not.isothermal) then
          read(iterdump) phiheat_grid
          read(iterdump) temperature_grid
       endif
    endif
    close(iterdump)
  end subroutine dump_records

  !> The state of outer iteration `niter` to the next of the two alternating dump files (evolve.F90:285-324)
  subroutine write_iteration_dump (niter)
    integer, intent(in) :: niter
    integer, save :: dumps_written = 0
    integer :: n
    n = niter
    write(timefile,"(A,F8.1)") "Time before writing iterdump: ", timestamp_wallclock ()
    dumps_written = dumps_written + 1
    call dump_records (dump_path(2 - mod(dumps_written, 2)), .true., n)       ! 1, 2, 1, 2, ...
    write(timefile,"(A,F8.1)") "Time after writing iterdump: ", timestamp_wallclock ()
  end subroutine write_iteration_dump

  !> Load the dump the restart flag selects into the driver's arrays (evolve.F90:328-426; every rank reads the
  !! file itself) and log what the reference logs about it (:418-422).
  subroutine start_from_dump (restart, niter)
    integer, intent(in) :: restart
    integer, intent(out) :: niter
    niter = 0
    write(timefile,"(A,F8.1)") "Time before reading iterdump: ", timestamp_wallclock ()
    call dump_records (dump_path(restart), .false., niter)
    write(logf,*) "Read iteration ",niter," from dump file"
    write(logf,*) "photon loss counter: ",photon_loss_all
#ifdef ALLFRAC
    write(logf,*) "Intermediate result for mean ionization fraction: ", &
         sum(xh_intermed(:,:,:,1))/real(mesh(1)*mesh(2)*mesh(3))
#else
    write(logf,*) "Intermediate result for mean ionization fraction: ", &
         sum(xh_intermed(:,:,:))/real(mesh(1)*mesh(2)*mesh(3))
#endif
    write(timefile,"(A,F8.1)") "Time after reading iterdump: ", timestamp_wallclock ()
  end subroutine start_from_dump

end module evolve
