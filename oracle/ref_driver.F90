!> TEST INFRASTRUCTURE ONLY -- not part of the product.
!!
!! Our own driver program, linked against the UNMODIFIED reference objects built
!! by oracle/ref_build.sh.  It repeats the reference's start-up order
!! (C2Ray.F90:108-198) and per-step preparation (C2Ray.F90:264-376), then calls
!! the reference's own hot-path routines (evolve3D evolve.F90:83, do_source
!! evolve_source.F90:58, cinterp column_density.f90:29, photoion_rates
!! radiation_photoionrates.F90:71, doric doric.f90:33) and dumps their inputs
!! and outputs as raw little-endian f64/f32/i32 streams.  All hot-path state of
!! the reference is public module data, which is what makes this possible
!! without touching a reference source file.
!!
!! Run in a scratch directory holding: results/ , test_sources.dat
!! (sourceprops.F90:248), the answers file (inputs/input_example_test format)
!! and driver.nml:
!!   &ctl mode='evolve'|'sweep'|'grid'|'cells'|'point'|'tables', nsteps=, x_init=, dens_file=,
!!        x_file=, dump_first=, dump_last=, ns_dump=, nrep=, out_dir=, t_file= /
!!   (mode 'sweep' also writes <tag>_nbox.txt: the sub-box count each source ended with; mode 'grid' calls
!!    master_slave_processing::do_grid for all sources at once and then evolve_point::evolve0D_global for every cell --
!!    with -DC2RAY_HIP_SHIM, the drop-in build, the shim's whole-mesh evolve0D_global_all)
!! usage: ref_driver <answers-file>
program ref_driver

  use precision, only: dp
  use clocks, only: setup_clocks
  use file_admin, only: stdinput, logf, flag_for_file_input
  use c2ray_parameters, only: cosmological, type_of_clumping, use_LLS, type_of_LLS, isothermal
  use my_mpi
  use output_module, only: setup_output
  use sizes, only: mesh
  use grid, only: grid_ini, dr, vol
  use radiation_tables, only: rad_ini, stellar_photo_thick_table, stellar_photo_thin_table, &
       stellar_heat_thick_table, stellar_heat_thin_table, xray_photo_thick_table, xray_photo_thin_table, &
       xray_heat_thick_table, xray_heat_thin_table
  use sed_parameters, only: use_xray_SED
  use radiative_cooling, only: setup_cool, coolin
  use thermalevolution, only: thermal
  use temperature_module, only: temperature_grid
  use ionfractions_module, only: ionstates
  use radiation_sed_parameters, only: S_star
  use radiation_photoionrates, only: photoion_rates, photrates
  use nbody, only: nbody_ini, NumZred, zred_array
  use cosmology, only: cosmology_init, redshift_evol, cosmo_evol, zred
  use material, only: material_ini
  use density_module, only: density_init, ndens
  use ionfractions_module, only: xh
  use clumping_module, only: set_clumping, load_clumping_model, clumping, clumping_grid
  use lls_module, only: set_LLS, coldensh_LLS, LLS_grid, R_max_LLS
  use times, only: time_ini, set_timesteps
  use sourceprops, only: source_properties_ini, source_properties, NumSrc, srcpos, &
       NormFlux_stellar, NormFlux_xray
  use photonstatistics, only: photon_loss, totrec, totcollisions, dh0, total_ion
  use evolve_data, only: evolve_ini, phih_grid, phiheat_grid, xh_av, xh_intermed, coldensh_out, &
       photon_loss_all, last_l, last_r, photon_loss_src_thread
  use evolve_source, only: do_source, sum_nbox, sum_nbox_all
  use master_slave_processing, only: do_grid
#ifdef C2RAY_HIP_SHIM
  use evolve_point, only: local_chemistry, evolve0D_global_all, evolve0D, evolve0D_global
#else
  use evolve_point, only: local_chemistry, evolve0D, evolve0D_global
#endif
  use c2ray_parameters, only: subboxsize, max_subbox
  use evolve, only: evolve3D
  use column_density, only: cinterp
  use doric_module, only: doric

  implicit none

  character(len=16)  :: mode = 'evolve'
  integer            :: nsteps = 1, dump_first = 1, dump_last = 1, ns_dump = 1, nrep = 1
  real(kind=dp)      :: x_init = -1.0_dp
  character(len=512) :: dens_file = 'none', x_file = 'none', out_dir = './dump/'
  character(len=512) :: lls_file = 'none', clump_file = 'none', t_file = 'none', xray_tables = 'none', x0_file = 'none'
  namelist /ctl/ mode, nsteps, x_init, dens_file, x_file, dump_first, dump_last, &
       ns_dump, nrep, out_dir, lls_file, clump_file, t_file, xray_tables, x0_file

  character(len=512) :: answers
  integer :: restart = 0, nz0 = 1, ierror = 0, nz, u, istep, ns, irep, gi, gj, gk, conv_flag
  integer :: nbox_before, nbox_src(4096) = 0
  integer :: nbx, qq, ci, cj, ck, lastpos_l(3), lastpos_r(3), rt(3), ncall
  real(kind=dp) :: end_time, sim_time, output_time, dt, actual_dt
  real(kind=dp) :: t_sweep
  integer(kind=8) :: c0, c1, crate
  real, allocatable :: factor(:,:,:), t_init(:,:,:)
  character(len=8) :: tag

  ! ---- start-up, in the order of C2Ray.F90:108-198 -------------------------
  call setup_clocks
  call mpi_setup()
  call get_command_argument(1, answers)
  open(unit=stdinput, file=trim(answers))
  call flag_for_file_input(.true.)
  open(newunit=u, file='driver.nml', status='old')
  read(u, nml=ctl)
  close(u)

  call setup_output()
  call grid_ini()
  call rad_ini()
  ! Builds with use_xray_SED=.true. (sed_parameters.f90:56): rad_ini integrates the X-ray tables over a local array it never
  ! fills (radiation_tables.F90:367 xray_SED; "not yet implemented", sed_parameters.f90:55).  The tables are INPUTS of the rate
  ! path (radiation_photoionrates.F90:268-270), so the fixture runs hand them in: thick(0:NumTau) then thin(0:NumTau), raw f64.
  if (use_xray_SED .and. trim(xray_tables) /= 'none') then
     open(newunit=u, file=trim(xray_tables), access='stream', form='unformatted', status='old')
     read(u) xray_photo_thick_table(:,1)
     read(u) xray_photo_thin_table(:,1)
     if (.not.isothermal) then          ! ... followed by the type's heating tables (radiation_tables.F90:84-85)
        read(u) xray_heat_thick_table(:,1)
        read(u) xray_heat_thin_table(:,1)
     endif
     close(u)
  endif
  if (.not.isothermal) call setup_cool()          ! C2Ray.F90:143 (reads ./tables/corocool.tab)
  call material_ini(restart, nz0, ierror)
  call nbody_ini(ierror)
  call source_properties_ini()
  call time_ini()
  call evolve_ini()
  sim_time = 0.0_dp
  call cosmology_init(zred_array(nz0), sim_time)
  call load_clumping_model(dr(1))

  if (trim(mode) == 'tables') then
     call dump_r8_1d('thick_table', stellar_photo_thick_table(:,1), size(stellar_photo_thick_table,1))
     call dump_r8_1d('thin_table', stellar_photo_thin_table(:,1), size(stellar_photo_thin_table,1))
     if (.not.isothermal) then
        call dump_r8_1d('heat_thick_table', stellar_heat_thick_table(:,1), size(stellar_heat_thick_table,1))
        call dump_r8_1d('heat_thin_table', stellar_heat_thin_table(:,1), size(stellar_heat_thin_table,1))
     endif
     stop
  endif

  if (trim(mode) == 'point') then
     call point_tests()
     stop
  endif

  ! ---- slices / time steps, in the order of C2Ray.F90:264-427 ---------------
  istep = 0
  slices: do nz = nz0, NumZred-1
     zred = zred_array(nz)
     call set_timesteps(zred, zred_array(nz+1), end_time, dt, output_time)
     call source_properties(zred, nz, end_time-sim_time, restart)
     call density_init(zred, nz)
     if (trim(dens_file) /= 'none') then
        ! our perturbation of the test problem's uniform density: ndens *= factor
        allocate(factor(mesh(1),mesh(2),mesh(3)))
        open(newunit=u, file=trim(dens_file), access='stream', form='unformatted', status='old')
        read(u) factor
        close(u)
        ndens = ndens*factor
        deallocate(factor)
     endif
     steps: do
        istep = istep + 1
        actual_dt = min(end_time-sim_time, dt)
        if (cosmological) then
           call redshift_evol(sim_time+0.5*actual_dt)
           call cosmo_evol()
        endif
        if (type_of_clumping /= 5) call set_clumping(zred)
        if (use_LLS .and. type_of_LLS /= 2) call set_LLS(zred)

        ! builds with type_of_LLS=2 / type_of_clumping=5: the reference would read its grids from
        ! N-body products; here they come from files written by make_golden.py
        if (istep == 1 .and. trim(lls_file) /= 'none') then
           if (.not. allocated(LLS_grid)) allocate(LLS_grid(mesh(1),mesh(2),mesh(3)))
           open(newunit=u, file=trim(lls_file), access='stream', form='unformatted', status='old')
           read(u) LLS_grid
           close(u)
        endif
        if (istep == 1 .and. trim(clump_file) /= 'none') then
           if (.not. allocated(clumping_grid)) allocate(clumping_grid(mesh(1),mesh(2),mesh(3)))
           open(newunit=u, file=trim(clump_file), access='stream', form='unformatted', status='old')
           read(u) clumping_grid
           close(u)
        endif

        if (istep == 1) then
#ifdef ALLFRAC
           ! builds with -DALLFRAC (ionfractions_module.F90:19-50): the ionized fraction comes in as before, the stored neutral
           ! fraction is set the way xfrac_restart_init does (:97-100) -- or read from x0_file where a test wants the two to
           ! be inconsistent on purpose (what distinguishes the build: evolve0D reads the STORED neutral fraction)
           if (x_init >= 0.0_dp) xh(:,:,:,1) = x_init
           if (trim(x_file) /= 'none') then
              open(newunit=u, file=trim(x_file), access='stream', form='unformatted', status='old')
              read(u) xh(:,:,:,1)
              close(u)
           endif
           xh(:,:,:,0) = 1.0_dp - xh(:,:,:,1)
           if (trim(x0_file) /= 'none') then
              open(newunit=u, file=trim(x0_file), access='stream', form='unformatted', status='old')
              read(u) xh(:,:,:,0)
              close(u)
           endif
#else
           if (x_init >= 0.0_dp) xh = x_init
           if (trim(x_file) /= 'none') then
              open(newunit=u, file=trim(x_file), access='stream', form='unformatted', status='old')
              read(u) xh
              close(u)
           endif
#endif
           ! non-isothermal builds: an initial temperature field (K, f32) instead of the uniform
           ! initial_temperature that temperature_array_init (temperature_module.F90:43) fills in
           if (.not.isothermal .and. trim(t_file) /= 'none') then
              allocate(t_init(mesh(1),mesh(2),mesh(3)))
              open(newunit=u, file=trim(t_file), access='stream', form='unformatted', status='old')
              read(u) t_init
              close(u)
              temperature_grid(:,:,:)%current = t_init
              temperature_grid(:,:,:)%average = t_init
              temperature_grid(:,:,:)%intermed = t_init
              deallocate(t_init)
           endif
        endif

        write(tag,'(A,I3.3)') 'step', istep

        if (trim(mode) == 'sweep') then
           ! one pass over all sources, exactly what pass_all_sources (evolve.F90:444)
           ! does around do_grid, without the chemistry
           call dump_inputs(tag)
           xh_av = xh
           xh_intermed = xh
           call system_clock(c0, crate)
           do irep = 1, nrep
              phih_grid = 0.0
              photon_loss = 0.0
              sum_nbox = 0
              if (.not.isothermal) phiheat_grid = 0.0
              do ns = 1, NumSrc
                 nbox_before = sum_nbox
                 call do_source(actual_dt, ns, 1)
                 if (irep == 1 .and. ns <= size(nbox_src)) nbox_src(ns) = sum_nbox - nbox_before
                 if (ns == ns_dump .and. irep == 1) &
                      call dump_r8(trim(tag)//'_coldensh_out', coldensh_out)
              enddo
           enddo
           call system_clock(c1)
           t_sweep = real(c1-c0,dp)/real(crate,dp)/real(nrep,dp)
           call dump_r8(trim(tag)//'_phih_grid', phih_grid)
           if (.not.isothermal) call dump_r8(trim(tag)//'_phiheat_grid', phiheat_grid)
           open(newunit=u, file=trim(out_dir)//trim(tag)//'_sweep.txt', status='replace')
           write(u,'(A,1X,ES26.17E3)') 'photon_loss', photon_loss(1)
           write(u,'(A,1X,I12)') 'sum_nbox', sum_nbox
           write(u,'(A,1X,ES26.17E3)') 'seconds_per_pass', t_sweep
           write(u,'(A,1X,I12)') 'nthreads', nthreads
           close(u)
           ! sub-boxes each source ended with (evolve_source.F90:219 adds them to sum_nbox): the cost of a source
           open(newunit=u, file=trim(out_dir)//trim(tag)//'_nbox.txt', status='replace')
           do ns = 1, min(NumSrc, size(nbox_src))
              write(u,'(I8)') nbox_src(ns)
           enddo
           close(u)
           stop
        endif

        if (trim(mode) == 'cells') then
           ! the per-cell call surface as such: source 1 traced through sub-boxes 1 and 2 by calling evolve0D cell by cell, shell
           ! after shell (any order that visits a cell after its upstream neighbours is valid: evolve_source.F90:227-591), with the
           ! sub-box limits moving as do_source moves them (:100-102, :128-136); then evolve0D_global for every cell of that box
           call dump_inputs(tag)
           xh_av = xh
           xh_intermed = xh
           phih_grid = 0.0
           if (.not.isothermal) phiheat_grid = 0.0
           coldensh_out(:,:,:) = 0.0
           ns = 1
           ncall = 0
           lastpos_r(:) = srcpos(:,ns) + min(max_subbox, mesh(:)/2 - 1 + mod(mesh(:),2))
           lastpos_l(:) = srcpos(:,ns) - min(max_subbox, mesh(:)/2)
           do nbx = 1, 2
              photon_loss_src_thread(:) = 0.0
              last_r(:) = min(srcpos(:,ns) + subboxsize*nbx, lastpos_r(:))
              last_l(:) = max(srcpos(:,ns) - subboxsize*nbx, lastpos_l(:))
              do qq = 0, subboxsize*nbx
                 do ck = -qq, qq
                    do cj = -qq, qq
                       do ci = -qq, qq
                          if (max(abs(ci), abs(cj), abs(ck)) /= qq) cycle
                          rt = srcpos(:,ns) + (/ ci, cj, ck /)
                          if (any(rt < last_l) .or. any(rt > last_r)) cycle
                          call evolve0D(actual_dt, rt, ns, 1)
                          ncall = ncall + 1
                       enddo
                    enddo
                 enddo
              enddo
           enddo
           call dump_r8(trim(tag)//'_coldensh_out', coldensh_out)
           call dump_r8(trim(tag)//'_phih_grid', phih_grid)
           conv_flag = 0
           do ck = last_l(3), last_r(3)
              do cj = last_l(2), last_r(2)
                 do ci = last_l(1), last_r(1)
                    call evolve0D_global(actual_dt, (/ modulo(ci-1,mesh(1))+1, modulo(cj-1,mesh(2))+1, modulo(ck-1,mesh(3))+1 /), conv_flag)
                 enddo
              enddo
           enddo
           call dump_x(trim(tag)//'_xh_av', xh_av)
           call dump_x(trim(tag)//'_xh_intermed', xh_intermed)
           open(newunit=u, file=trim(out_dir)//trim(tag)//'_cells.txt', status='replace')
           write(u,'(A,1X,ES26.17E3)') 'photon_loss_src', photon_loss_src_thread(1)
           write(u,'(A,1X,I12)') 'evolve0D_calls', ncall
           write(u,'(A,1X,I12)') 'conv_flag', conv_flag
           close(u)
           stop
        endif

        if (trim(mode) == 'grid') then
           ! pass_all_sources' core (evolve.F90:462-478: sum_nbox=0, local_chemistry=.false., do_grid) followed by
           ! global_pass' core (evolve.F90:548-555: evolve0D_global over the mesh), through the modules' public routines
           call dump_inputs(tag)
           xh_av = xh
           xh_intermed = xh
           phih_grid = 0.0
           photon_loss = 0.0
           if (.not.isothermal) phiheat_grid = 0.0
           sum_nbox = 0
           local_chemistry = .false.
           call do_grid(actual_dt, 1)
           call dump_r8(trim(tag)//'_phih_grid', phih_grid)
           conv_flag = 0
#ifdef C2RAY_HIP_SHIM
           call evolve0D_global_all(actual_dt, conv_flag)
#else
           do gk = 1, mesh(3)
              do gj = 1, mesh(2)
                 do gi = 1, mesh(1)
                    call evolve0D_global(actual_dt, (/ gi, gj, gk /), conv_flag)
                 enddo
              enddo
           enddo
#endif
           call dump_x(trim(tag)//'_xh_av', xh_av)
           call dump_x(trim(tag)//'_xh_intermed', xh_intermed)
           open(newunit=u, file=trim(out_dir)//trim(tag)//'_grid.txt', status='replace')
           write(u,'(A,1X,ES26.17E3)') 'photon_loss', photon_loss(1)
           write(u,'(A,1X,I12)') 'sum_nbox', sum_nbox
           write(u,'(A,1X,I12)') 'conv_flag', conv_flag
           write(u,'(A,1X,L1)') 'local_chemistry', local_chemistry
           close(u)
           stop
        endif

        ! mode 'evolve' (restart_flag=0) or 'restart' (restart_flag=3: the reference reads
        ! ./iterdump.bin through start_from_dump, evolve.F90:328, on the first step)
        if (istep >= dump_first .and. istep <= dump_last) then
           call dump_inputs(tag)
           if (.not.isothermal) call dump_temper(trim(tag)//'_temper_before')
        endif
        write(logf,*) 'REFDRIVER step ', istep
        if (trim(mode) == 'restart' .and. istep == 1) then
           call evolve3D(sim_time, actual_dt, 3)
        else
           call evolve3D(sim_time, actual_dt, 0)
        endif
        if (istep >= dump_first .and. istep <= dump_last) then
           call dump_x(trim(tag)//'_xh_after', xh)
           call dump_x(trim(tag)//'_xh_av', xh_av)
           call dump_x(trim(tag)//'_xh_intermed', xh_intermed)
           call dump_r8(trim(tag)//'_phih_grid', phih_grid)
           if (.not.isothermal) then
              call dump_r8(trim(tag)//'_phiheat_grid', phiheat_grid)
              call dump_temper(trim(tag)//'_temper_after')
           endif
           open(newunit=u, file=trim(out_dir)//trim(tag)//'_out.txt', status='replace')
           write(u,'(A,1X,ES26.17E3)') 'photon_loss_all', photon_loss_all(1)
           write(u,'(A,1X,I12)') 'sum_nbox_all', sum_nbox_all
           ! photon statistics of the step (photonstatistics.F90:82-228, called at evolve.F90:277)
           write(u,'(A,1X,ES26.17E3)') 'totrec', totrec
           write(u,'(A,1X,ES26.17E3)') 'totcollisions', totcollisions
           write(u,'(A,1X,ES26.17E3)') 'dh0', dh0
           write(u,'(A,1X,ES26.17E3)') 'total_ion', total_ion
           close(u)
        endif
        sim_time = sim_time + actual_dt
        if (istep >= nsteps) exit slices
        if (abs(sim_time-end_time) <= 1e-6*end_time) exit steps
     enddo steps
     if (cosmological) then
        call redshift_evol(sim_time)
        call cosmo_evol()
     endif
  enddo slices

contains

  subroutine dump_r8(name, a)
    character(len=*), intent(in) :: name
    real(kind=dp), intent(in) :: a(:,:,:)
    integer :: uu
    open(newunit=uu, file=trim(out_dir)//trim(name)//'.f64', access='stream', &
         form='unformatted', status='replace')
    write(uu) a
    close(uu)
  end subroutine dump_r8

  !> one of the ionization-fraction arrays: <name>.f64 holds the ionized fraction; builds with -DALLFRAC also leave the stored
  !! neutral fraction in <name>0.f64
#ifdef ALLFRAC
  subroutine dump_x(name, a)
    character(len=*), intent(in) :: name
    real(kind=dp), intent(in) :: a(:,:,:,0:)
    call dump_r8(name, a(:,:,:,1))
    call dump_r8(trim(name)//'0', a(:,:,:,0))
  end subroutine dump_x
#else
  subroutine dump_x(name, a)
    character(len=*), intent(in) :: name
    real(kind=dp), intent(in) :: a(:,:,:)
    call dump_r8(name, a)
  end subroutine dump_x
#endif

  !> temperature_grid as it lies in memory: (current, average, intermed) f32 per cell
  subroutine dump_temper(name)
    character(len=*), intent(in) :: name
    integer :: uu
    open(newunit=uu, file=trim(out_dir)//trim(name)//'.f32', access='stream', &
         form='unformatted', status='replace')
    write(uu) temperature_grid
    close(uu)
  end subroutine dump_temper

  subroutine dump_r8_1d(name, a, n)
    character(len=*), intent(in) :: name
    integer, intent(in) :: n
    real(kind=dp), intent(in) :: a(n)
    integer :: uu
    open(newunit=uu, file=trim(out_dir)//trim(name)//'.f64', access='stream', &
         form='unformatted', status='replace')
    write(uu) a
    close(uu)
  end subroutine dump_r8_1d

  !> everything the hot path reads for this step (SURVEY.md s8b "what the GPU box will run")
  subroutine dump_inputs(tag)
    character(len=*), intent(in) :: tag
    integer :: uu, is
    call dump_x(trim(tag)//'_xh_before', xh)
    open(newunit=uu, file=trim(out_dir)//trim(tag)//'_ndens.f32', access='stream', &
         form='unformatted', status='replace')
    write(uu) ndens
    close(uu)
    open(newunit=uu, file=trim(out_dir)//trim(tag)//'_in.txt', status='replace')
    write(uu,'(A,1X,I12)') 'mesh', mesh(1)
    write(uu,'(A,1X,ES26.17E3)') 'dt', actual_dt
    write(uu,'(A,1X,ES26.17E3)') 'dr1', dr(1)
    write(uu,'(A,1X,ES26.17E3)') 'dr2', dr(2)
    write(uu,'(A,1X,ES26.17E3)') 'dr3', dr(3)
    write(uu,'(A,1X,ES26.17E3)') 'vol', vol
    write(uu,'(A,1X,ES26.17E3)') 'coldensh_LLS', coldensh_LLS
    write(uu,'(A,1X,ES26.17E3)') 'clumping', real(clumping,dp)
    write(uu,'(A,1X,ES26.17E3)') 'S_star', S_star
    write(uu,'(A,1X,ES26.17E3)') 'zred', zred
    write(uu,'(A,1X,I12)') 'type_of_LLS', type_of_LLS
    write(uu,'(A,1X,I12)') 'type_of_clumping', type_of_clumping
    write(uu,'(A,1X,ES26.17E3)') 'R_max_LLS', R_max_LLS
    write(uu,'(A,1X,I12)') 'NumSrc', NumSrc
    do is = 1, NumSrc
       write(uu,'(A,1X,3(I8,1X),ES26.17E3)') 'src', srcpos(1,is), srcpos(2,is), srcpos(3,is), &
            NormFlux_stellar(is)
    enddo
    if (use_xray_SED) then                      ! NormFlux_xray (sourceprops.F90:381, :631), source by source
       do is = 1, NumSrc
          write(uu,'(A,1X,ES26.17E3)') 'xsrc', NormFlux_xray(is)
       enddo
    endif
    close(uu)
  end subroutine dump_inputs

  !> point-level known answers: cinterp on a filled cube, photoion_rates and doric on
  !! tabulated arguments.  Inputs come from files written by tests/golden/make_golden.py
  subroutine point_tests()
    integer :: uu, n, m, ii, jj, kk, r, cnt
    integer :: sp(3), pp(3)
    real(kind=dp), allocatable :: args(:,:), res(:,:)
    real(kind=dp) :: cd, path, xf(0:1), xfav(0:1), rhe
    type(photrates) :: phi

    ! --- cinterp: coldensh_out filled from file, all cells within radius r of the source
    open(newunit=uu, file='point_coldens.f64', access='stream', form='unformatted', status='old')
    read(uu) coldensh_out
    close(uu)
    open(newunit=uu, file='point_cinterp.txt', status='old')
    read(uu,*) sp(1), sp(2), sp(3), r
    close(uu)
    n = (2*r+1)**3 - 1
    allocate(res(2,n))
    cnt = 0
    do kk = -r, r
       do jj = -r, r
          do ii = -r, r
             if (ii == 0 .and. jj == 0 .and. kk == 0) cycle
             cnt = cnt + 1
             pp = sp + (/ ii, jj, kk /)
             call cinterp(pp, sp, cd, path)
             res(1,cnt) = cd
             res(2,cnt) = path
          enddo
       enddo
    enddo
    open(newunit=uu, file=trim(out_dir)//'point_cinterp_out.f64', access='stream', &
         form='unformatted', status='replace')
    write(uu) res
    close(uu)
    deallocate(res)

    ! --- photoion_rates: rows of (colum_in, colum_out, vol); needs NormFlux_stellar(1)
    !     from the source list, so load the sources of slice nz0 first
    zred = zred_array(nz0)
    call set_timesteps(zred, zred_array(nz0+1), end_time, dt, output_time)
    call source_properties(zred, nz0, end_time-sim_time, restart)
    open(newunit=uu, file='point_photo.f64', access='stream', form='unformatted', status='old')
    read(uu) m
    allocate(args(3,m), res(3,m))
    read(uu) args
    close(uu)
    do ii = 1, m
       phi = photoion_rates(args(1,ii), args(2,ii), args(3,ii), 1, 0.5_dp)
       res(1,ii) = phi%photo_cell_HI
       res(2,ii) = phi%photo_in
       res(3,ii) = phi%photo_out
    enddo
    open(newunit=uu, file=trim(out_dir)//'point_photo_out.f64', access='stream', &
         form='unformatted', status='replace')
    write(uu) NormFlux_stellar(1)
    write(uu) res
    close(uu)
    deallocate(args, res)

    ! --- doric: rows of (dt, temp0, rhe, rhh, xfh1_old, xfh_av1, phih); clumping set as in a step
    call set_clumping(zred)
    open(newunit=uu, file='point_doric.f64', access='stream', form='unformatted', status='old')
    read(uu) m
    allocate(args(7,m), res(4,m))
    read(uu) args
    close(uu)
    do ii = 1, m
       xf(1) = args(5,ii); xf(0) = 1.0_dp - xf(1)
       xfav(1) = args(6,ii); xfav(0) = 1.0_dp - xfav(1)
       rhe = args(3,ii)
       call doric(args(1,ii), args(2,ii), rhe, args(4,ii), xf, xfav, args(7,ii))
       res(1,ii) = xf(0); res(2,ii) = xf(1); res(3,ii) = xfav(0); res(4,ii) = xfav(1)
    enddo
    open(newunit=uu, file=trim(out_dir)//'point_doric_out.f64', access='stream', &
         form='unformatted', status='replace')
    write(uu) res
    close(uu)
    deallocate(args, res)

    if (.not.isothermal) call point_tests_thermal()
  end subroutine point_tests

  !> non-isothermal builds: photoion_rates' heating (radiation_photoionrates.F90:323-417), coolin
  !! (cooling.f90:38) and thermal (thermal.f90:22) on tabulated arguments
  subroutine point_tests_thermal()
    integer :: uu, m, ii
    real(kind=dp), allocatable :: args(:,:), res(:,:)
    real(kind=dp) :: xf(0:1), t_final, t_avg
    type(photrates) :: phi
    type(ionstates) :: ion

    ! --- heating: rows of (colum_in, colum_out, vol)
    open(newunit=uu, file='point_photo.f64', access='stream', form='unformatted', status='old')
    read(uu) m
    allocate(args(3,m), res(1,m))
    read(uu) args
    close(uu)
    do ii = 1, m
       phi = photoion_rates(args(1,ii), args(2,ii), args(3,ii), 1, 0.5_dp)
       res(1,ii) = phi%heat
    enddo
    open(newunit=uu, file=trim(out_dir)//'point_heat_out.f64', access='stream', &
         form='unformatted', status='replace')
    write(uu) res
    close(uu)
    deallocate(args, res)

    ! --- coolin: rows of (nucldens, eldens, temp0)
    open(newunit=uu, file='point_cool.f64', access='stream', form='unformatted', status='old')
    read(uu) m
    allocate(args(3,m), res(1,m))
    read(uu) args
    close(uu)
    xf = (/ 0.5_dp, 0.5_dp /)
    do ii = 1, m
       res(1,ii) = coolin(args(1,ii), args(2,ii), xf, args(3,ii))
    enddo
    open(newunit=uu, file=trim(out_dir)//'point_cool_out.f64', access='stream', &
         form='unformatted', status='replace')
    write(uu) res
    close(uu)
    deallocate(args, res)

    ! --- thermal: rows of (dt, T_initial, ndens_electron, ndens_atom, h_old(1), h_av(1), h(1), heat);
    !     zred is the first slice's (set above); outputs (final, average); -1 marks "not set"
    !     (thermal leaves its outputs untouched when T_initial <= minitemp, thermal.f90:83)
    open(newunit=uu, file='point_thermal.f64', access='stream', form='unformatted', status='old')
    read(uu) m
    allocate(args(8,m), res(2,m))
    read(uu) args
    close(uu)
    do ii = 1, m
       ion%h_old(1) = args(5,ii); ion%h_old(0) = 1.0_dp - args(5,ii)
       ion%h_av(1) = args(6,ii); ion%h_av(0) = 1.0_dp - args(6,ii)
       ion%h(1) = args(7,ii); ion%h(0) = 1.0_dp - args(7,ii)
       phi%heat = args(8,ii)
       t_final = -1.0_dp; t_avg = -1.0_dp
       call thermal(args(1,ii), args(2,ii), t_final, t_avg, args(3,ii), args(4,ii), ion, phi)
       res(1,ii) = t_final; res(2,ii) = t_avg
    enddo
    open(newunit=uu, file=trim(out_dir)//'point_thermal_out.f64', access='stream', &
         form='unformatted', status='replace')
    write(uu) zred
    write(uu) res
    close(uu)
  end subroutine point_tests_thermal

end program ref_driver
