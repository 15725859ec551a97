#!/bin/bash
# TEST INFRASTRUCTURE ONLY.  Builds the unmodified reference (garrelt/C2-Ray3Dm)
# hot path from the sources WHERE THEY LIE under /root/reference, with amdflang,
# into oracle/_ref/N<mesh>/ (git-ignored).  Nothing from /root/reference is
# copied into the repository.  The reference fixes the mesh size at compile time
# (sizes.f90:33; the reference's own CheckList line 2 tells the user to edit it),
# so the one `mesh=` line is rewritten by sed into the BUILD directory -- that
# generated sizes.f90 is a build output, never committed.
#
# Outputs per mesh N:
#   oracle/_ref/N<N>/c2ray_test        the reference's own program (C2Ray.F90 + nbody_test)
#   oracle/_ref/N<N>/ref_driver        OUR driver (oracle/ref_driver.F90) linked against the
#                                      reference objects; dumps f64 fixtures / times do_source
#   oracle/_ref/N<N>/ref_driver_omp    same with -fopenmp -DMY_OPENMP (TIMING ONLY: the
#                                      reference's OpenMP path has a data race)
# usage: oracle/ref_build.sh <mesh> [more meshes...]
set -euo pipefail
REF=${C2RAY_REFERENCE:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
FC=${FC:-/opt/rocm/bin/amdflang}
[ -d "$REF" ] || { echo "reference not present at $REF: nothing to build" >&2; exit 0; }
[ -x "$FC" ] || { echo "no Fortran compiler ($FC)" >&2; exit 1; }

# compile order = prerequisite order of the reference's makefile_core:40-45 (test target)
SRCS="precision.f90 mathconstants.f90 cgsconstants.f90 cgsastroconstants.f90 c2ray_parameters.f90
 cgsphotoconstants.f90 cosmoparms.f90 abundances.f90 atomic.f90 sed_parameters.f90 mrgrnk.f90
 ctrper.f90 romberg.f90 report_memory.f90 file_admin.f90 read_sm3d.f90 @sizes.f90 no_mpi.F90
 clocks.f90 nbody_test.F90 grid.F90 tped.f90 density_module.F90 clumping_module.F90 LLS.F90
 temperature_module.F90 ionfractions_module.F90 material.F90 cosmology.F90 cooling.f90
 radiation_sizes.f90 radiation_sed_parameters.F90 radiation_tables.F90 sourceprops.F90
 radiation_photoionrates.F90 thermal.f90 time_module.F90 doric.f90 photonstatistics.F90
 evolve_data.F90 column_density.f90 evolve_point.F90 evolve_source.F90 master_slave.F90
 evolve.F90 output.F90"

# Physics variants: like the mesh, the LLS and clumping models are compile-time parameters of
# c2ray_parameters.f90 that the reference's CheckList tells the user to edit ("LLS model", "clumping").
#   lls2   type_of_LLS=2       position-dependent LLS column (LLS_grid)
#   lls3   type_of_LLS=3       hard barrier at R_max_cMpc
#   clump5 type_of_clumping=5  pre-computed clumping grid (clumping_grid)
#   thermal isothermal=.false.  heating and cooling (thermal.f90); the run directory needs tables/corocool.tab
# Further compile-time switches of the path that the shipped parameters leave off (round 5):
#   pl     sed_parameters.f90 stellar_SED_type=2   power-law stellar SED (radiation_tables.F90:371-379 PL_SED): table fixtures
#   grey   c2ray_parameters.f90 grey=.true.         grey opacities (radiation_tables.F90:338-357): table fixtures
#   xray   sed_parameters.f90 use_xray_SED=.true.   the second ("P") source type of photoion_rates (radiation_photoionrates.F90:
#          133-137, 167-171).  The reference fills its X-ray tables from an UNINITIALISED local array (radiation_tables.F90:367,
#          :405-425: xray_SED is never set; sed_parameters.f90:55 "not yet implemented"), so ref_driver overwrites
#          xray_photo_thick/thin_table with tables it is handed (namelist xray_tables=) before any rate is evaluated.
# And the one PREPROCESSOR variant of the path's files (round 6):
#   allfrac  -DALLFRAC on every file (no parameter file rewritten): xh / xh_av / xh_intermed carry both fractions (:,:,:,0:1) and
#          the NEUTRAL fraction is stored, not derived as 1 - x (ionfractions_module.F90:19-50, evolve_point.F90:130-142, :341-350,
#          :394-399, evolve.F90:141-150, :179, :215, photonstatistics.F90).  No shipped makefile defines it; it compiles as it lies.
variant_file () {   # $1 = variant -> the reference file(s) the variant rewrites
  case "$1" in
    pl|xray) echo sed_parameters.f90 ;;
    allfrac) echo "" ;;
    xraythermal) echo "sed_parameters.f90 c2ray_parameters.f90" ;;     # use_xray_SED=.true. AND isothermal=.false.
    *) echo c2ray_parameters.f90 ;;
  esac
}
params_for_variant () {   # $1 = variant, $2 = output file
  case "$1" in
    lls2)   sed 's|^\( *integer,parameter :: type_of_LLS=\)1|\12|' "$REF/c2ray_parameters.f90" > "$2"; grep -q "type_of_LLS=2" "$2" ;;
    lls3)   sed 's|^\( *integer,parameter :: type_of_LLS=\)1|\13|' "$REF/c2ray_parameters.f90" > "$2"; grep -q "type_of_LLS=3" "$2" ;;
    thermal) sed 's|^\( *logical,parameter :: isothermal=\).true.|\1.false.|' "$REF/c2ray_parameters.f90" > "$2"; grep -q "isothermal=.false." "$2" ;;
    clump5) sed 's|^\( *integer,parameter :: type_of_clumping=\)1|\15|' "$REF/c2ray_parameters.f90" > "$2"; grep -q "type_of_clumping=5" "$2" ;;
    grey)   sed 's|^\( *logical,parameter :: grey = \).false.|\1.true.|' "$REF/c2ray_parameters.f90" > "$2"; grep -q "grey = .true." "$2" ;;
    pl)     sed 's|^\( *integer,parameter :: stellar_SED_type=\)1|\12|' "$REF/sed_parameters.f90" > "$2"; grep -q "stellar_SED_type=2" "$2" ;;
    xray)   sed 's|^\( *logical,parameter :: use_xray_SED=\).false.|\1.true.|' "$REF/sed_parameters.f90" > "$2"; grep -q "use_xray_SED=.true." "$2" ;;
    xraythermal)
      case "$(basename "$2")" in
        sed_parameters.f90) sed 's|^\( *logical,parameter :: use_xray_SED=\).false.|\1.true.|' "$REF/sed_parameters.f90" > "$2"; grep -q "use_xray_SED=.true." "$2" ;;
        *) sed 's|^\( *logical,parameter :: isothermal=\).true.|\1.false.|' "$REF/c2ray_parameters.f90" > "$2"; grep -q "isothermal=.false." "$2" ;;
      esac ;;
    *) echo "unknown variant $1" >&2; exit 1 ;;
  esac
}

build_variant () {   # $1 = mesh[:variant], $2 = subdir, $3 = extra flags
  local N=${1%%:*} V="" FLAGS="-DGFORT -O2 $3"
  [ "$1" != "$N" ] && V=${1#*:}
  [ "$V" = allfrac ] && FLAGS="$FLAGS -DALLFRAC"
  local B=$HERE/_ref/N$N${V:+_$V}/$2
  mkdir -p "$B"
  sed "s|^\( *integer,dimension(Ndim),parameter,public :: mesh=\)(/ 300, 300, 300 /)|\1(/ $N, $N, $N /)|" \
      "$REF/sizes.f90" > "$B/sizes.f90"
  grep -q "mesh=(/ $N, $N, $N /)" "$B/sizes.f90"
  local VF="" vf
  if [ -n "$V" ]; then
    VF=$(variant_file "$V")
    for vf in $VF; do params_for_variant "$V" "$B/$vf"; done
  fi
  local objs=""
  for s in $SRCS; do
    local src o
    if [ "${s#@}" != "$s" ]; then src="$B/${s#@}"; else src="$REF/$s"; fi
    for vf in $VF; do [ "$s" = "$vf" ] && src="$B/$vf"; done
    o="$B/$(basename "${s#@}" | sed 's/\.[fF]90$/.o/')"
    if [ ! -f "$o" ] || [ "$src" -nt "$o" ]; then
      ( cd "$B" && $FC $FLAGS -c "$src" -o "$o" 2>>"$B/build.log" )
    fi
    objs="$objs $o"
  done
  ( cd "$B" && $FC $FLAGS -c "$REF/C2Ray.F90" -o C2Ray.o 2>>build.log \
      && $FC $FLAGS -o c2ray_test $objs C2Ray.o 2>>build.log )
  ( cd "$B" && $FC $FLAGS -c "$HERE/ref_driver.F90" -o ref_driver.o 2>>build.log \
      && $FC $FLAGS -o ref_driver $objs ref_driver.o 2>>build.log )
}

# The reference's OWN program with its evolve modules swapped for the HIP drop-in
# (c2-ray3dm_amd/fortran/evolve_hip.F90 + libc2ray_hip.so): the integration test of INTEGRATION.md.
build_hip_dropin () {   # $1 = mesh[:variant]
  local N=${1%%:*} V="" FLAGS="-DGFORT -O2"
  [ "$1" != "$N" ] && V=${1#*:}
  [ "$V" = allfrac ] && FLAGS="$FLAGS -DALLFRAC"
  local D=$HERE/_ref/N$N${V:+_$V}
  local S=$D/serial B=$D/hip PKG=$HERE/../c2-ray3dm_amd
  [ -f "$PKG/libc2ray_hip.so" ] || { echo "libc2ray_hip.so not built: skipping drop-in program" >&2; return 0; }
  mkdir -p "$B"
  # the shim's evolve.mod / evolve_source.mod / photonstatistics.mod land in $B and shadow the reference's ($B before $S);
  # output.o of the serial build is reused as it is: it refers to the module's public variables by name
  ( cd "$B" && $FC $FLAGS -I"$S" -c "$PKG/fortran/evolve_hip.F90" -o evolve_hip.o 2>>build.log \
      && $FC $FLAGS -I"$B" -I"$S" -c "$REF/C2Ray.F90" -o C2Ray.o 2>>build.log \
      && $FC $FLAGS -DC2RAY_HIP_SHIM -I"$B" -I"$S" -c "$HERE/ref_driver.F90" -o ref_driver.o 2>>build.log )
  local objs=""
  for o in "$S"/*.o; do
    case "$(basename "$o")" in
      evolve_point.o|evolve_source.o|master_slave.o|evolve.o|C2Ray.o|ref_driver.o) ;;
      photonstatistics.o) ;;        # the shim brings its own module of that name (device-fed, evolve_hip.F90)
      *) objs="$objs $o" ;;
    esac
  done
  local LINK="-L$PKG -lc2ray_hip -Wl,-rpath,\$ORIGIN/../../../../c2-ray3dm_amd -Wl,-rpath,/opt/rocm/lib"
  # (timing comparison only, profiles/dropin_timing.py leg "hip-hoststats": the shim as it was before round 5 -- the driver's
  # own photonstatistics.o with its three serial mesh loops per step, xh_av copied back for them)
  if [ -z "$V" ] && [ "$N" -ge 128 ]; then
    mkdir -p "$D/hip_hoststats"
    ( cd "$D/hip_hoststats" && $FC $FLAGS -DC2R_REFERENCE_PHOTONSTATISTICS -I"$S" -c "$PKG/fortran/evolve_hip.F90" -o evolve_hip.o 2>>build.log \
        && $FC $FLAGS -I"$D/hip_hoststats" -I"$S" -c "$REF/C2Ray.F90" -o C2Ray.o 2>>build.log \
        && $FC $FLAGS -o c2ray_test_hip $objs "$S/photonstatistics.o" evolve_hip.o C2Ray.o $LINK 2>>build.log )
  fi
  # the reference's own program, and our fixture driver (ref_driver.F90: do_source / evolve3D / restart
  # modes), both with the HIP modules in place of the reference's evolve modules
  ( cd "$B" && $FC $FLAGS -o c2ray_test_hip $objs evolve_hip.o C2Ray.o $LINK 2>>build.log \
      && $FC $FLAGS -o ref_driver_hip $objs evolve_hip.o ref_driver.o $LINK 2>>build.log )
}

for N in "$@"; do
  build_variant "$N" serial ""
  build_hip_dropin "$N"
  if [ "${N%%:*}" = "$N" ]; then       # timing build only for the shipped parameters
    build_variant "$N" omp "-fopenmp -DMY_OPENMP"
  fi
  echo "built oracle/_ref/N${N/:/_}"
done
