# the EXACT sweep mode with the fast mode's rate arithmetic behind bit-identical column densities (isothermal): same-box A/B at the bench workload
#   git apply profiles/micro/hybrid_exact.patch; make -C c2-ray3dm_amd/csrc variant NAME=hyb EXTRA=-DC2R_EXP_HYBRID; git checkout c2-ray3dm_amd/csrc/kernels.hpp
# measured (round 3): 195.5 -> 183.5 ms per step (-6 %), sub-box counts equal, Gamma sums equal to 1e-14
for i in 1 2; do for v in base hyb; do if [ $v = base ]; then unset C2RAY_HIP_LIB; else export C2RAY_HIP_LIB=$PWD/c2-ray3dm_amd/libc2ray_hip_$v.so; fi
  python bench.py --sweep-mode exact --steps 3 --warmup 1 --no-cpu-baseline --no-other-mode --no-small-leg 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('$v exact mode', round(j['ms_per_step'],2), 'ms/step', j['check']['sum_nbox_last_step'], j['check']['phih_grid_sum'])"
done; done
