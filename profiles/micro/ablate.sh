# same-box A/B of variant builds on a frozen state (profiles/micro/ablate.py); builds are made before the call:
#   (the switches live in profiles/micro/ablate.patch since round 4, not in the shipped kernel: git apply profiles/micro/ablate.patch first,
#    git checkout c2-ray3dm_amd/csrc/kernels_sweep.hpp afterwards; the patch predates the round-5 file split)
#   make -C c2-ray3dm_amd/csrc variant NAME=<tag> EXTRA=-DC2R_ABLATE=<mask>      usage on the GPU box: bash profiles/micro/ablate.sh <tag> ...
for i in 1 2; do for v in base "$@"; do if [ $v = base ]; then unset C2RAY_HIP_LIB; else export C2RAY_HIP_LIB=$PWD/c2-ray3dm_amd/libc2ray_hip_$v.so; fi; python profiles/micro/ablate.py 3 2>&1 | grep -v amdgpu.ids; done; done
