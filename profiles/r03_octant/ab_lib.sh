# alternating: base lib C2R_OCTANT=0 vs variant lib with C2R_OCTANT=1
for i in 1 2 3; do
  unset C2RAY_HIP_LIB; C2R_OCTANT=0 python profiles/micro/ablate.py 2 | sed "s/^base/per-shell/"
  export C2RAY_HIP_LIB=$PWD/c2-ray3dm_amd/libc2ray_hip_$1.so; C2R_OCTANT=1 python profiles/micro/ablate.py 2
done
