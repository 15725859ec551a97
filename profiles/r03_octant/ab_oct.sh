# alternating same-box passes on a frozen state: per-shell launches vs strip levels (usage: ab_strip.sh "0 1 3" [passes])
for i in 1 2 3; do for v in $1; do C2R_OCTANT=$v python profiles/micro/ablate.py ${2:-3} | sed "s/^base/octant=$v/"; done; done
