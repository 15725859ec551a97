# The BASELINE.json configurations, one line each (fast sweep mode unless C2R_BENCH_SWEEP_MODE says otherwise):
#   bash profiles/micro/run_configs.sh <outdir>
OUT=$1; mkdir -p $OUT
run () { name=$1; shift; timeout 900 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 > $OUT/$name.json; python - $OUT/$name.json $name <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read())
print("%-22s %9.3f ms/step  nominal %.3e  visited/s %.3e  mean sub-boxes %.1f  frac %.3f" % (sys.argv[2], j["ms_per_step"], j["value"], j["config"]["visited_per_s"], j["config"]["mean_subboxes_per_source"][-1], j["roofline"]["frac"]))
PY
}
run c1_128_1src --mesh 128 --sources 1 --steps 50 --warmup 5
run c2_256_10src --mesh 256 --sources 10 --steps 20 --warmup 3
run c2_256_100src --mesh 256 --sources 100 --steps 10 --warmup 2
run c3_256_1000src --mesh 256 --sources 1000 --steps 3 --warmup 1
run c3_256_1000src_cold --mesh 256 --sources 1000 --x-init 2e-4 --steps 20 --warmup 3
run c3_256_125src --mesh 256 --sources 125 --steps 5 --warmup 1
run c_64_20000src_cold --mesh 64 --sources 20000 --x-init 2e-4 --steps 5 --warmup 1
run c4_504_10000src --mesh 504 --sources 10000 --density lognormal --steps 1 --warmup 0
