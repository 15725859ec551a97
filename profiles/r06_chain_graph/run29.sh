cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, time, numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
n, S = 256, 1000
tp = pkg.TestProblem(n); s = tp.step(1)
nd, xh = tp.fields(1, 0.999)
pos, nf = pkg.seeded_sources(n, S)
sh = pkg.static_source_share(S, 0, 8)
thick, thin, _ = pkg.build_tables()
for opts in ({}, {"chain_tail": 0}, {"chain_graph": 0}, {}, {"chain_tail": 0}, {"chain_graph": 0}):
    b = pkg.HipBackend(n, thick, thin, device=0, options=opts)
    b.set_step(s["dr1"], s["vol"], s["coldensh_LLS"], s["clumping"], s["temper"])
    b.set_sources(pos[sh], nf[sh]); b.load(ndens=nd, xh=xh)
    ev = pkg.Evolve(b); b.begin_step()
    ts = []
    for k in range(-5, 12):
        torch.cuda.synchronize(); t0 = time.perf_counter(); ev.iteration(k, s["dt"]); torch.cuda.synchronize()
        if k >= 0: ts.append(1e3 * (time.perf_counter() - t0))
    print(opts, "median ms", float(np.median(ts)), b.info().split("; chains ")[1].split("; exchanges")[0])
    b.close()
PY
