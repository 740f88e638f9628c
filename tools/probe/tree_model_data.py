"""Calibration data for a cost model of the sweeps: for every (scene, tree variant) the per-level structure (ADMM_HIP_VERBOSE lines, captured from
stderr) and the measured forward / backward time per ADMM iteration.  One JSON line per run on stdout.
usage: python tools/probe/tree_model_data.py   (GPU box)"""
import os, sys, json, subprocess, itertools
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
CHILD = r'''
import os, sys, json
sys.path.insert(0, %r)
from __graft_entry__ import load_package
pkg = load_package()
dims = tuple(int(v) for v in sys.argv[1:])
s = pkg.make_bar_system(*dims) if len(dims) == 3 else pkg.make_mixed_system(*dims)[0]
s.keep_z(False); s.initialize()
for _ in range(3): s.step(20)
s.enable_timing(1)
ph = dict(local_ms=0.0, solve_fwd_ms=0.0, solve_bwd_ms=0.0, total_ms=0.0)
for _ in range(2):
    s.step(20); tm = s.timing()
    for k in ph: ph[k] += tm[k] / 40.0
print("RESULT " + json.dumps(dict(nodes=s.info()["n_nodes"], levels=s.info()["n_levels"], **ph)))
''' % ROOT
scenes = [(13, 13, 50), (20, 20, 60), (24, 24, 75), (24, 24, 100)]
if os.environ.get("TREE_MODEL_SCENES"): scenes = [tuple(int(v) for v in q.split("x")) for q in os.environ["TREE_MODEL_SCENES"].split(",")]
variants = []
for leaf, merge, depth, rd in itertools.product((64, 128, 256, 512), (0, 100), (2, 3), (0, 4)):
    if merge == 0 and (depth == 3): continue
    variants.append({"ADMM_HIP_LEAF": str(leaf), "ADMM_HIP_MERGE": str(merge), "ADMM_HIP_MERGE_DEPTH": str(depth), "ADMM_HIP_ROOT_DEPTH": str(rd)})
for dims in scenes:
    for env in variants:
        e = dict(os.environ, ADMM_HIP_VERBOSE="1", **env)
        p = subprocess.run([sys.executable, "-c", CHILD] + [str(d) for d in dims], env=e, capture_output=True, text=True)
        lv = [l for l in p.stderr.splitlines() if l.startswith("admm_hip: level")]
        res = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
        if not res: print("FAILED", dims, env, p.stderr[-300:], file=sys.stderr); continue
        print(json.dumps(dict(dims=dims, env=env, level_lines=lv, **json.loads(res[0][7:]))), flush=True)
