import sys, os, subprocess, json
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for dims, merges in (((13, 13, 50), (0, 2000, 600)), ((16, 16, 65), (0, 5000, 1500)), ((32, 32, 163), (0, 40000, 8000, 2000))):
    for mg in merges:
        env = dict(os.environ, ADMM_HIP_MERGE=str(mg))
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--steps", "3", "--warmup", "1", "--dims"] + [str(d) for d in dims],
                           capture_output=True, text=True, env=env)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception:
            print(dims, mg, "FAILED", r.stderr[-500:]); continue
        p = d["roofline"]["phases_ms_per_iter"]
        print(dims, "merge", mg, "us/iter %.1f" % (d["ms_per_step"] / 20 * 1e3), "nnzL %d levels %d" % (d["config"]["nnz_L"], d["config"]["levels"]),
              "fwd %.3f bwd %.3f" % (p["solve_fwd_ms"], p["solve_bwd_ms"]), "factor_s %.1f" % d["config"]["factor_numeric_s"], flush=True)
