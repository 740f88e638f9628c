import sys, time, os, subprocess, json
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for dims in ((16, 16, 65), (20, 20, 100), (32, 32, 163)):
    for leaf in (16, 64, 128, 256):
        env = dict(os.environ, ADMM_HIP_LEAF=str(leaf))
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--steps", "3", "--warmup", "1", "--dims"] + [str(d) for d in dims],
                           capture_output=True, text=True, env=env)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        p = d["roofline"]["phases_ms_per_iter"]
        print(dims, "leaf", leaf, "us/iter %.1f" % (d["ms_per_step"] / 20 * 1e3), "nnzL %d levels %d" % (d["config"]["nnz_L"], d["config"]["levels"]),
              "fwd %.3f bwd %.3f" % (p["solve_fwd_ms"], p["solve_bwd_ms"]), flush=True)
