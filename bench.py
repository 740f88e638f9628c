#!/usr/bin/env python3
"""bench.py -- headline benchmark: ADMM iters/sec x #elements on the synthetic
1M-tet Neo-Hookean bar (BASELINE.json configs[3]).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A "step" is one frame = System::step() with 20 ADMM iterations (reference
deps/admm-elastic-sca/src/system/System.cpp:26-75): explicit forces, 20 x
(local step over every element, RHS assembly, pre-factored solve), velocity
update.  initialize() (ordering + factorization + upload) is excluded, as
BASELINE.md section 3 prescribes.  Positions/velocities stay resident in HBM.

Multi-GPU: elements shard across ranks and the partial right-hand sides meet in one
RCCL all-reduce per ADMM iteration.  Default (--shard subtree):
every rank owns whole subtrees of the elimination tree and the elements touching them,
so only the top separators' rows (a few hundred KB) are exchanged and only the top
levels of the solve are replicated; --shard contiguous is the plain form (contiguous
element ranges, whole RHS all-reduced, whole solve replicated).  Fixed total work:
"scaling": "strong".

Prints ONE JSON line on rank 0 (see the driver contract in the task statement)
with the extra objects "roofline" (dominant kernel, measured live with HIP
events on the kernel's own stream) and "cpu_baseline" (the compiled reference,
or the oracle port, timed on this host's cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s measured achievable)
VALU_ISSUE_SLOTS_PER_S = 1024 * 2.4e9 / 4.0   # NOMINAL: 256 CUs x 4 SIMDs at 2.4 GHz; one wave-wide VALU instruction occupies a SIMD for 4 cycles
# MEASURED on this part (tools/probe/valu_throughput.hip, profiles/r04/valu_roof.txt): with every SIMD of the chip saturated (8 waves of
# dependent fp64 chains each) a SIMD retires one wave-wide fp64 add / mul / fma / max / ldexp / compare / select per 2.00-2.15 ns -- 4 cycles at
# the ~2.0 GHz the chip sustains under fp64 load, not at 2.4 GHz -- and one v_rcp_f64 / v_rsq_f64 per 6.9 ns.  The floor below prices the
# launch's instruction counts (PMC) at the FASTEST measured rates.
VALU_NS_PER_INST_MEASURED = 2.00
VALU_NS_PER_TRANS_MEASURED = 6.92
RHS_BYTES_PER_NODE = 48.0      # rhs_gather_kernel: M x_bar read and b written per node (+ the slots, counted from info.rhs_slots)
# SURVEY.md section 8(d) prices the tet kernel at 472 B per tet and ADMM iteration (read 296 B + write 176 B) and the RHS assembly at
# 96 B per tet + 48 B per node: these ALGORITHMIC figures are the yardstick of roofline.achieved / frac (the contract's definition).
# What the round-3 kernels have to move at the least is less -- a production frame does not store z (72 B per tet: nobody reads it
# back, admm_hip_keep_z) and the RHS shares are summed per 64-tet block before they leave the kernel -- and is reported beside it
# (bytes_min, frac_of_min_bytes); roofline.traffic is what the counters saw.
LOCAL_BYTES_PER_TET = 472.0
LOCAL_BYTES_PER_TET_MIN = 400.0
RHS_BYTES_PER_TET = 96.0
ADMM_ITERS = 20


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--dims", type=int, nargs=3, default=[32, 32, 163], help="bar cubes nx ny nz (default: the 1M-tet bar)")
    p.add_argument("--config", choices=["bar", "mixed"], default="bar", help="bar = configs[3] (headline); mixed = configs[4]: 500k NH+StVK tets + 100k cloth tris")
    p.add_argument("--shard", choices=["subtree", "contiguous"], default="subtree",
                   help="N > 1: subtree = ranks own elimination subtrees + their elements, small per-iteration exchange (default); "
                        "contiguous = element ranges, full RHS all-reduce, replicated solve")
    p.add_argument("--timing-stride", type=int, default=10, help="HIP events around every k-th ADMM iteration of the timed region (1 = every iteration, all launches eager); "
                   "measured on identical frames (tools/probe/event_overhead.py): event-free 0.685-0.688 ms per iteration, k = 20 / 10: 0.690-0.693, k = 4: 0.695-0.696, k = 1: 0.71")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-extras", action="store_true", help="skip the class-API frame cost and the other BASELINE configs (N = 1 only)")
    p.add_argument("--cpu-dims", type=int, nargs=3, default=[16, 16, 65], help="bounded CPU-baseline sample (cubes)")
    return p.parse_args()


def cpu_baseline(dims):
    """The reference's own System::step (compiled from /root/reference into
    oracle/_ref/libadmm_ref.so, kind "reference"), or -- if that binary is not
    present -- our C restatement (kind "port"), timed on a bounded sample of the
    same workload: a smaller bar, same material/anchors/dt/iterations."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import checkers
    from __graft_entry__ import load_package
    pkg = load_package()
    kind = "reference" if checkers.have_ref() else "port"
    nx, ny, nz = dims          # the SAME bounded sample whichever checker is present
    x, tets = pkg.meshgen.bar(nx, ny, nz)
    m = pkg.meshgen.lumped_tet_mass(x, tets, 1000.0)
    s = checkers.Ref() if kind == "reference" else checkers.Oracle()
    s.settings(0.04, ADMM_ITERS)
    s.add_nodes(x.ravel(), np.repeat(m, 3))
    s.add_forces(pkg.KIND["TET_NH"], tets, [1e5, 1e5, 5])
    s.add_forces(pkg.KIND["ANCHOR"], pkg.meshgen.bar_anchor_nodes(nx, ny), [-1.0, 1.0])
    s.add_gravity([0.0, -9.8, 0.0])
    t0 = time.time()
    assert s.initialize()
    t_init = time.time() - t0
    s.time_steps(1)  # warm-up frame
    # the reference parallelises its local step with an OpenMP team of every hardware thread by default; on a 256-thread host
    # that is 3.6x slower than a moderate team (149.7 vs 41-43 ms per iteration on this sample): time a few team sizes on
    # consecutive frames and report the BEST one -- the baseline at its best, with the count it used
    lib = s.lib
    setter = getattr(lib, ("ref_" if kind == "reference" else "orc_") + "set_omp_threads", None)
    getter = getattr(lib, ("ref_" if kind == "reference" else "orc_") + "omp_threads")
    hw = os.cpu_count() or 1
    tried = {}
    teams = sorted({t for t in (16, 32, 64, getter(), hw) if 1 <= t <= hw}) if setter is not None else [getter()]
    frames = 1
    for team in teams:          # two single frames per team size, the faster one counts: the baseline at its best
        if setter is not None:
            setter(int(team))
        tried[int(team)] = min(s.time_steps(1), s.time_steps(1))
    cores = min(tried, key=tried.get)
    sec = tried[cores] * frames
    val = frames * ADMM_ITERS / sec * tets.shape[0]
    out = {"value": val, "unit": "ADMM iters/s x elements", "cores": int(cores), "kind": kind,
           "sample": "NH bar %dx%dx%d cubes = %d tets; 1 warm-up frame, then two frames of %d ADMM iters per OpenMP team size (%s threads), the fastest frame of the fastest team reported; initialize() %.1f s excluded; "
                     "%.1f ms/iter" % (nx, ny, nz, tets.shape[0], ADMM_ITERS, " / ".join(str(t) for t in teams), t_init, 1e3 * sec / (frames * ADMM_ITERS)),
           "ms_per_iter": 1e3 * sec / (frames * ADMM_ITERS), "cpu_model": _cpu_model(), "hardware_threads": hw,
           "ms_per_iter_by_team": {str(k): 1e3 * v / ADMM_ITERS for k, v in sorted(tried.items())}}
    if kind == "reference":       # which binary this was: built by oracle/Makefile from /root/reference in the build container
        import hashlib
        so = os.path.join(ROOT, "oracle", "_ref", "libadmm_ref.so")
        out["ref_so_sha256"] = hashlib.sha256(open(so, "rb").read()).hexdigest()
        out["ref_build"] = "g++ -std=c++11 -O2 -fopenmp, vendored Eigen 3.2.5 + cppoptlib, sources compiled where they lie (oracle/Makefile ref)"
    # the reference at the HEADLINE size, measured once when the full-size parity fixture was generated (make_golden_fullsize.py)
    try:
        g = np.load(os.path.join(ROOT, "tests", "golden", "traj_bar_1M.npz"))
        n_full = 6 * int(np.prod(g["dims"]))
        out["full_size_reference"] = {"value": int(g["iters"]) / float(g["ref_frame_s"]) * n_full, "frame_s": float(g["ref_frame_s"]), "initialize_s": float(g["ref_initialize_s"]),
                                      "threads": int(g["ref_threads"]), "cpu_model": str(g["ref_cpu"]), "build": str(g["ref_build"]),
                                      "note": "compiled reference, %d tets, one frame of %d ADMM iterations, build container (not this host)" % (n_full, int(g["iters"]))}
    except Exception:
        pass
    return out


def gpu_same_sample(pkg, dims, n_frames):
    """The GPU on the SAME bounded sample the CPU leg times (same bar, same frames of the same simulation: frame 1 = warm-up, then n_frames
    single frames of 20 ADMM iterations each, every one timed on its own like the CPU leg does) -- the only like-for-like pair of figures in
    this line.  State resident, graph replay (this size is below 100k nodes), initialize excluded."""
    nx, ny, nz = dims
    s = pkg.make_bar_system(nx, ny, nz)
    s.keep_z(False)
    s.initialize()
    s.step(ADMM_ITERS); s.sync()
    per = []
    for _ in range(n_frames):
        t = time.perf_counter()
        s.step(ADMM_ITERS); s.sync()
        per.append((time.perf_counter() - t) / ADMM_ITERS)
    assert np.isfinite(s.m_x).all()
    best = min(per)
    return {"ms_per_iter": 1e3 * best, "ms_per_iter_mean": 1e3 * float(np.mean(per)), "value": s.n_tets / best, "unit": "ADMM iters/s x elements",
            "frames": "2..%d of the simulation, each timed on its own; the fastest one reported (what the CPU leg reports), the mean beside it" % (n_frames + 1),
            "sample": "NH bar %dx%dx%d cubes = %d tets -- the cpu_baseline sample" % (nx, ny, nz, s.n_tets), "x_checksum": float(np.abs(s.m_x).sum())}


def other_configs(pkg, torch, steps):
    """The throughput variants of the other BASELINE.json configs on this GPU, in this run:
    configs[1] 5,400-tet NH bar, configs[2] 50,700-tet StVK bar, configs[4] mixed scene (1 GPU).  Same protocol as the
    headline (initialize excluded, 1 warm-up frame, `steps` frames of 20 ADMM iterations, state resident), default launch
    mode (graph replay)."""
    res = {}

    def measure(s, n_el, label, warm=3):
        s.keep_z(False)
        s.initialize()
        for _ in range(warm):       # warm-up frames (graph capture, cost-ordered launch, clocks) up to the STATED state ...
            s.step(ADMM_ITERS)
        s.sync()
        # ... which is kept (x, v, every force's u and warm start: the checkpoint of DESIGN section 7) and restored before each of the three timed
        # runs: all three time the SAME frames warm+1 .. warm+steps of the simulation (the local step's cost moves with the deformation:
        # consecutive windows of one simulation are not repeat measurements)
        ck = dict(x=s.m_x.copy(), v=s.m_v.copy(), loc=[s.read_local(b) for b in range(len(s.batches))])

        def rewind():
            s.m_x = ck["x"]; s.m_v = ck["v"]
            for bi, loc in enumerate(ck["loc"]):
                s.write_local(bi, u=loc["u"], state=loc["state"] if pkg.KIND_STATE[s.batches[bi][0]] else None)
            s.sync()
        runs, sums = [], []
        for _ in range(3):
            rewind()
            t = time.perf_counter()
            for _ in range(steps):
                s.step(ADMM_ITERS)
            s.sync()
            runs.append((time.perf_counter() - t) / steps)
            sums.append(float(np.abs(s.m_x).sum()))
        t = sorted(runs)[1]
        assert np.isfinite(s.m_x).all()
        inf = s.info()
        res[label] = {"value": ADMM_ITERS / t * n_el, "elements": int(n_el), "nodes": int(inf["n_nodes"]), "ms_per_iter": 1e3 * t / ADMM_ITERS,
                      "ms_per_iter_runs": [round(1e3 * r / ADMM_ITERS, 5) for r in runs],
                      "frames_timed": "%d..%d of the simulation in every run (state rewound between runs)" % (warm + 1, warm + steps),
                      "same_frames_every_run": bool(sums[0] == sums[1] == sums[2]),
                      "solve": "explicit inverse (one kernel)" if inf["dense_solve"] else "%d levels" % inf["n_levels"]}
    try:
        s = pkg.make_bar_system(10, 10, 9); measure(s, s.n_tets, "configs[1] NH bar 10x10x9 = 5400 tets"); del s
        s = pkg.make_bar_system(13, 13, 50, kind=pkg.KIND["TET_STVK"]); measure(s, s.n_tets, "configs[2] StVK bar 13x13x50 = 50700 tets"); del s
        s, _ = pkg.make_mixed_system(26, 26, 123, 158, 158); measure(s, s.n_elements, "configs[4] mixed: 498888 NH+StVK tets + 99856 cloth tris (+ hinges, anchors), 1 GPU"); del s
    except Exception as e:  # noqa: BLE001 -- side figures; never lose the headline over them
        res["error"] = repr(e)
    return res


def measured_roof(pm, sec):
    """The launch's VALU instruction counts (PMC) priced at the issue rates MEASURED on this chip with saturated SIMDs
    (tools/probe/valu_throughput.hip): the time below which no schedule of these instructions can finish."""
    n = pm["SQ_INSTS_VALU"]["per_launch"]
    tr = pm.get("SQ_INSTS_VALU_TRANS_F64", {}).get("per_launch")
    if tr is None:
        return {}
    floor_s = ((n - tr) * VALU_NS_PER_INST_MEASURED + tr * VALU_NS_PER_TRANS_MEASURED) * 1e-9 / 1024.0
    return {"trans_f64_insts": tr, "issue_floor_ms_measured_rates": 1e3 * floor_s, "valu_frac_measured_rates": floor_s / sec if sec > 0 else 0.0,
            "measured_rates": "%.2f ns per wave-wide fp64 VALU instruction and SIMD, %.2f ns per v_rcp/v_rsq_f64 (saturated chip, profiles/r04/valu_roof.txt)" % (VALU_NS_PER_INST_MEASURED, VALU_NS_PER_TRANS_MEASURED)}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def install_rccl(s, torch, dist, rank, world, local_rank):
    """The all-reduce inside the library (admm_hip_rccl_init): ncclAllReduce on the solver's stream, issued by the C step
    loop itself.  The ncclUniqueId travels from rank 0 through the torch process group.  Nothing here raises before every
    rank has taken part in the same collectives; the outcome is a consensus (MIN over the ranks): either all ranks use the
    communicator or all fall back."""
    import threading
    dev = torch.device("cuda", local_rank)
    ok = True
    uid = np.zeros(128, np.uint8)
    if rank == 0:
        try:
            uid = s.rccl_unique_id()
        except Exception as e:  # noqa: BLE001
            print("bench: ncclGetUniqueId failed: %r" % (e,), file=sys.stderr)
            ok = False
    t = torch.from_numpy(uid.copy()).to(dev)
    dist.broadcast(t, src=0)
    uid = t.cpu().numpy()
    flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if float(flag.item()) < 1.0:
        raise RuntimeError("ncclGetUniqueId failed on rank 0")
    state = {"ok": False, "cancel": False}
    seen = 0
    lock = threading.Lock()

    def init():     # collective; the library selects the context's device before ncclCommInitRank
        try:
            s.rccl_init(uid, rank, world)
        except Exception as e:  # noqa: BLE001
            print("bench: rank %d: %r" % (rank, e), file=sys.stderr)
            return
        with lock:      # a communicator that arrives after the deadline must not stay installed: the other ranks have moved on
            if state["cancel"]:
                s.set_rccl_comm(None)
            else:
                state["ok"] = True
    th = threading.Thread(target=init, daemon=True)
    th.start()
    th.join(timeout=float(os.environ.get("ADMM_BENCH_RCCL_INIT_TIMEOUT", "120")))
    with lock:
        if not state["ok"]:
            state["cancel"] = True
    flag.fill_(1.0 if state["ok"] else 0.0)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    good = float(flag.item()) >= 1.0
    if good:    # one checked all-reduce through the new communicator: 1 + 2 + ... + world
        probe = torch.full((8,), float(rank + 1), dtype=torch.float64, device=dev)
        probe[4:] = 1.0                                 # second half: a head count of the ranks the communicator really joins
        torch.cuda.synchronize()
        try:
            s.debug_allreduce(probe.data_ptr(), probe.numel())
            fine = bool((probe[:4] == world * (world + 1) / 2.0).all().item()) and bool((probe[4:] == float(world)).all().item())
            seen = int(round(float(probe[4].item())))
        except Exception:  # noqa: BLE001
            fine = False
        flag.fill_(1.0 if fine else 0.0)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        good = float(flag.item()) >= 1.0
    if not good:
        if state["ok"]:
            s.set_rccl_comm(None)
        raise RuntimeError("in-library RCCL communicator unavailable on some rank (here: init ok = %r)" % (state["ok"],))
    return seen


class Watchdog:
    """N > 1: a frame that makes no progress must end the run with a message, not hold the GPUs until the driver's own limit.
    The main thread calls beat(what[, limit_s]) at every step of the protocol; a daemon thread ends the process -- a plain
    os._exit(3), never a re-exec -- when the current phase has lasted longer than its limit, naming rank and phase (the
    launcher then stops the other ranks and relays the non-zero code).  After the warm-up the per-frame limit is
    max(ADMM_BENCH_FRAME_TIMEOUT_MIN [30 s], 100 x the warm-up frame time)."""

    def __init__(self, rank, world, limit_s=900.0, poll_s=0.25):
        import threading
        self.rank, self.world, self.poll_s = rank, world, poll_s
        self.t, self.what, self.limit, self.on = time.monotonic(), "start-up", float(limit_s), True
        self.lock = threading.Lock()
        self.th = threading.Thread(target=self._run, daemon=True)
        self.th.start()

    def beat(self, what, limit_s=None):
        with self.lock:
            self.t, self.what = time.monotonic(), what
            if limit_s is not None:
                self.limit = float(limit_s)

    def stop(self):
        with self.lock:
            self.on = False

    def _run(self):
        while True:
            time.sleep(self.poll_s)
            with self.lock:
                if not self.on:
                    return
                late = time.monotonic() - self.t
                what, limit = self.what, self.limit
            if late > limit:
                print("bench: rank %d of %d: no progress for %.1f s in phase '%s' (limit %.1f s) -- a hung collective or kernel; "
                      "exiting with code 3" % (self.rank, self.world, late, what, limit), file=sys.stderr)
                sys.stderr.flush()
                os._exit(3)


class _NoWatchdog:
    def beat(self, what, limit_s=None):
        pass

    def stop(self):
        pass


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: this process has not imported torch or touched a GPU
    yet, so it starts the N ranks as a fresh child (`python -m torch.distributed.run`, one process per GPU, rendezvous on
    127.0.0.1), relays the child's stdout (rank 0's JSON line) and stderr, and exits with the child's return code.  Never an
    exec, never after a HIP call."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(int(os.environ.get("ADMM_BENCH_MASTER_PORT", "0")) or _free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // a.gpus)))
    print("bench: --gpus %d without a launcher: starting %s" % (a.gpus, " ".join(cmd[1:8]) + " ... bench.py"), file=sys.stderr)
    sys.stderr.flush()
    return subprocess.call(cmd, env=env)


def main():
    a = parse()
    if a.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "RANK" not in os.environ and a.gpus > 1:
        sys.exit(self_launch(a))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:     # a silent 1-rank run under `--gpus 8` (or the reverse) would be reported as the wrong N
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (a.gpus, world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL between processes); before anything touches the GPU
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # test hooks (1-GPU boxes): ADMM_BENCH_SHARE_GPU=1 puts every rank on cuda:0, ADMM_BENCH_BACKEND=gloo replaces RCCL;
    # ADMM_BENCH_FAKE_WORLD=N (with BENCH_TIMING_EXPERIMENT=1): time ONE rank's kernels of an N-rank run with a no-op all-reduce
    # (wrong numbers, right launch sequence) -- what a rank computes per iteration, communication excluded
    fake_world = int(os.environ.get("ADMM_BENCH_FAKE_WORLD", "0"))
    if os.environ.get("ADMM_BENCH_SHARE_GPU"):
        local_rank = 0
    backend = os.environ.get("ADMM_BENCH_BACKEND", "nccl")
    if torch.cuda.device_count() <= local_rank:     # launcher that pins one visible device per rank
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dist = None
    wd = Watchdog(rank, world, float(os.environ.get("ADMM_BENCH_STARTUP_TIMEOUT", "900"))) if (world > 1 or os.environ.get("ADMM_BENCH_WATCHDOG") == "1") else _NoWatchdog()
    wd.beat("process group + scene set-up")
    fake_dist = fake_world > 1 and os.environ.get("ADMM_BENCH_FAKE_DIST") == "1" and "RANK" in os.environ
    if world > 1 or fake_dist:    # (fake_dist: a 1-rank RCCL group under torchrun whose all-reduce the fake-world run really calls -- wrong sums, real mechanics)
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    from __graft_entry__ import load_package
    pkg = load_package()

    nx, ny, nz = a.dims
    # the solver runs on torch's CURRENT stream (the fallback all-reduce through torch.distributed is ordered on it); made a
    # real stream, not the legacy default one, so that the library can capture its ADMM iteration as a HIP graph
    torch.cuda.set_stream(torch.cuda.Stream(device=local_rank))
    stream = torch.cuda.current_stream()
    t0 = time.time()
    if a.config == "mixed":      # BASELINE.json configs[4]: 26x26x123 cubes = 498,888 tets (half NH, half StVK) + 158x158 sym-plane cloth
        if a.dims == [32, 32, 163]:
            nx, ny, nz = 26, 26, 123
        s, _desc = pkg.make_mixed_system(nx, ny, nz, 158, 158, device_id=local_rank, rank=int(os.environ.get("ADMM_BENCH_FAKE_RANK", "0")) if fake_world > 1 else rank,
                                         world=fake_world if fake_world > 1 else world, stream=stream.cuda_stream, shard_mode=a.shard)
        s.n_tets = s.n_elements
        a.no_cpu_baseline = True
    else:
        s = pkg.make_bar_system(nx, ny, nz, device_id=local_rank, rank=int(os.environ.get("ADMM_BENCH_FAKE_RANK", "0")) if fake_world > 1 else rank,
                                world=fake_world if fake_world > 1 else world, stream=stream.cuda_stream, shard_mode=a.shard)
    if fake_world > 1:
        s.set_factor_local(False)      # ONE rank's share with identity "sums": the top of the tree can only be factored whole
    if fake_world > 1 and not fake_dist:
        s.set_allreduce(lambda ptr, count, strm: 0)
    hang = None
    if world > 1 or fake_dist:
        n3 = None
        holder = {}

        class _Ptr:  # zero-copy view of the C library's RHS buffer for torch.distributed
            def __init__(self, ptr, count):
                self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}

        hang = {"calls": 0, "rank": int(os.environ.get("ADMM_BENCH_TEST_HANG_RANK", "-1")), "after": int(os.environ.get("ADMM_BENCH_TEST_HANG_AFTER", "0")), "armed": False}

        def torch_hook(ptr, count, strm):
            hang["calls"] += 1
            if hang["armed"] and hang["rank"] == rank and hang["calls"] > hang["after"]:      # test hook (tests/test_sharding.py): this rank never reaches the collective again
                time.sleep(1e6)
            t = holder.get((ptr, count))      # (keyed by address AND count: the factorization's exchange buffers are freed again, and the allocator hands their addresses to other buffers)
            if t is None:
                t = torch.as_tensor(_Ptr(ptr, count), device=torch.device("cuda", local_rank))
                holder[(ptr, count)] = t
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return 0
        # RCCL inside the library: ncclAllReduce on the solver's stream straight from the C step loop.  torch.distributed's
        # all_reduce costs ~0.25 ms of host time per call (dispatcher, work object, stream events) -- as much as a whole ADMM
        # iteration takes at 8 GPUs -- and a Python hook re-enters the interpreter every iteration.  The communicator is
        # bootstrapped through the torch process group; any failure (on any rank) makes ALL ranks fall back to the torch hook.
        s.set_allreduce(torch_hook)
        comm_path = "torch.distributed.all_reduce (hook, backend %s)" % backend
        ranks_seen = None
        if backend == "nccl" and os.environ.get("ADMM_BENCH_TORCH_ALLREDUCE") != "1":
            try:
                ranks_seen = install_rccl(s, torch, dist, rank, world if not fake_dist else 1, local_rank)
                comm_path = "ncclAllReduce inside libadmm_hip.so"
            except Exception as e:  # noqa: BLE001
                if rank == 0:
                    print("bench: in-library RCCL unavailable (%r), using torch.distributed" % (e,), file=sys.stderr)
        if ranks_seen is None:       # the hook path: the same head count through the process group the hook uses
            ones = torch.ones(1, dtype=torch.float64, device=torch.device("cuda", local_rank) if backend == "nccl" else "cpu")
            dist.all_reduce(ones, op=dist.ReduceOp.SUM)
            ranks_seen = int(round(float(ones.item())))
        if ranks_seen != (world if not fake_dist else 1):      # an N-rank launch whose collective joins fewer ranks is not an N-GPU run
            raise SystemExit("bench.py: rank %d: the all-reduce path joins %d ranks, expected %d" % (rank, ranks_seen, world))
    s.keep_z(False)      # production frames: the tet batches' z is never read back (what host/admm/System.hpp does too)
    wd.beat("initialize (ordering, factorization, upload)")
    s.initialize()
    t_init = time.time() - t0
    info = s.info()
    n_tets = s.n_tets

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    wd.beat("barrier before the warm-up", float(os.environ.get("ADMM_BENCH_WARMUP_TIMEOUT", "300")))
    sync_all()
    t_w = time.perf_counter()
    for f in range(a.warmup):
        wd.beat("warm-up frame %d" % f)
        s.step(ADMM_ITERS)
    wd.beat("sync after the warm-up")
    sync_all()
    # per-frame limit of the timed region: 100 x the warm-up frame (it holds the first collective's lazy set-up and the graph capture)
    frame_limit = max(float(os.environ.get("ADMM_BENCH_FRAME_TIMEOUT_MIN", "30")), 100.0 * (time.perf_counter() - t_w) / max(a.warmup, 1))
    if os.environ.get("ADMM_BENCH_FRAME_TIMEOUT_MAX"):      # (tests: a fixed upper end, whatever the warm-up frame of a cold box took)
        frame_limit = min(frame_limit, float(os.environ["ADMM_BENCH_FRAME_TIMEOUT_MAX"]))
    # HIP events on the solver's stream around the phases of every TIMING_STRIDE-th ADMM iteration of the timed region (a frame's events
    # are read back after the NEXT frame has been queued); the other iterations run event-free.  An event is a barrier packet (~5 us of
    # lost launch overlap each): around every iteration they cost 3.7 % at one GPU, around every 10th 0.7 % (tools/probe/event_overhead.py).
    s.enable_timing(a.timing_stride)
    phase = dict(local_ms=0.0, rhs_ms=0.0, allreduce_ms=0.0, solve_fwd_ms=0.0, solve_bwd_ms=0.0, total_ms=0.0)
    if hang is not None:
        hang["calls"] = 0; hang["armed"] = True      # (test hook: ADMM_BENCH_TEST_HANG_AFTER counts the collectives of the TIMED region, however many initialize and the warm-up issued)
    t0 = time.perf_counter()
    for f in range(a.steps):
        wd.beat("timed frame %d: queueing" % f, frame_limit)
        s.step(ADMM_ITERS)
        if f > 0:      # frame f - 1's events, read AFTER frame f has been queued: the GPU never waits for the host between two frames
            wd.beat("timed frame %d: waiting for its events" % (f - 1))
            tm = s.timing_previous()
            for k in phase:
                phase[k] += tm[k]
    wd.beat("timed frame %d: waiting for its events" % (a.steps - 1))
    tm = s.timing()      # the last frame's (waits for its last event; positions stay on the device)
    for k in phase:
        phase[k] += tm[k]
    wd.beat("sync + barrier after the timed region")
    sync_all()
    elapsed = time.perf_counter() - t0
    wd.beat("results (MAX over the ranks, per-rank gather, read-back)", 300.0)
    rccl_async = s.rccl_async_error() if (world > 1 or fake_dist) else None      # ncclCommGetAsyncError after the timed region: raises (non-zero exit) if the communicator is broken
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    xs = s.m_x
    if not os.environ.get("BENCH_TIMING_EXPERIMENT"):   # (kernel-timing experiments with deliberately wrong arithmetic set this)
        assert np.isfinite(xs).all(), "non-finite positions"

    iters_total = a.steps * ADMM_ITERS
    value = iters_total / elapsed * n_tets
    # ---- roofline of the dominant kernel (per launch = per ADMM iteration) ----
    local_s = phase["local_ms"] * 1e-3 / iters_total
    fwd_s = phase["solve_fwd_ms"] * 1e-3 / iters_total
    bwd_s = phase["solve_bwd_ms"] * 1e-3 / iters_total
    # what THIS rank streams: its own subtrees' panels + the replicated top (one rank / contiguous shards: the whole factor), from the
    # library's own accounting (admm_hip_info.sweep_entries_*, csrc/partition.cpp shard_accounting); vectors: 72 B per node it sweeps over
    sweep_nodes = info["nodes_own"] + info["nodes_top"]
    fwd_bytes = 8.0 * (info["sweep_entries_own"] + info["sweep_entries_top"]) + 72.0 * sweep_nodes
    bwd_bytes = 8.0 * (info["sweep_entries_own"] + info["sweep_entries_top_bwd"]) + 72.0 * sweep_nodes
    panel_bytes = info["nnz_L"] * 8.0 + info["n_nodes"] * 24.0 * 3
    n_local = [int(s.local_elements(b).size) for b in range(len(s.batches))]      # this rank's elements per batch
    n_tets_local = n_local[0] if a.config == "bar" else sum(n for (k, _), n in zip(s.batches, n_local) if k in (pkg.KIND["TET_NH"], pkg.KIND["TET_STVK"]))
    local_name, local_bytes = "project_tet_kernel<NH>", LOCAL_BYTES_PER_TET * n_tets_local
    local_bytes_min = LOCAL_BYTES_PER_TET_MIN * n_tets_local
    if a.config == "mixed":      # one launch for all batches (project_multi_kernel): SURVEY 8(d)'s bytes per element of every kind in the scene
        per_kind = {"TET_NH": 472.0, "TET_STVK": 472.0, "TET_LINEAR": 472.0, "TET_VOLUME": 472.0, "TRI_STRAIN": 284.0, "TRI_AREA": 284.0, "TRI_FUNG": 284.0,
                    "BEND": 368.0, "SPRING": 144.0, "ANCHOR": 124.0, "COLLISION": 124.0}
        names = {v: k for k, v in pkg.KIND.items()}
        local_name = "project_multi_kernel (the scene's whole local step: " + " + ".join("%d %s" % (n, names.get(k, str(k))) for k, n in s.batches) + ")"
        local_bytes = sum(per_kind.get(names.get(k, ""), 472.0) * n for (k, _), n in zip(s.batches, n_local))
        local_bytes_min = local_bytes
    cands = {
        local_name: (local_bytes, local_s),
        "solve_fwd (gather+panel kernels, all levels)": (fwd_bytes, fwd_s),
        "solve_bwd_kernel (all levels)": (bwd_bytes, bwd_s),
    }
    dom = max(cands, key=lambda k: cands[k][1])
    by, sec = cands[dom]
    ach = by / sec / 1e9 if sec > 0 else 0.0
    # HBM traffic of the dominant kernel from the PMC passes committed under profiles/
    # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; same workload, 1 GPU).
    # gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half the bytes of coalesced streaming reads -- calibrated on
    # this code's own 8-B-per-lane pattern: epilogue_kernel reads 2 x 4.29 MB and shows 4.30 MB, prologue_kernel 12.86 -> 6.44 MB;
    # WRITE_SIZE is exact (8.57 / 12.86 MB).  So traffic = 2 x FETCH_SIZE + WRITE_SIZE.
    traffic = None
    valu = None
    sweeps = None
    bound = "hbm"
    pmc_file = os.path.join("profiles", os.environ.get("ADMM_BENCH_PMC", "r06/pmc_1M.json"))
    if not os.path.exists(os.path.join(ROOT, pmc_file)):
        pmc_file = os.path.join("profiles", "r05/pmc_1M.json")
    if not os.path.exists(os.path.join(ROOT, pmc_file)):
        pmc_file = os.path.join("profiles", "r04/pmc_1M.json")
    pmc_stamp_ok = None      # True / False: the counter file carries the hash of the sources it was collected on (None: an unstamped, older file)
    try:
        stamp = json.load(open(os.path.join(ROOT, pmc_file))).get("csrc_sha256")
        if stamp:
            pmc_stamp_ok = bool(stamp == pkg._build.source_hash())
    except Exception:  # noqa: BLE001
        pass
    try:
        if (nx, ny, nz) == (32, 32, 163) and world == 1:
            kern = json.load(open(os.path.join(ROOT, pmc_file)))["kernels"]
            tet_key = [k for k in kern if "project_tet_kernel<0, 5" in k][0]
            iters_pmc = float(kern[tet_key]["FETCH_SIZE"]["launches"])       # the PMC passes ran this many ADMM iterations (one tet launch each)

            def sweep_traffic(match):      # bytes per ADMM iteration over every kernel of a sweep: 2 x FETCH_SIZE + WRITE_SIZE (KiB per launch x launches)
                tot = 0.0
                for k, v in kern.items():
                    if any(m in k for m in match):
                        tot += (2.0 * v["FETCH_SIZE"]["per_launch"] * v["FETCH_SIZE"]["launches"] + v["WRITE_SIZE"]["per_launch"] * v["WRITE_SIZE"]["launches"]) * 1024.0
                return tot / iters_pmc
            fwd_alg = info["nnz_L"] * 8.0 + info["n_nodes"] * 24.0 * 3
            fwd_tr, bwd_tr = sweep_traffic(("solve_fwd_", "root_gather_kernel", "root_product_kernel")), sweep_traffic(("solve_bwd_kernel",))
            sweeps = {"forward": {"algorithmic_bytes": fwd_alg, "traffic": fwd_tr, "traffic_over_algorithmic": fwd_tr / fwd_alg, "ms": fwd_s * 1e3, "GB/s": fwd_alg / fwd_s / 1e9 if fwd_s > 0 else 0.0},
                      "backward": {"algorithmic_bytes": fwd_alg, "traffic": bwd_tr, "traffic_over_algorithmic": bwd_tr / fwd_alg, "ms": bwd_s * 1e3, "GB/s": fwd_alg / bwd_s / 1e9 if bwd_s > 0 else 0.0}}
            if dom.startswith("project_tet_kernel"):
                pm = kern[tet_key]
                traffic = (2.0 * pm["FETCH_SIZE"]["per_launch"] + pm["WRITE_SIZE"]["per_launch"]) * 1024.0
                # the kernel is fp64-VALU bound, not HBM bound.  valu_frac: the launch's wave-wide VALU instructions x the 4 cycles each
                # occupies a SIMD, over the chip's issue capacity for the launch's duration (a LOWER bound on VALU occupancy: fp64
                # multiplies, FMAs, divisions' and square roots' helper sequences take more than 4 cycles)
                busy = pm["SQ_ACTIVE_INST_VALU"]["per_launch"] / pm["SQ_WAVE_CYCLES"]["per_launch"]
                valu = {"valu_insts_per_launch": pm["SQ_INSTS_VALU"]["per_launch"], "fma_f64_insts": pm["SQ_INSTS_VALU_FMA_F64"]["per_launch"],
                        "valu_busy_per_wave": busy, "waves_per_simd": 2, "simd_valu_issue_frac": min(1.0, 2 * busy),
                        "lane_utilisation": pm["SQ_THREAD_CYCLES_VALU"]["per_launch"] / (64.0 * pm["SQ_ACTIVE_INST_VALU"]["per_launch"]),
                        "issue_floor_ms": 1e3 * pm["SQ_INSTS_VALU"]["per_launch"] / VALU_ISSUE_SLOTS_PER_S,
                        "valu_frac": (pm["SQ_INSTS_VALU"]["per_launch"] / VALU_ISSUE_SLOTS_PER_S) / sec if sec > 0 else 0.0}
                valu.update(measured_roof(pm, sec))
                bound = "valu"
        if a.config == "mixed" and world == 1 and dom.startswith("project_multi_kernel"):      # the scene's one-launch local step: counters of tools/pmc_collect.sh with PMC_MIXED=1
            pmc_file = os.path.join("profiles", "r06", "pmc_mixed.json")
            if not os.path.exists(os.path.join(ROOT, pmc_file)):
                pmc_file = os.path.join("profiles", "r05", "pmc_mixed.json")
            if not os.path.exists(os.path.join(ROOT, pmc_file)):
                pmc_file = os.path.join("profiles", "r04", "pmc_mixed.json")
            kern = json.load(open(os.path.join(ROOT, pmc_file)))["kernels"]
            pm = kern[[k for k in kern if "project_multi_kernel" in k][0]]
            traffic = (2.0 * pm["FETCH_SIZE"]["per_launch"] + pm["WRITE_SIZE"]["per_launch"]) * 1024.0
            busy = pm["SQ_ACTIVE_INST_VALU"]["per_launch"] / pm["SQ_WAVE_CYCLES"]["per_launch"]
            valu = {"valu_insts_per_launch": pm["SQ_INSTS_VALU"]["per_launch"], "fma_f64_insts": pm["SQ_INSTS_VALU_FMA_F64"]["per_launch"],
                    "valu_busy_per_wave": busy, "waves_per_simd": 2, "simd_valu_issue_frac": min(1.0, 2 * busy),
                    "lane_utilisation": pm["SQ_THREAD_CYCLES_VALU"]["per_launch"] / (64.0 * pm["SQ_ACTIVE_INST_VALU"]["per_launch"]),
                    "issue_floor_ms": 1e3 * pm["SQ_INSTS_VALU"]["per_launch"] / VALU_ISSUE_SLOTS_PER_S,
                    "valu_frac": (pm["SQ_INSTS_VALU"]["per_launch"] / VALU_ISSUE_SLOTS_PER_S) / sec if sec > 0 else 0.0}
            valu.update(measured_roof(pm, sec))
            bound = "valu"
    except Exception as e:  # noqa: BLE001 -- counters are side information
        traffic = None
        print("bench: PMC summary %s unusable: %r" % (pmc_file, e), file=sys.stderr)
    # the whole ADMM iteration against the HBM roof: algorithmic bytes of every kernel of one iteration over the iteration's time
    it_s = phase["total_ms"] * 1e-3 / iters_total
    # (the RHS slots: written once by the local kernels, read once by the gather -- one per (64-tet block, node) since the block-level pre-reduction)
    slots_local = info.get("rhs_slots", 4 * n_tets_local)
    it_bytes = (LOCAL_BYTES_PER_TET + RHS_BYTES_PER_TET) * n_tets_local + RHS_BYTES_PER_NODE * sweep_nodes + fwd_bytes + bwd_bytes
    it_bytes_min = LOCAL_BYTES_PER_TET_MIN * n_tets_local + 2.0 * 24.0 * slots_local + RHS_BYTES_PER_NODE * sweep_nodes + fwd_bytes + bwd_bytes
    iteration = None
    if a.config == "bar":
        iteration = {"bytes": it_bytes, "ms": it_s * 1e3, "GB/s": it_bytes / it_s / 1e9 if it_s > 0 else 0.0, "frac": (it_bytes / it_s / 1e9 / HBM_PEAK_GBS) if it_s > 0 else 0.0,
                     "bytes_min": it_bytes_min, "frac_of_min_bytes": (it_bytes_min / it_s / 1e9 / HBM_PEAK_GBS) if it_s > 0 else 0.0,
                     "what": "SURVEY 8(d) algorithmic bytes -- tet kernel 472 B/tet + RHS assembly 96 B/tet + 48 B/node + the factor panels and vectors once per sweep -- over the mean "
                             "ADMM iteration (HIP events, total_ms); bytes_min: what the kernels must move at the least (400 B/tet without z, one 24-byte slot per (64-tet block, node) "
                             "written and read once)" + ("; N > 1: THIS rank's elements, own subtrees and the replicated top (rank 0's figures; per_rank has every rank's), "
                             "the collective's time included in ms" if (world > 1 or fake_world > 1) else "")}
    if dom.startswith("project_multi_kernel"): bound = "valu"      # the same fp64 prox arithmetic as the tet kernel (no counter passes are kept for this scene: valu stays null)
    roof = {"bound": bound, "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": traffic, "bytes_per_launch": by, "avg_launch_ms": sec * 1e3, "valu": valu,
            "bytes_per_launch_min": (local_bytes_min if dom.startswith("project_") else by),
            "frac_of_min_bytes": ((local_bytes_min if dom.startswith("project_") else by) / sec / 1e9 / HBM_PEAK_GBS) if sec > 0 else 0.0,
            # what the launch really moved over what it took: the counters' bytes (traffic) / this run's launch time -- an ACHIEVED rate;
            # `achieved` above is SURVEY 8(d)'s yardstick (472 B per tet incl. the 72 B of z a production frame never stores)
            "achieved_moved": (traffic / sec / 1e9) if (traffic and sec > 0) else None,
            "frac_moved": (traffic / sec / 1e9 / HBM_PEAK_GBS) if (traffic and sec > 0) else None,
            "valu_frac_measured_rates": (valu or {}).get("valu_frac_measured_rates"),
            "valu_frac": (valu or {}).get("valu_frac"),
            "note": ("bound = valu: the dominant kernel is limited by fp64 VALU issue / dependency latency; achieved / peak / frac keep the HBM yardstick "
                     "the contract asks for, valu_frac is the fraction of the roof that actually binds" if bound == "valu" else None),
            # achieved / avg_launch_ms are THIS run's HIP events; traffic / valu / sweeps.traffic are PMC counters and cannot be collected in
            # the same process: they come from the committed rocprofv3 --pmc passes of the same command (tools/pmc_collect.sh)
            "traffic_source": (pmc_file if (traffic is not None or sweeps is not None) else None),
            "traffic_source_matches_this_tree": pmc_stamp_ok,
            "iteration": iteration, "sweeps": sweeps,
            "phases_ms_per_iter": {k: v / iters_total for k, v in phase.items()},
            "all": {k: {"GB/s": (v[0] / v[1] / 1e9 if v[1] > 0 else 0.0), "ms": v[1] * 1e3} for k, v in cands.items()}}

    out = {
        "metric": "ADMM iters/sec x #elements, 1M-tet Neo-Hookean; 1/2/4/8 MI355X",
        "value": value, "unit": "ADMM iters/s x elements", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": ("NH bar %dx%dx%d cubes (Kuhn split) = %d tets, %d nodes, mu=lambda=1e5, max_iterations 5, rho 1000, h 0.05, "
                                "z=0 face anchored, g=-9.8, dt 0.04, %d ADMM iters/frame" % (nx, ny, nz, n_tets, info["n_nodes"], ADMM_ITERS)) if a.config == "bar" else
                               ("mixed scene: bar %dx%dx%d (half NH, half StVK tets) + 158x158 sym-plane cloth (triangle strain + bend) + anchors = %d "
                                "tets+tris, %d nodes, %d ADMM iters/frame" % (nx, ny, nz, n_tets, info["n_nodes"], ADMM_ITERS)),
                   "admm_iters_per_step": ADMM_ITERS,
                   "launch_mode": ("every ADMM iteration launched eagerly with HIP events between its phases" if a.timing_stride <= 1 else
                                   "HIP events between the phases of every %d-th ADMM iteration (eager launches; %d samples per kernel feed roofline), "
                                   "event-free launches otherwise (eager at this size; one graph replay per iteration below 100k nodes)" % (a.timing_stride, a.steps * (ADMM_ITERS // a.timing_stride))),
                   "allreduce": (comm_path if (world > 1 or fake_dist) else None), "parallelism": ("1 GPU" if world == 1 else
                                   (("x%d: elements and elimination subtrees per rank, %s" % (world, "two small all-reduces per ADMM iteration (top rows in, the top's x out), the top's product split by rows across the ranks"
                                                                                                 if info["dist_top"] else "one small all-reduce (top separators) per ADMM iteration, top of the solve replicated"))
                                    if a.shard == "subtree" else "x%d: contiguous element shards, RHS all-reduce per ADMM iteration, replicated solve" % world)),
                   "nnz_L": info["nnz_L"], "supernodes": info["n_supernodes"], "levels": info["n_levels"],
                   "initialize_s": t_init, "factor_numeric_s": info["t_numeric_s"],
                   "factor_numeric_on": "gpu (multifrontal, fp64 MFMA products)" if info.get("device_factor") else "host", "host_threads": info["host_threads"],
                   "x_checksum": float(np.abs(xs).sum())},
        "roofline": roof,
    }
    if world > 1 or fake_dist or fake_world > 1:
        # the exchange, as the library accounts for it (admm_hip_info.comm_doubles_*): ONE collective per ADMM iteration (subtree shards: the
        # packed top rows; contiguous: the whole right-hand side) + under subtree shards one more per frame (the full x before the velocity update)
        out["comm"] = {"path": comm_path if (world > 1 or fake_dist) else "no-op hook (fake world)", "collectives_per_iter": (2 if info["dist_top"] else 1) if info["comm_doubles_iter"] else 0,
                       "bytes_per_iter": 8 * int(info["comm_doubles_iter"]),
                       "bytes_per_collective": ([8 * int(info["comm_doubles_iter"] - 3 * info["nodes_top"]), 8 * 3 * int(info["nodes_top"])] if info["dist_top"] else 8 * int(info["comm_doubles_iter"])),
                       "top": ("distributed: ONE root supernode (%d nodes), every rank streams its rows of the explicit inverse, the slices of x meet in the second collective" % info["nodes_top"]) if info["dist_top"]
                              else "replicated: every rank sweeps the top levels of the tree",
                       "collectives_per_frame_extra": 1 if info["comm_doubles_frame"] else 0, "bytes_per_frame_extra": 8 * int(info["comm_doubles_frame"]),
                       "allreduce_ms_per_iter": phase["allreduce_ms"] / iters_total,
                       "what": "allreduce_ms = pack + ncclAllReduce + unpack (subtree) resp. ncclAllReduce of the RHS (contiguous) between HIP events on the solver's "
                               "stream, every %d-th iteration; it includes the wait for the slowest rank's local step.  The per-frame collective sits in the frame's epilogue." % a.timing_stride}
        out["shard"] = {"mode": a.shard, "tets_local": n_tets_local, "nodes_own": int(info["nodes_own"]), "nodes_top": int(info["nodes_top"]),
                        "sweep_entries_own": int(info["sweep_entries_own"]), "sweep_entries_top": int(info["sweep_entries_top"]), "sweep_entries_top_bwd": int(info["sweep_entries_top_bwd"]),
                        "replicated_top_share_of_fwd_bytes": 8.0 * info["sweep_entries_top"] / fwd_bytes if fwd_bytes > 0 else 0.0,
                        "replicated_top_share_of_all_entries": info["sweep_entries_top"] / float(max(info["nnz_L"], 1))}
        # System::initialize's ONE solver.compute(A) (System.cpp:138-140) across the ranks: rank-local = every rank assembles, factors and keeps its own
        # subtrees + the replicated top (one all-reduce of the subtree roots' update matrices per factorization); otherwise the whole matrix on every rank
        out["factor"] = {"rank_local": bool(info["factor_local"]), "whole_mb": 1e-6 * info["panel_bytes"], "resident_mb": 8e-6 * info["factor_doubles_resident"],
                         "fronts_mb": 8e-6 * info["front_doubles"], "exchange_bytes_once": 8 * int(info["factor_exchange_doubles"]),
                         "numeric_s": info["t_numeric_s"], "on": "gpu" if info.get("device_factor") else "host",
                         "what": "rank 0's figures (per_rank.factor_mb_resident / front_mb / factor_numeric_s / initialize_s: every rank's)"}
        out["rccl_async_error"] = rccl_async
        out["graph_state"] = s.graph_state()
    if fake_world > 1:
        out["fake_world"] = fake_world      # a timing experiment: ONE rank's share of a fake_world-rank partition; sums replaced by the identity
        out["rccl_ranks_seen"] = ranks_seen if fake_dist else None
    if world > 1:      # per-rank phase times (ms per ADMM iteration), element counts and bytes: where the scaling goes
        keys = ["local_ms", "rhs_ms", "allreduce_ms", "solve_fwd_ms", "solve_bwd_ms", "total_ms"]
        extra = ["elements", "tets", "fwd_bytes", "bwd_bytes", "nodes_own"]
        extra_f = ["factor_mb_resident", "front_mb", "factor_numeric_s", "initialize_s"]      # rank-local factorization: what every rank factored, kept, and how long its initialize took
        vals = [phase[k] / iters_total for k in keys] + [float(info["n_elems_local"]), float(n_tets_local), fwd_bytes, bwd_bytes, float(info["nodes_own"])] + \
               [8e-6 * info["factor_doubles_resident"], 8e-6 * info["front_doubles"], float(info["t_numeric_s"]), float(t_init)]
        mine = torch.tensor(vals, dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        out["per_rank"] = {k: [round(float(t[i]), 4) for t in allr] for i, k in enumerate(keys)}
        for j, k in enumerate(extra):
            out["per_rank"][k] = [int(t[len(keys) + j]) for t in allr]
        for j, k in enumerate(extra_f):
            out["per_rank"][k] = [round(float(t[len(keys) + len(extra) + j]), 4) for t in allr]
        out["per_rank_summary"] = {k: {"slowest": max(out["per_rank"][k]), "fastest": min(out["per_rank"][k])} for k in keys}
        out["rccl_ranks_seen"] = ranks_seen              # head count from a checked all-reduce through the path config.allreduce names
        out["ranks_ok"] = bool(ranks_seen == a.gpus == world)
    if rank == 0 and world == 1 and fake_world <= 1 and not a.no_extras:
        # what an existing scene pays per frame through the class API (host/admm/System.hpp step(): m_x / m_v are public, so they
        # travel every frame): the same C-ABI sequence, upload_state -> step -> download_state, on page-locked vectors
        s.enable_timing(False)
        hx = s.m_x.copy(); hv = s.m_v.copy()
        s.pin_host(hx); s.pin_host(hv)
        s.upload_state(hx, hv); s.step(ADMM_ITERS); s.download_state(hx, hv)
        # The cost of a frame moves with the simulation (the line searches lengthen as the bar deforms), so both loops run the SAME frames:
        # checkpoint (x, v, every force's u and warm start), nf frames resident, rewind, the same nf frames through the class API -- twice.
        nf = max(a.steps, 8)
        ck = dict(x=s.m_x.copy(), v=s.m_v.copy(), loc=[s.read_local(b) for b in range(len(s.batches))])

        def rewind():
            s.m_x = ck["x"]; s.m_v = ck["v"]
            for bi, loc in enumerate(ck["loc"]):
                s.write_local(bi, u=loc["u"], state=loc["state"] if pkg.KIND_STATE[s.batches[bi][0]] else None)
            s.sync()
        tc = tr = 0.0
        sums = []
        for _ in range(2):
            rewind()
            t = time.perf_counter()
            for _ in range(nf):
                s.step(ADMM_ITERS)
            s.sync()
            tr += time.perf_counter() - t
            sums.append(float(np.abs(s.m_x).sum()))
            rewind()
            hx[:] = ck["x"]; hv[:] = ck["v"]
            t = time.perf_counter()
            for _ in range(nf):
                s.upload_state(hx, hv); s.step(ADMM_ITERS); s.download_state(hx, hv)
            tc += time.perf_counter() - t
            sums.append(float(np.abs(hx).sum()))
        same = bool(len(set(sums)) == 1)
        nf *= 2
        tc /= nf; tr /= nf
        t = time.perf_counter()          # the four transfers (+ the reordering kernels) on their own: what the boundary costs when nothing hides it
        for _ in range(8):
            s.upload_state(hx, hv); s.download_state(hx, hv)
        tx = (time.perf_counter() - t) / 8
        s.pin_host(hx, False); s.pin_host(hv, False)
        pcie_floor = 4.0 * 8.0 * 3 * info["n_nodes"] / 64e9         # four vectors over PCIe 5 x16 (64 GB/s each way)
        out["class_api"] = {"ms_per_step": 1e3 * tc, "resident_ms_per_step": 1e3 * tr, "value": ADMM_ITERS / tc * n_tets, "overhead_frac": tc / tr - 1.0,
                            "pcie_floor_frac": pcie_floor / tr, "frames": nf,
                            "transfers_alone_ms": 1e3 * tx, "transfers_alone_frac": tx / tr, "transfers_GBps": 4.0 * 8.0 * 3 * info["n_nodes"] / tx / 1e9,
                            "what": "admm_hip_upload_state(m_x, m_v) + admm_hip_step + admm_hip_download_state(m_x, m_v) per frame, event-free, "
                                    "vs. the SAME %d frames (state rewound to one checkpoint before every block) with the state resident" % nf,
                            "same_frames_both_ways": same}
        out["other_configs"] = other_configs(pkg, torch, a.steps)
    if rank == 0:
        if not a.no_cpu_baseline and world == 1:   # the CPU baseline is a rank-0, N=1 side figure
            try:
                out["cpu_baseline"] = cpu_baseline(a.cpu_dims)
                try:      # the like-for-like pair: the GPU on the CPU leg's own sample and frames (ms per ADMM iteration side by side; no ratio to the headline size)
                    cb = out["cpu_baseline"]
                    cb["gpu_same_sample"] = gpu_same_sample(pkg, a.cpu_dims, 2 * max(1, len(cb.get("ms_per_iter_by_team", {}))))
                    cb["same_sample_ms_per_iter"] = {"cpu_%s_%d_threads" % (cb["kind"], cb["cores"]): cb["ms_per_iter"], "gpu": cb["gpu_same_sample"]["ms_per_iter"]}
                except Exception as e:  # noqa: BLE001
                    out["cpu_baseline"]["gpu_same_sample"] = {"error": repr(e)}
            except Exception as e:  # the baseline is a reported side figure; never lose the GPU line over it
                out["cpu_baseline"] = {"value": None, "unit": "ADMM iters/s x elements", "cores": 0, "kind": "unavailable", "sample": repr(e)}
        print(json.dumps(out))
    wd.beat("tear-down", 300.0)
    if world > 1 or fake_dist:     # the library's RCCL communicator goes before the process group does
        torch.cuda.synchronize()
        del s
    if fake_dist and world == 1:
        dist.destroy_process_group()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    wd.stop()


if __name__ == "__main__":
    main()
