#!/usr/bin/env python3
"""Instruction audit of project_tet_kernel<NH, 5>: static instruction counts per code region (from the ISA of the shipped build,
hipcc -S) x how many times a WAVE executes each region (wave-level counters of the -DADMM_TET_PROFILE build, tools/tet_phase_profile.py:
the "REGIONS" line) = dynamic VALU instructions per wave by region, to be compared with the PMC total (SQ_INSTS_VALU / waves).

  python3 tools/audit/tet_inst_by_region.py "<REGIONS line of tet_phase_profile.py>" [SQ_INSTS_VALU per wave from pmc_1M.json]

Regions are found from the loop structure of the ISA (natural loops = backward branches):
  * the first large loop = the Jacobi sweep (three pair rotations, each behind its threshold test);
  * the second = the L-BFGS outer iteration; its largest inner loop = the More'-Thuente evaluation loop, split into "objective + gradient"
    (through the block holding the gradient's three reciprocals) and "step selection" (the rest); its one-block inner loops = the
    two-loop recursion / history shift;
  * everything outside the loops = load, F, SVD set-up and tail, orientation, first gradient, recomposition, epilogue (straight-line).
The log's two branches (table / near 1) are the blocks of the evaluation part that load the table resp. hold the 2^27 split constants.
"""
import collections, os, re, subprocess, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import isa_blocks as ib

VALU = ("f64_arith", "f64_trans", "mov", "select", "cmp", "lane", "cvt", "v_int")

def parse(path, sym):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sym in l.split(":")[0] and l.rstrip().split(";")[0].strip().endswith(":"))
    end = start
    while not lines[end].startswith(".Lfunc_end"): end += 1
    blocks, cur = [], ["entry", [], []]
    for l in lines[start + 1:end]:
        s = l.split(";")[0].strip()
        if not s: continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m: blocks.append(cur); cur = [m.group(1), [], []]; continue
        if s.startswith("."): continue
        cur[1].append(s)
        op = s.split()[0]
        if op.startswith("s_cbranch") or op == "s_branch": cur[2].append(s.split()[-1])
    blocks.append(cur)
    return blocks

def mix(blks):
    c = collections.Counter()
    for b in blks:
        for s in b[1]: c[ib.classify(s.split()[0])] += 1
    return c

def valu(c): return sum(v for k, v in c.items() if k in VALU)

def main():
    regions = {}
    if len(sys.argv) > 1:
        for tok in sys.argv[1].replace("REGIONS", "").split():
            k, v = tok.split(":"); regions[int(k)] = float(v)
    pmc = float(sys.argv[2]) if len(sys.argv) > 2 else None
    path = "/tmp/isa/admm_hip.s"
    if not os.path.exists(path):
        subprocess.check_call([sys.executable, os.path.join(HERE, "kernel_resources.py"), "project_tet_kernel<0, 5, false"], stdout=subprocess.DEVNULL)
    blocks = parse(path, "project_tet_kernelILi0ELi5ELb0")
    index = {b[0]: i for i, b in enumerate(blocks)}
    loops = sorted({(index[t], j) for j, b in enumerate(blocks) for t in b[2] if t in index and index[t] <= j})
    top = [l for l in loops if not any(o != l and o[0] <= l[0] and l[1] <= o[1] for o in loops)]
    big = [l for l in top if valu(mix(blocks[l[0]:l[1] + 1])) > 300]
    sweep, outer = big[0], big[1]
    inner = [l for l in loops if l != outer and outer[0] <= l[0] and l[1] <= outer[1]]
    has_tab = lambda l: any(s_.startswith("global_load_dwordx4") for b in blocks[l[0]:l[1] + 1] for s_ in b[1])
    l0 = min([l for l in inner if has_tab(l)], key=lambda l: l[1] - l[0])          # the smallest loop around a log-table load ...
    same = [l for l in inner if l0[0] - 1 <= l[0] <= l0[0]]                        # ... and every loop sharing its header: the evaluation loop
    ls = (min(l[0] for l in same), max(l[1] for l in same))
    hist = [l for l in inner if l[0] == l[1] and not (ls[0] <= l[0] <= ls[1])]
    lsb = blocks[ls[0]:ls[1] + 1]
    split = next(i for i, b in enumerate(lsb) if sum(1 for s in b[1] if s.startswith("v_rcp_f64")) >= 3)
    evalb, cstepb = lsb[:split + 1], lsb[split + 1:]
    log_tab = [b for b in evalb if any(s.startswith("global_load_dwordx4") for s in b[1])]
    log_near = [b for b in evalb if any("0x41a00000" in s for s in b[1])]
    eval_rest = [b for b in evalb if b not in log_tab and b not in log_near]
    outer_rest = [b for i, b in enumerate(blocks[outer[0]:outer[1] + 1], outer[0]) if not (ls[0] <= i <= ls[1]) and not any(h[0] == i for h in hist)]
    straight = [b for i, b in enumerate(blocks) if not (sweep[0] <= i <= sweep[1]) and not (outer[0] <= i <= outer[1])]
    sw = mix(blocks[sweep[0]:sweep[1] + 1])
    g = lambda k: regions.get(k, float("nan"))
    waves = 1.0
    rows = [
        ("straight-line: load, F, SVD set-up / sort / orientation, first gradient, recomposition, u / RHS epilogue", mix(straight), 1.0, "once per wave"),
        ("Jacobi sweep loop body (3 threshold tests + 3 rotations behind them)", sw, None, None),
        ("L-BFGS outer iteration without its inner loops (direction, restart test, updates)", mix(outer_rest), g(6), "outer iterations per wave"),
        ("two-loop recursion / history shift (scratch memory), per history pair", mix([blocks[h[0]] for h in hist]), g(7) / max(len(hist), 1) if hist else 0.0, "history pairs per wave (per loop)"),
        ("line search: trial point, objective + gradient without the logs", mix(eval_rest), g(8), "evaluations per wave"),
        ("  log, table branch (2 call sites)", mix(log_tab), g(10) / 2.0, "executions per wave / 2 call sites"),
        ("  log, near-1 branch (2 call sites)", mix(log_near), g(11) / 2.0, "executions per wave / 2 call sites"),
        ("line search: tests + step selection (mt_cstep) + interval update", mix(cstepb), g(9), "step selections per wave"),
    ]
    # the sweep body: rotations x (body - tests) / 3 + sweeps x tests
    test_cost = 3 * 12
    rot_static = (valu(sw) - test_cost) / 3.0
    print("project_tet_kernel<NH, 5, false>: %d blocks, %d instructions, %d VALU (static)" % (len(blocks), sum(sum(mix([b]).values()) for b in blocks), valu(mix(blocks))))
    print("%-108s %7s %7s %7s %7s %7s | %8s %9s" % ("region", "VALU", "fp64", "trans", "sel+cmp", "mov+int", "x/wave", "dyn VALU"))
    total = 0.0
    for name, c, times, what in rows:
        v = valu(c)
        if times is None:
            dyn = g(2) * rot_static + g(1) * test_cost
            tm = "%.2f rot" % g(2)
        else:
            dyn = v * times; tm = "%.2f" % times
        total += dyn if dyn == dyn else 0.0
        print("%-108s %7d %7d %7d %7d %7d | %8s %9.0f" % (name[:108], v, c["f64_arith"], c["f64_trans"], c["select"] + c["cmp"], c["mov"] + c["v_int"] + c["lane"] + c["cvt"], tm, dyn))
    print("   (sweep body priced as rotations x %.0f + sweeps x %d for the threshold tests: %.2f sweeps, %.2f rotations per wave)" % (rot_static, test_cost, g(1), g(2)))
    print("sum of the regions: %.0f VALU instructions per wave%s" % (total, "" if pmc is None else "; PMC SQ_INSTS_VALU per wave: %.0f (model / counter = %.2f)" % (pmc, total / pmc)))
    print("blocks: sweep %s..%s | outer %s..%s | line search %s..%s (evaluation through %s) | history loops %s" % (
        blocks[sweep[0]][0], blocks[sweep[1]][0], blocks[outer[0]][0], blocks[outer[1]][0], blocks[ls[0]][0], blocks[ls[1]][0], lsb[split][0], [blocks[h[0]][0] for h in hist]))

if __name__ == "__main__":
    main()
