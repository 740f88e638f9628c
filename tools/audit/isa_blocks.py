#!/usr/bin/env python3
"""Static instruction audit of one kernel's ISA (hipcc -S output).

usage: isa_blocks.py file.s kernel-symbol-substring [--dump]
Splits the kernel into basic blocks (labels), finds the natural loops from backward branches,
and prints per block / per loop: instruction count by class (fp64 arithmetic, transcendental,
moves, selects, compares, lane moves, integer, scalar, branch, nop, memory).
"""
import re, sys, collections

CLASSES = [
    ("f64_arith", re.compile(r"^v_(add|mul|fma|fmac|max|min|ldexp|div_scale|div_fmas|div_fixup|frexp_mant|fract|trunc|rndne|floor|ceil)_f64")),
    ("f64_trans", re.compile(r"^v_(rcp|rsq|sqrt)_f64")),
    ("mov", re.compile(r"^v_(mov_b32|mov_b64|accvgpr|pk_mov)")),
    ("select", re.compile(r"^v_cndmask")),
    ("cmp", re.compile(r"^v_cmp")),
    ("lane", re.compile(r"^v_(readlane|writelane|readfirstlane|permlane|bpermute)|^ds_(bpermute|permute|swizzle)")),
    ("cvt", re.compile(r"^v_cvt")),
    ("v_int", re.compile(r"^v_")),
    ("s_nop", re.compile(r"^s_nop")),
    ("s_wait", re.compile(r"^s_waitcnt")),
    ("branch", re.compile(r"^s_(cbranch|branch|setpc|call)")),
    ("salu", re.compile(r"^s_")),
    ("scratch", re.compile(r"^scratch_")),
    ("vmem", re.compile(r"^(global|flat|buffer)_")),
    ("lds", re.compile(r"^ds_")),
]


def classify(op):
    for name, rx in CLASSES:
        if rx.match(op):
            return name
    return "other"


def main():
    path, sym = sys.argv[1], sys.argv[2]
    dump = "--dump" in sys.argv
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if l.startswith("_Z") and sym in l.split(":")[0] and l.rstrip().split(";")[0].strip().endswith(":"):
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    end = start
    while not lines[end].strip().startswith("s_endpgm"):
        end += 1
    # keep going to the .Lfunc_end
    while not lines[end].startswith(".Lfunc_end"):
        end += 1
    body = lines[start:end]
    blocks = []  # (label, [ops], [branch targets])
    cur = ["entry", [], []]
    for l in body[1:]:
        s = l.split(";")[0].strip()
        if not s:
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            blocks.append(cur)
            cur = [m.group(1), [], []]
            continue
        if s.startswith("."):
            continue
        op = s.split()[0]
        cur[1].append(s)
        if op.startswith("s_cbranch") or op == "s_branch":
            cur[2].append(s.split()[-1])
    blocks.append(cur)
    index = {b[0]: i for i, b in enumerate(blocks)}
    tot = collections.Counter()
    for b in blocks:
        for s in b[1]:
            tot[classify(s.split()[0])] += 1
    print("kernel %s: %d blocks, %d instructions" % (sym, len(blocks), sum(tot.values())))
    print("  static mix:", dict(tot.most_common()))
    # loops: backward branch from block j to block i <= j  => blocks i..j (layout order) form the loop (reducible, laid out contiguously by LLVM)
    loops = []
    for j, b in enumerate(blocks):
        for t in b[2]:
            if t in index and index[t] <= j:
                loops.append((index[t], j))
    loops = sorted(set(loops))
    for (i, j) in loops:
        c = collections.Counter()
        for b in blocks[i:j + 1]:
            for s in b[1]:
                c[classify(s.split()[0])] += 1
        n = sum(c.values())
        valu = sum(v for k, v in c.items() if k in ("f64_arith", "f64_trans", "mov", "select", "cmp", "lane", "cvt", "v_int"))
        print("loop %s .. %s (%d blocks): %d instructions, %d VALU: %s" % (blocks[i][0], blocks[j][0], j - i + 1, n, valu, dict(c.most_common())))
    if dump:
        for b in blocks:
            print("== %s (%d)" % (b[0], len(b[1])))
            for s in b[1]:
                print("    " + s)


if __name__ == "__main__":
    main()
