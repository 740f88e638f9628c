#!/bin/bash
# round 4, GPU session C: parity after the range-known sqrt / reciprocal + auto tets-per-block, bench line, TPB rule across sizes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4c; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.err; python3 tools/bench_summary.py r4c < $O/bench.json
for cfg in "10x10x9 TET_NH 8" "10x10x30 TET_NH 10" "13x13x50 TET_STVK 8" "13x13x50 TET_STVK 14" "13x13x50 TET_NH 10" "20x20x60 TET_NH 8" "16x16x81 TET_NH 6"; do
  set -- $cfg
  timeout 600 python tools/probe/underfilled.py tpb dims=$1 kind=$2 frames=$3 >> $O/tpb_sizes.txt 2>&1
done
cat $O/tpb_sizes.txt
