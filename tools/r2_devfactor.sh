#!/bin/bash
# GPU box: device factorization -- time, flop rate, kernel breakdown (rocprofv3 stats)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
ADMM_HIP_VERBOSE=1 python tools/init_breakdown.py 2>&1 | grep -E "numeric factorization|initialize|recompute"
mkdir -p gpurun_out/devfactor
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/devfactor -- python3 $GRAFT_REPO_ROOT/tools/init_breakdown.py > /dev/null 2>&1)
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/devfactor/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
import shutil, os
os.makedirs('gpurun_out/devfactor', exist_ok=True); shutil.copy(f[0], 'gpurun_out/devfactor/init_kernel_stats.csv')
for r in rows[:12]:
    print("%-60s calls %6s total %10.3f ms avg %9.1f us" % (r['Name'][:60], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY
