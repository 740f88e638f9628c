#!/bin/bash
# round 4, GPU session G: the one-case step selection -- full GPU suite, then every judged artefact of the round on these sources
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4g; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 2400 bash tools/profile_round.sh r04 > $O/profile_round.log 2>&1; tail -5 $O/profile_round.log
PMC_MIXED=1 timeout 900 bash tools/pmc_collect.sh > $O/pmc_mixed.log 2>&1; tail -2 $O/pmc_mixed.log
timeout 900 bash tools/pmc_instmix.sh > $O/tet_instmix.txt 2>&1; tail -24 $O/tet_instmix.txt
timeout 600 python tools/tet_phase_profile.py frames=3 > $O/tet_phase_profile.txt 2>&1; tail -4 $O/tet_phase_profile.txt
ls gpurun_out gpurun_out/prof
