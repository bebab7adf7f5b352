#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/r3_final2_tests.log 2>&1
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/r3_final2_smoke.log 2>&1
timeout -k 10 900 python3 bench.py > $O/r3_final2_bench.json 2> $O/r3_final2_bench.err
bash scripts/profile_round.sh > $O/r3_final2_profile.log 2>&1
