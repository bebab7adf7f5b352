#!/bin/bash
# Round-end check on the GPU box (through gpurun): tests, smoke, the profile passes, then the default bench run -- last,
# so that its line carries the HBM traffic record the PMC passes have just produced for the current kernel source.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/verify_tests.log 2>&1
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/verify_smoke.log 2>&1
bash scripts/profile_round.sh > $O/verify_profile.log 2>&1
cp $O/r4p_spread_traffic.json profiles/r04_spread_traffic.json
timeout -k 10 900 python3 bench.py > $O/verify_bench.json 2> $O/verify_bench.err
