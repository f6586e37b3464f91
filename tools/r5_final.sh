#!/bin/bash
# round 5, final GPU call: the whole suite, then the round's profiles (tools/round_profiles.sh) on the final commit
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 ) > gpurun_out/r5_gputest_final.log 2>&1; tail -14 gpurun_out/r5_gputest_final.log
bash tools/round_profiles.sh
