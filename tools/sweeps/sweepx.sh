#!/bin/bash
# kernel-level view of the seed stage in sweep mode (GPU box): rocprofv3 kernel trace of tools/seed_probe.py
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export GSA_PROBE_KEEP=/tmp/seedprobe_keep
N=${SEEDX_N:-100000000}; V=${SEEDX_V:-both}
for x in "$@"; do
  rm -f gsalign_amd/csrc/build/k_seed.o; make -C gsalign_amd/csrc -j32 lib EXTRA="$x" > /tmp/mk.log 2>&1 || tail -5 /tmp/mk.log
  echo "=== $x"
  python tools/seed_probe.py $N $V 2>&1 | grep -E "^==|seed stats"
  rm -rf /tmp/sx; rocprofv3 --kernel-trace --stats -d /tmp/sx -o p -- python tools/seed_probe.py $N $V > /tmp/sx.log 2>&1
  python tools/rocprof_summary.py /tmp/sx/p_results.db 12 | cut -c1-160 | grep -E "k_dense|k_seed|kernel"
done
