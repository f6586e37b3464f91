#!/bin/bash
# Seed-kernel shape sweep on the GPU box (lanes per chunk x speculative sub-ranges x ...): rebuild k_seed.o with each set of -D
# flags and print the un-profiled seed-stage time of tools/seed_probe.py (one context alone, 100 Mb, family + tandem repeats).
#   gpurun -- 'bash tools/seedwg.sh "" "-DSEED_WG=64" "-DSEED_WG=64 -DNSUB=256"'  -> gpurun_out/seedwg.txt
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export GSA_PROBE_KEEP=/tmp/seedprobe_keep
N=${SEEDX_N:-100000000}; V=${SEEDX_V:-both}
mkdir -p gpurun_out; : > gpurun_out/seedwg.txt
for x in "$@"; do
  rm -f gsalign_amd/csrc/build/k_seed.o; make -C gsalign_amd/csrc -j32 lib EXTRA="$x" > /tmp/mk.log 2>&1 || tail -5 /tmp/mk.log
  echo "=== $x" >> gpurun_out/seedwg.txt
  python tools/seed_probe.py $N $V 2>&1 | grep -E "^==|seed stats|SEED_STATS|Error|error" >> gpurun_out/seedwg.txt
done
cat gpurun_out/seedwg.txt
