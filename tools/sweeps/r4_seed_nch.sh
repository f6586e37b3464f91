# round 4: the seed kernel with one / two chunks per wave (library variants), kernel alone on the bench workload (250 Mb, repeats)
export GSA_PROBE_KEEP=/tmp/seedprobe; mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/gsalign_amd/lib
for v in nch1 "" nch1s nch2s; do
  lib=$L/libgsa_hip${v:+_$v}.so; echo "=== ${v:-production (nch2)}"
  st=""; case "$v" in *s) st=1;; esac
  SEED_STATS=$st GSA_LIB_PATH=$lib python tools/seed_probe.py ${N:-250000000} ${VARS:-both,none} 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r4_seed_nch.txt
