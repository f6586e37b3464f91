cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for mode in "" torch; do for env in "X=1"; do
  echo "== mode=${mode:-plain} $env"
  env $env rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b -o p -- python tools/blit_probe.py $mode 2>&1 | grep "fill" | sort | uniq -c | sort -rn | head -4
  python tools/rocprof_summary.py gpurun_out/prof_b/p_results.db 5 | tail -4 | cut -c1-30,70-112
  rm -rf gpurun_out/prof_b
done; done
