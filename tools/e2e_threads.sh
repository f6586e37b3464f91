#!/bin/bash
# GPU box: the product end to end (GSAlign_hip at BASELINE configs[4]: index + FASTA from disk -> MAF + VCF) against the number of host pool threads (GSA_HOST_THREADS; the
# default is min(hardware threads, 32)).  The workload's files are made once by a short bench.py run (GSA_BENCH_KEEP), then the CLI runs three times per setting.
#   gpurun -- 'bash tools/e2e_threads.sh'   ->   one line per run: threads, total_s and the host-side terms of the -timing record
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export GSA_BENCH_KEEP=1 GSA_BENCH_TMP=/tmp/gb; mkdir -p $GSA_BENCH_TMP
timeout 900 python bench.py --workload human_full --steps 2 --warmup 1 --extra "" --no-cpu-baseline --no-side-legs 2>gpurun_out/e2e_threads_bench.err | tail -1 | cut -c1-200
IDX=$(ls -d /tmp/gb/human_full_* | head -1); IDX=${IDX%%.*}; Q=/tmp/gb/e2e_human_full_q.fa
ls -la /tmp/gb | head -20
for T in ${THREADS:-32 64 96 32 64 96 128}; do
  rm -f /tmp/gb/e2e_thr_out.*; sync      # (the previous run's 7.5 GB of output must not be written back beside this one)
  GSA_HOST_THREADS=$T timeout 300 gsalign_amd/bin/GSAlign_hip -i $IDX -q $Q -o /tmp/gb/e2e_thr_out -ctx 8 -timing -alen 5000 > /tmp/gb/e2e_thr.out 2> /tmp/gb/e2e_thr.err
  python - $T <<'PY'
import json, sys
t = open("/tmp/gb/e2e_thr.err").read()
try:
    d = json.loads([ln for ln in t.splitlines() if ln.startswith("GSA_TIMING ")][-1][len("GSA_TIMING "):])
    print("threads", sys.argv[1], " ".join(f"{k} {d[k]}" for k in ("total_s", "index_load_s", "gsa_create_s", "query_load_s", "align_many_s", "maf_format_s", "variants_s", "output_drain_after_align_s", "maf_write_s", "vcf_s", "host_threads") if k in d))
except Exception as e:
    print("threads", sys.argv[1], "FAILED", e, t[-400:])
PY
done
