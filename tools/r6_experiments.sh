#!/bin/bash
# Round 6's A/B experiments on the GPU box, as cited in profiles/r06_seed_wpw.txt and profiles/r06_experiments.txt (every run: bench.py --workload W --extra "" --no-cpu-baseline
# --no-side-legs --no-e2e, four contexts; one line per run: Gbp/s, ms per step, stage times of one context alone).
#   gpurun -- 'bash tools/r6_experiments.sh <mode>'     (library variants are built HERE first: cd gsalign_amd/csrc && make lib VARIANT=<v> EXTRA=<-D...>)
#   wpw        SEED_WPW = 1 / 8 / 10 (variants wpw1 wpw8 wpw10): one fat workgroup of independent waves per CU in the seed kernel
#   passes     LB_TPB = 128 / 64 (variants lbt128 lbt64): smaller workgroups in the fused passes
#   hwq        8 / 16 / 24 / 32 hardware queues
#   budget     the speculative seed kernel's give-up budget 256 .. 48 on the repeat workloads; SEED_NCH=2 (variant nch2) on the human index
#   dplane     per-lane DP class boundary 512 .. 8192 cells
#   dpsmall    k_dp_small on a stream of its own (option dp_small_side), with and without dp_side
#   footprint  the seed kernel's footprint on a CU: waves per CU x register bound (variants built with -DSEED_WPW=<n> -DSEED_MIN_WAVES=<b> [-DLHOP_N=512]; "r5" = -DSEED_WPW=1 -DSEED_MIN_WAVES=3)
#   contexts   3 .. 8 contexts in flight behind the adopted footprint
#   soak10     ten contexts six times over (30 steps each): does the look-back wait bound (5 s of wall clock) ever trip?
#   late       the fused passes at four elements per thread (variant lbi4: -DLB_ITEMS=4, 67 - 122 VGPRs instead of 111 - 213) and ten contexts, on the final tree
ulimit -c 0      # (a faulting experiment must not fill the box's disk with a core dump: the third call of the round did)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
export GSA_BENCH_KEEP=1 GSA_BENCH_TMP=/tmp/gb; mkdir -p $GSA_BENCH_TMP
run() {   # tag, workload, library variant ("-" = the product), env assignments, extra bench args
  L=$PWD/gsalign_amd/lib/libgsa_hip_$3.so; [ "$3" = "-" ] && L=$PWD/gsalign_amd/lib/libgsa_hip.so
  [ -f "$L" ] || { echo "run $1: $L is not built"; return; }
  env $4 GSA_LIB_PATH=$L GSA_BENCH_DETAIL=gpurun_out/r6x_detail_$1.json timeout 900 python bench.py --workload $2 --extra "" --no-cpu-baseline --no-side-legs --no-e2e $5 2>gpurun_out/r6x_$1.err | tail -1 > gpurun_out/r6x_$1.json
  python - <<P
import json
try:
    d=json.load(open("gpurun_out/r6x_$1.json")); print("run $1", d["value"], "Gbp/s", d["ms_per_step"], "ms", d.get("stage_ms_alone"))
except Exception as e:
    print("run $1 FAILED", e); print(open("gpurun_out/r6x_$1.err").read()[-800:])
P
}
F="--steps 10 --warmup 2"
case "${1:-}" in
  wpw)     for v in wpw1 - wpw8 wpw10; do run full_$v human_full $v GSA_X=0 "$F"; done ;;
  passes)  for v in - lbt128 lbt64; do run full_$v human_full $v GSA_X=0 "$F"; done ;;
  hwq)     for q in 8 16 24 32; do run full_hwq$q human_full - GSA_X=0 "$F --hwq $q"; done ;;
  budget)  for b in 256 128 96 64 48; do run hl_b$b human_like - GSA_SEED_BUDGET=$b ""; done
           for b in 256 96 64; do run adv_b$b adversarial - GSA_SEED_BUDGET=$b ""; done
           for b in 256 96; do run hum_b$b human - GSA_SEED_BUDGET=$b ""; done
           run full_b128 human_full - GSA_SEED_BUDGET=128 "$F"; run full_nch2 human_full nch2 GSA_X=0 "$F" ;;
  dplane)  for v in 512 1024 2048 4096 8192; do run full_lane$v human_full - GSA_DP_LANE=$v "$F"; done
           for v in 512 2048 8192; do run hum_lane$v human - GSA_DP_LANE=$v ""; done ;;
  dpsmall) run full_base human_full - GSA_X=0 "$F"; run full_ss human_full - GSA_DP_SMALL_SIDE=1 "$F"; run full_ss_ds human_full - "GSA_DP_SMALL_SIDE=1 GSA_DP_SIDE=1" "$F"
           run hum_base human - GSA_X=0 ""; run hum_ss human - GSA_DP_SMALL_SIDE=1 ""; run hum_ss_ds human - "GSA_DP_SMALL_SIDE=1 GSA_DP_SIDE=1" "" ;;
  footprint) for v in r5 - w8b4 w10b5 w9b5 w7b5 w6b5 w8b5lh; do run full_$v human_full $v GSA_X=0 "$F"; done; for v in r5 -; do run hum_$v human $v GSA_X=0 ""; done ;;
  contexts) for n in 3 4 5 6 7 8; do run full_ctx$n human_full - GSA_X=0 "--steps 20 --warmup 4 --inflight $n"; done
           for n in 4 6 8; do run hum_ctx$n human - GSA_X=0 "--inflight $n"; done; run hl_ctx6 human_like - GSA_X=0 "--inflight 6"; run yeast_ctx6 yeast - GSA_X=0 "--inflight 6" ;;
  late)    run late_base human_full - GSA_X=0 "$F"; run late_lbi4 human_full lbi4 GSA_X=0 "$F"; run late_ctx10 human_full - GSA_X=0 "$F --inflight 10"; run late_lbi4_hum human lbi4 GSA_X=0 "" ;;
  soak10)  for i in 1 2 3 4 5 6; do run soak10_$i human_full - GSA_X=0 "--steps 30 --warmup 4 --inflight 10"; grep -il "timed out\|error" gpurun_out/r6x_soak10_$i.err; done ;;
  *) echo "usage: r6_experiments.sh wpw|passes|hwq|budget|dplane|dpsmall|footprint|contexts|late|soak10" ;;
esac
