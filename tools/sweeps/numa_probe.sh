lscpu | grep -i "numa\|socket\|model name" | head -8
for d in /sys/class/drm/card*/device; do echo $d $(cat $d/numa_node 2>/dev/null) $(cat $d/vendor 2>/dev/null); done | head
N0=$(lscpu | grep "NUMA node0" | awk '{print $NF}'); N1=$(lscpu | grep "NUMA node1" | awk '{print $NF}')
echo "N0=$N0 N1=$N1"
for r in 1 2 3; do for n in "$N0" "$N1"; do [ -z "$n" ] && continue; taskset -c $n python bench.py --workload ecoli --steps 60 --warmup 10 --extra "" --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"cpus $n\", round(d[\"value\"],3), round(d[\"ms_per_step\"],3))"; done; done
