set -x
mkdir -p gpurun_out
python -m pytest tests/test_gpu_bundle.py -x -q 2>&1 | tail -5
python tools/copy_probe.py > gpurun_out/copy_probe.txt 2>&1; python tools/copy_probe.py torch >> gpurun_out/copy_probe.txt 2>&1
cat gpurun_out/copy_probe.txt
python bench.py --workload human --extra "" --no-cpu-baseline --steps 40 --warmup 6 > gpurun_out/b_human.json 2> gpurun_out/b_human.err; tail -3 gpurun_out/b_human.err
python tools/show_bench.py gpurun_out/b_human.json 2>/dev/null | head -30 || true
