"""bench.py --gpus N must really become N ranks (round-1 finding: the flag was parsed and ignored).  CPU check with the
--dry stub: `python bench.py --gpus 2 --dry` re-executes itself under torch.distributed.run, rank 0 prints one JSON line
with n_gpus = 2, and the staged result gather (shard.ResultStage / gather_staged: ONE genome's contigs dealt to the ranks by LPT,
every finished contig to rank 0, checked byte for byte inside the run) has left all contigs of the genome on rank 0."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def run(*args):
    env = dict(os.environ); env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_gpus_2_relaunches_as_two_ranks():
    d = run("--gpus", "2", "--dry", "--steps", "3", "--warmup", "1")
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["config"]["contigs_on_rank0_after_gather"] == 5 and d["config"]["world_size"] == 2
    assert d["scaling"] == "strong" and d["higher_is_better"] is True and d["value"] > 0


def test_gpus_1_runs_in_process():
    d = run("--dry", "--steps", "2", "--warmup", "0")
    assert d["n_gpus"] == 1


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr


CONTRACT_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"}


def test_dry_line_is_compact_and_complete():
    """Round 5's driver record had `parsed: null`: the line was 32 KB and the driver keeps less of stdout than that.  The LAST stdout
    line must be one small strict-JSON object with the contract's keys."""
    env = dict(os.environ); env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry", "--steps", "2", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    last = r.stdout.rstrip("\n").splitlines()[-1]
    assert len(last) < 4096
    d = json.loads(last, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))      # NaN / Infinity refused
    assert CONTRACT_KEYS <= set(d) and isinstance(d["config"]["workload"], str) and len(d["config"]["workload"]) <= 120


def test_compact_line_of_a_full_measurement():
    """The whole object of a real run (round 5's 32 KB line, kept under profiles/) through bench.compact_line: < 4 KB, strict JSON, the
    contract's keys + roofline + cpu_baseline + end_to_end, one entry per further workload."""
    sys.path.insert(0, ROOT)
    import bench
    full = json.loads(open(os.path.join(ROOT, "profiles", "r05_bench_default.json")).read().strip().splitlines()[-1])
    full["roofline"]["physical_frac"] = float("nan"); full["kernels"][0]["achieved"] = float("inf")           # what a zero-duration timer would produce
    line = bench.compact_line(bench._clean(full))
    assert len(line) < 4096 and "\n" not in line
    d = json.loads(line, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))
    assert CONTRACT_KEYS <= set(d)
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "physical_frac", "algorithmic_bytes_per_step"} <= set(d["roofline"])
    assert {"value", "unit", "cores", "kind", "sample", "parity_sample"} <= set(d["cpu_baseline"])
    assert {"total_s", "align_many_s"} <= set(d["end_to_end"])
    assert d["roofline"]["physical_frac"] is None and abs(d["roofline"]["frac"] - full["roofline"]["frac"]) < 1e-3
    assert [e["workload"] for e in d["extra_workloads"]] == [e["workload"] for e in full["extra_workloads"]]
    assert len(d["config"]["workload"]) <= 120 and abs(d["value"] - full["value"]) < 1e-3
    json.dumps(bench._clean(full), allow_nan=False)                  # the detail file is strict JSON as well
