"""Two ranks, the real aligner, one GPU: what a one-GPU box can prove about the multi-GPU path (the loops being sharded are
GSAlign.cpp:483-548 -- contigs -- and GSAlign.cpp:61-94 -- the chunks of one contig).  tests/two_rank_worker.py does the work; rank 0 compares
what it gathered with the one-rank result byte for byte.  No scaling claim: both ranks share the GPU."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def launch(mode, backend):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    env = dict(os.environ); env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "two_rank_worker.py"), "--mode", mode, "--backend", backend]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    out = r.stdout + r.stderr
    if "TWO_RANK_SKIP" in out:
        pytest.skip([ln for ln in out.splitlines() if "TWO_RANK_SKIP" in ln][0][:400])
    assert r.returncode == 0 and "TWO_RANK_OK" in r.stdout, out[-3000:]
    return [ln for ln in r.stdout.splitlines() if ln.startswith("TWO_RANK_OK")][0]


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_contig_shard_and_gather_equals_one_rank(backend):
    print(launch("shard", backend))


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_chunk_range_split_equals_one_rank(backend):
    print(launch("split", backend))
