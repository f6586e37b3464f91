"""The N>1 path on CPU: world_size-2 gloo run of the contig sharding + record gather."""
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT
from gsalign_amd import shard


def test_lpt_assignment_is_balanced_and_deterministic():
    lens = [248, 242, 198, 190, 181, 170, 159, 145, 138, 133, 135, 133, 114, 107, 101, 90, 83, 80, 58, 64, 46, 50, 156, 57]   # GRCh38 Mb
    a = shard.assign_contigs(lens, 8)
    assert sorted(i for r in a for i in r) == list(range(len(lens)))
    loads = [sum(lens[i] for i in r) for r in a]
    assert max(loads) <= 1.15 * (sum(lens) / 8)
    assert a == shard.assign_contigs(lens, 8)
    assert shard.assign_contigs([5, 3], 1) == [[0, 1]]


WORKER = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from gsalign_amd import shard
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
lens = [50, 40, 30, 20, 10]
mine = shard.assign_contigs(lens, world)[rank]
# fake finished blocks: contig c yields c+1 records whose bytes encode (contig, k)
recs, ids = [], []
for c in mine:
    for k in range(c + 1):
        r = np.zeros(40, np.uint8); r[0] = c; r[1] = k; recs.append(r); ids.append(c)
recs = np.array(recs, np.uint8).reshape(-1, 40); ids = np.array(ids, np.int32)
allr, alli = shard.gather_block_records(recs, ids)
assert alli.tolist() == sorted(alli.tolist()) and len(alli) == sum(c + 1 for c in range(5)), alli
for c in range(5):
    sel = allr[alli == c]
    assert sel[:, 0].tolist() == [c] * (c + 1) and sel[:, 1].tolist() == list(range(c + 1))
dist.barrier(); dist.destroy_process_group()
print("ok", rank)
"""


def test_gather_world2_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2
