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


def test_split_chunks():
    assert shard.split_chunks(10, 3) == [(0, 4), (4, 7), (7, 10)]
    assert shard.split_chunks(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]
    r = shard.split_chunks(25000, 2); assert r[0][1] == r[1][0] and r[1][1] == 25000


WORKER2 = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from gsalign_amd import shard, capi
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)

class StubAligner:                      # stands in for capi.Aligner: same call shapes, deterministic fake contents
    def seed_chunks(self, contig, beg, end):
        self.k = np.arange(beg * 7, end * 7, dtype=np.uint64) * np.uint64(1000003) + np.uint64(int(contig[:8].sum()))
        self.v = (self.k %% np.uint64(97)).astype(np.uint32); self.imported = []
        return self.k.size
    def export_hits(self): return self.k, self.v
    def import_hits(self, k, v): self.imported.append((np.array(k), np.array(v)))
    def align(self, ci, contig):
        n = int(contig.size %% 7) + 1
        B = np.zeros(n, capi.BLOCK_DT); B["score"] = np.arange(n) + ci * 100; B["n_frag"] = 2
        F = np.zeros(2 * n, capi.FRAG_DT); F["qpos"] = np.arange(2 * n) + ci
        s = np.frombuffer((b"ACGT-" * (ci + 3))[: 11 + ci], np.uint8)
        return dict(blocks=B, frags=F, aln1=s.copy(), aln2=s[::-1].copy())

rng = np.random.default_rng(5)
contigs = [rng.integers(65, 70, size=n).astype(np.uint8) for n in (50000, 41000, 30011, 20000, 10000, 777)]
g = StubAligner()
# (1) contig sharding + full-result gather: rank 0 ends up with what one rank alone computes
mine = {ci: g.align(ci, contigs[ci]) for ci in shard.assign_contigs([c.size for c in contigs], world)[rank]}
allr = shard.gather_results(mine, capi.BLOCK_DT, capi.FRAG_DT)
if rank == 0:
    assert sorted(allr) == list(range(len(contigs)))
    for ci, c in enumerate(contigs):
        w = g.align(ci, c)
        for k in w: assert np.array_equal(allr[ci][k], w[k]), (ci, k)
# (1b) the bench's gather (round 4): finished contigs staged as one tensor each (shard.ResultStage), point-to-point to rank 0 in exact sizes
stage = shard.ResultStage(torch.device("cpu"))
assign = shard.assign_contigs([c.size for c in contigs], world)
for ci in assign[rank]:
    w = g.align(ci, contigs[ci]); R = np.zeros(3 + ci, capi.REC_DT); R["w0"] = np.arange(3 + ci) * 7 + ci; R["w3"] = 11 * ci
    B, a1, a2 = np.ascontiguousarray(w["blocks"]), w["aln1"], w["aln2"]
    if rank != 0:
        stage.put(7, ci, [(B.ctypes.data, B.nbytes), (R.ctypes.data, R.nbytes), (a1.ctypes.data, a1.nbytes), (a2.ctypes.data, a2.nbytes)])
got, _ = shard.gather_staged(stage.take(7), max(len(x) for x in assign), device=torch.device("cpu"))
if rank == 0:
    ids = [shard.parse_staged(b, capi.BLOCK_DT, capi.REC_DT)[0] for b in got]
    assert ids == sorted(ids), ids                      # (the documented contract: the other ranks' contigs in contig order)
    seen = dict(shard.parse_staged(b, capi.BLOCK_DT, capi.REC_DT) for b in got)
    assert sorted(seen) == sorted(ci for r in range(1, world) for ci in assign[r])
    for ci, r in seen.items():
        w = g.align(ci, contigs[ci])
        R = np.zeros(3 + ci, capi.REC_DT); R["w0"] = np.arange(3 + ci) * 7 + ci; R["w3"] = 11 * ci
        assert np.array_equal(r["blocks"], w["blocks"]) and np.array_equal(r["recs"], R) and np.array_equal(r["aln1"], w["aln1"]) and np.array_equal(r["aln2"], w["aln2"]), ci
else:
    assert got == []
# (2) one contig seeded by chunk range on every rank, hits sent to the owner
n_chunks = 23
rngs = shard.split_chunks(n_chunks, world)
g.seed_chunks(contigs[0], *rngs[rank])
total = shard.exchange_hits(g, owner=0)
if rank == 0:
    full = StubAligner(); full.seed_chunks(contigs[0], 0, n_chunks)
    k = np.concatenate([g.k] + [x[0] for x in g.imported]); v = np.concatenate([g.v] + [x[1] for x in g.imported])
    o = np.argsort(k); assert np.array_equal(k[o], np.sort(full.k)) and np.array_equal(v[o], full.v[np.argsort(full.k)]) and total == k.size - g.k.size
dist.barrier(); dist.destroy_process_group()
print("ok", rank)
"""


def test_results_gather_and_hit_exchange_world2_gloo(tmp_path):
    """The two exchanges of the multi-GPU path with a stub aligner, world size 2 on CPU: (1) contigs sharded by LPT, complete
    results (blocks + gap records + gapped strings) gathered on rank 0 == the one-rank results; (2) one contig seeded by chunk
    range, hits exchanged to the owner == the hits of the whole contig."""
    script = tmp_path / "w2.py"
    script.write_text(WORKER2 % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29534")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29534", str(script)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2


def test_gather_world2_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("ok") == 2
