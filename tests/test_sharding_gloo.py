"""CPU: the N > 1 path of bench.py — stream sharding and its torch.distributed bookkeeping,
world_size 2 over gloo."""
import json
import os
import subprocess
import sys

import numpy as np

from folve_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shards_partition_the_streams():
    for n, w in ((512, 8), (64, 1), (7, 2), (3, 4)):
        shards = [sharding.shard_streams(n, w, r) for r in range(w)]
        flat = sorted(i for s in shards for i in s)
        assert flat == list(range(n))
        assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
        for r, s in enumerate(shards):
            assert all(sharding.owner_of(i, w) == r for i in s)
    assert len(sharding.shard_streams(512, 8, 3)) == 64         # cfg5: 64 streams per GPU


def test_world_size_2_gloo(tmp_path, oracle):
    out = os.path.join(str(tmp_path), "r.json")
    n_total = 7
    import socket
    with socket.socket() as sk:                 # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py"), out, str(n_total)]
    subprocess.run(cmd, check=True, env=env, timeout=600, cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    r = json.load(open(out))
    assert r["units"] == n_total * 1000 and r["mine"] == [0, 2, 4, 6]
    assert r["tmax"] >= r["dt0"] + 0.009                        # max over ranks, not rank 0's own time
    assert abs(r["rate"] - r["units"] / r["tmax"]) < 1e-6
    # the union of the two shards equals a single-process run
    h = (np.random.default_rng(3).standard_normal(300) / 10).astype(np.float32)
    for s in range(n_total):
        c = oracle.Convproc(1, 1, 300)
        c.impdata_create(0, 0, h, 0)
        sp = oracle.SoundProcessor.wrap(c)
        y = sp.run(np.random.default_rng(100 + s).uniform(-1, 1, (1000, 1)).astype(np.float32))
        assert abs(r["peaks"][s] - sp.max_output_value()) < 1e-7
        assert abs(r["sums"][s] - float(y.astype(np.float64).sum())) < 1e-9
