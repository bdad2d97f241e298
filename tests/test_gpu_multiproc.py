"""Two ranks (gloo rendezvous, both on the one GPU of the test box) encode disjoint GOP shards
through the real HIP path; the payloads must equal a single-process encode of the same GOPs and
bench.py must run under torch.distributed.run with world_size 2."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests.util import ROOT

pytestmark = pytest.mark.gpu

WORKER = r"""
import os, sys, pickle
sys.path.insert(0, os.environ["DCVC_ROOT"])
import torch, torch.distributed as dist
from vcm_ts_amd.dmc import DMC
from vcm_ts_amd.intra import IntraNoAR
from vcm_ts_amd.pipeline import GopEncoder, shard_gops, timed_region
from vcm_ts_amd.synthetic import frames
dist.init_process_group("gloo", init_method="env://")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda:0")
enc = GopEncoder(IntraNoAR().to(dev).eval(), DMC().to(dev).eval(), gop_size=3)
def gop(g):
    fr = frames(50 + g, 3, 64, 128)
    return [torch.from_numpy(fr[t:t+1]).to(dev) for t in range(3)]
mine = shard_gops(5, rank, world)
dt, coded = timed_region(lambda: {g: [c[2] for c in enc.encode_gop(gop(g), 1.0, 1.0, 1.0)[0]] for g in mine}, dev)
out = [None] * world
dist.all_gather_object(out, (mine, coded, dt))
if rank == 0:
    pickle.dump(out, open(os.environ["DCVC_OUT"], "wb"))
dist.destroy_process_group()
"""


def _run(args, env):
    e = dict(os.environ, **env)
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                           "127.0.0.1", "--master-port", "29541"] + args, env=e, capture_output=True, text=True, timeout=600)


def test_two_ranks_encode_disjoint_gops(tmp_path):
    import pickle

    script = os.path.join(tmp_path, "worker.py")
    open(script, "w").write(WORKER)
    out = os.path.join(tmp_path, "out.pkl")
    r = _run([script], {"DCVC_ROOT": ROOT, "DCVC_OUT": out})
    assert r.returncode == 0, r.stderr[-2000:]
    gathered = pickle.load(open(out, "rb"))
    seen = {}
    for mine, coded, dt in gathered:
        assert sorted(coded) == mine
        seen.update(coded)
    assert sorted(seen) == list(range(5))
    assert len({dt for _, _, dt in gathered}) == 1
    # single-process reference encode of the same GOPs
    from vcm_ts_amd.dmc import DMC
    from vcm_ts_amd.intra import IntraNoAR
    from vcm_ts_amd.pipeline import GopEncoder
    from vcm_ts_amd.synthetic import frames

    dev = torch.device("cuda:0")
    enc = GopEncoder(IntraNoAR().to(dev).eval(), DMC().to(dev).eval(), gop_size=3)
    for g in range(5):
        fr = frames(50 + g, 3, 64, 128)
        seq = [torch.from_numpy(fr[t : t + 1]).to(dev) for t in range(3)]
        assert [c[2] for c in enc.encode_gop(seq, 1.0, 1.0, 1.0)[0]] == seen[g], g


def test_bench_runs_with_world_size_two():
    r = _run([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--height", "128", "--width", "192",
              "--gop", "3", "--dist-backend", "gloo", "--no-cpu-baseline"], {})
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert {"roofline", "metric", "unit", "ms_per_step", "config"} <= set(d)
