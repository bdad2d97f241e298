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


def _run(args, env, nproc=2):
    e = dict(os.environ, **env)
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
                           "--master-addr", "127.0.0.1", "--master-port", "29541"] + args, env=e, capture_output=True,
                          text=True, timeout=900)


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


def test_bench_two_ranks_two_gop_streams_match_the_one_rank_run():
    """the driver's N > 1 form rehearsed as far as one GPU allows -- `bench.py --gpus 2
    --dist-backend gloo`, two ranks x two GOP streams each on the one card (four GOPs in flight: the per-GPU structure
    of the 8-GPU run) -- must join both ranks in the collective and code rank 0's first GOP into exactly the bytes the
    one-rank run produces (GOPs are independent: sharding changes nothing a decoder sees)."""
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--height", "256", "--width", "384",
            "--gop", "4", "--gop-streams", "2", "--no-cpu-baseline", "--no-parity-leg", "--no-extra-workloads"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    lines = {}
    for n in (1, 2):
        r = subprocess.run(base + ["--gpus", str(n), "--dist-backend", "gloo"], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        lines[n] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    one, two = lines[1], lines[2]
    assert two["n_gpus"] == 2 and two["config"]["collective_ranks"] == 2 and len(two["config"]["per_rank_frames_per_s"]) == 2
    assert two["config"]["gops_per_step_per_gpu"] == 2
    assert two["config"]["payload_sha16_gop0"] == one["config"]["payload_sha16_gop0"]
    assert two["config"]["bits_per_gop"] > 0 and two["config"]["fp16x3_range_status"] == 0
    for d in (one, two):
        c = d["config"]
        assert c["host_cpu_s_per_frame"] > 0 and c["host_threads"] >= 1 and c["hbm_bytes_reserved"] >= c["hbm_bytes_workspaces"] > 0


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` by itself (no torchrun around it, the form the driver uses): the parent starts
    one rank per GPU as a child torch.distributed.run before touching the GPU and hands back its exit code; a
    world size that contradicts --gpus is refused."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--height", "128",
           "--width", "192", "--gop", "3", "--dist-backend", "gloo", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["value"] > 0
    bad = subprocess.run(cmd[:3] + ["1"] + cmd[4:], env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True,
                         text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE" in (bad.stderr + bad.stdout)


DDP_WORKER = r"""
import os, sys, pickle
sys.path.insert(0, os.environ["DCVC_ROOT"])
import numpy as np, torch, torch.distributed as dist
from vcm_ts_amd.dcvc_hem import build_model, make_cfg
from vcm_ts_amd.synthetic import frames
dist.init_process_group("gloo", init_method="env://")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda:0")
model = build_model(make_cfg(lambdas=(85.0, 380.0)), precision="fp32").to(dev).train()
model.activate_modules_all()
g = torch.Generator().manual_seed(7)
model.dmc._noise_override = {"y": torch.rand(2, 96, 4, 4, generator=g) - 0.5, "mv_y": torch.rand(2, 64, 4, 4, generator=g) - 0.5,
                             "z": torch.rand(2, 64, 1, 1, generator=g) - 0.5, "mv_z": torch.rand(2, 64, 1, 1, generator=g) - 0.5}
ddp = torch.nn.parallel.DistributedDataParallel(model, find_unused_parameters=True)   # train_multi.py:179
opt = torch.optim.SGD(ddp.parameters(), lr=1e-4)
x = torch.from_numpy(np.stack([frames(70 + 10 * rank + i, 3, 64, 64) for i in range(2)])).to(dev)   # this rank's clip
before = {k: v.detach().clone() for k, v in model.dmc.named_parameters()}
dpb = {"ref_frame": x[:, 0], "ref_feature": None, "ref_y": None, "ref_mv_y": None}
for t in (1, 2):                               # single_step of train_multi.py:203-232, two P pictures
    opt.zero_grad()
    r = ddp("single_multi", x[:, t], x[:, t], "mse", ["bpp"], perceptual_loss=False, dpb=dpb)
    r["loss_to_opt"].backward()
    if t == 1:
        grads = {k: v.grad.detach().cpu().clone() for k, v in model.dmc.named_parameters() if v.grad is not None}
    opt.step()
    dpb = r["dpb"]
after = {k: v.detach().cpu() for k, v in model.dmc.named_parameters()}
out = [None] * world
dist.all_gather_object(out, (grads, {k: after[k] for k in ("recon_generation_net.recon_conv.weight", "optic_flow.moduleBasic.0.conv1.bias",
                                                          "y_q_basic", "feature_adaptor_P.weight", "feature_adaptor_I.weight")}))
if rank == 0:
    pickle.dump(out, open(os.environ["DCVC_OUT"], "wb"))
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world", [2, 4])
def test_ddp_training_step_averages_gradients_across_ranks(tmp_path, world):
    """trainer_multi.py's step with the model wrapped in DistributedDataParallel: after backward every rank holds
    the mean of the per-rank gradients, and parameters stay identical.  world 4 rehearses BASELINE configs[3]
    beyond two ranks (gloo rendezvous: all ranks share the one GPU of the test box)."""
    import pickle

    script = os.path.join(tmp_path, "ddp_worker.py")
    open(script, "w").write(DDP_WORKER)
    out = os.path.join(tmp_path, "ddp.pkl")
    r = _run([script], {"DCVC_ROOT": ROOT, "DCVC_OUT": out}, nproc=world)
    assert r.returncode == 0, r.stderr[-3000:]
    gathered = pickle.load(open(out, "rb"))
    assert len(gathered) == world
    (g0, p0) = gathered[0]
    assert "feature_adaptor_I.weight" in g0 and "feature_adaptor_P.weight" not in g0
    for g1, p1 in gathered[1:]:
        assert set(g0) == set(g1)
        for k in g0:
            assert torch.equal(g0[k], g1[k]), k                    # all-reduced: identical on every rank
        for k in p0:
            assert torch.isfinite(p0[k]).all() and torch.equal(p0[k], p1[k]), k   # so are the parameters after two steps
    # the all-reduced gradient is the mean of what each rank computes alone
    from vcm_ts_amd.dcvc_hem import build_model, make_cfg
    from vcm_ts_amd.synthetic import frames

    dev = torch.device("cuda:0")
    model = build_model(make_cfg(lambdas=(85.0, 380.0)), precision="fp32").to(dev).train()
    model.activate_modules_all()
    g = torch.Generator().manual_seed(7)
    model.dmc._noise_override = {"y": torch.rand(2, 96, 4, 4, generator=g) - 0.5, "mv_y": torch.rand(2, 64, 4, 4, generator=g) - 0.5,
                                 "z": torch.rand(2, 64, 1, 1, generator=g) - 0.5, "mv_z": torch.rand(2, 64, 1, 1, generator=g) - 0.5}
    acc = {}
    for rank in range(world):
        x = torch.from_numpy(np.stack([frames(70 + 10 * rank + i, 3, 64, 64) for i in range(2)])).to(dev)
        model.zero_grad(set_to_none=True)
        dpb = {"ref_frame": x[:, 0], "ref_feature": None, "ref_y": None, "ref_mv_y": None}
        res = model("single_multi", x[:, 1], x[:, 1], "mse", ["bpp"], perceptual_loss=False, dpb=dpb)
        res["loss_to_opt"].backward()
        for k, v in model.dmc.named_parameters():
            if v.grad is not None:
                acc[k] = acc.get(k, 0) + v.grad.detach().cpu() / world
    assert set(acc) == set(g0)
    for k in acc:
        torch.testing.assert_close(g0[k], acc[k], rtol=1e-4, atol=1e-6 + 1e-5 * float(acc[k].abs().max()), msg=k)


def test_bench_train_workload_with_world_size_two():
    r = _run([os.path.join(ROOT, "bench.py"), "--workload", "train", "--gpus", "2", "--steps", "2", "--warmup", "1",
              "--dist-backend", "gloo", "--no-cpu-baseline"], {})
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 8 and line["value"] > 0
    assert np.isfinite(line["config"]["final_loss"])


def test_bench_runs_over_rccl_with_one_rank():
    """`--dist-backend nccl` (= RCCL on ROCm) on the one GPU of the test box: a launcher with ONE rank makes
    bench.py create the RCCL communicator and run the barrier + MAX all-reduce of the timed region on the
    device -- the calls every rank makes at N > 1 (two ranks cannot share one GPU under RCCL)."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
                        "127.0.0.1", "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1",
                        "--warmup", "0", "--height", "128", "--width", "192", "--gop", "3", "--dist-backend", "nccl",
                        "--no-cpu-baseline", "--no-parity-leg", "--no-extra-workloads"],
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["config"]["dist_backend"] == "nccl" and d["value"] > 0


def test_training_bench_runs_ddp_over_rccl_with_one_rank():
    """The trainer step of bench.py --workload train under a one-rank launcher with the nccl (= RCCL) backend:
    DistributedDataParallel buckets and all-reduces the 17.5 M gradient floats over RCCL on the GPU."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
                        "127.0.0.1", "--master-port", "29548", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "train",
                        "--steps", "2", "--warmup", "1", "--dist-backend", "nccl", "--no-cpu-baseline"],
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and "ddp x1 (nccl)" in d["config"]["parallelism"] and d["value"] > 0
