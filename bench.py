"""bench.py -- encoded frames/s at 1920x1080, GOP 32 (BASELINE.json metric) on N MI355X.

One "step" = one GOP of 32 synthetic 1920x1080 frames (1 I + 31 P pictures, padded to
1088x1920) through the real encode path: all networks on the HIP kernels, symbol planes to the
host, rANS coding into the payload bytes, DPB kept on the device.  Inputs are resident in HBM
before the timed region.  With N > 1 every rank encodes its own GOPs (GOPs are independent:
weak scaling, no data-path collective); the timed region is bracketed by barriers and the
slowest rank's time is used.

Prints ONE JSON line (see the task contract) with two extra objects:
  roofline     achieved TFLOP/s of the dominant kernel (3x3 stride-1 fp32-MFMA convolution),
               = algorithmic conv FLOPs of its launches / their summed HIP-event durations,
               measured live on the launch stream in one extra (untimed) P picture
  cpu_baseline the CPU oracle (oracle/dcvc_ref.py, a torch-CPU port of the reference path)
               timed on this box's host cores on a bounded sample (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md, "Peak BF16/FP16 MFMA" (dense)
PEAK_HBM_TBS = 8.0             # MI355X_MICROARCH.md, HBM3E


def synth_sequence(dev, n, h, w, seed):
    """n frames (1,3,h,w) on the device: smooth field + global motion + noise (synthetic.py
    gives the base picture; the per-frame shifts are done on the device to keep start-up short)."""
    from vcm_ts_amd.synthetic import frames

    base = torch.from_numpy(frames(seed, 1, h, w)[0]).to(dev)
    g = torch.Generator(device=dev).manual_seed(seed)
    out = []
    for t in range(n):
        f = torch.roll(base, shifts=(t, -2 * t), dims=(1, 2))
        f = (f + torch.randn(f.shape, generator=g, device=dev) / 255.0).clamp_(0, 1)
        out.append(f[None].contiguous())
    return out


def host_cores():
    """CPU threads this process may really use: cgroup quota if set, else affinity, and never
    more than the 16-core host share a 1-GPU box is given."""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 16))


def host_cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def kernel_source_hash():
    """sha256 (16 hex) over the convolution kernel sources and the build flags: a committed PMC figure is only
    quoted for the kernels it was measured on."""
    import hashlib

    h = hashlib.sha256()
    for name in ("conv_mfma.hip", "conv_k32.hip", "Makefile"):
        with open(os.path.join(ROOT, "vcm_ts_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(precision, height, width):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic_<precision>.json, written by tools/rocprof_traffic.py with the gfx950 FETCH_SIZE x2
    correction of MI355X_MICROARCH.md); bench.py cannot run the profiler on itself.  Quoted only when the
    file was measured on THIS picture size and THESE kernel sources, else null with the reason."""
    path = os.path.join(ROOT, "profiles", f"pmc_traffic_{precision}.json")
    try:
        d = json.load(open(path))
    except (OSError, ValueError):
        return None, "no PMC pass committed for this precision"
    if (d.get("height"), d.get("width")) != (height, width):
        return None, f"committed PMC pass is for {d.get('height')}x{d.get('width')}, this run is {height}x{width}"
    if d.get("kernel_source_sha16") != kernel_source_hash():
        return None, "kernel sources changed since the committed PMC pass (re-run tools/rocprof_traffic.py)"
    return d["bytes_per_launch"], f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, {os.path.basename(path)}"


def parity_vs_reference(precision):
    """Standing of this arithmetic mode against the REFERENCE's own 1088x1920 runs, per depth in the GOP, from the committed
    output of the GPU suite (profiles/r04_parity_vs_reference.json, tools/parity_summary.py): the deviation a reader should
    hold next to `value`.  bench.py does not re-measure it (the fixtures are test data)."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r04_parity_vs_reference.json")))
    except (OSError, ValueError):
        return None
    out = {"tolerance": 1e-4, "source": "profiles/r04_parity_vs_reference.json (GPU suite, tests/test_gpu_codec.py)"}
    for name, case in d.get("cases", {}).items():
        if precision in case:
            out[name] = {"worst_relative_deviation_of_a_total": case[precision]["worst_total"],
                         "per_picture_P1_to_P7": case[precision]["per_picture_totals"]}
    return out


def cpu_baseline(h, w, threads):
    """One P-picture through the CPU oracle's networks (dmc_analysis) = the nets of DMC.compress; bounded
    sample so the default run stays within minutes.  Returns seconds."""
    from oracle import dcvc_ref as R
    from vcm_ts_amd.params import dmc_spec, seeded_state_dict
    from vcm_ts_amd.synthetic import frames

    torch.set_num_threads(threads)
    wd = seeded_state_dict(dmc_spec())
    fr = frames(7, 2, h, w)
    x0, x1 = torch.from_numpy(fr[0:1]), torch.from_numpy(fr[1:2])
    dpb = {"ref_frame": x0, "ref_feature": None, "ref_y": None, "ref_mv_y": None}
    with torch.no_grad():
        t0 = time.time()
        R.dmc_analysis(wd, x1, dpb, 1.0, 1.0)
        dt = time.time() - t0
    return dt


def cpu_rans_baseline(planes):
    """The oracle's C restatement of the reference's rANS coder (oracle/rans_ref.c, single thread like the
    reference: rans_interface.cpp has no threads) on the symbol planes of one real P picture of this run:
    (encode seconds, decode seconds, payload bytes).  planes: list of (symbols, indexes, cdf, sizes, offsets)."""
    import ctypes as C

    import numpy as np

    L = C.CDLL(os.path.join(ROOT, "oracle", "liboracle_ref.so"))
    L.ref_enc_new.restype = C.c_void_p
    L.ref_enc_free.argtypes = [C.c_void_p]
    L.ref_enc_encode_with_indexes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p,
                                              C.c_void_p]
    L.ref_enc_flush.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    L.ref_enc_flush.restype = C.c_void_p
    L.ref_free.argtypes = [C.c_void_p]
    L.ref_dec_new.restype = C.c_void_p
    L.ref_dec_free.argtypes = [C.c_void_p]
    L.ref_dec_set_stream.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.ref_dec_decode_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    arrs = [[np.ascontiguousarray(a, np.int32) for a in p] for p in planes]
    t0 = time.time()
    e = L.ref_enc_new()
    for s, i, c, l, o in arrs:
        L.ref_enc_encode_with_indexes(e, s.ctypes.data, i.ctypes.data, s.size, c.ctypes.data, c.shape[1], l.ctypes.data, o.ctypes.data)
    n = C.c_size_t()
    ptr = L.ref_enc_flush(e, C.byref(n))
    t_enc = time.time() - t0
    data = C.string_at(ptr, n.value)
    L.ref_free(ptr)
    L.ref_enc_free(e)
    t0 = time.time()
    d = L.ref_dec_new()
    L.ref_dec_set_stream(d, data, len(data))
    for s, i, c, l, o in arrs:
        out = np.empty(i.size, np.int32)
        L.ref_dec_decode_stream(d, i.ctypes.data, i.size, c.ctypes.data, c.shape[1], l.ctypes.data, o.ctypes.data, out.ctypes.data)
        assert np.array_equal(out, s)
    t_dec = time.time() - t0
    L.ref_dec_free(d)
    return t_enc, t_dec, len(data)


def picture_planes(p_net, views):
    """(symbols, indexes, cdf, sizes, offsets) of the six planes of a P picture, in coding order, on the host."""
    import numpy as np

    tabs = p_net._tables
    out = []
    for name, sym, idx in (("bit_estimator_z_mv", views["sym_mv_z"], None), ("scale", views["r_mv"]["sym"][0], views["r_mv"]["idx"][0]),
                           ("scale", views["r_mv"]["sym"][1], views["r_mv"]["idx"][1]), ("bit_estimator_z", views["sym_z"], None),
                           ("scale", views["r_y"]["sym"][0], views["r_y"]["idx"][0]), ("scale", views["r_y"]["sym"][1], views["r_y"]["idx"][1])):
        s = sym.cpu().numpy().reshape(-1)
        if idx is None:  # factorised prior: the CDF row is the channel (64 channels, (n, c, y, x) order)
            hw = s.size // 64
            i = np.repeat(np.arange(64, dtype=np.int32), hw)
        else:
            i = idx.cpu().numpy().reshape(-1)
        out.append((s, i) + tuple(tabs[name]))
    return out


class GopQuality:
    """Per-picture squared error of the reconstructions against the source pictures on the coded area
    (video_coder.py:145-151 crops the padding before computing PSNR), accumulated on the device."""

    def __init__(self, seq, height, width):
        self.seq, self.h, self.w, self.sse = seq, height, width, []

    def __call__(self, t, ref_frame):
        d = (ref_frame[..., : self.h, : self.w] - self.seq[t][..., : self.h, : self.w]).double()
        self.sse.append((d * d).sum())

    def psnr(self):
        mse = torch.stack(self.sse).cpu().numpy() / (3.0 * self.h * self.w)
        import numpy as np

        return 10.0 * np.log10(1.0 / mse)


def files_workload(args, dev, n_frames, nets=None):
    """the reference's file loop (video_coder.run_dcvc: PNG in, one .bin per picture out) around the same
    encoder, reported SEPARATELY from `value`: PNG decode + host->device + encode + rANS + .bin write per picture, from a
    temporary folder of synthetic 8-bit PNGs (written untimed).  vcm_ts_amd/run_codec.py.  Round 4: the folder's GOPs go
    through ConcurrentGopEncoder like the headline (args.gop_streams GOPs in flight, one shared pool of PNG-decoding
    threads; .bin bytes identical to the one-stream loop, which is timed beside it).
    nets: list of (i_frame_net, p_frame_net) pairs to reuse (one per GOP stream)."""
    import shutil
    import tempfile
    import time
    from concurrent.futures import ThreadPoolExecutor

    from vcm_ts_amd.run_codec import _nets, _save_array, encode_folder

    K = max(1, args.gop_streams)
    tmp = tempfile.mkdtemp(prefix="dcvc_files_")
    try:
        src, dst = os.path.join(tmp, "png"), os.path.join(tmp, "bin")
        os.makedirs(src)
        with ThreadPoolExecutor(max_workers=8) as pool:  # (untimed) PNG encoding of noise is slow: spread it
            for t, f in enumerate(synth_sequence(dev, n_frames, args.height, args.width, 11)):
                pool.submit(_save_array, f.squeeze(0).permute(1, 2, 0).cpu().numpy(), os.path.join(src, f"im{str(t + 1).zfill(5)}.png"))
        png_bytes = sum(os.path.getsize(os.path.join(src, n)) for n in os.listdir(src))
        if nets is None:  # built once, outside the timed loop (run_dcvc loads its models before its loop too)
            nets = [_nets(dev, args.precision) for _ in range(K)]
        common = dict(gop=args.gop, q=(1.0, 1.0, 1.0), device=str(dev), precision=args.precision)
        encode_folder(src, dst, None, max_frames=min(n_frames, args.gop + 2), nets=nets, gop_streams=K, **common)  # warm-up

        def timed(io_workers, streams):
            shutil.rmtree(dst, ignore_errors=True)
            torch.cuda.synchronize(dev)
            t0 = time.time()
            bits, size = encode_folder(src, dst, None, io_workers=io_workers, nets=nets, gop_streams=streams, **common)
            torch.cuda.synchronize(dev)
            return time.time() - t0, bits, size

        dt_inline, _, _ = timed(0, 1)   # PNGs decoded in the encode loop, one GOP at a time, as run_dcvc does
        dt_one, bits_one, _ = timed(8, 1)  # round 3's loop: 8 reader threads, one GOP stream
        dt, bits, size = timed(8, K)    # run_codec --gop-streams K: the folder's GOPs in flight together
        assert bits == bits_one, "the same .bin sizes whatever the number of GOP streams"
        return {"value": round(len(bits) / dt, 3), "unit": "frames/s", "frames": len(bits), "ms_per_frame": round(dt / len(bits) * 1e3, 2),
                "io_threads": 8, "gop_streams": min(K, (len(bits) + args.gop - 1) // args.gop),
                "one_gop_stream_frames_per_s": round(len(bits) / dt_one, 3),
                "inline_io_frames_per_s": round(len(bits) / dt_inline, 3),
                "png_mbytes_read": round(png_bytes / 1e6, 1), "bin_mbytes_written": round(sum(bits) / 8e6, 1),
                "workload": f"run_codec encode of a folder of {len(bits)} synthetic {size[1]}x{size[0]} PNGs into .bin files (GOP "
                            f"{args.gop}, {K} GOPs in flight, PNG decode and file writes inside the timed region; noise PNGs are "
                            f"the slowest to decode); = bench.py --workload files"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def train_workload(args, dev, rank, world):
    """BASELINE configs[2] (N=1) / configs[3] (N>1): a step is forward_one_frame + backward + AdamW on
    a batch of 4 256x256 pictures per GPU (`single` mode: one optimiser step per P picture,
    core/model/dcvc_hem.py:189-229), gradients all-reduced by DistributedDataParallel when N>1."""
    from vcm_ts_amd.dcvc_hem import build_model, make_cfg
    from vcm_ts_amd.pipeline import timed_region

    batch, size = 4, 256
    prec = "fp32" if args.precision == "fp32" else "fp16x3"
    model = build_model(make_cfg(lambdas=(85.0, 170.0, 380.0, 840.0)), precision=prec).to(dev).train()
    model.activate_modules_all()
    # --graph-training: replay each picture's launches from captured hipGraphs (DMC.graph_training; same launches, same
    # results: tests/test_gpu_wrapper.py).  Measured in round 4 it is SLOWER here (30.4 ms against 26.2 ms per step): the
    # step is bound by the GPU's own per-kernel gaps, not by the host, and a replayed graph runs its kernel nodes one
    # after the other, losing the overlap of the weight-gradient stream with the data-gradient chain
    model.dmc.graph_training = bool(getattr(args, "graph_training", False))
    net = model
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        # also with ONE rank under a launcher: DistributedDataParallel's bucketed all-reduce then runs over RCCL
        # on the single GPU, the code path of trainer_multi.py at N > 1
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev.index] if args.dist_backend == "nccl" else None,
                                                        find_unused_parameters=True)
    # the reference's optimiser (core/solver/optimizer.py:13); fused=True is torch's single-kernel implementation of the
    # same update (the default per-tensor-list one costs ~4 ms of host time per step here)
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4, betas=(0.9, 0.99), fused=True)
    g = torch.Generator().manual_seed(100 + rank)
    clip = torch.rand(batch, args.warmup + args.steps + 1, 3, size, size, generator=g).to(dev)

    def run(t0, n):
        r = None
        dpb = {"ref_frame": clip[:, t0], "ref_feature": None, "ref_y": None, "ref_mv_y": None}
        for t in range(t0 + 1, t0 + 1 + n):
            opt.zero_grad()
            r = net("single_multi", clip[:, t], clip[:, t], "mse", ["bpp"], perceptual_loss=False, dpb=dpb)
            r["loss_to_opt"].backward()
            opt.step()
            dpb = r["dpb"]
        return float(r["loss_to_opt"].detach()) if r is not None else float("nan")

    run(0, 1)  # one-off initialisation (not a step): weight packing, buffers, optimiser state
    run(0, args.warmup)
    dt, loss = timed_region(lambda: run(args.warmup, args.steps), dev)
    eng = model.dmc.engine()
    eng.profile = {}
    graphed, model.dmc.graph_training = model.dmc.graph_training, False  # (events go around Python-issued launches)
    run(0, 1)
    model.dmc.graph_training = graphed
    prof = eng.collect_profile()
    eng.profile = None
    dom = prof.get("conv3x3s1", {"flops": 0.0, "ms": 1.0, "launches": 0})
    peak = PEAK_F32_MFMA_TFLOPS if prec == "fp32" else PEAK_F16_MFMA_TFLOPS / 3.0
    achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
    out = {"metric": "trainer step pictures/sec (batch 4 x 256x256 per GPU, bpp+MSE, AdamW)", "value": round(world * batch * args.steps / dt, 2),
           "unit": "pictures/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32" if prec == "fp32" else "f32 (fp16x3 split-MFMA convolutions; weight gradients: split-bf16 MFMA for stride-1 layers, fp32 MFMA for stride-2 layers)", "data": "synthetic",
           "config": {"workload": "trainer.py / trainer_multi.py optimiser step (configs[2] at N=1, configs[3] at N>1): "
                                  "forward_one_frame + backward + AdamW, single mode, uniform-random clips, random-init weights",
                      "batch_per_gpu": batch, "global_batch": batch * world, "height": size, "width": size, "precision": prec,
                      "parallelism": f"ddp x{world} ({args.dist_backend})" if net is not model else "single GPU", "final_loss": round(loss, 4),
                      "launch_mode": "hipGraph replay of the picture's forward and reverse pass" if model.dmc.graph_training else "eager (Python-issued launches)"},
           "roofline": {"bound": "mfma", "kernel": "conv_mfma<3,1,*> forward + data-gradient launches of one step",
                        "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                        "traffic": None, "launches": dom["launches"]}}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import dcvc_ref as R  # checker timed as the CPU baseline, never on the product path
        from vcm_ts_amd.params import dmc_spec, seeded_state_dict

        cores = host_cores()
        torch.set_num_threads(cores)
        w = {k: v.clone().requires_grad_() for k, v in seeded_state_dict(dmc_spec()).items()}
        x = clip[:, :2].cpu()
        noise = {"y": torch.rand(batch, 96, size // 16, size // 16) - 0.5, "mv_y": torch.rand(batch, 64, size // 16, size // 16) - 0.5,
                 "z": torch.rand(batch, 64, size // 64, size // 64) - 0.5, "mv_z": torch.rand(batch, 64, size // 64, size // 64) - 0.5}
        t0 = time.time()
        with R.training_mode():
            o = R.dmc_forward_one_frame(w, x[:, 1], {"ref_frame": x[:, 0], "ref_feature": None, "ref_y": None, "ref_mv_y": None},
                                        w["mv_y_q_scale"], w["y_q_scale"], noise=noise)
        (o["bpp"] + torch.tensor([85.0, 170.0, 380.0, 840.0]) * o["mse"]).mean().backward()
        sec = time.time() - t0
        out["cpu_baseline"] = {"value": round(batch / sec, 3), "unit": "pictures/s", "cores": cores, "cpu_model": host_cpu_model(), "kind": "port",
                               "sample": f"1 step (forward + torch.autograd backward, no optimiser) of the oracle on the same batch: {sec:.1f} s"}
    return out


def self_launch(n):
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--gop", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-leg", action="store_true", help="skip the extra fp32-mode GOP (N=1 only)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL) on a multi-GPU node; gloo to rehearse on one GPU")
    ap.add_argument("--precision", default=os.environ.get("DCVC_PRECISION", "fp16x3"), choices=["fp32", "fp16x3"],
                    help="convolution arithmetic: exact fp32 MFMA or split-fp16 MFMA (3 products, fp32 accumulate)")
    ap.add_argument("--cpu-size", type=int, nargs=2, default=None, help="H W of the CPU sample (default: padded full size)")
    ap.add_argument("--gop-streams", type=int, default=2,
                    help="GOPs in flight per GPU (own codec instances and HIP stream each, one host thread); a step "
                         "codes this many GOPs")
    ap.add_argument("--workload", default="encode", choices=["encode", "decode", "train", "files"],
                    help="files: the PNG-folder -> .bin-folder loop of run_codec (file I/O inside the timed region); "
                         "encode: BASELINE configs[1] (the headline metric, default); decode: the same GOPs through "
                         "decompress (payloads made once, untimed); train: configs[2]/[3], one optimiser step of "
                         "trainer.py / trainer_multi.py per bench step (batch 4 of 256x256 per GPU, DDP over RCCL)")
    ap.add_argument("--graph-training", action="store_true",
                    help="--workload train: replay every picture from captured hipGraphs instead of issuing its launches from "
                         "Python (same results; slower on this runtime, see DESIGN.md 4b)")
    ap.add_argument("--no-extra-workloads", action="store_true",
                    help="N=1 encode run: skip the short decode and training-step measurements added to the JSON line")
    ap.add_argument("--strict-parity", action="store_true",
                    help="exit 3 (after printing the JSON line) when the fast mode misses the 1e-4 tolerance against the "
                         "fp32 mode or the range check fires; the line always carries within_tolerance")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by itself: start one rank per GPU (env:// rendezvous on 127.0.0.1, as
        # trainer_multi.py:16-39 does with mp.spawn) as a CHILD torch.distributed.run -- this process has not
        # touched the GPU yet and never will -- and hand back its exit code.  Rank 0 prints the JSON line.
        raise SystemExit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch one rank per GPU "
                         "(python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...) or let bench.py do it")
    # under a launcher (RANK set) the process group is created even for one rank: `torchrun --nproc-per-node 1
    # bench.py --dist-backend nccl` then runs the RCCL communicator, barrier and MAX-reduction of the timed
    # region on a single GPU, the same calls every rank makes at N > 1
    grouped = world > 1 or "RANK" in os.environ
    if grouped:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=args.dist_backend, init_method="env://")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    dev = torch.device("cuda", local % torch.cuda.device_count())
    torch.cuda.set_device(dev)

    if args.workload == "files":
        if world != 1:
            raise SystemExit("bench.py --workload files is a one-GPU measurement")
        r = files_workload(args, dev, args.gop * max(1, args.gop_streams))
        print(json.dumps({"metric": "encoded frames/sec at 1920x1080 GOP-32, PNG folder in, .bin folder out", "value": r["value"],
                          "unit": "frames/s", "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": round(r["ms_per_frame"] * r["frames"], 2),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": r}))
        return
    if args.workload == "train":
        out = train_workload(args, dev, rank, world)
        if rank == 0:
            print(json.dumps(out))
        if grouped:
            dist.destroy_process_group()
        return

    from vcm_ts_amd.dmc import DMC
    from vcm_ts_amd.intra import IntraNoAR
    from vcm_ts_amd.pipeline import ConcurrentGopEncoder, GopEncoder, pad_frame, timed_region

    # K GOPs of the sequence in flight per GPU (GOPs are independent: the sharding of SURVEY 8e applied
    # inside the GPU as well), each with its own codec instances and HIP stream
    K = max(1, args.gop_streams)
    cenc = ConcurrentGopEncoder(lambda: (IntraNoAR(precision=args.precision).to(dev).eval(),
                                         DMC(precision=args.precision).to(dev).eval()), gop_size=args.gop, streams=K)
    enc = cenc.encoders[0]
    i_net, p_net = enc.i_net, enc.p_net
    seqs = [[pad_frame(f) for f in synth_sequence(dev, args.gop, args.height, args.width, seed=rank * 16 + k)] for k in range(K)]
    seq = seqs[0]
    q_i, q_mv, q_y = 1.0, 1.0, 1.0

    # one-off initialisation (not a step): pack the weights, allocate the workspaces and the pinned
    # staging buffers by coding the first pictures of every stream once
    cenc.encode_gops([sq[:3] for sq in seqs], q_i, q_mv, q_y)
    bits = 0
    for _ in range(args.warmup):
        bits = sum(r[1] for r in cenc.encode_gops(seqs, q_i, q_mv, q_y))

    if args.workload == "decode":
        # payloads made once (untimed); a step decodes K GOPs, one host thread + HIP stream per GOP
        coded = [r[0] for r in cenc.encode_gops(seqs, q_i, q_mv, q_y)]
        for _ in range(max(1, args.warmup)):
            cenc.decode_gops(coded, args.height, args.width)

        def dwork():
            for _ in range(args.steps):
                cenc.decode_gops(coded, args.height, args.width)
            return 0

        dt, _ = timed_region(dwork, dev)
        bits = sum(len(p[2]) * 8 for p in coded[0])
        out = {"metric": "decoded frames/sec at 1920x1080 GOP-32", "value": round(K * args.gop * args.steps * world / dt, 3),
               "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32" if args.precision == "fp32" else "f32 (fp16x3 split-MFMA, fp32 accumulate)", "data": "synthetic",
               "config": {"workload": "decode of the configs[1] GOPs (reference bitstream format: 3 / 6 host rANS round trips per "
                                      f"I / P picture); {K} GOPs in flight per GPU, one host thread and HIP stream each",
                          "gop": args.gop, "height": args.height, "width": args.width, "precision": args.precision,
                          "bits_per_gop": int(bits)}}
        if rank == 0:
            print(json.dumps(out))
        if grouped:
            dist.destroy_process_group()
        return

    bits_gop0 = [0]
    last_coded = [None]

    def work():
        b = 0
        for _ in range(args.steps):
            res = cenc.encode_gops(seqs, q_i, q_mv, q_y)
            b = sum(r[1] for r in res)
            bits_gop0[0] = res[0][1]
            last_coded[0] = res[0][0]
        return b

    # what a rank costs the HOST while it codes (for sizing an 8-rank node without having one):
    # CPU seconds of this process over the timed region (every thread: the feeder, the rANS coder, torch's helpers),
    # its thread count, and the device memory it holds
    cpu0 = time.process_time()
    dt, bits, dt_local = timed_region(work, dev, with_local=True)
    host_cpu_s = time.process_time() - cpu0
    try:
        host_threads = len(os.listdir("/proc/self/task"))
    except OSError:
        host_threads = None
    import hashlib

    payload_sha = hashlib.sha256(b"".join(c[2] for c in last_coded[0])).hexdigest()[:16]
    bits = bits / K  # per GOP
    frames_total = K * args.gop * args.steps * world
    fps = frames_total / dt
    # the same GOP alone on the GPU (one stream), untimed for `value`: the one-stream rate, then once more with the
    # reconstructions' PSNR collected.  The range guard of the fast mode (every convolution output against the
    # magnitude split-fp16 operands can hold) is on in every launch of this program, the timed ones included
    eng_i, eng_p = i_net.engine(), p_net.engine()
    assert eng_i.range_check and eng_p.range_check
    dt1, bits1 = timed_region(lambda: enc.encode_gop(seq, q_i, q_mv, q_y)[1], dev)
    one_stream = round(args.gop / dt1, 3)
    quality = GopQuality(seq, args.height, args.width)
    bits1_again = enc.encode_gop(seq, q_i, q_mv, q_y, on_recon=quality)[1]
    assert bits1_again == bits1, "the same GOP codes the same bytes"
    saturation = eng_i.read_status() | eng_p.read_status()
    for e_ in cenc.encoders:
        saturation |= e_.i_net.engine().read_status() | e_.p_net.engine().read_status()
    psnr_fast = quality.psnr()

    # ---- roofline of the dominant kernel, measured live with HIP events on the launch stream
    def conv_roofline(i_n, p_n, precision):
        eng = p_n.engine()
        eng.profile = {}
        dpb0 = {"ref_frame": i_n.compress(seq[0], q_i)["x_hat"].clone(), "ref_feature": None, "ref_y": None, "ref_mv_y": None}
        dpb = p_n.compress(seq[1], dpb0, q_mv, q_y)["dpb"]
        views = p_n.compress(seq[2], dpb, q_mv, q_y)["_views"]
        torch.cuda.synchronize(dev)
        prof = eng.collect_profile()
        eng.profile = None
        # the HBM-bound kernels, timed in a second pass WITHOUT the side stream (the feature pyramid's convolutions
        # would otherwise run beside the warps and resamplers and lengthen them): every launch by itself
        eng.profile_hbm = {}
        fork, p_n.fork_features = p_n.fork_features, False
        dpb = p_n.compress(seq[1], dpb0, q_mv, q_y)["dpb"]
        p_n.compress(seq[2], dpb, q_mv, q_y)
        hbm = eng.collect_profile_hbm()
        eng.profile_hbm = None
        p_n.fork_features = fork
        # fast mode: the 3x3 stride-1 layers whose input segments are multiples of 32 channels (all heavy ones) run on
        # conv_k32 (profile key "...k32"); the few others stay on conv_mfma and are not part of the dominant kernel
        dom_key = "conv3x3s1k32" if precision != "fp32" and "conv3x3s1k32" in prof else "conv3x3s1"
        dom = prof.get(dom_key, {"flops": 0.0, "ms": 1.0, "launches": 0})
        all_flops = sum(v["flops"] for v in prof.values())
        all_ms = sum(v["ms"] for v in prof.values())
        achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        if precision == "fp32":
            peak, kname = PEAK_F32_MFMA_TFLOPS, "conv_mfma<3,1,2,*,false> (3x3 stride-1 convolutions, v_mfma_f32_32x32x2_f32)"
            peak_note = "dense fp32 MFMA peak"
        else:  # three fp16 MFMAs per algorithmic product: the ceiling for algorithmic FLOPs is 2500/3
            peak = PEAK_F16_MFMA_TFLOPS / 3.0
            kname = ("conv_k32<3,*> (3x3 stride-1 convolutions, 32-channel chunks, 3 x v_mfma_f32_16x16x32_f16 per product)"
                     if dom_key.endswith("k32") else
                     "conv_mfma<3,1,2,*,true> (3x3 stride-1 convolutions, 3 x v_mfma_f32_32x32x16_f16 per product)")
            peak_note = "dense fp16 MFMA peak 2500 TFLOP/s / 3 MFMAs per algorithmic product"
        traffic, traffic_src = pmc_traffic(precision, args.height, args.width)
        return {"bound": "mfma", "kernel": kname, "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "peak_note": peak_note, "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": round(dom.get("bytes", 0.0) / max(dom["launches"], 1)),
                "launches_per_p_frame": dom["launches"] // 2, "avg_launch_ms": round(dom["ms"] / max(dom["launches"], 1), 4),
                "measured": "HIP events around every launch, one GOP stream on the GPU (the kernel by itself)",
                "all_conv_tflops": round(all_flops / (all_ms * 1e-3) / 1e12, 2), "conv_ms_per_p_frame": round(all_ms / 2, 2),
                "power_note": "on all-zero operands (full 2.4 GHz clock, minimal power) the kernel takes 76 % of its random-data "
                              "time on the probe layer (0.338 vs 0.444 ms, profiles/r04_conv_k32_ablations_late.txt; round 3: 84 %): "
                              "on real data the board's power limit costs about a quarter, the rest is the kernel's phase "
                              "structure (stamps in profiles/r03_conv_k32_stamps.txt); DESIGN.md 4.1 item 9",
                "dominant_profile_key": dom_key,
                # the HBM-bound kernels of the path (warp, resamplers, dual prior, layout), per shape: HIP
                # events around every launch of the same two P pictures, algorithmic bytes (each operand once) / time,
                # against the HBM peak of MI355X_MICROARCH.md (8 TB/s; ~6.3 TB/s is what a streaming kernel reaches)
                "hbm_kernels": sorted(({**h, "frac_of_8_tb_s": None if h["tb_per_s"] is None else round(h["tb_per_s"] / PEAK_HBM_TBS, 3)}
                                       for h in hbm), key=lambda h: -h["algorithmic_bytes"] * h["launches"])}, views

    roofline, views = conv_roofline(i_net, p_net, args.precision)
    planes = picture_planes(p_net, views) if (rank == 0 and world == 1 and not args.no_cpu_baseline) else None
    # the same kernel while K GOP streams are in flight (what a rocprof trace of this command shows):
    # launches of the streams overlap, so each takes longer although together they finish sooner
    if K > 1:
        engines = [e_.p_net.engine() for e_ in cenc.encoders]
        for g in engines:
            g.profile = {}
        cenc.encode_gops([sq[:4] for sq in seqs], q_i, q_mv, q_y)
        both = [g.collect_profile().get(roofline["dominant_profile_key"], {"ms": 0.0, "launches": 0}) for g in engines]
        for g in engines:
            g.profile = None
        n_l = sum(b["launches"] for b in both)
        roofline["avg_launch_ms_with_all_gop_streams_in_flight"] = round(sum(b["ms"] for b in both) / max(n_l, 1), 4)

    import numpy as np

    out = {
        "metric": "encoded frames/sec at 1920x1080 GOP-32", "value": round(fps, 3), "unit": "frames/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.precision == "fp32" else "f32 (fp16x3 split-MFMA, fp32 accumulate)", "data": "synthetic",
        "config": {"workload": "1920x1080 GOP-32 synthetic sequence, single-rate encode (configs[1]); padded 1088x1920; "
                               f"a step codes {K} GOPs of 1 I + 31 P pictures per GPU, in flight together on {K} HIP "
                               "streams; random-init name-seeded weights",
                   "gop": args.gop, "height": args.height, "width": args.width, "precision": args.precision,
                   "parallelism": f"gop-sharded x{world} GPUs x{K} concurrent GOP streams per GPU",
                   "gops_per_step_per_gpu": K, "frames_per_step_per_gpu": K * args.gop,
                   "one_gop_stream_frames_per_s": one_stream,
                   "host_cpu_s_per_frame": round(host_cpu_s / (K * args.gop * args.steps), 5),
                   "host_cpu_cores_busy": round(host_cpu_s / dt_local, 2),
                   "host_threads": host_threads,
                   "hbm_bytes_reserved": int(torch.cuda.memory_reserved(dev)),
                   "hbm_bytes_workspaces": int(sum(e_.i_net.engine().bytes_reserved() + e_.p_net.engine().bytes_reserved()
                                                   for e_ in cenc.encoders)),
                   "host_note": "per RANK: CPU seconds of this process per coded frame over the timed region (all threads: "
                                "feeder, rANS coder), cores kept busy, threads alive, device memory held (caching allocator / "
                                "the engines' named workspaces); an 8-rank node needs 8 x these",
                   "payload_sha16_gop0": payload_sha,
                   "parity_vs_reference": parity_vs_reference(args.precision),
                   "device": torch.cuda.get_device_name(dev), "device_uuid": str(getattr(torch.cuda.get_device_properties(dev), "uuid", "")),
                   "bits_per_gop": int(bits), "bpp": round(bits / (args.gop * args.height * args.width), 4),
                   "psnr_db_gop_mean": round(float(psnr_fast.mean()), 4),
                   "psnr_note": "random-init weights: PSNR is a parity quantity here, not codec quality",
                   "fp16x3_range_status": int(saturation),
                   "fp16x3_range_note": "0 = no convolution output of ANY launch of this run (timed steps included: the guard is "
                                        "always on) exceeded |8188|, the magnitude split-fp16 operands can hold "
                                        "(DCVC_STATUS_ACT_SATURATED otherwise)"},
        "roofline": roofline,
    }
    parity_ok = True
    if world == 1 and args.precision != "fp32" and not args.no_parity_leg:
        # the same GOP once in exact-fp32 MFMA mode ("parity mode"): rate, bits and per-picture PSNR next to
        # the fast mode's, from this very run -- the fast mode's standing is what this object shows
        i32 = IntraNoAR(precision="fp32").to(dev).eval()
        p32 = DMC(precision="fp32").to(dev).eval()
        enc32 = GopEncoder(i32, p32, gop_size=args.gop)
        enc32.encode_gop(seq[:3], q_i, q_mv, q_y)  # warm-up: weight packing, buffers
        q32 = GopQuality(seq, args.height, args.width)
        dt32, bits32 = timed_region(lambda: enc32.encode_gop(seq, q_i, q_mv, q_y, on_recon=q32)[1], dev)
        psnr32 = q32.psnr()
        d_bpp = abs(bits32 - bits1) / max(bits32, 1)
        d_psnr = np.abs(psnr_fast - psnr32)
        d_psnr_gop = abs(float(psnr_fast.mean()) - float(psnr32.mean()))
        rel_psnr_gop = d_psnr_gop / max(abs(float(psnr32.mean())), 1.0)
        parity_ok = bool(d_bpp <= 1e-4 and rel_psnr_gop <= 1e-4 and saturation == 0)
        roof32, _ = conv_roofline(i32, p32, "fp32")
        out["parity_mode_fp32"] = {
            "value": round(args.gop / dt32, 3), "unit": "frames/s", "ms_per_step": round(dt32 * 1e3, 2), "bits_per_gop": int(bits32),
            "psnr_db_gop_mean": round(float(psnr32.mean()), 4),
            "fast_vs_fp32": {"bpp_rel_diff_gop": round(d_bpp, 8), "psnr_db_abs_diff_gop_mean": round(d_psnr_gop, 7),
                             "psnr_rel_diff_gop_mean": round(rel_psnr_gop, 8),
                             "psnr_db_abs_diff_max_over_pictures": round(float(d_psnr.max()), 6),
                             "tolerance": 1e-4, "within_tolerance": parity_ok,
                             "note": "GOP-level bpp and PSNR of the fp16x3 run against the exact-fp32 run of the same GOP "
                                     "(north_star: PSNR/bpp within 1e-4); a picture deep in the GOP may differ more when one "
                                     "symbol rounds the other way (printed as max over pictures)"},
            "roofline": roof32, "note": "same GOP (stream 0's sequence), one stream"}
    if world == 1 and not args.no_extra_workloads:
        # the other two workloads of this path, measured by the SAME command the driver runs (short versions
        # of --workload decode / --workload train), so that they are not builder-only numbers
        coded = [r[0] for r in cenc.encode_gops(seqs, q_i, q_mv, q_y)]
        cenc.decode_gops([c[:4] for c in coded], args.height, args.width)  # decoder-side buffers
        dtd, _ = timed_region(lambda: cenc.decode_gops(coded, args.height, args.width), dev)
        out["decode"] = {"value": round(K * args.gop / dtd, 3), "unit": "frames/s", "ms_per_step": round(dtd * 1e3, 2),
                         "workload": f"decode of the same {K} GOPs (reference bitstream: 3 / 6 host rANS round trips per I / P "
                                     "picture), one host thread and HIP stream per GOP; = bench.py --workload decode"}
        del coded
        # PNG -> .bin loop, I/O inside the timed region: K whole GOPs, on the encoders' own nets
        out["files"] = files_workload(args, dev, K * args.gop, nets=[(e_.i_net, e_.p_net) for e_ in cenc.encoders])
        for e_ in cenc.encoders:  # free the 1080p workspaces before the training model allocates its own
            e_.i_net.engine().release()
            e_.p_net.engine().release()
        torch.cuda.empty_cache()
        targs = argparse.Namespace(**vars(args))
        targs.steps, targs.warmup, targs.no_cpu_baseline = 5, 2, True
        tr = train_workload(targs, dev, rank, world)
        out["train"] = {"value": tr["value"], "unit": tr["unit"], "ms_per_step": tr["ms_per_step"], "steps": 5,
                        "workload": tr["config"]["workload"] + "; = bench.py --workload train", "roofline_frac": tr["roofline"]["frac"]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        ph, pw = (args.cpu_size if args.cpu_size else seq[0].shape[-2:])
        cores = host_cores()
        sec_all = cpu_baseline(int(ph), int(pw), cores)
        # the reference pins ONE thread while coding (video_coder.py:177): the same picture at full size on one thread
        # (~35 s; round 2 scaled a 1/16-area picture instead)
        sh, sw = int(ph), int(pw)
        sec_1 = cpu_baseline(sh, sw, 1)
        t_enc, t_dec, nbytes = cpu_rans_baseline(planes)
        out["cpu_baseline"] = {
            "value": round(1.0 / (sec_all + t_enc), 5), "unit": "frames/s", "cores": cores, "cpu_model": host_cpu_model(),
            "kind": "port",
            "sample": f"1 P picture {ph}x{pw}: networks of DMC.compress through oracle/dcvc_ref.py (torch-CPU fp32, {cores} threads) "
                      f"{sec_all:.1f} s + rANS encode of its 6 symbol planes through oracle/rans_ref.c (1 thread) {t_enc * 1e3:.0f} ms",
            "breakdown": {"nets_s_per_frame_all_cores": round(sec_all, 2), "nets_s_per_frame_1_thread": round(sec_1, 1),
                          "nets_1_thread_sample": f"{sh}x{sw} picture, measured at size",
                          "rans_encode_s_per_frame": round(t_enc, 4), "rans_decode_s_per_frame": round(t_dec, 4),
                          "rans_payload_bytes": nbytes, "rans_symbols": int(sum(p[0].size for p in planes)),
                          "end_to_end_fps_all_cores": round(1.0 / (sec_all + t_enc), 5),
                          "end_to_end_fps_1_thread": round(1.0 / (sec_1 + t_enc), 6)}}
    if grouped:
        # self-verification of a multi-GPU run: how many ranks the collective backend really joined (an all-reduce of
        # ones over the process group: RCCL when the backend is nccl) and what each rank delivered by its own clock
        on_gpu = dist.get_backend() == "nccl"
        one = torch.ones(1, dtype=torch.float64, device=dev if on_gpu else "cpu")
        dist.all_reduce(one)
        mine = torch.tensor([K * args.gop * args.steps / dt_local], dtype=torch.float64, device=dev if on_gpu else "cpu")
        every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(every, mine)
        out["config"]["dist_backend"] = args.dist_backend
        out["config"]["collective_ranks"] = int(round(float(one.item())))
        if on_gpu:
            out["config"]["rccl_ranks"] = int(round(float(one.item())))
        out["config"]["per_rank_frames_per_s"] = [round(float(t.item()), 3) for t in every]
    if rank == 0:
        print(json.dumps(out))
    if grouped:
        dist.destroy_process_group()
    if not parity_ok and args.strict_parity:
        sys.stderr.write("bench.py: the fp16x3 run is outside 1e-4 of the fp32 run (see parity_mode_fp32.fast_vs_fp32)\n")
        raise SystemExit(3)


if __name__ == "__main__":
    main()
