"""Folder-level encode / decode: the base-layer loop of the reference's
``video_coder.run_dcvc`` (/root/reference/video_coder.py:80-155) and its PNG conventions
(DCVC_HEM/src/utils/png_reader.py:10-46, stream_helper.py:148-153) on the MI355X path.

    python -m vcm_ts_amd.run_codec encode --frames DIR --bins DIR [--recon DIR] [--gop 32] [--q 1.0 1.0 1.0]
    python -m vcm_ts_amd.run_codec decode --bins DIR --recon DIR --height H --width W

Frames are ``im1.png`` / ``im00001.png`` ...; coded pictures are ``im00001.bin`` ... in the
reference's `.bin` format (an I picture every `gop` frames).  Unlike run_dcvc the encoder does not
run the decoder: its own reconstruction is bit-identical to what `decode` produces.
"""
from __future__ import annotations

import argparse
import os

import numpy as np
import torch
from PIL import Image

from . import stream as S
from .pipeline import pad_frame


class PNGReader:
    """im1.png or im00001.png naming, RGB float32 in [0, 1], (3, H, W)."""

    def __init__(self, folder):
        names = os.listdir(folder)
        if "im1.png" in names:
            self.width = 1
        elif "im00001.png" in names:
            self.width = 5
        else:
            raise ValueError("unknown image naming convention; expected im1.png or im00001.png")
        self.folder, self.index = folder, 1

    def path_of(self, index):
        return os.path.join(self.folder, f"im{str(index).zfill(self.width)}.png")

    @staticmethod
    def load_u8(path):
        """(H, W, 3) uint8"""
        return np.asarray(Image.open(path).convert("RGB"))

    @staticmethod
    def load(path):
        return PNGReader.load_u8(path).astype("float32").transpose(2, 0, 1) / 255.0

    def read_one_frame(self):
        path = self.path_of(self.index)
        if not os.path.exists(path):
            return None
        self.index += 1
        return self.load(path)

    def sequential(self):
        while (f := self.read_one_frame()) is not None:
            yield f

    def prefetching(self, workers=6, depth=12, raw=False):
        """Iterate over the remaining frames with `workers` threads decoding up to `depth` PNGs ahead (PIL and numpy
        release the GIL while they decode / convert; a 1920x1080 PNG costs 40-140 ms on one host thread, more than the
        GPU needs to code the picture).  Same arrays in the same order as read_one_frame(); raw=True: the (H, W, 3)
        uint8 pixels instead (u8_to_unit_float turns them into the same floats on the device)."""
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(max_workers=workers) as pool:
            pending = deque()
            while True:
                while len(pending) < depth:
                    path = self.path_of(self.index)
                    if not os.path.exists(path):
                        break
                    self.index += 1
                    pending.append(pool.submit(self.load_u8 if raw else self.load, path))
                if not pending:
                    return
                yield pending.popleft().result()


_LUT = {}


def u8_to_unit_float(u8: torch.Tensor):
    """(H, W, 3) uint8 on a device -> (1, 3, H, W) float32 with exactly the reference's values (png_reader.py: uint8 ->
    float32, / 255.0 on the host): a 256-entry table of those host-computed quotients, so that no device division
    (which torch may turn into a multiplication by a reciprocal) is involved.  4x fewer bytes cross PCIe."""
    lut = _LUT.get(u8.device)
    if lut is None:
        lut = _LUT[u8.device] = torch.from_numpy(np.arange(256).astype("float32") / 255.0).to(u8.device)
    return lut[u8.to(torch.int64)].permute(2, 0, 1)[None].contiguous()


def _save_array(a, path):
    Image.fromarray(np.clip(np.rint(a * 255), 0, 255).astype(np.uint8)).save(path)


class PNGWriters:
    """A bounded pool of PNG-encoding threads: at most 2 x workers pictures (25 MB of float32 each at 1080p) wait in
    host memory, and a failed write (disk full, bad path) surfaces in the caller -- at the next save or at close() --
    instead of being dropped with its future.  workers == 0: write inline, as the reference does."""

    def __init__(self, workers):
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor

        self.pool = ThreadPoolExecutor(max_workers=workers) if workers > 0 else None
        self.pending, self.limit = deque(), 2 * max(workers, 1)

    def submit(self, a, path):
        if self.pool is None:
            return _save_array(a, path)
        while len(self.pending) >= self.limit:
            self.pending.popleft().result()  # re-raises a writer's exception
        self.pending.append(self.pool.submit(_save_array, a, path))

    def close(self):
        try:
            while self.pending:
                self.pending.popleft().result()
        finally:
            if self.pool is not None:
                self.pool.shutdown(wait=True)

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        if et is None:
            self.close()
        elif self.pool is not None:  # already failing: do not mask the first error with a writer's
            self.pool.shutdown(wait=True, cancel_futures=True)


def save_torch_image(img: torch.Tensor, path, writers: PNGWriters = None):
    """stream_helper.py:148-153.  writers: a PNGWriters pool that takes the PNG encoding (the device -> host copy still
    happens here, while the tensor is valid)."""
    a = img.squeeze(0).permute(1, 2, 0).detach().cpu().numpy()
    if writers is None:
        _save_array(a, path)
    else:
        writers.submit(a, path)


def _nets(device, precision, i_ckpt=None, p_ckpt=None):
    from .dmc import DMC
    from .intra import IntraNoAR

    i_net, p_net = IntraNoAR(precision=precision), DMC(precision=precision)
    if i_ckpt:
        i_net.load_state_dict(S.get_state_dict(i_ckpt), strict=False)
    if p_ckpt:
        p_net.load_state_dict(S.get_state_dict(p_ckpt), strict=False)
    return i_net.to(device).eval(), p_net.to(device).eval()


def interpolate_log(min_val, max_val, num, decending=True):
    """DCVC_HEM/src/utils/common.py:23-31: `num` values spaced evenly in log(q) between the two anchors."""
    import numpy as np

    assert max_val > min_val > 0
    lo, hi = np.log(min_val), np.log(max_val)
    return np.exp(np.linspace(hi, lo, num) if decending else np.linspace(lo, hi, num))


def rate_point_q_scales(i_q_scales, y_q_scales, mv_y_q_scales, rate_count, quality):
    """video_coder.py:181-197: the (I, mv_y, y) q-scales of rate point `quality` out of `rate_count` points interpolated
    in log space between the first (coarsest) and the last (finest) anchor a model was trained with.  The arguments
    are the flattened q_scale tensors of the checkpoints (IntraNoAR.get_q_scales_from_ckpt / DMC.get_q_scales_from_ckpt)
    or of the live modules.  Returns (q_i, q_mv_y, q_y) in the order encode_folder takes them."""
    if not 0 <= quality < rate_count:
        raise ValueError(f"quality must be in [0, {rate_count})")
    pick = lambda qs: float(interpolate_log(float(qs[-1]), float(qs[0]), rate_count)[quality])
    return pick(i_q_scales), pick(mv_y_q_scales), pick(y_q_scales)


def encode_folder(frames_dir, bin_dir, recon_dir=None, gop=32, q=(1.0, 1.0, 1.0), device="cuda:0", precision=None,
                  i_ckpt=None, p_ckpt=None, max_frames=None, coder="host", io_workers=8, nets=None, gop_streams=1):
    """Returns (bits per frame list, (height, width)).  coder="device": payloads in the opt-in GPU
    format (include/dcvc_hip_rans.h) inside the same .bin containers; decode_folder reads both.
    io_workers: host threads decoding PNGs ahead of the encoder (0: read in the encode loop as run_dcvc does).
    nets: (i_frame_net, p_frame_net) already on the device -- or a list of such pairs, one per GOP stream -- instead
    of building them here.
    gop_streams (round 4): GOPs of the folder in flight together on the GPU (pipeline.ConcurrentGopEncoder: own codec
    instances, DPB and HIP stream each; stream k codes GOPs k, k + K, k + 2K, ...), fed by ONE pool of PNG-decoding
    threads.  GOPs are independent (every GOP starts from an I picture, video_coder.py:122-130), so the .bin files are
    byte-identical to the one-stream loop's (tests/test_gpu_codec.py); each stream holds its own workspace (~33 GB at
    1088x1920)."""
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor

    from .pipeline import ConcurrentGopEncoder

    os.makedirs(bin_dir, exist_ok=True)
    if recon_dir:
        os.makedirs(recon_dir, exist_ok=True)
    dev = torch.device(device)
    reader = PNGReader(frames_dir)
    n_frames = 0
    while os.path.exists(reader.path_of(n_frames + 1)) and (max_frames is None or n_frames < max_frames):
        n_frames += 1
    n_gops = (n_frames + gop - 1) // gop
    K = max(1, min(int(gop_streams), n_gops))
    pairs = [nets] if (nets is not None and not isinstance(nets, list)) else list(nets or [])
    made = iter(pairs[:K] + [None] * K)
    cenc = ConcurrentGopEncoder(lambda: next(made) or _nets(dev, precision, i_ckpt, p_ckpt), gop_size=gop, streams=K, coder=coder)
    size, bits = [None], {}
    pool = ThreadPoolExecutor(max_workers=io_workers) if io_workers > 0 else None

    def global_index(k, t):  # picture t of stream k's sequence -> 0-based frame number in the folder
        return ((t // gop) * K + k) * gop + t % gop

    def raw_frames(k):
        """Stream k's pictures in its coding order as (H, W, 3) uint8 arrays, decoded up to `depth` ahead by the shared pool."""
        order = [global_index(k, t) for t in range(((n_gops - k + K - 1) // K) * gop)]
        order = [g for g in order if g < n_frames]
        if pool is None:
            for g in order:
                yield reader.load_u8(reader.path_of(g + 1))
            return
        depth, pending, nxt = max(2, 2 * io_workers // K), deque(), 0
        while nxt < len(order) or pending:
            while nxt < len(order) and len(pending) < depth:
                pending.append(pool.submit(reader.load_u8, reader.path_of(order[nxt] + 1)))
                nxt += 1
            yield pending.popleft().result()

    def frames(k):
        # uint8 pixels go to the device through a ring of pinned buffers on a copy stream of their own: a pageable
        # `.to(device)` would be a synchronous copy queued BEHIND the previous picture's kernels, i.e. the host could
        # never run ahead of the GPU (measured: 26 ms of every picture's 66 spent blocked in that call)
        n, ring, done, copy_stream = 0, [], [], torch.cuda.Stream(dev) if dev.type == "cuda" else None
        for rgb in raw_frames(k):
            if copy_stream is not None:
                if not ring:
                    ring = [torch.empty(rgb.shape, dtype=torch.uint8).pin_memory() for _ in range(3)]
                    done = [None] * len(ring)
                assert tuple(rgb.shape) == tuple(ring[0].shape), "all frames must have one size"
                j = n % len(ring)
                if done[j] is not None:
                    done[j].synchronize()  # the copy that last read this pinned buffer (three pictures ago)
                ring[j].numpy()[...] = rgb
                with torch.cuda.stream(copy_stream):
                    d = ring[j].to(dev, non_blocking=True)
                    done[j] = torch.cuda.Event()
                    done[j].record(copy_stream)
                cur = torch.cuda.current_stream(dev)  # (this GOP stream's: ConcurrentGopEncoder pulls frames inside it)
                cur.wait_event(done[j])
                d.record_stream(cur)
                x = u8_to_unit_float(d)
            else:
                x = u8_to_unit_float(torch.from_numpy(np.array(rgb)).to(dev))
            if size[0] is None:
                size[0] = tuple(x.shape[-2:])
            assert tuple(x.shape[-2:]) == size[0], "all frames must have one size"
            n += 1
            yield pad_frame(x)

    def sink_of(k):
        def sink(kind, qidx, payload, t):
            g = global_index(k, t)
            path = os.path.join(bin_dir, f"im{str(g + 1).zfill(5)}.bin")
            if kind == "I":
                S.encode_i(size[0][0], size[0][1], qidx[0], payload, path)
            else:
                S.encode_p(payload, qidx[0], qidx[1], path)
            bits[g] = S.filesize(path) * 8

        return sink

    def recon_of(k):
        def on_recon(t, ref_frame):
            h, w = size[0]
            save_torch_image(ref_frame[..., :h, :w], os.path.join(recon_dir, f"im{str(global_index(k, t) + 1).zfill(5)}.png"), savers)

        return on_recon if recon_dir else None

    # (GopEncoder reads the split-fp16 range guard once per GOP and raises lib.KernelError: no .bin of a clamped GOP
    # is reported as a success)
    try:
        with PNGWriters(io_workers) as savers, torch.no_grad():
            cenc.encode_gops([frames(k) for k in range(K)], q[0], q[1], q[2], sinks=[sink_of(k) for k in range(K)],
                             on_recons=[recon_of(k) for k in range(K)])
    finally:
        if pool is not None:
            pool.shutdown(wait=True, cancel_futures=True)
    return [bits[g] for g in sorted(bits)], size[0]


def decode_folder(bin_dir, recon_dir, height, width, gop=32, device="cuda:0", precision=None, i_ckpt=None, p_ckpt=None,
                  io_workers=8):
    os.makedirs(recon_dir, exist_ok=True)
    dev = torch.device(device)
    i_net, p_net = _nets(dev, precision, i_ckpt, p_ckpt)
    i_net.update()
    p_net.update()
    t, dpb = 0, None

    def range_guard():  # once per GOP: raises lib.KernelError if a split-fp16 kernel clamped an activation
        i_net.engine().check_status()
        p_net.engine().check_status()

    with PNGWriters(io_workers) as savers, torch.no_grad():
        while True:
            path = os.path.join(bin_dir, f"im{str(t + 1).zfill(5)}.bin")
            if not os.path.exists(path):
                break
            if t % gop == 0:
                if t:
                    range_guard()
                h, w, qi, payload = S.decode_i(path)
                assert (h, w) == (height, width)
                x_hat = i_net.decompress(payload, h, w, qi / 100, check_range=False)["x_hat"]
                dpb = {"ref_frame": x_hat, "ref_feature": None, "ref_y": None, "ref_mv_y": None}
            else:
                qmv, qy, payload = S.decode_p(path)
                dpb = p_net.decompress(dpb, payload, height, width, qmv / 100, qy / 100, check_range=False)["dpb"]
            save_torch_image(dpb["ref_frame"][..., :height, :width], os.path.join(recon_dir, f"im{str(t + 1).zfill(5)}.png"), savers)
            t += 1
        if t:
            range_guard()
    return t


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    sub = ap.add_subparsers(dest="cmd", required=True)
    e = sub.add_parser("encode")
    e.add_argument("--frames", required=True)
    e.add_argument("--bins", required=True)
    e.add_argument("--recon")
    e.add_argument("--q", type=float, nargs=3, default=None, metavar=("I", "MV_Y", "Y"),
                   help="explicit q-scales (default 1 1 1 when no rate point is selected)")
    e.add_argument("--rate-count", type=int, default=None,
                   help="with --quality: the reference's rate-point selection (video_coder.py RATE_COUNT / QUALITY): "
                        "q-scales interpolated in log space between the anchors stored in the checkpoints")
    e.add_argument("--quality", type=int, default=None)
    e.add_argument("--gop-streams", type=int, default=1,
                   help="GOPs of the folder in flight together on the GPU (same .bin bytes; ~33 GB of workspace each at 1080p)")
    e.add_argument("--coder", default="host", choices=["host", "device"],
                   help="host: the reference's bitstream (default); device: opt-in GPU entropy coder, own format")
    d = sub.add_parser("decode")
    d.add_argument("--bins", required=True)
    d.add_argument("--recon", required=True)
    d.add_argument("--height", type=int, required=True)
    d.add_argument("--width", type=int, required=True)
    for p in (e, d):
        p.add_argument("--gop", type=int, default=32)
        p.add_argument("--io-workers", type=int, default=8, help="host threads for PNG decoding / encoding (0: inline)")
        p.add_argument("--device", default="cuda:0")
        p.add_argument("--precision", default=None, choices=["fp32", "fp16x3"])
        p.add_argument("--i-ckpt")
        p.add_argument("--p-ckpt")
    a = ap.parse_args()
    if a.cmd == "encode":
        if (a.rate_count is None) != (a.quality is None) or (a.q is not None and a.rate_count is not None):
            ap.error("give either --q, or --rate-count together with --quality")
        q = tuple(a.q) if a.q is not None else (1.0, 1.0, 1.0)
        if a.rate_count is not None:
            from .dmc import DMC
            from .intra import IntraNoAR
            from .params import dmc_spec, intra_spec, seeded_state_dict

            if a.i_ckpt and a.p_ckpt:
                i_qs = IntraNoAR.get_q_scales_from_ckpt(a.i_ckpt)
                y_qs, mv_qs = DMC.get_q_scales_from_ckpt(a.p_ckpt)
            else:  # no checkpoint: the anchors of the name-seeded synthetic weights the nets are built with
                i_qs = seeded_state_dict(intra_spec())["q_scale"].reshape(-1)
                sd = seeded_state_dict(dmc_spec())
                y_qs, mv_qs = sd["y_q_scale"].reshape(-1), sd["mv_y_q_scale"].reshape(-1)
            q = rate_point_q_scales(i_qs, y_qs, mv_qs, a.rate_count, a.quality)
            print(f"rate point {a.quality} of {a.rate_count}: q_i {q[0]:.4f}  q_mv_y {q[1]:.4f}  q_y {q[2]:.4f}")
        bits, size = encode_folder(a.frames, a.bins, a.recon, a.gop, q, a.device, a.precision, a.i_ckpt, a.p_ckpt,
                                   coder=a.coder, io_workers=a.io_workers, gop_streams=a.gop_streams)
        print(f"{len(bits)} pictures, {size[0]}x{size[1]}, {sum(bits)} bits, {sum(bits) / (len(bits) * size[0] * size[1]):.4f} bpp")
    else:
        n = decode_folder(a.bins, a.recon, a.height, a.width, a.gop, a.device, a.precision, a.i_ckpt, a.p_ckpt, io_workers=a.io_workers)
        print(f"{n} pictures decoded")


if __name__ == "__main__":
    main()
