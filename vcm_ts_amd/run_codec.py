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
from .pipeline import GopEncoder, pad_frame


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


def save_torch_image(img: torch.Tensor, path, pool=None):
    """stream_helper.py:148-153.  pool: a ThreadPoolExecutor that takes the PNG encoding (the device -> host copy still
    happens here, while the tensor is valid)."""
    a = img.squeeze(0).permute(1, 2, 0).detach().cpu().numpy()
    if pool is None:
        _save_array(a, path)
    else:
        pool.submit(_save_array, a, path)


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
                  i_ckpt=None, p_ckpt=None, max_frames=None, coder="host", io_workers=8, nets=None):
    """Returns (bits per frame list, (height, width)).  coder="device": payloads in the opt-in GPU
    format (include/dcvc_hip_rans.h) inside the same .bin containers; decode_folder reads both.
    io_workers: host threads decoding PNGs ahead of the encoder (0: read in the encode loop as run_dcvc does).
    nets: (i_frame_net, p_frame_net) already on the device, instead of building them here."""
    os.makedirs(bin_dir, exist_ok=True)
    if recon_dir:
        os.makedirs(recon_dir, exist_ok=True)
    dev = torch.device(device)
    enc = GopEncoder(*(nets if nets is not None else _nets(dev, precision, i_ckpt, p_ckpt)), gop_size=gop, coder=coder)
    reader = PNGReader(frames_dir)
    size, bits = [None], []

    def frames():
        # uint8 pixels go to the device through a ring of pinned buffers on a copy stream of their own: a pageable
        # `.to(device)` would be a synchronous copy queued BEHIND the previous picture's kernels, i.e. the host could
        # never run ahead of the GPU (measured: 26 ms of every picture's 66 spent blocked in that call)
        n, ring, done, copy_stream = 0, [], [], torch.cuda.Stream(dev) if dev.type == "cuda" else None
        for rgb in reader.prefetching(workers=io_workers, depth=2 * io_workers, raw=True) if io_workers > 0 else reader.sequential():
            if max_frames is not None and n >= max_frames:
                return
            if rgb.dtype == np.uint8 and copy_stream is not None:
                if not ring:
                    ring = [torch.empty(rgb.shape, dtype=torch.uint8).pin_memory() for _ in range(3)]
                    done = [None] * len(ring)
                assert tuple(rgb.shape) == tuple(ring[0].shape), "all frames must have one size"
                k = n % len(ring)
                if done[k] is not None:
                    done[k].synchronize()  # the copy that last read this pinned buffer (three pictures ago)
                ring[k].numpy()[...] = rgb
                with torch.cuda.stream(copy_stream):
                    d = ring[k].to(dev, non_blocking=True)
                    done[k] = torch.cuda.Event()
                    done[k].record(copy_stream)
                cur = torch.cuda.current_stream(dev)
                cur.wait_event(done[k])
                d.record_stream(cur)
                x = u8_to_unit_float(d)
            elif rgb.dtype == np.uint8:
                x = u8_to_unit_float(torch.from_numpy(np.array(rgb)).to(dev))
            else:
                x = torch.from_numpy(rgb)[None].to(dev)
            if size[0] is None:
                size[0] = tuple(x.shape[-2:])
            assert tuple(x.shape[-2:]) == size[0], "all frames must have one size"
            n += 1
            yield pad_frame(x)

    def sink(kind, qidx, payload, t):
        path = os.path.join(bin_dir, f"im{str(t + 1).zfill(5)}.bin")
        if kind == "I":
            S.encode_i(size[0][0], size[0][1], qidx[0], payload, path)
        else:
            S.encode_p(payload, qidx[0], qidx[1], path)
        bits.append(S.filesize(path) * 8)

    from concurrent.futures import ThreadPoolExecutor

    def on_recon(t, ref_frame):
        if recon_dir:
            h, w = size[0]
            save_torch_image(ref_frame[..., :h, :w], os.path.join(recon_dir, f"im{str(t + 1).zfill(5)}.png"), savers)

    with ThreadPoolExecutor(max_workers=max(io_workers, 1)) as savers, torch.no_grad():
        enc.encode_gop(frames(), q[0], q[1], q[2], sink=sink, on_recon=on_recon)
    return bits, size[0]


def decode_folder(bin_dir, recon_dir, height, width, gop=32, device="cuda:0", precision=None, i_ckpt=None, p_ckpt=None,
                  io_workers=8):
    from concurrent.futures import ThreadPoolExecutor

    os.makedirs(recon_dir, exist_ok=True)
    dev = torch.device(device)
    i_net, p_net = _nets(dev, precision, i_ckpt, p_ckpt)
    i_net.update()
    p_net.update()
    t, dpb = 0, None
    with ThreadPoolExecutor(max_workers=max(io_workers, 1)) as savers, torch.no_grad():
        while True:
            path = os.path.join(bin_dir, f"im{str(t + 1).zfill(5)}.bin")
            if not os.path.exists(path):
                break
            if t % gop == 0:
                h, w, qi, payload = S.decode_i(path)
                assert (h, w) == (height, width)
                x_hat = i_net.decompress(payload, h, w, qi / 100)["x_hat"]
                dpb = {"ref_frame": x_hat, "ref_feature": None, "ref_y": None, "ref_mv_y": None}
            else:
                qmv, qy, payload = S.decode_p(path)
                dpb = p_net.decompress(dpb, payload, height, width, qmv / 100, qy / 100)["dpb"]
            save_torch_image(dpb["ref_frame"][..., :height, :width], os.path.join(recon_dir, f"im{str(t + 1).zfill(5)}.png"), savers)
            t += 1
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
                                   coder=a.coder, io_workers=a.io_workers)
        print(f"{len(bits)} pictures, {size[0]}x{size[1]}, {sum(bits)} bits, {sum(bits) / (len(bits) * size[0] * size[1]):.4f} bpp")
    else:
        n = decode_folder(a.bins, a.recon, a.height, a.width, a.gop, a.device, a.precision, a.i_ckpt, a.p_ckpt, io_workers=a.io_workers)
        print(f"{n} pictures decoded")


if __name__ == "__main__":
    main()
