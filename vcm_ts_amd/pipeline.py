"""GOP encode loop over the two codecs: the hot loop of the reference's
``video_coder.run_dcvc`` (/root/reference/video_coder.py:80-155) with the redundant decode
pass removed.

The reference calls ``encode_decode`` per frame, i.e. compress + file write + file read +
decompress, because its ``compress`` hides the encoder-side DPB behind a misspelt key
(video_model.py:344).  Here ``compress`` returns the DPB it already computed (clamped like the
decoder's, so both sides hold bit-identical references), and a GOP is a strict chain of
compress calls.  GOPs are independent (each starts from an I picture and an empty DPB), so
several GOPs shard across GPUs / processes with no exchange (``shard_gops``).
"""
from __future__ import annotations

import torch

from . import stream as S


def pad_frame(x: torch.Tensor):
    """Pad right/bottom to multiples of 64 with zeros (video_coder.py:111-117)."""
    h, w = x.shape[-2:]
    l, r, t, b = S.get_padding_size(h, w)
    return torch.nn.functional.pad(x, (l, r, t, b), mode="constant", value=0)


def shard_gops(n_gops: int, rank: int, world: int):
    """GOP g -> rank g mod world: no data-path collective is needed."""
    return [g for g in range(n_gops) if g % world == rank]


def timed_region(fn, device=None, with_local=False):
    """Run fn() between barriers and return the slowest rank's wall time in seconds (what
    bench.py reports): barrier + device sync on both sides, MAX over ranks.  with_local: also this
    rank's own time, as a third value."""
    import time

    import torch.distributed as dist

    multi = dist.is_available() and dist.is_initialized()  # also a one-rank group: same barrier + reduction path

    def fence():
        if multi:
            dist.barrier()
        if device is not None and torch.device(device).type == "cuda":
            torch.cuda.synchronize(device)

    fence()
    t0 = time.time()
    result = fn()
    fence()
    dt = local = time.time() - t0
    if multi:
        on_gpu = device is not None and dist.get_backend() == "nccl"
        t = torch.tensor([dt], dtype=torch.float64, device=device if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return (dt, result, local) if with_local else (dt, result)


class GopEncoder:
    def __init__(self, i_frame_net, p_frame_net, gop_size=32, coder="host", graphs=False):
        """coder="device": payloads in the opt-in GPU format of include/dcvc_hip_rans.h (symbol planes
        never leave the device; decode_gop recognises them by their magic).  graphs=True: P pictures
        are replayed as captured hipGraphs (same results; for picture sizes where ~235 launches per
        picture make the host the bottleneck)."""
        assert coder in ("host", "device") and not (graphs and coder != "host")
        self.i_net, self.p_net, self.gop, self.coder, self.graphs = i_frame_net, p_frame_net, int(gop_size), coder, graphs
        self.i_net.update()
        self.p_net.update()

    def encode_gop(self, frames, q_i, q_mv_y, q_y, sink=None, on_recon=None):
        """frames: iterable of padded (1, 3, H, W) device tensors, the first coded as an I
        picture.  Returns (list of payload bytes with their headers' q indexes, total bits of
        the payloads + headers, final DPB).  `sink(kind, q_indexes, payload, t)` may persist the
        coded pictures; `on_recon(t, ref_frame)` sees each reconstruction while it is still valid."""
        res = {}
        for _ in self.encode_steps(frames, q_i, q_mv_y, q_y, res, sink=sink, on_recon=on_recon):
            pass
        return res["coded"], res["bits"], res["dpb"]

    def encode_steps(self, frames, q_i, q_mv_y, q_y, res, sink=None, on_recon=None):
        """encode_gop as a generator that yields after every picture it has enqueued, so that several
        encoders can be interleaved by one host thread (ConcurrentGopEncoder).  Fills `res` with
        "coded", "bits", "dpb" when exhausted."""
        q_i, qi_idx = S.get_rounded_q(q_i)
        q_mv_y, qmv_idx = S.get_rounded_q(q_mv_y)
        q_y, qy_idx = S.get_rounded_q(q_y)
        out, bits, dpb, prev = [], 0, None, None

        def retire(item):  # host half of a picture: wait for its planes, rANS-code them
            nonlocal bits
            kind, q, pending, t, guards = item
            payload = pending.finish()
            for check in guards:  # range guard of the split-fp16 kernels: a GOP whose activations were
                check()           # clamped must not be written out as if it were fine -- raises lib.KernelError
            out.append((kind, q, payload))
            bits += (len(payload) + (14 if kind == "I" else 8)) * 8  # >IIHI / >HHI headers
            if sink is not None:
                sink(kind, q, payload, t)

        # Software pipeline: the kernels of picture t are enqueued (its symbol planes follow them
        # to pinned host memory asynchronously) BEFORE picture t-1 is entropy-coded on the host,
        # so the GPU works on t while the CPU codes t-1.  The DPB never leaves the device.
        for t, x in enumerate(frames):
            if t % self.gop == 0:
                r = self.i_net.compress(x, q_i, defer=True, coder=self.coder, check_range=False)
                dpb = {"ref_frame": r["x_hat"], "ref_feature": None, "ref_y": None, "ref_mv_y": None}
                item = ("I", (qi_idx,), r["pending"], t)
            else:
                r = self.p_net.compress(x, dpb, q_mv_y, q_y, defer=True, coder=self.coder, graph=self.graphs, check_range=False)
                dpb = r["dpb"]
                item = ("P", (qmv_idx, qy_idx), r["pending"], t)
            # once per GOP (behind its last picture; a trailing partial GOP is covered after the loop): the status
            # word of both engines, read asynchronously -- the copy rides behind the picture's kernels and is looked
            # at when that picture is retired, so the host never drains the GPU for it
            guards = ()
            if t % self.gop == self.gop - 1:
                guards = tuple(c for c in (self.i_net.engine().status_snapshot(), self.p_net.engine().status_snapshot()) if c)
            item = item + (guards,)
            if on_recon is not None:  # reconstruction == what the decoder will produce (clamped)
                on_recon(t, dpb["ref_frame"])
            if prev is not None:
                retire(prev)
            prev = item
            yield t
        if prev is not None:
            retire(prev)
            if not prev[4]:  # a trailing partial GOP: its pictures have not been looked at yet (a synchronising read;
                self.i_net.engine().check_status()  # after a whole GOP the snapshot above has already covered everything
                self.p_net.engine().check_status()  # and nothing is drained here)
        res.update(coded=out, bits=bits, dpb=dpb)

    def decode_gop(self, coded, height, width):
        """Inverse of encode_gop (the reference decoder path): returns the list of x_hat."""
        from .entropy import DRANS_MAGIC

        recs, dpb, deferred = [], None, set()
        for kind, q, payload in coded:
            # device-format pictures need no host round trip: enqueue them all, read the status once
            dev_fmt = payload[:4] == DRANS_MAGIC
            net = self.i_net if kind == "I" else self.p_net
            if dev_fmt:
                deferred.add(net)
            if kind == "I":
                x_hat = net.decompress(payload, height, width, q[0] / 100, defer_check=dev_fmt, check_range=False)["x_hat"]
                dpb = {"ref_frame": x_hat, "ref_feature": None, "ref_y": None, "ref_mv_y": None}
            else:
                dpb = net.decompress(dpb, payload, height, width, q[0] / 100, q[1] / 100, defer_check=dev_fmt,
                                     check_range=False)["dpb"]
            recs.append(dpb["ref_frame"].clone())
        for net in deferred:
            net.device_coder().check()
        self.i_net.engine().check_status()  # range guard of the split-fp16 kernels, once per call
        self.p_net.engine().check_status()
        return recs


class ConcurrentGopEncoder:
    """Several GOPs of a sequence in flight on ONE GPU: each GOP has its own codec instances (its
    own DPB and workspace) and its own HIP stream, and one host thread feeds them round-robin, a
    picture at a time.  GOPs are independent, so this is GOP sharding applied inside a
    GPU: while one GOP runs its 1/16- and 1/64-resolution stages (tens of small kernels that leave
    most of the 256 CUs idle) the other GOP's full-resolution convolutions fill the chip.  Payloads
    are byte-identical to sequential encoding; two streams measured +13 % frames/s at 1080p, a third
    adds nothing.  Every stream holds a full workspace (about 33 GB at 1088x1920, 130 GB at
    2176x3840): use one stream for 4K pictures."""

    def __init__(self, make_nets, gop_size=32, streams=2, coder="host"):
        """make_nets() -> (i_frame_net, p_frame_net) on the target device, called once per stream."""
        self.encoders = [GopEncoder(*make_nets(), gop_size=gop_size, coder=coder) for _ in range(int(streams))]
        dev = self.encoders[0].p_net.device
        self.device = dev
        self.streams = [torch.cuda.Stream(dev) for _ in self.encoders]

    def encode_gops(self, sequences, q_i, q_mv_y, q_y, sinks=None, on_recons=None):
        """sequences: up to `streams` iterables of padded pictures (one sequence of whole GOPs each; an iterable is
        pulled INSIDE its stream, so a generator may upload its pictures there).  Returns a list of
        (coded, bits, dpb) in the same order.  sinks / on_recons: per-sequence callbacks of GopEncoder.encode_gop."""
        assert len(sequences) <= len(self.encoders)
        cur = torch.cuda.current_stream(self.device)
        results = [{} for _ in sequences]
        gens = []
        for k, seq in enumerate(sequences):
            self.streams[k].wait_stream(cur)  # the pictures were produced on the caller's stream
            gens.append(self.encoders[k].encode_steps(seq, q_i, q_mv_y, q_y, results[k], sink=sinks[k] if sinks else None,
                                                      on_recon=on_recons[k] if on_recons else None))
        live = list(range(len(gens)))
        while live:
            for k in list(live):
                with torch.cuda.stream(self.streams[k]):
                    try:
                        next(gens[k])
                    except StopIteration:
                        live.remove(k)
        for s in self.streams[: len(sequences)]:
            cur.wait_stream(s)
        return [(r["coded"], r["bits"], r["dpb"]) for r in results]

    def decode_gops(self, coded_gops, height, width):
        """Decode up to `streams` GOPs at once: one host thread and one HIP stream per GOP (the
        reference bitstream forces six host round trips per P picture; while one GOP's thread waits
        for its symbols the other GOP's kernels run).  Returns one list of reconstructions per GOP."""
        import threading

        assert len(coded_gops) <= len(self.encoders)
        cur = torch.cuda.current_stream(self.device)
        out, errs = [None] * len(coded_gops), []

        def work(k):
            try:
                with torch.cuda.stream(self.streams[k]):
                    out[k] = self.encoders[k].decode_gop(coded_gops[k], height, width)
            except BaseException as ex:  # re-raised in the caller's thread
                errs.append(ex)

        for k in range(len(coded_gops)):
            self.streams[k].wait_stream(cur)
        threads = [threading.Thread(target=work, args=(k,)) for k in range(len(coded_gops))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errs:
            raise errs[0]
        for s in self.streams[: len(coded_gops)]:
            cur.wait_stream(s)
        return out
