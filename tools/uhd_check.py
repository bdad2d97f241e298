"""Robustness probe: one I + two P pictures at 3840x2160 (padded 2176x3840), encode -> decode identity."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcm_ts_amd.dmc import DMC
from vcm_ts_amd.intra import IntraNoAR
from vcm_ts_amd.pipeline import GopEncoder, pad_frame
from bench import synth_sequence
dev = torch.device("cuda:0")
enc = GopEncoder(IntraNoAR(precision="fp16x3").to(dev).eval(), DMC(precision="fp16x3").to(dev).eval(), 32)
seq = [pad_frame(f) for f in synth_sequence(dev, 3, 2160, 3840, 0)]
print("padded", tuple(seq[0].shape))
enc.encode_gop(seq, 1.0, 1.0, 1.0)
torch.cuda.synchronize(); t0 = time.perf_counter()
coded, bits, dpb = enc.encode_gop(seq, 1.0, 1.0, 1.0)
torch.cuda.synchronize(); t1 = time.perf_counter()
last = dpb["ref_frame"].clone()
recs = enc.decode_gop(coded, 2160, 3840)
print("encode 3 pictures %.1f ms; decode identical: %s; bpp %.3f; HBM buffers %.1f GB" % (
    1e3 * (t1 - t0), torch.equal(recs[-1], last), bits / (3 * 2160 * 3840),
    (enc.p_net.engine().bytes_reserved() + enc.i_net.engine().bytes_reserved()) / 1e9))
