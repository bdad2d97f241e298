"""Decoder-side timing at 1080p: decompress of I and P pictures (6 host rANS decodes interleaved
with the networks per P picture), and the reference-faithful encode_decode call."""
import os, sys, time, tempfile
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcm_ts_amd.dmc import DMC
from vcm_ts_amd.intra import IntraNoAR
from vcm_ts_amd.pipeline import GopEncoder, pad_frame
from bench import synth_sequence
prec = sys.argv[1] if len(sys.argv) > 1 else "fp16x3"
dev = torch.device("cuda:0")
i_net, p_net = IntraNoAR(precision=prec).to(dev).eval(), DMC(precision=prec).to(dev).eval()
enc = GopEncoder(i_net, p_net, 32)
seq = [pad_frame(f) for f in synth_sequence(dev, 8, 1080, 1920, 0)]
coded, bits, _ = enc.encode_gop(seq, 1.0, 1.0, 1.0)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    recs = enc.decode_gop(coded, 1080, 1920)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"decode 8 pictures: {1e3*(t1-t0):.1f} ms = {8/(t1-t0):.2f} frames/s")
with tempfile.TemporaryDirectory() as td:
    dpb = {"ref_frame": i_net.compress(seq[0], 1.0)["x_hat"], "ref_feature": None, "ref_y": None, "ref_mv_y": None}
    for t in range(1, 5):
        r = p_net.encode_decode(seq[t], dpb, os.path.join(td, "p.bin"), pic_width=1920, pic_height=1080, mv_y_q_scale=1.0, y_q_scale=1.0)
        dpb = r["dpb"]
        print(f"encode_decode P{t}: enc {1e3*r['encoding_time']:.1f} ms dec {1e3*r['decoding_time']:.1f} ms bits {r['bit']}")
