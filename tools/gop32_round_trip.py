import sys, os, torch
sys.path.insert(0, os.getcwd())
from bench import synth_sequence
from vcm_ts_amd.dmc import DMC
from vcm_ts_amd.intra import IntraNoAR
from vcm_ts_amd.pipeline import GopEncoder, pad_frame
dev = torch.device("cuda:0")
for prec in ("fp16x3", "fp32"):
    i, d = IntraNoAR(precision=prec).to(dev).eval(), DMC(precision=prec).to(dev).eval()
    enc = GopEncoder(i, d, gop_size=32)
    seq = [pad_frame(f) for f in synth_sequence(dev, 32, 1080, 1920, seed=3)]
    recs = []
    coded, bits, dpb = enc.encode_gop(seq, 1.0, 1.0, 1.0, on_recon=lambda t, r: recs.append(r.clone()))
    dec = enc.decode_gop(coded, 1080, 1920)
    ok = all(torch.equal(a, b) for a, b in zip(recs, dec))
    print(prec, "GOP-32 1080p encode->decode identical reconstructions:", ok, "bits", bits)
    assert ok
    i.engine().release(); d.engine().release(); del i, d, enc; torch.cuda.empty_cache()

# two GOPs in flight on two HIP streams (bench.py's default) against the same GOPs coded one after the other: same bytes?
from vcm_ts_amd.pipeline import ConcurrentGopEncoder
make = lambda: (IntraNoAR(precision="fp16x3").to(dev).eval(), DMC(precision="fp16x3").to(dev).eval())
cenc = ConcurrentGopEncoder(make, gop_size=32, streams=2)
seqs = [[pad_frame(f) for f in synth_sequence(dev, 32, 1080, 1920, seed=10 + k)] for k in range(2)]
both = cenc.encode_gops(seqs, 1.0, 1.0, 1.0)
for rep in range(2):
    again = cenc.encode_gops(seqs, 1.0, 1.0, 1.0)
    assert all(a[0] == b[0] for a, b in zip(both, again)), "two concurrent runs differ"
solo = [cenc.encoders[0].encode_gop(seqs[k], 1.0, 1.0, 1.0) for k in range(2)]
same = all(both[k][0] == solo[k][0] for k in range(2))
print("GOP-32 1080p: two GOPs on two streams (3 runs) vs one after the other: identical payloads:", same,
      "bytes", [sum(len(p) if isinstance(p, (bytes, bytearray)) else len(p[-1]) for p in both[k][0]) for k in range(2)])
assert same
