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
