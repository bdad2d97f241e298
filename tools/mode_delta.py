"""fp16x3 (fast) vs fp32 (parity) convolution modes on the same GPU and the same 1080p pictures:
per-picture payload size and PSNR of the reconstruction, as SURVEY section 7 asks of a fast mode."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcm_ts_amd.dmc import DMC
from vcm_ts_amd.intra import IntraNoAR
from vcm_ts_amd.pipeline import GopEncoder, pad_frame
from bench import synth_sequence
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
seq = [pad_frame(f) for f in synth_sequence(dev, n, 1080, 1920, 0)]
res = {}
for prec in ("fp32", "fp16x3"):
    enc = GopEncoder(IntraNoAR(precision=prec).to(dev).eval(), DMC(precision=prec).to(dev).eval(), 32)
    rows = []
    def on_recon(t, rec, rows=rows):
        mse = torch.mean((rec[..., :1080, :] - seq[t][..., :1080, :]) ** 2).item()
        rows.append(10 * np.log10(1.0 / mse))
    coded, bits, _ = enc.encode_gop(seq, 1.0, 1.0, 1.0, on_recon=on_recon)
    res[prec] = ([len(c[2]) * 8 for c in coded], rows)
    del enc; torch.cuda.empty_cache()
print("pic  bits_fp32  bits_fp16x3  rel_diff   psnr_fp32  psnr_fp16x3  |dPSNR| dB")
for t in range(n):
    b0, b1 = res["fp32"][0][t], res["fp16x3"][0][t]
    p0, p1 = res["fp32"][1][t], res["fp16x3"][1][t]
    print(f"{t:3d} {b0:10d} {b1:11d}  {abs(b1-b0)/b0:9.2e} {p0:10.5f} {p1:11.5f}  {abs(p1-p0):9.2e}")
tb0, tb1 = sum(res["fp32"][0]), sum(res["fp16x3"][0])
print(f"total bits rel diff {abs(tb1-tb0)/tb0:.2e}; max |dPSNR| {max(abs(a-b) for a,b in zip(res['fp32'][1], res['fp16x3'][1])):.2e} dB")
