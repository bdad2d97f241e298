import ctypes, os, torch
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libtr_read.so"))
out = torch.zeros(64 * 8, device="cuda:0")
L.probe(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
o = out.cpu().view(64, 8)
ok = True
for l in range(64):
    r, h = l & 31, l >> 5
    want = [(8 * h + j) * 100 + r for j in range(8)]
    got = [int(v) for v in o[l].tolist()]
    if got != want:
        ok = False
        print("lane", l, "got", got, "want", want)
print("tr_read operand map as expected:", ok)
