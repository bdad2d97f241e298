"""Host profile of run_codec.encode_folder on a folder of synthetic 1080p PNGs (where the file loop's time goes).
usage: python3 tools/py_profile_files.py [n_frames] [io_workers]"""
import cProfile, io, os, pstats, shutil, sys, tempfile, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcm_ts_amd.run_codec import _nets, encode_folder, save_torch_image
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
tmp = tempfile.mkdtemp(prefix="dcvc_files_")
try:
    src, dst = os.path.join(tmp, "png"), os.path.join(tmp, "bin")
    os.makedirs(src)
    g = torch.Generator().manual_seed(3)
    for t in range(n):
        save_torch_image(torch.rand(1, 3, 1080, 1920, generator=g), os.path.join(src, f"im{str(t + 1).zfill(5)}.png"))
    nets = _nets(torch.device("cuda:0"), "fp16x3")
    encode_folder(src, dst, None, 32, (1.0, 1.0, 1.0), "cuda:0", None, max_frames=3, nets=nets)
    shutil.rmtree(dst)
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    t0 = time.time()
    pr.enable()
    encode_folder(src, dst, None, 32, (1.0, 1.0, 1.0), "cuda:0", None, io_workers=workers, nets=nets)
    torch.cuda.synchronize()
    pr.disable()
    print(f"{n} frames, {workers} io workers: {(time.time() - t0) / n * 1e3:.1f} ms per frame")
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
    print(s.getvalue())
finally:
    shutil.rmtree(tmp, ignore_errors=True)
