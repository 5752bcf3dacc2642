"""GPU debug: where does the 32x32 kernel differ from the fp32 path for the all-dimensions test field?"""
import sys
import numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, "pytorch-motion-planner_amd"); sys.path.insert(0, ".")
import nfopp
from nfopp import _lib
import os
if os.environ.get("DBG_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["DBG_LIB"])
    _lib.ABI_VERSION = None
lib = __import__("ctypes").CDLL(_lib.LIB_PATH)
for name, (res, args) in _lib._SIGNATURES.items():
    if hasattr(lib, name):
        fn = getattr(lib, name); fn.restype, fn.argtypes = res, args
_lib._lib = lib
print("library:", _lib.LIB_PATH, flush=True)
F32 = np.float32
torch.random.manual_seed(11)
onf = nfopp.ONF(0.4, 2.5, use_cos=True, use_normal_init=True, bias=True, angle_encoding=True).to("cuda")
rng = np.random.default_rng(3)
for n in (1, 33, 300, 4099):
    x = rng.uniform(-4, 6, (n, 3)).astype(F32); x[:, 2] = rng.uniform(-3.3, 3.3, n)
for n in (70001, 65536, 65537, 131072, 131072, 131072, 1044480):
    x = rng.uniform(-4, 6, (n, 3)).astype(F32); x[:, 2] = rng.uniform(-3.3, 3.3, n)
    xt = torch.tensor(x, device="cuda")
    outs = {}
    for path in (0, 1):
        _lib.check(lib.nfopp_set_matrix_path(path))
        outs[path] = onf.forward_with_grad(xt).cpu().numpy()
    d = np.abs(outs[1] - outs[0])
    scale = np.abs(outs[0]).max(0)
    bad = np.flatnonzero((d / scale).max(1) > 1e-4)
    print(n, "cols max rel", (d / scale).max(0), "bad rows", len(bad), bad[:12], "chunk of first", bad[:3] // 256 if len(bad) else None, flush=True)
    if len(bad):
        r = bad[0]
        print("  row", r, "x32", outs[1][r], "fp32", outs[0][r], "lane", r % 256)
        print("  bad rows mod 256 histogram (first 20 distinct):", np.unique(bad % 256)[:40])
        print("  bad rows // 256 distinct:", np.unique(bad // 256)[:40])
