"""Dev tool: where do the two matrix paths' input gradients differ?  Regresses the per-point difference of d logit / d y
on the float64 per-feature contributions, so a wrong table entry shows up as a coefficient on that feature."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pytorch-motion-planner_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import gpu_common as gc  # noqa: E402
from nfopp import _lib  # noqa: E402
from oracle import nfopp_oracle as orc  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "a"
z = np.load(os.path.join(ROOT, "tests", "golden", "g1_onf.npz"))
onf, cfg = gc.make_onf(z[tag + "_cfg"], z[tag + "_params"])
x = z[tag + "_x"]
rng = np.random.default_rng(70000)
xs = np.ascontiguousarray(x[rng.integers(0, len(x), 70000)], np.float32)
if os.environ.get("NFOPP_DEV_LIB"):
    _lib.LIB_PATH = os.environ["NFOPP_DEV_LIB"]
lib = _lib.load()
outs = []
for path in (0, 1):
    _lib.check(lib.nfopp_set_matrix_path(path))
    outs.append(onf.forward_with_grad(torch.tensor(xs, device="cuda")).cpu().numpy())
o0, o1 = outs
for c, name in enumerate(["logit", "gx", "gy", "gth"][:o0.shape[1]]):
    d = np.abs(o1[:, c] - o0[:, c])
    print(name, "max diff %.3e  rows wrong %d" % (d.max(), (d > 1e-3 * np.abs(o0[:, c]).max()).sum()))
d = np.abs(o1[:, 2] - o0[:, 2])
wrong = np.nonzero(d > 1e-4 * np.abs(o0[:, 2]).max())[0]
print("wrong rows:", len(wrong), "first", wrong[:40].tolist())
for m in (16, 32, 64, 128, 256):
    print("row %% %d histogram:" % m, np.bincount(wrong % m, minlength=m).tolist())
print("row // 4096 histogram:", np.bincount(wrong // 4096).tolist())
p = orc.unpack_params(np.asarray(z[tag + "_params"], np.float32), cfg)
p = {k: (np.asarray(v, np.float64) if v is not None else None) for k, v in p.items()}
X = xs.astype(np.float64)
e = X[:, :2] @ p["we"].T + (p["be"] if p["be"] is not None else 0)
n_sin = e.shape[1] if not cfg.use_cos else e.shape[1] // 2
feats = [np.sin(e[:, :n_sin])] + ([np.cos(e[:, n_sin:])] if cfg.use_cos else [])
dfe = [np.cos(e[:, :n_sin])] + ([-np.sin(e[:, n_sin:])] if cfg.use_cos else [])
fe, dfe = np.concatenate(feats, 1), np.concatenate(dfe, 1)
fin = fe
if cfg.angle_encoding:
    za = (X[:, 2:3] + p["ang_b"][None]) * p["ang_f"][None]
    fin = np.concatenate([fe, np.sin(za[:, :cfg.angle_dim]), np.cos(za[:, cfg.angle_dim:])], 1)
a1 = fin @ p["w1"].T + p["b1"]
h1 = np.maximum(a1, 0)
a2 = h1 @ p["w2"].T + p["b2"]
w3 = p["w3"].reshape(-1)
dh2 = w3[:100][None] * (a2 > 0)
dh1 = (dh2 @ p["w2"]) * (a1 > 0)
dfin = dh1 @ p["w1"] + w3[100:][None]
contrib_y = dfin[:, :fe.shape[1]] * dfe * p["we"][:, 1][None]     # per-feature share of d logit / d y
print("f64 gy vs path0 %.2e, vs path1 %.2e" % (np.abs(contrib_y.sum(1) - o0[:, 2]).max(), np.abs(contrib_y.sum(1) - o1[:, 2]).max()))
