"""Dev tool: print the actual parity errors (HIP vs golden) so tolerances/margins are visible."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "pytorch-motion-planner_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import gpu_common as gc
from oracle import nfopp_oracle as orc
z = np.load(os.path.join(ROOT, "tests/golden/g1_onf.npz"))
for tag in "abc":
    onf, cfg = gc.make_onf(z[tag + "_cfg"], z[tag + "_params"])
    x = z[tag + "_x"]
    out = onf.forward_with_grad(torch.tensor(x, device="cuda")).cpu().numpy()
    d = x.shape[1]
    lo, go = orc.onf_forward_grad(z[tag + "_params"], cfg, x)
    print("G1 %s: logit vs golden %.2e  grad vs golden %.2e | oracle vs golden %.2e %.2e" % (
        tag, gc.scaled_err(out[:, 0], z[tag + "_logit"]), gc.scaled_err(out[:, 1:1 + d], z[tag + "_grad"]),
        gc.scaled_err(lo, z[tag + "_logit"]), gc.scaled_err(go, z[tag + "_grad"])))
for name in ("traj_n100_default.npz", "traj_n100_hard.npz", "traj_n256_default.npz"):
    z = np.load(os.path.join(ROOT, "tests/golden", name))
    onf, cfg = gc.make_onf(z["cfg"], z["params"])
    hp = orc.Hyper.from_npz(z)
    eng = gc.engine_from_state(onf, gc.state_of(z, "s0_"), hp)
    eng.optimize_trajectory(z["g3_t"][None])
    print("G3 %s: traj %.2e lam %.2e cm %.2e m %.2e" % (name, np.abs(eng.traj.cpu().numpy()[0] - z["g3_traj"]).max(),
          np.abs(eng.lam.cpu().numpy()[0] - z["g3_lam"]).max(), np.abs(eng.cm.cpu().numpy()[0] - z["g3_cm"]).max(),
          gc.scaled_err(eng.adam_m.cpu().numpy()[0], z["g3_adam_m"])))
