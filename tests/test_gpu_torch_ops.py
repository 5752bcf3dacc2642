"""GPU: golden steps driven through torch.ops.nfopp.* (csrc/torch_ops.cpp, the PyTorch-ROCm extension form of the boundary)
instead of the ctypes binding: the same kernels behind TORCH_CHECK-validated tensor arguments on the current stream."""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden, max_abs

pytestmark = pytest.mark.gpu

gc = pytest.importorskip("gpu_common")
import nfopp  # noqa: E402
from nfopp import torch_ops  # noqa: E402
from oracle import nfopp_oracle as orc  # noqa: E402

F32 = np.float32


def _cfg_args(cfg):
    return (float(cfg.mean), float(cfg.sigma), bool(cfg.use_cos), bool(cfg.bias), 10 if cfg.angle_encoding else 0)


@pytest.mark.parametrize("tag", ["a", "c"])
def test_onf_ops_vs_golden(tag):
    ops = torch_ops.load()
    z = load_golden("g1_onf.npz")
    onf, cfg = gc.make_onf(z[tag + "_cfg"], z[tag + "_params"])
    x = torch.tensor(z[tag + "_x"], device="cuda")
    d = x.shape[1]
    out = ops.onf_fwd_bwd_input(onf.flat_parameters, x, *_cfg_args(cfg))
    lg = ops.onf_logits(onf.flat_parameters, x, *_cfg_args(cfg))
    torch.cuda.synchronize()
    assert gc.scaled_err(out[:, 0].cpu().numpy(), z[tag + "_logit"]) < 1e-5
    assert gc.scaled_err(out[:, 1:1 + d].cpu().numpy(), z[tag + "_grad"]) < 5e-5
    assert lg.shape == (len(x), 1) and torch.equal(lg[:, 0], out[:, 0])
    with pytest.raises(RuntimeError, match="must be \\[P, %d\\]" % d):
        ops.onf_fwd_bwd_input(onf.flat_parameters, x[:, :1].contiguous(), *_cfg_args(cfg))
    with pytest.raises(RuntimeError, match="contiguous"):
        ops.onf_fwd_bwd_input(onf.flat_parameters, x.t().contiguous().t(), *_cfg_args(cfg))


@pytest.mark.parametrize("name", ["traj_n100_default.npz", "traj_n100_hard.npz"])
def test_traj_step_and_reparametrize_ops_vs_golden(name):
    ops = torch_ops.load()
    z = load_golden(name)
    onf, cfg = gc.make_onf(z["cfg"], z["params"])
    hp = gc.hyper_from(orc.Hyper.from_npz(z))
    s = gc.state_of(z, "s0_")
    eng = gc.engine_from_state(onf, s, hp)     # buffers + band only: the step below goes through torch.ops
    eng.t.copy_(torch.tensor(z["g3_t"][None]))
    eng.adam_step += 1
    hyper = torch_ops.hyper_list(hp.to_c(eng.adam_step))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):              # the ops take the CURRENT stream of the tensors' device
        ops.traj_step(onf.flat_parameters, *_cfg_args(cfg), eng.traj, eng.start, eng.goal, eng.lam, eng.cm, eng.adam_m,
                      eng.adam_v, eng.t, 0, 0, 0, 0, eng.onf_out, eng.hinv_band, eng.half_width, eng.interior[0],
                      eng.interior[1], hyper, eng.terms, None, None)
    side.synchronize()
    assert max_abs(eng.traj.cpu().numpy()[0], z["g3_traj"]) < 2e-6
    assert max_abs(eng.lam.cpu().numpy()[0], z["g3_lam"]) < 2e-6
    assert max_abs(eng.cm.cpu().numpy()[0], z["g3_cm"]) < 1e-6
    assert gc.scaled_err(eng.adam_m.cpu().numpy()[0], z["g3_adam_m"]) < 1e-5
    # the same reparametrisation through both bindings: bit-identical
    ref = gc.engine_from_state(onf, gc.state_of(z, "g3_"), hp)
    ref.reparametrize()
    eng2 = gc.engine_from_state(onf, gc.state_of(z, "g3_"), hp)
    ops.reparametrize(eng2.traj, eng2.start, eng2.goal, eng2.lam, eng2.cm, eng2.u, None)
    torch.cuda.synchronize()
    assert torch.equal(eng2.traj, ref.traj) and torch.equal(eng2.lam, ref.lam) and torch.equal(eng2.cm, ref.cm)
    with pytest.raises(RuntimeError, match="hyper must hold the 18 floats"):
        ops.traj_step(onf.flat_parameters, *_cfg_args(cfg), eng.traj, eng.start, eng.goal, eng.lam, eng.cm, eng.adam_m,
                      eng.adam_v, eng.t, 0, 0, 0, 0, eng.onf_out, eng.hinv_band, eng.half_width, 0, 0, hyper[:5], None, None, None)


def test_onf_train_step_op_vs_golden():
    ops = torch_ops.load()
    z = load_golden("g7_onf_train.npz")
    onf, cfg = gc.make_onf(z["cfg"], z["params_before"])
    x, y = torch.tensor(z["x"].astype(F32), device="cuda"), torch.tensor(z["labels"].astype(F32), device="cuda")
    params = onf.flat_parameters.clone()
    m, v = torch.tensor(z["adam_m_before"], device="cuda"), torch.tensor(z["adam_v_before"], device="cuda")
    lr, b1, b2, eps = float(z["lr"]), float(z["beta1"]), float(z["beta2"]), float(z["eps"])
    step = int(z["adam_step_before"]) + 1
    grad = ops.onf_train_step(params, m, v, x, y, *_cfg_args(cfg), b2, 1 - b1, 1 - b2, eps, lr / (1 - b1 ** step),
                              math.sqrt(1 - b2 ** step))
    torch.cuda.synchronize()
    n = params.numel()
    g = grad.cpu().numpy()
    assert abs(float(g[n]) - float(z["loss"])) < 2e-6 and g[n + 1] == len(x)
    assert max_abs(g[:n], z["grad"]) < 3e-6 * max(1.0, float(np.abs(z["grad"]).max()))
    # Adam ran on the kernel's own gradient (the ctypes test feeds the golden one): parameters agree to the gradient's rounding
    assert max_abs(params.cpu().numpy(), z["params_after"]) < 2e-5
    assert max_abs(m.cpu().numpy(), z["adam_m_after"]) < 1e-6
